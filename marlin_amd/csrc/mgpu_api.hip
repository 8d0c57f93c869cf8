#include <stdio.h>
#include <stdlib.h>
// mgpu_api.hip -- host runtime behind the C ABI of include/marlin_gpu.h.
//
// Mirror construction (reference formats -> HBM layouts), per-batch plan lowering, arena
// layout, launches on one HIP stream, result decode.  No torch, no CPU compute fallback:
// every documented failure returns an MGPU_E_* code.
#include <time.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <memory>
#include <mutex>
#include <vector>
#include <map>
#include <tuple>
#include "mgpu_types.h"
#include "mg_lev.h"

typedef unsigned long long u64;

extern "C" {
void mgk_scatter(hipStream_t st, const ScatterItem *items, uint32_t nitems);
void mgk_range_rows(hipStream_t st, const RangeRow *rows, uint32_t nrows, uint32_t max_w);
void mgk_formula(hipStream_t st, const SqDev *sqs, const uint32_t *blob, uint32_t max_nblk, uint32_t nsq, uint32_t nmain,
                 uint2 *tile_list, uint32_t *tile_count, uint32_t tile_cap, uint32_t *fh_list, uint32_t *fh_count, uint32_t fh_cap);
void mgk_rank(hipStream_t st, const SqDev *sqs, const uint32_t *blob, uint32_t max_n, uint32_t nsq,
              uint32_t max_tab_slots, int with_len, const uint2 *tile_list, const uint32_t *tile_count, uint32_t tile_cap,
              unsigned long long *pool, uint32_t *pool_used, uint64_t pool_cap);
void mgk_docproc(hipStream_t st, const SqDev *sqs, uint32_t nsq, const uint2 *tile_list, const uint32_t *tile_count,
                 uint32_t tile_cap, uint32_t lds_bins, int pre);
void mgk_fast_build(hipStream_t st, const SqDev *sqs, uint32_t nsq, const uint32_t *elig_list, uint32_t nelig,
                    uint2 *tile_list_b, uint32_t *tile_count_b);
void mgk_facet_recount(hipStream_t st, const SqDev *sqs, const void *jobs, uint32_t njobs, uint32_t max_k);
void mgk_fkeys(hipStream_t st, const void *jobs, uint32_t njobs, uint32_t max_n);
void mgk_card(hipStream_t st, const void *jobs, uint32_t njobs, uint32_t max_n);
void mgk_check_zero(hipStream_t st, const void *p, size_t bytes, uint32_t *flag);
void mgk_select(hipStream_t st, const SelJob *jobs, const void *outs, uint32_t njobs, uint32_t max_m, int passes, int finals,
                void *items_mem, uint32_t cap_items, uint32_t *counters, uint32_t *fin_list);
void mgk_row_op(hipStream_t st, const unsigned long long *a, const unsigned long long *b, unsigned long long *out,
                uint32_t W, int op, unsigned long long *card);
void mgk_scatter_list(hipStream_t st, const uint32_t *src, uint64_t n, unsigned long long *dst);
void mgk_fetch(hipStream_t st, void *dst, const void *src_pinned, size_t bytes);
void mgk_facet_hash(hipStream_t st, const SelJob *jobs, const void *outs, const uint32_t *list, const uint32_t *count, uint32_t cap);
void mgk_merge_postings(hipStream_t st, uint32_t num_postings, const uint32_t *old_pid, const int32_t *new_dense, const unsigned long long *new_off,
                        const int32_t *old_dense, const unsigned long long *old_off, const uint32_t *old_doc,
                        const unsigned long long *delta_off, const uint32_t *delta_doc, uint32_t *new_doc,
                        const uint32_t *winbase, uint32_t nwin, uint32_t *post_win);
void mgk_hl_match(hipStream_t st, const uint32_t *tok_chars, const uint32_t *tok_off, const uint32_t *tok_job, uint32_t ntok,
                  const void *terms, const uint32_t *term_chars, const uint32_t *job_term_off, int32_t *out);
void mgk_term_scan(hipStream_t st, const TermJob *jobs, const TermDict *dicts, uint32_t ndicts, uint32_t max_words, const uint32_t *term_chars,
                   uint2 *cand, uint32_t cand_cap, uint32_t *hits, uint32_t *counts, uint32_t cap);
}

// Scratch of mgpu_posting_op_noalloc (include/marlin_gpu.h): two operand rows, the result row and the cardinality
// word, in device memory that comes with the code object instead of from hipMalloc.  In host code this name is
// only a handle: the device address is looked up once per device (noalloc_scratch_of) and every use below goes
// through that pointer.  Each GPU the library runs on holds its own copy until the process ends.
#define NA_ROW_BYTES ((size_t)MGPU_NOALLOC_MAX_DOCS / 8)
#define NA_SCRATCH_BYTES (3 * NA_ROW_BYTES + 256)
static __device__ __align__(256) unsigned char g_noalloc_scratch[NA_SCRATCH_BYTES];
#define NA_MAX_DEVICES 64
static char *g_na_ptr[NA_MAX_DEVICES];

struct FkJob { const uint32_t *hist; const uint32_t *ord2id; const uint32_t *touched; const uint32_t *ntouched;
               u64 *key_hi, *key_lo; u64 *andor; };
struct CardJob { const uint32_t *ord2id; const uint32_t *touched; const uint32_t *ntouched; uint32_t *out_ids; uint32_t *out_n;
                 uint32_t *clean_hist; };   // histogram to put back to zero (fields no facet job consumes), or NULL
struct SelOut { u64 *out_hi, *out_lo; uint32_t *out_pl; const uint32_t *payload; };
struct RecountJob { uint32_t sq; uint32_t field; u64 *out_hi; const u64 *out_lo; const uint32_t *out_n; };

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    return fail(MGPU_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)

extern "C" const char *mgpu_last_error(void) { return g_err.c_str(); }
extern "C" int mgpu_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

// ------------------------------------------------------------------ mirror
struct mgpu_shard {
    int device;
    ShardDev dev;                       // device pointers
    std::vector<void *> allocs;
    uint64_t bytes;
    // host-side facts used when lowering plans
    std::vector<int32_t> post_dense;    // per pid
    std::vector<uint64_t> post_off;     // sparse CSR offsets (per pid)
    std::vector<uint32_t> post_card;    // docs per posting (bounds the hit count of a plan)
    std::vector<uint32_t> post_winbase; // per pid: base into post_win, or ~0 (dense, empty or short posting)
    bool rec_len_ok = false;            // the rank kernel may carry a record's length in the top byte of its docid word
    uint32_t fac_single = 0;            // bit f: every document has <= 1 value in facet field f, and every facet section is staged whole by the rank kernel
    uint64_t np_n[MGPU_MAX_FIELDS];
    uint32_t np_multi = 0;              // bit i: some document appears more than once in num_dbi[i] (a list-valued field)
    double avg_dw;                      // mean docword entries per doc
    double avg_fv[MGPU_MAX_FIELDS];     // mean facet values per doc
    uint32_t n_dense;
    uint32_t rec_stride;
    // ---- growth (mgpu_shard_apply_delta, row f-2): the arrays new documents append to are allocated with headroom
    uint32_t cap_docs = 0;              // docids the per-document arrays and the bitmap rows have room for (N <= cap_docs)
    uint64_t n_dw = 0, cap_dw = 0, n_pp = 0, cap_pp = 0;          // docword entries, decoded positions (and the pool copy behind the records)
    uint64_t n_fa[MGPU_MAX_FIELDS] = {0}, cap_fa[MGPU_MAX_FIELDS] = {0};   // facet values per field
    uint32_t cap_ord[MGPU_MAX_FIELDS] = {0};                      // ordinals per field (fa_ord2id)
    uint32_t rec_hdr = 0, rec_FA = 0, rec_nnum_in = 0;            // the record layout, fixed when the mirror is built
    uint64_t rec_pool_at = 0;                                     // u32 index in rec where the position pool copy starts
    bool all_staged = false;
    uint64_t n_sparse = 0;                                        // entries in post_doc
    std::vector<uint32_t> fa_sorted_id[MGPU_MAX_FIELDS], fa_sorted_ord[MGPU_MAX_FIELDS];   // facet ids ascending -> their ordinals
};
// fraction of headroom a mirror is built with (MGPU_MIRROR_HEADROOM, default 0.125; 0 = none: every delta is refused)
static double mirror_headroom() {
    const char *e = getenv("MGPU_MIRROR_HEADROOM");               // read when a mirror is built
    const double v = e ? atof(e) : 0.125;
    return v < 0 ? 0.0 : v > 4 ? 4.0 : v;
}
static uint64_t with_headroom(uint64_t n, uint64_t slack) { const double h = mirror_headroom(); return h > 0 ? n + (uint64_t)((double)n * h) + slack : n; }

// the same with room for `cap` elements (cap >= v.size()); the tail is zeroed
template <typename T>
static int upload_cap(mgpu_shard *s, const std::vector<T> &v, size_t cap, const T **out) {
    if (cap < v.size()) cap = v.size();
    size_t bytes = std::max<size_t>(cap, 1) * sizeof(T);
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    s->bytes += bytes;
    if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    if (cap > v.size()) HIPCHK(hipMemset((char *)p + v.size() * sizeof(T), 0, (cap - v.size()) * sizeof(T)));
    *out = (const T *)p;
    return MGPU_OK;
}
template <typename T>
static int upload(mgpu_shard *s, const std::vector<T> &v, const T **out) {
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    s->bytes += bytes;
    if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T *)p;
    return MGPU_OK;
}

static inline const uint8_t *rd_vint(const uint8_t *buf, int *value) {   // read_vint, sindex.c:340-351
    uint8_t b = *buf++;
    int i = b & 0x7F;
    for (int shift = 7; (b & 0x80) != 0; shift += 7) { b = *buf++; i |= (b & 0x7F) << shift; }
    *value = i;
    return buf;
}

// ------------------------------------------------------------------ mbmap format (mbmap.c:183-283, cont.c)
static inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
// walks one stored posting; emit(docid) in ascending order.  Returns MGPU_OK / MGPU_E_ARG.
template <class F>
static int mbmap_walk(const mgpu_mbmap_t *b, F emit, uint64_t *count) {
    *count = 0;
    if (!b || !b->header) return MGPU_OK;                    // absent key: mbmap_load_bmap returns NULL, an empty set
    if (b->header_len < 2) return fail(MGPU_E_ARG, "mbmap header shorter than its count");
    const uint32_t num_c = rd16(b->header);
    if ((uint64_t)b->header_len < 2 + 2 * (uint64_t)num_c) return fail(MGPU_E_ARG, "mbmap header shorter than its container list");
    if (num_c && (!b->cont || !b->cont_len)) return fail(MGPU_E_ARG, "mbmap containers missing");
    int64_t prev_cid = -1;
    uint64_t total = 0;
    for (uint32_t i = 0; i < num_c; i++) {
        const uint32_t cid = rd16(b->header + 2 + 2 * i);
        if ((int64_t)cid <= prev_cid) return fail(MGPU_E_ARG, "mbmap container ids not ascending");
        prev_cid = cid;
        const uint8_t *c = b->cont[i];
        if (!c) continue;                                    // mdb_get failed in the reference: container skipped
        const uint32_t len = b->cont_len[i];
        if (len < 4) return fail(MGPU_E_ARG, "mbmap container shorter than its header");
        if (rd16(c) != cid) return fail(MGPU_E_ARG, "mbmap container id does not match its key");
        uint32_t card = rd16(c + 2);
        const uint32_t base = cid << 16;
        // cont_cardinality (cont.c:543-548): card 0 with a first payload word of 0xFFFF is the full container
        if (card == 0 && len >= 6 && rd16(c + 4) == 0xFFFF) card = 65536;
        if (card <= 4096) {                                  // array container (cont_is_array, cont.c:118-123)
            if ((uint64_t)len < 4 + 2 * (uint64_t)card) return fail(MGPU_E_ARG, "mbmap array container shorter than its cardinality");
            int32_t prev = -1;
            for (uint32_t j = 0; j < card; j++) {
                const uint32_t v = rd16(c + 4 + 2 * j);
                if ((int32_t)v <= prev) return fail(MGPU_E_ARG, "mbmap array container not ascending");
                prev = (int32_t)v;
                emit(base + v);
            }
            total += card;
        } else {                                             // bitset container: 1024 u64 words from byte 4 (unaligned)
            if (len < 4 + 8192) return fail(MGPU_E_ARG, "mbmap bitset container shorter than 8192 bytes");
            uint64_t seen = 0;
            for (uint32_t w = 0; w < 1024; w++) {
                uint64_t x; memcpy(&x, c + 4 + 8 * (size_t)w, 8);
                seen += (uint64_t)__builtin_popcountll(x);
                while (x) { const int r = __builtin_ctzll(x); x &= x - 1; emit(base + 64 * w + (uint32_t)r); }
            }
            if (seen != card) return fail(MGPU_E_ARG, "mbmap bitset container cardinality does not match its bits");
            total += seen;
        }
    }
    *count = total;
    return MGPU_OK;
}

extern "C" int mgpu_mbmap_decode(const mgpu_mbmap_t *b, uint32_t *out, uint64_t cap, uint64_t *n) {
    uint64_t at = 0, cnt = 0;
    const int rc = mbmap_walk(b, [&](uint32_t d) { if (at < cap && out) out[at] = d; at++; }, &cnt);
    if (n) *n = rc ? 0 : cnt;
    if (rc) return rc;
    if (cnt > cap) return fail(MGPU_E_LIMIT, "mgpu_mbmap_decode: output buffer smaller than the posting");
    return MGPU_OK;
}

extern "C" void mgpu_shard_destroy(mgpu_shard_t *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void *p : s->allocs) (void)hipFree(p);
    delete s;
}

extern "C" uint64_t mgpu_shard_bytes(const mgpu_shard_t *s) { return s ? s->bytes : 0; }

// ---- what the docid2data blobs of a run of documents decode to (sindex.c:273-326, 409-496).  Offsets are relative to
// the run: document i of the run owns dw[dw_off[i] .. dw_off[i + 1]) and so on.
struct DocRun {
    std::vector<uint32_t> dw_off, pp_off;                  // [ndocs + 1]
    std::vector<uint2> dw;                                 // {wid, info}
    std::vector<uint32_t> pp;                              // decoded positions: (priority << 16) + position
    std::vector<std::vector<double>> cols;                 // [num_numbers][ndocs]
    std::vector<std::vector<uint32_t>> fa_off, fa_ids;     // [num_facets][ndocs + 1], facet ids
};
// blob of document i of the run = dd[dd_off[i], dd_off[i + 1])
static int decode_docs(const uint8_t *dd, const uint64_t *dd_off, uint32_t ndocs, int nn, int nf, DocRun &R) {
    R.dw_off.assign((size_t)ndocs + 1, 0); R.pp_off.assign((size_t)ndocs + 1, 0);
    R.cols.assign(nn, std::vector<double>(ndocs, 0.0));
    R.fa_off.assign(nf, std::vector<uint32_t>((size_t)ndocs + 1, 0));
    R.fa_ids.assign(nf, std::vector<uint32_t>());
    for (uint32_t i = 0; i < ndocs; i++) {
        R.dw_off[i] = (uint32_t)R.dw.size(); R.pp_off[i] = (uint32_t)R.pp.size();
        for (int f = 0; f < nf; f++) R.fa_off[f][i] = (uint32_t)R.fa_ids[f].size();
        const uint8_t *blob = dd + dd_off[i];
        const uint64_t bsz = dd_off[i + 1] - dd_off[i];
        if (bsz < 4) continue;
        uint32_t off; memcpy(&off, blob, 4);
        for (int k = 0; k < nn; k++) memcpy(&R.cols[k][i], blob + 4 + (size_t)k * 8, 8);
        const uint8_t *fpos = blob + 4 + (size_t)nn * 8;
        for (int f = 0; f < nf; f++) {
            uint32_t cnt; memcpy(&cnt, fpos, 4);
            for (uint32_t j = 0; j < cnt; j++) { uint32_t id; memcpy(&id, fpos + 4 + (size_t)j * 4, 4); R.fa_ids[f].push_back(id); }
            fpos += 4 + (size_t)cnt * 4;
        }
        if ((uint64_t)off + 2 > bsz) continue;
        const uint8_t *wp = blob + off;
        uint16_t numwid; memcpy(&numwid, wp, 2);
        const uint32_t pool_base = (uint32_t)R.pp.size();
        for (uint32_t w = 0; w < numwid; w++) {
            uint32_t wid, info; memcpy(&wid, wp + 2 + (size_t)w * 8, 4); memcpy(&info, wp + 6 + (size_t)w * 8, 4);
            if (info & 1u) { R.dw.push_back(make_uint2(wid, info)); continue; }
            // fill_positions (docrank.c:21-45): freq, then {priority, count, vints, 0xFF} groups
            const uint8_t *c = wp + (info >> 9);
            int freq = *c++;
            const uint32_t start = (uint32_t)R.pp.size() - pool_base;
            uint32_t n = 0;
            while (freq > 0) {
                int priority = *c++; int count = *c++;
                freq -= count;
                while (count > 0) { int position; c = rd_vint(c, &position); R.pp.push_back(((uint32_t)priority << 16) + (uint32_t)position); n++; count--; }
                c++;
            }
            if (n > 255 || start >= (1u << 23)) return fail(MGPU_E_LIMIT, "word frequency or document too large for the mirror");
            R.dw.push_back(make_uint2(wid, (n << 1) | (start << 9)));
        }
    }
    R.dw_off[ndocs] = (uint32_t)R.dw.size(); R.pp_off[ndocs] = (uint32_t)R.pp.size();
    for (int f = 0; f < nf; f++) R.fa_off[f][ndocs] = (uint32_t)R.fa_ids[f].size();
    return MGPU_OK;
}
// The records of a run of documents (layout: ShardDev::rec), document i of the run = docid first_doc + i.  base_dw / base_pp:
// entries of the shard's docword / position arrays before the run.  fa_ords[f] = the run's facet values as ordinals.
// rec: ndocs * stride / 4 words, zeroed; rlen: ndocs bytes.  Clears s->all_staged when a document's facet section is
// not staged whole by the rank kernel.
static void build_records(mgpu_shard *s, const DocRun &R, const std::vector<std::vector<uint32_t>> &fa_ords, uint32_t first_doc,
                          uint64_t base_dw, uint64_t base_pp, uint32_t *rec, uint8_t *rlen) {
    const uint32_t S = s->rec_stride, hdr = s->rec_hdr, FA = s->rec_FA, ws_c = (hdr + FA) / 8;
    const int nnum_in = (int)s->rec_nnum_in, nf = (int)R.fa_off.size();
    const uint32_t ndocs = (uint32_t)R.dw_off.size() - 1;
    const bool pool_in = !getenv("MGPU_NO_POOLIN");                  // measurement switch, read when records are built
    for (uint32_t i = 0; i < ndocs; i++) {
        const uint32_t doc = first_doc + i;
        uint32_t *r = rec + (size_t)i * (S / 4);
        const uint32_t nwords = R.dw_off[i + 1] - R.dw_off[i];
        uint32_t fb = 0;
        for (int f = 0; f < nf; f++) fb += 4 + 4 * (R.fa_off[f][i + 1] - R.fa_off[f][i]);
        fb = (fb + 7) & ~7u;
        const bool facets_in = nf > 0 && fb <= FA;
        if (!facets_in || fb > 4 * MG_PCAP) s->all_staged = false;
        uint32_t nin = (S - 8 * ws_c) / 8;
        if (nin > nwords) nin = nwords;
        uint32_t used = 8 * (ws_c + nin);                     // bytes
        r[0] = (nwords > 0xFFFF ? 0xFFFFu : nwords) | (nin << 16);
        r[1] = (uint32_t)(s->rec_pool_at + base_pp + R.pp_off[i]);
        r[2] = (uint32_t)(base_dw + R.dw_off[i] + nin);
        r[3] = (uint32_t)nnum_in | ((facets_in ? 1u : 0u) << 8) | (ws_c << 16);
        for (int k = 0; k < nnum_in; k++) memcpy(&r[4 + 2 * k], &R.cols[k][i], 8);
        if (facets_in) {
            uint32_t *fp = r + hdr / 4;
            for (int f = 0; f < nf; f++) {
                const uint32_t a = R.fa_off[f][i], b = R.fa_off[f][i + 1];
                *fp++ = b - a;
                for (uint32_t j = a; j < b; j++) *fp++ = fa_ords[f][j];
            }
        }
        for (uint32_t j = 0; j < nin; j++) { r[2 * (ws_c + j)] = R.dw[R.dw_off[i] + j].x; r[2 * (ws_c + j) + 1] = R.dw[R.dw_off[i] + j].y; }
        // the document's decoded positions right behind its docwords when everything fits: the rank kernel's
        // position reads then land in the lines it has just fetched (the lanes that hold them do not take them
        // for docword entries: the header's entry count says where the entries end)
        const uint32_t npos = R.pp_off[i + 1] - R.pp_off[i];
        if (pool_in && nin == nwords && npos > 0 && used + 4 * npos <= S) {
            for (uint32_t j = 0; j < npos; j++) r[used / 4 + j] = R.pp[R.pp_off[i] + j];
            r[1] = (uint32_t)((size_t)doc * (S / 4) + used / 4);
            used += 4 * npos;
        }
        rlen[i] = (uint8_t)((used + 15) / 16);
    }
}

static int set_num_order(mgpu_shard *s, int i, const uint32_t *doc, uint64_t n, const uint32_t **np_out, const uint64_t **grp_out);
static int shard_create_impl(int device, const mgpu_shard_desc_t *d, mgpu_shard_t **out);
extern "C" int mgpu_shard_create(int device, const mgpu_shard_desc_t *d, mgpu_shard_t **out) {
    if (!d || !out) return fail(MGPU_E_ARG, "null argument");
    if (!d->post_mbmap) return shard_create_impl(device, d, out);
    // postings handed over as stored (mbmap header + containers): expand them here (mbmap_load_bmap + bmap_iterate)
    std::vector<uint64_t> off((size_t)d->num_postings + 1, 0);
    std::vector<uint32_t> doc;
    for (uint32_t p = 0; p < d->num_postings; p++) {
        off[p] = doc.size();
        uint64_t cnt = 0;
        const int rc = mbmap_walk(&d->post_mbmap[p], [&](uint32_t x) { doc.push_back(x); }, &cnt);
        if (rc) return rc;
    }
    off[d->num_postings] = doc.size();
    if (doc.empty()) doc.push_back(0);
    mgpu_shard_desc_t d2 = *d;
    d2.post_mbmap = nullptr; d2.post_off = off.data(); d2.post_doc = doc.data();
    return shard_create_impl(device, &d2, out);
}
static int shard_create_impl(int device, const mgpu_shard_desc_t *d, mgpu_shard_t **out) {
    if (d->num_numbers > MGPU_MAX_FIELDS || d->num_facets > MGPU_MAX_FIELDS)
        return fail(MGPU_E_LIMIT, "more than 32 numeric or facet fields");
    HIPCHK(hipSetDevice(device));
    mgpu_shard *s = new mgpu_shard();
    s->device = device; s->bytes = 0; s->n_dense = 0;
    memset(&s->dev, 0, sizeof(s->dev));
    ShardDev &sd = s->dev;
    const uint32_t N = d->max_docid + 1;
    s->cap_docs = (uint32_t)std::min<uint64_t>(with_headroom(N, 4096), 0xFFFFFFF0ull);
    // W counts the words that hold documents; the row stride leaves room for cap_docs (rows grow in place)
    sd.N = N; sd.W = (N + 63) / 64; sd.Wp = ((s->cap_docs + 63) / 64 + 31) & ~31u;
    sd.num_postings = d->num_postings;
    sd.num_facets = d->num_facets; sd.num_numbers = d->num_numbers;
    int rc = MGPU_OK;

    // ---- postings: dense rows for cardinality >= N/32 (a row costs no more than the list), pid 0 always
    {
        s->post_dense.assign(d->num_postings, -1);
        s->post_off.assign((size_t)d->num_postings + 1, 0);
        s->post_card.assign(d->num_postings, 0);
        for (uint32_t p = 0; p < d->num_postings; p++) s->post_card[p] = (uint32_t)std::min<uint64_t>(d->post_off[p + 1] - d->post_off[p], N);
        std::vector<uint32_t> sparse_doc;
        std::vector<uint32_t> dense_pids;
        for (uint32_t p = 0; p < d->num_postings; p++) {
            uint64_t n = d->post_off[p + 1] - d->post_off[p];
            bool dense = (p == 0) || (n > 0 && n * 32 >= N);
            s->post_off[p] = sparse_doc.size();
            if (dense) { s->post_dense[p] = (int32_t)dense_pids.size(); dense_pids.push_back(p); }
            else sparse_doc.insert(sparse_doc.end(), d->post_doc + d->post_off[p], d->post_doc + d->post_off[p + 1]);
        }
        s->post_off[d->num_postings] = sparse_doc.size();
        s->n_sparse = sparse_doc.size();
        if (sparse_doc.size() >= 0xFFFFFFFFull) { delete s; return fail(MGPU_E_LIMIT, "more than 2^32 sparse posting entries in one shard"); }
        for (uint32_t doc : sparse_doc) if (doc >= N) { delete s; return fail(MGPU_E_ARG, "posting docid beyond max_docid"); }
        // window index of the long sparse postings (see ShardDev::post_win)
        std::vector<uint32_t> post_win;
        s->post_winbase.assign(d->num_postings, 0xFFFFFFFFu);
        {
            const uint32_t nwin = (sd.W + MG_FORMULA_WORDS - 1) / MG_FORMULA_WORDS;
            for (uint32_t p = 0; p < d->num_postings; p++) {
                if (s->post_dense[p] >= 0) continue;
                const uint64_t b = s->post_off[p], n = s->post_off[p + 1] - b;
                if (n <= MG_SHORT_POSTING) continue;
                if (post_win.size() + nwin + 1 >= 0xFFFFFFFFull) { delete s; return fail(MGPU_E_LIMIT, "posting window index too large"); }
                s->post_winbase[p] = (uint32_t)post_win.size();
                uint64_t at = 0;
                for (uint32_t j = 0; j <= nwin; j++) {
                    const uint64_t lim = (uint64_t)j * 64 * MG_FORMULA_WORDS;
                    while (at < n && sparse_doc[b + at] < lim) {
                        if (at && sparse_doc[b + at] < sparse_doc[b + at - 1]) { delete s; return fail(MGPU_E_ARG, "posting docids not ascending"); }
                        at++;
                    }
                    post_win.push_back(j == nwin ? (uint32_t)n : (uint32_t)at);
                }
            }
        }
        std::vector<u64> rows((size_t)dense_pids.size() * sd.Wp, 0);
        for (size_t r = 0; r < dense_pids.size(); r++) {
            uint32_t p = dense_pids[r];
            u64 *row = rows.data() + r * sd.Wp;
            for (uint64_t i = d->post_off[p]; i < d->post_off[p + 1]; i++) {
                uint32_t doc = d->post_doc[i];
                if (doc >= N) { delete s; return fail(MGPU_E_ARG, "posting docid beyond max_docid"); }
                row[doc >> 6] |= 1ull << (doc & 63);
            }
        }
        s->n_dense = (uint32_t)dense_pids.size();
        const u64 *p_rows; const int32_t *p_pd; const u64 *p_off; const uint32_t *p_doc;
        if ((rc = upload(s, rows, &p_rows)) || (rc = upload(s, s->post_dense, &p_pd)) ||
            (rc = upload(s, s->post_off, (const uint64_t **)&p_off)) || (rc = upload(s, sparse_doc, &p_doc)) ||
            (rc = upload(s, post_win, &sd.post_win))) {
            mgpu_shard_destroy(s); return rc;
        }
        sd.dense = (const uint64_t *)p_rows; sd.post_dense = p_pd; sd.post_off = (const uint64_t *)p_off; sd.post_doc = p_doc;
    }

    // ---- docid2data blobs -> docword CSR + position pool + facet ordinals + numeric columns + doc-major records
    {
        DocRun R;
        if ((rc = decode_docs(d->dd, d->dd_off + 1, N > 1 ? N - 1 : 0, d->num_numbers, d->num_facets, R))) { mgpu_shard_destroy(s); return rc; }
        const uint32_t nd = N > 1 ? N - 1 : 0;
        // per-docid arrays: entry 0 belongs to docid 0, which no document has
        std::vector<uint32_t> dw_off(N + 1, 0), pp_off(N + 1, 0);
        for (uint32_t i = 0; i <= nd; i++) { dw_off[1 + i] = R.dw_off[i]; pp_off[1 + i] = R.pp_off[i]; }
        if (!nd) { dw_off[N] = 0; pp_off[N] = 0; }
        s->avg_dw = N > 1 ? (double)R.dw.size() / (double)(N - 1) : 0.0;
        s->n_dw = R.dw.size(); s->n_pp = R.pp.size();
        s->cap_dw = with_headroom(s->n_dw, 65536); s->cap_pp = with_headroom(s->n_pp, 65536);
        sd.dw_len = (uint32_t)R.dw.size(); sd.pp_len = (uint32_t)R.pp.size();
        const size_t capN = (size_t)s->cap_docs + 1;
        if ((rc = upload_cap(s, dw_off, capN, &sd.dw_off)) || (rc = upload_cap(s, R.dw, s->cap_dw, &sd.dw)) ||
            (rc = upload_cap(s, pp_off, capN, &sd.pp_off)) || (rc = upload_cap(s, R.pp, s->cap_pp, &sd.pp))) { mgpu_shard_destroy(s); return rc; }
        for (int i = 0; i < d->num_numbers; i++) {
            std::vector<double> col(N, 0.0);
            if (nd) memcpy(col.data() + 1, R.cols[i].data(), (size_t)nd * sizeof(double));
            if ((rc = upload_cap(s, col, capN, &sd.num_col[i]))) { mgpu_shard_destroy(s); return rc; }
        }
        std::vector<std::vector<uint32_t>> fa_ords(d->num_facets);
        for (int f = 0; f < d->num_facets; f++) {
            std::vector<uint32_t> fa_off(N + 1, 0);
            for (uint32_t i = 0; i <= nd; i++) fa_off[1 + i] = R.fa_off[f][i];
            std::vector<uint32_t> uniq(R.fa_ids[f]);
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            std::vector<uint32_t> &ords = fa_ords[f];
            ords.resize(R.fa_ids[f].size());
            for (size_t i = 0; i < ords.size(); i++)
                ords[i] = (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), R.fa_ids[f][i]) - uniq.begin());
            sd.fa_nord[f] = (uint32_t)uniq.size();
            s->avg_fv[f] = N > 1 ? (double)ords.size() / (double)(N - 1) : 0.0;
            s->n_fa[f] = ords.size(); s->cap_fa[f] = with_headroom(ords.size(), 65536);
            s->cap_ord[f] = (uint32_t)std::min<uint64_t>(with_headroom(uniq.size(), 4096), 0xFFFFFFF0ull);
            // ordinals are handed out in ascending id order here; documents added later (mgpu_shard_apply_delta) append
            // theirs, so the ascending list of ids with their ordinals is kept apart (facet recount: id -> posting)
            s->fa_sorted_id[f] = uniq;
            s->fa_sorted_ord[f].resize(uniq.size());
            for (size_t i = 0; i < uniq.size(); i++) s->fa_sorted_ord[f][i] = (uint32_t)i;
            if ((rc = upload_cap(s, fa_off, capN, &sd.fa_off[f])) || (rc = upload_cap(s, ords, s->cap_fa[f], &sd.fa_ord[f])) ||
                (rc = upload_cap(s, uniq, s->cap_ord[f], &sd.fa_ord2id[f])) || (rc = upload(s, uniq, &sd.fa_sorted_id[f]))) { mgpu_shard_destroy(s); return rc; }
        }
        // ---- doc-major records.  The stages after the match formula visit an effectively random
        // 0.1-10 % of the documents; with column/CSR arrays every visit costs a cache line per array.
        // A record gathers what those stages read (numbers, facet ordinals, docwords) into one
        // aligned block so that a hit costs one or two adjacent lines.  The CSR arrays stay: they
        // hold the tail of documents that do not fit their record and serve the overflow kernel.
        {
            const int nnum_in = d->num_numbers <= 4 ? d->num_numbers : 0;
            const uint32_t hdr = 16 + 8 * (uint32_t)nnum_in;
            auto facet_bytes = [&](uint32_t i) {
                uint32_t b = 0;
                for (int f = 0; f < d->num_facets; f++) b += 4 + 4 * (R.fa_off[f][i + 1] - R.fa_off[f][i]);
                return (b + 7) & ~7u;
            };
            // The facet area has ONE size per shard (what 90 % of the documents need, at most 64 B), so that the docword
            // entries of every record start at the same 8-byte slot `ws`: the rank kernel's lanes know their role without
            // reading a header.  A document whose facet section is larger keeps it in the per-field CSR (facets_in = 0).
            uint32_t FA = 0;
            if (d->num_facets > 0 && nd) {
                std::vector<uint32_t> fh(9, 0);                       // facet bytes in units of 8, 8 = "more than 56"
                for (uint32_t i = 0; i < nd; i++) fh[std::min<uint32_t>(facet_bytes(i) / 8, 8)]++;
                uint32_t acc = 0, u = 8;
                for (uint32_t c = 0; c <= 8; c++) { acc += fh[c]; if ((uint64_t)acc * 10 >= (uint64_t)nd * 9) { u = c; break; } }
                FA = std::min<uint32_t>(u, 8) * 8;
            }
            // stride: the smallest multiple of 128 B that holds 90 % of the documents whole (<= 512)
            std::vector<uint32_t> hist(5, 0);
            for (uint32_t i = 0; i < nd; i++) {
                uint32_t need = hdr + FA + 8 * (R.dw_off[i + 1] - R.dw_off[i]);
                uint32_t c = (need + 127) / 128; if (c > 4) c = 4;
                hist[c]++;
            }
            uint32_t S = 512, acc = 0;
            for (uint32_t c = 1; c <= 4; c++) { acc += hist[c]; if ((uint64_t)acc * 10 >= (uint64_t)nd * 9) { S = c * 128; break; } }
            if (hdr + FA + 8 > S) { FA = 0; }                          // degenerate schema: no facet area
            s->rec_hdr = hdr; s->rec_FA = FA; s->rec_nnum_in = (uint32_t)nnum_in; s->rec_stride = S;
            // Word 1 of a record is the u32 index of the document's positions in ONE array: the records (with room for
            // cap_docs of them), then a copy of the shard's position pool.  Positions that fit behind a document's
            // docwords are placed there (index into the record itself), the others are found in the copy; the kernel
            // adds the index to one base pointer either way.
            s->rec_pool_at = (uint64_t)s->cap_docs * (S / 4);
            if (s->rec_pool_at + s->cap_pp >= 0xFFFFFFFFull) { mgpu_shard_destroy(s); return fail(MGPU_E_LIMIT, "records + positions beyond 2^32 words in one shard"); }
            std::vector<uint32_t> rec((size_t)N * (S / 4), 0);
            std::vector<uint8_t> rlen(N, 0);
            s->all_staged = d->num_facets > 0;
            build_records(s, R, fa_ords, 1, 0, 0, rec.data() + (S / 4), rlen.data() + (N > 1 ? 1 : 0));
            {
                void *p = nullptr;
                const size_t words = (size_t)s->rec_pool_at + (size_t)s->cap_pp;
                HIPCHK(hipMalloc(&p, std::max<size_t>(words, 1) * 4));
                s->allocs.push_back(p); s->bytes += words * 4;
                HIPCHK(hipMemset(p, 0, std::max<size_t>(words, 1) * 4));
                if (!rec.empty()) HIPCHK(hipMemcpy(p, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
                if (!R.pp.empty()) HIPCHK(hipMemcpy((uint32_t *)p + s->rec_pool_at, R.pp.data(), R.pp.size() * 4, hipMemcpyHostToDevice));
                sd.rec = (const uint32_t *)p;
                sd.rec_words = (uint32_t)words;
            }
            { const uint8_t *pl; if ((rc = upload_cap(s, rlen, (size_t)s->cap_docs + 1, &pl))) { mgpu_shard_destroy(s); return rc; } sd.rec_len = pl; }
            // Off unless MGPU_RECLEN is set: measured on the bench corpus the predicated loads change nothing (0.660 vs
            // 0.655 ms per batch: the kernel is bound by instruction issue and latency, not by bytes) while the length
            // gather adds a third of a fabric request per hit to the counters (profiles/README.md, round 3)
            s->rec_len_ok = s->cap_docs <= (1u << 24) && getenv("MGPU_RECLEN") != nullptr;
            sd.rec_ws = (hdr + FA) / 8;
            if (s->all_staged)
                for (int f = 0; f < d->num_facets; f++) {
                    bool single = true;
                    for (uint32_t i = 0; i < nd && single; i++) single = R.fa_off[f][i + 1] - R.fa_off[f][i] <= 1;
                    if (single) s->fac_single |= 1u << f;
                }
            sd.rec_stride = S; sd.rec_nnum = (uint32_t)nnum_in;
        }
    }
    // ---- num_dbi docids in key order
    for (int i = 0; i < d->num_numbers; i++)
        if ((rc = set_num_order(s, i, d->np_doc[i], d->np_n[i], &sd.np_doc[i], &sd.grp_has[i]))) { mgpu_shard_destroy(s); return rc; }
    *out = s;
    return MGPU_OK;
}

// ------------------------------------------------------------------ f-2: a committed bulk write as a delta
// num_dbi of one field (the complete order: value descending, docid ascending) -> np_doc, grp_has, np_multi
static int set_num_order(mgpu_shard *s, int i, const uint32_t *doc, uint64_t n, const uint32_t **np_out, const uint64_t **grp_out) {
    ShardDev &sd = s->dev;
    const uint32_t N = sd.N;
    s->np_n[i] = n;
    std::vector<uint32_t> v(doc, doc + n);
    int rc;
    if ((rc = upload(s, v, np_out))) return rc;
    // what fast-rank mode needs to know about the field: which idnum2dbl groups exist, and whether a document has
    // several values
    std::vector<uint64_t> grp(((size_t)s->cap_docs / 1000 + 64) / 64 + 1, 0), seen(sd.W ? sd.W : 1, 0);
    s->np_multi &= ~(1u << i);
    for (uint32_t x : v) {
        if (x >= N) return fail(MGPU_E_ARG, "num_dbi docid beyond max_docid");
        grp[(x / 1000) >> 6] |= 1ull << ((x / 1000) & 63);
        if ((seen[x >> 6] >> (x & 63)) & 1ull) s->np_multi |= 1u << i;
        seen[x >> 6] |= 1ull << (x & 63);
    }
    return upload(s, grp, grp_out);
}
static void drop_alloc(mgpu_shard *s, const void *p, size_t bytes) {
    if (!p) return;
    for (size_t i = 0; i < s->allocs.size(); i++) if (s->allocs[i] == p) { s->allocs[i] = s->allocs.back(); s->allocs.pop_back(); break; }
    (void)hipFree((void *)p);
    s->bytes -= std::min<uint64_t>(s->bytes, bytes);
}

extern "C" uint32_t mgpu_shard_max_docid(const mgpu_shard_t *s) { return s ? s->dev.N - 1 : 0; }
extern "C" uint32_t mgpu_shard_doc_capacity(const mgpu_shard_t *s) { return s ? s->cap_docs - 1 : 0; }

extern "C" int mgpu_shard_apply_delta(mgpu_shard_t *s, const mgpu_shard_delta_t *d) {
    if (!s || !d || !d->post_off || (d->num_postings && !d->post_doc && d->post_off[d->num_postings])) return fail(MGPU_E_ARG, "null argument");
    ShardDev &sd = s->dev;
    const uint32_t N0 = sd.N, N1 = d->max_docid + 1;
    if (N1 < N0) return fail(MGPU_E_ARG, "a delta cannot take documents away");
    if (N1 > s->cap_docs) return fail(MGPU_E_LIMIT, "the mirror has no room for that many more documents: rebuild it");
    if (d->num_postings < sd.num_postings) return fail(MGPU_E_ARG, "a delta cannot take postings away");
    const uint32_t nd = N1 - N0, P0 = sd.num_postings, P1 = d->num_postings;
    if (nd && (!d->dd_off || !d->dd)) return fail(MGPU_E_ARG, "blobs of the new documents missing");
    for (int i = 0; i < sd.num_numbers; i++) if (d->np_n[i] && !d->np_doc[i]) return fail(MGPU_E_ARG, "num_dbi order missing");
    HIPCHK(hipSetDevice(s->device));
    int rc;
    // ---- 1. what can be checked is checked before the mirror is touched: MGPU_E_ARG / MGPU_E_LIMIT leave it as it was.
    // A device failure further down (MGPU_E_HIP: out of memory, a lost device) may leave it half-updated -- the rows
    // have their new bits before the tables are swapped: the caller destroys such a mirror and builds a new one.
    // pid scheme: every old posting exactly once, in order
    std::vector<uint32_t> old_pid(P1);
    {
        uint32_t next = 0;
        for (uint32_t p = 0; p < P1; p++) {
            const uint32_t op = d->old_pid ? d->old_pid[p] : (p < P0 ? p : 0xFFFFFFFFu);
            if (op != 0xFFFFFFFFu) { if (op != next) return fail(MGPU_E_ARG, "old pids must appear once each, ascending"); next++; }
            old_pid[p] = op;
        }
        if (next != P0) return fail(MGPU_E_ARG, "the delta's pid map does not cover the mirror's postings");
        if (P1 && old_pid[0] != 0) return fail(MGPU_E_ARG, "pid 0 (all documents) must stay pid 0");
    }
    for (uint32_t p = 0; p < P1; p++) {
        if (d->post_off[p + 1] < d->post_off[p]) return fail(MGPU_E_ARG, "delta posting offsets must ascend");
        for (uint64_t i = d->post_off[p]; i < d->post_off[p + 1]; i++) {
            const uint32_t x = d->post_doc[i];
            if (x < N0 || x >= N1) return fail(MGPU_E_ARG, "a delta posting holds a docid that is not new");
            if (i > d->post_off[p] && x <= d->post_doc[i - 1]) return fail(MGPU_E_ARG, "delta posting docids not ascending");
        }
    }
    // the new documents
    DocRun R;
    if ((rc = decode_docs(d->dd, d->dd_off, nd, sd.num_numbers, sd.num_facets, R))) return rc;
    if (s->n_dw + R.dw.size() > s->cap_dw || s->n_pp + R.pp.size() > s->cap_pp || s->n_dw + R.dw.size() >= 0xFFFFFFFFull)
        return fail(MGPU_E_LIMIT, "the mirror has no room for the new documents' words: rebuild it");
    // facet ids -> ordinals; ids the field has not seen get the next ordinals
    std::vector<std::vector<uint32_t>> fa_ords(sd.num_facets), new_ids(sd.num_facets);
    std::vector<std::vector<uint32_t>> sorted_id(sd.num_facets), sorted_ord(sd.num_facets);
    for (int f = 0; f < sd.num_facets; f++) {
        if (s->n_fa[f] + R.fa_ids[f].size() > s->cap_fa[f]) return fail(MGPU_E_LIMIT, "the mirror has no room for the new documents' facet values: rebuild it");
        std::vector<uint32_t> fresh;
        const std::vector<uint32_t> &sid = s->fa_sorted_id[f];
        for (uint32_t id : R.fa_ids[f]) if (!std::binary_search(sid.begin(), sid.end(), id)) fresh.push_back(id);
        std::sort(fresh.begin(), fresh.end());
        fresh.erase(std::unique(fresh.begin(), fresh.end()), fresh.end());
        if ((uint64_t)sd.fa_nord[f] + fresh.size() > s->cap_ord[f]) return fail(MGPU_E_LIMIT, "the mirror has no room for the new facet values: rebuild it");
        new_ids[f] = fresh;                                               // ordinal of fresh[i] = fa_nord + i
        // merged ascending id list with ordinals
        sorted_id[f].reserve(sid.size() + fresh.size()); sorted_ord[f].reserve(sid.size() + fresh.size());
        size_t a = 0, b = 0;
        while (a < sid.size() || b < fresh.size()) {
            if (b >= fresh.size() || (a < sid.size() && sid[a] < fresh[b])) { sorted_id[f].push_back(sid[a]); sorted_ord[f].push_back(s->fa_sorted_ord[f][a]); a++; }
            else { sorted_id[f].push_back(fresh[b]); sorted_ord[f].push_back(sd.fa_nord[f] + (uint32_t)b); b++; }
        }
        fa_ords[f].resize(R.fa_ids[f].size());
        for (size_t i = 0; i < fa_ords[f].size(); i++) {
            const size_t at = (size_t)(std::lower_bound(sorted_id[f].begin(), sorted_id[f].end(), R.fa_ids[f][i]) - sorted_id[f].begin());
            fa_ords[f][i] = sorted_ord[f][at];
        }
    }
    const uint32_t S = s->rec_stride;
    std::vector<uint32_t> rec((size_t)nd * (S / 4), 0);
    std::vector<uint8_t> rlen(nd, 0);
    bool staged = s->all_staged;
    {
        const bool keep = s->all_staged;
        build_records(s, R, fa_ords, N0, s->n_dw, s->n_pp, rec.data(), rlen.data());
        staged = s->all_staged; s->all_staged = keep;                     // committed below
    }
    // postings: dense rows keep their rows, everything else is a run of the new sparse array
    std::vector<int32_t> new_dense(P1, -1);
    std::vector<uint64_t> new_off((size_t)P1 + 1, 0);
    std::vector<uint32_t> new_card(P1, 0), winbase(P1, 0xFFFFFFFFu);
    const uint32_t W1 = (N1 + 63) / 64, nwin = (W1 + MG_FORMULA_WORDS - 1) / MG_FORMULA_WORDS;
    uint64_t nsp = 0, nwi = 0;
    std::vector<ScatterItem> items;
    for (uint32_t p = 0; p < P1; p++) {
        const uint32_t op = old_pid[p];
        const uint64_t dn = d->post_off[p + 1] - d->post_off[p];
        new_off[p] = nsp;
        uint64_t n = dn;
        if (op != 0xFFFFFFFFu) {
            new_dense[p] = s->post_dense[op];
            n += s->post_card[op];
            if (new_dense[p] < 0) nsp += s->post_off[op + 1] - s->post_off[op];
        }
        new_card[p] = (uint32_t)std::min<uint64_t>(n, N1);
        if (new_dense[p] < 0) {
            nsp += dn;
            const uint64_t len = nsp - new_off[p];
            if (len > MG_SHORT_POSTING) { winbase[p] = (uint32_t)nwi; nwi += (uint64_t)nwin + 1; }
        }
    }
    new_off[P1] = nsp;
    if (nsp >= 0xFFFFFFFFull || nwi >= 0xFFFFFFFFull) return fail(MGPU_E_LIMIT, "more than 2^32 sparse posting entries or window entries in one shard");
    // ---- 2. device work: the delta's docids, the merged sparse array and its window index, the dense rows' new bits
    uint32_t *d_delta = nullptr, *d_newdoc = nullptr, *d_win = nullptr, *d_oldpid = nullptr, *d_winbase = nullptr;
    u64 *d_deltaoff = nullptr, *d_newoff = nullptr; int32_t *d_newdense = nullptr; ScatterItem *d_items = nullptr;
    auto cleanup_tmp = [&]() { if (d_delta) (void)hipFree(d_delta); if (d_deltaoff) (void)hipFree(d_deltaoff); if (d_oldpid) (void)hipFree(d_oldpid);
                               if (d_winbase) (void)hipFree(d_winbase); if (d_items) (void)hipFree(d_items); };
    auto bail = [&](hipError_t e) { cleanup_tmp(); if (d_newdoc) (void)hipFree(d_newdoc); if (d_win) (void)hipFree(d_win);
                                     if (d_newoff) (void)hipFree(d_newoff); if (d_newdense) (void)hipFree(d_newdense);
                                     return fail(MGPU_E_HIP, hipGetErrorString(e)); };
    hipError_t e;
    const uint64_t ndelta = d->post_off[P1];
#define TRY(x) do { if ((e = (x)) != hipSuccess) return bail(e); } while (0)
    TRY(hipMalloc((void **)&d_delta, std::max<uint64_t>(ndelta, 1) * 4));
    TRY(hipMalloc((void **)&d_deltaoff, ((size_t)P1 + 1) * 8));
    TRY(hipMalloc((void **)&d_oldpid, std::max<size_t>(P1, 1) * 4));
    TRY(hipMalloc((void **)&d_winbase, std::max<size_t>(P1, 1) * 4));
    TRY(hipMalloc((void **)&d_newdoc, std::max<uint64_t>(nsp, 1) * 4));
    TRY(hipMalloc((void **)&d_win, std::max<uint64_t>(nwi, 1) * 4));
    TRY(hipMalloc((void **)&d_newoff, ((size_t)P1 + 1) * 8));
    TRY(hipMalloc((void **)&d_newdense, std::max<size_t>(P1, 1) * 4));
    if (ndelta) TRY(hipMemcpy(d_delta, d->post_doc, ndelta * 4, hipMemcpyHostToDevice));
    TRY(hipMemcpy(d_deltaoff, d->post_off, ((size_t)P1 + 1) * 8, hipMemcpyHostToDevice));
    if (P1) {
        TRY(hipMemcpy(d_oldpid, old_pid.data(), (size_t)P1 * 4, hipMemcpyHostToDevice));
        TRY(hipMemcpy(d_winbase, winbase.data(), (size_t)P1 * 4, hipMemcpyHostToDevice));
        TRY(hipMemcpy(d_newdense, new_dense.data(), (size_t)P1 * 4, hipMemcpyHostToDevice));
    }
    TRY(hipMemcpy(d_newoff, new_off.data(), ((size_t)P1 + 1) * 8, hipMemcpyHostToDevice));
    for (uint32_t p = 0; p < P1; p++) {
        if (new_dense[p] < 0) continue;
        const uint64_t a = d->post_off[p], n = d->post_off[p + 1] - a;
        for (uint64_t at = 0; at < n; at += MG_SCATTER_CHUNK) {
            ScatterItem it; it.src = d_delta + a + at; it.dst = (uint64_t *)sd.dense + (size_t)new_dense[p] * sd.Wp;
            it.n = (uint32_t)std::min<uint64_t>(n - at, MG_SCATTER_CHUNK); it.pad = 0;
            items.push_back(it);
        }
    }
    if (!items.empty()) {
        TRY(hipMalloc((void **)&d_items, items.size() * sizeof(ScatterItem)));
        TRY(hipMemcpy(d_items, items.data(), items.size() * sizeof(ScatterItem), hipMemcpyHostToDevice));
    }
    // the merged array is built next to the live one; the rows' new bits are the first change of the mirror itself
    mgk_merge_postings(nullptr, P1, d_oldpid, d_newdense, (const unsigned long long *)d_newoff, sd.post_dense, (const unsigned long long *)sd.post_off, sd.post_doc,
                       (const unsigned long long *)d_deltaoff, d_delta, d_newdoc, d_winbase, nwin, d_win);
    TRY(hipStreamSynchronize(nullptr));
    TRY(hipGetLastError());
    if (!items.empty()) mgk_scatter(nullptr, d_items, (uint32_t)items.size());
    TRY(hipStreamSynchronize(nullptr));
    TRY(hipGetLastError());
    // ---- 3. the appended tails of the per-document arrays
    auto put = [&](const void *dst_base, size_t elem, size_t at, const void *src, size_t n) -> hipError_t {
        if (!n) return hipSuccess;
        return hipMemcpy((char *)dst_base + at * elem, src, n * elem, hipMemcpyHostToDevice);
    };
    {
        std::vector<uint32_t> o(nd + 1);
        for (uint32_t i = 0; i <= nd; i++) o[i] = (uint32_t)(s->n_dw + R.dw_off[i]);
        TRY(put(sd.dw_off, 4, N0, o.data(), (size_t)nd + 1));
        for (uint32_t i = 0; i <= nd; i++) o[i] = (uint32_t)(s->n_pp + R.pp_off[i]);
        TRY(put(sd.pp_off, 4, N0, o.data(), (size_t)nd + 1));
        TRY(put(sd.dw, sizeof(uint2), s->n_dw, R.dw.data(), R.dw.size()));
        TRY(put(sd.pp, 4, s->n_pp, R.pp.data(), R.pp.size()));
        TRY(put(sd.rec, 4, s->rec_pool_at + s->n_pp, R.pp.data(), R.pp.size()));
        for (int i = 0; i < sd.num_numbers; i++) TRY(put(sd.num_col[i], 8, N0, R.cols[i].data(), nd));
        for (int f = 0; f < sd.num_facets; f++) {
            for (uint32_t i = 0; i <= nd; i++) o[i] = (uint32_t)(s->n_fa[f] + R.fa_off[f][i]);
            TRY(put(sd.fa_off[f], 4, N0, o.data(), (size_t)nd + 1));
            TRY(put(sd.fa_ord[f], 4, s->n_fa[f], fa_ords[f].data(), fa_ords[f].size()));
            TRY(put(sd.fa_ord2id[f], 4, sd.fa_nord[f], new_ids[f].data(), new_ids[f].size()));
        }
        TRY(put(sd.rec, 4, (size_t)N0 * (S / 4), rec.data(), rec.size()));
        TRY(put(sd.rec_len, 1, N0, rlen.data(), nd));
    }
#undef TRY
    cleanup_tmp();
    // ---- 4. commit: the new posting tables replace the old ones
    drop_alloc(s, sd.post_doc, (size_t)std::max<uint64_t>(s->n_sparse, 1) * 4);
    drop_alloc(s, sd.post_win, 0);
    drop_alloc(s, sd.post_off, ((size_t)P0 + 1) * 8);
    drop_alloc(s, sd.post_dense, (size_t)std::max<uint32_t>(P0, 1) * 4);
    s->allocs.push_back(d_newdoc); s->allocs.push_back(d_win); s->allocs.push_back(d_newoff); s->allocs.push_back(d_newdense);
    s->bytes += nsp * 4 + nwi * 4 + ((size_t)P1 + 1) * 8 + (size_t)P1 * 4;
    sd.post_doc = d_newdoc; sd.post_win = d_win; sd.post_off = (const uint64_t *)d_newoff; sd.post_dense = d_newdense;
    sd.num_postings = P1; sd.N = N1; sd.W = W1;
    s->post_dense = new_dense; s->post_off = new_off; s->post_card = new_card; s->post_winbase = winbase; s->n_sparse = nsp;
    s->n_dw += R.dw.size(); s->n_pp += R.pp.size();
    sd.dw_len = (uint32_t)s->n_dw; sd.pp_len = (uint32_t)s->n_pp;
    s->avg_dw = N1 > 1 ? (double)s->n_dw / (double)(N1 - 1) : 0.0;
    if (!staged) { s->all_staged = false; s->fac_single = 0; }
    for (int f = 0; f < sd.num_facets; f++) {
        s->n_fa[f] += fa_ords[f].size();
        sd.fa_nord[f] += (uint32_t)new_ids[f].size();
        s->avg_fv[f] = N1 > 1 ? (double)s->n_fa[f] / (double)(N1 - 1) : 0.0;
        for (uint32_t i = 0; i < nd; i++) if (R.fa_off[f][i + 1] - R.fa_off[f][i] > 1) s->fac_single &= ~(1u << f);
        s->fa_sorted_id[f].swap(sorted_id[f]); s->fa_sorted_ord[f].swap(sorted_ord[f]);
        const uint32_t *old = sd.fa_sorted_id[f];
        if ((rc = upload(s, s->fa_sorted_id[f], &sd.fa_sorted_id[f]))) return rc;
        drop_alloc(s, old, 0);
    }
    // num_dbi: the complete order again (4 bytes per document and field)
    for (int i = 0; i < sd.num_numbers; i++) {
        const uint32_t *old_np = sd.np_doc[i]; const uint64_t *old_grp = sd.grp_has[i];
        const uint64_t old_n = s->np_n[i];
        if ((rc = set_num_order(s, i, d->np_doc[i], d->np_n[i], &sd.np_doc[i], &sd.grp_has[i]))) return rc;
        drop_alloc(s, old_np, (size_t)std::max<uint64_t>(old_n, 1) * 4); drop_alloc(s, old_grp, 0);
    }
    return MGPU_OK;
}

// ------------------------------------------------------------------ context
struct Arena {
    char *dev = nullptr; size_t cap = 0;
};

struct mgpu_ctx {
    int device;
    hipStream_t stream = nullptr;
    Arena arena;
    char *pin = nullptr; size_t pin_cap = 0;          // pinned staging for uploads
    char *pin_out = nullptr; size_t pin_out_cap = 0;  // pinned buffer for results
    u64 *pool = nullptr; uint64_t pool_cap = 0;       // overflow position pool, in u64 entries
    // Self-cleaned pool: facet histograms and select histograms.  It is all zero between batches because
    // every kernel that consumes a bin puts the zero back (k_fkeys / the select kernels / k_card; the wide
    // select passes); a batch that does not run to its end leaves `cln_dirty` set and the next one clears it.
    Arena cln; bool cln_dirty = true;
    uint32_t *pool_used = nullptr;
    char *na_scratch = nullptr;                        // this device's g_noalloc_scratch (mgpu_posting_op_noalloc)
    char *na_pin = nullptr;                            // pinned bounce buffer of that entry point: [card 256 B | one row]
    struct Pending *pend = nullptr;                    // the batch in flight (execute_begin .. execute_end)
    // term lookup (mgpu_lookup_terms): job table + term blob, hit list, counter; grown on demand, kept between calls
    char *lk_in = nullptr; size_t lk_in_cap = 0;
    uint32_t *lk_hits = nullptr; uint32_t lk_hits_cap = 0;
    uint32_t *lk_count = nullptr;
    char *hl_buf = nullptr; size_t hl_cap = 0;                // highlight matcher (mgpu_highlight_match): inputs + result, grown on demand
    uint2 *lk_cand = nullptr; uint32_t lk_cand_cap = 0;      // (job, word) pairs the signature filter let through
    hipEvent_t ev[MGPU_NUM_STAGES + 1] = {};
    int profiling;
    float stage_ms[MGPU_NUM_STAGES];
    float total_ms;
    float host_ms[3];               // last mgpu_execute: lowering, launch + wait, decode (wall clock)
    uint64_t stage_bytes[MGPU_NUM_STAGES];
};

static void pending_free(struct Pending *p);
extern "C" void mgpu_ctx_destroy(mgpu_ctx_t *c);
// (a failure part-way leaves nothing behind: the half-built context goes through mgpu_ctx_destroy)
static int ctx_create_steps(mgpu_ctx *c, int device, uint64_t scratch_bytes) {
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (!scratch_bytes) scratch_bytes = 64ull << 20;
    c->pool_cap = scratch_bytes / 8;
    HIPCHK(hipMalloc((void **)&c->pool, c->pool_cap * 8));
    HIPCHK(hipMalloc((void **)&c->pool_used, 256));
    for (int i = 0; i <= MGPU_NUM_STAGES; i++) HIPCHK(hipEventCreate(&c->ev[i]));
    // set-up of the allocation-free entry point: this device's address of the static scratch array (each GPU has
    // its own copy at its own address) and a pinned bounce buffer for its results
    {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        if (!g_na_ptr[device]) {
            void *ptr = nullptr;
            HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_noalloc_scratch)));
            g_na_ptr[device] = (char *)ptr;
        }
    }
    c->na_scratch = g_na_ptr[device];
    HIPCHK(hipHostMalloc((void **)&c->na_pin, NA_ROW_BYTES + 256, hipHostMallocDefault));
    return MGPU_OK;
}
extern "C" int mgpu_ctx_create(int device, uint64_t scratch_bytes, mgpu_ctx_t **out) {
    if (!out) return fail(MGPU_E_ARG, "null argument");
    if (device < 0 || device >= NA_MAX_DEVICES) return fail(MGPU_E_LIMIT, "device index above 63");
    HIPCHK(hipSetDevice(device));
    mgpu_ctx *c = new mgpu_ctx();
    c->device = device; c->profiling = 0; c->total_ms = 0;
    memset(c->stage_ms, 0, sizeof(c->stage_ms)); memset(c->stage_bytes, 0, sizeof(c->stage_bytes));
    const int rc = ctx_create_steps(c, device, scratch_bytes);
    if (rc != MGPU_OK) { mgpu_ctx_destroy(c); return rc; }          // (the thread's error text is the failing step's)
    *out = c;
    return MGPU_OK;
}

extern "C" void mgpu_ctx_destroy(mgpu_ctx_t *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    pending_free(c->pend);
    if (c->arena.dev) (void)hipFree(c->arena.dev);
    if (c->cln.dev) (void)hipFree(c->cln.dev);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    if (c->na_pin) (void)hipHostFree(c->na_pin);
    if (c->pool) (void)hipFree(c->pool);
    if (c->pool_used) (void)hipFree(c->pool_used);
    if (c->lk_in) (void)hipFree(c->lk_in);
    if (c->lk_hits) (void)hipFree(c->lk_hits);
    if (c->lk_count) (void)hipFree(c->lk_count);
    if (c->lk_cand) (void)hipFree(c->lk_cand);
    if (c->hl_buf) (void)hipFree(c->hl_buf);
    for (int i = 0; i <= MGPU_NUM_STAGES; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void mgpu_ctx_set_profiling(mgpu_ctx_t *c, int on) { if (c) c->profiling = on; }

static const char *k_stage_names[MGPU_NUM_STAGES] = {
    "upload+clear", "scatter", "formula+compact", "rank", "facets+aggs", "facet-keys", "select", "download" };
extern "C" const char *mgpu_stage_name(int s) { return (s >= 0 && s < MGPU_NUM_STAGES) ? k_stage_names[s] : ""; }

extern "C" int mgpu_last_host_timing(const mgpu_ctx_t *c, float *host_ms) {
    if (!c || !host_ms) return fail(MGPU_E_ARG, "bad argument");
    memcpy(host_ms, c->host_ms, sizeof(c->host_ms));
    return MGPU_OK;
}
static double wall_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

extern "C" int mgpu_last_timing(const mgpu_ctx_t *c, float *total_ms, float *stage_ms) {
    if (!c) return MGPU_E_ARG;
    if (total_ms) *total_ms = c->total_ms;
    if (stage_ms) memcpy(stage_ms, c->stage_ms, sizeof(c->stage_ms));
    return MGPU_OK;
}
extern "C" int mgpu_last_bytes(const mgpu_ctx_t *c, uint64_t *stage_bytes) {
    if (!c || !stage_bytes) return MGPU_E_ARG;
    memcpy(stage_bytes, c->stage_bytes, sizeof(c->stage_bytes));
    return MGPU_OK;
}

static int ensure_arena(mgpu_ctx *c, size_t bytes) {
    if (bytes <= c->arena.cap) return MGPU_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->arena.dev) HIPCHK(hipFree(c->arena.dev));
    size_t cap = std::max(bytes + bytes / 4, (size_t)(16u << 20));
    c->arena.dev = nullptr; c->arena.cap = 0;
    // The arena of a batch is sized by the squeries' hit BOUNDS (about 60 bytes per possible hit and squery: hit lists,
    // keys, per-hit facet ordinals), and every context in flight holds its own: a batch of one-word queries that may match
    // half of a 10M-document index wants tens of GB.  No room: first without the growth margin, then a limit the caller can
    // act on (fewer queries per batch, or fewer contexts in flight) -- the context stays usable for a smaller batch.
    if (hipMalloc((void **)&c->arena.dev, cap) != hipSuccess) {
        (void)hipGetLastError();
        c->arena.dev = nullptr; cap = bytes;
        if (hipMalloc((void **)&c->arena.dev, cap) != hipSuccess) {
            (void)hipGetLastError();
            c->arena.dev = nullptr;
            char msg[200];
            snprintf(msg, sizeof(msg), "the batch needs %.1f GB of device scratch (hit bounds of its squeries) and the device has no room: fewer queries per batch or fewer contexts in flight",
                     (double)bytes / 1e9);
            return fail(MGPU_E_LIMIT, msg);
        }
    }
    c->arena.cap = cap;
    return MGPU_OK;
}
static int ensure_pin(char **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return MGPU_OK;
    if (*p) HIPCHK(hipHostFree(*p));
    size_t ncap = std::max(bytes + bytes / 2, (size_t)(1u << 20));
    *p = nullptr; *cap = 0;
    HIPCHK(hipHostMalloc((void **)p, ncap, hipHostMallocDefault));
    *cap = ncap;
    return MGPU_OK;
}

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes, size_t align = 256) { off = (off + align - 1) / align * align; size_t r = off; off += bytes; return r; }
};

// result header copied back per squery
struct ResHdr {
    uint32_t num_hits, err, ovf_cnt, rank_n;
    uint32_t fast_flag, total_hits, num_rank, pad0;      // fast-rank: SqDev::fast_flag / total_hits / num_rank
    uint32_t fac_n[MGPU_MAX_FIELDS];
    uint32_t card_n[MGPU_MAX_AGGS];
    AggPart agg[MGPU_MAX_AGGS];
};

struct SqHost {                          // host bookkeeping for one squery of the batch
    size_t hdr_off;                      // into results region
    uint32_t k;
    size_t rank_hi, rank_lo, rank_pl;    // results-region offsets
    int nfac; int fac_field[MGPU_MAX_FIELDS]; uint32_t fac_k[MGPU_MAX_FIELDS];
    size_t fac_hi[MGPU_MAX_FIELDS], fac_lo[MGPU_MAX_FIELDS], fac_pl[MGPU_MAX_FIELDS];
    int ncard; int card_agg[MGPU_MAX_AGGS]; size_t card_ids_off[MGPU_MAX_AGGS]; uint32_t card_cap[MGPU_MAX_AGGS];
    bool card_in_res[MGPU_MAX_AGGS];      // the id set lies in the results region (comes back with the one download), else in the scratch region
    size_t docid_map_off, hits_off, key_hi_off, key_lo_off;
    int rank_job, fac_job[MGPU_MAX_FIELDS];
    KeyField kf[8]; int nkf;
    uint64_t formula_rows;               // rows read by the formula kernel
    uint64_t scatter_docs;
    uint64_t sparse_docs;                // sparse posting entries of the terms (read once per squery by k_formula)
};

// ---- state of one batch between its launch (execute_begin) and its decode (execute_end); kept in the context
// and reused from batch to batch, so that the vectors keep their capacity
struct PendingScatter { int sq; int row; const uint32_t *src_dev; uint64_t n; };
struct PendingRange { int sq; int row; int field; uint64_t lo, hi; };       // a range row built from the column (RangeRow)
// a filter operand that is not a dense row (a sparse posting, a range of a num_dbi order) is scattered into a row of the batch's
// zero region; the squeries of one shard that name the same operand share the row and the scatter
struct RowKey {
    const void *sh; uint32_t kind, a; uint64_t lo, hi;
    bool operator<(const RowKey &o) const { return std::tie(sh, kind, a, lo, hi) < std::tie(o.sh, o.kind, o.a, o.lo, o.hi); }
};
struct JobFix { int sq; int kind; int field; };       // kind 0 rank, 1 facet
struct Offs {      // arena offsets per squery (zero or nonzero region relative)
    size_t rows, hist[MGPU_MAX_FIELDS], touched[MGPU_MAX_FIELDS], ntouched[MGPU_MAX_FIELDS], facord[MGPU_MAX_FIELDS], andor, ovfcnt, docmap, win, rawcnt, hitsraw, hits, khi, klo, ovfl, aggpart;
    size_t fkhi[MGPU_MAX_FIELDS], fklo[MGPU_MAX_FIELDS];
    size_t dbmap;
    int aux_index;       // fast-eligible squery: index of its zero-typo twin in the batch, else -1
};
struct SelOff { size_t cand0, cand1, hist, ctl, outidx, andor; uint32_t mcap; };
struct Pending {
    bool active = false;
    int n = 0;
    mgpu_shard_t *const *shards = nullptr; const mgpu_plan_t *plans = nullptr;
    std::vector<SqDev> sqd;
    std::vector<SqHost> sqh;
    std::vector<uint32_t> blob;
    std::vector<PendingScatter> pscat;
    std::vector<SelJob> seljobs; std::vector<SelOut> selouts;
    std::vector<FkJob> fkjobs; std::vector<CardJob> cardjobs;
    std::vector<JobFix> jobfix;
    std::vector<int> nrows;
    std::vector<Offs> offs;
    std::vector<SelOff> seloff;
    std::vector<uint32_t> sl, dl, fo, pl;       // per-squery scratch lists
    std::vector<ScatterItem> items;
    // what the decode needs from the launch
    size_t res_bytes = 0, nz_base = 0, up_bytes = 0, up_base = 0, up_sel = 0, up_selout = 0;
    uint32_t max_selm = 0;
    bool fuse_fk = false, prof = false;
    int n_all = 0;                               // n plus the zero-typo twins of the fast-eligible squeries
    std::vector<int> aux_parent;
    std::vector<uint32_t> elig_list;
    std::vector<RecountJob> recount;
    int nev = 0;                                 // events recorded (profiling)
    double t_begin = 0, t_lowered = 0;
};


static int build_key_layout(const mgpu_plan_t &p, KeyField *kf, int *nkf) {
    int n = 0, acc = 0;
    bool seen[8] = {false, false, false, false, false, false, false, false};
    auto add = [&](int kind, int bits) {
        if (seen[kind]) return;
        seen[kind] = true;
        kf[n].kind = (int8_t)kind; kf[n].bits = (uint8_t)bits; acc += bits; kf[n].shift = (uint8_t)(128 - acc); kf[n].hi = 0; n++;
    };
    for (int i = 0; i < MGPU_MAX_RULES; i++) {
        int r = p.rules[i];
        if (r == MGPU_R_DONE || r == MGPU_R_MAX) break;
        switch (r) {
            case MGPU_R_TYPO: add(KF_TYPO, 8); break;
            case MGPU_R_PROX: add(KF_PROX, 16); break;
            case MGPU_R_POS: add(KF_POS, 16); break;
            case MGPU_R_EXACT: add(KF_EXACT, 8); break;
            case MGPU_R_FIELD: add(KF_FIELD, 8); break;
            case MGPU_R_COMP: case MGPU_R_COMP_ASC:
                if (p.rank_by < 0) return fail(MGPU_E_ARG, "compare rule without rank_by");
                if (!seen[KF_COMP_DESC] && !seen[KF_COMP_ASC]) add(r == MGPU_R_COMP ? KF_COMP_DESC : KF_COMP_ASC, 64);
                break;
            default: return fail(MGPU_E_UNSUPPORTED, "rank rule not supported (geo)");
        }
    }
    // fields outside the rule list ride in the low bits so the record can be rebuilt; they only
    // order documents the reference's comparator calls equal
    add(KF_TYPO, 8); add(KF_PROX, 16); add(KF_POS, 16); add(KF_EXACT, 8); add(KF_FIELD, 8);
    if (acc > 128) return fail(MGPU_E_LIMIT, "rank key wider than 128 bits");
    *nkf = n;
    return MGPU_OK;
}

static void decode_key(const KeyField *kf, int nkf, u64 hi, u64 lo, mgpu_rank_t *r) {
    unsigned __int128 key = ((unsigned __int128)hi << 64) | lo;
    r->compare = 0.0;
    for (int i = 0; i < nkf; i++) {
        u64 v = (u64)(key >> kf[i].shift);
        if (kf[i].bits < 64) v &= ((1ull << kf[i].bits) - 1);
        switch (kf[i].kind) {
            case KF_TYPO: r->typos = (uint8_t)v; break;
            case KF_PROX: r->proximity = (uint16_t)v; break;
            case KF_POS: r->position = (uint16_t)v; break;
            case KF_EXACT: r->exact = (uint8_t)(255 - v); break;
            case KF_FIELD: r->field = (uint8_t)v; break;
            case KF_COMP_DESC: v = ~v; /* fallthrough */
            case KF_COMP_ASC: {
                u64 bits = (v & 0x8000000000000000ull) ? (v ^ 0x8000000000000000ull) : ~v;
                memcpy(&r->compare, &bits, 8);
            } break;
        }
    }
}

extern "C" void mgpu_sqout_free(mgpu_sqout_t *o) {
    if (!o) return;
    free(o->ranks);
    for (int f = 0; f < MGPU_MAX_FIELDS; f++) free(o->fc[f]);
    for (int a = 0; a < MGPU_MAX_AGGS; a++) free(o->aggs[a].card_ids);
    free(o->dbg_docid_map); free(o->dbg_all_ranks);
    memset(o, 0, sizeof(*o));
}

static void pending_free(Pending *p) { delete p; }
static int execute_begin(mgpu_ctx_t *c, mgpu_shard_t *const *shards, const mgpu_plan_t *plans, int n);
static int execute_end(mgpu_ctx_t *c, mgpu_sqout_t *outs, bool *completed);

// Launch half of mgpu_execute: lowers the plans, lays out the arena, enqueues every copy and kernel of the batch on
// the context's stream and returns without waiting.  `shards` and `plans` must stay valid until mgpu_execute_end.
extern "C" int mgpu_execute_begin(mgpu_ctx_t *c, mgpu_shard_t *const *shards, const mgpu_plan_t *plans, int n) {
    if (!c || !shards || !plans || n <= 0) return fail(MGPU_E_ARG, "bad argument");
    if (c->pend && c->pend->active) return fail(MGPU_E_ARG, "a batch is already in flight on this context");
    if (!c->pend) c->pend = new Pending();
    const int rc = execute_begin(c, shards, plans, n);
    c->pend->active = rc == MGPU_OK;
    return rc;
}
// Wait + decode half: outs[i] is filled as by mgpu_execute.
extern "C" int mgpu_execute_end(mgpu_ctx_t *c, mgpu_sqout_t *outs) {
    if (!c || !outs) return fail(MGPU_E_ARG, "bad argument");
    if (!c->pend || !c->pend->active) return fail(MGPU_E_ARG, "no batch in flight on this context");
    const int n = c->pend->n;
    for (int i = 0; i < n; i++) memset(&outs[i], 0, sizeof(outs[i]));
    bool completed = false;
    const int rc = execute_end(c, outs, &completed);
    c->pend->active = false;
    // an early return (HIP failure) hands nothing back: whatever the decode had already allocated is released
    // here, so that the caller owns arrays only when the batch ran to its end
    if (!completed) for (int i = 0; i < n; i++) mgpu_sqout_free(&outs[i]);
    return rc;
}

extern "C" int mgpu_execute(mgpu_ctx_t *c, mgpu_shard_t *const *shards, const mgpu_plan_t *plans, int n,
                            mgpu_sqout_t *outs) {
    if (!outs) return fail(MGPU_E_ARG, "bad argument");
    for (int i = 0; i < n; i++) memset(&outs[i], 0, sizeof(outs[i]));
    const int rc = mgpu_execute_begin(c, shards, plans, n);
    if (rc) return rc;
    return mgpu_execute_end(c, outs);
}

static int execute_begin(mgpu_ctx_t *c, mgpu_shard_t *const *shards, const mgpu_plan_t *plans, int n) {
    if (n > 65535) return fail(MGPU_E_LIMIT, "batch larger than 65535");
    HIPCHK(hipSetDevice(c->device));
    const double t_begin = wall_ms();
    Pending &P = *c->pend;
    P.n = n; P.shards = shards; P.plans = plans;

    // ---------------- lower plans into host staging ----------------
    P.sqd.resize(2 * (size_t)n);                           // room for a zero-typo twin per squery; every used element is memset below
    SqDev *sqd = P.sqd.data();
    std::vector<SqHost> &sqh = P.sqh; sqh.resize(2 * (size_t)n);
    std::vector<int> &aux_parent = P.aux_parent; aux_parent.clear();
    std::vector<uint32_t> &elig_list = P.elig_list; elig_list.clear();
    std::vector<RecountJob> &recount = P.recount; recount.clear();
    int n_all = n;
    std::vector<uint32_t> &blob = P.blob; blob.clear();
    std::vector<PendingScatter> &pscat = P.pscat; pscat.clear();
    std::vector<SelJob> &seljobs = P.seljobs; std::vector<SelOut> &selouts = P.selouts;
    std::vector<FkJob> &fkjobs = P.fkjobs; fkjobs.clear();
    std::vector<CardJob> &cardjobs = P.cardjobs; cardjobs.clear();
    std::vector<JobFix> &jobfix = P.jobfix; jobfix.clear();

    Bump res;          // results region (inside the zero region, first)
    Bump zer;          // rest of the zero region
    Bump nz;           // non-zero region
    Bump cln;          // self-cleaned pool (its own allocation, see mgpu_ctx)
    std::vector<int> &nrows = P.nrows; nrows.assign(2 * (size_t)n, 0);
    std::vector<PendingRange> prange;
    std::vector<RangeRow> ranges;
    const bool use_range_rows = !getenv("MGPU_NO_RANGE_ROWS");
    std::map<RowKey, int> rowmap;                    // filter operand -> row of its shard's pool
    std::map<const void *, int> pool_rows;           // rows per shard pool
    uint32_t max_nblk = 1, max_N = 1, max_pblocks = 1, max_fk = 1, max_selm = 1, max_tab = 1, max_fac_k = 0, max_fac_nord = 0, max_lds_bins = 0;
    bool sel_wide = false;            // some selection may need the wide (many-workgroup) passes or the stand-alone sorts
    size_t sel_items = 1;             // workgroups those selections can ask for in one wide launch

    std::vector<Offs> &offs = P.offs; offs.resize(2 * (size_t)n);
    std::vector<SelOff> &seloff = P.seloff; seloff.clear();
    std::vector<uint32_t> &sl = P.sl, &dl = P.dl, &fo = P.fo, &pl = P.pl;
    for (int i = 0; i < n_all; i++) {
        // i >= n: the zero-typo twin of squery aux_parent[i - n] (get_matching_zero_typo_docids, squery.c:280-328): the
        // same formula over the postings of the distance-0 words only, no filter; nothing but its bitmap is wanted
        const bool aux = i >= n;
        const int pi = aux ? aux_parent[i - n] : i;
        const mgpu_plan_t &p = plans[pi];
        mgpu_shard *sh = shards[pi];
        if (!sh || sh->device != c->device) return fail(MGPU_E_ARG, "shard not on the context's device");
        if (p.num_terms < 0 || p.num_terms > MGPU_MAX_TERMS || p.num_words > MGPU_MAX_WORDS)
            return fail(MGPU_E_LIMIT, "too many query terms or words");
        if (p.num_terms == 2 || (p.num_terms > 3 && (p.num_terms & 1))) return fail(MGPU_E_ARG, "num_terms must be 0, 1, 3 or even");
        SqDev &q = sqd[i]; SqHost &h = sqh[i];
        memset(&q, 0, sizeof(q)); memset(&h, 0, sizeof(h));
        q.sh = sh->dev;
        const ShardDev &sd = sh->dev;
        q.num_words = p.num_words;
        q.empty = (!aux && p.has_filter && p.filter_null) ? 1 : 0;
        q.aux = aux ? 1 : 0;
        offs[i].aux_index = -1;
        q.nblk = (sd.W + MG_WORDS_PER_BLOCK - 1) / MG_WORDS_PER_BLOCK;
        max_nblk = std::max(max_nblk, q.nblk);
        h.formula_rows = 0; h.scatter_docs = 0; h.sparse_docs = 0;

        auto use_posting = [&](uint32_t pid, int *row_io, std::vector<uint32_t> *dense_list) -> int {
            // returns 0 and either appends a dense row index or schedules a scatter into *row_io
            if (pid >= sd.num_postings) return fail(MGPU_E_ARG, "posting id out of range");
            int32_t di = sh->post_dense[pid];
            if (di >= 0) { dense_list->push_back((uint32_t)di); return MGPU_OK; }
            uint64_t cnt = sh->post_off[pid + 1] - sh->post_off[pid];
            if (!cnt) return MGPU_OK;
            const RowKey key{sh, 0u, pid, 0, 0};
            auto it = rowmap.find(key);
            if (it != rowmap.end()) { *row_io = it->second; return MGPU_OK; }      // another squery of this shard brought it in
            *row_io = pool_rows[sh]++; rowmap[key] = *row_io;
            pscat.push_back({i, *row_io, sd.post_doc + sh->post_off[pid], cnt});
            h.scatter_docs += cnt;
            return MGPU_OK;
        };

        // a term's sparse postings are not expanded into rows: k_formula ORs the part of each list that
        // falls into a workgroup's window straight into an LDS tile (list table below)
        sl.clear();
        auto use_term_posting = [&](uint32_t pid, int t, std::vector<uint32_t> *dense_list) -> int {
            if (pid >= sd.num_postings) return fail(MGPU_E_ARG, "posting id out of range");
            int32_t di = sh->post_dense[pid];
            if (di >= 0) { dense_list->push_back((uint32_t)di); return MGPU_OK; }
            uint64_t cnt = sh->post_off[pid + 1] - sh->post_off[pid];
            if (!cnt) return MGPU_OK;
            sl.push_back((uint32_t)t); sl.push_back((uint32_t)sh->post_off[pid]); sl.push_back((uint32_t)cnt);
            sl.push_back(sh->post_winbase[pid]);
            h.sparse_docs += cnt;
            return MGPU_OK;
        };

        // terms (num_terms == 0 -> every used docid, shard.c:51-53)
        int nt = p.num_terms;
        uint64_t tcard[MGPU_MAX_TERMS + 1] = {0};        // |term t| <= sum of its postings' sizes
        if (nt == 0) {
            q.num_terms = 1;
            q.term_dense_off[0] = (uint32_t)blob.size();
            blob.push_back((uint32_t)sh->post_dense[0]);
            q.term_dense_off[1] = (uint32_t)blob.size();
            q.term_row[0] = -1;
            q.term_sl[0] = q.term_sl[1] = 0;
            h.formula_rows += 1;
        } else {
            q.num_terms = nt;
            for (int t = 0; t < nt; t++) {
                dl.clear();
                const int row = -1;
                uint64_t tc = 0;
                q.term_sl[t] = (uint32_t)(sl.size() / 4);
                if (p.term_nzero[t] > p.term_npost[t]) return fail(MGPU_E_ARG, "term_nzero above term_npost");
                const uint32_t npost_t = aux ? p.term_nzero[t] : p.term_npost[t];
                for (uint32_t j = 0; j < npost_t; j++) {
                    int rc = use_term_posting(p.term_post[t][j], t, &dl);
                    if (rc) return rc;
                    tc += sh->post_card[p.term_post[t][j]];
                }
                q.term_sl[t + 1] = (uint32_t)(sl.size() / 4);
                tcard[t] = std::min<uint64_t>(tc, sd.N);
                std::sort(dl.begin(), dl.end());
                dl.erase(std::unique(dl.begin(), dl.end()), dl.end());
                q.term_dense_off[t] = (uint32_t)blob.size();
                blob.insert(blob.end(), dl.begin(), dl.end());
                q.term_dense_off[t + 1] = (uint32_t)blob.size();
                q.term_row[t] = row;
                h.formula_rows += dl.size() + (row >= 0 ? 1 : 0);
            }
        }
        // Upper bound of the hit count from the match formula alone (a filter only removes hits): the
        // per-hit arrays below are sized by it instead of by the shard.  k_formula / k_order check the real count
        // against it and reports MGPU_E_LIMIT rather than writing past an array if the bound were wrong.
        {
            uint64_t hb = sd.N;
            if (q.empty) hb = 0;
            else if (nt == 1) hb = tcard[0];
            else if (nt == 3) hb = std::min(tcard[0], tcard[1]) + tcard[2];
            else if (nt > 3) {
                uint64_t mn = sd.N;
                for (int g = 0; g < nt; g += 2) {
                    uint64_t gs = tcard[g] + (g > 0 ? tcard[g - 1] : 0) + (g < nt - 2 ? tcard[g + 1] : 0);
                    mn = std::min(mn, gs);
                }
                hb = mn + tcard[nt - 1];
            }
            q.hit_cap = (uint32_t)std::min<uint64_t>(hb, sd.N);
            max_N = std::max(max_N, q.hit_cap);
        }
        // perform_ranking's switch (docrank.c:565-587): could this squery be ranked on a sample?
        if (!aux && !p.full_scan && p.rank_by >= 0 && q.hit_cap > p.full_scan_threshold) {
            if (p.rank_by >= sd.num_numbers) return fail(MGPU_E_ARG, "rank_by field out of range");
            q.fast_eligible = 1; q.fast_thresh = p.full_scan_threshold; q.rank_asc = p.rank_asc ? 1 : 0;
            q.np_rank = sd.np_doc[p.rank_by]; q.np_rank_n = sh->np_n[p.rank_by];
            if ((sh->np_multi >> p.rank_by) & 1u) q.fast_bad |= 2;
            for (int a = 0; a < p.num_aggs && a < MGPU_MAX_AGGS; a++) if (p.agg_type[a] == MGPU_A_CARDINALITY) q.fast_bad |= 1;
            for (int f = 0; f < sd.num_facets; f++) {
                q.fac_pid_base[f] = p.facet_pid_base[f];
                if (p.max_facet_results > 0 && ((p.facet_enabled >> f) & 1u) && sd.fa_nord[f] && !p.facet_pid_base[f]) q.fast_bad |= 4;
            }
            offs[i].aux_index = n_all;
            aux_parent.push_back(i);
            elig_list.push_back((uint32_t)i);
            n_all++;
        }
        // filter program
        if (!aux && p.has_filter && !p.filter_null) {
            int depth = 0;
            q.fops_off = (uint32_t)blob.size();
            q.num_fops = p.num_fops;
            fo.clear();
            for (int j = 0; j < p.num_fops; j++) {
                const mgpu_fop_t &op = p.fops[j];
                int a = 0;
                switch (op.op) {
                    case MGPU_F_PUSH_POSTING: {
                        std::vector<uint32_t> dl; int row = -1;
                        int rc = use_posting((uint32_t)op.a, &row, &dl);
                        if (rc) return rc;
                        if (!dl.empty()) a = (int)dl[0];
                        else if (row >= 0) a = ~row;
                        else { fo.push_back(MGPU_F_PUSH_EMPTY); fo.push_back(0); depth++; continue; }
                        depth++; h.formula_rows++;
                    } break;
                    case MGPU_F_PUSH_NUMRANGE: {
                        if (op.a < 0 || op.a >= sd.num_numbers || op.hi > sh->np_n[op.a] || op.lo > op.hi)
                            return fail(MGPU_E_ARG, "numeric range out of bounds");
                        if (op.hi == op.lo) { fo.push_back(MGPU_F_PUSH_EMPTY); fo.push_back(0); depth++; continue; }
                        const RowKey key{sh, 1u, (uint32_t)op.a, op.lo, op.hi};
                        auto it = rowmap.find(key);
                        int row;
                        if (it != rowmap.end()) row = it->second;
                        else {
                            row = pool_rows[sh]++; rowmap[key] = row;
                            // a long range of a single-valued field comes from the column (RangeRow), the rest is scattered
                            if (use_range_rows && !((sh->np_multi >> op.a) & 1u) && (op.hi - op.lo) * MG_RANGE_ROW_FRACTION > sd.N) {
                                prange.push_back({i, row, op.a, op.lo, op.hi});
                                h.scatter_docs += 2ull * sd.N;                 // (8 B per document of the shard, in the scatter stage's 4-B units)
                            } else {
                                pscat.push_back({i, row, sd.np_doc[op.a] + op.lo, op.hi - op.lo});
                                h.scatter_docs += op.hi - op.lo;
                            }
                        }
                        a = ~row; depth++; h.formula_rows++;
                    } break;
                    case MGPU_F_PUSH_EMPTY: depth++; break;
                    case MGPU_F_AND: case MGPU_F_OR: if (depth < 2) return fail(MGPU_E_ARG, "filter program underflow"); depth--; break;
                    case MGPU_F_DOCS_ANDNOT: if (depth < 1) return fail(MGPU_E_ARG, "filter program underflow"); break;
                    default: return fail(MGPU_E_ARG, "bad filter op");
                }
                if (depth > MG_MAX_STACK) return fail(MGPU_E_LIMIT, "filter nesting deeper than 8");
                fo.push_back((uint32_t)op.op); fo.push_back((uint32_t)a);
            }
            if (depth != 1) return fail(MGPU_E_ARG, "filter program must leave one value");
            q.num_fops = (int)(fo.size() / 2);
            blob.insert(blob.end(), fo.begin(), fo.end());
        }
        // flattened plan for k_formula (staged into LDS by every workgroup)
        {
            pl.clear();
            uint32_t at = 0;
            for (int t = 0; t <= q.num_terms; t++) {
                pl.push_back(at);
                if (t < q.num_terms) at += q.term_dense_off[t + 1] - q.term_dense_off[t];
            }
            for (int t = 0; t < q.num_terms; t++)
                for (uint32_t x = q.term_dense_off[t]; x < q.term_dense_off[t + 1]; x++) {
                    if (blob[x] >= (1u << 27)) return fail(MGPU_E_LIMIT, "more than 2^27 dense rows");
                    pl.push_back(blob[x] | ((uint32_t)t << 27));
                }
            q.plan_fops = (uint32_t)pl.size();
            for (int j = 0; j < q.num_fops; j++) { pl.push_back(blob[q.fops_off + 2 * j]); pl.push_back(blob[q.fops_off + 2 * j + 1]); }
            q.plan_off = (uint32_t)blob.size(); q.plan_len = (uint32_t)pl.size();
            blob.insert(blob.end(), pl.begin(), pl.end());
            while (blob.size() & 3) blob.push_back(0);       // the list table is read as uint4
            q.sl_off = (uint32_t)blob.size();
            blob.insert(blob.end(), sl.begin(), sl.end());
        }
        if (aux) {
            // nothing downstream of the formula looks at a twin: bitmap + count only
            q.do_rank = -1; q.rank_by = -1; q.nproc_blocks = 1;
            Offs &o = offs[i];
            h.hdr_off = res.take(sizeof(ResHdr));
            o.docmap = nz.take((size_t)sd.Wp * 8);
            o.win = nz.take((size_t)((q.nblk + 1) / 2) * sizeof(uint2)); o.rawcnt = zer.take(4, 4);
            o.andor = zer.take(32); o.ovfcnt = zer.take(4);
            continue;
        }
        // wid table
        {
            // sparse while it fits LDS (load <= 1/8: a word that is not in the table almost always meets an
            // empty slot on its first probe, which the rank kernel batches), else load <= 1/2
            uint32_t cap = 2; while (cap < 8 * p.num_wids + 2) cap <<= 1;
            if (cap > MG_LDS_HASH) { cap = 2; while (cap < 2 * p.num_wids + 2) cap <<= 1; }
            if (blob.size() & 1) blob.push_back(0);          // 8-byte aligned slots
            q.hash_off = (uint32_t)blob.size(); q.hash_mask = cap - 1;
            if (cap <= MG_LDS_HASH) max_tab = std::max(max_tab, cap);
            size_t base = blob.size();
            blob.resize(base + (size_t)cap * 2, 0);
            // an empty slot's key is MG_WID_EMPTY, not 0: the zero padding of a record is probed like a word and must
            // meet "empty or another key", never "equal"
            for (uint32_t j = 0; j < cap; j++) blob[base + (size_t)j * 2] = MG_WID_EMPTY;
            for (uint32_t j = 0; j < p.num_wids; j++) {
                const mgpu_wid_entry_t &e = p.wids[j];
                if (e.wid == 0 || e.wid >= 0x80000000u) return fail(MGPU_E_ARG, "wid 0 or above 2^31 in wid table");
                uint32_t hsh = (uint32_t)((uint64_t)(e.wid & 0xFFFFFFu) * MG_WID_HASH_K); hsh ^= hsh >> 15;      // = hash_wid of the kernels
                uint32_t slot = hsh & (cap - 1);
                while (blob[base + slot * 2] != MG_WID_EMPTY && blob[base + slot * 2] != e.wid) slot = (slot + 1) & (cap - 1);
                blob[base + slot * 2] = e.wid;
                blob[base + slot * 2 + 1] = (uint32_t)e.dist | ((uint32_t)e.mterm << 8) | ((uint32_t)e.exact_mask << 16);
            }
        }
        q.exact_present = p.exact_present;
        q.do_rank = (p.hits_per_page > 0 && p.num_words > 0) ? 1 : 0;
        q.rank_by = p.rank_by;
        if (p.rank_by >= sd.num_numbers) return fail(MGPU_E_ARG, "rank_by field out of range");
        int rc = build_key_layout(p, h.kf, &h.nkf);
        if (rc) return rc;
        q.nkf = h.nkf; memcpy(q.kf, h.kf, sizeof(h.kf));
        long kk = (long)p.hits_per_page * (long)p.page_num;
        if (kk < 0) kk = 0;
        if (kk > MG_SEL_FINAL) return fail(MGPU_E_LIMIT, "hitsPerPage * page above 2048");
        h.k = (uint32_t)kk;

        // histograms: enabled facets (when maxFacetResults > 0) and cardinality-agg fields
        uint32_t hmask = 0;
        if (p.max_facet_results > 0) hmask |= p.facet_enabled;
        q.num_aggs = p.num_aggs;
        if (p.num_aggs > MGPU_MAX_AGGS) return fail(MGPU_E_LIMIT, "too many aggregations");
        for (int a = 0; a < p.num_aggs; a++) {
            q.agg_type[a] = p.agg_type[a]; q.agg_field[a] = p.agg_field[a];
            if (p.agg_type[a] == MGPU_A_CARDINALITY) {
                if (p.agg_field[a] < 0 || p.agg_field[a] >= sd.num_facets) return fail(MGPU_E_ARG, "cardinality field out of range");
                hmask |= 1u << p.agg_field[a];
            } else if (p.agg_field[a] < 0 || p.agg_field[a] >= sd.num_numbers) return fail(MGPU_E_ARG, "agg field out of range");
        }
        if (sd.num_facets < 32) hmask &= (1u << sd.num_facets) - 1;
        q.hist_mask = hmask;
        q.nproc_blocks = std::min<uint32_t>(512, std::max<uint32_t>(1, (sd.N + 1023) / 1024));
        max_pblocks = std::max(max_pblocks, q.nproc_blocks);

        // ---- arena layout for this squery
        Offs &o = offs[i];
        h.hdr_off = res.take(sizeof(ResHdr));
        if (h.k) {
            h.rank_hi = res.take((size_t)h.k * 8); h.rank_lo = res.take((size_t)h.k * 8); h.rank_pl = res.take((size_t)h.k * 4);
        }
        h.nfac = 0;
        if (p.max_facet_results > 0) {
            for (int f = 0; f < sd.num_facets; f++) {
                if (!((p.facet_enabled >> f) & 1u) || sd.fa_nord[f] == 0) continue;
                uint32_t fk = std::min<uint32_t>((uint32_t)p.max_facet_results * 2u, sd.fa_nord[f]);
                if (fk > MG_SEL_FINAL) return fail(MGPU_E_LIMIT, "2*maxFacetResults (capped by distinct values) above 2048");
                int j = h.nfac++;
                h.fac_field[j] = f; h.fac_k[j] = fk;
                max_fac_k = std::max(max_fac_k, fk); max_fac_nord = std::max(max_fac_nord, sd.fa_nord[f]);
                h.fac_hi[j] = res.take((size_t)fk * 8); h.fac_lo[j] = res.take((size_t)fk * 8); h.fac_pl[j] = res.take((size_t)fk * 4);
            }
        }
        o.rows = 0;   // filled after nrows is final (below)
        for (int f = 0; f < sd.num_facets; f++) if ((hmask >> f) & 1u) {
            o.hist[f] = cln.take((size_t)sd.fa_nord[f] * 4 + 4);
            if (sd.fa_nord[f] <= MG_LDS_HIST) max_lds_bins = std::max(max_lds_bins, sd.fa_nord[f]);
            o.ntouched[f] = zer.take(4, 4);
            o.touched[f] = nz.take((size_t)sd.fa_nord[f] * 4 + 4);
            o.facord[f] = nz.take((size_t)std::max<uint32_t>(q.hit_cap, 1) * 4);
        }
        o.andor = zer.take(32); o.ovfcnt = zer.take(4);
        // the dense docid_map exists only for the parity capture (k_formula emits the hits itself)
        o.docmap = ((p.capture & 1) || q.fast_eligible) ? nz.take((size_t)sd.Wp * 8) : (size_t)-1;
        if (q.fast_eligible) o.dbmap = nz.take((size_t)sd.Wp * 8);
        if ((q.nblk + 1) / 2 > MG_ORDER_WINDOWS) return fail(MGPU_E_LIMIT, "shard larger than 2048 formula windows (67M docids)");
        o.win = nz.take((size_t)((q.nblk + 1) / 2) * sizeof(uint2)); o.rawcnt = zer.take(4, 4);
        const size_t hc = std::max<size_t>(q.hit_cap, 1);
        o.hitsraw = nz.take(hc * 4);
        o.hits = nz.take(hc * 4);
        o.khi = nz.take(hc * 8); o.klo = nz.take(hc * 8);
        o.ovfl = nz.take(hc * 4);
        o.aggpart = nz.take((size_t)q.nproc_blocks * std::max(1, p.num_aggs) * sizeof(AggPart));
        for (int j = 0; j < h.nfac; j++) {
            int f = h.fac_field[j];
            // (a directly counted field lists one entry per hit instead of one per distinct value)
            const size_t nkeys = ((sh->fac_single >> f) & 1u) ? std::max<size_t>(sd.fa_nord[f], q.hit_cap) : sd.fa_nord[f];
            o.fkhi[f] = nz.take(nkeys * 8); o.fklo[f] = nz.take(nkeys * 8);
            max_fk = std::max(max_fk, sd.fa_nord[f]);
        }
        h.ncard = 0;
        for (int a = 0; a < p.num_aggs; a++) {
            if (p.agg_type[a] != MGPU_A_CARDINALITY) continue;
            int j = h.ncard++;
            h.card_agg[j] = a; h.card_cap[j] = sd.fa_nord[p.agg_field[a]];
            // a small field's id set rides in the results region: one download per batch, where a copy of its own per squery
            // (131 000 of them over the 4096-query stream with a cardinality aggregation) held the rate at 7.8k queries/s
            h.card_in_res[j] = h.card_cap[j] <= MG_CARD_IN_RES;
            h.card_ids_off[j] = h.card_in_res[j] ? res.take((size_t)h.card_cap[j] * 4 + 4, 8) : nz.take((size_t)h.card_cap[j] * 4 + 4);
            max_fk = std::max(max_fk, sd.fa_nord[p.agg_field[a]]);
        }
        // selection jobs
        h.rank_job = -1;
        if (h.k) {
            h.rank_job = (int)seloff.size();
            SelOff so; so.mcap = q.hit_cap;
            so.cand0 = nz.take(hc * 4); so.cand1 = nz.take(hc * 4);
            so.hist = cln.take(MG_SEL_BINS * 4); so.ctl = zer.take(SEL_WORDS * 4); so.outidx = nz.take((size_t)MG_SEL_FINAL * 4);
            seloff.push_back(so); jobfix.push_back({i, 0, 0});
            max_selm = std::max(max_selm, q.hit_cap);
            if (q.hit_cap > MG_SEL_MID_KEYS || h.k > MG_SEL_SOLO_K) { sel_wide = true; sel_items += std::min<size_t>(64, ((size_t)q.hit_cap + 2047) / 2048); }
        }
        for (int j = 0; j < h.nfac; j++) {
            int f = h.fac_field[j];
            h.fac_job[j] = (int)seloff.size();
            SelOff so; so.mcap = sd.fa_nord[f];
            so.cand0 = nz.take((size_t)so.mcap * 4); so.cand1 = nz.take((size_t)so.mcap * 4);
            so.hist = cln.take(MG_SEL_BINS * 4); so.ctl = zer.take(SEL_WORDS * 4); so.outidx = nz.take((size_t)MG_SEL_FINAL * 4);
            so.andor = zer.take(32, 8);
            seloff.push_back(so); jobfix.push_back({i, 1, j});
            max_selm = std::max(max_selm, so.mcap);
            // (a field counted straight from the rank kernel is selected over one key per hit)
            if (std::max(so.mcap, q.hit_cap) > MG_SEL_MID_KEYS || h.fac_k[j] > MG_SEL_SOLO_K) {
                sel_wide = true; sel_items += std::min<size_t>(64, ((size_t)std::max(so.mcap, q.hit_cap) + 2047) / 2048);
            }
        }
    }
    // batch-wide work list of hit tiles, filled by k_order
    size_t tile_cap = 0;
    for (int i = 0; i < n; i++) tile_cap += ((size_t)sqd[i].hit_cap + MG_BLOCK - 1) / MG_BLOCK;
    if (tile_cap > 0xFFFFFFFFull) return fail(MGPU_E_LIMIT, "batch too large for the tile work list");
    const size_t off_tilecnt = zer.take(4, 4);
    const size_t off_todocnt = zer.take(8 * 4, 8);            // select work-list counters
    const size_t off_tiles = nz.take(std::max<size_t>(tile_cap, 1) * sizeof(uint2));
    // batches with fast-eligible squeries: a second work list, built once the samples are known
    const size_t off_tilecnt_b = zer.take(4, 4);
    const size_t off_tiles_b = elig_list.empty() ? 0 : nz.take(std::max<size_t>(tile_cap, 1) * sizeof(uint2));
    const size_t off_todo = nz.take(2 * (seloff.size() + 1) * 4);                         // final list, mid list
    const size_t off_fhcnt = zer.take(2 * 4, 8);                                          // k_facet_hash job lists: counters ...
    const size_t off_fhlist = nz.take(2 * (seloff.size() + 1) * 4);                       // ... and the two lists
    // A job never asks for more than min(64, ceil(capacity / 2048)) workgroups per wide launch, and only the jobs counted
    // above can ask at all (the others finish in k_select_first / k_select_mid): the lists, and with them the grids of
    // the wide launches, are sized for those (the bench stream, where the hit BOUNDS of a few dozen squeries per batch
    // exceed one workgroup's reach, still fills the 2048 workgroups a wide launch is capped at: no change there).  Should
    // a job find no room all the same, the host-driven rounds pick it up.
    const uint32_t sel_cap_items = (uint32_t)sel_items;
    const size_t off_items = nz.take((size_t)sel_cap_items * 4 * sizeof(uint2));
    // scratch rows (zero region), now that row counts are known
    // filter operand rows: one pool per shard of the batch, every squery of the shard addresses it
    {
        std::map<const void *, size_t> pool_off;
        for (const auto &kv : pool_rows) pool_off[kv.first] = zer.take((size_t)kv.second * ((const mgpu_shard *)kv.first)->dev.Wp * 8);
        for (int i = 0; i < n; i++) { auto it = pool_off.find(shards[i]); offs[i].rows = it == pool_off.end() ? 0 : it->second; }
    }

    // scatter work items
    std::vector<ScatterItem> &items = P.items; items.clear();
    // region bases
    const size_t res_bytes = (res.off + 255) / 256 * 256;
    const size_t zero_bytes = res_bytes + ((zer.off + 255) / 256 * 256);
    const size_t nz_base = zero_bytes;
    const size_t nz_bytes = (nz.off + 255) / 256 * 256;
    // upload region layout
    Bump up;
    const size_t up_sq = up.take(sizeof(SqDev) * (size_t)n_all);
    const size_t up_blob = up.take(std::max<size_t>(blob.size(), 1) * 4);
    size_t nitems = 0;
    for (auto &ps : pscat) nitems += (ps.n + MG_SCATTER_CHUNK - 1) / MG_SCATTER_CHUNK;
    const size_t up_items = up.take(std::max<size_t>(nitems, 1) * sizeof(ScatterItem));
    const size_t up_ranges = up.take(std::max<size_t>(prange.size(), 1) * sizeof(RangeRow));
    const size_t up_sel = up.take(std::max<size_t>(seloff.size(), 1) * sizeof(SelJob));
    const size_t up_selout = up.take(std::max<size_t>(seloff.size(), 1) * sizeof(SelOut));
    size_t nfk = 0, ncard = 0;
    for (int i = 0; i < n; i++) { nfk += sqh[i].nfac; ncard += sqh[i].ncard; }
    const size_t up_fk = up.take(std::max<size_t>(nfk, 1) * sizeof(FkJob));
    const size_t up_card = up.take(std::max<size_t>(ncard, 1) * sizeof(CardJob));
    size_t nfac_elig = 0;
    for (uint32_t e : elig_list) nfac_elig += sqh[e].nfac;
    const size_t up_elig = up.take(std::max<size_t>(elig_list.size(), 1) * 4);
    const size_t up_recount = up.take(std::max<size_t>(nfac_elig, 1) * sizeof(RecountJob));
    const size_t up_bytes = (up.off + 255) / 256 * 256;
    const size_t up_base = nz_base + nz_bytes;
    const size_t total = up_base + up_bytes;

    int rc = ensure_arena(c, total);
    if (rc) return rc;
    if ((rc = ensure_pin(&c->pin, &c->pin_cap, up_bytes))) return rc;
    const size_t cln_bytes = (cln.off + 255) / 256 * 256;
    if (cln_bytes > c->cln.cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->cln.dev) HIPCHK(hipFree(c->cln.dev));
        c->cln.dev = nullptr; c->cln.cap = 0; c->cln_dirty = true;
        const size_t cap = std::max(cln_bytes + cln_bytes / 4, (size_t)(1u << 20));
        HIPCHK(hipMalloc((void **)&c->cln.dev, cap));
        c->cln.cap = cap;
    }
    char *CL = c->cln.dev;
    if ((rc = ensure_pin(&c->pin_out, &c->pin_out_cap, res_bytes))) return rc;
    char *A = c->arena.dev;
    char *R = A;                       // results region
    char *Z = A + res_bytes;           // remaining zero region
    char *NZ = A + nz_base;
    char *UP = A + up_base;

    // fix up device pointers
    for (int i = n; i < n_all; i++) {                      // the zero-typo twins
        SqDev &q = sqd[i]; SqHost &h = sqh[i]; Offs &o = offs[i];
        ResHdr *hdr = (ResHdr *)(R + h.hdr_off);
        q.rows = nullptr;
        q.docid_map = (uint64_t *)(NZ + o.docmap);
        q.win_info = (uint2 *)(NZ + o.win); q.raw_count = (uint32_t *)(Z + o.rawcnt); q.hits_raw = nullptr; q.hits = nullptr;
        q.num_hits = &hdr->num_hits; q.err = &hdr->err; q.ovf_cnt = (uint32_t *)(Z + o.ovfcnt);
        q.key_andor = (uint64_t *)(Z + o.andor);
    }
    for (int i = 0; i < n; i++) {
        SqDev &q = sqd[i]; SqHost &h = sqh[i]; Offs &o = offs[i];
        ResHdr *hdr = (ResHdr *)(R + h.hdr_off);
        if (q.fast_eligible) {
            const SqDev &tw = sqd[o.aux_index];
            q.zmap = (const uint64_t *)(NZ + offs[o.aux_index].docmap);
            q.z_count = &((ResHdr *)(R + sqh[o.aux_index].hdr_off))->num_hits;
            (void)tw;
            q.dbmap = (uint64_t *)(NZ + o.dbmap);
            q.fast_flag = &hdr->fast_flag; q.total_hits = &hdr->total_hits; q.num_rank = &hdr->num_rank;
            sqd[o.aux_index].fast_flag = q.fast_flag;          // the twin looks at it too: no work for the twin of a squery that is not sampled
        }
        q.rows = (uint64_t *)(Z + o.rows);
        q.docid_map = o.docmap == (size_t)-1 ? nullptr : (uint64_t *)(NZ + o.docmap);
        q.win_info = (uint2 *)(NZ + o.win); q.raw_count = (uint32_t *)(Z + o.rawcnt); q.hits_raw = (uint32_t *)(NZ + o.hitsraw);
        q.num_hits = &hdr->num_hits; q.err = &hdr->err; q.ovf_cnt = (uint32_t *)(Z + o.ovfcnt);
        q.hits = (uint32_t *)(NZ + o.hits);
        q.key_hi = (uint64_t *)(NZ + o.khi); q.key_lo = (uint64_t *)(NZ + o.klo);
        q.key_andor = (uint64_t *)(Z + o.andor);
        q.ovf_list = (uint32_t *)(NZ + o.ovfl);
        for (int f = 0; f < q.sh.num_facets; f++) if ((q.hist_mask >> f) & 1u) {
            q.hist[f] = (uint32_t *)(CL + o.hist[f]);
            q.ntouched[f] = (uint32_t *)(Z + o.ntouched[f]);
            q.touched[f] = (uint32_t *)(NZ + o.touched[f]);
            q.fac_ord[f] = (uint32_t *)(NZ + o.facord[f]);
        }
        q.agg_part = (AggPart *)(NZ + o.aggpart);
        q.agg_out = hdr->agg;
        h.docid_map_off = o.docmap == (size_t)-1 ? 0 : nz_base + o.docmap; h.hits_off = nz_base + o.hits;
        h.key_hi_off = nz_base + o.khi; h.key_lo_off = nz_base + o.klo;
    }
    for (auto &ps : pscat) {
        SqDev &q = sqd[ps.sq];
        uint64_t *dst = q.rows + (size_t)ps.row * q.sh.Wp;
        for (uint64_t at = 0; at < ps.n; at += MG_SCATTER_CHUNK) {
            ScatterItem it; it.src = ps.src_dev + at; it.dst = dst;
            it.n = (uint32_t)std::min<uint64_t>(MG_SCATTER_CHUNK, ps.n - at); it.pad = 0;
            items.push_back(it);
        }
    }
    uint32_t range_max_w = 0;
    for (auto &pr : prange) {
        SqDev &q = sqd[pr.sq];
        RangeRow r; r.col = q.sh.num_col[pr.field]; r.dst = q.rows + (size_t)pr.row * q.sh.Wp; r.order = q.sh.np_doc[pr.field];
        r.lo = pr.lo; r.hi = pr.hi; r.N = q.sh.N; r.W = q.sh.W;
        ranges.push_back(r);
        range_max_w = std::max(range_max_w, r.W);
    }
    // rows of one column next to each other: a group of them shares the column's loads (k_range_rows)
    std::stable_sort(ranges.begin(), ranges.end(), [](const RangeRow &a, const RangeRow &b) { return (uintptr_t)a.col < (uintptr_t)b.col; });
    // need_keys: the rank kernel has nothing to do when no hits are requested
    for (int i = 0; i < n; i++) if (!sqh[i].k) sqd[i].do_rank = -1;
    // facet counting rides on the rank kernel's record visit whenever that kernel reads the records
    {
        const bool fuse = !getenv("MGPU_NO_FUSED_FACETS");
        for (int i = 0; i < n; i++) {
            SqDev &q = sqd[i];
            const bool reads_records = q.do_rank == 1 || (q.do_rank == 0 && q.rank_by >= 0 && (uint32_t)q.rank_by < q.sh.rec_nnum);
            q.fac_in_rank = (fuse && q.hist_mask && reads_records) ? 1 : 0;
        }
    }

    // facet keys: when every facet list of the batch fits the one-workgroup selections, those derive
    // the keys from the touched bins themselves and k_fkeys is not launched
    const bool fuse_fk = max_fac_nord <= MG_SEL_MID_KEYS && max_fac_k <= MG_SEL_SOLO_K && !getenv("MGPU_NO_FUSED_FKEYS");
    const bool use_fdirect = !getenv("MGPU_NO_DIRECT_FACETS");
    const bool use_fhash = !getenv("MGPU_NO_HASH_FACETS");
    // The bound is on hit_cap, which the host computes from posting cardinalities and is often far above the hit count: a
    // squery above it falls back to per-hit returning atomics in k_docproc and, for a 100k-value field, to lists the wide
    // select launches take -- although it may have a few thousand hits.  MGPU_FDIRECT_CAP lifts the bound (the kernels
    // cope: tests/test_gpu_fastrank.py); measured on the bench stream, facets+aggs 0.083 -> 0.036 ms but select 0.37 -> 0.43
    // (lists of one key per HIT instead of one per touched bin for the squeries of 50-100k hits), so the default stays.
    const uint32_t fdirect_max_cap = getenv("MGPU_FDIRECT_CAP") ? (uint32_t)strtoul(getenv("MGPU_FDIRECT_CAP"), nullptr, 10) : MG_FDIRECT_MAX_CAP;
    uint32_t nfhash = 0;
    seljobs.resize(seloff.size()); selouts.resize(seloff.size());
    for (size_t j = 0; j < seloff.size(); j++) {
        const SelOff &so = seloff[j]; const JobFix &jf = jobfix[j];
        SqDev &q = sqd[jf.sq]; SqHost &h = sqh[jf.sq];
        ResHdr *hdr = (ResHdr *)(R + h.hdr_off);
        SelJob &sj = seljobs[j]; SelOut &so2 = selouts[j];
        memset(&sj, 0, sizeof(sj));
        sj.cand[0] = (uint32_t *)(NZ + so.cand0); sj.cand[1] = (uint32_t *)(NZ + so.cand1);
        sj.hist = (uint32_t *)(CL + so.hist); sj.ctl = (uint32_t *)(Z + so.ctl);
        sj.out_idx = (uint32_t *)(NZ + so.outidx); sj.m_cap = so.mcap;
        if (jf.kind == 0) {
            sj.key_hi = q.key_hi; sj.key_lo = q.key_lo; sj.m_ptr = q.fast_eligible ? q.num_rank : q.num_hits; sj.k = h.k;
            sj.out_n = &hdr->rank_n; sj.mask = q.key_andor; sj.skip_ones = 0;
            so2.out_hi = (u64 *)(R + h.rank_hi); so2.out_lo = (u64 *)(R + h.rank_lo);
            so2.out_pl = (uint32_t *)(R + h.rank_pl); so2.payload = q.hits;
        } else {
            int f = h.fac_field[jf.field];
            Offs &o = offs[jf.sq];
            q.fkey_hi[f] = (uint64_t *)(NZ + o.fkhi[f]); q.fkey_lo[f] = (uint64_t *)(NZ + o.fklo[f]);
            sj.key_hi = q.fkey_hi[f]; sj.key_lo = q.fkey_lo[f];
            sj.m_ptr = q.ntouched[f];
            sj.k = h.fac_k[jf.field]; sj.out_n = &hdr->fac_n[jf.field]; sj.mask = (const uint64_t *)(Z + so.andor); sj.skip_ones = 1;
            so2.out_hi = (u64 *)(R + h.fac_hi[jf.field]); so2.out_lo = (u64 *)(R + h.fac_lo[jf.field]);
            so2.out_pl = (uint32_t *)(R + h.fac_pl[jf.field]); so2.payload = nullptr;
            if (q.fast_eligible) recount.push_back({(uint32_t)jf.sq, (uint32_t)f, so2.out_hi, so2.out_lo, &hdr->fac_n[jf.field]});
            FkJob fj; fj.hist = q.hist[f]; fj.ord2id = q.sh.fa_ord2id[f]; fj.touched = q.touched[f]; fj.ntouched = q.ntouched[f];
            fj.key_hi = (u64 *)q.fkey_hi[f]; fj.key_lo = (u64 *)q.fkey_lo[f];
            fj.andor = (u64 *)(Z + so.andor);
            if (fuse_fk) { sj.gen_touched = fj.touched; sj.gen_hist = fj.hist; sj.gen_ord2id = fj.ord2id; }
            else fkjobs.push_back(fj);
            // a sparse field (too many bins for an LDS histogram) of single-valued documents: counted straight from
            // the rank kernel, the per-hit ordinals are the touched list (SqDev::fdirect_mask)
            bool card_on_f = false;
            for (int a = 0; a < q.num_aggs; a++) card_on_f |= q.agg_type[a] == MGPU_A_CARDINALITY && q.agg_field[a] == f;
            if (fuse_fk && use_fdirect && q.fac_in_rank && ((shards[jf.sq]->fac_single >> f) & 1u) && q.sh.fa_nord[f] > MG_LDS_HIST &&
                !card_on_f && q.hit_cap <= fdirect_max_cap) {
                q.fdirect_mask |= 1u << f;
                sj.gen_touched = q.fac_ord[f]; sj.m_ptr = q.num_hits;
                // ... and with few hits not even the histogram: k_facet_hash counts the ordinals in LDS.  A squery that may
                // be ranked on a sample (fast_eligible) qualifies when its threshold is at least MG_FHASH_MAX_M: "few enough hits
                // for hashing" then implies "not sampled", so k_order (which files the job by the hit count BEFORE a sample
                // replaces the list) and the kernels that skip their own counting for hashed squeries (they see the count
                // AFTER it) cannot disagree.  With a lower threshold a sampled squery would fall between the two and nobody
                // would count the field (found in round 3): no hash jobs for such a squery.
                if (use_fhash && (!q.fast_eligible || q.fast_thresh >= MG_FHASH_MAX_M) && sj.k <= MG_FHASH_MAX_K && q.fhash_njobs < 4) {
                    q.fhash_mask |= 1u << f; sj.fhash = 1; q.fhash_jobs[q.fhash_njobs++] = (uint32_t)j; nfhash++;
                }
            }
        }
    }
    for (int i = 0; i < n; i++) {
        SqHost &h = sqh[i]; SqDev &q = sqd[i];
        ResHdr *hdr = (ResHdr *)(R + h.hdr_off);
        for (int j = 0; j < h.ncard; j++) {
            int f = q.agg_field[h.card_agg[j]];
            CardJob cj; cj.ord2id = q.sh.fa_ord2id[f]; cj.touched = q.touched[f]; cj.ntouched = q.ntouched[f];
            cj.out_ids = (uint32_t *)(h.card_in_res[j] ? R + h.card_ids_off[j] : A + nz_base + h.card_ids_off[j]); cj.out_n = &hdr->card_n[j];
            bool has_facet_job = false, seen = false;
            for (int x = 0; x < h.nfac; x++) has_facet_job |= h.fac_field[x] == f;
            for (int x = 0; x < j; x++) seen |= q.agg_field[h.card_agg[x]] == f;          // two cardinality aggs on one field
            cj.clean_hist = (has_facet_job || seen) ? nullptr : q.hist[f];
            cardjobs.push_back(cj);
        }
    }
    // staging copy
    memcpy(c->pin + up_sq, sqd, sizeof(SqDev) * (size_t)n_all);
    if (!elig_list.empty()) memcpy(c->pin + up_elig, elig_list.data(), elig_list.size() * 4);
    if (!recount.empty()) memcpy(c->pin + up_recount, recount.data(), recount.size() * sizeof(RecountJob));
    if (!blob.empty()) memcpy(c->pin + up_blob, blob.data(), blob.size() * 4);
    if (!items.empty()) memcpy(c->pin + up_items, items.data(), items.size() * sizeof(ScatterItem));
    if (!ranges.empty()) memcpy(c->pin + up_ranges, ranges.data(), ranges.size() * sizeof(RangeRow));
    if (!seljobs.empty()) { memcpy(c->pin + up_sel, seljobs.data(), seljobs.size() * sizeof(SelJob));
                            memcpy(c->pin + up_selout, selouts.data(), selouts.size() * sizeof(SelOut)); }
    if (!fkjobs.empty()) memcpy(c->pin + up_fk, fkjobs.data(), fkjobs.size() * sizeof(FkJob));
    if (!cardjobs.empty()) memcpy(c->pin + up_card, cardjobs.data(), cardjobs.size() * sizeof(CardJob));

    if (getenv("MGPU_DEBUG")) {
        uint64_t wids = 0; for (int i = 0; i < n; i++) wids += plans[i].num_wids;
        fprintf(stderr, "[mgpu] n=%d zero=%.1fMB scratch=%.1fMB upload=%.2fMB results=%.2fMB max_tab=%u slots mean_wids=%.1f sel_jobs=%zu sizeof(SqDev)=%zu\n",
                n, zero_bytes / 1e6, nz_bytes / 1e6, up_bytes / 1e6, res_bytes / 1e6, max_tab, (double)wids / n, seloff.size(), sizeof(SqDev));
    }
    // ---------------- launches ----------------
    const double t_lowered = wall_ms();
    hipStream_t st = c->stream;
    const SqDev *d_sq = (const SqDev *)(UP + up_sq);
    const uint32_t *d_blob = (const uint32_t *)(UP + up_blob);
    const bool prof = c->profiling != 0;
    int evi = 0;
    auto mark = [&]() { if (prof) (void)hipEventRecord(c->ev[evi++], st); };
    // MGPU_TRACE_SUBMIT=1: wall clock after every enqueue step, printed when the launch half took more than 2 ms
    static const bool trace_submit = getenv("MGPU_TRACE_SUBMIT") != nullptr;
    double tt[12]; int ntt = 0;
    auto tick = [&]() { if (trace_submit && ntt < 12) tt[ntt++] = wall_ms(); };
    tick();
    // Copies.  On this runtime a hipMemcpyAsync enqueued behind kernels of its own stream holds the calling host
    // thread until those kernels have run (measured: 4-11 ms with eight batches in flight, profiles/README.md), a
    // kernel launch returns at once.  So the upload goes FIRST, while the stream is still idle (the previous batch
    // of this context was waited for), and the results are written to pinned host memory by a kernel (mgk_fetch).
    // MGPU_COPY_MODE: 0 = both by hipMemcpyAsync (round 1), 1 = upload by hipMemcpyAsync first, results by kernel
    // (default), 2 = both by kernel.
    static const int copy_mode = getenv("MGPU_COPY_MODE") ? atoi(getenv("MGPU_COPY_MODE")) : 1;
    (void)hipEventRecord(c->ev[0], st); evi = 1;
    if (copy_mode == 1) HIPCHK(hipMemcpyAsync(UP, c->pin, up_bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(A, 0, zero_bytes, st));
    if (c->cln_dirty && c->cln.cap) HIPCHK(hipMemsetAsync(CL, 0, c->cln.cap, st));
    c->cln_dirty = true;                  // until this batch has run to its end
    HIPCHK(hipMemsetAsync(c->pool_used, 0, 4, st));
    tick();
    if (copy_mode == 0) HIPCHK(hipMemcpyAsync(UP, c->pin, up_bytes, hipMemcpyHostToDevice, st));
    else if (copy_mode == 2) mgk_fetch(st, UP, c->pin, up_bytes);             // up_bytes is a multiple of 256, both buffers 256-byte aligned
    mark();
    tick();
    mgk_scatter(st, (const ScatterItem *)(UP + up_items), (uint32_t)items.size());
    mgk_range_rows(st, (const RangeRow *)(UP + up_ranges), (uint32_t)ranges.size(), range_max_w);
    mark();
    uint2 *d_tiles = (uint2 *)(NZ + off_tiles); uint32_t *d_tilecnt = (uint32_t *)(Z + off_tilecnt);
    uint32_t *d_fhlist = (uint32_t *)(NZ + off_fhlist), *d_fhcnt = (uint32_t *)(Z + off_fhcnt);
    mgk_formula(st, d_sq, d_blob, max_nblk, (uint32_t)n_all, (uint32_t)n, d_tiles, d_tilecnt, (uint32_t)tile_cap, d_fhlist, d_fhcnt, nfhash);
    if (!elig_list.empty()) {
        // fast-rank (docrank.c:478-560): aggregations over ALL hits of the squeries k_order flagged, then their hit
        // lists become the samples and the work list is rebuilt
        bool any_aggs = false;
        for (uint32_t e : elig_list) any_aggs |= sqd[e].num_aggs > 0;
        if (any_aggs) mgk_docproc(st, d_sq, (uint32_t)n, d_tiles, d_tilecnt, (uint32_t)tile_cap, 0, 1);
        mgk_fast_build(st, d_sq, (uint32_t)n, (const uint32_t *)(UP + up_elig), (uint32_t)elig_list.size(),
                       (uint2 *)(NZ + off_tiles_b), (uint32_t *)(Z + off_tilecnt_b));
        d_tiles = (uint2 *)(NZ + off_tiles_b); d_tilecnt = (uint32_t *)(Z + off_tilecnt_b);
    }
    mark();
    tick();
    bool rank_len = true;                  // every shard of the batch lets the rank kernel pack record lengths
    for (int i = 0; i < n; i++) rank_len = rank_len && shards[i]->rec_len_ok;
    mgk_rank(st, d_sq, d_blob, max_N, (uint32_t)n, max_tab, rank_len ? 1 : 0, d_tiles, d_tilecnt, (uint32_t)tile_cap, c->pool, c->pool_used, c->pool_cap);
    mark();
    mgk_docproc(st, d_sq, (uint32_t)n, d_tiles, d_tilecnt, (uint32_t)tile_cap, max_lds_bins, 0);
    mark();
    tick();
    mgk_fkeys(st, UP + up_fk, (uint32_t)fkjobs.size(), max_fk);
    mgk_card(st, UP + up_card, (uint32_t)cardjobs.size(), max_fk);
    mark();
    mgk_facet_hash(st, (const SelJob *)(UP + up_sel), UP + up_selout, d_fhlist, d_fhcnt, nfhash);
    mgk_select(st, (const SelJob *)(UP + up_sel), UP + up_selout, (uint32_t)seljobs.size(), max_selm, MG_SEL_PASSES, 0,
               NZ + off_items, sel_wide ? sel_cap_items : 0u, (uint32_t *)(Z + off_todocnt), (uint32_t *)(NZ + off_todo));
    mgk_facet_recount(st, d_sq, UP + up_recount, (uint32_t)recount.size(), max_fac_k);
    mark();
    tick();
    if (copy_mode == 0) HIPCHK(hipMemcpyAsync(c->pin_out, R, res_bytes, hipMemcpyDeviceToHost, st));
    else mgk_fetch(st, c->pin_out, R, res_bytes);          // results written to pinned host memory by the GPU
    if (prof) { (void)hipEventRecord(c->ev[evi++], st); } else { (void)hipEventRecord(c->ev[1], st); }
    tick();
    HIPCHK(hipGetLastError());
    if (trace_submit && ntt >= 7 && tt[6] - t_begin > 2.0)
        fprintf(stderr, "[submit] lower %.3f | memsets %.3f h2d %.3f scatter+formula %.3f rank+docproc %.3f fkeys+select %.3f d2h %.3f ms\n",
                tt[0] - t_begin, tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4], tt[6] - tt[5]);
    P.res_bytes = res_bytes; P.nz_base = nz_base; P.up_bytes = up_bytes; P.up_base = up_base; P.up_sel = up_sel; P.up_selout = up_selout;
    P.max_selm = max_selm; P.fuse_fk = fuse_fk; P.prof = prof; P.nev = evi;
    P.t_begin = t_begin; P.t_lowered = t_lowered;
    c->host_ms[0] = (float)(t_lowered - t_begin);
    return MGPU_OK;
}

static int execute_end(mgpu_ctx_t *c, mgpu_sqout_t *outs, bool *completed) {
    Pending &P = *c->pend;
    const int n = P.n;
    mgpu_shard_t *const *shards = P.shards; const mgpu_plan_t *plans = P.plans;
    SqDev *sqd = P.sqd.data();
    std::vector<SqHost> &sqh = P.sqh;
    std::vector<SelOff> &seloff = P.seloff; std::vector<JobFix> &jobfix = P.jobfix; std::vector<SelJob> &seljobs = P.seljobs;
    const size_t res_bytes = P.res_bytes, nz_base = P.nz_base, up_bytes = P.up_bytes, up_sel = P.up_sel, up_selout = P.up_selout;
    const uint32_t max_selm = P.max_selm;
    const bool fuse_fk = P.fuse_fk, prof = P.prof;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    char *A = c->arena.dev, *R = A, *UP = A + P.up_base, *CL = c->cln.dev;
    const double t_wait0 = wall_ms();
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());

    // jobs the first pass launch did not finish (lists of more than 32768 keys, or final sets above
    // MG_SEL_SMALL): continuation rounds of passes + stand-alone final sorts
    for (int round = 0; round < 20; round++) {
        bool again = false;
        for (size_t j = 0; j < seloff.size() && !again; j++) {
            const JobFix &jf = jobfix[j]; SqHost &h = sqh[jf.sq];
            ResHdr *hdr = (ResHdr *)(c->pin_out + h.hdr_off);
            uint32_t on = jf.kind == 0 ? hdr->rank_n : hdr->fac_n[jf.field];
            if (on == 0xFFFFFFFFu) again = true;
        }
        if (!again) break;
        mgk_select(st, (const SelJob *)(UP + up_sel), UP + up_selout, (uint32_t)seljobs.size(), max_selm, MG_SEL_PASSES, 1, nullptr, 0, nullptr, nullptr);
        HIPCHK(hipMemcpyAsync(c->pin_out, R, res_bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }

    if (prof) {
        float tot = 0;
        for (int sidx = 0; sidx < MGPU_NUM_STAGES; sidx++) {
            float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[sidx], c->ev[sidx + 1]);
            c->stage_ms[sidx] = ms; tot += ms;
        }
        c->total_ms = tot;
    } else {
        float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
        c->total_ms = ms;
    }

    // ---------------- decode ----------------
    const double t_waited = wall_ms();
    memset(c->stage_bytes, 0, sizeof(c->stage_bytes));
    int worst = MGPU_OK;
    for (int i = 0; i < n; i++) {
        const mgpu_plan_t &p = plans[i];
        SqHost &h = sqh[i]; SqDev &q = sqd[i]; mgpu_shard *sh = shards[i];
        mgpu_sqout_t &o = outs[i];
        const ResHdr *hdr = (const ResHdr *)(c->pin_out + h.hdr_off);
        const bool fast = q.fast_eligible && hdr->fast_flag;
        o.num_hits = fast ? hdr->total_hits : hdr->num_hits;
        o.fast_rank = fast ? 1 : 0;
        if (hdr->err >= 0x40000000u) {
            // a checked accessor refused an access (mgpu_kernels.hip: CK sites)
            char msg[96]; snprintf(msg, sizeof(msg), "checked accessor: site %u, index (low 24 bits) %u", (hdr->err >> 24) & 0x3Fu, hdr->err & 0xFFFFFFu);
            o.status = MGPU_E_HIP; worst = o.status; g_err = msg; continue;
        }
        if (hdr->err) { o.status = -(int)hdr->err; worst = o.status; g_err = hdr->err == (uint32_t)(-MGPU_E_LIMIT) ? (fast ? "fast-rank: the sample outgrows the reference's rank buffer, or rank_by has several values per document" : "hit count above the planned bound")
                                                                     : hdr->err == (uint32_t)(-MGPU_E_UNSUPPORTED) ? "fast-rank with a cardinality aggregation (the reference dereferences NULL there) or without facet_pid_base"
                                                                     : "device reported an error (position scratch pool exhausted)"; continue; }
        const uint64_t M = hdr->num_hits;
        // ranks
        uint32_t rn = h.k ? hdr->rank_n : 0;
        if (rn == 0xFFFFFFFFu) { o.status = MGPU_E_LIMIT; worst = o.status; g_err = "selection did not converge"; continue; }
        o.rank_count = (int)rn;
        o.ranks = (mgpu_rank_t *)calloc(std::max<uint32_t>(rn, 1), sizeof(mgpu_rank_t));
        const u64 *rhi = (const u64 *)(c->pin_out + h.rank_hi), *rlo = (const u64 *)(c->pin_out + h.rank_lo);
        const uint32_t *rpl = (const uint32_t *)(c->pin_out + h.rank_pl);
        for (uint32_t j = 0; j < rn; j++) { decode_key(h.kf, h.nkf, rhi[j], rlo[j], &o.ranks[j]); o.ranks[j].docid = rpl[j]; }
        // facets
        for (int j = 0; j < h.nfac; j++) {
            int f = h.fac_field[j];
            uint32_t fn = hdr->fac_n[j];
            if (fn == 0xFFFFFFFFu) { o.status = MGPU_E_LIMIT; worst = o.status; g_err = "facet selection did not converge"; fn = 0; }
            o.rcount[f] = (int)fn;
            o.fc[f] = (mgpu_facet_t *)calloc(std::max<uint32_t>(fn, 1), sizeof(mgpu_facet_t));
            const u64 *fhi = (const u64 *)(c->pin_out + h.fac_hi[j]), *flo = (const u64 *)(c->pin_out + h.fac_lo[j]);
            for (uint32_t x = 0; x < fn; x++) { o.fc[f][x].facet_id = (uint32_t)flo[x]; o.fc[f][x].count = 0xFFFFFFFFu - (uint32_t)fhi[x]; }
        }
        // aggs
        for (int a = 0; a < p.num_aggs; a++) {
            o.aggs[a].sum = hdr->agg[a].sum; o.aggs[a].min = hdr->agg[a].mn; o.aggs[a].max = hdr->agg[a].mx;
            o.aggs[a].count = (int64_t)hdr->agg[a].cnt;
        }
        for (int j = 0; j < h.ncard; j++) {
            int a = h.card_agg[j];
            uint32_t cn = hdr->card_n[j];
            o.aggs[a].card_n = cn;
            o.aggs[a].card_ids = (uint32_t *)malloc(std::max<uint32_t>(cn, 1) * 4);
            if (cn > h.card_cap[j]) { cn = h.card_cap[j]; o.aggs[a].card_n = cn; }
            if (cn && h.card_in_res[j]) memcpy(o.aggs[a].card_ids, c->pin_out + h.card_ids_off[j], (size_t)cn * 4);
            else if (cn) HIPCHK(hipMemcpy(o.aggs[a].card_ids, A + nz_base + h.card_ids_off[j], (size_t)cn * 4, hipMemcpyDeviceToHost));
            std::sort(o.aggs[a].card_ids, o.aggs[a].card_ids + cn);
        }
        // debug captures
        if (p.capture & 1) {
            o.dbg_words = q.sh.W;
            o.dbg_docid_map = (uint64_t *)malloc(std::max<size_t>(q.sh.W, 1) * 8);
            HIPCHK(hipMemcpy(o.dbg_docid_map, A + h.docid_map_off, (size_t)q.sh.W * 8, hipMemcpyDeviceToHost));
        }
        if ((p.capture & 2) && h.k && !fast) {
            std::vector<u64> khi(M), klo(M); std::vector<uint32_t> hits(M);
            if (M) {
                HIPCHK(hipMemcpy(khi.data(), A + h.key_hi_off, M * 8, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(klo.data(), A + h.key_lo_off, M * 8, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(hits.data(), A + h.hits_off, M * 4, hipMemcpyDeviceToHost));
            }
            o.dbg_all_ranks = (mgpu_rank_t *)calloc(std::max<uint64_t>(M, 1), sizeof(mgpu_rank_t));
            for (uint64_t j = 0; j < M; j++) { decode_key(h.kf, h.nkf, khi[j], klo[j], &o.dbg_all_ranks[j]); o.dbg_all_ranks[j].docid = hits[j]; }
        }
        // algorithmic bytes (DESIGN.md byte model)
        const uint64_t B = (uint64_t)q.sh.W * 8;
        c->stage_bytes[1] += h.scatter_docs * 4;
        c->stage_bytes[2] += h.formula_rows * B + h.sparse_docs * 4 + 12 * M;   // rows + lists in, raw hits out, k_order in + out
        if (h.k) c->stage_bytes[3] += M * (uint64_t)(4 + 16 + 8.0 * sh->avg_dw + (p.rank_by >= 0 ? 8 : 0) + 16);
        // facet work that rides on the rank kernel's record visit: the ordinal it leaves per hit and field (4 B
        // written), and for the directly counted fields the histogram update (8 B read-modify-write)
        if (q.fac_in_rank)
            for (int f = 0; f < q.sh.num_facets; f++) if ((q.hist_mask >> f) & 1u)
                c->stage_bytes[3] += M * (uint64_t)(4 + (((q.fdirect_mask >> f) & 1u) ? 8 : 0));
        {
            double per = 0;
            for (int f = 0; f < q.sh.num_facets; f++) if ((q.hist_mask >> f) & 1u) {
                if ((q.fdirect_mask >> f) & 1u) continue;                 // counted in the rank stage
                per += q.fac_in_rank ? 4 + 4 : 4 + 8 + 4.0 * sh->avg_fv[f];       // ordinal + bin, or docid + record header + values
            }
            int na = 0; for (int a = 0; a < p.num_aggs; a++) if (p.agg_type[a] != MGPU_A_CARDINALITY) na++;
            per += na * (4 + 8);
            c->stage_bytes[4] += (uint64_t)(M * per);
        }
        for (int j = 0; j < h.nfac; j++) {
            // touched bins (at most one per facet value of a hit): bin index + count + id in, 16-byte key out
            const int f = h.fac_field[j];
            const double touched = std::min<double>((double)q.sh.fa_nord[f], (double)M * sh->avg_fv[f]);
            c->stage_bytes[fuse_fk ? 6 : 5] += (uint64_t)(touched * 28.0);   // the select kernels derive the keys when fuse_fk
        }
        if (h.k) c->stage_bytes[6] += 16 * M * (M > MG_SEL_SMALL ? 2 : 1);
    }
    c->stage_bytes[0] = up_bytes; c->stage_bytes[7] = res_bytes;
    // the batch ran to its end: the self-cleaned pool is zero again (MGPU_CHECK_CLEAN makes sure of it) -- unless a
    // job gave up (error word, selection that did not converge): its bins were never exchanged back to zero
    c->cln_dirty = worst != MGPU_OK;
    if (getenv("MGPU_CHECK_CLEAN") && c->cln.cap && worst == MGPU_OK) {
        uint32_t *flag = (uint32_t *)A;              // the arena is free again
        HIPCHK(hipMemsetAsync(flag, 0, 4, st));
        mgk_check_zero(st, CL, c->cln.cap, flag);
        uint32_t hflag = 0;
        HIPCHK(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (hflag) { c->cln_dirty = true; return fail(MGPU_E_HIP, "self-cleaned pool was left dirty by this batch"); }
    }
    c->host_ms[1] = (float)(t_waited - t_wait0);         // time spent waiting in execute_end (the launch half returned earlier)
    c->host_ms[2] = (float)(wall_ms() - t_waited);
    *completed = true;
    return worst;
}

// ------------------------------------------------------------------ primitives
extern "C" int mgpu_posting_op(mgpu_ctx_t *c, mgpu_shard_t *s, int op, uint32_t pid_a, uint32_t pid_b,
                               uint64_t *out_words, uint32_t nwords, uint64_t *card) {
    if (!c || !s || op < 0 || op > 2) return fail(MGPU_E_ARG, "bad argument");
    if (pid_a >= s->dev.num_postings || pid_b >= s->dev.num_postings) return fail(MGPU_E_ARG, "posting id out of range");
    HIPCHK(hipSetDevice(c->device));
    const ShardDev &sd = s->dev;
    const size_t rowb = (size_t)sd.Wp * 8;
    int rc = ensure_arena(c, rowb * 3 + 4096 + 2 * sizeof(ScatterItem) * ((size_t)sd.N / MG_SCATTER_CHUNK + 2));
    if (rc) return rc;
    char *A = c->arena.dev;
    u64 *ra = (u64 *)A, *rb = (u64 *)(A + rowb), *ro = (u64 *)(A + 2 * rowb);
    unsigned long long *dcard = (unsigned long long *)(A + 3 * rowb);
    ScatterItem *ditems = (ScatterItem *)(A + 3 * rowb + 256);
    hipStream_t st = c->stream;
    HIPCHK(hipMemsetAsync(A, 0, 3 * rowb + 256, st));
    std::vector<ScatterItem> items;
    const u64 *srca = nullptr, *srcb = nullptr;
    auto prep = [&](uint32_t pid, u64 *row, const u64 **src) {
        int32_t di = s->post_dense[pid];
        if (di >= 0) { *src = (const u64 *)sd.dense + (size_t)di * sd.Wp; return; }
        *src = row;
        uint64_t cnt = s->post_off[pid + 1] - s->post_off[pid];
        for (uint64_t at = 0; at < cnt; at += MG_SCATTER_CHUNK) {
            ScatterItem it; it.src = sd.post_doc + s->post_off[pid] + at; it.dst = (uint64_t *)row;
            it.n = (uint32_t)std::min<uint64_t>(MG_SCATTER_CHUNK, cnt - at); it.pad = 0;
            items.push_back(it);
        }
    };
    prep(pid_a, ra, &srca); prep(pid_b, rb, &srcb);
    if (!items.empty()) {
        HIPCHK(hipMemcpyAsync(ditems, items.data(), items.size() * sizeof(ScatterItem), hipMemcpyHostToDevice, st));
        mgk_scatter(st, ditems, (uint32_t)items.size());
    }
    mgk_row_op(st, srca, srcb, ro, sd.W, op, dcard);
    unsigned long long hc = 0;
    HIPCHK(hipMemcpyAsync(&hc, dcard, 8, hipMemcpyDeviceToHost, st));
    if (out_words) {
        if (nwords < sd.W) return fail(MGPU_E_ARG, "output buffer too small");
        HIPCHK(hipMemcpyAsync(out_words, ro, (size_t)sd.W * 8, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    if (card) *card = hc;
    return MGPU_OK;
}

// The same operation for callers that cannot allocate (serving loops and harnesses that forbid hipMalloc, code that
// runs while a capture is open elsewhere in the process; the call waits for its own stream, so it is not capturable
// itself).  No hipMalloc / hipFree / hipHostMalloc and no host heap use in here: the operand rows,
// the result row and the cardinality word live in this device's copy of g_noalloc_scratch (address looked up at
// mgpu_ctx_create), sparse postings are scattered by a kernel that takes the list as its arguments (no item table
// to upload), and results come back through the context's pinned bounce buffer.  Every call on a GPU shares that
// GPU's scratch array, which still holds the previous call's rows: the regions a call accumulates into (scattered
// operand rows, the cardinality word) are cleared first, the result row is written whole.  Calls that overlap on one
// GPU (two contexts, two threads) race on it.
extern "C" int mgpu_posting_op_noalloc(mgpu_ctx_t *c, mgpu_shard_t *s, int op, uint32_t pid_a, uint32_t pid_b,
                                       uint64_t *out_words, uint32_t nwords, uint64_t *card) {
    if (!c || !s || op < 0 || op > 2) return fail(MGPU_E_ARG, "bad argument");
    if (pid_a >= s->dev.num_postings || pid_b >= s->dev.num_postings) return fail(MGPU_E_ARG, "posting id out of range");
    if (s->device != c->device) return fail(MGPU_E_ARG, "shard not on the context's device");
    const ShardDev &sd = s->dev;
    const size_t rowb = (size_t)sd.Wp * 8;
    if (rowb > NA_ROW_BYTES)
        return fail(MGPU_E_LIMIT, "mgpu_posting_op_noalloc: shard has more than MGPU_NOALLOC_MAX_DOCS docids (static scratch holds three rows of that size)");
    if (out_words && nwords < sd.W) return fail(MGPU_E_ARG, "output buffer too small");
    if (!c->na_scratch || !c->na_pin) return fail(MGPU_E_HIP, "context has no static scratch (mgpu_ctx_create did not finish)");
    if (c->pend && c->pend->active) return fail(MGPU_E_ARG, "a batch is in flight on this context");
    HIPCHK(hipSetDevice(c->device));
    char *S = c->na_scratch;                                   // the device address, never the array's name
    u64 *ra = (u64 *)S, *rb = (u64 *)(S + NA_ROW_BYTES), *ro = (u64 *)(S + 2 * NA_ROW_BYTES);
    unsigned long long *dcard = (unsigned long long *)(S + 3 * NA_ROW_BYTES);
    hipStream_t st = c->stream;
    HIPCHK(hipMemsetAsync(dcard, 0, 256, st));
    const u64 *src[2] = {nullptr, nullptr};
    const uint32_t pids[2] = {pid_a, pid_b};
    u64 *rows[2] = {ra, rb};
    for (int i = 0; i < 2; i++) {
        const int32_t di = s->post_dense[pids[i]];
        if (di >= 0) { src[i] = (const u64 *)sd.dense + (size_t)di * sd.Wp; continue; }
        src[i] = rows[i];
        HIPCHK(hipMemsetAsync(rows[i], 0, rowb, st));           // last call's row is still there
        const uint64_t cnt = s->post_off[pids[i] + 1] - s->post_off[pids[i]];
        mgk_scatter_list(st, sd.post_doc + s->post_off[pids[i]], cnt, (unsigned long long *)rows[i]);
    }
    mgk_row_op(st, src[0], src[1], ro, sd.W, op, dcard);
    HIPCHK(hipMemcpyAsync(c->na_pin, dcard, 8, hipMemcpyDeviceToHost, st));
    if (out_words) HIPCHK(hipMemcpyAsync(c->na_pin + 256, ro, (size_t)sd.W * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (card) memcpy(card, c->na_pin, 8);
    if (out_words) memcpy(out_words, c->na_pin + 256, (size_t)sd.W * 8);
    return MGPU_OK;
}

// ------------------------------------------------------------------ f-1: term lookup on the device
struct mgpu_dict {
    int device;
    uint32_t *chars = nullptr, *off = nullptr, *wid = nullptr;
    uint32_t *sig = nullptr;                         // mg_lev_sig per word: what k_term_filter reads
    uint32_t nwords = 0; uint64_t nchars = 0;
};

extern "C" int mgpu_dict_create(int device, const uint32_t *chars, const uint32_t *word_off, const uint32_t *word_wid,
                                uint32_t num_words, mgpu_dict_t **out) {
    if (!out || !word_off || !word_wid || (num_words && !chars)) return fail(MGPU_E_ARG, "null argument");
    // the kernel trusts the offset table: check it here, once
    for (uint32_t i = 0; i < num_words; i++)
        if (word_off[i + 1] < word_off[i]) return fail(MGPU_E_ARG, "word offsets must ascend");
    const uint64_t nchars = num_words ? word_off[num_words] : 0;
    if (word_off[0] != 0) return fail(MGPU_E_ARG, "word offsets must start at 0");
    HIPCHK(hipSetDevice(device));
    mgpu_dict *d = new mgpu_dict();
    d->device = device; d->nwords = num_words; d->nchars = nchars;
    auto bail = [&](hipError_t e) { mgpu_dict_destroy(d); return fail(MGPU_E_HIP, hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMalloc((void **)&d->chars, (nchars + 4) * 4)) != hipSuccess) return bail(e);
    if ((e = hipMalloc((void **)&d->off, ((size_t)num_words + 1) * 4)) != hipSuccess) return bail(e);
    if ((e = hipMalloc((void **)&d->wid, ((size_t)num_words + 1) * 4)) != hipSuccess) return bail(e);
    if (nchars && (e = hipMemcpy(d->chars, chars, nchars * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    if ((e = hipMemcpy(d->off, word_off, ((size_t)num_words + 1) * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    if (num_words && (e = hipMemcpy(d->wid, word_wid, (size_t)num_words * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    {
        std::vector<uint32_t> sig((size_t)num_words + 1, 0u);
        for (uint32_t i = 0; i < num_words; i++) sig[i] = mg_lev_sig(chars + word_off[i], (int)std::min<uint32_t>(word_off[i + 1] - word_off[i], 1u << 20));
        if ((e = hipMalloc((void **)&d->sig, ((size_t)num_words + 1) * 4)) != hipSuccess) return bail(e);
        if ((e = hipMemcpy(d->sig, sig.data(), ((size_t)num_words + 1) * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    }
    *out = d;
    return MGPU_OK;
}

extern "C" void mgpu_dict_destroy(mgpu_dict_t *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->chars) (void)hipFree(d->chars);
    if (d->off) (void)hipFree(d->off);
    if (d->wid) (void)hipFree(d->wid);
    if (d->sig) (void)hipFree(d->sig);
    delete d;
}

extern "C" uint64_t mgpu_dict_bytes(const mgpu_dict_t *d) { return d ? (d->nchars + 4) * 4 + ((uint64_t)d->nwords + 1) * 12 : 0; }

extern "C" void mgpu_term_hits_free(mgpu_term_hit_t *hits) { free(hits); }

extern "C" int mgpu_lookup_terms(mgpu_ctx_t *c, const mgpu_term_lookup_t *lookups, int n, mgpu_term_hit_t **hits, uint32_t *nhits) {
    if (!c || !hits || !nhits || n < 0 || (n && !lookups)) return fail(MGPU_E_ARG, "bad argument");
    *hits = nullptr; *nhits = 0;
    if (n == 0) return MGPU_OK;
    if (c->pend && c->pend->active) return fail(MGPU_E_ARG, "a batch is in flight on this context");
    if (n > 65535) return fail(MGPU_E_LIMIT, "more than 65535 lookups per call (one grid row each)");
    HIPCHK(hipSetDevice(c->device));
    // job table (sorted by dictionary: the filter kernel walks a dictionary once for all of its terms) + term blob
    std::vector<uint32_t> order((size_t)n);
    for (int i = 0; i < n; i++) {
        const mgpu_term_lookup_t &l = lookups[i];
        if (!l.dict || !l.chars || l.len < 1 || l.len > MGPU_MAX_TERM_LEN) return fail(MGPU_E_ARG, "term of 0 or more than MGPU_MAX_TERM_LEN characters, or no dictionary");
        if (l.dict->device != c->device) return fail(MGPU_E_ARG, "dictionary lives on another device");
        order[(size_t)i] = (uint32_t)i;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (uintptr_t)lookups[a].dict < (uintptr_t)lookups[b].dict; });
    std::vector<TermJob> jobs((size_t)n);
    std::vector<TermDict> dicts;
    std::vector<uint32_t> blob;
    uint32_t max_words = 0;
    for (int x = 0; x < n; x++) {
        const uint32_t i = order[(size_t)x];
        const mgpu_term_lookup_t &l = lookups[i];
        TermJob &j = jobs[(size_t)x];
        j.chars = l.dict->chars; j.off = l.dict->off; j.wid = l.dict->wid; j.nwords = l.dict->nwords;
        j.term_off = (uint32_t)blob.size(); j.tlen = l.len; j.prefix = l.prefix ? 1 : 0; j.maxdist = mg_lev_maxdist(l.len);
        j.tsig = mg_lev_sig(l.chars, l.len); j.lookup = i; j.pad = 0;
        blob.insert(blob.end(), l.chars, l.chars + l.len);
        if (dicts.empty() || dicts.back().sig != l.dict->sig) {
            TermDict td; td.sig = l.dict->sig; td.nwords = l.dict->nwords; td.job_lo = (uint32_t)x; td.job_hi = (uint32_t)x; td.pad = 0;
            dicts.push_back(td);
        }
        dicts.back().job_hi = (uint32_t)x + 1;
        max_words = std::max(max_words, j.nwords);
    }
    if (dicts.size() > 65535) return fail(MGPU_E_LIMIT, "more than 65535 dictionaries per call (one grid row each)");
    const size_t jobs_b = (jobs.size() * sizeof(TermJob) + 255) & ~(size_t)255, dicts_b = (dicts.size() * sizeof(TermDict) + 255) & ~(size_t)255,
                 blob_b = (blob.size() * 4 + 255) & ~(size_t)255;
    if (c->lk_in_cap < jobs_b + dicts_b + blob_b) {
        if (c->lk_in) { (void)hipFree(c->lk_in); c->lk_in = nullptr; c->lk_in_cap = 0; }
        const size_t cap = std::max<size_t>(jobs_b + dicts_b + blob_b, 1u << 20);
        HIPCHK(hipMalloc((void **)&c->lk_in, cap));
        c->lk_in_cap = cap;
    }
    if (!c->lk_count) HIPCHK(hipMalloc((void **)&c->lk_count, 256));
    if (!c->lk_hits) { HIPCHK(hipMalloc((void **)&c->lk_hits, (size_t)(1u << 20) * 12)); c->lk_hits_cap = 1u << 20; }
    if (!c->lk_cand) { HIPCHK(hipMalloc((void **)&c->lk_cand, (size_t)(1u << 22) * 8)); c->lk_cand_cap = 1u << 22; }
    hipStream_t st = c->stream;
    HIPCHK(hipMemcpyAsync(c->lk_in, jobs.data(), jobs.size() * sizeof(TermJob), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->lk_in + jobs_b, dicts.data(), dicts.size() * sizeof(TermDict), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->lk_in + jobs_b + dicts_b, blob.data(), blob.size() * 4, hipMemcpyHostToDevice, st));
    uint32_t total = 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        uint32_t counts[2] = {0, 0};
        HIPCHK(hipMemsetAsync(c->lk_count, 0, 8, st));
        mgk_term_scan(st, (const TermJob *)c->lk_in, (const TermDict *)(c->lk_in + jobs_b), (uint32_t)dicts.size(), max_words,
                      (const uint32_t *)(c->lk_in + jobs_b + dicts_b), c->lk_cand, c->lk_cand_cap, c->lk_hits, c->lk_count, c->lk_hits_cap);
        HIPCHK(hipMemcpyAsync(counts, c->lk_count, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        total = counts[0];
        if (counts[1] <= c->lk_cand_cap && total <= c->lk_hits_cap) break;
        if (attempt == 2) return fail(MGPU_E_HIP, "term scan counts changed between runs");
        // a list was too short: its counter says how long it has to be
        if (counts[1] > c->lk_cand_cap) {
            (void)hipFree(c->lk_cand); c->lk_cand = nullptr; c->lk_cand_cap = 0;
            HIPCHK(hipMalloc((void **)&c->lk_cand, (size_t)counts[1] * 8));
            c->lk_cand_cap = counts[1];
        }
        if (total > c->lk_hits_cap) {
            (void)hipFree(c->lk_hits); c->lk_hits = nullptr; c->lk_hits_cap = 0;
            HIPCHK(hipMalloc((void **)&c->lk_hits, (size_t)total * 12));
            c->lk_hits_cap = total;
        }
    }
    if (total) {
        mgpu_term_hit_t *h = (mgpu_term_hit_t *)malloc((size_t)total * sizeof(mgpu_term_hit_t));
        if (!h) return fail(MGPU_E_LIMIT, "out of host memory");
        hipError_t e = hipMemcpy(h, c->lk_hits, (size_t)total * 12, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(h); return fail(MGPU_E_HIP, hipGetErrorString(e)); }
        *hits = h;
    }
    *nhits = total;
    return MGPU_OK;
}

// ------------------------------------------------------------------ f-4: highlight matcher
extern "C" int mgpu_highlight_match(mgpu_ctx_t *c, const uint32_t *tok_chars, const uint32_t *tok_off, const uint32_t *tok_job, uint32_t ntok,
                                    const mgpu_hl_term_t *terms, uint32_t nterms, const uint32_t *term_chars, uint32_t nterm_chars,
                                    const uint32_t *job_term_off, uint32_t njobs, int32_t *match_len) {
    if (!c || !tok_off || (ntok && (!tok_job || !match_len)) || !job_term_off || (nterms && !terms)) return fail(MGPU_E_ARG, "null argument");
    if (!ntok) return MGPU_OK;
    if (c->pend && c->pend->active) return fail(MGPU_E_ARG, "a batch is in flight on this context");
    // the kernel trusts the tables: check them here
    if (tok_off[0] != 0) return fail(MGPU_E_ARG, "token offsets must start at 0");
    for (uint32_t i = 0; i < ntok; i++) {
        if (tok_off[i + 1] < tok_off[i]) return fail(MGPU_E_ARG, "token offsets must ascend");
        if (tok_job[i] >= njobs) return fail(MGPU_E_ARG, "token names a job that does not exist");
    }
    for (uint32_t j = 0; j < njobs; j++) if (job_term_off[j + 1] < job_term_off[j] || job_term_off[j + 1] > nterms) return fail(MGPU_E_ARG, "job term ranges malformed");
    for (uint32_t j = 0; j < nterms; j++) if ((uint64_t)terms[j].off + terms[j].len > nterm_chars) return fail(MGPU_E_ARG, "term beyond the term characters");
    const size_t nchars = tok_off[ntok];
    if (nchars && !tok_chars) return fail(MGPU_E_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_chars = al(std::max<size_t>(nchars, 1) * 4), b_off = al(((size_t)ntok + 1) * 4), b_job = al((size_t)ntok * 4),
                 b_terms = al(std::max<size_t>(nterms, 1) * sizeof(mgpu_hl_term_t)), b_tch = al(std::max<size_t>(nterm_chars, 1) * 4),
                 b_jto = al(((size_t)njobs + 1) * 4), b_out = al((size_t)ntok * 4);
    const size_t need = b_chars + b_off + b_job + b_terms + b_tch + b_jto + b_out;
    if (c->hl_cap < need) {
        if (c->hl_buf) { (void)hipFree(c->hl_buf); c->hl_buf = nullptr; c->hl_cap = 0; }
        const size_t cap = std::max<size_t>(need, (size_t)4 << 20);
        HIPCHK(hipMalloc((void **)&c->hl_buf, cap));
        c->hl_cap = cap;
    }
    char *p = c->hl_buf;
    char *d_chars = p; p += b_chars; char *d_off = p; p += b_off; char *d_job = p; p += b_job; char *d_terms = p; p += b_terms;
    char *d_tch = p; p += b_tch; char *d_jto = p; p += b_jto; char *d_out = p;
    hipStream_t st = c->stream;
    if (nchars) HIPCHK(hipMemcpyAsync(d_chars, tok_chars, nchars * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_off, tok_off, ((size_t)ntok + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_job, tok_job, (size_t)ntok * 4, hipMemcpyHostToDevice, st));
    if (nterms) HIPCHK(hipMemcpyAsync(d_terms, terms, (size_t)nterms * sizeof(mgpu_hl_term_t), hipMemcpyHostToDevice, st));
    if (nterm_chars) HIPCHK(hipMemcpyAsync(d_tch, term_chars, (size_t)nterm_chars * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_jto, job_term_off, ((size_t)njobs + 1) * 4, hipMemcpyHostToDevice, st));
    mgk_hl_match(st, (const uint32_t *)d_chars, (const uint32_t *)d_off, (const uint32_t *)d_job, ntok, d_terms, (const uint32_t *)d_tch, (const uint32_t *)d_jto, (int32_t *)d_out);
    HIPCHK(hipMemcpyAsync(match_len, d_out, (size_t)ntok * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    return MGPU_OK;
}

// ------------------------------------------------------------------ (e): RCCL all-gather of result slots
#include <rccl/rccl.h>
struct mgpu_comm {
    int device, rank, world, nslots;
    size_t max_bytes;
    ncclComm_t comm;
    hipStream_t stream;
    std::vector<char *> d_send, d_recv;
    std::vector<hipEvent_t> done;
};
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(MGPU_E_HIP, std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)

extern "C" int mgpu_comm_unique_id(void *id) {
    if (!id) return fail(MGPU_E_ARG, "null argument");
    static_assert(sizeof(ncclUniqueId) == MGPU_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return MGPU_OK;
}

extern "C" void mgpu_comm_destroy(mgpu_comm_t *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (char *p : c->d_send) if (p) (void)hipFree(p);
    for (char *p : c->d_recv) if (p) (void)hipFree(p);
    for (hipEvent_t e : c->done) (void)hipEventDestroy(e);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mgpu_comm_create(int device, const void *id, int rank, int world, int nslots, size_t max_bytes, mgpu_comm_t **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world || nslots < 1 || nslots > 256 || !max_bytes) return fail(MGPU_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(device));
    mgpu_comm *c = new mgpu_comm();
    c->device = device; c->rank = rank; c->world = world; c->nslots = nslots; c->max_bytes = max_bytes;
    c->comm = nullptr; c->stream = nullptr;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t nr = ncclCommInitRank(&c->comm, world, u, rank);
    if (nr != ncclSuccess) { c->comm = nullptr; mgpu_comm_destroy(c); return fail(MGPU_E_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(nr)); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->stream = nullptr; mgpu_comm_destroy(c); return fail(MGPU_E_HIP, "hipStreamCreate"); }
    for (int i = 0; i < nslots; i++) {
        char *a = nullptr, *b = nullptr; hipEvent_t e;
        if (hipMalloc((void **)&a, max_bytes) != hipSuccess || hipMalloc((void **)&b, max_bytes * (size_t)world) != hipSuccess ||
            hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            mgpu_comm_destroy(c);
            return fail(MGPU_E_HIP, "comm buffers");
        }
        c->d_send.push_back(a); c->d_recv.push_back(b); c->done.push_back(e);
    }
    *out = c;
    return MGPU_OK;
}

extern "C" int mgpu_comm_rank(const mgpu_comm_t *c) { return c ? c->rank : -1; }
extern "C" int mgpu_comm_world(const mgpu_comm_t *c) { return c ? c->world : 0; }

extern "C" int mgpu_comm_allgather_begin(mgpu_comm_t *c, int slot, const void *send, size_t bytes, void *recv) {
    if (!c || slot < 0 || slot >= c->nslots || !send || !recv || !bytes) return fail(MGPU_E_ARG, "bad argument");
    if (bytes > c->max_bytes) return fail(MGPU_E_LIMIT, "gather larger than the comm's buffers");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_send[slot], send, bytes, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(ncclAllGather(c->d_send[slot], c->d_recv[slot], bytes, ncclUint8, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(recv, c->d_recv[slot], bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(c->done[slot], c->stream));
    return MGPU_OK;
}

extern "C" int mgpu_comm_allgather_end(mgpu_comm_t *c, int slot) {
    if (!c || slot < 0 || slot >= c->nslots) return fail(MGPU_E_ARG, "bad argument");
    HIPCHK(hipEventSynchronize(c->done[slot]));
    return MGPU_OK;
}

extern "C" void *mgpu_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void mgpu_host_free(void *p) { if (p) (void)hipHostFree(p); }
