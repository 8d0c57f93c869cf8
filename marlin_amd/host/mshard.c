/* mshard.c -- in-memory shard store + builder (see mshard.h). */
#include "mshard.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------ farmhash32 (mk) */
/* Google FarmHash, farmhashmk::Hash32 (MIT licence; the reference vendors the C port at
 * src/utils/farmhash-c.c:858-951).  Restated from the published algorithm. */
static inline uint32_t fh_fetch32(const char *p) { uint32_t r; memcpy(&r, p, 4); return r; }
static inline uint32_t fh_ror32(uint32_t v, int s) { return s == 0 ? v : (v >> s) | (v << (32 - s)); }
#define FH_C1 0xcc9e2d51u
#define FH_C2 0x1b873593u
static inline uint32_t fh_fmix(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
static inline uint32_t fh_mur(uint32_t a, uint32_t h) {
    a *= FH_C1; a = fh_ror32(a, 17); a *= FH_C2; h ^= a; h = fh_ror32(h, 19);
    return h * 5 + 0xe6546b64u;
}

uint32_t m_farmhash32_mk(const char *s, size_t len) {
    if (len <= 4) {
        uint32_t b = 0, c = 9;
        for (size_t i = 0; i < len; i++) { signed char v = s[i]; b = b * FH_C1 + (uint32_t)v; c ^= b; }
        return fh_fmix(fh_mur(b, fh_mur((uint32_t)len, c)));
    }
    if (len <= 12) {
        uint32_t a = (uint32_t)len, b = (uint32_t)len * 5, c = 9, d = b;
        a += fh_fetch32(s); b += fh_fetch32(s + len - 4); c += fh_fetch32(s + ((len >> 1) & 4));
        return fh_fmix(fh_mur(c, fh_mur(b, fh_mur(a, d))));
    }
    if (len <= 24) {
        uint32_t a = fh_fetch32(s - 4 + (len >> 1)), b = fh_fetch32(s + 4), c = fh_fetch32(s + len - 8);
        uint32_t d = fh_fetch32(s + (len >> 1)), e = fh_fetch32(s), f = fh_fetch32(s + len - 4);
        uint32_t h = d * FH_C1 + (uint32_t)len;
        a = fh_ror32(a, 12) + f; h = fh_mur(c, h) + a;
        a = fh_ror32(a, 3) + c;  h = fh_mur(e, h) + a;
        a = fh_ror32(a + f, 12) + d; h = fh_mur(b, h) + a;
        return fh_fmix(h);
    }
    uint32_t h = (uint32_t)len, g = FH_C1 * (uint32_t)len, f = g;
    uint32_t a0 = fh_ror32(fh_fetch32(s + len - 4) * FH_C1, 17) * FH_C2;
    uint32_t a1 = fh_ror32(fh_fetch32(s + len - 8) * FH_C1, 17) * FH_C2;
    uint32_t a2 = fh_ror32(fh_fetch32(s + len - 16) * FH_C1, 17) * FH_C2;
    uint32_t a3 = fh_ror32(fh_fetch32(s + len - 12) * FH_C1, 17) * FH_C2;
    uint32_t a4 = fh_ror32(fh_fetch32(s + len - 20) * FH_C1, 17) * FH_C2;
    h ^= a0; h = fh_ror32(h, 19); h = h * 5 + 0xe6546b64u;
    h ^= a2; h = fh_ror32(h, 19); h = h * 5 + 0xe6546b64u;
    g ^= a1; g = fh_ror32(g, 19); g = g * 5 + 0xe6546b64u;
    g ^= a3; g = fh_ror32(g, 19); g = g * 5 + 0xe6546b64u;
    f += a4; f = fh_ror32(f, 19) + 113;
    size_t iters = (len - 1) / 20;
    do {
        uint32_t a = fh_fetch32(s), b = fh_fetch32(s + 4), c = fh_fetch32(s + 8);
        uint32_t d = fh_fetch32(s + 12), e = fh_fetch32(s + 16);
        h += a; g += b; f += c;
        h = fh_mur(d, h) + e; g = fh_mur(c, g) + a; f = fh_mur(b + e * FH_C1, f) + d;
        f += g; g += f; s += 20;
    } while (--iters != 0);
    g = fh_ror32(g, 11) * FH_C1; g = fh_ror32(g, 17) * FH_C1;
    f = fh_ror32(f, 11) * FH_C1; f = fh_ror32(f, 17) * FH_C1;
    h = fh_ror32(h + g, 19); h = h * 5 + 0xe6546b64u; h = fh_ror32(h, 17) * FH_C1;
    h = fh_ror32(h + f, 19); h = h * 5 + 0xe6546b64u; h = fh_ror32(h, 17) * FH_C1;
    return h;
}

uint32_t m_facet_id(const char *s, size_t len) {
    uint32_t x = m_farmhash32_mk(s, len) * FH_C1;          /* debug_tweak32, farmhash-c.c:287-293 */
    x = __builtin_bswap32(x);
    return ~x;
}

/* ------------------------------------------------------------------ vint */
static inline uint8_t *write_vint(uint8_t *buf, int vi) {      /* sindex.c:328-338 */
    uint32_t i = (uint32_t)vi;
    while ((i & ~0x7Fu) != 0) { *buf++ = (uint8_t)((i & 0x7f) | 0x80); i >>= 7; }
    *buf++ = (uint8_t)i;
    return buf;
}

const uint8_t *m_read_vint(const uint8_t *buf, int *value) {   /* sindex.c:340-351 */
    uint8_t b = *buf++;
    int i = b & 0x7F;
    for (int shift = 7; (b & 0x80) != 0; shift += 7) { b = *buf++; i |= (b & 0x7F) << shift; }
    *value = i;
    return buf;
}

/* ------------------------------------------------------------------ builder */
typedef struct { uint32_t wid; uint32_t position; int priority; } widpos_t;
typedef struct { uint32_t *a; uint32_t n, cap; } u32vec_t;
typedef struct { double *val; uint32_t *doc; uint64_t n, cap; } numpairs_t;
typedef struct { uint32_t *id; uint32_t *doc; uint64_t n, cap; } fpairs_t;

struct mshard_builder {
    int num_strings, num_numbers, num_bools, num_facets;
    mdict_t *dict;
    uint32_t docid;
    int in_doc;
    /* per wid: its first three twids and the number of valid ones */
    uint32_t *wid_tw;          /* [cap_w * 3] */
    uint8_t  *wid_lim;
    uint32_t  cap_w;
    /* per document */
    widpos_t *wp; uint32_t nwp, cap_wp;
    double    num_data[M_MAX_FIELDS];
    u32vec_t  facet_data[M_MAX_FIELDS];
    /* accumulated */
    uint8_t  *dd; uint64_t dd_n, dd_cap;
    uint64_t *dd_off; uint64_t off_cap;
    numpairs_t np[M_MAX_FIELDS];
    fpairs_t   fp[M_MAX_FIELDS];
    u32vec_t   bp[M_MAX_FIELDS][2];
    /* facetid2str */
    m_u32map_t fstr_idx;       /* id -> index into fstr */
    uint32_t  *fstr_id; char **fstr; uint32_t nfstr, cap_fstr;
    uint64_t   total_wp;       /* sum of unique wids over docs */
    uint32_t   raw_max_wid;    /* largest word id given through msb_add_raw_wordpos */
};

static void u32vec_push(u32vec_t *v, uint32_t x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 8; v->a = realloc(v->a, v->cap * sizeof(uint32_t)); }
    v->a[v->n++] = x;
}

mshard_builder_t *msb_new(int num_strings, int num_numbers, int num_bools, int num_facets) {
    if (num_numbers > M_MAX_FIELDS || num_facets > M_MAX_FIELDS || num_bools > M_MAX_FIELDS) return NULL;
    mshard_builder_t *b = calloc(1, sizeof(*b));
    b->num_strings = num_strings; b->num_numbers = num_numbers;
    b->num_bools = num_bools; b->num_facets = num_facets;
    b->dict = mdict_new();
    b->off_cap = 1024;
    b->dd_off = calloc(b->off_cap, sizeof(uint64_t));
    m_u32map_init(&b->fstr_idx);
    return b;
}

uint32_t msb_begin_doc(mshard_builder_t *b) {
    b->docid++;
    b->in_doc = 1;
    b->nwp = 0;
    memset(b->num_data, 0, sizeof(b->num_data));           /* sindex_docdata_init, sindex.c:257-267 */
    for (int i = 0; i < b->num_facets; i++) b->facet_data[i].n = 0;
    return b->docid;
}

static void ensure_wid_cap(mshard_builder_t *b, uint32_t wid) {
    if (wid < b->cap_w) return;
    uint32_t nc = b->cap_w ? b->cap_w : 1024;
    while (nc <= wid) nc *= 2;
    b->wid_tw = realloc(b->wid_tw, (size_t)nc * 3 * sizeof(uint32_t));
    b->wid_lim = realloc(b->wid_lim, nc);
    memset(b->wid_lim + b->cap_w, 0, nc - b->cap_w);
    b->cap_w = nc;
}

static inline void push_widpos(mshard_builder_t *b, uint32_t wid, int priority, uint32_t position) {
    if (b->nwp == b->cap_wp) {
        b->cap_wp = b->cap_wp ? b->cap_wp * 2 : 64;
        b->wp = realloc(b->wp, b->cap_wp * sizeof(widpos_t));
    }
    b->wp[b->nwp].wid = wid; b->wp[b->nwp].priority = priority; b->wp[b->nwp].position = position;
    b->nwp++;
}

/* string_new_word_pos (sindex.c:658-726) */
void msb_add_word(mshard_builder_t *b, int i_priority, const m_chr_t *chars, int length, int position) {
    uint32_t tw[M_LEVLIMIT] = {0, 0, 0};
    uint32_t wid = mdict_exists(b->dict, chars, length, tw);
    if (wid == 0) {
        wid = mdict_insert(b->dict, chars, length, tw);
        ensure_wid_cap(b, wid);
        int limit = length >= M_LEVLIMIT ? M_LEVLIMIT : length;
        b->wid_lim[wid] = (uint8_t)limit;
        for (int i = 0; i < limit; i++) b->wid_tw[(size_t)wid * 3 + i] = tw[i];
    }
    push_widpos(b, wid, i_priority, (uint32_t)position);
}

/* One word occurrence by word id, past the dictionary (format tests: tests/test_docdata_format.py feeds the reference's
 * own fixtures through here); the word gets a posting but no dictionary entry and no top-level ids */
void msb_add_raw_wordpos(mshard_builder_t *b, uint32_t wid, int priority, uint32_t position) {
    ensure_wid_cap(b, wid);
    if (wid > b->raw_max_wid) b->raw_max_wid = wid;
    push_widpos(b, wid, priority, position);
}
/* a facet id of the current document as it is, without a string behind it */
static void add_facet_id(mshard_builder_t *b, int f_priority, uint32_t facet_id, const char *str);
void msb_add_raw_facet(mshard_builder_t *b, int f_priority, uint32_t facet_id) {
    add_facet_id(b, f_priority, facet_id, "");
}

struct an_data { mshard_builder_t *b; int priority; };
static void an_cb(const m_chr_t *chars, int length, int position, void *data) {
    struct an_data *ad = data;
    msb_add_word(ad->b, ad->priority, chars, length, position);
}

void msb_add_string(mshard_builder_t *b, int i_priority, const char *str) {
    struct an_data ad = { b, i_priority };
    m_analyze_for_indexing(str, an_cb, &ad);
}

void msb_add_number(mshard_builder_t *b, int i_priority, double d, int is_indexed) {
    b->num_data[i_priority] = d;                            /* sindex.c:831,851 */
    if (!is_indexed) return;
    numpairs_t *np = &b->np[i_priority];                    /* index_number, sindex.c:528-535 */
    if (np->n == np->cap) {
        np->cap = np->cap ? np->cap * 2 : 1024;
        np->val = realloc(np->val, np->cap * sizeof(double));
        np->doc = realloc(np->doc, np->cap * sizeof(uint32_t));
    }
    np->val[np->n] = d; np->doc[np->n] = b->docid; np->n++;
}

static void add_facet_id(mshard_builder_t *b, int f_priority, uint32_t facet_id, const char *str) {
    if (m_u32map_get(&b->fstr_idx, facet_id, -1) < 0) {     /* facetid2str, sindex.c:603-625 */
        if (b->nfstr == b->cap_fstr) {
            b->cap_fstr = b->cap_fstr ? b->cap_fstr * 2 : 256;
            b->fstr_id = realloc(b->fstr_id, b->cap_fstr * sizeof(uint32_t));
            b->fstr = realloc(b->fstr, b->cap_fstr * sizeof(char *));
        }
        b->fstr_id[b->nfstr] = facet_id; b->fstr[b->nfstr] = strdup(str);
        m_u32map_set(&b->fstr_idx, facet_id, (int32_t)b->nfstr);
        b->nfstr++;
    }
    u32vec_push(&b->facet_data[f_priority], facet_id);      /* sindex.c:628 */
    fpairs_t *fp = &b->fp[f_priority];                      /* facetid2bmap, sindex.c:630-631 */
    if (fp->n == fp->cap) {
        fp->cap = fp->cap ? fp->cap * 2 : 1024;
        fp->id = realloc(fp->id, fp->cap * sizeof(uint32_t));
        fp->doc = realloc(fp->doc, fp->cap * sizeof(uint32_t));
    }
    fp->id[fp->n] = facet_id; fp->doc[fp->n] = b->docid; fp->n++;
}

void msb_add_facet_str(mshard_builder_t *b, int f_priority, const char *str) {
    add_facet_id(b, f_priority, m_facet_id(str, strlen(str)), str);
}

void msb_add_facet_num(mshard_builder_t *b, int f_priority, double d) {
    char numstr[32];
    snprintf(numstr, sizeof(numstr), "%f", d);              /* index_number_facet, sindex.c:645-649 */
    msb_add_facet_str(b, f_priority, numstr);
}

void msb_add_bool(mshard_builder_t *b, int i_priority, int val) {
    u32vec_t *v = &b->bp[i_priority][val ? 1 : 0];
    if (v->n && v->a[v->n - 1] == b->docid) return;
    u32vec_push(v, b->docid);
}

static int cmp_widpos(const void *pa, const void *pb) {     /* compare_wpos, sindex.c:51-57 */
    const widpos_t *a = pa, *c = pb;
    if (a->wid != c->wid) return a->wid < c->wid ? -1 : 1;
    if (a->priority != c->priority) return a->priority < c->priority ? -1 : 1;
    if (a->position != c->position) return a->position < c->position ? -1 : 1;
    return 0;
}

static void dd_reserve(mshard_builder_t *b, uint64_t extra) {
    if (b->dd_n + extra <= b->dd_cap) return;
    uint64_t nc = b->dd_cap ? b->dd_cap : (1 << 16);
    while (nc < b->dd_n + extra) nc *= 2;
    b->dd = realloc(b->dd, nc);
    b->dd_cap = nc;
}

/* store_docdata + gen_wordpos_data (sindex.c:273-326, 409-496) */
void msb_end_doc(mshard_builder_t *b) {
    size_t size = (size_t)b->num_numbers * sizeof(double) + sizeof(uint32_t);
    size += (size_t)b->num_facets * sizeof(uint32_t);
    for (int i = 0; i < b->num_facets; i++) size += (size_t)b->facet_data[i].n * sizeof(uint32_t);
    size = (size + 7) & ~(size_t)7;

    uint32_t n = b->nwp;
    if (n > 1) qsort(b->wp, n, sizeof(widpos_t), cmp_widpos);
    uint32_t numwid = 0;
    for (uint32_t x = 0; x < n; x++) if (x == 0 || b->wp[x].wid != b->wp[x - 1].wid) numwid++;

    /* upper bound of the word part: header + infos + (freq, priority, count, 5-byte vint, 0xFF) */
    uint64_t wp_bound = n ? 2 + (uint64_t)numwid * 8 + (uint64_t)n * 9 + 16 : 0;
    dd_reserve(b, size + wp_bound + 8);
    uint8_t *base = b->dd + b->dd_n;
    memset(base, 0, size);
    uint32_t off32 = (uint32_t)size;
    memcpy(base, &off32, 4);
    memcpy(base + 4, b->num_data, (size_t)b->num_numbers * sizeof(double));
    uint8_t *fpos = base + 4 + (size_t)b->num_numbers * 8;
    for (int i = 0; i < b->num_facets; i++) {
        uint32_t cnt = b->facet_data[i].n;
        memcpy(fpos, &cnt, 4); fpos += 4;
        if (cnt) { memcpy(fpos, b->facet_data[i].a, (size_t)cnt * 4); fpos += (size_t)cnt * 4; }
    }

    uint64_t wlen = 0;
    if (n) {
        uint8_t *data = base + size;
        uint16_t nw16 = (uint16_t)numwid;
        memcpy(data, &nw16, 2);
        uint8_t *infos = data + 2;
        uint8_t *dpos = infos + (size_t)numwid * 8;
        uint32_t last_wid = 0;
        uint8_t last_priority = 0xFF;
        uint8_t *wid_count = NULL;
        uint32_t wi = 0;
        uint32_t x = 0;
        while (x < n) {
            widpos_t *wp = &b->wp[x];
            if (wp->wid != last_wid) {
                last_wid = wp->wid;
                uint32_t freq = 1;
                while (x + freq < n && b->wp[x + freq].wid == wp->wid) freq++;
                uint32_t info;
                uint8_t *ip = infos + (size_t)wi * 8;
                wi++;
                memcpy(ip, &wp->wid, 4);
                if (freq == 1) {
                    info = 1u | (((uint32_t)wp->priority & 0xFF) << 1) | ((wp->position & 0x7FFFFF) << 9);
                    memcpy(ip + 4, &info, 4);
                    x++;
                    continue;
                }
                info = (((uint32_t)(dpos - data)) & 0x7FFFFF) << 9;
                memcpy(ip + 4, &info, 4);
                last_priority = 0xFF;
                *dpos++ = (uint8_t)freq;
            }
            if (last_priority != (uint8_t)wp->priority) {
                last_priority = (uint8_t)wp->priority;
                *dpos++ = (uint8_t)wp->priority;
                wid_count = dpos;
                *wid_count = 0;
                dpos++;
            }
            *wid_count = (uint8_t)(*wid_count + 1);
            dpos = write_vint(dpos, (int)wp->position);
            if (!((x + 1) < n && b->wp[x + 1].wid == last_wid &&
                  (uint8_t)b->wp[x + 1].priority == last_priority)) {
                *dpos++ = 0xFF;
            }
            x++;
        }
        wlen = (uint64_t)(dpos - data);
    }
    b->total_wp += numwid;

    if ((uint64_t)b->docid + 2 >= b->off_cap) {
        uint64_t nc = b->off_cap * 2;
        b->dd_off = realloc(b->dd_off, nc * sizeof(uint64_t));
        memset(b->dd_off + b->off_cap, 0, (nc - b->off_cap) * sizeof(uint64_t));
        b->off_cap = nc;
    }
    b->dd_off[b->docid] = b->dd_n;
    b->dd_n += size + wlen;
    /* keep every blob 8-byte aligned so doubles can be read in place (LMDB values are) */
    { const uint64_t end = b->dd_n; b->dd_n = (b->dd_n + 7) & ~(uint64_t)7; memset(b->dd + end, 0, b->dd_n - end); }
    b->dd_off[b->docid + 1] = b->dd_n;
    b->in_doc = 0;
}

/* ------------------------------------------------------------------ finalize */
typedef struct { double v; uint32_t d; } npair_t;
static int cmp_npair(const void *pa, const void *pb) {     /* double_compare (sindex.c:61-64) + INTEGERDUP */
    const npair_t *a = pa, *b = pb;
    if (a->v != b->v) return a->v > b->v ? -1 : 1;
    if (a->d != b->d) return a->d < b->d ? -1 : 1;
    return 0;
}
typedef struct { uint32_t id; uint32_t d; } fpair_t;
static int cmp_fpair(const void *pa, const void *pb) {
    const fpair_t *a = pa, *b = pb;
    if (a->id != b->id) return a->id < b->id ? -1 : 1;
    if (a->d != b->d) return a->d < b->d ? -1 : 1;
    return 0;
}
typedef struct { uint32_t id; char *s; } fstr_t;
static int cmp_fstr(const void *pa, const void *pb) {
    const fstr_t *a = pa, *b = pb;
    return a->id < b->id ? -1 : (a->id > b->id);
}

mshard_t *msb_finalize(mshard_builder_t *b) {
    mshard_t *s = calloc(1, sizeof(*s));
    s->max_docid = b->docid;
    s->num_strings = b->num_strings; s->num_numbers = b->num_numbers;
    s->num_bools = b->num_bools; s->num_facets = b->num_facets;
    s->dict = b->dict;
    s->dd = b->dd; s->dd_off = b->dd_off;
    if (!s->dd) s->dd = calloc(8, 1);
    uint32_t W = mdict_word_count(b->dict), T = mdict_tword_count(b->dict);
    if (b->raw_max_wid > W) { W = b->raw_max_wid; ensure_wid_cap(b, W); }
    s->word_count = W; s->tword_count = T;

    /* wid2bmap: count, then fill, walking the blobs in docid order */
    s->wp_off = calloc((size_t)W + 2, sizeof(uint64_t));
    s->tp_off = calloc((size_t)T + 2, sizeof(uint64_t));
    s->tw_off = calloc((size_t)T + 2, sizeof(uint64_t));
    uint32_t *tstamp = calloc((size_t)T + 1, sizeof(uint32_t));
    for (int pass = 0; pass < 2; pass++) {
        uint64_t *wcur = NULL, *tcur = NULL;
        if (pass == 1) {
            /* counts sit at off[id + 1]; an inclusive scan makes off[id] the start of id's list */
            for (uint32_t w = 0; w <= W; w++) s->wp_off[w + 1] += s->wp_off[w];
            for (uint32_t t = 0; t <= T; t++) s->tp_off[t + 1] += s->tp_off[t];
            s->wp_doc = malloc((s->wp_off[W + 1] ? s->wp_off[W + 1] : 1) * sizeof(uint32_t));
            s->tp_doc = malloc((s->tp_off[T + 1] ? s->tp_off[T + 1] : 1) * sizeof(uint32_t));
            wcur = malloc(((size_t)W + 2) * sizeof(uint64_t));
            tcur = malloc(((size_t)T + 2) * sizeof(uint64_t));
            memcpy(wcur, s->wp_off, ((size_t)W + 2) * sizeof(uint64_t));
            memcpy(tcur, s->tp_off, ((size_t)T + 2) * sizeof(uint64_t));
            memset(tstamp, 0, ((size_t)T + 1) * sizeof(uint32_t));
        }
        for (uint32_t d = 1; d <= s->max_docid; d++) {
            const uint8_t *blob = s->dd + s->dd_off[d];
            uint64_t bsz = s->dd_off[d + 1] - s->dd_off[d];
            uint32_t off; memcpy(&off, blob, 4);
            if (off + 2 > bsz) continue;
            const uint8_t *wpd = blob + off;
            uint16_t numwid; memcpy(&numwid, wpd, 2);
            /* a doc without words has no word part; its padding reads as numwid == 0 */
            for (uint32_t i = 0; i < numwid; i++) {
                uint32_t wid; memcpy(&wid, wpd + 2 + (size_t)i * 8, 4);
                if (pass == 0) s->wp_off[wid + 1]++;
                else s->wp_doc[wcur[wid]++] = d;
                int lim = b->wid_lim[wid];
                for (int k = 0; k < lim; k++) {
                    uint32_t tw = b->wid_tw[(size_t)wid * 3 + k];
                    if (tstamp[tw] == d) continue;
                    tstamp[tw] = d;
                    if (pass == 0) s->tp_off[tw + 1]++;
                    else s->tp_doc[tcur[tw]++] = d;
                }
            }
        }
        if (pass == 1) { free(wcur); free(tcur); }
    }
    free(tstamp);

    /* twid2widbmap: wids ascending under each twid */
    for (uint32_t w = 1; w <= W; w++)
        for (int k = 0; k < b->wid_lim[w]; k++) s->tw_off[b->wid_tw[(size_t)w * 3 + k] + 1]++;
    {
        for (uint32_t t = 0; t <= T; t++) s->tw_off[t + 1] += s->tw_off[t];
        s->tw_wid = malloc((s->tw_off[T + 1] ? s->tw_off[T + 1] : 1) * sizeof(uint32_t));
        uint64_t *cur = malloc(((size_t)T + 2) * sizeof(uint64_t));
        memcpy(cur, s->tw_off, ((size_t)T + 2) * sizeof(uint64_t));
        for (uint32_t w = 1; w <= W; w++)
            for (int k = 0; k < b->wid_lim[w]; k++) s->tw_wid[cur[b->wid_tw[(size_t)w * 3 + k]]++] = w;
        free(cur);
    }

    /* facetid2bmap */
    for (int f = 0; f < b->num_facets; f++) {
        fpairs_t *fp = &b->fp[f];
        fpair_t *pairs = malloc((fp->n ? fp->n : 1) * sizeof(fpair_t));
        for (uint64_t i = 0; i < fp->n; i++) { pairs[i].id = fp->id[i]; pairs[i].d = fp->doc[i]; }
        qsort(pairs, fp->n, sizeof(fpair_t), cmp_fpair);
        uint64_t m = 0;                                        /* a bitmap holds a doc once */
        for (uint64_t i = 0; i < fp->n; i++)
            if (i == 0 || pairs[i].id != pairs[i - 1].id || pairs[i].d != pairs[i - 1].d) pairs[m++] = pairs[i];
        uint32_t nid = 0;
        for (uint64_t i = 0; i < m; i++) if (i == 0 || pairs[i].id != pairs[i - 1].id) nid++;
        s->nfid[f] = nid;
        s->fid[f] = malloc((nid ? nid : 1) * sizeof(uint32_t));
        s->fp_off[f] = calloc((size_t)nid + 1, sizeof(uint64_t));
        s->fp_doc[f] = malloc((m ? m : 1) * sizeof(uint32_t));
        uint32_t k = 0;
        for (uint64_t i = 0; i < m; i++) {
            if (i == 0 || pairs[i].id != pairs[i - 1].id) { s->fid[f][k] = pairs[i].id; s->fp_off[f][k] = i; k++; }
            s->fp_doc[f][i] = pairs[i].d;
        }
        s->fp_off[f][nid] = m;
        free(pairs); free(fp->id); free(fp->doc);
    }
    /* facetid2str */
    {
        fstr_t *fs = malloc((b->nfstr ? b->nfstr : 1) * sizeof(fstr_t));
        for (uint32_t i = 0; i < b->nfstr; i++) { fs[i].id = b->fstr_id[i]; fs[i].s = b->fstr[i]; }
        qsort(fs, b->nfstr, sizeof(fstr_t), cmp_fstr);
        s->nfstr = b->nfstr;
        s->fstr_id = malloc((b->nfstr ? b->nfstr : 1) * sizeof(uint32_t));
        s->fstr = malloc((b->nfstr ? b->nfstr : 1) * sizeof(char *));
        for (uint32_t i = 0; i < b->nfstr; i++) { s->fstr_id[i] = fs[i].id; s->fstr[i] = fs[i].s; }
        free(fs); free(b->fstr_id); free(b->fstr);
        m_u32map_free(&b->fstr_idx);
    }
    /* boolid2bmap */
    for (int i = 0; i < b->num_bools; i++)
        for (int v = 0; v < 2; v++) { s->bp_n[i][v] = b->bp[i][v].n; s->bp_doc[i][v] = b->bp[i][v].a; }
    /* num_dbi */
    for (int i = 0; i < b->num_numbers; i++) {
        numpairs_t *np = &b->np[i];
        npair_t *pairs = malloc((np->n ? np->n : 1) * sizeof(npair_t));
        for (uint64_t k = 0; k < np->n; k++) { pairs[k].v = np->val[k]; pairs[k].d = np->doc[k]; }
        /* already sorted when values fall as docids rise (the common synthetic case) */
        int sorted = 1;
        for (uint64_t k = 1; k < np->n && sorted; k++) if (cmp_npair(&pairs[k - 1], &pairs[k]) > 0) sorted = 0;
        if (!sorted) qsort(pairs, np->n, sizeof(npair_t), cmp_npair);
        uint64_t m = 0;
        for (uint64_t k = 0; k < np->n; k++)
            if (k == 0 || pairs[k].v != pairs[k - 1].v || pairs[k].d != pairs[k - 1].d) pairs[m++] = pairs[k];
        s->np_n[i] = m;
        s->np_val[i] = malloc((m ? m : 1) * sizeof(double));
        s->np_doc[i] = malloc((m ? m : 1) * sizeof(uint32_t));
        for (uint64_t k = 0; k < m; k++) { s->np_val[i][k] = pairs[k].v; s->np_doc[i][k] = pairs[k].d; }
        free(pairs); free(np->val); free(np->doc);
    }
    for (int i = 0; i < b->num_facets; i++) free(b->facet_data[i].a);
    free(b->wp); free(b->wid_tw); free(b->wid_lim);
    free(b);
    return s;
}

void mshard_free(mshard_t *s) {
    if (!s) return;
    mdict_free(s->dict);
    free(s->dd); free(s->dd_off);
    free(s->wp_off); free(s->wp_doc); free(s->tp_off); free(s->tp_doc); free(s->tw_off); free(s->tw_wid);
    for (int f = 0; f < s->num_facets; f++) { free(s->fid[f]); free(s->fp_off[f]); free(s->fp_doc[f]); }
    for (uint32_t i = 0; i < s->nfstr; i++) free(s->fstr[i]);
    free(s->fstr_id); free(s->fstr);
    for (int i = 0; i < s->num_bools; i++) for (int v = 0; v < 2; v++) free(s->bp_doc[i][v]);
    for (int i = 0; i < s->num_numbers; i++) { free(s->np_val[i]); free(s->np_doc[i]); }
    free(s->used_doc);
    free(s);
}

const char *mshard_lookup_facet(const mshard_t *s, uint32_t facet_id) {
    uint32_t lo = 0, hi = s->nfstr;
    while (lo < hi) {
        uint32_t mid = (lo + hi) / 2;
        if (s->fstr_id[mid] < facet_id) lo = mid + 1; else hi = mid;
    }
    return (lo < s->nfstr && s->fstr_id[lo] == facet_id) ? s->fstr[lo] : NULL;
}

uint64_t mshard_wid_posting(const mshard_t *s, uint32_t wid, const uint32_t **docs) {
    if (wid == 0 || wid > s->word_count) { *docs = NULL; return 0; }
    uint64_t n = s->wp_off[wid + 1] - s->wp_off[wid];
    *docs = n ? s->wp_doc + s->wp_off[wid] : NULL;
    return n;
}

uint64_t mshard_twid_posting(const mshard_t *s, uint32_t twid, const uint32_t **docs) {
    if (twid == 0 || twid > s->tword_count) { *docs = NULL; return 0; }
    uint64_t n = s->tp_off[twid + 1] - s->tp_off[twid];
    *docs = n ? s->tp_doc + s->tp_off[twid] : NULL;
    return n;
}

uint64_t mshard_twid_wids(const mshard_t *s, uint32_t twid, const uint32_t **wids) {
    if (twid == 0 || twid > s->tword_count) { *wids = NULL; return 0; }
    uint64_t n = s->tw_off[twid + 1] - s->tw_off[twid];
    *wids = n ? s->tw_wid + s->tw_off[twid] : NULL;
    return n;
}

uint64_t mshard_facet_posting(const mshard_t *s, int f, uint32_t facet_id, const uint32_t **docs) {
    *docs = NULL;
    if (f < 0 || f >= s->num_facets) return 0;
    uint32_t lo = 0, hi = s->nfid[f];
    while (lo < hi) {
        uint32_t mid = (lo + hi) / 2;
        if (s->fid[f][mid] < facet_id) lo = mid + 1; else hi = mid;
    }
    if (lo >= s->nfid[f] || s->fid[f][lo] != facet_id) return 0;
    uint64_t n = s->fp_off[f][lo + 1] - s->fp_off[f][lo];
    *docs = s->fp_doc[f] + s->fp_off[f][lo];
    return n;
}

uint64_t mshard_bool_posting(const mshard_t *s, int i, int val, const uint32_t **docs) {
    *docs = NULL;
    if (i < 0 || i >= s->num_bools) return 0;
    *docs = s->bp_doc[i][val ? 1 : 0];
    return s->bp_n[i][val ? 1 : 0];
}

const uint8_t *mshard_docdata(const mshard_t *s, uint32_t docid, uint32_t *size) {
    if (docid == 0 || docid > s->max_docid) { if (size) *size = 0; return NULL; }
    if (size) *size = (uint32_t)(s->dd_off[docid + 1] - s->dd_off[docid]);
    return s->dd + s->dd_off[docid];
}

/* ------------------------------------------------------------------ synthetic corpora */
static inline uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline uint32_t rnd_range(uint64_t *st, uint32_t a, uint32_t b) {   /* inclusive, like randint */
    return a + (uint32_t)(splitmix64(st) % (uint64_t)(b - a + 1));
}

int msynth_route(uint64_t g, int num_shards) {
    uint64_t x = g * 0x9e3779b97f4a7c15ULL;
    x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 32;
    return (int)(x % (uint64_t)num_shards);
}

mshard_t *msynth_datagen(uint32_t rend, uint64_t seed, const char **facets, int nfacets,
                         uint32_t *facet_counts) {
    /* indexedFields str, strlist, facet, facetlist, spell | id, num, numlist | bool;
     * facetFields facet, facetlist, facetonly (test/py/pytests.py:85-103) */
    mshard_builder_t *b = msb_new(5, 3, 1, 3);
    static const char *strs[4] = { "aaaaaaa attr", "aaaaaaa bbbbbbb attr", "aaaaaaa bbbbbbb ccccccc attr",
                                   "aaaaaaa bbbbbbb ccccccc ddddddd attr" };
    static const char *sl[4] = { "zzzzzzz", "yyyyyyy", "xxxxxxx", "wwwwwww" };
    static const char *fo[4] = { "oneone", "twotwo", "threethree", "fourfour" };
    uint64_t st = seed;
    int flen = nfacets - 1;
    if (facet_counts) memset(facet_counts, 0, sizeof(uint32_t) * (size_t)nfacets);
    int state = 0;
    for (uint32_t i = 1; i <= rend; i++) {
        int pick[4];
        pick[0] = (int)rnd_range(&st, 0, 5);
        pick[1] = (int)rnd_range(&st, 6, 50);
        pick[2] = (int)rnd_range(&st, 51, (uint32_t)flen);
        pick[3] = (int)rnd_range(&st, 51, (uint32_t)flen);
        if (facet_counts) for (int k = 0; k < 4; k++) facet_counts[pick[k]]++;
        msb_begin_doc(b);
        /* field order as printed by data_gen.py:41-80 */
        msb_add_facet_str(b, 0, facets[pick[0]]);
        msb_add_string(b, 2, facets[pick[0]]);
        for (int k = 1; k < 4; k++) { msb_add_facet_str(b, 1, facets[pick[k]]); msb_add_string(b, 3, facets[pick[k]]); }
        msb_add_bool(b, 0, i % 10 == 0);
        if (i < 51) msb_add_string(b, 4, (i % 5 == 0) ? "eeeeeee" : "eeeeeed");
        msb_add_string(b, 0, strs[state]);
        for (int k = 0; k <= state; k++) msb_add_string(b, 1, sl[k]);
        for (int k = 0; k <= state; k++) msb_add_facet_str(b, 2, fo[k]);
        msb_add_number(b, 1, (double)((rend + 1) - i), 1);
        uint32_t mul = 2 * (uint32_t)(state + 1);
        msb_add_number(b, 2, (double)(i + rend * mul), 1);
        msb_add_number(b, 2, (double)(i + 1 + rend * mul), 1);
        msb_add_number(b, 2, (double)(i + 3 + rend * mul), 1);
        msb_add_number(b, 0, (double)i, 1);
        msb_end_doc(b);
        state = (state + 1) & 3;
    }
    return msb_finalize(b);
}

/* bijection on [0, 2^k) used to scatter frequency ranks over the 4-letter stems */
static inline uint32_t perm_rank(uint32_t r, uint32_t vocab) {
    uint32_t bits = 1; while ((1u << bits) < vocab) bits++;
    uint32_t mask = (1u << bits) - 1;
    uint32_t x = r;
    do {                        /* cycle-walk so the image stays below vocab */
        x = (x * 0x9E3779B1u + 0x7F4A7C15u) & mask;
        x ^= x >> (bits / 2 + 1);
        x = (x * 0x85EBCA6Bu) & mask;
    } while (x >= vocab);
    return x;
}

int msynth_zipf_word(uint32_t vocab, uint64_t seed, uint32_t r, char *out) {
    /* note: perm_rank is a composition of bijections on [0,2^bits) restricted by cycle walking */
    uint32_t code = perm_rank(r, vocab);
    uint64_t st = seed ^ (0xD6E8FEB86659FD93ULL * (uint64_t)(r + 1));
    uint64_t h = splitmix64(&st);
    int len = 4 + (int)(h % 7);
    h >>= 8;
    for (int i = 3; i >= 0; i--) { out[i] = (char)('a' + code % 26); code /= 26; }
    for (int i = 4; i < len; i++) { out[i] = (char)('a' + h % 26); h /= 26; }
    out[len] = 0;
    return len;
}

mshard_t *msynth_zipf(uint64_t total_docs, uint32_t vocab, uint32_t hc_buckets, uint64_t seed,
                      int shard, int num_shards) {
    return msynth_zipf_first(total_docs, vocab, hc_buckets, seed, shard, num_shards, total_docs);
}
/* the same corpus cut off after its first `first_docs` documents: what the shard held before the rest was written */
mshard_t *msynth_zipf_first(uint64_t total_docs, uint32_t vocab, uint32_t hc_buckets, uint64_t seed,
                            int shard, int num_shards, uint64_t first_docs) {
    /* strings: str | numbers: num, price | bools: flag | facets: cat, hc */
    mshard_builder_t *b = msb_new(1, 2, 1, 2);
    /* the vocabulary goes in first, so wid = rank + 1 in every shard */
    {
        char w[16]; m_chr_t cw[16]; uint32_t tw[3];
        for (uint32_t r = 0; r < vocab; r++) {
            int len = msynth_zipf_word(vocab, seed, r, w);
            for (int i = 0; i < len; i++) cw[i] = (m_chr_t)w[i];
            uint32_t wid = mdict_insert(b->dict, cw, len, tw);
            ensure_wid_cap(b, wid);
            b->wid_lim[wid] = 3;
            for (int i = 0; i < 3; i++) b->wid_tw[(size_t)wid * 3 + i] = tw[i];
        }
    }
    const double lnv = log((double)vocab + 1.0);
    const double ln65 = log(65.0);
    char buf[32];
    for (uint64_t g = 1; g <= total_docs && g <= first_docs; g++) {
        if (msynth_route(g, num_shards) != shard) continue;
        uint64_t st = seed ^ (g * 0xA24BAED4963EE407ULL);
        msb_begin_doc(b);
        uint32_t ntok = rnd_range(&st, 8, 24);
        for (uint32_t p = 1; p <= ntok; p++) {
            double u = (double)(splitmix64(&st) >> 11) * (1.0 / 9007199254740992.0);
            uint32_t r = (uint32_t)exp(u * lnv) - 1;           /* log-uniform ~ Zipf s=1 */
            if (r >= vocab) r = vocab - 1;
            push_widpos(b, r + 1, 0, p);
        }
        double u = (double)(splitmix64(&st) >> 11) * (1.0 / 9007199254740992.0);
        uint32_t cat = (uint32_t)exp(u * ln65) - 1; if (cat > 63) cat = 63;
        snprintf(buf, sizeof(buf), "cat%02u", cat);
        msb_add_facet_str(b, 0, buf);
        uint32_t hc = (uint32_t)(splitmix64(&st) % hc_buckets);
        snprintf(buf, sizeof(buf), "b%06u", hc);
        msb_add_facet_str(b, 1, buf);
        msb_add_number(b, 0, (double)(total_docs + 1 - g), 1);
        msb_add_number(b, 1, (double)(splitmix64(&st) % 1000000ULL), 1);
        msb_add_bool(b, 0, g % 10 == 0);
        msb_end_doc(b);
    }
    return msb_finalize(b);
}
