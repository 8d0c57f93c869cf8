// mgpu_types.h -- structures shared by the host runtime and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include "marlin_gpu.h"

#define MG_BLOCK        256          // threads per workgroup (4 waves of 64)
#define MG_WORDS_PER_BLOCK 256       // u64 bitmap words per hit-count block (blk_cnt granularity)
#define MG_FORMULA_WORDS 512         // u64 bitmap words (= 32768 docids, one "window") owned by a formula workgroup
#define MG_ORDER_WINDOWS 2048        // formula windows per squery that k_order handles (= 67M docids per shard)
#define MG_SHORT_POSTING 32          // sparse postings up to this length carry no window index: every window scans them whole
#define MG_MAX_STACK    8            // filter program stack depth
#define MG_MCAP         8            // matched docwords kept per document in LDS before the overflow path
#define MG_PCAP         12           // (slot,position) pairs per document in LDS before the overflow path
#define MG_SEL_BINS     2048         // radix-select digit = 11 bits
#define MG_SEL_DIGIT    11
#define MG_SEL_PASSES   3            // select passes launched per round (lists <= 32768 finish inside one launch)
#define MG_SEL_SMALL    512          // final sets up to this size use the small sort workgroup
#define MG_SEL_FINAL    2048         // candidates sorted by the final block
#define MG_SCATTER_CHUNK 2048        // docids per scatter work item
#define MG_LDS_HASH     4096         // wid table slots staged in LDS (32 KiB)
#define MG_WID_HASH_K   0x9E3779u    // wid table hash: low 32 bits of (wid & 0xFFFFFF) * K, folded with >> 15
#define MG_WID_EMPTY    0xFFFFFFFFu  // key of an empty wid table slot
#define MG_LDS_HIST     8192         // facet bins privatised in LDS (32 KiB)

// Device view of one shard mirror.
struct ShardDev {
    uint32_t N;                      // bits per row = max_docid + 1
    uint32_t W;                      // u64 words per row
    uint32_t Wp;                     // row stride in u64 words (multiple of 32 = 256 B)
    uint32_t num_postings;
    const uint64_t *dense;           // [n_dense][Wp]
    const int32_t  *post_dense;      // [num_postings] dense row index or -1
    const uint64_t *post_off;        // [num_postings + 1] into post_doc (sparse postings only)
    const uint32_t *post_doc;
    // window index of the long sparse postings: post_win[base + j] = first element (relative to the posting's
    // start) whose docid is >= j * 64 * MG_FORMULA_WORDS, j = 0..nwin (the last entry = the posting's length)
    const uint32_t *post_win;
    // docwords: CSR over docids; entry = {wid, info}; info bit0 = single position inline
    const uint32_t *dw_off;          // [N + 1]
    const uint2    *dw;              // {wid, info}
    const uint32_t *pp_off;          // [N + 1] decoded position pool ((priority << 16) + position)
    const uint32_t *pp;
    // doc-major records: one aligned record of rec_stride bytes per docid:
    //   u32 nwords | nin << 16 ; u32 pbase ; u32 ovf (= dw_off[doc] + nin) ; u32 nnum_in | facets_in << 8 | ws << 16
    //   f64 x nnum_in ; [facet area of a shard-constant size: per field u32 count + ordinals] ; from 8-byte slot ws
    //   (= rec_ws, the same in every record): nin x {wid, info} ; then, when everything fits, the document's decoded
    //   positions with bit 31 set (pbase = 0x80000000 | their u32 index in rec) ; zero padding
    const uint32_t *rec;
    const uint8_t  *rec_len;         // [N] 16-byte units of each record that hold anything, or NULL (docids beyond 24 bits)
    uint32_t rec_ws;                 // first docword slot of every record
    uint32_t rec_words;              // u32 words in rec (checked accessors)
    uint32_t rec_stride;             // bytes: 128, 256, 384 or 512
    uint32_t rec_nnum;               // numbers held in every record (all of them when num_numbers <= 4, else 0)
    // facets
    int32_t  num_facets, num_numbers;
    const uint32_t *fa_off[MGPU_MAX_FIELDS];    // [N + 1]
    const uint32_t *fa_ord[MGPU_MAX_FIELDS];    // dense ordinals
    const uint32_t *fa_ord2id[MGPU_MAX_FIELDS]; // ordinal -> facet id
    const uint32_t *fa_sorted_id[MGPU_MAX_FIELDS]; // the field's facet ids ascending (position = offset of the id's posting from fac_pid_base)
    uint32_t fa_nord[MGPU_MAX_FIELDS];
    // numbers
    const double   *num_col[MGPU_MAX_FIELDS];   // [N]
    const uint32_t *np_doc[MGPU_MAX_FIELDS];
    // idnum2dbl (sindex.c:528-565): bit g of grp_has[i] = some document of docids [1000 g, 1000 g + 1000) indexed a
    // number of field i, i.e. the group's array exists (what get_num_value reads in fast-rank mode)
    const uint64_t *grp_has[MGPU_MAX_FIELDS];
    // extents of the data-indexed arrays (checked accessors, mgpu_kernels.hip): entries of dw, of pp
    uint32_t dw_len, pp_len;
};

// layout of the packed 128-bit rank key: fields in rank-rule order, most significant first
struct KeyField { int8_t kind; uint8_t bits; uint8_t shift; uint8_t hi; };   // hi: 1 = in key_hi
// kinds
enum { KF_TYPO, KF_PROX, KF_POS, KF_EXACT, KF_FIELD, KF_COMP_DESC, KF_COMP_ASC };

struct SelJob {                      // one top-k selection (rank keys of a squery, or facet bins)
    const uint64_t *key_hi, *key_lo;
    uint32_t *cand[2];               // ping-pong candidate index buffers (capacity m_cap)
    uint32_t *hist;                  // [MG_SEL_BINS]
    uint32_t *ctl;                   // control words, see SEL_* below
    const uint32_t *m_ptr;           // number of keys (device word)
    uint32_t k;
    uint32_t m_cap;
    uint32_t *out_idx;               // [k] selected indices, sorted by (key, idx)
    uint32_t *out_n;                 // number selected = min(k, m)
    const uint64_t *mask;            // [2] varying bits (or ^ and): hi, lo ; NULL = all bits vary
    uint32_t skip_ones;              // facet jobs: never select keys with key_hi == ~0 (zero counts)
    // facet jobs whose keys the select stage derives itself (no k_fkeys launch): key i is
    // (0xFFFFFFFF - gen_hist[b], b) with b = gen_touched[i], emitted as gen_ord2id[b]; key_hi/key_lo are then scratch
    // the selecting workgroup may fill; NULL = the keys are already in key_hi/key_lo
    const uint32_t *gen_touched, *gen_hist, *gen_ord2id;
    // facet jobs of a directly counted field (SqDev::fdirect_mask) whose distinct values k_facet_hash counts in an LDS
    // hash table straight from the hits' ordinals whenever the squery has at most MG_FHASH_MAX_M hits: no histogram
    // is touched then (the rank kernel skips its atomics, k_select_first skips the job)
    uint32_t fhash;
};
#define MG_FHASH_SMALL_M 1536u       // ... of which the 256-lane / 4096-slot size of k_facet_hash takes those up to here
#define MG_FHASH_MAX_M 8192u         // most hits for which a sparse facet field is counted by hashing in LDS
#define MG_FHASH_MAX_K 56u           // largest k of such a job (its merge buffer holds 9 k keys)
#define MG_FAC_NONE   0xFFFFFFFFu     // fac_ord: the document has no value in the field
#define MG_FAC_LOOKUP 0xFFFFFFFEu     // fac_ord: several values, or not staged: k_docproc reads the record / CSR itself
#define MG_SEL_MID_KEYS 131072u      // longest fresh list one 1024-thread workgroup selects (k_select_mid)
#define MG_FDIRECT_MAX_CAP MG_SEL_MID_KEYS // largest hit bound (SqDev::hit_cap) of a squery whose sparse single-valued field is counted from the rank kernel
#define MG_SEL_SOLO_K 512u           // largest k of the one-workgroup selections (= MG_SEL_SMALL: their final set is ranked in LDS)
#define MG_CARD_IN_RES 4096u         // a cardinality aggregation over a field of at most this many values returns its id set in the results region
// ctl words
enum { SEL_NCAND, SEL_KREM, SEL_NWIN, SEL_DONE, SEL_TICKET, SEL_NEXT, SEL_BITPOS, SEL_PASS, SEL_WORDS = 16 };

struct AggPart { double sum, mn, mx; unsigned long long cnt; };

// Device-side descriptor of one (shard, query) pair.
struct SqDev {
    ShardDev sh;
    // ---- formula ----
    int32_t  num_terms, num_words;
    int32_t  empty;                  // plan says: result is empty (filter_null)
    uint32_t term_dense_off[MGPU_MAX_TERMS + 1];   // into blob (u32 dense row indices)
    int32_t  term_row[MGPU_MAX_TERMS];             // scratch row index holding scattered sparse postings, or -1
    uint64_t *rows;                  // scratch rows [nrows][Wp], zeroed per query
    // the plan flattened for k_formula, plan_len u32 words at blob + plan_off:
    //   [0 .. num_terms]   dn[t]: dense rows of term t = list entries [dn[t], dn[t + 1])
    //   [num_terms + 1 ..] the dense-row list, entry = row index | term << 27
    //   [plan_fops ..]     the filter ops {op, a} pairs; a >= 0 dense row, < 0 ~scratch row
    uint32_t plan_off, plan_len, plan_fops;   // plan_fops: word index of the first filter op inside the plan
    // sparse postings of the terms: 4 u32 per list {term, first element in post_doc, length, post_win base or ~0}
    // at blob + sl_off (16-byte aligned), lists of term t = [term_sl[t], term_sl[t + 1])
    uint32_t sl_off;
    uint32_t term_sl[MGPU_MAX_TERMS + 1];
    // ---- filter ----
    int32_t  num_fops;
    uint32_t fops_off;               // into blob: 4 u32 per op {op, a, _, _}; operand a: dense row (>=0) or ~scratch row
    // ---- outputs of formula/compact ----
    uint64_t *docid_map;             // [Wp], or NULL: only written for the parity captures
    uint2    *win_info;              // [ceil(nblk / 2)] per formula window: {hits, base of its range in hits_raw}
    uint32_t *hits_raw;              // [hit_cap] hits grouped by window, windows in completion order
    uint32_t *raw_count;             // device word, zeroed per query: hits_raw entries handed out
    uint32_t nblk;
    uint32_t *num_hits;              // device word
    uint32_t *hits;                  // [hit_cap] ascending docids
    uint32_t hit_cap;                // host-side upper bound of the hit count; sizes every per-hit array
    // ---- rank ----
    int32_t  do_rank;                // hits_per_page > 0 && num_words > 0
    int32_t  rank_by;
    uint32_t hash_off, hash_mask;    // wid table in blob: 2 u32 per slot {wid, val}; wid 0 = empty
    uint16_t exact_present;
    int32_t  nkf;
    KeyField kf[8];
    uint64_t *key_hi, *key_lo;       // [N]
    uint64_t *key_andor;             // [4] and_hi, and_lo, or_hi, or_lo
    uint32_t *ovf_cnt;               // overflow doc list counter
    uint32_t *ovf_list;              // [N] hit indices whose position lists overflowed
    // ---- doc processing (facets + aggs) ----
    uint32_t hist_mask;              // bit f: histogram field f
    // bit f: field f is counted straight from the rank kernel (fire-and-forget atomics on hist[f], one per hit:
    // every document has at most one value and its facet section is staged); the hit's ordinals in fac_ord[f]
    // then ARE the list of touched bins (with repeats, which the exchange in the key derivation turns into
    // zero counts), k_docproc skips the field and no touched list is built
    uint32_t fdirect_mask;
    // bit f (a subset of fdirect_mask): with at most MG_FHASH_MAX_M hits the field is not counted into hist[f] at all,
    // k_facet_hash counts it from fac_ord[f] (SelJob::fhash)
    uint32_t fhash_mask;
    uint32_t fhash_njobs, fhash_jobs[4];   // the select jobs of those fields (k_order hands them to k_facet_hash by hit count)
    int32_t  fac_in_rank;            // the rank kernel leaves every hit's facet ordinals in fac_ord while it holds the records
    uint32_t *fac_ord[MGPU_MAX_FIELDS];   // [hit_cap] per histogram field: the hit's ordinal, MG_FAC_NONE, or MG_FAC_LOOKUP
    uint32_t *hist[MGPU_MAX_FIELDS]; // [fa_nord[f]] zeroed per query
    uint32_t *touched[MGPU_MAX_FIELDS];  // [fa_nord[f]] bins hit at least once, in first-touch order
    uint32_t *ntouched[MGPU_MAX_FIELDS]; // device word, zeroed per query
    int32_t  num_aggs;
    int32_t  agg_type[MGPU_MAX_AGGS];
    int32_t  agg_field[MGPU_MAX_AGGS];
    AggPart *agg_part;               // [nproc_blocks][num_aggs]
    AggPart *agg_out;                // [num_aggs]
    uint32_t nproc_blocks;
    // facet selection keys
    uint64_t *fkey_hi[MGPU_MAX_FIELDS], *fkey_lo[MGPU_MAX_FIELDS];
    // ---- fast-rank (perform_fast_ranking, docrank.c:478-560) ----
    int32_t  aux;                    // 1 = the zero-typo twin of a fast-eligible squery: only its docid_map and count are wanted
    int32_t  fast_eligible;          // not full_scan and rank_by set: ranked on a sample when *total_hits > fast_thresh
    uint32_t fast_thresh;
    int32_t  rank_asc;
    int32_t  fast_bad;               // what makes the mode impossible: 1 cardinality agg (-> MGPU_E_UNSUPPORTED),
                                     // 2 rank_by field with several values per document, 4 facet postings unknown (-> MGPU_E_LIMIT / UNSUPPORTED)
    const uint64_t *zmap;            // zero_typo_docid_map: the twin's docid_map
    const uint32_t *z_count;         // its cardinality (the twin's num_hits word)
    uint64_t *dbmap;                 // [Wp] documents already ranked
    uint32_t *fast_flag;             // result word: 1 = the sample was ranked
    uint32_t *total_hits;            // result word: num_hits of the squery when fast_flag (num_hits then counts the sample)
    uint32_t *num_rank;              // device word: keys the rank selection looks at (sample size - 1, docrank.c:550)
    const uint32_t *np_rank; uint64_t np_rank_n;    // num_dbi[rank_by] docids in key order
    uint32_t fac_pid_base[MGPU_MAX_FIELDS];
    // ---- status ----
    uint32_t *err;                   // device word, nonzero = MGPU_E_* (negated)
};

// a numeric-range filter operand built from the field's column (one value per document: ShardDev::num_col) instead of
// scattered from its sorted docid list: positions [lo, hi) of the num_dbi order are the documents whose value lies between
// the values at the two ends of the range (the host cut the range at value boundaries), so the row is one coalesced pass
// over the column -- 8 B per document of the shard, where the scatter costs an atomic per docid of the range
struct RangeRow { const double *col; uint64_t *dst; const uint32_t *order; uint64_t lo, hi; uint32_t N, W; };
#define MG_RANGE_ROW_FRACTION 14u    // a range of more than N / 14 docids is built from the column (27G scattered docids/s against 375G column values/s)
struct ScatterItem {                 // one chunk of a sorted docid list OR-ed into a scratch row
    const uint32_t *src;
    uint64_t *dst;
    uint32_t n;
    uint32_t pad;
};

// one (term, dictionary) pair of the device term lookup (k_term_scan)
struct TermJob {
    const uint32_t *chars, *off, *wid;   // the dictionary: flat code points, word offsets (nwords + 1), word ids
    uint32_t nwords;
    uint32_t term_off;                   // the term's code points in the term blob
    int32_t tlen, prefix, maxdist;
    uint32_t tsig;                       // mg_lev_sig of the term
    uint32_t lookup, pad;                // the caller's index of this lookup (the job table is sorted by dictionary)
};
// one dictionary of a lookup call and the run of jobs (sorted by dictionary) that read it
struct TermDict {
    const uint32_t *sig;                 // mg_lev_sig per word
    uint32_t nwords, job_lo, job_hi, pad;
};
