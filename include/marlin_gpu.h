/* marlin_gpu.h -- C ABI of libmarlin_gpu.so, the MI355X shard-query engine.
 *
 * Drop-in boundary for Marlin's per-shard query path.  Marlin itself has no plugin or FFI
 * interface: the seam is the pair of C symbols in src/main/squery.h:59-60
 *      void execute_squery(void *w);      void sqresult_free(struct query *, struct squery_result *);
 * A replacement squery.c keeps those two symbols, turns its `struct squery` into the flat
 * `mgpu_plan_t` below (term lookup stays on the host trie), calls mgpu_execute() and copies
 * the `mgpu_sqout_t` back into `struct squery_result` (INTEGRATION.md shows the stub).
 * Plain C, plain pointers and sizes, int status codes (0 = ok); no torch / HIP types.
 *
 * Every entry point cites the reference code whose work it takes over
 * (paths relative to /root/reference/src/main).
 */
#ifndef MARLIN_GPU_H
#define MARLIN_GPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGPU_MAX_TERMS   32
#define MGPU_MAX_WORDS   16
#define MGPU_MAX_FIELDS  32
#define MGPU_MAX_AGGS    16
#define MGPU_MAX_RULES   10
#define MGPU_MAX_K       1000      /* MAX_HITS_LIMIT, index.h:22 */

enum {
    MGPU_OK = 0,
    MGPU_E_HIP = -1,          /* a HIP call failed; see mgpu_last_error() */
    MGPU_E_ARG = -2,          /* malformed argument */
    MGPU_E_LIMIT = -3,        /* beyond a documented limit of this implementation */
    MGPU_E_SCRATCH = -4,      /* position scratch pool exhausted (raise mgpu_ctx_create's pool) */
    MGPU_E_UNSUPPORTED = -5   /* e.g. R_GEO rule; fast-rank where the reference itself misbehaves (see mgpu_plan_t) */
};

/* sort.h:7-18 */
enum { MGPU_R_TYPO, MGPU_R_PROX, MGPU_R_POS, MGPU_R_EXACT, MGPU_R_FIELD, MGPU_R_GEO, MGPU_R_DONE,
       MGPU_R_COMP, MGPU_R_COMP_ASC, MGPU_R_MAX };
/* aggs.h:15-23 */
enum { MGPU_A_ROOT, MGPU_A_MAX, MGPU_A_MIN, MGPU_A_AVG, MGPU_A_STATS, MGPU_A_CARDINALITY };

typedef struct mgpu_shard mgpu_shard_t;   /* HBM mirror of one shard */
typedef struct mgpu_ctx   mgpu_ctx_t;     /* one execution context = one HIP stream + workspace */

const char *mgpu_last_error(void);
int mgpu_device_count(void);

/* ---------------------------------------------------------------- mirror ------------------
 * Source formats: the sets stored by mbmap (mbmap.c:253-283 header + container keys; read as stored through
 * post_mbmap / mgpu_mbmap_decode, or pre-expanded to docid lists by the caller), the
 * docid2data blob (sindex.c:273-326, 409-496) and num_dbi (sindex.c:528-535).  All arrays are
 * host memory, copied during the call.  Posting ids ("pid") are chosen by the caller; pid 0 is
 * reserved for "all used docids" (shard_get_all_docids, shard.c:51-53). */
/* One posting exactly as mbmap stores it in LMDB (mbmap.c:183-283): `header` is the value at key id,
 * [u16 num_c][u16 cid x num_c]; cont[i] is the value at key id + cid_i + 1, a container buffer
 * [u16 id][u16 card][payload]: card ascending u16 values when card <= 4096 (CUTOFF, cont.h:8, cont.c:118-130), else
 * 1024 little-endian u64 bitset words (BCUTOFF); a full container is stored as card = 0 with payload[0] = 0xFFFF
 * (cont_cardinality, cont.c:543-548).  A NULL cont[i] is a container whose mdb_get failed (mbmap.c:276-278): skipped. */
typedef struct mgpu_mbmap {
    const uint8_t *header; uint32_t header_len;
    const uint8_t *const *cont; const uint32_t *cont_len;      /* [num_c] */
} mgpu_mbmap_t;
/* bmap_iterate (bmap.c:370-374, cont.c:590-610) over such a posting: ascending docids into out[0..cap), *n = its
 * cardinality.  Host code only (no GPU needed).  MGPU_E_ARG for a malformed value, MGPU_E_LIMIT when cap is too small
 * (*n is still set; out may be NULL with cap 0 to ask for the size). */
int mgpu_mbmap_decode(const mgpu_mbmap_t *b, uint32_t *out, uint64_t cap, uint64_t *n);

typedef struct mgpu_shard_desc {
    uint32_t max_docid;                 /* docids 1..max_docid */
    int      num_numbers, num_facets;
    /* postings, CSR by pid; docids ascending */
    uint32_t        num_postings;       /* pids 0..num_postings-1 */
    const uint64_t *post_off;           /* [num_postings + 1] */
    const uint32_t *post_doc;
    /* ... or, instead, every posting in the reference's own stored form: [num_postings] entries (an entry with a NULL
     * header is an absent key = empty posting); when set, post_off / post_doc are ignored */
    const mgpu_mbmap_t *post_mbmap;
    /* docid2data, raw blobs in the reference layout */
    const uint64_t *dd_off;             /* [max_docid + 2], blob of doc d = dd[dd_off[d], dd_off[d+1]) */
    const uint8_t  *dd;
    /* num_dbi[i] docids in key order (value descending, docid ascending); values stay on the host */
    uint64_t        np_n[MGPU_MAX_FIELDS];
    const uint32_t *np_doc[MGPU_MAX_FIELDS];
} mgpu_shard_desc_t;

/* Builds the mirror on `device`: dense u64 rows for postings with >= 1/32 of the docs, sorted
 * docid lists for the rest, docword CSR + decoded position pool, per-field facet ordinals and
 * numeric columns decoded from the blobs.  Takes over mbmap_load_bmap + mdb_get(docid2data). */
int  mgpu_shard_create(int device, const mgpu_shard_desc_t *desc, mgpu_shard_t **out);
void mgpu_shard_destroy(mgpu_shard_t *s);
/* bytes of HBM held by the mirror */
uint64_t mgpu_shard_bytes(const mgpu_shard_t *s);

/* ---------------------------------------------------------------- mirror maintenance (row f-2) ------------------
 * A committed bulk write (si_write_end, sindex.c:184-254: the write cache's mbmaps saved with mbmap_save, mbmap.c:183-251;
 * one docid2data blob per new document, sindex.c:273-326; the new num_dbi keys, sindex.c:528-535) appends documents:
 * docids above the old maximum, which every posting they touch gains at its end, and possibly postings, words and facet
 * values the shard did not have.  mgpu_shard_apply_delta brings the mirror up to date from exactly that -- the new
 * documents' blobs and, per posting, the docids ADDED -- instead of a rebuild: the sparse posting array is re-laid on the
 * device (old run + new docids per posting, k_merge_postings) and its window index recomputed (k_build_winidx), dense rows
 * take their new bits in place (k_scatter), records / docwords / positions / columns / facet ordinals of the new documents
 * are appended into the headroom the mirror was built with (MGPU_MIRROR_HEADROOM, default 12.5 % + 4096 documents).
 * MGPU_E_LIMIT = no room left, MGPU_E_ARG = not a valid delta (nothing was changed either way): rebuild the mirror with
 * mgpu_shard_create.  Any other error (a device failure part-way) leaves the mirror half-updated: destroy it and rebuild.
 * The caller makes sure no batch that names this shard is in flight during the call.  Deleted or replaced documents are not a delta: rebuild. */
typedef struct mgpu_shard_delta {
    uint32_t max_docid;                 /* after the write; the new documents are old max_docid + 1 .. max_docid */
    uint32_t num_postings;              /* pids of the scheme after the write: 0..num_postings-1 */
    const uint32_t *old_pid;            /* [num_postings] the posting's pid before the write (ascending), 0xFFFFFFFF = a new
                                         * posting; NULL = the old pids keep their numbers and new postings follow them */
    const uint64_t *post_off;           /* [num_postings + 1] CSR by (new) pid of the docids ADDED to each posting: all of */
    const uint32_t *post_doc;           /* them new documents, ascending */
    const uint64_t *dd_off;             /* [new documents + 1]: blob of new document i (docid old max + 1 + i) = */
    const uint8_t  *dd;                 /* dd[dd_off[i], dd_off[i + 1]), reference layout as in mgpu_shard_desc_t */
    uint64_t        np_n[MGPU_MAX_FIELDS];      /* num_dbi[i] after the write, complete (4 bytes per document and field: */
    const uint32_t *np_doc[MGPU_MAX_FIELDS];    /* re-sent rather than merged) */
} mgpu_shard_delta_t;
int      mgpu_shard_apply_delta(mgpu_shard_t *s, const mgpu_shard_delta_t *delta);
uint32_t mgpu_shard_max_docid(const mgpu_shard_t *s);
uint32_t mgpu_shard_doc_capacity(const mgpu_shard_t *s);    /* largest max_docid a delta may bring */

/* ---------------------------------------------------------------- plan -------------------- */
/* filter program: postfix form of the reference's filter tree (filter_apply.c:207-227).
 * `docs` is the pre-filter docid_map, as in filter_apply's last argument. */
enum {
    MGPU_F_PUSH_POSTING,   /* a = pid            (str_eq_filter / bool_eq_filter) */
    MGPU_F_PUSH_NUMRANGE,  /* a = numeric field, lo..hi = [lo, hi) into np_doc (num_*_filter) */
    MGPU_F_PUSH_EMPTY,
    MGPU_F_AND,            /* pop 2, push a & b  (oper_and) */
    MGPU_F_OR,             /* pop 2, push a | b  (oper_or)  */
    MGPU_F_DOCS_ANDNOT     /* top = docs & ~top  (ne_filter / nin_filter) */
};
typedef struct mgpu_fop { int32_t op; int32_t a; uint64_t lo, hi; } mgpu_fop_t;

/* one entry of the wid table built from all_wordids + per-term wordids + exact wids
 * (squery.c:48-70, 100-111, 159-166; consumed by docrank.c:84,117,152,256) */
typedef struct mgpu_wid_entry {
    uint32_t wid;
    uint8_t  dist;        /* all_wordids value, 0xFF when absent */
    uint8_t  mterm;       /* get_matching_term: first term whose wordids holds wid, 0xFF none */
    uint16_t exact_mask;  /* bit i: wid is the exact wid of query word i */
} mgpu_wid_entry_t;

typedef struct mgpu_plan {
    /* terms: termdata[t].tbmap = OR of these postings (process_termresult, squery.c:72-118) */
    int32_t  num_terms;                         /* 0, 1, 3 or 2n (query.c:489-603) */
    int32_t  num_words;
    uint32_t term_npost[MGPU_MAX_TERMS];
    const uint32_t *term_post[MGPU_MAX_TERMS];  /* pids; the postings of words at distance 0 come FIRST ... */
    uint32_t term_nzero[MGPU_MAX_TERMS];        /* ... this many of them: termdata[t].tzero_typo_bmap (squery.c:107-113) */
    uint16_t exact_present;                     /* bit i: exact_docid_map[i] != NULL */
    /* wid table */
    uint32_t num_wids;
    const mgpu_wid_entry_t *wids;
    /* filter */
    int32_t  has_filter;
    int32_t  filter_null;                       /* filter_apply returned NULL -> empty result */
    int32_t  num_fops;
    const mgpu_fop_t *fops;
    /* config (query.h:47-68) */
    int32_t  hits_per_page, page_num;
    int32_t  max_facet_results;
    uint32_t facet_enabled;                     /* bit f */
    int32_t  rank_by;
    int32_t  rank_asc;
    int32_t  rules[MGPU_MAX_RULES];             /* q->rank_rule, MGPU_R_DONE terminated */
    /* perform_ranking's switch (docrank.c:565-587): with full_scan == 0, rank_by >= 0 and more hits than the threshold the
     * shard is ranked on a sample (perform_fast_ranking, docrank.c:478-560: zero-typo documents first, then the first
     * 1000 in num_dbi[rank_by] order, then every n-th document) and the facet counts of the top 2N are recounted exactly
     * (squery.c:398-408,439-446); mgpu_sqout_t.fast_rank reports it.  Needs facet_pid_base when facets are enabled.
     * MGPU_E_UNSUPPORTED where the reference itself misbehaves in this mode: a cardinality aggregation (NULL
     * dereference, metric-aggs.c:23-45); MGPU_E_LIMIT when the sample outgrows the reference's rank buffer
     * (docrank.c:491) or rank_by is a field with several values per document. */
    int32_t  full_scan;
    uint32_t full_scan_threshold;
    uint32_t facet_pid_base[MGPU_MAX_FIELDS];   /* pid of facetid2bmap[smallest facet id of field f]; the field's other
                                                 * ids follow in ascending id order.  0 = not given */
    /* aggs (metric-aggs.h) */
    int32_t  num_aggs;
    int32_t  agg_type[MGPU_MAX_AGGS];
    int32_t  agg_field[MGPU_MAX_AGGS];
    /* debug captures for parity tests: 1 = docid_map words, 2 = all rank records */
    int32_t  capture;
} mgpu_plan_t;

/* ---------------------------------------------------------------- results ----------------- */
typedef struct mgpu_rank {       /* struct docrank (docrank.h:5-15), same layout */
    double   compare;
    uint32_t docid;
    uint32_t distance;
    uint16_t position;
    uint16_t proximity;
    uint8_t  typos;
    uint8_t  exact;
    uint8_t  field;
    uint8_t  shard_id;
} mgpu_rank_t;

typedef struct mgpu_facet { uint32_t facet_id; uint32_t count; } mgpu_facet_t;

typedef struct mgpu_agg_out { double sum, min, max; int64_t count; uint32_t card_n; uint32_t *card_ids; } mgpu_agg_out_t;

typedef struct mgpu_sqout {      /* struct squery_result (squery.h:34-46), flat; malloc'd arrays */
    int32_t   status;
    uint32_t  num_hits;
    int32_t   rank_count;
    mgpu_rank_t *ranks;                          /* sorted by rank rules, ties by docid */
    int32_t   rcount[MGPU_MAX_FIELDS];           /* fh[f].rcount */
    mgpu_facet_t *fc[MGPU_MAX_FIELDS];           /* top 2*maxFacetResults by count, ties by facet id */
    mgpu_agg_out_t aggs[MGPU_MAX_AGGS];
    int32_t   fast_rank;                         /* sq->fast_rank: the shard was ranked on a sample */
    uint64_t *dbg_docid_map; uint32_t dbg_words;
    mgpu_rank_t *dbg_all_ranks;
} mgpu_sqout_t;

/* ---------------------------------------------------------------- execution --------------- */
/* scratch_bytes: size of the overflow position pool (0 = default 64 MiB) */
int  mgpu_ctx_create(int device, uint64_t scratch_bytes, mgpu_ctx_t **out);
void mgpu_ctx_destroy(mgpu_ctx_t *c);

/* execute_squery (squery.c:459-536) for n (shard, plan) pairs in one batch of launches; all
 * shards must live on the context's device.  outs[i] is filled (free with mgpu_sqout_free). */
int  mgpu_execute(mgpu_ctx_t *c, mgpu_shard_t *const *shards, const mgpu_plan_t *plans, int n,
                  mgpu_sqout_t *outs);
/* The same in two halves, for callers that keep several batches in flight (the threadpool of query.c:435-452
 * with a batch as the unit of work): begin lowers the plans and enqueues every copy and kernel on the context's
 * stream, then returns; end waits and fills outs.  One batch per context at a time; shards and plans must stay
 * valid until end.  mgpu_execute = begin + end. */
int  mgpu_execute_begin(mgpu_ctx_t *c, mgpu_shard_t *const *shards, const mgpu_plan_t *plans, int n);
int  mgpu_execute_end(mgpu_ctx_t *c, mgpu_sqout_t *outs);
/* sqresult_free (squery.c:336-356) */
void mgpu_sqout_free(mgpu_sqout_t *o);

/* device time of the last mgpu_execute on this context, measured with HIP events on the
 * context's stream: total ms, and per-stage ms (see mgpu_stage_name) */
#define MGPU_NUM_STAGES 8
int  mgpu_last_timing(const mgpu_ctx_t *c, float *total_ms, float *stage_ms /* [MGPU_NUM_STAGES] */);
const char *mgpu_stage_name(int stage);
/* host wall-clock ms of the last mgpu_execute: [0] plan lowering + staging, [1] launches + wait, [2] result decode */
int  mgpu_last_host_timing(const mgpu_ctx_t *c, float *host_ms /* [3] */);
/* algorithmic bytes moved by the last mgpu_execute per stage (DESIGN.md byte model) */
int  mgpu_last_bytes(const mgpu_ctx_t *c, uint64_t *stage_bytes /* [MGPU_NUM_STAGES] */);
/* enable/disable per-stage event timing (adds event records between kernels) */
void mgpu_ctx_set_profiling(mgpu_ctx_t *c, int on);

/* ---------------------------------------------------------------- primitives --------------
 * kernel-level entry points used by the parity tests against bmap.c / cont.c */
/* bmap_and / oper_or / bmap_andnot / bmap_cardinality on two postings of a mirror; op: 0 and,
 * 1 or, 2 andnot.  out_words may be NULL; returns the cardinality in *card. */
int mgpu_posting_op(mgpu_ctx_t *c, mgpu_shard_t *s, int op, uint32_t pid_a, uint32_t pid_b,
                    uint64_t *out_words, uint32_t nwords, uint64_t *card);


/* mgpu_posting_op for callers that cannot allocate.  No counterpart in the reference: it exists for callers
 * that may not call hipMalloc -- tight serving loops, harnesses that forbid it, code that runs while another stream
 * of the process is under HIP graph capture (the call itself waits for its stream and hands its results to the
 * host, so it is not something to capture; it also refuses a context with a batch in flight).
 * Same arguments, results and error codes as mgpu_posting_op; its scratch (two operand rows, the result row, the
 * cardinality word) is a static device array of 3 * MGPU_NOALLOC_MAX_DOCS / 8 + 256 bytes (6 MiB + 256 B) that
 * every GPU the library runs on holds until the process ends.  Shards with more than MGPU_NOALLOC_MAX_DOCS docids
 * are refused with MGPU_E_LIMIT.  The call allocates nothing (device, pinned or heap); the context's set-up
 * (mgpu_ctx_create) looked up the array's device address and pinned the result bounce buffer.  All calls on one
 * GPU share that GPU's array: calls that overlap on one GPU race, serialise them. */
#define MGPU_NOALLOC_MAX_DOCS (16u << 20)
int mgpu_posting_op_noalloc(mgpu_ctx_t *c, mgpu_shard_t *s, int op, uint32_t pid_a, uint32_t pid_b,
                            uint64_t *out_words, uint32_t nwords, uint64_t *card);

/* ------------------------------------------------------------------ f-1: term lookup on the device
 * Replaces the typo walk of dtrie_lookup_term (lookup_typo / node_lev, /root/reference/src/main/dtrie.c:710-791, reached
 * from dtrie.c:805-823 when the term allows typos and is not a short prefix): for every (term, dictionary) pair the
 * words within the term's edit distance and their least distances (add_wordid_to_result, dtrie.c:633-640; a prefix
 * term gives a matching node's distance to every word below it, node_walk dtrie.c:642-658).  The device holds the
 * dictionary as a flat word list, not a trie: every word is scored on its own with the walk's row arithmetic
 * (marlin_amd/csrc/mg_lev.h), one lane per word.  The twid and no-typo modes (dtrie.c:664-700) are exact descents and
 * stay with the caller.
 *
 * mgpu_dict_create: chars[word_off[i] .. word_off[i+1]) = the code points of the word whose id is word_wid[i];
 * word_off has num_words + 1 ascending entries.  Words longer than MGPU_MAX_TERM_LEN + 2 cannot match any term.
 * mgpu_lookup_terms: n lookups in one launch; *hits receives a malloc'd array of {lookup index, wid, distance} in no
 * particular order (every (lookup, wid) at most once), freed with mgpu_term_hits_free.  Terms of 0 or more than
 * MGPU_MAX_TERM_LEN characters are refused with MGPU_E_ARG. */
#define MGPU_MAX_TERM_LEN 127
typedef struct mgpu_dict mgpu_dict_t;
typedef struct mgpu_term_lookup {
    const mgpu_dict_t *dict;
    const uint32_t *chars;       /* the term's code points */
    int32_t len;
    int32_t prefix;              /* term_t.prefix: the last query word, matched as a prefix */
} mgpu_term_lookup_t;
typedef struct mgpu_term_hit { uint32_t lookup, wid, dist; } mgpu_term_hit_t;
int  mgpu_dict_create(int device, const uint32_t *chars, const uint32_t *word_off, const uint32_t *word_wid,
                      uint32_t num_words, mgpu_dict_t **out);
void mgpu_dict_destroy(mgpu_dict_t *d);
uint64_t mgpu_dict_bytes(const mgpu_dict_t *d);
int  mgpu_lookup_terms(mgpu_ctx_t *c, const mgpu_term_lookup_t *lookups, int n, mgpu_term_hit_t **hits, uint32_t *nhits);
void mgpu_term_hits_free(mgpu_term_hit_t *hits);

/* ------------------------------------------------------------------ f-4: hit materialisation, the matcher of highlight()
 * form_result ends by highlighting the stored strings of every hit of the page (hit_query_processing,
 * /root/reference/src/main/query.c:189-310 -> highlight(), highlight.c:215-345).  The compute of it is is_match
 * (highlight.c:195-213) over every (token, query term) pair: levenshtein() (highlight.c:166-192) returns how many leading
 * characters of the token match the term best, within the term's typo allowance.  mgpu_highlight_match does that for the
 * tokens of ALL strings of ALL hits of a batch of queries in one launch, one lane per token: match_len[i] = the value
 * is_match leaves in token i for the FIRST term of its job (query) that matches, or -1.  Tokenizing and mark assembly stay
 * with the caller (marlin_amd/host/mhighlight.c: m_highlight_batch).  Tokens and terms are code points (at most 127 each);
 * job j's terms are terms[job_term_off[j] .. job_term_off[j + 1]), in query order. */
typedef struct mgpu_hl_term { uint32_t off, len; int32_t max_dist, prefix; } mgpu_hl_term_t;   /* off / len into term_chars */
int mgpu_highlight_match(mgpu_ctx_t *c, const uint32_t *tok_chars, const uint32_t *tok_off /* [ntok + 1] */, const uint32_t *tok_job /* [ntok] */,
                         uint32_t ntok, const mgpu_hl_term_t *terms, uint32_t nterms, const uint32_t *term_chars, uint32_t nterm_chars,
                         const uint32_t *job_term_off /* [njobs + 1] */, uint32_t njobs, int32_t *match_len /* [ntok] */);

/* ------------------------------------------------------------------ (e): the cross-GPU exchange
 * One process per GPU owns a block of the index's shards; after a batch every process holds fixed-size result slots
 * for its (query, shard) pairs and ONE all-gather hands every slot to every process, which then merges its share of the
 * queries (form_result, /root/reference/src/main/query.c:312-383: the reference merges in shared memory after
 * wait_for_workers, query.c:449).  mgpu_comm_* is that all-gather: RCCL over xGMI, issued from C on a stream of its own.
 *
 * mgpu_comm_unique_id: 128 bytes the caller hands to every rank by whatever means it has (rank 0 makes them).
 * mgpu_comm_create: ncclCommInitRank for `rank` of `world` on `device`; `nslots` buffer sets of up to `max_bytes` per
 * rank, so that several gathers can be in flight.
 * mgpu_comm_allgather_begin: copies `bytes` from `send` (host) to the device, all-gathers, copies world * bytes back to
 * `recv` (host), all enqueued on the comm's stream; returns without waiting.  Every rank must issue its begins in the
 * same order.  mgpu_comm_allgather_end waits for that slot's copies.  `send` / `recv` should be pinned
 * (mgpu_host_alloc) so that the copies are asynchronous. */
typedef struct mgpu_comm mgpu_comm_t;
#define MGPU_COMM_ID_BYTES 128
int  mgpu_comm_unique_id(void *id /* [MGPU_COMM_ID_BYTES] */);
int  mgpu_comm_create(int device, const void *id, int rank, int world, int nslots, size_t max_bytes, mgpu_comm_t **out);
void mgpu_comm_destroy(mgpu_comm_t *c);
int  mgpu_comm_rank(const mgpu_comm_t *c);
int  mgpu_comm_world(const mgpu_comm_t *c);
int  mgpu_comm_allgather_begin(mgpu_comm_t *c, int slot, const void *send, size_t bytes, void *recv);
int  mgpu_comm_allgather_end(mgpu_comm_t *c, int slot);
/* pinned host memory for the buffers above */
void *mgpu_host_alloc(size_t bytes);
void  mgpu_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
