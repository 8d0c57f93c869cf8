/* mplan.h -- host side of the GPU shard query: mirror export and plan lowering.
 *
 * This is the part of execute_squery that stays on the CPU (squery.c:144-167 lookup_terms:
 * exact trie descents, twid -> wid expansion, wid -> distance maps; filter-tree leaves resolved to
 * posting ids / key ranges; the typo walks go to the device with msq_prepare_batch_dev) and hands a flat
 * mgpu_plan_t to libmarlin_gpu.so.  It never
 * touches the oracle; if the GPU library reports an error the error is returned.
 */
#ifndef MARLIN_AMD_MPLAN_H
#define MARLIN_AMD_MPLAN_H

#include "mcommon.h"
#include "mshard.h"
#include "mquery.h"
#include "marlin_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one shard's HBM mirror plus the posting-id scheme used to address it */
typedef struct mmirror {
    const mshard_t *s;
    mgpu_shard_t   *g;
    int       device;
    uint32_t  W, T;                      /* word_count, tword_count */
    uint32_t  fbase[M_MAX_FIELDS];       /* pid of field f's first facet posting */
    uint32_t  bbase;                     /* pid of bool field 0 / false */
    uint32_t  num_postings;
    mgpu_dict_t *gd;                     /* the shard's dictionary on the device (term lookup, row f-1); NULL if s->dict is NULL */
} mmirror_t;

/* pid 0 = all docs, 1..W = wid2bmap, W+1..W+T = twid2bmap, then facetid2bmap, then boolid2bmap */
int  mmirror_create(const mshard_t *s, int device, mmirror_t **out);
void mmirror_destroy(mmirror_t *m);
/* row f-2: `snew` = the mirror's shard after a committed append-only bulk write; the mirror takes the difference as a
 * delta (mgpu_shard_apply_delta) instead of being rebuilt.  No batch on this mirror may be in flight.  MGPU_E_LIMIT: no
 * headroom left, nothing changed: mmirror_create a new one.  MGPU_E_UNSUPPORTED: not an append. */
int  mmirror_apply_shard(mmirror_t *m, const mshard_t *snew);
/* the posting-id scheme alone, no device memory (g and gd stay NULL): enough to build plans, e.g. in CPU-only tests */
mmirror_t *mmirror_scheme(const mshard_t *s);

/* bump allocator behind the pointers of a plan (or of a whole batch of plans): one reset per batch, no malloc per term */
typedef struct m_arena m_arena_t;
m_arena_t *m_arena_new(void);
void m_arena_reset(m_arena_t *a);
void m_arena_free(m_arena_t *a);

typedef struct mplan {
    mgpu_plan_t p;
    mgpu_wid_entry_t *wids;              /* = p.wids: all_wordids (dist != 0xFF) + exact-only entries */
    m_arena_t *arena;                    /* owned storage behind the pointers in p */
} mplan_t;

/* lookup_terms + process_termresult + filter leaf resolution for one shard.  Whether the squery is then ranked on a
 * sample (fast-rank, docrank.c:571) is decided on the device from the hit count; the plan carries full_scan and
 * full_scan_threshold for it. */
int  mplan_build(const mmirror_t *m, const m_query_t *q, int capture, mplan_t *out);
/* the same after the trie walk: tr[t] = dtrie_lookup_term's result per term, exact_wid[i] = dtrie_lookup_exact's result
 * per query word; q->terms / q->words are not read, tr is not modified */
int  mplan_from_terms(const mmirror_t *m, const m_query_t *q, m_termresult_t *tr, const uint32_t *exact_wid,
                      int capture, mplan_t *out);
void mplan_free(mplan_t *pl);

/* execute_squery for one (shard, query) pair with a plan built by the caller */
int  msq_execute_plan(mgpu_ctx_t *ctx, const mmirror_t *m, const m_query_t *q, const mplan_t *pl, int shard_idx, m_sqresult_t *out);

/* execute_squery for nq queries x num_shards shards in one GPU batch.
 * mirrors[s] for s < num_shards must live on ctx's device.
 * outs[qi * num_shards + s] receives shard s's result of query qi. */
int  msq_execute_batch(mgpu_ctx_t *ctx, mmirror_t *const *mirrors, int num_shards,
                       const m_query_t *const *queries, int nq, int capture, m_sqresult_t *outs);

/* The same in two halves, so a caller can keep the term lookups (dtrie walk; not part of the
 * GPU hot path) apart from the batch execution: prepare = lookup_terms + filter leaf
 * resolution for every (query, shard); execute_prepared = everything after it. */
typedef struct msq_prepared msq_prepared_t;
int  msq_prepare_batch(mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries,
                       int nq, int capture, msq_prepared_t **out);
int  msq_execute_plans(mgpu_ctx_t *ctx, const mmirror_t *const *ms, const m_query_t *const *qs, const mplan_t *const *pls,
                       const int *shard_idx, int n, m_sqresult_t *outs);
int  msq_execute_prepared(mgpu_ctx_t *ctx, const msq_prepared_t *pp, m_sqresult_t *outs);
/* execute_prepared in two halves (one batch in flight per context): submit enqueues the GPU work and returns,
 * collect waits for it and decodes; pp must stay alive in between */
int  msq_submit_prepared(mgpu_ctx_t *ctx, const msq_prepared_t *pp);
int  msq_collect_prepared(mgpu_ctx_t *ctx, const msq_prepared_t *pp, m_sqresult_t *outs);
/* msq_prepare_batch with the typo walks of lookup_terms run on the device: ONE mgpu_lookup_terms call scores every
 * (query, shard, term) that dtrie_lookup_term would walk the trie for (dtrie.c:763-791); the twid and no-typo modes,
 * exact word ids and the rest of the plan stay on the host.  ctx must be idle (no batch in flight) and on the mirrors'
 * device. */
int  msq_prepare_batch_dev(mgpu_ctx_t *ctx, mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries,
                           int nq, int capture, msq_prepared_t **out);
void msq_prepared_free(msq_prepared_t *pp);
/* The two halves of a prepare, apart.  resolve = the DICTIONARY half of lookup_terms for nq queries x num_shards shards
 * (dtrie_lookup_term per term, dtrie_lookup_exact per word; ctx != NULL runs the typo walks on the device), kept as flat
 * {twid, (wid, distance) list} records: the termresult_t of dtrie.h:94-97.  plan = everything after it
 * (process_termresult squery.c:72-118: posting selection, all_wordids, zero-typo sets; exact wids squery.c:159-166; filter
 * leaves; config), written into a reusable prepared batch whose storage is one arena.  The resolved set must outlive
 * the prepared batch built from it. */
typedef struct msq_resolved msq_resolved_t;
int  msq_resolve_batch(mgpu_ctx_t *ctx, mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries, int nq,
                       msq_resolved_t **out);
void msq_resolved_free(msq_resolved_t *r);
int  msq_resolved_count(const msq_resolved_t *r);
msq_prepared_t *msq_prepared_new(void);
/* inspection (tests): plan i of a prepared batch; term t of pair i of a resolved set -> its (wid, distance) list, *twid;
 * dtrie_lookup_exact's result for word w of pair i */
const mgpu_plan_t *msq_prepared_plan(const msq_prepared_t *pp, int i);
uint32_t msq_resolved_term(const msq_resolved_t *r, int i, int t, uint32_t *twid, const uint32_t **wid, const uint8_t **dist);
uint32_t msq_resolved_exact(const msq_resolved_t *r, int i, int w);
int  msq_plan_resolved(const msq_resolved_t *r, int capture, msq_prepared_t *pp);
/* serving loop: nctx contexts = nctx batches in flight, two contexts per host thread (a thread submits its next
 * batch before it decodes and merges the current one); step i runs batch (first + i) % nprep through
 * submit / collect + form_result (mplan.c) */
int  msq_run_pipeline(mgpu_ctx_t *const *ctxs, int nctx, const msq_prepared_t *const *preps, int nprep,
                      int first, int count, long *hits_out);
/* the same loop with its threads kept between runs (created once, parked on a condition variable):
 * create -> warm (every batch through every context once: first-use allocations) -> run ... -> destroy.
 * msq_pipeline_error: message of the first failed step of the last run (worker threads' own error text). */
typedef struct msq_pipeline msq_pipeline_t;
int  msq_pipeline_create(mgpu_ctx_t *const *ctxs, int nctx, msq_pipeline_t **out);
int  msq_pipeline_warm(msq_pipeline_t *p, const msq_prepared_t *const *preps, int nprep);
int  msq_pipeline_run(msq_pipeline_t *p, const msq_prepared_t *const *preps, int nprep, int first, int count, long *hits_out);
/* the serving loop with the plans built INSIDE it: step i = msq_plan_resolved(res[(first + i) % nres]) + submit /
 * collect + form_result, all in the pipeline's own threads */
int  msq_pipeline_warm_resolved(msq_pipeline_t *p, const msq_resolved_t *const *res, int nres);
int  msq_pipeline_run_resolved(msq_pipeline_t *p, const msq_resolved_t *const *res, int nres, int first, int count, long *hits_out);
double msq_pipeline_plan_ms(const msq_pipeline_t *p);
/* ... and with the dictionary lookups inside it too: step i = lookup_terms for `batch` queries x num_shards shards (typo
 * walks on the step's context: mgpu_lookup_terms; exact descents on the host) + plans + submit / collect + form_result.
 * warm != 0: every context takes every batch once (first / count ignored). */
int  msq_pipeline_run_queries(msq_pipeline_t *p, mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries, int nq,
                              int batch, int first, int count, int warm, long *hits_out);
double msq_pipeline_lookup_ms(const msq_pipeline_t *p);
/* several GPUs: every batch of the loop ends with ONE RCCL all-gather of its result slots, issued from the pipeline's
 * threads in step order; the rank merges queries rank, rank + world, ... of each batch (mplan.c) */
int  msq_pipeline_set_gather(msq_pipeline_t *p, mgpu_comm_t *comm, int pairs, int k, int num_facets, int fmax, int num_aggs);
const char *msq_pipeline_error(const msq_pipeline_t *p);
void msq_pipeline_destroy(msq_pipeline_t *p);

/* host threads used for plan building inside msq_execute_batch (default 1) */
void msq_set_plan_threads(int n);
mgpu_shard_t *mmirror_gpu(mmirror_t *m);

#ifdef __cplusplus
}
#endif
#endif
