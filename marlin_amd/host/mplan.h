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

typedef struct mplan {
    mgpu_plan_t p;
    /* owned storage behind the pointers in p */
    uint32_t *term_post[MGPU_MAX_TERMS];
    mgpu_wid_entry_t *wids;
    mgpu_fop_t *fops;
    int fops_cap;
} mplan_t;

/* lookup_terms + process_termresult + filter leaf resolution for one shard.  Whether the squery is then ranked on a
 * sample (fast-rank, docrank.c:571) is decided on the device from the hit count; the plan carries full_scan and
 * full_scan_threshold for it. */
int  mplan_build(const mmirror_t *m, const m_query_t *q, int capture, mplan_t *out);
/* the same after the trie walk: tr[t] = dtrie_lookup_term's result per term (wordids get the words under a twid
 * added), exact_wid[i] = dtrie_lookup_exact's result per query word; q->terms / q->words are not read */
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
const char *msq_pipeline_error(const msq_pipeline_t *p);
void msq_pipeline_destroy(msq_pipeline_t *p);

/* host threads used for plan building inside msq_execute_batch (default 1) */
void msq_set_plan_threads(int n);
mgpu_shard_t *mmirror_gpu(mmirror_t *m);

#ifdef __cplusplus
}
#endif
#endif
