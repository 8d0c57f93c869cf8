/* orc.h -- CPU ORACLE for the shard-query hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's execute_squery pipeline
 * (/root/reference/src/main/squery.c:459-536 and the functions it calls), kept
 * deliberately sequential and close to the reference's control flow.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it; the
 * product path (marlin_amd/) never does.
 *
 * Pinning: the oracle is checked against the reference's own known answers
 * (test/robot/ *.robot files, test/py/pytests.py re-encoded under tests/golden/) and, in the
 * build container, against the reference's bmap.c / cont.c / sort.c / hashtable.c /
 * farmhash-c.c compiled unmodified into oracle/_ref/ (oracle/Makefile).
 *
 * Differences from the reference, on purpose:
 *   - bitmaps are dense u64 words, not roaring containers (same sets; validated against
 *     oracle/_ref bmap.c);
 *   - fast-rank mode (docrank.c:478-560) is restated as the reference runs it; where the reference itself
 *     misbehaves (a cardinality aggregation: NULL dereference; a sample larger than its rank buffer)
 *     orc_execute_squery returns -5 / -3 instead.
 */
#ifndef MARLIN_ORACLE_ORC_H
#define MARLIN_ORACLE_ORC_H

#include "../marlin_amd/host/mcommon.h"
#include "../marlin_amd/host/mshard.h"
#include "../marlin_amd/host/mquery.h"

#ifdef __cplusplus
extern "C" {
#endif

/* execute_squery (squery.c:459-536).  flags: 1 = capture docid_map words, 2 = capture every
 * rank record in ascending docid order (before sort_results), 4 = keep the reference's
 * unsorted quickselect prefix instead of sorting the returned ranks. */
int orc_execute_squery(const mshard_t *s, const m_query_t *q, int shard_idx, int flags, m_sqresult_t *out);

/* restated hashtable.c, exposed for the _ref cross-check */
typedef struct orc_cell { uint32_t key, value; } orc_cell_t;
typedef struct orc_hashtable { orc_cell_t *cells; uint32_t array_size, population; } orc_hashtable_t;
orc_hashtable_t *orc_hashtable_new(uint32_t size);
orc_cell_t *orc_hashtable_insert(orc_hashtable_t *h, uint32_t key);
orc_cell_t *orc_hashtable_lookup(orc_hashtable_t *h, uint32_t key);
void orc_hashtable_free(orc_hashtable_t *h);

/* restated sort.c */
void orc_rank_sort(size_t n, m_docrank_t *a, const int *rules);                          /* sort.c:134-183 */
void orc_rank_partial_sort(m_docrank_t *a, size_t l, size_t r, size_t m, const int *rules); /* sort.c:208-214 */
/* restated ks_partialsort(facet_sort) (ksort.h:260-285, squery.c:17-18) */
void orc_facet_partialsort(m_facet_count_t *a, size_t l, size_t r, size_t m);
/* restated ks_introsort(facetsort) (ksort.h:180-230, query.c:11-12) */
void orc_facet_introsort(size_t n, m_facet_count_t *a);

/* dense-bitmap primitives (set semantics of bmap.c / cont.c) */
void     orc_bm_or(uint64_t *dst, const uint64_t *src, size_t words);
void     orc_bm_and(uint64_t *dst, const uint64_t *src, size_t words);
void     orc_bm_andnot(uint64_t *dst, const uint64_t *src, size_t words);
uint64_t orc_bm_cardinality(const uint64_t *a, size_t words);
void     orc_bm_from_list(uint64_t *dst, const uint32_t *docs, uint64_t n);

/* the merge half of form_result (query.c:312-383, 39-92; metric-aggs.c merges), restated literally in orc_merge.c;
 * out is freed with m_result_free */
void orc_form_result(const m_query_t *q, m_sqresult_t *sq, int num_shards, m_result_t *out);

/* seconds this thread spent inside mdict_lookup_term / mdict_lookup_exact since its last reset */
double orc_lookup_time(int reset);

/* cpu_baseline at more than one core: the reference's search_pool shape (query.c:435-452, marlin.c:38-44: one
 * execute_squery job per (query, shard), a pool of threads).  nthreads workers take the nq x nshards jobs from a
 * shared counter.  Returns wall seconds; *lookup_cpu_s / *busy_cpu_s = seconds summed over the workers spent in
 * dictionary lookups / in execute_squery as a whole; *hits = sum of num_hits (a checksum). */
double orc_bench_pool(const mshard_t *const *shards, int nshards, const m_query_t *const *queries, int nq, int nthreads,
                      double *lookup_cpu_s, double *busy_cpu_s, uint64_t *hits);

#ifdef __cplusplus
}
#endif
#endif
