/* mquery.h -- query construction and cross-shard merge, host side.
 *
 * Mirrors the reference's query objects for this path: query_new / default query_cfg
 * (index.c:1365-1378), analyze + generate_query_terms (index.c:1150-1156, query.c:489-603),
 * rank_rule assembly (index.c:1134-1148) and the merge half of form_result
 * (query.c:312-383, 39-92).  JSON parsing, hit materialisation and highlighting stay in
 * the reference (out of scope).
 */
#ifndef MARLIN_AMD_MQUERY_H
#define MARLIN_AMD_MQUERY_H

#include "mcommon.h"

#ifdef __cplusplus
extern "C" {
#endif

m_query_t *mq_new(int num_facets);                 /* defaults of query_config_new */
void       mq_free(m_query_t *q);
/* q->text = text; words via the search analyzer; terms via generate_query_terms.
 * returns 0, or -1 when the query has more than M_MAX_QWORDS words */
int        mq_set_text(m_query_t *q, const char *text);
/* call after cfg.rank_by / rank_sort / rank_asc / rank_algo are final (index.c:1134-1148) */
void       mq_build_rank_rule(m_query_t *q);

m_filter_t *mf_new(int type);
void        mf_add_child(m_filter_t *parent, m_filter_t *child);
void        mf_free(m_filter_t *f);

int  mq_add_agg(m_query_t *q, int type, int field);

/* merged result of one query over all shards (the non-JSON part of form_result) */
typedef struct m_result {
    int          total_hits;
    int          num_hits;                          /* hits on the requested page */
    int          page_s, page_e;
    int          total_ranks;
    m_docrank_t *ranks;                             /* all merged ranks, sorted; page = [page_s,page_e) */
    int          num_facets;
    int          fcount[M_MAX_FIELDS];
    m_facet_count_t *fc[M_MAX_FIELDS];
    int          num_aggs;
    m_agg_t      aggs[M_MAX_AGGS];
    bool         full_scan;
} m_result_t;

/* form_result, merge part (query.c:312-383) + process_facet_results (query.c:39-92).
 * sq[i] is shard i's result.  Ordering among equal keys is canonical here (docs of equal
 * rank by shard then docid; facets of equal count by facet id) where the reference's is
 * implementation-defined -- see DESIGN.md "ties". */
void m_form_result(const m_query_t *q, m_sqresult_t *sq, int num_shards, m_result_t *out);
void m_result_free(m_result_t *r);

/* highlight (highlight.c:215-345): `str` with the leading part of every token that matches a query term wrapped in
 * <mark> </mark>, or NULL when nothing matches; snip_num_words as in the reference (its callers pass 0).  mhighlight.c */
char *m_highlight(const char *str, const m_query_t *q, int snip_num_words);
void  m_highlight_free(char *s);
/* the same for n strings with the matcher on the device (one launch for all tokens); ctx idle, on any device */
struct mgpu_ctx;
int   m_highlight_batch(struct mgpu_ctx *ctx, const char *const *strs, const m_query_t *const *qs, const int *snip, int n, char **out);

/* __sort_lt (sort.c:46-100) */
bool m_rank_lt(const m_docrank_t *a, const m_docrank_t *b, const int *rules);
/* total order used wherever this implementation must break the reference's ties */
bool m_rank_lt_canonical(const m_docrank_t *a, const m_docrank_t *b, const int *rules);
void m_rank_sort_canonical(m_docrank_t *a, size_t n, const int *rules);

/* Result slots: fixed-size wire form of one shard's squery_result for the cross-GPU gather that
 * replaces the in-process hand-off of query.c:319-383 (k rank records, fmax facet entries per
 * facet field, num_aggs aggregation partials; cardinality id sets travel in a second, variable-length buffer:
 * m_slot_card_count / _pack on the sending side, m_slot_card_attach per sending rank after the gather). */
size_t m_slot_bytes(int k, int num_facets, int fmax, int num_aggs);
void m_slot_pack(const m_sqresult_t *r, int k, int num_facets, int fmax, int num_aggs, uint8_t *buf);
void m_slot_unpack(const uint8_t *buf, int k, int num_facets, int fmax, int num_aggs, int shard_idx, m_sqresult_t *out);
void m_slot_pack_batch(const m_sqresult_t *rs, int n, int k, int num_facets, int fmax, int num_aggs, uint8_t *buf);
uint64_t m_slot_card_count(const m_sqresult_t *rs, int n, int num_aggs);
void m_slot_card_pack(const m_sqresult_t *rs, int n, int num_aggs, uint32_t *buf);
int  m_slot_card_attach(const uint8_t *slots, const uint32_t *card, uint64_t card_n, int rank, int nranks, int nq, int local_shards,
                        int k, int num_facets, int fmax, int num_aggs, int q_first, int q_step, m_sqresult_t *sq);
/* gathered layout: [rank][query][local shard]; output sq[query * total_shards + shard] */
void m_slot_unpack_gathered(const uint8_t *buf, int nranks, int nq, int local_shards, int k, int num_facets,
                            int fmax, int num_aggs, m_sqresult_t *sq);
void m_slot_unpack_gathered_subset(const uint8_t *buf, int nranks, int nq, int local_shards, int k, int num_facets,
                                   int fmax, int num_aggs, int q_first, int q_step, m_sqresult_t *sq);
/* form_result for nq queries; returns the sum of total_hits (a checksum) */
long m_form_result_batch(const m_query_t *const *queries, m_sqresult_t *sq, int nq, int num_shards,
                         m_result_t *outs, int keep);

#ifdef __cplusplus
}
#endif
#endif
