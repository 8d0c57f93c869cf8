/* mcommon.h -- shared host-side types for the MI355X shard-query path.
 *
 * These mirror the reference's query-side structs closely enough that the parity tests
 * read like the reference's own: struct docrank (docrank.h:5-15), facet_count_t
 * (squery.h:23-27), SORT_RULE (sort.h:7-18), FILTER_TYPE (filter.h:8-20),
 * AGG_TYPE (aggs.h:15-23), struct query_cfg (query.h:47-68), term_t (dtrie.h:84-88).
 * Plain C99, no GPU types.
 */
#ifndef MARLIN_AMD_MCOMMON_H
#define MARLIN_AMD_MCOMMON_H

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M_LEVLIMIT        3      /* dtrie.h:16 */
#define M_MAX_HITS_LIMIT  1000   /* index.h:22 */
#define M_PARTIAL_SORT_LIMIT 100 /* sort.h:5 */
#define M_MAX_WORD_LEN    127    /* analyzer token[128] */
#define M_MAX_QWORDS      16     /* limit of this implementation (reference: unbounded VLA, docrank.c:223) */
#define M_MAX_QTERMS      (2 * M_MAX_QWORDS)
#define M_MAX_FIELDS      32     /* limit of this implementation (reference MAX_FIELDS 0xFF, common.h:32) */
#define M_MAX_RULES       10     /* R_MAX + 1, query.h:61 */
#define M_MAX_AGGS        16

typedef uint32_t m_chr_t;          /* utf8proc code point, word.h */

/* docrank.h:5-15 -- identical field order and sizes (24 bytes). */
typedef struct m_docrank {
    double   compare;
    uint32_t docid;
    uint32_t distance;
    uint16_t position;
    uint16_t proximity;
    uint8_t  typos;
    uint8_t  exact;
    uint8_t  field;
    uint8_t  shard_id;
} m_docrank_t;

/* squery.h:23-27 */
typedef struct m_facet_count {
    uint8_t  shard_id;
    uint32_t facet_id;
    uint32_t count;
} m_facet_count_t;

/* sort.h:7-18 */
typedef enum {
    MR_TYPO, MR_PROX, MR_POS, MR_EXACT, MR_FIELD, MR_GEO, MR_DONE, MR_COMP, MR_COMP_ASC, MR_MAX
} m_sort_rule;

/* filter.h:8-20 */
typedef enum {
    MF_AND, MF_OR, MF_NIN, MF_EQ, MF_NE, MF_GT, MF_GTE, MF_LT, MF_LTE, MF_RANGE, MF_ERROR
} m_filter_type;

/* field types the filter leaves care about (mapping.h F_TYPE subset) */
typedef enum { MT_STRING, MT_NUMBER, MT_BOOLEAN, MT_OTHER } m_field_type;

/* aggs.h:15-23 */
typedef enum { MA_ROOT, MA_MAX, MA_MIN, MA_AVG, MA_STATS, MA_CARDINALITY, MA_ERROR } m_agg_type;

/* query.h:11-36 */
typedef enum { MPREFIX_LAST, MPREFIX_ALL, MPREFIX_NONE } m_prefix_type;
typedef enum { MTYPO_OK, MTYPO_NONE } m_typos_type;

typedef struct m_word {
    int   length;
    m_chr_t chars[M_MAX_WORD_LEN + 1];
} m_word_t;

/* dtrie.h:84-88 */
typedef struct m_term {
    m_word_t word;
    uint8_t  prefix;
    uint8_t  typos;
} m_term_t;

/* filter.h:24-38 (parsed tree; parsing itself stays in the reference's filter.c) */
typedef struct m_filter {
    int     type;            /* m_filter_type */
    int     field_type;      /* m_field_type */
    int     i_priority;      /* schema->i_priority (numbers, bools) */
    int     f_priority;      /* schema->f_priority (facets) */
    int     is_facet;
    uint32_t facet_id;       /* farmhash32(strval) -- computed where the string is known */
    double  numval;
    double  numval2;
    int     numcmp1;
    int     numcmp2;
    int     num_children;
    struct m_filter **children;
} m_filter_t;

/* One metric aggregation (metric-aggs.h:8-42), flat children of the root agg (aggs.c:56-90). */
typedef struct m_agg {
    int    type;             /* m_agg_type */
    int    field;            /* i_priority for max/min/avg/stats, f_priority for cardinality */
    /* accumulated state, same meaning as agg_max/min/avg/stats/card */
    double value;            /* max / min */
    double sum, min, max;
    int    count;
    /* cardinality: distinct facet ids seen (sorted ascending, owned) */
    uint32_t *card_ids;
    uint32_t  card_n;
} m_agg_t;

/* query.h:47-68 */
typedef struct m_query_cfg {
    int      prefix;                 /* m_prefix_type */
    int      typos;                  /* m_typos_type */
    uint16_t hits_per_page;
    uint16_t max_hits;
    uint16_t max_facet_results;
    int      num_rules;
    int      rank_by;                /* numeric field priority or -1 */
    bool     rank_sort;
    bool     rank_asc;
    bool     full_scan;
    uint32_t full_scan_threshold;
    int      rank_algo[M_MAX_RULES];
    int      facet_enabled[M_MAX_FIELDS];
} m_query_cfg_t;

/* query.h:70-84 */
typedef struct m_query {
    char       *text;
    int         num_words;
    m_word_t    words[M_MAX_QWORDS];
    int         num_terms;
    m_term_t    terms[M_MAX_QTERMS];
    m_filter_t *filter;
    int         num_aggs;            /* 0 = no "aggs" in the query */
    m_agg_t     aggs[M_MAX_AGGS];
    m_query_cfg_t cfg;
    uint16_t    page_num;
    int         rank_rule[M_MAX_RULES];
} m_query_t;

/* squery.h:34-46 -- what one shard hands back to form_result. */
typedef struct m_sqresult {
    uint32_t      num_hits;
    int           rank_count;
    m_docrank_t  *ranks;                          /* rank_count valid, sorted (ties) */
    int           num_facets;
    int           rcount[M_MAX_FIELDS];           /* fh[f].rcount */
    m_facet_count_t *fc[M_MAX_FIELDS];            /* fc[f][0..rcount) */
    int           num_aggs;
    m_agg_t       aggs[M_MAX_AGGS];
    bool          fast_rank;
    /* optional debug captures used by the parity tests */
    uint64_t     *dbg_docid_map;                  /* dense bitmap words of docid_map, or NULL */
    uint32_t      dbg_words;
    m_docrank_t  *dbg_all_ranks;                  /* num_hits records in ascending docid order, or NULL */
} m_sqresult_t;

void m_sqresult_free(m_sqresult_t *r);
void m_sqresult_free_batch(m_sqresult_t *rs, int n);

#ifdef __cplusplus
}
#endif
#endif
