/* orc_merge.c -- TEST INFRASTRUCTURE: the merge half of the reference's form_result, restated literally, so that the
 * merged GPU-vs-oracle comparisons check the product's m_form_result against something other than itself.
 *
 *   orc_form_result            query.c:312-383 (hit totals, paging, rank concatenation + rank_sort, agg merge)
 *   orc_process_facet_results  query.c:39-92   (single shard: introsort + cut; several: find-or-append through the
 *                                               reference's hashtable into a worst-case array, introsort, cut)
 *   agg merges                 metric-aggs.c:47-73 (cardinality: OR of the shards' id sets) and the max / min / avg /
 *                              stats merge functions of the same file
 * The sorts are the reference's own (orc_sort.c: ks_introsort over facetsort, rank_sort), so ties come out as the
 * reference leaves them; the product orders ties canonically and tests/util.py compares tie classes.
 * Nothing under marlin_amd/ may call this.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

static int cmp_u32(const void *a, const void *b) {
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

static void orc_agg_merge(m_agg_t *into, const m_agg_t *from) {
    switch (into->type) {
        case MA_MAX: if (from->value > into->value) into->value = from->value; break;       /* max_agg_merge */
        case MA_MIN: if (from->value < into->value) into->value = from->value; break;       /* min_agg_merge */
        case MA_AVG: into->sum += from->sum; into->count += from->count; break;             /* avg_agg_merge */
        case MA_STATS:                                                                       /* stats_agg_merge */
            into->sum += from->sum; into->count += from->count;
            if (from->min < into->min) into->min = from->min;
            if (from->max > into->max) into->max = from->max;
            break;
        case MA_CARDINALITY: {                         /* card_agg_merge: oper_add of the shard's bitmap, OR-ed at the end */
            const uint32_t n = into->card_n + from->card_n;
            uint32_t *u = malloc((n ? n : 1) * sizeof(uint32_t));
            uint32_t m = 0;
            for (uint32_t i = 0; i < into->card_n; i++) u[m++] = into->card_ids[i];
            for (uint32_t i = 0; i < from->card_n; i++) u[m++] = from->card_ids[i];
            qsort(u, m, sizeof(uint32_t), cmp_u32);
            uint32_t w = 0;
            for (uint32_t i = 0; i < m; i++) if (!w || u[w - 1] != u[i]) u[w++] = u[i];
            free(into->card_ids);
            into->card_ids = u; into->card_n = w;
        } break;
        default: break;
    }
}

/* query.c:39-92 */
static m_facet_count_t *orc_process_facet_results(const m_query_t *q, m_sqresult_t *sq, int num_shards, int f, int *count) {
    m_facet_count_t *fc;
    int rcount = 0;
    if (num_shards == 1) {
        rcount = sq[0].rcount[f];
        fc = malloc((rcount ? rcount : 1) * sizeof(m_facet_count_t));
        if (rcount) memcpy(fc, sq[0].fc[f], (size_t)rcount * sizeof(m_facet_count_t));
        orc_facet_introsort((size_t)rcount, fc);
        if (rcount > q->cfg.max_facet_results) rcount = q->cfg.max_facet_results;
        *count = rcount;
        return fc;
    }
    size_t cap = (size_t)num_shards * (size_t)q->cfg.max_facet_results * 2;
    fc = malloc((cap ? cap : 1) * sizeof(m_facet_count_t));
    orc_hashtable_t *h = orc_hashtable_new(512);
    for (int i = 0; i < num_shards; i++) {
        const int size = sq[i].rcount[f];
        for (int j = 0; j < size; j++) {
            const m_facet_count_t *ifc = &sq[i].fc[f][j];
            orc_cell_t *c;
            if ((c = orc_hashtable_lookup(h, ifc->facet_id)) != NULL) fc[c->value].count += ifc->count;
            else {
                c = orc_hashtable_insert(h, ifc->facet_id);
                memcpy(&fc[rcount], ifc, sizeof(m_facet_count_t));
                fc[rcount].shard_id = (uint8_t)i;          /* the shard stamps its id on its list (squery.c:425) */
                c->value = (uint32_t)rcount;
                rcount++;
            }
        }
    }
    orc_facet_introsort((size_t)rcount, fc);
    if (rcount > q->cfg.max_facet_results) rcount = q->cfg.max_facet_results;
    *count = rcount;
    orc_hashtable_free(h);
    return fc;
}

/* query.c:312-383 */
void orc_form_result(const m_query_t *q, m_sqresult_t *sq, int num_shards, m_result_t *out) {
    memset(out, 0, sizeof(*out));
    int total_hits = 0, total_ranks = 0;
    bool full_scan = true;
    for (int i = 0; i < num_shards; i++) { total_hits += (int)sq[i].num_hits; total_ranks += sq[i].rank_count; }
    int page_s = q->cfg.hits_per_page * (q->page_num - 1);
    if (page_s > total_hits) page_s = total_hits;
    int page_e = page_s + q->cfg.hits_per_page;
    if (page_e > total_hits) page_e = total_hits;
    out->total_hits = total_hits; out->page_s = page_s; out->page_e = page_e; out->num_hits = page_e - page_s;

    out->ranks = malloc((total_ranks ? total_ranks : 1) * sizeof(m_docrank_t));
    out->total_ranks = total_ranks;
    out->num_aggs = q->num_aggs;
    if (num_shards == 1) {
        memcpy(out->ranks, sq[0].ranks, (size_t)sq[0].rank_count * sizeof(m_docrank_t));
        orc_rank_sort((size_t)sq[0].rank_count, out->ranks, q->rank_rule);
        for (int j = 0; j < sq[0].rank_count; j++) out->ranks[j].shard_id = 0;
        if (sq[0].fast_rank) full_scan = false;
        for (int a = 0; a < q->num_aggs; a++) {
            out->aggs[a] = sq[0].aggs[a];
            out->aggs[a].card_ids = NULL;
            if (sq[0].aggs[a].card_n) {
                out->aggs[a].card_ids = malloc(sq[0].aggs[a].card_n * sizeof(uint32_t));
                memcpy(out->aggs[a].card_ids, sq[0].aggs[a].card_ids, sq[0].aggs[a].card_n * sizeof(uint32_t));
            }
        }
    } else {
        int pos = 0;
        for (int a = 0; a < q->num_aggs; a++) { out->aggs[a] = q->aggs[a]; out->aggs[a].card_ids = NULL; out->aggs[a].card_n = 0; }
        for (int i = 0; i < num_shards; i++) {
            for (int j = 0; j < sq[i].rank_count; j++) { out->ranks[pos] = sq[i].ranks[j]; out->ranks[pos].shard_id = (uint8_t)i; pos++; }
            if (sq[i].fast_rank) full_scan = false;
            for (int a = 0; a < q->num_aggs; a++) orc_agg_merge(&out->aggs[a], &sq[i].aggs[a]);
        }
        orc_rank_sort((size_t)total_ranks, out->ranks, q->rank_rule);
    }
    out->full_scan = full_scan;

    int nf = 0;
    for (int i = 0; i < num_shards; i++) if (sq[i].num_facets > nf) nf = sq[i].num_facets;
    out->num_facets = nf;
    for (int f = 0; f < nf; f++) {
        if (!q->cfg.facet_enabled[f]) continue;
        out->fc[f] = orc_process_facet_results(q, sq, num_shards, f, &out->fcount[f]);
    }
}
