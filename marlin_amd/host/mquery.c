/* mquery.c -- see mquery.h */
#include "mquery.h"
#include "mshard.h"
#include "marlin_gpu.h"
#include <stdlib.h>
#include <string.h>
#include <float.h>

void m_sqresult_free(m_sqresult_t *r) {
    if (!r) return;
    free(r->ranks);
    for (int f = 0; f < M_MAX_FIELDS; f++) free(r->fc[f]);
    for (int i = 0; i < r->num_aggs; i++) free(r->aggs[i].card_ids);
    free(r->dbg_docid_map);
    free(r->dbg_all_ranks);
    memset(r, 0, sizeof(*r));
}

/* query_config_new (index.c:1365-1378) + set_default_facets (index.c:74-78) */
m_query_t *mq_new(int num_facets) {
    m_query_t *q = calloc(1, sizeof(*q));
    q->cfg.hits_per_page = 20;            /* DEF_HITS_PER_PAGE, index.h:19-24 */
    q->cfg.max_facet_results = 10;
    q->cfg.full_scan_threshold = 25000;
    q->cfg.max_hits = 500;
    q->cfg.rank_by = -1;
    q->cfg.rank_sort = false;
    q->cfg.rank_asc = false;
    q->cfg.num_rules = 5;                 /* default_rule, sort.c:8-9 */
    q->cfg.rank_algo[0] = MR_TYPO; q->cfg.rank_algo[1] = MR_PROX; q->cfg.rank_algo[2] = MR_FIELD;
    q->cfg.rank_algo[3] = MR_POS;  q->cfg.rank_algo[4] = MR_EXACT;
    for (int i = 0; i < num_facets && i < M_MAX_FIELDS; i++) q->cfg.facet_enabled[i] = 1;
    q->page_num = 1;
    mq_build_rank_rule(q);
    return q;
}

void mf_free(m_filter_t *f) {
    if (!f) return;
    for (int i = 0; i < f->num_children; i++) mf_free(f->children[i]);
    free(f->children);
    free(f);
}

void mq_free(m_query_t *q) {
    if (!q) return;
    free(q->text);
    mf_free(q->filter);
    for (int i = 0; i < q->num_aggs; i++) free(q->aggs[i].card_ids);
    free(q);
}

m_filter_t *mf_new(int type) {
    m_filter_t *f = calloc(1, sizeof(*f));
    f->type = type;
    return f;
}

void mf_add_child(m_filter_t *p, m_filter_t *c) {
    p->children = realloc(p->children, sizeof(m_filter_t *) * (size_t)(p->num_children + 1));
    p->children[p->num_children++] = c;
}

int mq_add_agg(m_query_t *q, int type, int field) {
    if (q->num_aggs >= M_MAX_AGGS) return -1;
    m_agg_t *a = &q->aggs[q->num_aggs++];
    memset(a, 0, sizeof(*a));
    a->type = type; a->field = field;
    /* initial states of parse_*_agg (metric-aggs.c:191-218, 312-338, 373-399) */
    if (type == MA_MAX) a->value = -DBL_MAX;
    if (type == MA_MIN) a->value = DBL_MAX;
    if (type == MA_STATS) { a->min = DBL_MAX; a->max = -DBL_MAX; }
    return 0;
}

/* index.c:1134-1148 */
void mq_build_rank_rule(m_query_t *q) {
    int rcount = 0;
    if (q->cfg.rank_by >= 0) {
        if (q->cfg.rank_sort) {
            q->rank_rule[0] = q->cfg.rank_asc ? MR_COMP_ASC : MR_COMP;
            q->rank_rule[q->cfg.num_rules + 1] = MR_DONE;
            rcount++;
        } else {
            q->rank_rule[q->cfg.num_rules] = q->cfg.rank_asc ? MR_COMP_ASC : MR_COMP;
            q->rank_rule[q->cfg.num_rules + 1] = MR_DONE;
        }
    } else {
        q->rank_rule[q->cfg.num_rules] = MR_DONE;
    }
    memcpy(&q->rank_rule[rcount], q->cfg.rank_algo, (size_t)q->cfg.num_rules * sizeof(int));
}

static void word_cb(const m_chr_t *chars, int length, int position, void *data) {
    (void)position;
    m_query_t *q = data;                                   /* query_string_word_cb, index.c:909-917 */
    if (q->num_words >= M_MAX_QWORDS) { q->num_words++; return; }
    m_word_t *w = &q->words[q->num_words++];
    w->length = length;
    memcpy(w->chars, chars, sizeof(m_chr_t) * (size_t)length);
}

static inline void word_cat(m_word_t *dst, const m_word_t *src) {     /* wordcat, word.c */
    int n = src->length;
    if (dst->length + n > M_MAX_WORD_LEN) n = M_MAX_WORD_LEN - dst->length;
    memcpy(dst->chars + dst->length, src->chars, sizeof(m_chr_t) * (size_t)n);
    dst->length += n;
}

/* generate_query_terms (query.c:489-603) */
static void generate_query_terms(m_query_t *q) {
    q->num_terms = 0;
    int nw = q->num_words;
    if (nw == 0) return;
    size_t tl = q->text ? strlen(q->text) : 0;
    int ends_space = tl > 0 && q->text[tl - 1] == ' ';
    if (nw == 1) {
        m_term_t *t = &q->terms[q->num_terms++];
        memset(t, 0, sizeof(*t));
        t->word = q->words[0];
        if (q->cfg.prefix != MPREFIX_NONE) t->prefix = 1;
        if (q->cfg.typos == MTYPO_OK && t->word.length > M_LEVLIMIT) t->typos = 1;
        if (ends_space) t->prefix = 0;
        return;
    }
    if (nw == 2) {
        for (int i = 0; i < 2; i++) {
            m_term_t *t = &q->terms[q->num_terms++];
            memset(t, 0, sizeof(*t));
            t->word = q->words[i];
            if (q->cfg.prefix == MPREFIX_ALL) t->prefix = 1;
            else if (q->cfg.prefix == MPREFIX_LAST && i == 1) t->prefix = ends_space ? 0 : 1;
            if (q->cfg.typos == MTYPO_OK && t->word.length > M_LEVLIMIT) t->typos = 1;
        }
        m_term_t *t = &q->terms[q->num_terms++];
        memset(t, 0, sizeof(*t));
        t->word = q->words[0];
        word_cat(&t->word, &q->words[1]);
        t->typos = 0;
        if (q->cfg.prefix != MPREFIX_NONE) t->prefix = 1;
        return;
    }
    m_term_t ct;
    memset(&ct, 0, sizeof(ct));
    if (q->cfg.prefix != MPREFIX_NONE) ct.prefix = 1;
    for (int i = 0; i < nw; i++) {
        word_cat(&ct.word, &q->words[i]);
        m_term_t *t = &q->terms[q->num_terms++];
        memset(t, 0, sizeof(*t));
        t->word = q->words[i];
        if (q->cfg.typos == MTYPO_OK && t->word.length > M_LEVLIMIT) t->typos = 1;
        if (q->cfg.prefix == MPREFIX_ALL) t->prefix = 1;
        else if (q->cfg.prefix == MPREFIX_LAST && i == nw - 1) t->prefix = ends_space ? 0 : 1;
        if (i != nw - 1) {
            m_term_t *tt = &q->terms[q->num_terms++];
            memset(tt, 0, sizeof(*tt));
            tt->word = q->words[i];
            word_cat(&tt->word, &q->words[i + 1]);
            if (q->cfg.prefix != MPREFIX_NONE && i + 2 == nw) tt->prefix = 1;
        }
    }
    q->terms[q->num_terms++] = ct;
}

int mq_set_text(m_query_t *q, const char *text) {
    free(q->text);
    q->text = text ? strdup(text) : NULL;
    q->num_words = 0;
    q->num_terms = 0;
    if (text) m_analyze_for_search(text, word_cb, q);
    if (q->num_words > M_MAX_QWORDS) { q->num_words = 0; return -1; }
    generate_query_terms(q);
    return 0;
}

#define M_MAX_SHARDS_MERGE 64           /* run-merge fast path of m_form_result; more shards use the full sort */
#define FACET_FAST_CAP 256
#define FACET_FAST_TAB_LOG 10
#define FACET_FAST_TAB (1 << FACET_FAST_TAB_LOG)

/* ------------------------------------------------------------------ comparators */
bool m_rank_lt(const m_docrank_t *a, const m_docrank_t *b, const int *rules) {
    for (int c = 0;; c++) {
        switch (rules[c]) {
            case MR_TYPO:  if (a->typos != b->typos) return a->typos < b->typos; break;
            case MR_PROX:  if (a->proximity != b->proximity) return a->proximity < b->proximity; break;
            case MR_POS:   if (a->position != b->position) return a->position < b->position; break;
            case MR_EXACT: if (a->exact != b->exact) return a->exact > b->exact; break;
            case MR_COMP:  if (a->compare != b->compare) return a->compare > b->compare; break;
            case MR_COMP_ASC: if (a->compare != b->compare) return a->compare < b->compare; break;
            case MR_FIELD: if (a->field != b->field) return a->field < b->field; break;
            case MR_GEO:   if (a->distance != b->distance) return a->distance < b->distance; break;
            default: return false;
        }
    }
}

bool m_rank_lt_canonical(const m_docrank_t *a, const m_docrank_t *b, const int *rules) {
    if (m_rank_lt(a, b, rules)) return true;
    if (m_rank_lt(b, a, rules)) return false;
    if (a->shard_id != b->shard_id) return a->shard_id < b->shard_id;
    return a->docid < b->docid;
}

/* three-way form of m_rank_lt_canonical in one walk over the rules: < 0 when a sorts first */
static inline int rank_cmp_canonical(const m_docrank_t *a, const m_docrank_t *b, const int *rules) {
    for (int c = 0;; c++) {
        switch (rules[c]) {
            case MR_TYPO:  if (a->typos != b->typos) return a->typos < b->typos ? -1 : 1; break;
            case MR_PROX:  if (a->proximity != b->proximity) return a->proximity < b->proximity ? -1 : 1; break;
            case MR_POS:   if (a->position != b->position) return a->position < b->position ? -1 : 1; break;
            case MR_EXACT: if (a->exact != b->exact) return a->exact > b->exact ? -1 : 1; break;
            case MR_COMP:  if (a->compare != b->compare) return a->compare > b->compare ? -1 : 1; break;
            case MR_COMP_ASC: if (a->compare != b->compare) return a->compare < b->compare ? -1 : 1; break;
            case MR_FIELD: if (a->field != b->field) return a->field < b->field ? -1 : 1; break;
            case MR_GEO:   if (a->distance != b->distance) return a->distance < b->distance ? -1 : 1; break;
            default:
                if (a->shard_id != b->shard_id) return a->shard_id < b->shard_id ? -1 : 1;
                return a->docid < b->docid ? -1 : (a->docid > b->docid);
        }
    }
}

/* merge sort on the canonical order that starts from the runs the caller names (run_end[r] = one past
 * run r, each run already in canonical order: a shard's own top-2N list), so that merging S shards
 * costs log2(S) passes; with nruns = 0 it is a plain bottom-up merge sort. */
static void rank_merge_runs(m_docrank_t *a, size_t n, const int *rules, const size_t *run_end, int nruns) {
    if (n < 2) return;
    m_docrank_t stack_tmp[256];
    size_t bounds_a[2 * M_MAX_SHARDS_MERGE + 2], bounds_b[2 * M_MAX_SHARDS_MERGE + 2];
    m_docrank_t *tmp = n <= 256 ? stack_tmp : malloc(n * sizeof(m_docrank_t));
    if (nruns <= 0 || nruns > 2 * M_MAX_SHARDS_MERGE) { if (tmp != stack_tmp) free(tmp); m_rank_sort_canonical(a, n, rules); return; }
    size_t *cur = bounds_a, *nxt = bounds_b;
    for (int r = 0; r < nruns; r++) cur[r] = run_end[r];
    m_docrank_t *src = a, *dst = tmp;
    while (nruns > 1) {
        int o = 0; size_t lo = 0;
        for (int r = 0; r < nruns; r += 2) {
            const size_t mid = cur[r], hi = (r + 1 < nruns) ? cur[r + 1] : cur[r];
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = rank_cmp_canonical(&src[j], &src[i], rules) < 0 ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
            nxt[o++] = hi; lo = hi;
        }
        nruns = o;
        m_docrank_t *t = src; src = dst; dst = t;
        size_t *tb = cur; cur = nxt; nxt = tb;
    }
    if (src != a) memcpy(a, src, n * sizeof(m_docrank_t));
    if (tmp != stack_tmp) free(tmp);
}

/* bottom-up merge sort on the canonical order */
void m_rank_sort_canonical(m_docrank_t *a, size_t n, const int *rules) {
    if (n < 2) return;
    m_docrank_t *tmp = malloc(n * sizeof(m_docrank_t));
    m_docrank_t *src = a, *dst = tmp;
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = rank_cmp_canonical(&src[j], &src[i], rules) < 0 ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        m_docrank_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(m_docrank_t));
    free(tmp);
}

static int cmp_fc(const void *pa, const void *pb) {
    const m_facet_count_t *a = pa, *b = pb;
    if (a->count != b->count) return a->count > b->count ? -1 : 1;
    if (a->facet_id != b->facet_id) return a->facet_id < b->facet_id ? -1 : 1;
    return 0;
}

static int cmp_fc_id(const void *pa, const void *pb) {
    const m_facet_count_t *a = pa, *b = pb;
    if (a->facet_id != b->facet_id) return a->facet_id < b->facet_id ? -1 : 1;
    return (int)a->shard_id - (int)b->shard_id;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y);
}

/* *_agg_merge (metric-aggs.c:66-73, 178-189, 245-250, 304-310, 365-371) */
static void agg_merge(m_agg_t *into, const m_agg_t *from) {
    switch (into->type) {
        case MA_MAX: if (from->value > into->value) into->value = from->value; break;
        case MA_MIN: if (from->value < into->value) into->value = from->value; break;
        case MA_AVG: into->sum += from->sum; into->count += from->count; break;
        case MA_STATS:
            into->sum += from->sum; into->count += from->count;
            if (from->min < into->min) into->min = from->min;
            if (from->max > into->max) into->max = from->max;
            break;
        case MA_CARDINALITY: {
            /* oper_or of the per-shard facet-id bitmaps == sorted set union */
            uint32_t n = into->card_n + from->card_n;
            uint32_t *u = malloc((n ? n : 1) * sizeof(uint32_t));
            if (into->card_n) memcpy(u, into->card_ids, into->card_n * sizeof(uint32_t));
            if (from->card_n) memcpy(u + into->card_n, from->card_ids, from->card_n * sizeof(uint32_t));
            qsort(u, n, sizeof(uint32_t), cmp_u32);
            uint32_t m = 0;
            for (uint32_t i = 0; i < n; i++) if (i == 0 || u[i] != u[i - 1]) u[m++] = u[i];
            free(into->card_ids);
            into->card_ids = u; into->card_n = m;
        } break;
        default: break;
    }
}

void m_form_result(const m_query_t *q, m_sqresult_t *sq, int num_shards, m_result_t *out) {
    memset(out, 0, sizeof(*out));
    out->full_scan = true;
    int total_hits = 0, total_ranks = 0;
    for (int i = 0; i < num_shards; i++) {
        total_hits += (int)sq[i].num_hits;
        total_ranks += sq[i].rank_count;
        if (sq[i].fast_rank) out->full_scan = false;
    }
    int page_s = q->cfg.hits_per_page * (q->page_num - 1);
    if (page_s > total_hits) page_s = total_hits;
    int page_e = page_s + q->cfg.hits_per_page;
    if (page_e > total_hits) page_e = total_hits;
    out->total_hits = total_hits;
    out->page_s = page_s; out->page_e = page_e;
    out->num_hits = page_e - page_s;

    /* ranks: concat with shard_id stamped, then sort (query.c:343-378) */
    out->ranks = malloc((total_ranks ? total_ranks : 1) * sizeof(m_docrank_t));
    int pos = 0;
    /* each shard's list arrives in canonical order (checked while copying): the sort is then a merge
     * of num_shards runs; a list that is not sorted falls back to the full sort */
    size_t run_end[M_MAX_SHARDS_MERGE];
    int nruns = 0; bool runs_ok = num_shards <= M_MAX_SHARDS_MERGE;
    for (int i = 0; i < num_shards; i++) {
        for (int j = 0; j < sq[i].rank_count; j++) {
            out->ranks[pos] = sq[i].ranks[j];
            out->ranks[pos].shard_id = (uint8_t)i;
            if (runs_ok && j && rank_cmp_canonical(&out->ranks[pos - 1], &out->ranks[pos], q->rank_rule) > 0) runs_ok = false;
            pos++;
        }
        if (runs_ok && sq[i].rank_count) run_end[nruns++] = (size_t)pos;
    }
    out->total_ranks = total_ranks;
    if (runs_ok) rank_merge_runs(out->ranks, (size_t)total_ranks, q->rank_rule, run_end, nruns);
    else m_rank_sort_canonical(out->ranks, (size_t)total_ranks, q->rank_rule);

    /* aggs: merged in shard order into the query's fresh aggs (query.c:373-375) */
    out->num_aggs = q->num_aggs;
    for (int a = 0; a < q->num_aggs; a++) {
        if (num_shards == 1) {
            out->aggs[a] = sq[0].aggs[a];
            if (sq[0].aggs[a].card_n) {
                out->aggs[a].card_ids = malloc(sq[0].aggs[a].card_n * sizeof(uint32_t));
                memcpy(out->aggs[a].card_ids, sq[0].aggs[a].card_ids, sq[0].aggs[a].card_n * sizeof(uint32_t));
            } else out->aggs[a].card_ids = NULL;
        } else {
            out->aggs[a] = q->aggs[a];
            out->aggs[a].card_ids = NULL; out->aggs[a].card_n = 0;
            for (int i = 0; i < num_shards; i++) agg_merge(&out->aggs[a], &sq[i].aggs[a]);
        }
    }

    /* facets (query.c:39-92): sum each shard's top-2N lists by facet id, sort, cut to N */
    int nf = 0;
    for (int i = 0; i < num_shards; i++) if (sq[i].num_facets > nf) nf = sq[i].num_facets;
    out->num_facets = nf;
    for (int f = 0; f < nf; f++) {
        if (!q->cfg.facet_enabled[f]) continue;
        int cap = 0;
        for (int i = 0; i < num_shards; i++) cap += sq[i].rcount[f];
        m_facet_count_t *fc = malloc((cap ? cap : 1) * sizeof(m_facet_count_t));
        int rcount = 0;
        if (cap <= FACET_FAST_CAP) {
            /* the common size (num_shards x 2N entries): find-or-append through a small open-addressing
             * table (shards in order, so an id keeps its lowest shard), then a bounded insertion keeps the
             * best max_facet_results by (count desc, id asc) */
            int16_t tab[FACET_FAST_TAB];
            memset(tab, 0xFF, sizeof(tab));
            for (int i = 0; i < num_shards; i++)
                for (int j = 0; j < sq[i].rcount[f]; j++) {
                    const m_facet_count_t *e = &sq[i].fc[f][j];
                    uint32_t h = (e->facet_id * 2654435761u) >> (32 - FACET_FAST_TAB_LOG);
                    while (tab[h] >= 0 && fc[tab[h]].facet_id != e->facet_id) h = (h + 1) & (FACET_FAST_TAB - 1);
                    if (tab[h] >= 0) fc[tab[h]].count += e->count;
                    else { tab[h] = (int16_t)rcount; fc[rcount] = *e; fc[rcount].shard_id = (uint8_t)i; rcount++; }
                }
            const int want = q->cfg.max_facet_results < rcount ? q->cfg.max_facet_results : rcount;
            if (want > 0 && want <= 64) {
                m_facet_count_t best[64];
                int nb = 0;
                for (int k = 0; k < rcount; k++) {
                    const m_facet_count_t e = fc[k];
                    if (nb == want) {
                        const m_facet_count_t *w = &best[nb - 1];
                        if (!(e.count > w->count || (e.count == w->count && e.facet_id < w->facet_id))) continue;
                    }
                    int at = nb < want ? nb++ : nb - 1;
                    while (at > 0 && (e.count > best[at - 1].count ||
                                      (e.count == best[at - 1].count && e.facet_id < best[at - 1].facet_id))) { best[at] = best[at - 1]; at--; }
                    best[at] = e;
                }
                memcpy(fc, best, (size_t)nb * sizeof(m_facet_count_t));
                rcount = nb;
            } else {
                qsort(fc, (size_t)rcount, sizeof(m_facet_count_t), cmp_fc);
                if (rcount > q->cfg.max_facet_results) rcount = q->cfg.max_facet_results;
            }
            out->fc[f] = fc;
            out->fcount[f] = rcount;
            continue;
        }
        /* same sums as the reference's find-or-append walk, without its quadratic search:
         * order by (facet id, shard), fold equal ids into the first (= lowest shard) entry */
        for (int i = 0; i < num_shards; i++)
            for (int j = 0; j < sq[i].rcount[f]; j++) { fc[rcount] = sq[i].fc[f][j]; fc[rcount].shard_id = (uint8_t)i; rcount++; }
        qsort(fc, (size_t)rcount, sizeof(m_facet_count_t), cmp_fc_id);
        int w = 0;
        for (int k = 0; k < rcount; k++) {
            if (w && fc[w - 1].facet_id == fc[k].facet_id) fc[w - 1].count += fc[k].count;
            else fc[w++] = fc[k];
        }
        rcount = w;
        qsort(fc, (size_t)rcount, sizeof(m_facet_count_t), cmp_fc);
        if (rcount > q->cfg.max_facet_results) rcount = q->cfg.max_facet_results;
        out->fc[f] = fc;
        out->fcount[f] = rcount;
    }
}

void m_result_free(m_result_t *r) {
    free(r->ranks);
    for (int f = 0; f < M_MAX_FIELDS; f++) free(r->fc[f]);
    for (int a = 0; a < r->num_aggs; a++) free(r->aggs[a].card_ids);
    memset(r, 0, sizeof(*r));
}

/* struct sizes, so language bindings can verify their mirrors */
size_t m_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(m_docrank_t);
        case 1: return sizeof(m_facet_count_t);
        case 2: return sizeof(m_agg_t);
        case 3: return sizeof(m_query_cfg_t);
        case 4: return sizeof(m_query_t);
        case 5: return sizeof(m_sqresult_t);
        case 6: return sizeof(m_result_t);
        case 7: return sizeof(m_filter_t);
        case 8: return sizeof(m_word_t);
        case 9: return sizeof(m_term_t);
        case 10: return sizeof(mshard_t);
        case 11: return sizeof(mgpu_plan_t);
        default: return 0;
    }
}

/* ------------------------------------------------------------------ result slots
 * Fixed-size wire form of one shard's squery_result, for the cross-GPU gather that replaces
 * the shared-memory hand-off of query.c:319-383.  Cardinality id sets are variable-length: a slot carries only their
 * sizes (m_agg_t.card_n), the ids travel in a second buffer (m_slot_card_*), concatenated in (result, agg) order, and
 * are attached again after the gather. */
typedef struct m_slot_hdr {
    uint32_t num_hits; int32_t rank_count; int32_t num_facets; int32_t num_aggs;
    int32_t rcount[M_MAX_FIELDS];
} m_slot_hdr_t;

size_t m_slot_bytes(int k, int num_facets, int fmax, int num_aggs) {
    size_t b = sizeof(m_slot_hdr_t) + (size_t)k * sizeof(m_docrank_t) +
               (size_t)num_facets * (size_t)fmax * sizeof(m_facet_count_t) + (size_t)num_aggs * sizeof(m_agg_t);
    return (b + 15) & ~(size_t)15;
}

void m_slot_pack(const m_sqresult_t *r, int k, int num_facets, int fmax, int num_aggs, uint8_t *buf) {
    m_slot_hdr_t h;
    memset(&h, 0, sizeof(h));
    h.num_hits = r->num_hits; h.rank_count = r->rank_count < k ? r->rank_count : k;
    h.num_facets = num_facets; h.num_aggs = num_aggs;
    for (int f = 0; f < num_facets; f++) h.rcount[f] = r->rcount[f] < fmax ? r->rcount[f] : fmax;
    memcpy(buf, &h, sizeof(h));
    uint8_t *p = buf + sizeof(h);
    if (h.rank_count) memcpy(p, r->ranks, (size_t)h.rank_count * sizeof(m_docrank_t));
    p += (size_t)k * sizeof(m_docrank_t);
    for (int f = 0; f < num_facets; f++) {
        if (h.rcount[f]) memcpy(p, r->fc[f], (size_t)h.rcount[f] * sizeof(m_facet_count_t));
        p += (size_t)fmax * sizeof(m_facet_count_t);
    }
    for (int a = 0; a < num_aggs; a++) {
        m_agg_t ag = r->aggs[a]; ag.card_ids = NULL;          /* card_n stays: the ids follow in the card buffer */
        memcpy(p, &ag, sizeof(ag)); p += sizeof(ag);
    }
}

void m_slot_unpack(const uint8_t *buf, int k, int num_facets, int fmax, int num_aggs, int shard_idx, m_sqresult_t *out) {
    m_slot_hdr_t h;
    memcpy(&h, buf, sizeof(h));
    memset(out, 0, sizeof(*out));
    out->num_hits = h.num_hits; out->rank_count = h.rank_count; out->num_facets = num_facets; out->num_aggs = num_aggs;
    const uint8_t *p = buf + sizeof(h);
    out->ranks = malloc((size_t)(h.rank_count ? h.rank_count : 1) * sizeof(m_docrank_t));
    memcpy(out->ranks, p, (size_t)h.rank_count * sizeof(m_docrank_t));
    p += (size_t)k * sizeof(m_docrank_t);
    for (int f = 0; f < num_facets; f++) {
        out->rcount[f] = h.rcount[f];
        out->fc[f] = malloc((size_t)(h.rcount[f] ? h.rcount[f] : 1) * sizeof(m_facet_count_t));
        memcpy(out->fc[f], p, (size_t)h.rcount[f] * sizeof(m_facet_count_t));
        for (int x = 0; x < h.rcount[f]; x++) out->fc[f][x].shard_id = (uint8_t)shard_idx;
        p += (size_t)fmax * sizeof(m_facet_count_t);
    }
    for (int a = 0; a < num_aggs; a++) {
        memcpy(&out->aggs[a], p, sizeof(m_agg_t)); p += sizeof(m_agg_t);
        out->aggs[a].card_ids = NULL; out->aggs[a].card_n = 0;   /* set again by m_slot_card_attach (the slot keeps the size) */
    }
}

/* ---- the cardinality side channel: ids of every (result, agg) of a batch, concatenated */
uint64_t m_slot_card_count(const m_sqresult_t *rs, int n, int num_aggs) {
    uint64_t t = 0;
    for (int i = 0; i < n; i++) for (int a = 0; a < num_aggs; a++) if (rs[i].aggs[a].type == MA_CARDINALITY) t += rs[i].aggs[a].card_n;
    return t;
}
void m_slot_card_pack(const m_sqresult_t *rs, int n, int num_aggs, uint32_t *buf) {
    uint64_t at = 0;
    for (int i = 0; i < n; i++) for (int a = 0; a < num_aggs; a++) {
        const m_agg_t *ag = &rs[i].aggs[a];
        if (ag->type != MA_CARDINALITY || !ag->card_n) continue;
        memcpy(buf + at, ag->card_ids, (size_t)ag->card_n * sizeof(uint32_t)); at += ag->card_n;
    }
}
/* one rank's gathered slots (nq x local_shards, in pack order) and its card buffer: gives every cardinality agg of
 * sq[...] its ids back.  The slots are walked in pack order whether or not the caller unpacked them all (q_first /
 * q_step select the queries that were: m_slot_unpack_gathered_subset; 0 / 1 = all).  Returns 0, or -1 when the buffer
 * is shorter than the slots say. */
int m_slot_card_attach(const uint8_t *slots, const uint32_t *card, uint64_t card_n, int rank, int nranks, int nq, int local_shards,
                       int k, int num_facets, int fmax, int num_aggs, int q_first, int q_step, m_sqresult_t *sq) {
    size_t sb = m_slot_bytes(k, num_facets, fmax, num_aggs);
    int total = nranks * local_shards;
    uint64_t at = 0;
    for (int q = 0; q < nq; q++)
        for (int sidx = 0; sidx < local_shards; sidx++) {
            const uint8_t *p = slots + ((size_t)q * local_shards + sidx) * sb + sizeof(m_slot_hdr_t) +
                               (size_t)k * sizeof(m_docrank_t) + (size_t)num_facets * (size_t)fmax * sizeof(m_facet_count_t);
            const int mine = q >= q_first && (q - q_first) % q_step == 0;
            for (int a = 0; a < num_aggs; a++) {
                m_agg_t ag; memcpy(&ag, p + (size_t)a * sizeof(m_agg_t), sizeof(ag));
                if (ag.type != MA_CARDINALITY || !ag.card_n) continue;
                if (at + ag.card_n > card_n) return -1;
                if (mine) {
                    m_sqresult_t *o = &sq[(size_t)((q - q_first) / q_step) * total + (size_t)rank * local_shards + sidx];
                    o->aggs[a].card_ids = malloc((size_t)ag.card_n * sizeof(uint32_t));
                    memcpy(o->aggs[a].card_ids, card + at, (size_t)ag.card_n * sizeof(uint32_t));
                    o->aggs[a].card_n = ag.card_n;
                }
                at += ag.card_n;
            }
        }
    return 0;
}

void m_sqresult_free_batch(m_sqresult_t *rs, int n) { for (int i = 0; i < n; i++) m_sqresult_free(&rs[i]); }

/* pack / unpack / merge whole batches (one call per step from the launcher) */
void m_slot_pack_batch(const m_sqresult_t *rs, int n, int k, int num_facets, int fmax, int num_aggs, uint8_t *buf) {
    size_t sb = m_slot_bytes(k, num_facets, fmax, num_aggs);
    for (int i = 0; i < n; i++) m_slot_pack(&rs[i], k, num_facets, fmax, num_aggs, buf + (size_t)i * sb);
}

/* gathered layout: [rank][query][local shard]; output: sq[query * total_shards + rank * local + s] */
void m_slot_unpack_gathered(const uint8_t *buf, int nranks, int nq, int local_shards, int k, int num_facets,
                            int fmax, int num_aggs, m_sqresult_t *sq) {
    size_t sb = m_slot_bytes(k, num_facets, fmax, num_aggs);
    int total = nranks * local_shards;
    for (int r = 0; r < nranks; r++)
        for (int q = 0; q < nq; q++)
            for (int s = 0; s < local_shards; s++) {
                int shard = r * local_shards + s;
                m_slot_unpack(buf + (((size_t)r * nq + q) * local_shards + s) * sb, k, num_facets, fmax, num_aggs,
                              shard, &sq[(size_t)q * total + shard]);
            }
}

/* the same for every q_step-th query starting at q_first (each rank of an N-rank job merges its own
 * share of the batch out of the gathered slots): sq[i * total + shard] for the i-th selected query */
void m_slot_unpack_gathered_subset(const uint8_t *buf, int nranks, int nq, int local_shards, int k, int num_facets,
                                   int fmax, int num_aggs, int q_first, int q_step, m_sqresult_t *sq) {
    size_t sb = m_slot_bytes(k, num_facets, fmax, num_aggs);
    int total = nranks * local_shards;
    int i = 0;
    for (int q = q_first; q < nq; q += q_step, i++)
        for (int r = 0; r < nranks; r++)
            for (int s = 0; s < local_shards; s++) {
                int shard = r * local_shards + s;
                m_slot_unpack(buf + (((size_t)r * nq + q) * local_shards + s) * sb, k, num_facets, fmax, num_aggs,
                              shard, &sq[(size_t)i * total + shard]);
            }
}

/* returns the sum of total_hits over the batch (a checksum the launcher prints) */
long m_form_result_batch(const m_query_t *const *queries, m_sqresult_t *sq, int nq, int num_shards,
                         m_result_t *outs, int keep) {
    long hits = 0;
    for (int q = 0; q < nq; q++) {
        m_result_t tmp;
        m_result_t *o = keep ? &outs[q] : &tmp;
        m_form_result(queries[q], sq + (size_t)q * num_shards, num_shards, o);
        hits += o->total_hits;
        if (!keep) m_result_free(o);
    }
    return hits;
}
