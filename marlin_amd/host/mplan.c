/* mplan.c -- see mplan.h */
#include "mplan.h"
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stdio.h>
#include <pthread.h>
#include <time.h>

/* ------------------------------------------------------------------ mirror export */
int mmirror_create(const mshard_t *s, int device, mmirror_t **out) {
    mmirror_t *m = calloc(1, sizeof(*m));
    m->s = s; m->device = device;
    m->W = s->word_count; m->T = s->tword_count;
    uint32_t pid = 1 + m->W + m->T;
    for (int f = 0; f < s->num_facets; f++) { m->fbase[f] = pid; pid += s->nfid[f]; }
    m->bbase = pid; pid += 2u * (uint32_t)s->num_bools;
    m->num_postings = pid;

    uint64_t total = s->has_used ? s->used_n : s->max_docid;
    total += s->wp_off[m->W + 1] + s->tp_off[m->T + 1];
    for (int f = 0; f < s->num_facets; f++) total += s->fp_off[f][s->nfid[f]];
    for (int i = 0; i < s->num_bools; i++) total += s->bp_n[i][0] + s->bp_n[i][1];
    uint64_t *off = malloc(((size_t)pid + 1) * sizeof(uint64_t));
    uint32_t *doc = malloc((total ? total : 1) * sizeof(uint32_t));
    uint64_t at = 0;
    uint32_t p = 0;
    off[p++] = 0;                                           /* pid 0: every used docid (shard.c:51-53) */
    if (s->has_used) { memcpy(doc, s->used_doc, s->used_n * sizeof(uint32_t)); at = s->used_n; }
    else for (uint32_t d = 1; d <= s->max_docid; d++) doc[at++] = d;
    for (uint32_t w = 1; w <= m->W; w++) {
        off[p++] = at;
        uint64_t n = s->wp_off[w + 1] - s->wp_off[w];
        memcpy(doc + at, s->wp_doc + s->wp_off[w], n * sizeof(uint32_t)); at += n;
    }
    for (uint32_t t = 1; t <= m->T; t++) {
        off[p++] = at;
        uint64_t n = s->tp_off[t + 1] - s->tp_off[t];
        memcpy(doc + at, s->tp_doc + s->tp_off[t], n * sizeof(uint32_t)); at += n;
    }
    for (int f = 0; f < s->num_facets; f++) {
        for (uint32_t k = 0; k < s->nfid[f]; k++) {
            off[p++] = at;
            uint64_t n = s->fp_off[f][k + 1] - s->fp_off[f][k];
            memcpy(doc + at, s->fp_doc[f] + s->fp_off[f][k], n * sizeof(uint32_t)); at += n;
        }
    }
    for (int i = 0; i < s->num_bools; i++) {
        for (int v = 0; v < 2; v++) {
            off[p++] = at;
            if (s->bp_n[i][v]) memcpy(doc + at, s->bp_doc[i][v], s->bp_n[i][v] * sizeof(uint32_t));
            at += s->bp_n[i][v];
        }
    }
    off[p] = at;

    mgpu_shard_desc_t d;
    memset(&d, 0, sizeof(d));
    d.max_docid = s->max_docid;
    d.num_numbers = s->num_numbers; d.num_facets = s->num_facets;
    d.num_postings = pid; d.post_off = off; d.post_doc = doc;
    d.dd_off = s->dd_off; d.dd = s->dd;
    for (int i = 0; i < s->num_numbers; i++) { d.np_n[i] = s->np_n[i]; d.np_doc[i] = s->np_doc[i]; }
    int rc = mgpu_shard_create(device, &d, &m->g);
    free(off); free(doc);
    if (rc) { free(m); return rc; }
    /* the dictionary as a flat word list, for the device-side term lookup */
    if (s->dict) {
        uint32_t *chars = NULL, *woff = NULL, *wwid = NULL, nw = 0;
        if (mdict_export_words(s->dict, &chars, &woff, &wwid, &nw)) { mgpu_shard_destroy(m->g); free(m); return MGPU_E_ARG; }
        rc = mgpu_dict_create(device, chars, woff, wwid, nw, &m->gd);
        free(chars); free(woff); free(wwid);
        if (rc) { mgpu_shard_destroy(m->g); free(m); return rc; }
    }
    *out = m;
    return 0;
}

void mmirror_destroy(mmirror_t *m) {
    if (!m) return;
    mgpu_dict_destroy(m->gd);
    mgpu_shard_destroy(m->g);
    free(m);
}

/* ------------------------------------------------------------------ plan */
void mplan_free(mplan_t *pl) {
    for (int t = 0; t < MGPU_MAX_TERMS; t++) free(pl->term_post[t]);
    free(pl->wids); free(pl->fops);
    memset(pl, 0, sizeof(*pl));
}

static void fop_push(mplan_t *pl, int op, int a, uint64_t lo, uint64_t hi) {
    if (pl->p.num_fops == pl->fops_cap) {
        pl->fops_cap = pl->fops_cap ? pl->fops_cap * 2 : 16;
        pl->fops = realloc(pl->fops, sizeof(mgpu_fop_t) * (size_t)pl->fops_cap);
    }
    mgpu_fop_t *f = &pl->fops[pl->p.num_fops++];
    f->op = op; f->a = a; f->lo = lo; f->hi = hi;
}

static int check_op(int f, double a, double b) {            /* check_double_oper, filter_apply.c:127-144 */
    switch (f) {
        case MF_LT: return a < b;
        case MF_LTE: return a <= b;
        case MF_GT: return a > b;
        case MF_GTE: return a >= b;
        default: return 0;
    }
}

/* first index k with val[k] <= x (values descending) */
static uint64_t lb_desc(const double *val, uint64_t n, double x) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (val[mid] > x) lo = mid + 1; else hi = mid; }
    return lo;
}

/* Lowers one filter node (post order, filter_apply.c:207-213).  Returns 1 when the node
 * leaves a bitmap on the stack, 0 when the reference would leave fr_bmap NULL. */
static int lower_filter(const mmirror_t *m, const m_filter_t *f, mplan_t *pl) {
    const mshard_t *s = m->s;
    switch (f->type) {
        case MF_EQ: case MF_NE: {
            int have = 0;
            if (f->field_type == MT_NUMBER) {                        /* num_eq_filter */
                int fi = f->i_priority;
                if (fi >= 0 && fi < s->num_numbers) {
                    uint64_t n = s->np_n[fi];
                    uint64_t a = lb_desc(s->np_val[fi], n, f->numval), b = a;
                    while (b < n && s->np_val[fi][b] == f->numval) b++;
                    fop_push(pl, MGPU_F_PUSH_NUMRANGE, fi, a, b);
                } else fop_push(pl, MGPU_F_PUSH_EMPTY, 0, 0, 0);
                have = 1;
            } else if (f->field_type == MT_STRING) {                 /* str_eq_filter */
                if (f->is_facet) {
                    int ff = f->f_priority;
                    int found = 0;
                    if (ff >= 0 && ff < s->num_facets) {
                        uint32_t lo = 0, hi = s->nfid[ff];
                        while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (s->fid[ff][mid] < f->facet_id) lo = mid + 1; else hi = mid; }
                        if (lo < s->nfid[ff] && s->fid[ff][lo] == f->facet_id) {
                            fop_push(pl, MGPU_F_PUSH_POSTING, (int)(m->fbase[ff] + lo), 0, 0);
                            found = 1;
                        }
                    }
                    if (!found) fop_push(pl, MGPU_F_PUSH_EMPTY, 0, 0, 0);
                    have = 1;
                }
            } else if (f->field_type == MT_BOOLEAN) {                /* bool_eq_filter */
                int bi = f->i_priority;
                if (bi >= 0 && bi < s->num_bools)
                    fop_push(pl, MGPU_F_PUSH_POSTING, (int)(m->bbase + 2u * (uint32_t)bi + (f->numval != 0.0 ? 1u : 0u)), 0, 0);
                else fop_push(pl, MGPU_F_PUSH_EMPTY, 0, 0, 0);
                have = 1;
            }
            if (f->type == MF_NE) {                                  /* ne_filter */
                if (!have) fop_push(pl, MGPU_F_PUSH_EMPTY, 0, 0, 0);
                fop_push(pl, MGPU_F_DOCS_ANDNOT, 0, 0, 0);
                have = 1;
            }
            return have;
        }
        case MF_OR: case MF_NIN: case MF_AND: {
            int cnt = 0;
            for (int i = 0; i < f->num_children; i++) {
                if (lower_filter(m, f->children[i], pl)) {
                    cnt++;
                    if (cnt > 1) fop_push(pl, f->type == MF_AND ? MGPU_F_AND : MGPU_F_OR, 0, 0, 0);
                }
            }
            if (cnt == 0) fop_push(pl, MGPU_F_PUSH_EMPTY, 0, 0, 0);  /* oper_and / oper_or of nothing */
            if (f->type == MF_NIN) fop_push(pl, MGPU_F_DOCS_ANDNOT, 0, 0, 0);
            return 1;
        }
        case MF_GT: case MF_GTE: case MF_LT: case MF_LTE: {          /* num_filter */
            /* children of a leaf are evaluated first and ignored (none in practice) */
            if (f->field_type != MT_NUMBER) return 0;
            int fi = f->i_priority;
            if (fi < 0 || fi >= s->num_numbers) { fop_push(pl, MGPU_F_PUSH_EMPTY, 0, 0, 0); return 1; }
            uint64_t n = s->np_n[fi];
            const double *val = s->np_val[fi];
            if (f->type == MF_GT || f->type == MF_GTE) {
                uint64_t lo = 0, hi = n;                             /* first index failing the test */
                while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (check_op(f->type, val[mid], f->numval)) lo = mid + 1; else hi = mid; }
                fop_push(pl, MGPU_F_PUSH_NUMRANGE, fi, 0, lo);
            } else {
                uint64_t lo = 0, hi = n;                             /* first index passing the test */
                while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (check_op(f->type, val[mid], f->numval)) hi = mid; else lo = mid + 1; }
                fop_push(pl, MGPU_F_PUSH_NUMRANGE, fi, lo, n);
            }
            return 1;
        }
        case MF_RANGE: {                                             /* num_range_filter */
            if (f->field_type != MT_NUMBER) return 0;
            int fi = f->i_priority;
            if (fi < 0 || fi >= s->num_numbers) { fop_push(pl, MGPU_F_PUSH_EMPTY, 0, 0, 0); return 1; }
            uint64_t n = s->np_n[fi];
            const double *val = s->np_val[fi];
            /* the cursor starts at the first key <= numval2 (MDB_SET_RANGE on a descending tree);
             * keys failing the upper test are skipped, the first key failing the lower test ends
             * the walk (filter_apply.c:182-191).  Keys descend, so both tests are monotone. */
            uint64_t a = lb_desc(val, n, f->numval2), b;
            if (f->numcmp2 == MF_LT) {
                uint64_t lo = a, hi = n;
                while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (val[mid] < f->numval2) hi = mid; else lo = mid + 1; }
                a = lo;
            } else if (f->numcmp2 != MF_LTE) {
                a = n;
            }
            {
                uint64_t lo = a, hi = n;
                while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (check_op(f->numcmp1, val[mid], f->numval)) lo = mid + 1; else hi = mid; }
                b = lo;
            }
            fop_push(pl, MGPU_F_PUSH_NUMRANGE, fi, a, b);
            return 1;
        }
        default:
            return 0;
    }
}

/* pre (may be NULL): term results that the device already produced; pre_have[t] says which */
static int mplan_build_with(const mmirror_t *m, const m_query_t *q, int capture, mplan_t *pl, m_termresult_t *pre, const uint8_t *pre_have) {
    if (q->num_terms > MGPU_MAX_TERMS || q->num_words > MGPU_MAX_WORDS) { memset(pl, 0, sizeof(*pl)); return MGPU_E_LIMIT; }
    const mshard_t *s = m->s;
    /* lookup_terms (squery.c:144-167): the trie walk stays on the host unless the device did it */
    m_termresult_t tr[M_MAX_QTERMS];
    uint32_t exact_wid[M_MAX_QWORDS];
    for (int t = 0; t < q->num_terms; t++) {
        if (pre && pre_have[t]) { tr[t] = pre[t]; memset(&pre[t], 0, sizeof(pre[t])); }      /* ownership moves here */
        else mdict_lookup_term(s->dict, &q->terms[t], &tr[t]);
    }
    for (int i = 0; i < q->num_words; i++) exact_wid[i] = mdict_lookup_exact(s->dict, &q->words[i]);
    int rc = mplan_from_terms(m, q, tr, exact_wid, capture, pl);
    for (int t = 0; t < q->num_terms; t++) m_termresult_free(&tr[t]);
    return rc;
}

int mplan_build(const mmirror_t *m, const m_query_t *q, int capture, mplan_t *pl) { return mplan_build_with(m, q, capture, pl, NULL, NULL); }

/* Everything of the plan after the trie walk.  tr[t] = dtrie_lookup_term's result for term t (its wordids map
 * is extended here with the words under a twid, as set_wids_under_twid does, squery.c:62-70); exact_wid[i] =
 * dtrie_lookup_exact's result for query word i (0 = none).  q->terms / q->words themselves are not read, so a
 * caller whose dictionary is not an mdict (the reference's dtrie: shim/squery_gpu.c) can use it as is. */
int mplan_from_terms(const mmirror_t *m, const m_query_t *q, m_termresult_t *tr, const uint32_t *exact_in,
                     int capture, mplan_t *pl) {
    memset(pl, 0, sizeof(*pl));
    const mshard_t *s = m->s;
    mgpu_plan_t *p = &pl->p;
    p->num_terms = q->num_terms; p->num_words = q->num_words;
    if (q->num_terms > MGPU_MAX_TERMS || q->num_words > MGPU_MAX_WORDS) return MGPU_E_LIMIT;

    m_u32map_t all; m_u32map_init(&all);
    /* process_termresult (squery.c:72-118) */
    for (int t = 0; t < q->num_terms; t++) {
        const uint32_t *docs;
        if (tr[t].twid == 0 && tr[t].wordids.n == 0) { p->term_npost[t] = 0; continue; }
        if (tr[t].twid) {
            uint64_t n = mshard_twid_posting(s, tr[t].twid, &docs);
            pl->term_post[t] = malloc(sizeof(uint32_t));
            if (n) {
                pl->term_post[t][0] = m->W + tr[t].twid;
                p->term_npost[t] = 1; p->term_nzero[t] = 1;          /* a twid matches with zero typos (squery.c:87-88) */
                const uint32_t *wids;
                uint64_t nw = mshard_twid_wids(s, tr[t].twid, &wids);
                for (uint64_t i = 0; i < nw; i++) { m_u32map_set(&tr[t].wordids, wids[i], 0); m_u32map_set(&all, wids[i], 0); }
            }
            continue;
        }
        pl->term_post[t] = malloc(sizeof(uint32_t) * (tr[t].wordids.n ? tr[t].wordids.n : 1));
        uint32_t np = 0, nz = 0;
        for (uint32_t k = 0; k < tr[t].wordids.cap; k++) {
            uint32_t wid = tr[t].wordids.keys[k];
            if (!wid) continue;
            if (!mshard_wid_posting(s, wid, &docs)) continue;
            m_u32map_set(&all, wid, tr[t].wordids.vals[k]);
            pl->term_post[t][np++] = wid;
            if (tr[t].wordids.vals[k] == 0) {                        /* zero typos: kept in front (tzero_typo_bmap) */
                uint32_t x = pl->term_post[t][nz]; pl->term_post[t][nz] = wid; pl->term_post[t][np - 1] = x; nz++;
            }
        }
        p->term_npost[t] = np; p->term_nzero[t] = nz;
    }
    for (int t = 0; t < q->num_terms; t++) p->term_post[t] = pl->term_post[t];

    /* exact wids (squery.c:159-166) */
    uint32_t exact_wid[M_MAX_QWORDS];
    for (int i = 0; i < q->num_words; i++) {
        const uint32_t *docs;
        uint32_t wid = exact_in[i];
        exact_wid[i] = 0;
        if (wid && mshard_wid_posting(s, wid, &docs)) { exact_wid[i] = wid; p->exact_present |= (uint16_t)(1u << i); }
    }
    /* wid table: all_wordids distance, first matching term, exact bits */
    {
        uint32_t cap = all.n + (uint32_t)q->num_words + 1;
        pl->wids = malloc(sizeof(mgpu_wid_entry_t) * cap);
        uint32_t nw = 0;
        for (uint32_t k = 0; k < all.cap; k++) {
            uint32_t wid = all.keys[k];
            if (!wid) continue;
            mgpu_wid_entry_t *e = &pl->wids[nw++];
            e->wid = wid; e->dist = (uint8_t)all.vals[k]; e->mterm = 0xFF; e->exact_mask = 0;
            for (int t = 0; t < q->num_terms; t++)
                if (m_u32map_get(&tr[t].wordids, wid, 0xFF) != 0xFF) { e->mterm = (uint8_t)t; break; }
            for (int i = 0; i < q->num_words; i++) if (exact_wid[i] == wid) e->exact_mask |= (uint16_t)(1u << i);
        }
        for (int i = 0; i < q->num_words; i++) {
            if (!exact_wid[i] || m_u32map_get(&all, exact_wid[i], -1) >= 0) continue;
            int dup = 0;
            for (uint32_t k = all.n; k < nw; k++) if (pl->wids[k].wid == exact_wid[i]) { pl->wids[k].exact_mask |= (uint16_t)(1u << i); dup = 1; }
            if (dup) continue;
            mgpu_wid_entry_t *e = &pl->wids[nw++];
            e->wid = exact_wid[i]; e->dist = 0xFF; e->mterm = 0xFF; e->exact_mask = (uint16_t)(1u << i);
        }
        p->num_wids = nw; p->wids = pl->wids;
    }
    m_u32map_free(&all);

    /* filter */
    if (q->filter) {
        p->has_filter = 1;
        p->filter_null = lower_filter(m, q->filter, pl) ? 0 : 1;
        p->fops = pl->fops;
    }
    /* config */
    p->hits_per_page = q->cfg.hits_per_page; p->page_num = q->page_num;
    p->max_facet_results = q->cfg.max_facet_results;
    for (int f = 0; f < s->num_facets && f < 32; f++) if (q->cfg.facet_enabled[f]) p->facet_enabled |= 1u << f;
    p->rank_by = q->cfg.rank_by; p->rank_asc = q->cfg.rank_asc;
    p->full_scan = q->cfg.full_scan ? 1 : 0; p->full_scan_threshold = q->cfg.full_scan_threshold;
    for (int f = 0; f < s->num_facets && f < 32; f++) p->facet_pid_base[f] = m->fbase[f];
    for (int i = 0; i < MGPU_MAX_RULES; i++) p->rules[i] = q->rank_rule[i];
    p->num_aggs = q->num_aggs;
    for (int a = 0; a < q->num_aggs; a++) { p->agg_type[a] = q->aggs[a].type; p->agg_field[a] = q->aggs[a].field; }
    p->capture = capture;
    return 0;
}

/* ------------------------------------------------------------------ execution */
int msq_plan_threads = 1;
void msq_set_plan_threads(int n) { msq_plan_threads = n; }

struct plan_job {
    mmirror_t *const *mirrors; int num_shards; const m_query_t *const *queries; int capture;
    mplan_t *plans; mgpu_plan_t *gp; mgpu_shard_t **gs; int lo, hi, rc;
    m_termresult_t *pre; uint8_t *pre_have;        /* per (query, shard) M_MAX_QTERMS entries each, or NULL */
};
static void *plan_worker(void *arg) {
    struct plan_job *j = arg;
    for (int i = j->lo; i < j->hi; i++) {
        int qi = i / j->num_shards, sidx = i % j->num_shards;
        int rc = mplan_build_with(j->mirrors[sidx], j->queries[qi], j->capture, &j->plans[i],
                                  j->pre ? j->pre + (size_t)i * M_MAX_QTERMS : NULL, j->pre ? j->pre_have + (size_t)i * M_MAX_QTERMS : NULL);
        if (rc) j->rc = rc;
        j->gp[i] = j->plans[i].p;
        j->gs[i] = j->mirrors[sidx]->g;
    }
    return NULL;
}

mgpu_shard_t *mmirror_gpu(mmirror_t *m) { return m ? m->g : NULL; }

struct msq_prepared {
    int n, nq, num_shards;
    mmirror_t *const *mirrors; const m_query_t **queries;
    mplan_t *plans; mgpu_plan_t *gp; mgpu_shard_t **gs;
};

/* lookup_terms (squery.c:144-167) + filter leaf resolution for every (query, shard) pair */
static int prepare_common(mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries,
                          int nq, int capture, m_termresult_t *pre, uint8_t *pre_have, msq_prepared_t **out) {
    int n = nq * num_shards;
    msq_prepared_t *pp = calloc(1, sizeof(*pp));
    pp->n = n; pp->nq = nq; pp->num_shards = num_shards; pp->mirrors = mirrors;
    pp->queries = calloc((size_t)(nq ? nq : 1), sizeof(*pp->queries));
    memcpy(pp->queries, queries, (size_t)nq * sizeof(*pp->queries));
    pp->plans = calloc((size_t)(n ? n : 1), sizeof(mplan_t));
    pp->gp = calloc((size_t)(n ? n : 1), sizeof(mgpu_plan_t));
    pp->gs = calloc((size_t)(n ? n : 1), sizeof(mgpu_shard_t *));
    int rc = 0;
    int nthreads = msq_plan_threads;
    if (nthreads > n) nthreads = n;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    struct plan_job jobs[64];
    pthread_t th[64];
    for (int t = 0; t < nthreads; t++) {
        jobs[t].mirrors = mirrors; jobs[t].num_shards = num_shards; jobs[t].queries = queries;
        jobs[t].capture = capture; jobs[t].plans = pp->plans; jobs[t].gp = pp->gp; jobs[t].gs = pp->gs;
        jobs[t].lo = (int)((long)n * t / nthreads); jobs[t].hi = (int)((long)n * (t + 1) / nthreads); jobs[t].rc = 0;
        jobs[t].pre = pre; jobs[t].pre_have = pre_have;
    }
    if (nthreads == 1) plan_worker(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, plan_worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    for (int t = 0; t < nthreads; t++) if (jobs[t].rc) rc = jobs[t].rc;
    if (rc) { msq_prepared_free(pp); pp = NULL; }
    *out = pp;
    return rc;
}

int msq_prepare_batch(mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries,
                      int nq, int capture, msq_prepared_t **out) {
    return prepare_common(mirrors, num_shards, queries, nq, capture, NULL, NULL, out);
}

int msq_prepare_batch_dev(mgpu_ctx_t *ctx, mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries,
                          int nq, int capture, msq_prepared_t **out) {
    *out = NULL;
    const int n = nq * num_shards;
    if (n <= 0) return prepare_common(mirrors, num_shards, queries, nq, capture, NULL, NULL, out);
    m_termresult_t *pre = calloc((size_t)n * M_MAX_QTERMS, sizeof(*pre));
    uint8_t *have = calloc((size_t)n * M_MAX_QTERMS, 1);
    /* every term that dtrie_lookup_term would walk the trie for (mode 2) becomes one device lookup */
    size_t nl = 0;
    for (int i = 0; i < n; i++) {
        const m_query_t *q = queries[i / num_shards];
        if (!mirrors[i % num_shards]->gd || q->num_terms > M_MAX_QTERMS) continue;
        for (int t = 0; t < q->num_terms; t++)
            if (mdict_lookup_mode(&q->terms[t]) == 2 && q->terms[t].word.length >= 1 && q->terms[t].word.length <= MGPU_MAX_TERM_LEN) nl++;
    }
    mgpu_term_lookup_t *lk = malloc(sizeof(*lk) * (nl ? nl : 1));
    uint32_t *slot = malloc(sizeof(uint32_t) * (nl ? nl : 1));          /* lookup -> index into pre[] */
    nl = 0;
    for (int i = 0; i < n; i++) {
        const m_query_t *q = queries[i / num_shards];
        if (!mirrors[i % num_shards]->gd || q->num_terms > M_MAX_QTERMS) continue;
        for (int t = 0; t < q->num_terms; t++) {
            const m_term_t *tm = &q->terms[t];
            if (mdict_lookup_mode(tm) != 2 || tm->word.length < 1 || tm->word.length > MGPU_MAX_TERM_LEN) continue;
            lk[nl].dict = mirrors[i % num_shards]->gd; lk[nl].chars = tm->word.chars; lk[nl].len = tm->word.length; lk[nl].prefix = tm->prefix;
            slot[nl] = (uint32_t)((size_t)i * M_MAX_QTERMS + (size_t)t);
            have[slot[nl]] = 1;                                      /* an empty result is a result too */
            nl++;
        }
    }
    int rc = 0;
    static int trace = -1;
    if (trace < 0) trace = getenv("MSQ_PREP_TRACE") != NULL;
    struct timespec ts0, ts1, ts2;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    size_t nhits_total = 0;
    /* at most 65535 lookups per call */
    for (size_t lo = 0; lo < nl && !rc; lo += 65535) {
        const int cnt = (int)(nl - lo < 65535 ? nl - lo : 65535);
        mgpu_term_hit_t *hits = NULL; uint32_t nh = 0;
        rc = mgpu_lookup_terms(ctx, lk + lo, cnt, &hits, &nh);
        if (rc) break;
        for (uint32_t h = 0; h < nh; h++) {
            if (hits[h].lookup >= (uint32_t)cnt) { rc = MGPU_E_HIP; break; }
            m_termresult_t *tr = &pre[slot[lo + hits[h].lookup]];
            /* add_wordid_to_result (dtrie.c:633-640): least distance */
            if ((int)hits[h].dist < m_u32map_get(&tr->wordids, hits[h].wid, 0xFF)) m_u32map_set(&tr->wordids, hits[h].wid, (int32_t)hits[h].dist);
        }
        nhits_total += nh;
        mgpu_term_hits_free(hits);
    }
    free(lk); free(slot);
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    if (!rc) rc = prepare_common(mirrors, num_shards, queries, nq, capture, pre, have, out);
    if (trace) {
        clock_gettime(CLOCK_MONOTONIC, &ts2);
        fprintf(stderr, "[prepare_dev] %zu lookups, %zu hits: device scan + maps %.3f ms, plans %.3f ms\n", nl, nhits_total,
                (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6, (ts2.tv_sec - ts1.tv_sec) * 1e3 + (ts2.tv_nsec - ts1.tv_nsec) * 1e-6);
    }
    for (size_t i = 0; i < (size_t)n * M_MAX_QTERMS; i++) if (pre[i].wordids.cap) m_termresult_free(&pre[i]);
    free(pre); free(have);
    return rc;
}

void msq_prepared_free(msq_prepared_t *pp) {
    if (!pp) return;
    for (int i = 0; i < pp->n; i++) mplan_free(&pp->plans[i]);
    free(pp->plans); free(pp->gp); free(pp->gs); free(pp->queries); free(pp);
}

int msq_execute_batch(mgpu_ctx_t *ctx, mmirror_t *const *mirrors, int num_shards,
                      const m_query_t *const *queries, int nq, int capture, m_sqresult_t *outs) {
    msq_prepared_t *pp = NULL;
    int rc = msq_prepare_batch(mirrors, num_shards, queries, nq, capture, &pp);
    if (rc) { memset(outs, 0, sizeof(*outs) * (size_t)nq * (size_t)num_shards); return rc; }
    rc = msq_execute_prepared(ctx, pp, outs);
    msq_prepared_free(pp);
    return rc;
}

/* the part of execute_squery after lookup_terms: bitmap formula, filter, ranking, facets, aggs,
 * top-2N (squery.c:480-536) as one GPU batch -- in two halves: submit enqueues the batch on the context's
 * stream and returns, collect waits for it and fills outs.  One batch in flight per context. */
int msq_submit_prepared(mgpu_ctx_t *ctx, const msq_prepared_t *pp) {
    return mgpu_execute_begin(ctx, pp->gs, pp->gp, pp->n);
}

/* mgpu_sqout_t -> what execute_squery leaves in struct squery_result (squery.h:34-46); takes over g's arrays */
static int sqout_to_result(const m_query_t *q, int sidx, int num_facets, mgpu_sqout_t *g, m_sqresult_t *o) {
    memset(o, 0, sizeof(*o));
    if (g->status) return g->status;
    o->num_hits = g->num_hits;
    o->rank_count = g->rank_count;
    o->ranks = (m_docrank_t *)g->ranks; g->ranks = NULL;        /* identical 24-byte layout */
    for (int j = 0; j < o->rank_count; j++) o->ranks[j].shard_id = 0;
    o->num_facets = num_facets;
    for (int f = 0; f < o->num_facets; f++) {
        if (!q->cfg.facet_enabled[f]) continue;
        int rcnt = g->rcount[f];
        o->rcount[f] = rcnt;
        o->fc[f] = calloc((size_t)(rcnt ? rcnt : 1), sizeof(m_facet_count_t));
        for (int x = 0; x < rcnt; x++) {
            o->fc[f][x].facet_id = g->fc[f][x].facet_id; o->fc[f][x].count = g->fc[f][x].count;
            o->fc[f][x].shard_id = (uint8_t)sidx;
        }
    }
    o->num_aggs = q->num_aggs;
    for (int a = 0; a < q->num_aggs; a++) {
        m_agg_t *ag = &o->aggs[a];
        *ag = q->aggs[a]; ag->card_ids = NULL; ag->card_n = 0;
        const mgpu_agg_out_t *ga = &g->aggs[a];
        switch (ag->type) {
            case MA_MAX: if (ga->max > ag->value) ag->value = ga->max; break;
            case MA_MIN: if (ga->min < ag->value) ag->value = ga->min; break;
            case MA_AVG: ag->sum += ga->sum; ag->count += (int)ga->count; break;
            case MA_STATS:
                ag->sum += ga->sum; ag->count += (int)ga->count;
                if (ga->min < ag->min) ag->min = ga->min;
                if (ga->max > ag->max) ag->max = ga->max;
                break;
            case MA_CARDINALITY: ag->card_ids = ga->card_ids; ag->card_n = ga->card_n; g->aggs[a].card_ids = NULL; break;
            default: break;
        }
    }
    o->fast_rank = g->fast_rank != 0;
    o->dbg_docid_map = g->dbg_docid_map; o->dbg_words = g->dbg_words; g->dbg_docid_map = NULL;
    o->dbg_all_ranks = (m_docrank_t *)g->dbg_all_ranks; g->dbg_all_ranks = NULL;
    return 0;
}

int msq_collect_prepared(mgpu_ctx_t *ctx, const msq_prepared_t *pp, m_sqresult_t *outs) {
    int n = pp->n, num_shards = pp->num_shards;
    mmirror_t *const *mirrors = pp->mirrors;
    const m_query_t *const *queries = (const m_query_t *const *)pp->queries;
    mgpu_sqout_t *go = calloc((size_t)(n ? n : 1), sizeof(mgpu_sqout_t));
    int rc = mgpu_execute_end(ctx, go);
    for (int i = 0; i < n; i++) {
        int sidx = i % num_shards;
        if (rc) { memset(&outs[i], 0, sizeof(outs[i])); continue; }
        int r1 = sqout_to_result(queries[i / num_shards], sidx, mirrors[sidx]->s->num_facets, &go[i], &outs[i]);
        if (r1) { rc = r1; m_sqresult_free(&outs[i]); }
    }
    for (int i = 0; i < n; i++) mgpu_sqout_free(&go[i]);
    free(go);
    return rc;
}

/* execute_squery for ONE (shard, query) whose plan the caller built (mplan_from_terms): the call a replacement
 * squery.c makes per shard (shim/squery_gpu.c) */
int msq_execute_plan(mgpu_ctx_t *ctx, const mmirror_t *m, const m_query_t *q, const mplan_t *pl, int shard_idx, m_sqresult_t *out) {
    mgpu_sqout_t g;
    mgpu_shard_t *gs = m->g;
    memset(out, 0, sizeof(*out));
    int rc = mgpu_execute(ctx, &gs, &pl->p, 1, &g);
    if (rc) return rc;
    rc = sqout_to_result(q, shard_idx, m->s->num_facets, &g, out);
    if (rc) m_sqresult_free(out);
    mgpu_sqout_free(&g);
    return rc;
}

int msq_execute_prepared(mgpu_ctx_t *ctx, const msq_prepared_t *pp, m_sqresult_t *outs) {
    int rc = msq_submit_prepared(ctx, pp);
    if (rc) { memset(outs, 0, sizeof(*outs) * (size_t)pp->n); return rc; }
    return msq_collect_prepared(ctx, pp, outs);
}

/* ------------------------------------------------------------------ pipelined serving loop
 * The shape query.c:426-484 has in the reference (a threadpool of workers, each running
 * execute_squery jobs and handing them to form_result), with a batch as the unit of work:
 * nthreads host threads, one GPU context each, started ONCE (msq_pipeline_create) and parked on a
 * condition variable between runs; in a run thread t takes steps t, t + nthreads, ... and for each
 * step executes prepared batch (first + step) % nprep and merges every query of it (form_result).
 * msq_pipeline_warm takes every batch through every context once, so that each context's arena,
 * pinned buffers and histogram pool have their final size before a caller starts a clock. */
/* MSQ_PIPE_TRACE=1: every step's host-side timeline (ms since the run started) on stderr after the run */
typedef struct { int step, t; double sub0, sub1, col0, col1, mrg1; float gpu_ms; } pipe_trace_t;
#define PIPE_TRACE_MAX 4096
static double pipe_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

struct msq_pipeline {
    int nthreads, nctx;
    int trace_on; double t_run0; pipe_trace_t *trace; int ntrace;
    pthread_t th[64];
    pthread_mutex_t mu; pthread_cond_t go, done;
    unsigned gen; int pending, quit;
    /* thread t owns contexts 2t and 2t + 1 (the last thread only one when nctx is odd): while it decodes and
     * merges one batch, its next one is already queued on the GPU */
    mgpu_ctx_t *ctx[128];
    /* the current run */
    const msq_prepared_t *const *preps; int nprep, first, count, every;
    long hits[64]; int rc[64];
    char err[256];
};
typedef struct { msq_pipeline_t *p; int t; } pipe_arg_t;

static void pipe_fail(msq_pipeline_t *p, int t, int rc) {
    /* mgpu_last_error is per thread: keep this thread's message for the caller */
    pthread_mutex_lock(&p->mu);
    if (!p->err[0]) snprintf(p->err, sizeof(p->err), "%s", mgpu_last_error());
    pthread_mutex_unlock(&p->mu);
    p->rc[t] = rc;
}

/* wait for the batch in flight on ctx, merge it (form_result) */
static int pipe_collect(msq_pipeline_t *p, int t, mgpu_ctx_t *ctx, const msq_prepared_t *pp, pipe_trace_t *tr) {
    m_sqresult_t *outs = calloc((size_t)(pp->n ? pp->n : 1), sizeof(m_sqresult_t));
    if (!outs) { p->rc[t] = -1; return -1; }
    if (tr) tr->col0 = pipe_now() - p->t_run0;
    int rc = msq_collect_prepared(ctx, pp, outs);
    if (tr) { tr->col1 = pipe_now() - p->t_run0; mgpu_last_timing(ctx, &tr->gpu_ms, NULL); }
    if (!rc) p->hits[t] += m_form_result_batch((const m_query_t *const *)pp->queries, outs, pp->nq, pp->num_shards, NULL, 0);
    m_sqresult_free_batch(outs, pp->n);
    free(outs);
    if (tr) {
        tr->mrg1 = pipe_now() - p->t_run0;
        pthread_mutex_lock(&p->mu);
        if (p->ntrace < PIPE_TRACE_MAX) p->trace[p->ntrace++] = *tr;
        pthread_mutex_unlock(&p->mu);
    }
    if (rc) pipe_fail(p, t, rc);
    return rc;
}

static void pipe_steps(msq_pipeline_t *p, int t) {
    p->hits[t] = 0; p->rc[t] = 0;
    mgpu_ctx_t *cx[2] = { p->ctx[2 * t], (2 * t + 1 < p->nctx) ? p->ctx[2 * t + 1] : NULL };
    const int ncx = cx[1] ? 2 : 1;
    if (p->every) {
        /* warm-up run: every context of this thread takes every batch once */
        for (int k = 0; k < ncx; k++)
            for (int i = 0; i < p->count; i++) {
                const msq_prepared_t *pp = p->preps[(p->first + i) % p->nprep];
                int rc = msq_submit_prepared(cx[k], pp);
                if (rc) { pipe_fail(p, t, rc); return; }
                if (pipe_collect(p, t, cx[k], pp, NULL)) return;
            }
        return;
    }
    /* steps t, t + nthreads, ...: the next step is submitted before the current one is collected */
    const msq_prepared_t *inflight[2] = { NULL, NULL };
    pipe_trace_t tr[2];
    int slot = 0;
    for (int i = t; i < p->count || inflight[0] || inflight[1]; i += p->nthreads) {
        if (i < p->count) {
            const msq_prepared_t *pp = p->preps[(p->first + i) % p->nprep];
            tr[slot].step = i; tr[slot].t = t; tr[slot].sub0 = pipe_now() - p->t_run0;
            int rc = msq_submit_prepared(cx[slot], pp);
            tr[slot].sub1 = pipe_now() - p->t_run0;
            if (rc) { pipe_fail(p, t, rc); break; }
            inflight[slot] = pp;
        }
        /* collect the older batch: the other slot when there are two, else this one */
        const int older = ncx == 2 ? slot ^ 1 : slot;
        if (inflight[older]) {
            const msq_prepared_t *pp = inflight[older];
            inflight[older] = NULL;
            if (pipe_collect(p, t, cx[older], pp, p->trace_on ? &tr[older] : NULL)) break;
        }
        if (ncx == 2) slot ^= 1;
    }
    /* after a failure: drain what is still in flight so that the contexts stay usable */
    for (int k = 0; k < ncx; k++) if (inflight[k]) {
        m_sqresult_t *outs = calloc((size_t)(inflight[k]->n ? inflight[k]->n : 1), sizeof(m_sqresult_t));
        if (outs) { msq_collect_prepared(cx[k], inflight[k], outs); m_sqresult_free_batch(outs, inflight[k]->n); free(outs); }
    }
}

static void *pipe_thread(void *arg) {
    pipe_arg_t *a = arg;
    msq_pipeline_t *p = a->p; const int t = a->t;
    free(a);
    unsigned seen = 0;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (p->gen == seen && !p->quit) pthread_cond_wait(&p->go, &p->mu);
        if (p->quit) { pthread_mutex_unlock(&p->mu); return NULL; }
        seen = p->gen;
        pthread_mutex_unlock(&p->mu);
        pipe_steps(p, t);
        pthread_mutex_lock(&p->mu);
        if (--p->pending == 0) pthread_cond_signal(&p->done);
        pthread_mutex_unlock(&p->mu);
    }
}

int msq_pipeline_create(mgpu_ctx_t *const *ctxs, int nctx, msq_pipeline_t **out) {
    if (!ctxs || !out || nctx < 1 || nctx > 128) return MGPU_E_ARG;
    msq_pipeline_t *p = calloc(1, sizeof(*p));
    if (!p) return -1;
    const int nthreads = (nctx + 1) / 2;
    p->nthreads = nthreads; p->nctx = nctx;
    p->trace_on = getenv("MSQ_PIPE_TRACE") != NULL;
    if (p->trace_on) p->trace = calloc(PIPE_TRACE_MAX, sizeof(pipe_trace_t));
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->go, NULL); pthread_cond_init(&p->done, NULL);
    for (int t = 0; t < nctx; t++) p->ctx[t] = ctxs[t];
    for (int t = 0; t < nthreads; t++) {
        pipe_arg_t *a = malloc(sizeof(*a));
        a->p = p; a->t = t;
        if (pthread_create(&p->th[t], NULL, pipe_thread, a)) { free(a); p->nthreads = t; msq_pipeline_destroy(p); return -1; }
    }
    *out = p;
    return 0;
}

static int pipe_dispatch(msq_pipeline_t *p, const msq_prepared_t *const *preps, int nprep, int first, int count,
                         int every, long *hits_out) {
    if (!p || !preps || nprep < 1 || count < 0) return MGPU_E_ARG;
    pthread_mutex_lock(&p->mu);
    p->preps = preps; p->nprep = nprep; p->first = first; p->count = count; p->every = every;
    p->err[0] = 0;
    p->ntrace = 0; p->t_run0 = pipe_now();
    p->pending = p->nthreads;
    p->gen++;
    pthread_cond_broadcast(&p->go);
    while (p->pending) pthread_cond_wait(&p->done, &p->mu);
    pthread_mutex_unlock(&p->mu);
    long hits = 0; int rc = 0;
    for (int t = 0; t < p->nthreads; t++) { hits += p->hits[t]; if (p->rc[t] && !rc) rc = p->rc[t]; }
    if (hits_out) *hits_out = hits;
    if (p->trace_on && !every) {
        fprintf(stderr, "[pipe] run of %d steps took %.3f ms\n", count, pipe_now() - p->t_run0);
        for (int i = 0; i < p->ntrace; i++) {
            const pipe_trace_t *x = &p->trace[i];
            fprintf(stderr, "[pipe] step %3d thr %d submit %.3f-%.3f collect %.3f-%.3f merged %.3f gpu %.3f\n",
                    x->step, x->t, x->sub0, x->sub1, x->col0, x->col1, x->mrg1, x->gpu_ms);
        }
    }
    return rc;
}

int msq_pipeline_run(msq_pipeline_t *p, const msq_prepared_t *const *preps, int nprep, int first, int count, long *hits_out) {
    return pipe_dispatch(p, preps, nprep, first, count, 0, hits_out);
}

int msq_pipeline_warm(msq_pipeline_t *p, const msq_prepared_t *const *preps, int nprep) {
    return pipe_dispatch(p, preps, nprep, 0, nprep, 1, NULL);
}

const char *msq_pipeline_error(const msq_pipeline_t *p) { return p ? p->err : ""; }

void msq_pipeline_destroy(msq_pipeline_t *p) {
    if (!p) return;
    pthread_mutex_lock(&p->mu);
    p->quit = 1;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
    for (int t = 0; t < p->nthreads; t++) pthread_join(p->th[t], NULL);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->go); pthread_cond_destroy(&p->done);
    free(p->trace);
    free(p);
}

/* one-shot form: threads created, run, joined */
int msq_run_pipeline(mgpu_ctx_t *const *ctxs, int nctx, const msq_prepared_t *const *preps, int nprep,
                     int first, int count, long *hits_out) {
    msq_pipeline_t *p = NULL;
    int rc = msq_pipeline_create(ctxs, nctx, &p);
    if (rc) return rc;
    rc = msq_pipeline_run(p, preps, nprep, first, count, hits_out);
    msq_pipeline_destroy(p);
    return rc;
}
