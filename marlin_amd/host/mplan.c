/* mplan.c -- see mplan.h */
#include "mplan.h"
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stdio.h>
#include <pthread.h>
#include <time.h>

/* ------------------------------------------------------------------ mirror export */
/* the posting-id scheme alone (no device): what plan building needs of a mirror */
mmirror_t *mmirror_scheme(const mshard_t *s) {
    mmirror_t *m = calloc(1, sizeof(*m));
    if (!m) return NULL;
    m->s = s; m->device = -1;
    m->W = s->word_count; m->T = s->tword_count;
    uint32_t pid = 1 + m->W + m->T;
    for (int f = 0; f < s->num_facets; f++) { m->fbase[f] = pid; pid += s->nfid[f]; }
    m->bbase = pid; pid += 2u * (uint32_t)s->num_bools;
    m->num_postings = pid;
    return m;
}

int mmirror_create(const mshard_t *s, int device, mmirror_t **out) {
    mmirror_t *m = mmirror_scheme(s);
    if (!m) return MGPU_E_LIMIT;
    m->device = device;
    const uint32_t pid = m->num_postings;

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

/* first element of the ascending list [a, a + n) that is > lim */
static uint64_t tail_from(const uint32_t *a, uint64_t n, uint32_t lim) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (a[mid] <= lim) lo = mid + 1; else hi = mid; }
    return lo;
}

/* Row f-2: the shard `snew` is what the mirror's shard became after a committed, append-only bulk write (the old
 * documents keep their docids, words and top-level words their ids: sindex.c hands out docids and the dtrie word ids in
 * insertion order).  The mirror is brought up to date from the difference -- per posting the docids above the old
 * maximum, the new documents' blobs, the new num_dbi order -- with mgpu_shard_apply_delta instead of a rebuild; the
 * dictionary mirror is re-exported only when words were added.  MGPU_E_LIMIT: the mirror's headroom is used up, nothing
 * was changed, build a new mirror (mmirror_create).  MGPU_E_UNSUPPORTED: not an append (documents deleted). */
int mmirror_apply_shard(mmirror_t *m, const mshard_t *snew) {
    if (!m || !m->g || !snew) return MGPU_E_ARG;
    const mshard_t *so = m->s;
    /* a list of the docids in use (sdata->used_bmap) must only have grown at its end: no document deleted */
    if (so->has_used != snew->has_used) return MGPU_E_UNSUPPORTED;
    if (so->has_used) {
        if (snew->used_n < so->used_n || tail_from(snew->used_doc, snew->used_n, so->max_docid) != so->used_n ||
            memcmp(snew->used_doc, so->used_doc, so->used_n * sizeof(uint32_t))) return MGPU_E_UNSUPPORTED;
    }
    if (snew->max_docid < so->max_docid || snew->word_count < so->word_count || snew->tword_count < so->tword_count ||
        snew->num_facets != so->num_facets || snew->num_numbers != so->num_numbers || snew->num_bools != so->num_bools) return MGPU_E_UNSUPPORTED;
    mmirror_t *nm = mmirror_scheme(snew);
    if (!nm) return MGPU_E_LIMIT;
    const uint32_t P1 = nm->num_postings, old_max = so->max_docid;
    uint32_t *old_pid = malloc(sizeof(uint32_t) * (size_t)(P1 ? P1 : 1));
    uint64_t *off = malloc(sizeof(uint64_t) * ((size_t)P1 + 1));
    /* an upper bound of the delta's size: every list's tail */
    uint64_t cap = 0, at = 0;
    int rc = 0;
    if (!old_pid || !off) rc = MGPU_E_LIMIT;
    uint32_t *doc = NULL;
    for (int pass = 0; pass < 2 && !rc; pass++) {
        /* pass 0 sizes, pass 1 fills */
        if (pass == 1) { doc = malloc(sizeof(uint32_t) * (size_t)(cap ? cap : 1)); if (!doc) { rc = MGPU_E_LIMIT; break; } }
        uint32_t p = 0;
        at = 0;
#define TAIL(list, n, oldp) do { const uint32_t *l_ = (list); const uint64_t n_ = (n), t_ = tail_from(l_, n_, old_max); \
            off[p] = at; old_pid[p] = (oldp); if (pass == 1 && n_ > t_) memcpy(doc + at, l_ + t_, (n_ - t_) * sizeof(uint32_t)); at += n_ - t_; p++; } while (0)
        off[p] = 0; old_pid[p] = 0;                                   /* pid 0: every docid in use */
        if (snew->has_used) {
            const uint64_t t0 = so->used_n, n0 = snew->used_n - t0;
            if (pass == 1 && n0) memcpy(doc + at, snew->used_doc + t0, n0 * sizeof(uint32_t));
            at += n0;
        } else {
            if (pass == 1) for (uint32_t d = old_max + 1; d <= snew->max_docid; d++) doc[at + (d - old_max - 1)] = d;
            at += snew->max_docid - old_max;
        }
        p++;
        for (uint32_t w = 1; w <= nm->W; w++)
            TAIL(snew->wp_doc + snew->wp_off[w], snew->wp_off[w + 1] - snew->wp_off[w], w <= m->W ? w : 0xFFFFFFFFu);
        for (uint32_t t = 1; t <= nm->T; t++)
            TAIL(snew->tp_doc + snew->tp_off[t], snew->tp_off[t + 1] - snew->tp_off[t], t <= m->T ? m->W + t : 0xFFFFFFFFu);
        for (int f = 0; f < snew->num_facets && !rc; f++) {
            uint32_t j = 0;                                           /* walks the old field's ids */
            for (uint32_t k = 0; k < snew->nfid[f]; k++) {
                uint32_t op = 0xFFFFFFFFu;
                if (j < so->nfid[f] && so->fid[f][j] == snew->fid[f][k]) op = m->fbase[f] + j++;
                else if (j < so->nfid[f] && so->fid[f][j] < snew->fid[f][k]) { rc = MGPU_E_UNSUPPORTED; break; }   /* an old facet value is gone */
                TAIL(snew->fp_doc[f] + snew->fp_off[f][k], snew->fp_off[f][k + 1] - snew->fp_off[f][k], op);
            }
            if (!rc && j != so->nfid[f]) rc = MGPU_E_UNSUPPORTED;
        }
        for (int i = 0; i < snew->num_bools && !rc; i++)
            for (int v = 0; v < 2; v++) TAIL(snew->bp_doc[i][v], snew->bp_n[i][v], m->bbase + 2u * (uint32_t)i + (uint32_t)v);
#undef TAIL
        off[P1] = at;
        if (pass == 0) cap = at;
    }
    if (!rc) {
        mgpu_shard_delta_t d;
        memset(&d, 0, sizeof(d));
        d.max_docid = snew->max_docid; d.num_postings = P1; d.old_pid = old_pid; d.post_off = off; d.post_doc = doc;
        /* blobs of the new documents: rebased offsets */
        const uint32_t nd = snew->max_docid - old_max;
        uint64_t *ddo = malloc(sizeof(uint64_t) * ((size_t)nd + 1));
        if (!ddo) rc = MGPU_E_LIMIT;
        else {
            const uint64_t base = snew->dd_off[old_max + 1];
            for (uint32_t i = 0; i <= nd; i++) ddo[i] = snew->dd_off[old_max + 1 + i] - base;
            d.dd_off = ddo; d.dd = snew->dd + base;
            for (int i = 0; i < snew->num_numbers; i++) { d.np_n[i] = snew->np_n[i]; d.np_doc[i] = snew->np_doc[i]; }
            rc = mgpu_shard_apply_delta(m->g, &d);
            free(ddo);
        }
    }
    free(old_pid); free(off); free(doc);
    if (!rc && snew->dict && (snew->word_count != so->word_count || !m->gd)) {
        /* words were added: the flat dictionary of the term lookup is exported again */
        uint32_t *chars = NULL, *woff = NULL, *wwid = NULL, nw = 0;
        if (mdict_export_words(snew->dict, &chars, &woff, &wwid, &nw)) rc = MGPU_E_ARG;
        else {
            mgpu_dict_t *gd = NULL;
            rc = mgpu_dict_create(m->device, chars, woff, wwid, nw, &gd);
            free(chars); free(woff); free(wwid);
            if (!rc) { mgpu_dict_destroy(m->gd); m->gd = gd; }
        }
    }
    if (!rc) {
        m->s = snew; m->W = nm->W; m->T = nm->T; m->bbase = nm->bbase; m->num_postings = nm->num_postings;
        memcpy(m->fbase, nm->fbase, sizeof(m->fbase));
    }
    free(nm);
    return rc;
}

void mmirror_destroy(mmirror_t *m) {
    if (!m) return;
    mgpu_dict_destroy(m->gd);
    mgpu_shard_destroy(m->g);
    free(m);
}

/* ------------------------------------------------------------------ arena
 * Everything a batch of plans points to (posting-id lists, wid tables, filter programs, the plan array itself) comes out
 * of one bump allocator that its owner resets between batches: no malloc per term or per plan.  Blocks never move, so
 * pointers stay valid until the reset; a reset that finds several blocks replaces them by one of their total size.
 * The arena also owns the scratch wid table a plan is built with (all_wordids of squery.c:100-105): generation-stamped
 * slots, never cleared. */
typedef struct m_ablk { struct m_ablk *next; size_t cap, used; } m_ablk_t;
struct m_arena {
    m_ablk_t *head;
    uint32_t *hk, *hv, *hg;      /* scratch table: key (wid), value (index into the plan's wid entries), generation */
    uint32_t hcap, gen;
};
#define ABLK_HDR ((sizeof(m_ablk_t) + 15) & ~(size_t)15)

m_arena_t *m_arena_new(void) { return calloc(1, sizeof(m_arena_t)); }

static void *arena_alloc(m_arena_t *a, size_t bytes) {
    bytes = (bytes + 15) & ~(size_t)15;
    m_ablk_t *b = a->head;
    if (!b || b->used + bytes > b->cap) {
        size_t cap = b ? b->cap * 2 : (size_t)1 << 16;
        if (cap < bytes) cap = bytes;
        m_ablk_t *nb = malloc(ABLK_HDR + cap);
        if (!nb) return NULL;
        nb->next = b; nb->cap = cap; nb->used = 0;
        a->head = b = nb;
    }
    void *p = (char *)b + ABLK_HDR + b->used;
    b->used += bytes;
    return p;
}

void m_arena_reset(m_arena_t *a) {
    if (!a || !a->head) return;
    if (!a->head->next) { a->head->used = 0; return; }
    size_t total = 0;
    for (m_ablk_t *b = a->head; b;) { m_ablk_t *n = b->next; total += b->cap; free(b); b = n; }
    a->head = NULL;
    m_ablk_t *nb = malloc(ABLK_HDR + total);
    if (nb) { nb->next = NULL; nb->cap = total; nb->used = 0; a->head = nb; }
}

void m_arena_free(m_arena_t *a) {
    if (!a) return;
    for (m_ablk_t *b = a->head; b;) { m_ablk_t *n = b->next; free(b); b = n; }
    free(a->hk); free(a->hv); free(a->hg);
    free(a);
}

/* scratch table for one plan with at most `n` distinct keys */
static int wt_begin(m_arena_t *a, uint32_t n) {
    uint32_t need = 64;
    while (need < 4 * n) need <<= 1;
    if (need > a->hcap) {
        free(a->hk); free(a->hv); free(a->hg);
        a->hk = malloc(sizeof(uint32_t) * need); a->hv = malloc(sizeof(uint32_t) * need); a->hg = calloc(need, sizeof(uint32_t));
        a->hcap = a->hk && a->hv && a->hg ? need : 0; a->gen = 0;
        if (!a->hcap) return -1;
    }
    if (++a->gen == 0) { memset(a->hg, 0, sizeof(uint32_t) * a->hcap); a->gen = 1; }
    return 0;
}
static inline uint32_t wt_mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; return h; }
/* slot of `wid`: *found says whether it is there; an absent key's slot is where it goes */
static inline uint32_t wt_find(const m_arena_t *a, uint32_t wid, int *found) {
    const uint32_t mask = a->hcap - 1;
    for (uint32_t i = wt_mix(wid) & mask;; i = (i + 1) & mask) {
        if (a->hg[i] != a->gen) { *found = 0; return i; }
        if (a->hk[i] == wid) { *found = 1; return i; }
    }
}

/* ------------------------------------------------------------------ plan */
void mplan_free(mplan_t *pl) {
    m_arena_free(pl->arena);
    memset(pl, 0, sizeof(*pl));
}

typedef struct { mgpu_fop_t *v; int n, cap; } fops_t;
static void fop_push(fops_t *fo, int op, int a, uint64_t lo, uint64_t hi) {
    if (fo->n == fo->cap) return;                            /* cannot happen: cap is the tree's upper bound */
    mgpu_fop_t *f = &fo->v[fo->n++];
    f->op = op; f->a = a; f->lo = lo; f->hi = hi;
}
static int filter_nodes(const m_filter_t *f) {
    int n = 1;
    for (int i = 0; i < f->num_children; i++) n += filter_nodes(f->children[i]);
    return n;
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
static int lower_filter(const mmirror_t *m, const m_filter_t *f, fops_t *pl) {
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


/* One (query, shard) plan from flat term results: everything of lookup_terms after the dictionary (process_termresult,
 * squery.c:72-118; exact wids, squery.c:159-166), the filter leaves, the config.  Term t's dictionary result is
 * {twid[t], (t_wid[t][k], t_dist[t][k]) for k < t_n[t]} (dtrie.h:94-97); exact_in[i] = dtrie_lookup_exact of word i.
 * Everything the plan points to is taken from `A`; *wids_out = the wid table (all_wordids + exact-only entries). */
static int plan_build_flat(const mmirror_t *m, const m_query_t *q, const uint32_t *twid, const uint32_t *const *t_wid,
                           const uint8_t *const *t_dist, const uint32_t *t_n, const uint32_t *exact_in, int capture,
                           m_arena_t *A, mgpu_plan_t *p, mgpu_wid_entry_t **wids_out) {
    memset(p, 0, sizeof(*p));
    if (wids_out) *wids_out = NULL;
    const mshard_t *s = m->s;
    p->num_terms = q->num_terms; p->num_words = q->num_words;
    if (q->num_terms > MGPU_MAX_TERMS || q->num_words > MGPU_MAX_WORDS) return MGPU_E_LIMIT;
    const int nt = q->num_terms, nwords = q->num_words;

    /* upper bound of the distinct wids this plan can name */
    uint32_t ub = (uint32_t)nwords + 1;
    for (int t = 0; t < nt; t++) {
        if (twid[t]) { const uint32_t *w; ub += (uint32_t)mshard_twid_wids(s, twid[t], &w); }
        else ub += t_n[t];
    }
    if (wt_begin(A, ub)) return MGPU_E_LIMIT;
    mgpu_wid_entry_t *W = arena_alloc(A, sizeof(mgpu_wid_entry_t) * ub);
    if (!W) return MGPU_E_LIMIT;
    uint32_t nw = 0;
    /* all_wordids: the LAST term that names a word sets its distance (kh_set, squery.c:56-60), the FIRST is its matching
     * term (get_matching_term walks the terms in order, docrank.c:114-123) */
#define ALL_SET(wid_, dist_, t_) do { int fnd_; const uint32_t sl_ = wt_find(A, (wid_), &fnd_); \
        if (fnd_) W[A->hv[sl_]].dist = (uint8_t)(dist_); \
        else { A->hg[sl_] = A->gen; A->hk[sl_] = (wid_); A->hv[sl_] = nw; \
               W[nw].wid = (wid_); W[nw].dist = (uint8_t)(dist_); W[nw].mterm = (uint8_t)(t_); W[nw].exact_mask = 0; nw++; } } while (0)
    for (int t = 0; t < nt; t++) {
        const uint32_t *docs;
        if (twid[t] == 0 && t_n[t] == 0) continue;
        if (twid[t]) {
            uint32_t *tp = arena_alloc(A, sizeof(uint32_t));
            if (!tp) return MGPU_E_LIMIT;
            p->term_post[t] = tp;
            if (mshard_twid_posting(s, twid[t], &docs)) {
                tp[0] = m->W + twid[t];
                p->term_npost[t] = 1; p->term_nzero[t] = 1;          /* a twid matches with zero typos (squery.c:87-88) */
                const uint32_t *wids;
                const uint64_t n = mshard_twid_wids(s, twid[t], &wids);          /* set_wids_under_twid, squery.c:62-70 */
                for (uint64_t i = 0; i < n; i++) ALL_SET(wids[i], 0, t);
            }
            continue;
        }
        uint32_t *tp = arena_alloc(A, sizeof(uint32_t) * t_n[t]);
        if (!tp) return MGPU_E_LIMIT;
        p->term_post[t] = tp;
        uint32_t np = 0, nz = 0;
        const uint32_t *tw = t_wid[t]; const uint8_t *td = t_dist[t];
        for (uint32_t k = 0; k < t_n[t]; k++) {
            const uint32_t wid = tw[k];
            if (!wid || !mshard_wid_posting(s, wid, &docs)) continue;
            ALL_SET(wid, td[k], t);
            tp[np++] = wid;
            if (td[k] == 0) { const uint32_t x = tp[nz]; tp[nz] = wid; tp[np - 1] = x; nz++; }   /* zero typos: kept in front (tzero_typo_bmap) */
        }
        p->term_npost[t] = np; p->term_nzero[t] = nz;
    }
#undef ALL_SET
    /* exact wids (squery.c:159-166): bits on the entries they share with all_wordids, entries of their own otherwise */
    for (int i = 0; i < nwords; i++) {
        const uint32_t *docs;
        const uint32_t wid = exact_in[i];
        if (!wid || !mshard_wid_posting(s, wid, &docs)) continue;
        p->exact_present |= (uint16_t)(1u << i);
        int fnd; const uint32_t sl = wt_find(A, wid, &fnd);
        if (fnd) { W[A->hv[sl]].exact_mask |= (uint16_t)(1u << i); continue; }
        A->hg[sl] = A->gen; A->hk[sl] = wid; A->hv[sl] = nw;
        W[nw].wid = wid; W[nw].dist = 0xFF; W[nw].mterm = 0xFF; W[nw].exact_mask = (uint16_t)(1u << i); nw++;
    }
    p->num_wids = nw; p->wids = W;
    if (wids_out) *wids_out = W;

    /* filter */
    if (q->filter) {
        fops_t fo; fo.n = 0; fo.cap = 4 * filter_nodes(q->filter) + 4;
        fo.v = arena_alloc(A, sizeof(mgpu_fop_t) * (size_t)fo.cap);
        if (!fo.v) return MGPU_E_LIMIT;
        p->has_filter = 1;
        p->filter_null = lower_filter(m, q->filter, &fo) ? 0 : 1;
        p->fops = fo.v; p->num_fops = fo.n;
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

/* Everything of the plan after the trie walk, from the reference's own result type: tr[t] = dtrie_lookup_term's result
 * for term t (a khash of wid -> distance there, an m_u32map here); exact_in[i] = dtrie_lookup_exact's result for query
 * word i (0 = none).  q->terms / q->words themselves are not read, so a caller whose dictionary is not an mdict (the
 * reference's dtrie: shim/squery_gpu.c) can use it as is.  tr is not modified. */
int mplan_from_terms(const mmirror_t *m, const m_query_t *q, m_termresult_t *tr, const uint32_t *exact_in,
                     int capture, mplan_t *pl) {
    memset(pl, 0, sizeof(*pl));
    if (q->num_terms > MGPU_MAX_TERMS || q->num_words > MGPU_MAX_WORDS) return MGPU_E_LIMIT;
    pl->arena = m_arena_new();
    if (!pl->arena) return MGPU_E_LIMIT;
    uint32_t twid[M_MAX_QTERMS], t_n[M_MAX_QTERMS];
    const uint32_t *t_wid[M_MAX_QTERMS]; const uint8_t *t_dist[M_MAX_QTERMS];
    for (int t = 0; t < q->num_terms; t++) {
        const m_u32map_t *mp = &tr[t].wordids;
        twid[t] = tr[t].twid; t_n[t] = 0; t_wid[t] = NULL; t_dist[t] = NULL;
        if (twid[t] || !mp->n) continue;
        uint32_t *w = arena_alloc(pl->arena, sizeof(uint32_t) * mp->n);
        uint8_t *d = arena_alloc(pl->arena, mp->n);
        if (!w || !d) return MGPU_E_LIMIT;
        uint32_t n = 0;
        for (uint32_t k = 0; k < mp->cap; k++) if (mp->keys[k]) { w[n] = mp->keys[k]; d[n] = (uint8_t)mp->vals[k]; n++; }
        t_n[t] = n; t_wid[t] = w; t_dist[t] = d;
    }
    return plan_build_flat(m, q, twid, t_wid, t_dist, t_n, exact_in, capture, pl->arena, &pl->p, &pl->wids);
}

/* lookup_terms (squery.c:144-167) with the host dictionary, then the plan */
int mplan_build(const mmirror_t *m, const m_query_t *q, int capture, mplan_t *pl) {
    if (q->num_terms > MGPU_MAX_TERMS || q->num_words > MGPU_MAX_WORDS) { memset(pl, 0, sizeof(*pl)); return MGPU_E_LIMIT; }
    const mshard_t *s = m->s;
    m_termresult_t tr[M_MAX_QTERMS];
    uint32_t exact_wid[M_MAX_QWORDS];
    for (int t = 0; t < q->num_terms; t++) mdict_lookup_term(s->dict, &q->terms[t], &tr[t]);
    for (int i = 0; i < q->num_words; i++) exact_wid[i] = mdict_lookup_exact(s->dict, &q->words[i]);
    int rc = mplan_from_terms(m, q, tr, exact_wid, capture, pl);
    for (int t = 0; t < q->num_terms; t++) m_termresult_free(&tr[t]);
    return rc;
}

/* ------------------------------------------------------------------ execution */
int msq_plan_threads = 1;
void msq_set_plan_threads(int n) { msq_plan_threads = n; }
mgpu_shard_t *mmirror_gpu(mmirror_t *m) { return m ? m->g : NULL; }

/* ---- resolved queries: the dictionary half of lookup_terms for nq queries x num_shards shards, kept flat.
 * Pair i = (query i / num_shards, shard i % num_shards); its term t is entry i * M_MAX_QTERMS + t. */
struct msq_resolved {
    int n, nq, num_shards;
    mmirror_t *const *mirrors; const m_query_t **queries;
    uint32_t *twid;            /* [n * M_MAX_QTERMS] */
    uint32_t *t_off;           /* [n * M_MAX_QTERMS + 1] into wid / dist */
    uint32_t *wid; uint8_t *dist;
    uint32_t *exact;           /* [n * M_MAX_QWORDS] */
    size_t nwd;
};

void msq_resolved_free(msq_resolved_t *r) {
    if (!r) return;
    free(r->queries); free(r->twid); free(r->t_off); free(r->wid); free(r->dist); free(r->exact); free(r);
}
int msq_resolved_count(const msq_resolved_t *r) { return r ? r->nq : 0; }

struct resolve_job { mmirror_t *const *mirrors; int num_shards; const m_query_t *const *queries; m_termresult_t *tr; const uint8_t *have;
                     uint32_t *exact; int lo, hi; };
static void *resolve_worker(void *arg) {
    struct resolve_job *j = arg;
    for (int i = j->lo; i < j->hi; i++) {
        const m_query_t *q = j->queries[i / j->num_shards];
        const mshard_t *s = j->mirrors[i % j->num_shards]->s;
        if (q->num_terms > M_MAX_QTERMS || q->num_words > M_MAX_QWORDS) continue;
        for (int t = 0; t < q->num_terms; t++)
            if (!j->have[(size_t)i * M_MAX_QTERMS + t]) mdict_lookup_term(s->dict, &q->terms[t], &j->tr[(size_t)i * M_MAX_QTERMS + t]);
        for (int w = 0; w < q->num_words; w++) j->exact[(size_t)i * M_MAX_QWORDS + w] = mdict_lookup_exact(s->dict, &q->words[w]);
    }
    return NULL;
}

/* dtrie_lookup_term for every term and dtrie_lookup_exact for every word of every (query, shard) pair.  ctx != NULL: the
 * typo walks (dtrie.c:763-791) run on the device, ONE mgpu_lookup_terms call for the whole set (row f-1; ctx must be idle
 * and on the mirrors' device); the exact descents stay on the host (msq_plan_threads threads). */
static int resolve_batch_impl(mgpu_ctx_t *ctx, mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries, int nq,
                              int nthreads, msq_resolved_t **out) {
    *out = NULL;
    if (nq < 0 || num_shards < 1) return MGPU_E_ARG;
    const int n = nq * num_shards;
    for (int i = 0; i < nq; i++) if (queries[i]->num_terms > M_MAX_QTERMS || queries[i]->num_words > M_MAX_QWORDS) return MGPU_E_LIMIT;
    msq_resolved_t *r = calloc(1, sizeof(*r));
    m_termresult_t *tr = calloc((size_t)(n ? n : 1) * M_MAX_QTERMS, sizeof(*tr));
    uint8_t *have = calloc((size_t)(n ? n : 1) * M_MAX_QTERMS, 1);
    if (!r || !tr || !have) { free(r); free(tr); free(have); return MGPU_E_LIMIT; }
    r->n = n; r->nq = nq; r->num_shards = num_shards; r->mirrors = mirrors;
    r->queries = calloc((size_t)(nq ? nq : 1), sizeof(*r->queries));
    r->twid = calloc((size_t)(n ? n : 1) * M_MAX_QTERMS, sizeof(uint32_t));
    r->t_off = calloc((size_t)(n ? n : 1) * M_MAX_QTERMS + 1, sizeof(uint32_t));
    r->exact = calloc((size_t)(n ? n : 1) * M_MAX_QWORDS, sizeof(uint32_t));
    if (!r->queries || !r->twid || !r->t_off || !r->exact) { msq_resolved_free(r); free(tr); free(have); return MGPU_E_LIMIT; }
    memcpy(r->queries, queries, (size_t)nq * sizeof(*r->queries));
    int rc = 0;
    static int trace = -1;
    if (trace < 0) trace = getenv("MSQ_PREP_TRACE") != NULL;
    struct timespec ts0, ts1, ts2;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    size_t nl = 0, nhits_total = 0;
    /* device lookups: run l (dv_off[l] .. dv_off[l + 1]) of dv_wid / dv_dist is the result of term slot dv_slot_of[l] */
    uint32_t *dv_off = NULL, *dv_wid = NULL, *dv_slot_of = NULL; uint8_t *dv_dist = NULL; size_t dv_n = 0;
    struct resolve_job jobs[64]; pthread_t th[64]; int nworkers = 0, spawned = 0;
    if (ctx && n) {
        /* every term that dtrie_lookup_term would walk the trie for (mode 2) becomes one device lookup */
        for (int i = 0; i < n; i++) {
            const m_query_t *q = queries[i / num_shards];
            if (!mirrors[i % num_shards]->gd) continue;
            for (int t = 0; t < q->num_terms; t++)
                if (mdict_lookup_mode(&q->terms[t]) == 2 && q->terms[t].word.length >= 1 && q->terms[t].word.length <= MGPU_MAX_TERM_LEN) nl++;
        }
        mgpu_term_lookup_t *lk = malloc(sizeof(*lk) * (nl ? nl : 1));
        uint32_t *slot = malloc(sizeof(uint32_t) * (nl ? nl : 1));          /* lookup -> index into tr[] */
        dv_slot_of = malloc(sizeof(uint32_t) * (nl ? nl : 1));
        if (!lk || !slot || !dv_slot_of) rc = MGPU_E_LIMIT;
        nl = 0;
        for (int i = 0; i < n && !rc; i++) {
            const m_query_t *q = queries[i / num_shards];
            if (!mirrors[i % num_shards]->gd) continue;
            for (int t = 0; t < q->num_terms; t++) {
                const m_term_t *tm = &q->terms[t];
                if (mdict_lookup_mode(tm) != 2 || tm->word.length < 1 || tm->word.length > MGPU_MAX_TERM_LEN) continue;
                lk[nl].dict = mirrors[i % num_shards]->gd; lk[nl].chars = tm->word.chars; lk[nl].len = tm->word.length; lk[nl].prefix = tm->prefix;
                slot[nl] = (uint32_t)((size_t)i * M_MAX_QTERMS + (size_t)t);
                have[slot[nl]] = 1;                                      /* an empty result is a result too */
                nl++;
            }
        }
        /* The hits of the device ({lookup, wid, distance}, any order) become one (wid, distance) run per lookup: counted
         * per lookup, placed, then each short run ordered by wid with the least distance kept (add_wordid_to_result,
         * dtrie.c:633-640) -- no hash map and no malloc per term.  At most 65535 lookups per call. */
        if (!rc && !(dv_off = calloc(nl + 2, sizeof(uint32_t)))) rc = MGPU_E_LIMIT;
        /* the host's share (exact descents, twid terms: every slot without `have`) runs in worker threads WHILE the
         * device scores the typo walks */
        if (!rc) {
            nworkers = nthreads < 1 ? 1 : nthreads > 64 ? 64 : nthreads;
            if (nworkers > n) nworkers = n;
            for (int t = 0; t < nworkers; t++) {
                jobs[t] = (struct resolve_job){ mirrors, num_shards, queries, tr, have, r->exact,
                                                (int)((long)n * t / nworkers), (int)((long)n * (t + 1) / nworkers) };
                if (pthread_create(&th[t], NULL, resolve_worker, &jobs[t])) { nworkers = t; rc = MGPU_E_LIMIT; break; }
            }
            spawned = 1;
        }
        for (size_t lo = 0; lo < nl && !rc; lo += 65535) {
            const int cnt = (int)(nl - lo < 65535 ? nl - lo : 65535);
            mgpu_term_hit_t *hits = NULL; uint32_t nh = 0;
            rc = mgpu_lookup_terms(ctx, lk + lo, cnt, &hits, &nh);
            if (rc) break;
            uint32_t *nw = realloc(dv_wid, sizeof(uint32_t) * (dv_n + nh + 1));
            if (nw) dv_wid = nw;
            uint8_t *nd = realloc(dv_dist, dv_n + nh + 1);
            if (nd) dv_dist = nd;
            if (!nw || !nd) { rc = MGPU_E_LIMIT; mgpu_term_hits_free(hits); break; }
            uint32_t *o = dv_off + lo + 1;                       /* o[l] = end of lookup l's run, then its fill position */
            for (uint32_t h = 0; h < nh && !rc; h++) { if (hits[h].lookup >= (uint32_t)cnt) rc = MGPU_E_HIP; else o[hits[h].lookup]++; }
            if (rc) { mgpu_term_hits_free(hits); break; }
            uint32_t at = (uint32_t)dv_n;
            for (int l = 0; l < cnt; l++) { const uint32_t c2 = o[l]; o[l] = at; at += c2; }
            for (uint32_t h = 0; h < nh; h++) { const uint32_t x = o[hits[h].lookup]++; dv_wid[x] = hits[h].wid; dv_dist[x] = (uint8_t)hits[h].dist; }
            /* now o[l] = end of run l; its start is the end of run l - 1 (or the old dv_n) */
            uint32_t w = (uint32_t)dv_n, b = (uint32_t)dv_n;
            for (int l = 0; l < cnt; l++) {
                const uint32_t e = o[l], w0 = w;
                for (uint32_t x = b; x < e; x++) {               /* insertion into [w0, w), ordered by wid */
                    const uint32_t wd = dv_wid[x]; const uint8_t ds = dv_dist[x];
                    uint32_t y = w;
                    while (y > w0 && dv_wid[y - 1] > wd) y--;
                    if (y > w0 && dv_wid[y - 1] == wd) { if (ds < dv_dist[y - 1]) dv_dist[y - 1] = ds; continue; }
                    for (uint32_t z = w; z > y; z--) { dv_wid[z] = dv_wid[z - 1]; dv_dist[z] = dv_dist[z - 1]; }
                    dv_wid[y] = wd; dv_dist[y] = ds; w++;
                }
                b = e; o[l] = w;
            }
            dv_n = w;
            nhits_total += nh;
            mgpu_term_hits_free(hits);
        }
        /* dv_off[l + 1] holds the end of run l; entry 0 its start */
        if (!rc) for (size_t l = 0; l < nl; l++) dv_slot_of[l] = slot[l];
        free(lk); free(slot);
    }
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    if (spawned) for (int t = 0; t < nworkers; t++) pthread_join(th[t], NULL);
    if (!rc && n) {
        if (!spawned) {
            if (nthreads > n) nthreads = n;
            if (nthreads < 1) nthreads = 1;
            if (nthreads > 64) nthreads = 64;
            for (int t = 0; t < nthreads; t++) {
                jobs[t] = (struct resolve_job){ mirrors, num_shards, queries, tr, have, r->exact,
                                                (int)((long)n * t / nthreads), (int)((long)n * (t + 1) / nthreads) };
            }
            if (nthreads == 1) resolve_worker(&jobs[0]);
            else {
                int made = 0;
                while (made < nthreads && !pthread_create(&th[made], NULL, resolve_worker, &jobs[made])) made++;
                for (int t = made; t < nthreads; t++) resolve_worker(&jobs[t]);          /* no thread to be had: here and now */
                for (int t = 0; t < made; t++) pthread_join(th[t], NULL);
            }
        }
        /* flatten: host results from their maps, device results from their runs (lookups were made in slot order) */
        size_t tot = dv_n;
        for (size_t k = 0; k < (size_t)n * M_MAX_QTERMS; k++) tot += tr[k].wordids.n;
        if (tot >= 0xFFFFFFFFull) rc = MGPU_E_LIMIT;
        else {
            r->wid = malloc(sizeof(uint32_t) * (tot ? tot : 1)); r->dist = malloc(tot ? tot : 1);
            if (!r->wid || !r->dist) rc = MGPU_E_LIMIT;
            size_t at = 0, l = 0;
            for (size_t k = 0; k < (size_t)n * M_MAX_QTERMS && !rc; k++) {
                r->t_off[k] = (uint32_t)at;
                r->twid[k] = tr[k].twid;
                if (l < nl && dv_slot_of && dv_slot_of[l] == k) {
                    const uint32_t a = dv_off[l], b = dv_off[l + 1];
                    memcpy(r->wid + at, dv_wid + a, sizeof(uint32_t) * (size_t)(b - a)); memcpy(r->dist + at, dv_dist + a, (size_t)(b - a));
                    at += b - a; l++;
                    continue;
                }
                const m_u32map_t *mp = &tr[k].wordids;
                for (uint32_t x = 0; x < mp->cap; x++) if (mp->keys[x]) { r->wid[at] = mp->keys[x]; r->dist[at] = (uint8_t)mp->vals[x]; at++; }
            }
            if (!rc) { r->t_off[(size_t)n * M_MAX_QTERMS] = (uint32_t)at; r->nwd = at; }
        }
    }
    if (trace) {
        clock_gettime(CLOCK_MONOTONIC, &ts2);
        fprintf(stderr, "[resolve] %d pairs, %zu device lookups, %zu hits: device scan + maps %.3f ms, host descents + flatten %.3f ms\n", n, nl, nhits_total,
                (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6, (ts2.tv_sec - ts1.tv_sec) * 1e3 + (ts2.tv_nsec - ts1.tv_nsec) * 1e-6);
    }
    for (size_t k = 0; k < (size_t)(n ? n : 1) * M_MAX_QTERMS; k++) if (tr[k].wordids.cap) m_termresult_free(&tr[k]);
    free(tr); free(have);
    free(dv_off); free(dv_wid); free(dv_dist); free(dv_slot_of);
    if (rc) { msq_resolved_free(r); return rc; }
    *out = r;
    return 0;
}
int msq_resolve_batch(mgpu_ctx_t *ctx, mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries, int nq,
                      msq_resolved_t **out) {
    return resolve_batch_impl(ctx, mirrors, num_shards, queries, nq, msq_plan_threads, out);
}

/* ---- prepared batches: the plans of a resolved set, all in one arena */
struct msq_prepared {
    int n, nq, num_shards;
    mmirror_t *const *mirrors; const m_query_t **queries;      /* borrowed from the resolved set when `borrowed` */
    mgpu_plan_t *gp; mgpu_shard_t **gs;
    m_arena_t *arena;
    int borrowed;
};

msq_prepared_t *msq_prepared_new(void) {
    msq_prepared_t *pp = calloc(1, sizeof(*pp));
    if (pp && !(pp->arena = m_arena_new())) { free(pp); pp = NULL; }
    return pp;
}

/* process_termresult + exact wids + filter leaves + config for every pair of `r`, written into pp (whose previous
 * contents are dropped; its arena is reused).  Single-threaded: the serving loop runs one of these per host thread. */
int msq_plan_resolved(const msq_resolved_t *r, int capture, msq_prepared_t *pp) {
    if (!r || !pp) return MGPU_E_ARG;
    m_arena_reset(pp->arena);
    if (!pp->borrowed) free(pp->queries);
    pp->n = r->n; pp->nq = r->nq; pp->num_shards = r->num_shards; pp->mirrors = r->mirrors;
    pp->queries = r->queries; pp->borrowed = 1;
    const int n = r->n;
    pp->gp = arena_alloc(pp->arena, sizeof(mgpu_plan_t) * (size_t)(n ? n : 1));
    pp->gs = arena_alloc(pp->arena, sizeof(mgpu_shard_t *) * (size_t)(n ? n : 1));
    if (!pp->gp || !pp->gs) return MGPU_E_LIMIT;
    int rc = 0;
    for (int i = 0; i < n; i++) {
        const m_query_t *q = r->queries[i / r->num_shards];
        const mmirror_t *m = r->mirrors[i % r->num_shards];
        const uint32_t *t_wid[M_MAX_QTERMS]; const uint8_t *t_dist[M_MAX_QTERMS]; uint32_t t_n[M_MAX_QTERMS];
        const size_t b = (size_t)i * M_MAX_QTERMS;
        for (int t = 0; t < q->num_terms; t++) {
            t_wid[t] = r->wid + r->t_off[b + t]; t_dist[t] = r->dist + r->t_off[b + t];
            t_n[t] = r->t_off[b + t + 1] - r->t_off[b + t];
        }
        const int r1 = plan_build_flat(m, q, r->twid + b, t_wid, t_dist, t_n, r->exact + (size_t)i * M_MAX_QWORDS, capture,
                                       pp->arena, &pp->gp[i], NULL);
        if (r1) rc = r1;
        pp->gs[i] = m->g;
    }
    return rc;
}

/* test / inspection access: plan i of a prepared batch; term t of pair i of a resolved set */
const mgpu_plan_t *msq_prepared_plan(const msq_prepared_t *pp, int i) { return (pp && i >= 0 && i < pp->n) ? &pp->gp[i] : NULL; }
uint32_t msq_resolved_term(const msq_resolved_t *r, int i, int t, uint32_t *twid, const uint32_t **wid, const uint8_t **dist) {
    if (!r || i < 0 || i >= r->n || t < 0 || t >= M_MAX_QTERMS) return 0;
    const size_t k = (size_t)i * M_MAX_QTERMS + (size_t)t;
    if (twid) *twid = r->twid[k];
    if (wid) *wid = r->wid + r->t_off[k];
    if (dist) *dist = r->dist + r->t_off[k];
    return r->t_off[k + 1] - r->t_off[k];
}
uint32_t msq_resolved_exact(const msq_resolved_t *r, int i, int w) {
    return (r && i >= 0 && i < r->n && w >= 0 && w < M_MAX_QWORDS) ? r->exact[(size_t)i * M_MAX_QWORDS + w] : 0;
}

static int prepare_common(mgpu_ctx_t *ctx, mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries,
                          int nq, int capture, msq_prepared_t **out) {
    *out = NULL;
    msq_resolved_t *r = NULL;
    int rc = msq_resolve_batch(ctx, mirrors, num_shards, queries, nq, &r);
    if (rc) return rc;
    msq_prepared_t *pp = msq_prepared_new();
    if (!pp) { msq_resolved_free(r); return MGPU_E_LIMIT; }
    rc = msq_plan_resolved(r, capture, pp);
    /* the prepared batch outlives the resolved set: it keeps its own copy of the query pointers */
    pp->queries = r->queries; r->queries = NULL; pp->borrowed = 0;
    msq_resolved_free(r);
    if (rc) { msq_prepared_free(pp); return rc; }
    *out = pp;
    return 0;
}

/* lookup_terms (squery.c:144-167) + filter leaf resolution for every (query, shard) pair */
int msq_prepare_batch(mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries,
                      int nq, int capture, msq_prepared_t **out) {
    return prepare_common(NULL, mirrors, num_shards, queries, nq, capture, out);
}

int msq_prepare_batch_dev(mgpu_ctx_t *ctx, mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries,
                          int nq, int capture, msq_prepared_t **out) {
    return prepare_common(ctx, mirrors, num_shards, queries, nq, capture, out);
}

void msq_prepared_free(msq_prepared_t *pp) {
    if (!pp) return;
    m_arena_free(pp->arena);
    if (!pp->borrowed) free(pp->queries);
    free(pp);
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

/* the same for n pairs in ONE batch of launches: every shard of a query, or of several queries (a replacement
 * execute_query, shim/query_gpu.c); all mirrors on the context's device.  outs[i] is filled on success only. */
int msq_execute_plans(mgpu_ctx_t *ctx, const mmirror_t *const *ms, const m_query_t *const *qs, const mplan_t *const *pls,
                      const int *shard_idx, int n, m_sqresult_t *outs) {
    if (!ctx || n <= 0) return MGPU_E_ARG;
    mgpu_sqout_t *g = calloc((size_t)n, sizeof(*g));
    mgpu_shard_t **gs = malloc(sizeof(*gs) * (size_t)n);
    mgpu_plan_t *gp = malloc(sizeof(*gp) * (size_t)n);
    if (!g || !gs || !gp) { free(g); free(gs); free(gp); return MGPU_E_LIMIT; }
    for (int i = 0; i < n; i++) { gs[i] = ms[i]->g; gp[i] = pls[i]->p; memset(&outs[i], 0, sizeof(outs[i])); }
    int rc = mgpu_execute(ctx, gs, gp, n, g);
    for (int i = 0; i < n && !rc; i++) {
        rc = sqout_to_result(qs[i], shard_idx[i], ms[i]->s->num_facets, &g[i], &outs[i]);
        if (rc) for (int k = 0; k <= i; k++) m_sqresult_free(&outs[k]);
    }
    for (int i = 0; i < n; i++) mgpu_sqout_free(&g[i]);
    free(g); free(gs); free(gp);
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
    /* the current run: prepared batches (preps), or resolved sets whose plans every step builds first (res) */
    const msq_prepared_t *const *preps; int nprep, first, count, every;
    const msq_resolved_t *const *res;
    /* ... or raw queries (q_all: q_nb batches of q_batch queries): every step also does the dictionary half, with the
     * typo walks on the step's own context (row f-1), before it builds its plans */
    const m_query_t *const *q_all; int q_nb, q_batch, q_ns, q_threads; mmirror_t *const *q_mirrors;
    msq_resolved_t *slot_res[128];        /* per context: the resolved set behind slot_pp (query runs) */
    double lookup_ms[64];                 /* per thread: wall time of the dictionary half in the last run */
    msq_prepared_t *slot_pp[128];         /* per context: the plans of the batch it has in flight (resolved runs) */
    double plan_ms[64];                   /* per thread: wall time spent building plans in the last run */
    long hits[64]; int rc[64];
    char err[256];
    /* several GPUs: this process owns `gl_shards` of the index's shards; after a batch its result slots
     * are all-gathered over RCCL (mgpu_comm_*, issued from these threads in step order) and it merges its share of the
     * batch's queries: query rank, rank + world, ... */
    mgpu_comm_t *comm; int g_k, g_nf, g_fmax, g_nagg, g_pairs; size_t g_slot_b;
    uint8_t *g_send[128], *g_recv[128];   /* per context: pinned slot buffers */
    pthread_mutex_t g_mu; pthread_cond_t g_cv; long g_turn; int g_fail;
};
typedef struct { msq_pipeline_t *p; int t; } pipe_arg_t;

static void pipe_fail(msq_pipeline_t *p, int t, int rc) {
    /* mgpu_last_error is per thread: keep this thread's message for the caller */
    pthread_mutex_lock(&p->mu);
    if (!p->err[0]) snprintf(p->err, sizeof(p->err), "%s", mgpu_last_error());
    pthread_mutex_unlock(&p->mu);
    p->rc[t] = rc;
}

/* the cross-GPU step of batch `step` (context slot ci): slots packed, all-gathered in step order, this rank's share of
 * the queries unpacked and merged.  Returns the share's summed totalHits through *hits. */
static int pipe_gather(msq_pipeline_t *p, int ci, long step, const msq_prepared_t *pp, m_sqresult_t *outs, long *hits) {
    const int world = mgpu_comm_world(p->comm), rank = mgpu_comm_rank(p->comm);
    if (pp->n != p->g_pairs) return MGPU_E_ARG;
    for (int i = 0; i < pp->n; i++) for (int a = 0; a < outs[i].num_aggs; a++)
        if (outs[i].aggs[a].type == MA_CARDINALITY) return MGPU_E_UNSUPPORTED;      /* variable-length id sets: marlin_amd/dist.py's second gather */
    m_slot_pack_batch(outs, pp->n, p->g_k, p->g_nf, p->g_fmax, p->g_nagg, p->g_send[ci]);
    /* every rank issues its gathers in step order, whatever its threads do */
    pthread_mutex_lock(&p->g_mu);
    while (p->g_turn != step && !p->g_fail) pthread_cond_wait(&p->g_cv, &p->g_mu);
    int rc = p->g_fail ? MGPU_E_HIP : mgpu_comm_allgather_begin(p->comm, ci, p->g_send[ci], (size_t)pp->n * p->g_slot_b, p->g_recv[ci]);
    if (rc) p->g_fail = 1; else p->g_turn = step + 1;
    pthread_cond_broadcast(&p->g_cv);
    pthread_mutex_unlock(&p->g_mu);
    if (rc) return rc;
    if ((rc = mgpu_comm_allgather_end(p->comm, ci))) return rc;
    const int total = world * pp->num_shards;
    int mine = 0;
    for (int q = rank; q < pp->nq; q += world) mine++;
    if (!mine) { *hits = 0; return 0; }
    m_sqresult_t *sq = calloc((size_t)mine * (size_t)total, sizeof(*sq));
    const m_query_t **qs = malloc(sizeof(*qs) * (size_t)mine);
    if (!sq || !qs) { free(sq); free(qs); return MGPU_E_LIMIT; }
    m_slot_unpack_gathered_subset(p->g_recv[ci], world, pp->nq, pp->num_shards, p->g_k, p->g_nf, p->g_fmax, p->g_nagg, rank, world, sq);
    for (int i = 0; i < mine; i++) qs[i] = pp->queries[rank + i * world];
    *hits = m_form_result_batch(qs, sq, mine, total, NULL, 0);
    m_sqresult_free_batch(sq, mine * total);
    free(sq); free(qs);
    return 0;
}

/* wait for the batch in flight on ctx, merge it (form_result) */
static int pipe_collect(msq_pipeline_t *p, int t, int ci, long step, mgpu_ctx_t *ctx, const msq_prepared_t *pp, pipe_trace_t *tr) {
    m_sqresult_t *outs = calloc((size_t)(pp->n ? pp->n : 1), sizeof(m_sqresult_t));
    if (!outs) { p->rc[t] = -1; return -1; }
    if (tr) tr->col0 = pipe_now() - p->t_run0;
    int rc = msq_collect_prepared(ctx, pp, outs);
    if (tr) { tr->col1 = pipe_now() - p->t_run0; mgpu_last_timing(ctx, &tr->gpu_ms, NULL); }
    if (!rc && p->comm && step >= 0) {
        long h = 0;
        rc = pipe_gather(p, ci, step, pp, outs, &h);
        if (!rc) p->hits[t] += h;
    } else if (!rc && !p->comm) p->hits[t] += m_form_result_batch((const m_query_t *const *)pp->queries, outs, pp->nq, pp->num_shards, NULL, 0);
    if (rc && p->comm) {                                      /* nobody may wait for a turn that will not come */
        pthread_mutex_lock(&p->g_mu); p->g_fail = 1; pthread_cond_broadcast(&p->g_cv); pthread_mutex_unlock(&p->g_mu);
    }
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

/* the batch of step i on context slot k of thread t: a prepared one, or plans built here and now from a resolved set
 * (lookup_terms after the dictionary: process_termresult, squery.c:72-118) */
static const msq_prepared_t *pipe_batch(msq_pipeline_t *p, int t, int ci, int i, int *rc) {
    *rc = 0;
    if (!p->res && !p->q_all) return p->preps[(p->first + i) % p->nprep];
    if (!p->slot_pp[ci] && !(p->slot_pp[ci] = msq_prepared_new())) { *rc = MGPU_E_LIMIT; return NULL; }
    const msq_resolved_t *r;
    if (p->q_all) {
        /* the context is idle here: its previous batch was collected before the slot came round again */
        const double tl = pipe_now();
        msq_resolved_free(p->slot_res[ci]); p->slot_res[ci] = NULL;
        *rc = resolve_batch_impl(p->ctx[ci], p->q_mirrors, p->q_ns, p->q_all + (size_t)((p->first + i) % p->q_nb) * (size_t)p->q_batch,
                                 p->q_batch, p->q_threads > 0 ? p->q_threads : 1, &p->slot_res[ci]);
        p->lookup_ms[t] += pipe_now() - tl;
        if (*rc) return NULL;
        r = p->slot_res[ci];
    } else r = p->res[(p->first + i) % p->nprep];
    const double t0 = pipe_now();
    *rc = msq_plan_resolved(r, 0, p->slot_pp[ci]);
    p->plan_ms[t] += pipe_now() - t0;
    return *rc ? NULL : p->slot_pp[ci];
}

static void pipe_steps(msq_pipeline_t *p, int t) {
    p->hits[t] = 0; p->rc[t] = 0; p->plan_ms[t] = 0; p->lookup_ms[t] = 0;
    mgpu_ctx_t *cx[2] = { p->ctx[2 * t], (2 * t + 1 < p->nctx) ? p->ctx[2 * t + 1] : NULL };
    const int ncx = cx[1] ? 2 : 1;
    if (p->every) {
        /* warm-up run: every context of this thread takes every batch once */
        for (int k = 0; k < ncx; k++)
            for (int i = 0; i < p->count; i++) {
                int rc;
                const msq_prepared_t *pp = pipe_batch(p, t, 2 * t + k, i, &rc);
                if (!rc) rc = msq_submit_prepared(cx[k], pp);
                if (rc) { pipe_fail(p, t, rc); return; }
                if (pipe_collect(p, t, 2 * t + k, -1, cx[k], pp, NULL)) return;       /* warm-up: no exchange */
            }
        return;
    }
    /* steps t, t + nthreads, ...: the next step is submitted before the current one is collected */
    const msq_prepared_t *inflight[2] = { NULL, NULL };
    long inflight_step[2] = { -1, -1 };
    pipe_trace_t tr[2];
    int slot = 0;
    for (int i = t; i < p->count || inflight[0] || inflight[1]; i += p->nthreads) {
        if (i < p->count) {
            tr[slot].step = i; tr[slot].t = t; tr[slot].sub0 = pipe_now() - p->t_run0;
            int rc;
            const msq_prepared_t *pp = pipe_batch(p, t, 2 * t + slot, i, &rc);
            if (!rc) rc = msq_submit_prepared(cx[slot], pp);
            tr[slot].sub1 = pipe_now() - p->t_run0;
            if (rc) { pipe_fail(p, t, rc); break; }
            inflight[slot] = pp; inflight_step[slot] = i;
        }
        /* collect the older batch: the other slot when there are two, else this one */
        const int older = ncx == 2 ? slot ^ 1 : slot;
        if (inflight[older]) {
            const msq_prepared_t *pp = inflight[older];
            inflight[older] = NULL;
            if (pipe_collect(p, t, 2 * t + older, inflight_step[older], cx[older], pp, p->trace_on ? &tr[older] : NULL)) break;
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
    pthread_mutex_init(&p->g_mu, NULL); pthread_cond_init(&p->g_cv, NULL);
    for (int t = 0; t < nctx; t++) p->ctx[t] = ctxs[t];
    for (int t = 0; t < nthreads; t++) {
        pipe_arg_t *a = malloc(sizeof(*a));
        a->p = p; a->t = t;
        if (pthread_create(&p->th[t], NULL, pipe_thread, a)) { free(a); p->nthreads = t; msq_pipeline_destroy(p); return -1; }
    }
    *out = p;
    return 0;
}

static int pipe_dispatch(msq_pipeline_t *p, const msq_prepared_t *const *preps, const msq_resolved_t *const *res, int nprep,
                         int first, int count, int every, long *hits_out) {
    if (!p || (!preps && !res && !p->q_all) || nprep < 1 || count < 0) return MGPU_E_ARG;
    pthread_mutex_lock(&p->mu);
    p->preps = preps; p->res = res; p->nprep = nprep; p->first = first; p->count = count; p->every = every;
    p->err[0] = 0;
    p->ntrace = 0; p->t_run0 = pipe_now();
    p->g_turn = 0; p->g_fail = 0;
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
    return pipe_dispatch(p, preps, NULL, nprep, first, count, 0, hits_out);
}

int msq_pipeline_warm(msq_pipeline_t *p, const msq_prepared_t *const *preps, int nprep) {
    return pipe_dispatch(p, preps, NULL, nprep, 0, nprep, 1, NULL);
}

/* The same loop from resolved sets: step i FIRST builds the plans of res[(first + i) % nres] (process_termresult for
 * every (query, shard) of the batch, in the thread that then submits it), so that everything of execute_squery after
 * the dictionary lookups happens inside the run. */
int msq_pipeline_run_resolved(msq_pipeline_t *p, const msq_resolved_t *const *res, int nres, int first, int count, long *hits_out) {
    return pipe_dispatch(p, NULL, res, nres, first, count, 0, hits_out);
}
int msq_pipeline_warm_resolved(msq_pipeline_t *p, const msq_resolved_t *const *res, int nres) {
    return pipe_dispatch(p, NULL, res, nres, 0, nres, 1, NULL);
}
/* The same loop from the queries themselves: step i takes queries [(first + i) % nb * batch, + batch) through the WHOLE of
 * lookup_terms (squery.c:144-167) -- dictionary lookups with the typo walks on the step's own context (mgpu_lookup_terms),
 * exact descents on the host, then process_termresult -- and then submits it; nq = nb * batch queries, mirrors[s] for
 * s < num_shards on the contexts' device.  The arrays must stay alive until the run returns. */
int msq_pipeline_run_queries(msq_pipeline_t *p, mmirror_t *const *mirrors, int num_shards, const m_query_t *const *queries, int nq,
                             int batch, int first, int count, int warm, long *hits_out) {
    if (!p || !mirrors || !queries || batch < 1 || nq < batch || num_shards < 1) return MGPU_E_ARG;
    p->q_all = queries; p->q_nb = nq / batch; p->q_batch = batch; p->q_ns = num_shards; p->q_mirrors = mirrors;
    /* helper threads of a step's dictionary half: the plan-thread budget shared out among the pipeline's threads */
    p->q_threads = msq_plan_threads / (p->nthreads > 0 ? p->nthreads : 1);
    if (p->q_threads < 1) p->q_threads = 1;
    const int rc = warm ? pipe_dispatch(p, NULL, NULL, p->q_nb, 0, p->q_nb, 1, NULL) : pipe_dispatch(p, NULL, NULL, p->q_nb, first, count, 0, hits_out);
    p->q_all = NULL;
    return rc;
}
double msq_pipeline_lookup_ms(const msq_pipeline_t *p) {
    double t = 0;
    if (p) for (int i = 0; i < p->nthreads; i++) t += p->lookup_ms[i];
    return t;
}
/* host milliseconds the last run's threads spent building plans, summed over the threads */
double msq_pipeline_plan_ms(const msq_pipeline_t *p) {
    double t = 0;
    if (p) for (int i = 0; i < p->nthreads; i++) t += p->plan_ms[i];
    return t;
}

/* Several GPUs: from now on every batch of the loop ends with the all-gather of its result slots over `comm` (created
 * with at least as many slots as the pipeline has contexts) and this rank merges its share of the queries.  `pairs` =
 * (query, shard) pairs of a batch on this rank; k / num_facets / fmax / num_aggs = the slot shape (m_slot_bytes), the
 * same on every rank.  One gather per context slot runs here, in slot order, so that the communicator and the buffers
 * have been used before a caller starts a clock.  A batch with a cardinality aggregation is refused
 * (MGPU_E_UNSUPPORTED): its id sets are variable-length (marlin_amd/dist.py carries them in a second gather). */
int msq_pipeline_set_gather(msq_pipeline_t *p, mgpu_comm_t *comm, int pairs, int k, int num_facets, int fmax, int num_aggs) {
    if (!p || !comm || pairs < 1) return MGPU_E_ARG;
    const int world = mgpu_comm_world(comm);
    p->g_k = k; p->g_nf = num_facets; p->g_fmax = fmax; p->g_nagg = num_aggs; p->g_pairs = pairs;
    p->g_slot_b = m_slot_bytes(k, num_facets, fmax, num_aggs);
    const size_t n = (size_t)pairs * p->g_slot_b;
    for (int i = 0; i < p->nctx; i++) {
        mgpu_host_free(p->g_send[i]); mgpu_host_free(p->g_recv[i]);
        p->g_send[i] = mgpu_host_alloc(n); p->g_recv[i] = mgpu_host_alloc(n * (size_t)world);
        if (!p->g_send[i] || !p->g_recv[i]) return MGPU_E_LIMIT;
        memset(p->g_send[i], 0, n);
    }
    for (int i = 0; i < p->nctx; i++) {
        int rc = mgpu_comm_allgather_begin(comm, i, p->g_send[i], n, p->g_recv[i]);
        if (!rc) rc = mgpu_comm_allgather_end(comm, i);
        if (rc) { snprintf(p->err, sizeof(p->err), "%s", mgpu_last_error()); return rc; }
    }
    p->comm = comm;
    return 0;
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
    for (int i = 0; i < 128; i++) msq_prepared_free(p->slot_pp[i]);
    for (int i = 0; i < 128; i++) msq_resolved_free(p->slot_res[i]);
    for (int i = 0; i < 128; i++) { mgpu_host_free(p->g_send[i]); mgpu_host_free(p->g_recv[i]); }
    pthread_mutex_destroy(&p->g_mu); pthread_cond_destroy(&p->g_cv);
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
