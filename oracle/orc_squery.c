/* orc_squery.c -- ORACLE (test infrastructure): execute_squery restated on the CPU.
 * See orc.h for what this is and who may call it.  Every function cites the reference
 * code it follows (paths relative to /root/reference/src/main). */
#include "orc.h"
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <time.h>

/* seconds spent in the dictionary (trie) lookups since the last reset: lets bench.py report the
 * CPU baseline with and without the term lookup, which the GPU path keeps on the host too */
static __thread double orc_lookup_seconds = 0.0;
static double orc_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
double orc_lookup_time(int reset) { double v = orc_lookup_seconds; if (reset) orc_lookup_seconds = 0.0; return v; }

/* ------------------------------------------------------------------ dense bitmaps */
void orc_bm_or(uint64_t *d, const uint64_t *s, size_t w) { for (size_t i = 0; i < w; i++) d[i] |= s[i]; }
void orc_bm_and(uint64_t *d, const uint64_t *s, size_t w) { for (size_t i = 0; i < w; i++) d[i] &= s[i]; }
void orc_bm_andnot(uint64_t *d, const uint64_t *s, size_t w) { for (size_t i = 0; i < w; i++) d[i] &= ~s[i]; }
uint64_t orc_bm_cardinality(const uint64_t *a, size_t w) {
    uint64_t c = 0;
    for (size_t i = 0; i < w; i++) c += (uint64_t)__builtin_popcountll(a[i]);
    return c;
}
void orc_bm_from_list(uint64_t *d, const uint32_t *docs, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) d[docs[i] >> 6] |= 1ull << (docs[i] & 63);
}

/* ------------------------------------------------------------------ hashtable.c restated */
static inline uint32_t integer_hash(uint32_t h) {            /* hashtable.c:12-19 */
    h ^= h >> 16; h *= 0x85ebca6b; h ^= h >> 13; h *= 0xc2b2ae35; h ^= h >> 16;
    return h;
}

orc_hashtable_t *orc_hashtable_new(uint32_t size) {          /* hashtable.c:24-31 */
    orc_hashtable_t *h = calloc(1, sizeof(*h));
    h->array_size = size;
    h->cells = calloc(size, sizeof(orc_cell_t));
    return h;
}

orc_cell_t *orc_hashtable_lookup(orc_hashtable_t *h, uint32_t key) {   /* hashtable.c:33-39 */
    for (uint32_t i = integer_hash(key) & (h->array_size - 1);; i = (i + 1 != h->array_size) ? i + 1 : 0) {
        if (h->cells[i].key == key) return &h->cells[i];
        if (!h->cells[i].key) return NULL;
    }
}

static void ht_repopulate(orc_hashtable_t *h, uint32_t desired) {      /* hashtable.c:41-63 */
    orc_cell_t *old = h->cells;
    uint32_t old_size = h->array_size;
    h->array_size = desired;
    h->cells = calloc(desired, sizeof(orc_cell_t));
    for (uint32_t c = 0; c < old_size; c++) {
        if (!old[c].key) continue;
        for (uint32_t i = integer_hash(old[c].key) & (desired - 1);; i = (i + 1 != desired) ? i + 1 : 0) {
            if (!h->cells[i].key) { h->cells[i] = old[c]; break; }
        }
    }
    free(old);
}

orc_cell_t *orc_hashtable_insert(orc_hashtable_t *h, uint32_t key) {   /* hashtable.c:65-83 */
    for (;;) {
        int resized = 0;
        for (uint32_t i = integer_hash(key) & (h->array_size - 1);; i = (i + 1 != h->array_size) ? i + 1 : 0) {
            if (h->cells[i].key == key) return &h->cells[i];
            if (h->cells[i].key == 0) {
                if ((h->population + 1) * 4 >= h->array_size * 3) {
                    ht_repopulate(h, h->array_size * 2);
                    resized = 1;
                    break;
                }
                ++h->population;
                h->cells[i].key = key;
                return &h->cells[i];
            }
        }
        (void)resized;
    }
}

void orc_hashtable_free(orc_hashtable_t *h) { if (h) { free(h->cells); free(h); } }

/* ------------------------------------------------------------------ context */
typedef struct {
    const mshard_t *s;
    const m_query_t *q;
    size_t W;
    int num_terms;
    m_termresult_t tr[M_MAX_QTERMS];
    uint64_t *tbmap[M_MAX_QTERMS];
    uint64_t *tzero[M_MAX_QTERMS];
    m_u32map_t all_wordids;
    const uint32_t *exact_docs[M_MAX_QWORDS];
    uint64_t exact_n[M_MAX_QWORDS];
    orc_hashtable_t *fh[M_MAX_FIELDS];
    m_agg_t *aggs;
    int num_aggs;
    uint32_t *card_buf[M_MAX_AGGS];
    uint64_t card_n[M_MAX_AGGS], card_cap[M_MAX_AGGS];
    bool fast_rank;                     /* sq->fast_rank: perform_doc_processing leaves the aggs alone (docrank.c:409) */
} ctx_t;

static uint64_t *bm_new(const ctx_t *c) { return calloc(c->W ? c->W : 1, sizeof(uint64_t)); }
static uint64_t *bm_dup(const ctx_t *c, const uint64_t *a) {
    uint64_t *r = malloc((c->W ? c->W : 1) * sizeof(uint64_t));
    memcpy(r, a, c->W * sizeof(uint64_t));
    return r;
}

static bool list_exists(const uint32_t *docs, uint64_t n, uint32_t docid) {   /* bmap_exists, bmap.c:554-560 */
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (docs[mid] < docid) lo = mid + 1; else hi = mid;
    }
    return lo < n && docs[lo] == docid;
}

/* process_termresult (squery.c:72-118) */
static void process_termresult(ctx_t *c, int t) {
    m_termresult_t *tr = &c->tr[t];
    c->tbmap[t] = bm_new(c);
    c->tzero[t] = bm_new(c);
    if (tr->twid == 0 && tr->wordids.n == 0) return;
    const uint32_t *docs;
    if (tr->twid) {
        uint64_t n = mshard_twid_posting(c->s, tr->twid, &docs);     /* get_twid_to_docids, squery.c:39-46 */
        orc_bm_from_list(c->tbmap[t], docs, n);
        orc_bm_from_list(c->tzero[t], docs, n);
        if (n) {                                                     /* set_wids_under_twid, squery.c:62-70 */
            const uint32_t *wids;
            uint64_t nw = mshard_twid_wids(c->s, tr->twid, &wids);
            for (uint64_t i = 0; i < nw; i++) {
                m_u32map_set(&tr->wordids, wids[i], 0);
                m_u32map_set(&c->all_wordids, wids[i], 0);
            }
        }
        return;
    }
    for (uint32_t k = 0; k < tr->wordids.cap; k++) {                 /* kh_foreach, squery.c:101-111 */
        uint32_t wid = tr->wordids.keys[k];
        if (!wid) continue;
        int dist = tr->wordids.vals[k];
        uint64_t n = mshard_wid_posting(c->s, wid, &docs);
        if (!docs) continue;
        m_u32map_set(&c->all_wordids, wid, dist);
        orc_bm_from_list(c->tbmap[t], docs, n);
        if (dist == 0) orc_bm_from_list(c->tzero[t], docs, n);
    }
}

/* lookup_terms (squery.c:144-167) */
static void lookup_terms(ctx_t *c) {
    for (int i = 0; i < c->num_terms; i++) {
        double t0 = orc_now();
        mdict_lookup_term(c->s->dict, &c->q->terms[i], &c->tr[i]);
        orc_lookup_seconds += orc_now() - t0;
        process_termresult(c, i);
    }
    for (int i = 0; i < c->q->num_words; i++) {
        double t0 = orc_now();
        uint32_t wid = mdict_lookup_exact(c->s->dict, &c->q->words[i]);
        orc_lookup_seconds += orc_now() - t0;
        c->exact_docs[i] = NULL; c->exact_n[i] = 0;
        if (wid) c->exact_n[i] = mshard_wid_posting(c->s, wid, &c->exact_docs[i]);
    }
}

/* get_term_docids (squery.c:170-187) */
static uint64_t *get_term_docids(ctx_t *c, uint64_t **maps, int pos) {
    uint64_t *r = bm_new(c);
    if (pos > 0) orc_bm_or(r, maps[pos - 1], c->W);
    orc_bm_or(r, maps[pos], c->W);
    if (pos < c->num_terms - 2) orc_bm_or(r, maps[pos + 1], c->W);
    return r;
}

/* get_matching_docids / get_matching_zero_typo_docids (squery.c:212-328) */
static uint64_t *get_matching_docids(ctx_t *c, uint64_t **maps) {
    int nt = c->num_terms;
    if (nt == 0) {                                                   /* shard_get_all_docids, shard.c:51-53 */
        uint64_t *r = bm_new(c);
        if (c->s->has_used) for (uint64_t i = 0; i < c->s->used_n; i++) { uint32_t d = c->s->used_doc[i]; r[d >> 6] |= 1ull << (d & 63); }
        else for (uint32_t d = 1; d <= c->s->max_docid; d++) r[d >> 6] |= 1ull << (d & 63);
        return r;
    }
    if (nt == 1) return bm_dup(c, maps[0]);
    uint64_t *ret;
    if (nt == 3) {
        ret = bm_dup(c, maps[0]);
        orc_bm_and(ret, maps[1], c->W);
    } else {
        ret = NULL;
        for (int i = 0; i < nt; i += 2) {
            uint64_t *b = get_term_docids(c, maps, i);
            if (!ret) ret = b;
            else { orc_bm_and(ret, b, c->W); free(b); }
        }
    }
    if (orc_bm_cardinality(maps[nt - 1], c->W)) orc_bm_or(ret, maps[nt - 1], c->W);
    return ret;
}

/* ------------------------------------------------------------------ filters */
static bool check_double_oper(int f, double a, double b) {           /* filter_apply.c:127-144 */
    switch (f) {
        case MF_LT: return a < b;
        case MF_LTE: return a <= b;
        case MF_GT: return a > b;
        case MF_GTE: return a >= b;
        default: return false;
    }
}

/* returns the node's fr_bmap (may be NULL exactly where the reference leaves it NULL) */
static uint64_t *filter_apply(ctx_t *c, const m_filter_t *f, const uint64_t *docs) {   /* filter_apply.c:207-213 */
    uint64_t **kids = calloc((size_t)(f->num_children ? f->num_children : 1), sizeof(uint64_t *));
    for (int i = 0; i < f->num_children; i++) kids[i] = filter_apply(c, f->children[i], docs);
    uint64_t *r = NULL;
    const mshard_t *s = c->s;
    switch (f->type) {
        case MF_EQ: case MF_NE: {
            if (f->field_type == MT_NUMBER) {                        /* num_eq_filter, filter_apply.c:33-56 */
                r = bm_new(c);
                int fi = f->i_priority;
                if (fi >= 0 && fi < s->num_numbers) {
                    uint64_t n = s->np_n[fi], lo = 0, hi = n;
                    while (lo < hi) {                                /* MDB_SET_KEY on a descending tree */
                        uint64_t mid = (lo + hi) / 2;
                        if (s->np_val[fi][mid] > f->numval) lo = mid + 1; else hi = mid;
                    }
                    for (uint64_t k = lo; k < n && s->np_val[fi][k] == f->numval; k++) {
                        uint32_t d = s->np_doc[fi][k];
                        r[d >> 6] |= 1ull << (d & 63);
                    }
                }
            } else if (f->field_type == MT_STRING) {                 /* str_eq_filter, filter_apply.c:17-31 */
                if (f->is_facet) {
                    const uint32_t *pd;
                    uint64_t n = mshard_facet_posting(s, f->f_priority, f->facet_id, &pd);
                    r = bm_new(c);
                    orc_bm_from_list(r, pd, n);
                }
            } else if (f->field_type == MT_BOOLEAN) {                /* bool_eq_filter, filter_apply.c:8-15 */
                const uint32_t *pd;
                uint64_t n = mshard_bool_posting(s, f->i_priority, f->numval != 0.0, &pd);
                r = bm_new(c);
                orc_bm_from_list(r, pd, n);
            }
            if (f->type == MF_NE) {                                  /* ne_filter, filter_apply.c:77-86 */
                uint64_t *b = bm_dup(c, docs);
                if (r) { orc_bm_andnot(b, r, c->W); free(r); }
                r = b;
            }
        } break;
        case MF_OR: case MF_NIN: {                                   /* or_filter / nin_filter, filter_apply.c:89-125 */
            r = bm_new(c);
            for (int i = 0; i < f->num_children; i++) if (kids[i]) orc_bm_or(r, kids[i], c->W);
            if (f->type == MF_NIN) {
                uint64_t *b = bm_dup(c, docs);
                orc_bm_andnot(b, r, c->W);
                free(r);
                r = b;
            }
        } break;
        case MF_AND: {                                               /* and_filter + oper_and, bmap.c:431-453 */
            int cnt = 0;
            for (int i = 0; i < f->num_children; i++) {
                if (!kids[i]) continue;
                if (cnt == 0) r = bm_dup(c, kids[i]); else orc_bm_and(r, kids[i], c->W);
                cnt++;
            }
            if (cnt == 0) r = bm_new(c);
        } break;
        case MF_GT: case MF_GTE: case MF_LT: case MF_LTE: {          /* num_filter, filter_apply.c:146-168 */
            if (f->field_type != MT_NUMBER) break;
            r = bm_new(c);
            int fi = f->i_priority;
            if (fi < 0 || fi >= s->num_numbers) break;
            uint64_t n = s->np_n[fi];
            bool gt = (f->type == MF_GT || f->type == MF_GTE);
            if (gt) {
                for (uint64_t k = 0; k < n; k++) {
                    if (!check_double_oper(f->type, s->np_val[fi][k], f->numval)) break;
                    uint32_t d = s->np_doc[fi][k]; r[d >> 6] |= 1ull << (d & 63);
                }
            } else {
                for (uint64_t k = n; k-- > 0;) {
                    if (!check_double_oper(f->type, s->np_val[fi][k], f->numval)) break;
                    uint32_t d = s->np_doc[fi][k]; r[d >> 6] |= 1ull << (d & 63);
                }
            }
        } break;
        case MF_RANGE: {                                             /* num_range_filter, filter_apply.c:170-194 */
            if (f->field_type != MT_NUMBER) break;
            r = bm_new(c);
            int fi = f->i_priority;
            if (fi < 0 || fi >= s->num_numbers) break;
            uint64_t n = s->np_n[fi], lo = 0, hi = n;
            while (lo < hi) {                                        /* MDB_SET_RANGE: first key <= numval2 */
                uint64_t mid = (lo + hi) / 2;
                if (s->np_val[fi][mid] > f->numval2) lo = mid + 1; else hi = mid;
            }
            for (uint64_t k = lo; k < n; k++) {
                double kd = s->np_val[fi][k];
                if (check_double_oper(f->numcmp2, kd, f->numval2)) {
                    if (check_double_oper(f->numcmp1, kd, f->numval)) {
                        uint32_t d = s->np_doc[fi][k]; r[d >> 6] |= 1ull << (d & 63);
                    } else break;
                }
            }
        } break;
        default: break;
    }
    for (int i = 0; i < f->num_children; i++) free(kids[i]);
    free(kids);
    return r;
}

/* ------------------------------------------------------------------ ranking */
typedef struct { uint32_t wid; uint32_t info; } winfo_t;             /* wid_info_t, sindex.h:94-100 */
#define WI_IS_POS(i)   ((i).info & 1u)
#define WI_PRIO(i)     (((i).info >> 1) & 0xFFu)
#define WI_OFFSET(i)   ((i).info >> 9)

typedef struct { int *a; int n, cap; } ivec_t;
static void ivec_push(ivec_t *v, int x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 8; v->a = realloc(v->a, sizeof(int) * (size_t)v->cap); }
    v->a[v->n++] = x;
}
static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return x < y ? -1 : (x > y); }

/* fill_positions (docrank.c:14-47) */
static void fill_positions(const winfo_t *info, const uint8_t *head, ivec_t *p) {
    if (WI_IS_POS(*info)) {
        ivec_push(p, (int)((WI_PRIO(*info) << 16) + WI_OFFSET(*info)));
        return;
    }
    const uint8_t *c = head + WI_OFFSET(*info);
    int freq = *c++;
    while (freq > 0) {
        int priority = *c++;
        int count = *c++;
        freq -= count;
        while (count > 0) {
            int position;
            c = m_read_vint(c, &position);
            ivec_push(p, (priority << 16) + position);
            count--;
        }
        c++;                                                         /* 0xFF separator */
    }
}

/* get_priority_position (docrank.c:49-61) */
static void get_priority_position(const winfo_t *info, const uint8_t *head, int *priority, int *position) {
    if (WI_IS_POS(*info)) { *priority = (int)WI_PRIO(*info); *position = (int)WI_OFFSET(*info); return; }
    const uint8_t *p = head + WI_OFFSET(*info);
    p++;
    *priority = *p;
    p += 2;
    m_read_vint(p, position);
}

static inline int all_dist(const ctx_t *c, uint32_t wid) { return m_u32map_get(&c->all_wordids, wid, 0xFF); }

/* get_matching_term (docrank.c:114-123) */
static int get_matching_term(const ctx_t *c, uint32_t wid) {
    for (int i = 0; i < c->num_terms; i++)
        if (m_u32map_get(&c->tr[i].wordids, wid, 0xFF) != 0xFF) return i;
    return 0xFF;
}

/* rank_single_term (docrank.c:63-112) */
static void rank_single_term(const ctx_t *c, m_docrank_t *rank, const uint8_t *dpos) {
    uint16_t numwid; memcpy(&numwid, dpos, 2);
    const winfo_t *info = (const winfo_t *)(dpos + 2);
    int min_dist = 0xFF - 1;
    if (c->exact_docs[0] && list_exists(c->exact_docs[0], c->exact_n[0], rank->docid)) {
        rank->exact = 1; rank->typos = 0; min_dist = 0;
    }
    for (int i = 0; i < numwid; i++) {
        winfo_t wi; memcpy(&wi, &info[i], 8);
        int dist = all_dist(c, wi.wid);
        if (dist <= min_dist) {
            min_dist = dist;
            if (dist > rank->typos) continue;
            if (dist < rank->typos) { rank->typos = (uint8_t)dist; rank->field = 0xFF; rank->position = 0xFFFF; }
            int priority, position;
            get_priority_position(&wi, dpos, &priority, &position);
            if (priority > rank->field) continue;
            if (priority < rank->field) { rank->field = (uint8_t)priority; rank->position = (uint16_t)position; }
            else if (position < rank->position) rank->position = (uint16_t)position;
        }
    }
}

static int prox_walk(const ivec_t *a, const ivec_t *b) {             /* docrank.c:188-204, 303-319 */
    int i = 0, j = 0, mindiff = 8;
    while (i < a->n && j < b->n) {
        int pos1 = a->a[i], pos2 = b->a[j];
        int diff = abs(pos1 - pos2);
        if (diff != 0 && diff < mindiff) { mindiff = diff; if (mindiff == 1) break; }
        if (pos1 > pos2) j++; else i++;
    }
    return mindiff;
}

/* rank_three_terms (docrank.c:125-209) */
static void rank_three_terms(const ctx_t *c, m_docrank_t *rank, const uint8_t *dpos) {
    const int num_words = 2, num_terms = 3;
    int typos[2] = { 0xFF, 0xFF };
    ivec_t positions[3]; memset(positions, 0, sizeof(positions));
    for (int i = 0; i < num_words; i++) {
        if (c->exact_docs[i] && list_exists(c->exact_docs[i], c->exact_n[i], rank->docid)) {
            rank->exact++; typos[i] = 0;
        }
    }
    uint16_t numwid; memcpy(&numwid, dpos, 2);
    const winfo_t *info = (const winfo_t *)(dpos + 2);
    for (int i = 0; i < numwid; i++) {
        winfo_t wi; memcpy(&wi, &info[i], 8);
        int dist = all_dist(c, wi.wid);
        if (dist != 0xFF) {
            int mterm = get_matching_term(c, wi.wid);
            if (mterm == 2) { rank->proximity = 1; typos[0] = typos[1] = 0; }
            else if (mterm < 2 && dist < typos[mterm]) typos[mterm] = dist;
            if (mterm < 3) fill_positions(&wi, dpos, &positions[mterm]);
        }
    }
    uint32_t best_position = 0xFFFFFFFF;
    rank->typos = 0;
    for (int x = 0; x < num_terms; x++) {
        if (positions[x].n) {
            qsort(positions[x].a, (size_t)positions[x].n, sizeof(int), cmp_int);
            if ((uint32_t)positions[x].a[0] < best_position) best_position = (uint32_t)positions[x].a[0];
        }
        if (x < num_words) rank->typos = (uint8_t)(rank->typos + typos[x]);
    }
    rank->field = (uint8_t)(best_position >> 16);
    rank->position = (uint16_t)(best_position & 0xFFFF);
    if (rank->proximity != 1) rank->proximity = (uint16_t)prox_walk(&positions[0], &positions[1]);
    for (int i = 0; i < 3; i++) free(positions[i].a);
}

static inline int term_to_word_idx(int term, int num_terms) {        /* docrank.c:211-214, quirk kept */
    if ((term != 0) == (num_terms - 1)) return 0;
    return term / 2;
}

/* rank_many_terms (docrank.c:218-328) */
static void rank_many_terms(const ctx_t *c, m_docrank_t *rank, const uint8_t *dpos) {
    int num_words = c->q->num_words, num_terms = c->num_terms;
    int typos[M_MAX_QWORDS], proximity[M_MAX_QWORDS];
    ivec_t positions[M_MAX_QWORDS]; memset(positions, 0, sizeof(positions));
    for (int i = 0; i < num_words; i++) {
        if (c->exact_docs[i] && list_exists(c->exact_docs[i], c->exact_n[i], rank->docid)) {
            rank->exact++; typos[i] = 0;
        } else typos[i] = 0xFF;
        proximity[i] = 8;
    }
    rank->proximity = 0;
    uint16_t numwid; memcpy(&numwid, dpos, 2);
    const winfo_t *info = (const winfo_t *)(dpos + 2);
    for (int i = 0; i < numwid; i++) {
        winfo_t wi; memcpy(&wi, &info[i], 8);
        int dist = all_dist(c, wi.wid);
        if (dist != 0xFF) {
            int mterm = get_matching_term(c, wi.wid);
            if (mterm == num_terms - 1) {
                rank->proximity = (uint16_t)(num_words - 1);
                memset(typos, 0, sizeof(int) * (size_t)num_words);
            }
            int widx = term_to_word_idx(mterm, num_terms);
            if (dist < typos[widx]) typos[widx] = dist;
            fill_positions(&wi, dpos, &positions[widx]);
            if ((mterm % 2) == 1) {
                proximity[widx] = 1;
                if (mterm != num_terms - 1) {
                    typos[widx + 1] = 0;
                    fill_positions(&wi, dpos, &positions[widx + 1]);
                }
            }
        }
    }
    uint32_t best_position = 0xFFFFFFFF;
    rank->typos = 0;
    for (int x = 0; x < num_words; x++) {
        if (positions[x].n) {
            qsort(positions[x].a, (size_t)positions[x].n, sizeof(int), cmp_int);
            if ((uint32_t)positions[x].a[0] < best_position) best_position = (uint32_t)positions[x].a[0];
        }
        rank->typos = (uint8_t)(rank->typos + typos[x]);
    }
    rank->field = (uint8_t)(best_position >> 16);
    rank->position = (uint16_t)(best_position & 0xFFFF);
    if (rank->proximity != num_words - 1) {
        for (int x = 0; x < num_words - 1; x++) {
            if (proximity[x] != 1) proximity[x] = prox_walk(&positions[x], &positions[x + 1]);
            rank->proximity = (uint16_t)(rank->proximity + proximity[x]);
        }
    }
    for (int i = 0; i < num_words; i++) free(positions[i].a);
}

/* calculate_rank (docrank.c:338-369) */
static void calculate_rank(const ctx_t *c, m_docrank_t *rank, const uint8_t *dpos) {
    int num_words = c->q->num_words;
    if (num_words == 0) {
        rank->position = 0; rank->proximity = 0; rank->exact = 0; rank->field = 0; rank->typos = 0;
        return;
    }
    rank->field = 0xFF; rank->typos = 0xFF; rank->position = 0xFFFF; rank->exact = 0;
    if (num_words == 1) { rank->proximity = 0; rank_single_term(c, rank, dpos); }
    else if (num_words == 2) { rank->proximity = 0xFFFF; rank_three_terms(c, rank, dpos); }
    else { rank->proximity = 0xFFFF; rank_many_terms(c, rank, dpos); }
}

/* consume_*_agg (metric-aggs.c:23-45, 146-159, 223-230, 281-289, 342-350), data != NULL branch */
static void aggs_consume(ctx_t *c, const uint8_t *blob) {
    const mshard_t *s = c->s;
    for (int a = 0; a < c->num_aggs; a++) {
        m_agg_t *ag = &c->aggs[a];
        if (ag->type == MA_CARDINALITY) {
            const uint8_t *fpos = blob + 4 + (size_t)s->num_numbers * 8;
            for (int i = 0; i < s->num_facets; i++) {
                uint32_t cnt; memcpy(&cnt, fpos, 4);
                if (i == ag->field) {
                    for (uint32_t j = 0; j < cnt; j++) {
                        uint32_t id; memcpy(&id, fpos + 4 + (size_t)j * 4, 4);
                        if (c->card_n[a] == c->card_cap[a]) {
                            c->card_cap[a] = c->card_cap[a] ? c->card_cap[a] * 2 : 256;
                            c->card_buf[a] = realloc(c->card_buf[a], c->card_cap[a] * sizeof(uint32_t));
                        }
                        c->card_buf[a][c->card_n[a]++] = id;
                    }
                    break;
                }
                fpos += (size_t)cnt * 4 + 4;
            }
            continue;
        }
        double val; memcpy(&val, blob + 4 + (size_t)ag->field * 8, 8);      /* get_num_value, metric-aggs.c:105-112 */
        if (val == -DBL_MAX) continue;
        switch (ag->type) {
            case MA_MAX: if (val > ag->value) ag->value = val; break;
            case MA_MIN: if (val < ag->value) ag->value = val; break;
            case MA_AVG: ag->sum += val; ag->count++; break;
            case MA_STATS:
                ag->sum += val;
                if (val < ag->min) ag->min = val;
                if (val > ag->max) ag->max = val;
                ag->count++;
                break;
            default: break;
        }
    }
}

/* perform_doc_rank + perform_doc_processing (docrank.c:396-443) */
static void perform_doc_rank(ctx_t *c, m_docrank_t *rank) {
    const m_query_t *q = c->q;
    uint32_t bsz;
    const uint8_t *blob = mshard_docdata(c->s, rank->docid, &bsz);
    if (!blob || bsz == 0) {                                         /* docrank.c:420-428 */
        rank->typos = 0xFF; rank->position = 0xFFFF; rank->proximity = 0xFFFF;
        rank->compare = q->cfg.rank_asc ? DBL_MAX : -DBL_MAX;
        return;
    }
    if (q->cfg.hits_per_page > 0 && q->num_words > 0) {
        uint32_t off; memcpy(&off, blob, 4);
        calculate_rank(c, rank, blob + off);
    } else {
        rank->position = 0; rank->proximity = 0; rank->exact = 0; rank->field = 0; rank->typos = 0;
    }
    if (q->cfg.max_facet_results) {                                  /* process_facet_result, docrank.c:371-389 */
        const uint8_t *fpos = blob + 4 + (size_t)c->s->num_numbers * 8;
        for (int i = 0; i < c->s->num_facets; i++) {
            uint32_t cnt; memcpy(&cnt, fpos, 4);
            if (q->cfg.facet_enabled[i]) {
                for (uint32_t j = 0; j < cnt; j++) {
                    uint32_t id; memcpy(&id, fpos + 4 + (size_t)j * 4, 4);
                    orc_hashtable_insert(c->fh[i], id)->value += 1;
                }
            }
            fpos += (size_t)cnt * 4 + 4;
        }
    }
    if (q->cfg.rank_by >= 0) memcpy(&rank->compare, blob + 4 + (size_t)q->cfg.rank_by * 8, 8);
    if (c->num_aggs && !c->fast_rank) aggs_consume(c, blob);
}

static inline bool bm_test(const uint64_t *m, uint32_t d) { return (m[d >> 6] >> (d & 63)) & 1u; }
static inline void bm_set(uint64_t *m, uint32_t d) { m[d >> 6] |= 1ull << (d & 63); }

/* consume_*_agg with data == NULL (calculate_agg, docrank.c:445-448): the value comes from idnum2dbl, one array of
 * 1000 doubles per (docid group, field) that exists once ANY document of the group indexed a number of that field
 * (index_number, sindex.c:528-565) and is zero-filled otherwise; a group that does not exist yields "no value"
 * (get_num_value, metric-aggs.c:105-142).  grp_has[field] = bitmap over the groups. */
static void aggs_consume_fast(ctx_t *c, uint32_t docid, uint64_t *const *grp_has) {
    const mshard_t *s = c->s;
    for (int a = 0; a < c->num_aggs; a++) {
        m_agg_t *ag = &c->aggs[a];
        if (ag->type == MA_CARDINALITY) continue;                    /* refused before we get here */
        if (!bm_test(grp_has[ag->field], docid / 1000)) continue;
        double val = 0.0;
        uint32_t bsz; const uint8_t *blob = mshard_docdata(s, docid, &bsz);
        if (blob && bsz >= 4 + (size_t)(ag->field + 1) * 8) memcpy(&val, blob + 4 + (size_t)ag->field * 8, 8);
        if (val == -DBL_MAX) continue;
        switch (ag->type) {
            case MA_MAX: if (val > ag->value) ag->value = val; break;
            case MA_MIN: if (val < ag->value) ag->value = val; break;
            case MA_AVG: ag->sum += val; ag->count++; break;
            case MA_STATS:
                ag->sum += val;
                if (val < ag->min) ag->min = val;
                if (val > ag->max) ag->max = val;
                ag->count++;
                break;
            default: break;
        }
    }
}

/* perform_fast_ranking (docrank.c:478-560).  Returns the rank array (capacity as the reference mallocs it) and the
 * count it reports, or NULL with *err set when the reference itself would misbehave: a cardinality agg (its consume
 * dereferences the NULL data pointer, metric-aggs.c:23-45) or more sampled documents than its buffer holds. */
static m_docrank_t *perform_fast_ranking(ctx_t *c, const uint64_t *docid_map, const uint64_t *zero_map,
                                         uint32_t totalcount, uint32_t *resultcount, int *err) {
    const m_query_t *q = c->q; const mshard_t *s = c->s;
    *err = 0;
    for (int a = 0; a < c->num_aggs; a++) if (c->aggs[a].type == MA_CARDINALITY) { *err = -5; return NULL; }
    int fcount = (int)(totalcount / 100) < M_MAX_HITS_LIMIT * 10 ? (int)(totalcount / 100) : M_MAX_HITS_LIMIT * 10;
    if (fcount < M_MAX_HITS_LIMIT * 5) fcount = M_MAX_HITS_LIMIT * 5;
    const uint32_t cap = (uint32_t)(M_MAX_HITS_LIMIT + fcount + 10);
    m_docrank_t *ranks = calloc(cap, sizeof(m_docrank_t));
    uint64_t *dbmap = bm_new(c);
    const uint64_t *rmap = docid_map;
    uint32_t rankpos = 0;
    uint64_t zt_card = orc_bm_cardinality(zero_map, c->W);
    if (zt_card) {
        if (zt_card <= M_MAX_HITS_LIMIT) {                           /* setup_zero_typo_ranks */
            for (size_t w = 0; w < c->W; w++) {
                uint64_t x = zero_map[w];
                while (x) {
                    int r = __builtin_ctzll(x); x &= x - 1;
                    uint32_t d = (uint32_t)(w * 64 + (size_t)r);
                    ranks[rankpos].docid = d; perform_doc_rank(c, &ranks[rankpos]); rankpos++;
                    bm_set(dbmap, d);
                }
            }
        } else rmap = zero_map;
    }
    /* documents in ranking order: num_dbi[rank_by] first to last, or last to first for ascending (MDB_LAST / MDB_PREV) */
    {
        const int f = q->cfg.rank_by;
        const uint64_t n = s->np_n[f];
        for (uint64_t i = 0; i < n; i++) {
            uint32_t d = s->np_doc[f][q->cfg.rank_asc ? n - 1 - i : i];
            if (d > s->max_docid || !bm_test(rmap, d)) continue;
            if (bm_test(dbmap, d)) continue;
            ranks[rankpos].docid = d; perform_doc_rank(c, &ranks[rankpos]);
            bm_set(dbmap, d);
            rankpos++;
            if (rankpos >= M_MAX_HITS_LIMIT) break;
        }
    }
    /* the rest: every skip_count-th document of rmap in docid order (setup_skip_ranks) */
    {
        const uint32_t skip_count = totalcount / (uint32_t)fcount + 1;
        uint32_t counter = 0;
        for (size_t w = 0; w < c->W && !*err; w++) {
            uint64_t x = rmap[w];
            while (x) {
                int r = __builtin_ctzll(x); x &= x - 1;
                uint32_t d = (uint32_t)(w * 64 + (size_t)r);
                if (counter++ % skip_count != 0) continue;
                if (bm_test(dbmap, d)) continue;
                if (rankpos >= cap) { *err = -3; break; }            /* the reference writes past its malloc here */
                ranks[rankpos].docid = d; perform_doc_rank(c, &ranks[rankpos]); rankpos++;
            }
        }
    }
    free(dbmap);
    if (*err) { free(ranks); return NULL; }
    *resultcount = rankpos - 1;                                      /* docrank.c:550: the last ranked document is dropped */
    if (c->num_aggs) {                                               /* aggregates over ALL matching documents */
        uint64_t *grp_has[M_MAX_FIELDS];
        const size_t gw = ((size_t)s->max_docid / 1000 + 64) / 64;
        for (int i = 0; i < s->num_numbers; i++) {
            grp_has[i] = calloc(gw, sizeof(uint64_t));
            for (uint64_t k = 0; k < s->np_n[i]; k++) bm_set(grp_has[i], s->np_doc[i][k] / 1000);
        }
        for (size_t w = 0; w < c->W; w++) {
            uint64_t x = docid_map[w];
            while (x) { int r = __builtin_ctzll(x); x &= x - 1; aggs_consume_fast(c, (uint32_t)(w * 64 + (size_t)r), grp_has); }
        }
        for (int i = 0; i < s->num_numbers; i++) free(grp_has[i]);
    }
    return ranks;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y);
}

/* ------------------------------------------------------------------ execute_squery */
int orc_execute_squery(const mshard_t *s, const m_query_t *q, int shard_idx, int flags, m_sqresult_t *out) {
    memset(out, 0, sizeof(*out));
    ctx_t *c = calloc(1, sizeof(*c));
    c->s = s; c->q = q;
    c->W = ((size_t)s->max_docid + 64) / 64;
    c->num_terms = q->num_terms;
    m_u32map_init(&c->all_wordids);
    for (int i = 0; i < s->num_facets; i++)                          /* init_facet_hash, squery.c:27-37 */
        c->fh[i] = q->cfg.facet_enabled[i] ? orc_hashtable_new(1024) : NULL;

    lookup_terms(c);
    uint64_t *docid_map = get_matching_docids(c, c->tbmap);
    uint64_t *zero_map = get_matching_docids(c, c->tzero);           /* computed and, in full-scan mode, unused */

    if (q->filter) {                                                 /* squery_apply_filters, squery.c:377-395 */
        uint64_t *fb = filter_apply(c, q->filter, docid_map);
        if (fb) { orc_bm_and(docid_map, fb, c->W); free(fb); }
        else memset(docid_map, 0, c->W * sizeof(uint64_t));
    }
    c->num_aggs = q->num_aggs;                                       /* aggs_dup, squery.c:496-498 */
    out->num_aggs = q->num_aggs;
    for (int a = 0; a < q->num_aggs; a++) { out->aggs[a] = q->aggs[a]; out->aggs[a].card_ids = NULL; out->aggs[a].card_n = 0; }
    c->aggs = out->aggs;

    uint32_t num_hits = (uint32_t)orc_bm_cardinality(docid_map, c->W);
    out->num_hits = num_hits;
    int rc = 0;
    /* perform_ranking (docrank.c:565-587) */
    m_docrank_t *ranks;
    uint32_t resultcount = num_hits;
    c->fast_rank = !q->cfg.full_scan && num_hits > q->cfg.full_scan_threshold && q->cfg.rank_by >= 0;
    out->fast_rank = c->fast_rank;
    if (c->fast_rank) {
        ranks = perform_fast_ranking(c, docid_map, zero_map, num_hits, &resultcount, &rc);
        if (!ranks) goto done;
    } else {
        ranks = malloc((num_hits ? num_hits : 1) * sizeof(m_docrank_t));
        memset(ranks, 0, (num_hits ? num_hits : 1) * sizeof(m_docrank_t));   /* malloc garbage in the reference */
        uint32_t rp = 0;
        for (size_t w = 0; w < c->W; w++) {                              /* bmap_iterate -> setup_ranks */
            uint64_t x = docid_map[w];
            while (x) {
                int r = __builtin_ctzll(x);
                x &= x - 1;
                ranks[rp].docid = (uint32_t)(w * 64 + (size_t)r);
                perform_doc_rank(c, &ranks[rp]);
                rp++;
            }
        }
        if (flags & 2) {
            out->dbg_all_ranks = malloc((num_hits ? num_hits : 1) * sizeof(m_docrank_t));
            memcpy(out->dbg_all_ranks, ranks, num_hits * sizeof(m_docrank_t));
        }
    }

    /* sort_facets (squery.c:411-457) */
    out->num_facets = s->num_facets;
    for (int i = 0; i < s->num_facets; i++) {
        if (!q->cfg.facet_enabled[i]) continue;
        orc_hashtable_t *h = c->fh[i];
        m_facet_count_t *fc = malloc((h->population ? h->population : 1) * sizeof(m_facet_count_t));
        memset(fc, 0, (h->population ? h->population : 1) * sizeof(m_facet_count_t));
        uint32_t j = 0;
        for (uint32_t x = 0; x < h->array_size; x++) {
            if (h->cells[x].key) { fc[j].facet_id = h->cells[x].key; fc[j].count = h->cells[x].value; j++; }
        }
        int rcount = q->cfg.max_facet_results * 2;
        if ((int)h->population > rcount) orc_facet_partialsort(fc, 0, h->population - 1, (size_t)rcount);
        else rcount = (int)h->population;
        if (c->fast_rank) {                                          /* get_facet_totalcount, squery.c:398-408,439-446 */
            for (int x = 0; x < rcount; x++) {
                const uint32_t *docs;
                uint64_t n = mshard_facet_posting(s, i, fc[x].facet_id, &docs);
                uint32_t cnt = 0;
                for (uint64_t k = 0; k < n; k++) cnt += bm_test(docid_map, docs[k]) ? 1u : 0u;
                fc[x].count = cnt;
            }
        }
        for (int x = 0; x < rcount; x++) fc[x].shard_id = (uint8_t)shard_idx;
        out->fc[i] = fc;
        out->rcount[i] = rcount;
    }

    /* sort_results (squery.c:358-375) */
    {
        int sortcnt = q->cfg.hits_per_page * q->page_num;
        if (q->cfg.hits_per_page > 0) {
            if (resultcount >= M_PARTIAL_SORT_LIMIT && resultcount > (uint32_t)sortcnt)
                orc_rank_partial_sort(ranks, 0, resultcount - 1, (size_t)sortcnt, q->rank_rule);
            else orc_rank_sort(resultcount, ranks, q->rank_rule);
        }
        out->rank_count = ((uint32_t)sortcnt < resultcount) ? sortcnt : (int)resultcount;
        if (!(flags & 4)) orc_rank_sort((size_t)out->rank_count, ranks, q->rank_rule);   /* form_result, query.c:347 */
    }
    out->ranks = ranks;

    /* cardinality aggs: the bitmap of facet ids becomes a sorted id set */
    for (int a = 0; a < q->num_aggs; a++) {
        if (out->aggs[a].type != MA_CARDINALITY) continue;
        qsort(c->card_buf[a], c->card_n[a], sizeof(uint32_t), cmp_u32);
        uint64_t m = 0;
        for (uint64_t i = 0; i < c->card_n[a]; i++)
            if (i == 0 || c->card_buf[a][i] != c->card_buf[a][i - 1]) c->card_buf[a][m++] = c->card_buf[a][i];
        out->aggs[a].card_ids = c->card_buf[a]; out->aggs[a].card_n = (uint32_t)m;
        c->card_buf[a] = NULL;
    }

done:
    if (flags & 1) {
        out->dbg_docid_map = docid_map; out->dbg_words = (uint32_t)c->W;
    } else free(docid_map);
    free(zero_map);
    for (int i = 0; i < c->num_terms; i++) { m_termresult_free(&c->tr[i]); free(c->tbmap[i]); free(c->tzero[i]); }
    for (int i = 0; i < s->num_facets; i++) orc_hashtable_free(c->fh[i]);
    for (int a = 0; a < M_MAX_AGGS; a++) free(c->card_buf[a]);
    m_u32map_free(&c->all_wordids);
    free(c);
    return rc;
}

/* ------------------------------------------------------------------ cpu_baseline on several cores */
#include <pthread.h>
#include <malloc.h>
typedef struct {
    const mshard_t *const *shards; int nshards; const m_query_t *const *queries; int nq;
    int *next; pthread_mutex_t *mu;
    double lookup_s, busy_s; uint64_t hits;
} pool_job_t;

static void *pool_worker(void *arg) {
    pool_job_t *j = arg;
    orc_lookup_time(1);
    const int total = j->nq * j->nshards;
    for (;;) {
        pthread_mutex_lock(j->mu);
        const int i = (*j->next)++;
        pthread_mutex_unlock(j->mu);
        if (i >= total) break;
        m_sqresult_t out;
        const double t0 = orc_now();
        orc_execute_squery(j->shards[i % j->nshards], j->queries[i / j->nshards], i % j->nshards, 0, &out);
        j->busy_s += orc_now() - t0;
        j->hits += out.num_hits;
        m_sqresult_free(&out);
    }
    j->lookup_s = orc_lookup_time(1);
    return NULL;
}

double orc_bench_pool(const mshard_t *const *shards, int nshards, const m_query_t *const *queries, int nq, int nthreads,
                      double *lookup_cpu_s, double *busy_cpu_s, uint64_t *hits) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    /* per-query maps of a few MB: keep them in the threads' arenas instead of mmap/munmap per job (page faults
     * under one mm lock serialise the pool) */
    mallopt(M_MMAP_THRESHOLD, 32 << 20);          /* glibc's maximum */
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    pool_job_t jobs[256]; pthread_t th[256];
    int next = 0; pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    const double t0 = orc_now();
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (pool_job_t){ shards, nshards, queries, nq, &next, &mu, 0.0, 0.0, 0 };
        pthread_create(&th[t], NULL, pool_worker, &jobs[t]);
    }
    double lk = 0, busy = 0; uint64_t h = 0;
    for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); lk += jobs[t].lookup_s; busy += jobs[t].busy_s; h += jobs[t].hits; }
    const double wall = orc_now() - t0;
    if (lookup_cpu_s) *lookup_cpu_s = lk;
    if (busy_cpu_s) *busy_cpu_s = busy;
    if (hits) *hits = h;
    return wall;
}
