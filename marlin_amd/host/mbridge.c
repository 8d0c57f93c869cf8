/* mbridge.c -- see mbridge.h */
#include "mbridge.h"
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t id, prio; uint64_t off, n; } mbr_post_t;
typedef struct { mbr_post_t *v; size_t n, cap; } mbr_postvec_t;
typedef struct { uint32_t *v; uint64_t n, cap; } mbr_u32vec_t;
typedef struct { double *val; uint32_t *doc; uint64_t n, cap; } mbr_numvec_t;
typedef struct { uint32_t docid; uint64_t off; uint32_t len; } mbr_dd_t;

struct mbridge {
    int num_numbers, num_bools, num_facets;
    uint32_t word_count, tword_count, max_docid;
    mbr_postvec_t post[5];          /* by kind */
    mbr_u32vec_t pool;              /* decoded members of every posting */
    mbr_numvec_t num[M_MAX_FIELDS];
    mbr_dd_t *dd; size_t ndd, capdd;
    uint8_t *blob; uint64_t nblob, capblob;
    uint32_t *used; uint64_t nused;
};

mbridge_t *mbr_new(int num_numbers, int num_bools, int num_facets, uint32_t word_count, uint32_t tword_count) {
    if (num_numbers < 0 || num_numbers > M_MAX_FIELDS || num_bools < 0 || num_bools > M_MAX_FIELDS ||
        num_facets < 0 || num_facets > M_MAX_FIELDS) return NULL;
    mbridge_t *b = calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->num_numbers = num_numbers; b->num_bools = num_bools; b->num_facets = num_facets;
    b->word_count = word_count; b->tword_count = tword_count;
    return b;
}

void mbr_free(mbridge_t *b) {
    if (!b) return;
    for (int k = 0; k < 5; k++) free(b->post[k].v);
    free(b->pool.v);
    for (int i = 0; i < M_MAX_FIELDS; i++) { free(b->num[i].val); free(b->num[i].doc); }
    free(b->dd); free(b->blob); free(b->used);
    free(b);
}

int mbr_add_posting(mbridge_t *b, mbr_kind kind, uint32_t id, uint32_t priority, const mgpu_mbmap_t *stored) {
    if (!b || (int)kind < 0 || (int)kind > MBR_BOOL) return MGPU_E_ARG;
    if ((kind == MBR_WID && (id == 0 || id > b->word_count || priority)) ||
        ((kind == MBR_TWID || kind == MBR_TWID_WIDS) && (id == 0 || id > b->tword_count || priority)) ||
        (kind == MBR_FACET && priority >= (uint32_t)b->num_facets) ||
        (kind == MBR_BOOL && (priority >= (uint32_t)b->num_bools || id > 1))) return MGPU_E_ARG;
    uint64_t n = 0;
    int rc = mgpu_mbmap_decode(stored, NULL, 0, &n);
    if (rc && rc != MGPU_E_LIMIT) return rc;
    if (b->pool.n + n > b->pool.cap) {
        uint64_t cap = b->pool.cap ? b->pool.cap * 2 : 1u << 16;
        while (cap < b->pool.n + n) cap *= 2;
        uint32_t *v = realloc(b->pool.v, cap * sizeof(uint32_t));
        if (!v) return MGPU_E_LIMIT;
        b->pool.v = v; b->pool.cap = cap;
    }
    if (n && (rc = mgpu_mbmap_decode(stored, b->pool.v + b->pool.n, n, &n))) return rc;
    mbr_postvec_t *pv = &b->post[kind];
    if (pv->n == pv->cap) {
        size_t cap = pv->cap ? pv->cap * 2 : 1024;
        mbr_post_t *v = realloc(pv->v, cap * sizeof(*v));
        if (!v) return MGPU_E_LIMIT;
        pv->v = v; pv->cap = cap;
    }
    pv->v[pv->n++] = (mbr_post_t){id, priority, b->pool.n, n};
    b->pool.n += n;
    return 0;
}

int mbr_add_docdata(mbridge_t *b, uint32_t docid, const void *blob, size_t len) {
    if (!b || docid == 0 || len > 0xFFFFFFFFu) return MGPU_E_ARG;
    if (b->ndd == b->capdd) {
        size_t cap = b->capdd ? b->capdd * 2 : 4096;
        mbr_dd_t *v = realloc(b->dd, cap * sizeof(*v));
        if (!v) return MGPU_E_LIMIT;
        b->dd = v; b->capdd = cap;
    }
    if (b->nblob + len > b->capblob) {
        uint64_t cap = b->capblob ? b->capblob * 2 : 1u << 20;
        while (cap < b->nblob + len) cap *= 2;
        uint8_t *v = realloc(b->blob, cap);
        if (!v) return MGPU_E_LIMIT;
        b->blob = v; b->capblob = cap;
    }
    memcpy(b->blob + b->nblob, blob, len);
    b->dd[b->ndd++] = (mbr_dd_t){docid, b->nblob, (uint32_t)len};
    b->nblob += len;
    if (docid > b->max_docid) b->max_docid = docid;
    return 0;
}

int mbr_add_num(mbridge_t *b, int i, double value, uint32_t docid) {
    if (!b || i < 0 || i >= b->num_numbers) return MGPU_E_ARG;
    mbr_numvec_t *nv = &b->num[i];
    if (nv->n && (nv->val[nv->n - 1] < value || (nv->val[nv->n - 1] == value && nv->doc[nv->n - 1] >= docid)))
        return MGPU_E_ARG;                                   /* not in num_dbi's cursor order */
    if (nv->n == nv->cap) {
        uint64_t cap = nv->cap ? nv->cap * 2 : 4096;
        double *v = realloc(nv->val, cap * sizeof(double));
        if (!v) return MGPU_E_LIMIT;
        nv->val = v;
        uint32_t *d = realloc(nv->doc, cap * sizeof(uint32_t));
        if (!d) return MGPU_E_LIMIT;
        nv->doc = d; nv->cap = cap;
    }
    nv->val[nv->n] = value; nv->doc[nv->n] = docid; nv->n++;
    return 0;
}

int mbr_set_used(mbridge_t *b, const uint32_t *docids, uint64_t n) {
    if (!b) return MGPU_E_ARG;
    for (uint64_t i = 0; i < n; i++) if (docids[i] == 0 || (i && docids[i] <= docids[i - 1])) return MGPU_E_ARG;
    free(b->used);
    b->used = malloc((n ? n : 1) * sizeof(uint32_t));
    if (!b->used) return MGPU_E_LIMIT;
    if (n) memcpy(b->used, docids, n * sizeof(uint32_t));
    b->nused = n;
    if (n && docids[n - 1] > b->max_docid) b->max_docid = docids[n - 1];
    return 0;
}

static int cmp_post_id(const void *a, const void *b) {
    const mbr_post_t *x = a, *y = b;
    if (x->prio != y->prio) return x->prio < y->prio ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

/* CSR indexed by id 1..count ([count + 2] offsets, the layout of mshard_t's wp / tp / tw arrays) */
static void csr_by_id(const mbridge_t *b, const mbr_postvec_t *pv, uint32_t count, uint64_t **off_out, uint32_t **doc_out) {
    uint64_t *off = calloc((size_t)count + 2, sizeof(uint64_t));
    uint64_t total = 0;
    for (size_t i = 0; i < pv->n; i++) total += pv->v[i].n;
    uint32_t *doc = malloc((total ? total : 1) * sizeof(uint32_t));
    /* lengths at off[id + 1], then prefix sums */
    for (size_t i = 0; i < pv->n; i++) off[pv->v[i].id + 1] += pv->v[i].n;
    for (uint32_t w = 1; w <= count + 1; w++) off[w] += off[w - 1];
    /* a key stored twice cannot happen in LMDB; the last one wins the slot */
    for (size_t i = 0; i < pv->n; i++)
        memcpy(doc + off[pv->v[i].id], b->pool.v + pv->v[i].off, pv->v[i].n * sizeof(uint32_t));
    *off_out = off; *doc_out = doc;
}

mshard_t *mbr_finish(mbridge_t *b) {
    if (!b) return NULL;
    mshard_t *s = calloc(1, sizeof(*s));
    if (!s) { mbr_free(b); return NULL; }
    s->num_numbers = b->num_numbers; s->num_bools = b->num_bools; s->num_facets = b->num_facets;
    s->num_strings = 0; s->dict = NULL;
    /* postings may name docids beyond the last docdata / used docid only in a corrupt store; size by what was seen */
    s->max_docid = b->max_docid;
    s->word_count = b->word_count; s->tword_count = b->tword_count;
    csr_by_id(b, &b->post[MBR_WID], b->word_count, &s->wp_off, &s->wp_doc);
    csr_by_id(b, &b->post[MBR_TWID], b->tword_count, &s->tp_off, &s->tp_doc);
    csr_by_id(b, &b->post[MBR_TWID_WIDS], b->tword_count, &s->tw_off, &s->tw_wid);
    /* facets: per field, ids ascending */
    mbr_postvec_t *fv = &b->post[MBR_FACET];
    qsort(fv->v, fv->n, sizeof(mbr_post_t), cmp_post_id);
    size_t at = 0;
    for (int f = 0; f < s->num_facets; f++) {
        size_t lo = at;
        uint64_t total = 0;
        while (at < fv->n && fv->v[at].prio == (uint32_t)f) { total += fv->v[at].n; at++; }
        uint32_t cnt = (uint32_t)(at - lo);
        s->nfid[f] = cnt;
        s->fid[f] = malloc((cnt ? cnt : 1) * sizeof(uint32_t));
        s->fp_off[f] = calloc((size_t)cnt + 1, sizeof(uint64_t));
        s->fp_doc[f] = malloc((total ? total : 1) * sizeof(uint32_t));
        uint64_t o = 0;
        for (uint32_t k = 0; k < cnt; k++) {
            const mbr_post_t *p = &fv->v[lo + k];
            s->fid[f][k] = p->id; s->fp_off[f][k] = o;
            memcpy(s->fp_doc[f] + o, b->pool.v + p->off, p->n * sizeof(uint32_t));
            o += p->n;
        }
        s->fp_off[f][cnt] = o;
    }
    /* bools */
    for (size_t i = 0; i < b->post[MBR_BOOL].n; i++) {
        const mbr_post_t *p = &b->post[MBR_BOOL].v[i];
        free(s->bp_doc[p->prio][p->id]);
        s->bp_doc[p->prio][p->id] = malloc((p->n ? p->n : 1) * sizeof(uint32_t));
        memcpy(s->bp_doc[p->prio][p->id], b->pool.v + p->off, p->n * sizeof(uint32_t));
        s->bp_n[p->prio][p->id] = p->n;
    }
    /* numbers: ownership moves */
    for (int i = 0; i < s->num_numbers; i++) {
        s->np_n[i] = b->num[i].n;
        s->np_val[i] = b->num[i].val ? b->num[i].val : malloc(sizeof(double));
        s->np_doc[i] = b->num[i].doc ? b->num[i].doc : malloc(sizeof(uint32_t));
        b->num[i].val = NULL; b->num[i].doc = NULL;
    }
    /* docdata: blobs in docid order */
    s->dd_off = calloc((size_t)s->max_docid + 2, sizeof(uint64_t));
    s->dd = malloc(b->nblob ? b->nblob : 1);
    {
        uint32_t *len = calloc((size_t)s->max_docid + 2, sizeof(uint32_t));
        size_t *src = calloc((size_t)s->max_docid + 2, sizeof(size_t));
        for (size_t i = 0; i < b->ndd; i++) { len[b->dd[i].docid] = b->dd[i].len; src[b->dd[i].docid] = i + 1; }
        uint64_t o = 0;
        for (uint32_t d = 0; d <= s->max_docid; d++) {
            s->dd_off[d] = o;
            if (src[d]) { memcpy(s->dd + o, b->blob + b->dd[src[d] - 1].off, len[d]); o += len[d]; }
        }
        s->dd_off[s->max_docid + 1] = o;
        free(len); free(src);
    }
    if (b->used) { s->used_doc = b->used; s->used_n = b->nused; s->has_used = 1; b->used = NULL; }
    mbr_free(b);
    return s;
}
