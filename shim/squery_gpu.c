/* squery_gpu.c -- drop-in replacement for Marlin's src/main/squery.c on MI355X.
 *
 * Compiled against the reference's own headers (never against copies): it exports the two symbols that are the seam
 * of the shard query path, squery.h:59-60,
 *      void execute_squery(void *w);
 *      void sqresult_free(struct query *q, struct squery_result *sqres);
 * with the reference's calling convention (query.c:446 threadpool job, query.c:461 inline call, query.c:468-470
 * free), and two set-up calls a maintainer adds where a shard is opened / after a bulk commit (sindex.c:246):
 *      void mgpu_shim_execute_squeries(struct squery *sq, int n);  -- n (query, shard) pairs as one GPU batch (query_gpu.c)
 *      int  mgpu_shim_attach_shard(struct shard *s, int device);   -- mirror the shard's LMDB into HBM
 *      int  mgpu_shim_refresh_shard(struct shard *s);              -- after a commit: rebuild + swap if LMDB moved on
 *      void mgpu_shim_detach_shard(struct shard *s);
 * What stays on the host is what north_star keeps there: the exact descents of the dtrie (twid / no-typo lookups,
 * dtrie_lookup_exact), LMDB, filter and agg parsing.  The typo walk of dtrie_lookup_term (dtrie.c:763-791) runs on the
 * device against a flat copy of the shard's dictionary (row f-1: mgpu_lookup_terms).  Everything between lookup_terms and the filled squery_result runs in
 * libmarlin_gpu.so through the host library (mplan_from_terms -> msq_execute_plan).
 *
 * There is NO CPU fallback in here: without an attached mirror, or on a GPU error, the shard reports zero hits and
 * logs with M_ERR, the reference's own convention for degraded results (squery.h has no error channel).
 *
 * Build: `make shim` (build container only; needs /root/reference for the headers).  Link a Marlin build with this
 * object in place of squery.o, plus -lmarlin_host -lmarlin_gpu.
 */
#include "squery.h"
#include "workers.h"
#include "debug.h"
#include "docrank.h"
#include "utils.h"
#include "sindex.h"
#include "sdata.h"
#include "filter.h"
#include "bmap.h"
#include "mbmap.h"
#include "aggs.h"
#include "metric-aggs.h"
#include "farmhash-c.h"

#include "mplan.h"       /* marlin_amd/host: m_query_t, mplan_from_terms, msq_execute_plan, mmirror_t */
#include "mbridge.h"     /* marlin_amd/host: mshard_t filled from a foreign store */

/* ---- the layouts this file moves by pointer must be the reference's */
_Static_assert(sizeof(struct docrank) == sizeof(m_docrank_t) && sizeof(struct docrank) == sizeof(mgpu_rank_t), "docrank layout");
_Static_assert(offsetof(struct docrank, docid) == offsetof(m_docrank_t, docid) && offsetof(struct docrank, shard_id) == offsetof(m_docrank_t, shard_id) &&
               offsetof(struct docrank, position) == offsetof(m_docrank_t, position) && offsetof(struct docrank, typos) == offsetof(m_docrank_t, typos), "docrank fields");
_Static_assert(sizeof(facet_count_t) == sizeof(m_facet_count_t) && offsetof(facet_count_t, facet_id) == offsetof(m_facet_count_t, facet_id) &&
               offsetof(facet_count_t, count) == offsetof(m_facet_count_t, count), "facet_count layout");
_Static_assert((int)R_TYPO == MR_TYPO && (int)R_PROX == MR_PROX && (int)R_POS == MR_POS && (int)R_EXACT == MR_EXACT && (int)R_FIELD == MR_FIELD &&
               (int)R_GEO == MR_GEO && (int)R_DONE == MR_DONE && (int)R_COMP == MR_COMP && (int)R_COMP_ASC == MR_COMP_ASC && (int)R_MAX == MR_MAX, "SORT_RULE values");
_Static_assert((int)F_AND == MF_AND && (int)F_OR == MF_OR && (int)F_NIN == MF_NIN && (int)F_EQ == MF_EQ && (int)F_NE == MF_NE && (int)F_GT == MF_GT &&
               (int)F_GTE == MF_GTE && (int)F_LT == MF_LT && (int)F_LTE == MF_LTE && (int)F_RANGE == MF_RANGE, "FILTER_TYPE values");
_Static_assert((int)AGG_MAX == MA_MAX && (int)AGG_MIN == MA_MIN && (int)AGG_AVG == MA_AVG && (int)AGG_STATS == MA_STATS && (int)AGG_CARDINALITY == MA_CARDINALITY, "AGG_TYPE values");
_Static_assert((int)PREFIX_LAST == MPREFIX_LAST && (int)PREFIX_NONE == MPREFIX_NONE && (int)TYPO_OK == MTYPO_OK, "query_cfg enums");
_Static_assert(R_MAX + 1 <= M_MAX_RULES, "rank_rule length");

/* ------------------------------------------------------------------ attached mirrors */
#define SHIM_MAX_SHARDS 256
#define SHIM_CTX_PER_DEVICE 16
/* One generation of a shard's mirror.  Queries take a reference under the table's read lock (held for the lookup only,
 * not for the GPU call) and drop it when their results are copied out; a refresh swaps the table entry under the write
 * lock and retires the old generation, which is destroyed by whoever drops the last reference. */
struct shim_mirror { mshard_t *src; mmirror_t *mirror; int device; size_t txnid; int refs, retired; };
struct shim_entry { struct shard *shard; struct shim_mirror *m; };
static struct shim_entry g_shards[SHIM_MAX_SHARDS];
static int g_nshards;                                   /* entries in use are below this index */
static pthread_rwlock_t g_shards_lock = PTHREAD_RWLOCK_INITIALIZER;
static pthread_mutex_t g_ref_lock = PTHREAD_MUTEX_INITIALIZER;

static void mirror_destroy(struct shim_mirror *m) {
    if (!m) return;
    mmirror_destroy(m->mirror);
    mshard_free(m->src);
    free(m);
}
static struct shim_mirror *mirror_acquire(const struct shard *s) {
    struct shim_mirror *m = NULL;
    pthread_rwlock_rdlock(&g_shards_lock);
    for (int i = 0; i < g_nshards; i++) if (g_shards[i].shard == s) { m = g_shards[i].m; break; }
    if (m) { pthread_mutex_lock(&g_ref_lock); m->refs++; pthread_mutex_unlock(&g_ref_lock); }
    pthread_rwlock_unlock(&g_shards_lock);
    return m;
}
static void mirror_release(struct shim_mirror *m) {
    if (!m) return;
    pthread_mutex_lock(&g_ref_lock);
    const int last = --m->refs == 0 && m->retired;
    pthread_mutex_unlock(&g_ref_lock);
    if (last) mirror_destroy(m);
}
static void mirror_retire(struct shim_mirror *m) {        /* after the table no longer points to it */
    if (!m) return;
    pthread_mutex_lock(&g_ref_lock);
    m->retired = 1;
    const int idle = m->refs == 0;
    pthread_mutex_unlock(&g_ref_lock);
    if (idle) mirror_destroy(m);
}
/* execution contexts (one HIP stream + workspace each), handed out per call: execute_squery runs on search_pool
 * threads and on h2o threads alike (query.c:446,461) */
static struct { mgpu_ctx_t *ctx[SHIM_CTX_PER_DEVICE]; int busy[SHIM_CTX_PER_DEVICE]; } g_ctx[64];
static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_ctx_free = PTHREAD_COND_INITIALIZER;

static mgpu_ctx_t *ctx_acquire(int device, int *slot) {
    mgpu_ctx_t *c = NULL;
    pthread_mutex_lock(&g_ctx_lock);
    for (;;) {
        for (int i = 0; i < SHIM_CTX_PER_DEVICE; i++) {
            if (g_ctx[device].busy[i]) continue;
            if (!g_ctx[device].ctx[i] && mgpu_ctx_create(device, 0, &g_ctx[device].ctx[i])) {
                M_ERR("gpu squery: mgpu_ctx_create: %s", mgpu_last_error());
                pthread_mutex_unlock(&g_ctx_lock);
                return NULL;
            }
            g_ctx[device].busy[i] = 1; *slot = i; c = g_ctx[device].ctx[i];
            break;
        }
        if (c) break;
        pthread_cond_wait(&g_ctx_free, &g_ctx_lock);
    }
    pthread_mutex_unlock(&g_ctx_lock);
    return c;
}

static void ctx_release(int device, int slot) {
    pthread_mutex_lock(&g_ctx_lock);
    g_ctx[device].busy[slot] = 0;
    pthread_cond_signal(&g_ctx_free);
    pthread_mutex_unlock(&g_ctx_lock);
}

/* ---- LMDB -> bridge.  A posting dbi holds, per set, the header at key IDPRIORITY(id, priority) and one value per
 * container at key header + cid + 1 (mbmap.c:183-283): the low 16 bits of a key are 0 for headers only. */
struct used_iter { uint32_t *v; uint64_t n, cap; };
static void collect_used(uint32_t docid, void *data) {
    struct used_iter *u = data;
    if (u->n == u->cap) { u->cap = u->cap ? u->cap * 2 : 4096; u->v = realloc(u->v, u->cap * sizeof(uint32_t)); }
    u->v[u->n++] = docid;
}

static int bridge_posting_dbi(mbridge_t *mb, MDB_txn *txn, MDB_dbi dbi, mbr_kind kind) {
    MDB_cursor *cur;
    if (mdb_cursor_open(txn, dbi, &cur) != 0) return MGPU_E_ARG;
    MDB_val key, data;
    int rc = 0;
    /* a header names at most 65535 containers (u16 count): room for all of them */
    const uint8_t **cont = malloc(sizeof(*cont) * 65536);
    uint32_t *cont_len = malloc(sizeof(*cont_len) * 65536);
    if (!cont || !cont_len) { free(cont); free(cont_len); mdb_cursor_close(cur); return MGPU_E_LIMIT; }
    for (MDB_cursor_op op = MDB_FIRST; !rc && mdb_cursor_get(cur, &key, &data, op) == 0; op = MDB_NEXT) {
        uint64_t k;
        if (key.mv_size != sizeof(k)) continue;
        memcpy(&k, key.mv_data, sizeof(k));
        if (k & 0xFFFFu) continue;                                  /* a container value, fetched with its header */
        const uint32_t id = (uint32_t)(k >> 32), prio = (uint32_t)((k >> 16) & 0xFFFFu);
        /* Word, top-level-word and twid->wid sets are stored once for all fields (priority 0) and once per field
         * (priority = field + 1: string_new_word_pos, sindex.c:676-699); the query path reads the priority-0 ones only
         * (squery.c:39-46).  The boolean database also holds the dtrie's free-node maps under id 0xFFFFFFFF in another
         * format (dtrie.c:490-527): not postings. */
        if ((kind == MBR_WID || kind == MBR_TWID || kind == MBR_TWID_WIDS) && prio != 0) continue;
        if (kind == MBR_BOOL && id > 1) continue;
        if (data.mv_size < 2) continue;
        uint16_t num_c; memcpy(&num_c, data.mv_data, 2);
        if (data.mv_size < 2u + 2u * num_c) { rc = MGPU_E_ARG; break; }
        for (uint32_t i = 0; i < num_c; i++) {
            uint16_t cid; memcpy(&cid, (const uint8_t *)data.mv_data + 2 + 2 * i, 2);
            uint64_t ck = k + cid + 1;
            MDB_val ckey = { sizeof(ck), &ck }, cval;
            if (mdb_get(txn, dbi, &ckey, &cval) == 0) { cont[i] = cval.mv_data; cont_len[i] = (uint32_t)cval.mv_size; }
            else { cont[i] = NULL; cont_len[i] = 0; }              /* mbmap_load_bmap logs and carries on (mbmap.c:276-278) */
        }
        mgpu_mbmap_t stored = { data.mv_data, (uint32_t)data.mv_size, cont, cont_len };
        rc = mbr_add_posting(mb, kind, id, prio, &stored);
    }
    mdb_cursor_close(cur);
    free(cont); free(cont_len);
    return rc;
}

void mgpu_shim_detach_shard(struct shard *s) {
    struct shim_mirror *old = NULL;
    pthread_rwlock_wrlock(&g_shards_lock);
    for (int i = 0; i < g_nshards; i++) {
        if (g_shards[i].shard != s) continue;
        old = g_shards[i].m;
        memset(&g_shards[i], 0, sizeof(g_shards[i]));
        break;
    }
    pthread_rwlock_unlock(&g_shards_lock);
    mirror_retire(old);                                    /* queries that hold it finish on it */
}

/* Every word of the shard's dtrie as a flat list (the input of mgpu_dict_create): a depth-first walk over the mapped
 * nodes with the layout of dtrie.h:36-51 -- children sorted by character, the word id behind the child array when
 * has_wid is set (get_widpos, dtrie.c:24-28).  Runs under the trie's read lock. */
struct dict_out { uint32_t *chars, *off, *wid; uint32_t n, nchars, ccap, wcap; chr_t *path; uint32_t pcap; int oom; };
static void dict_walk(struct dtrie *dt, const struct dnode *n, uint32_t depth, struct dict_out *o) {
    if (o->oom) return;
    /* a node whose word id is 0 holds no word: every get_wid user skips it (dtrie.c:648-651, 742-746) */
    if (n->has_wid && depth && *(const uint32_t *)((const uint8_t *)n + sizeof(struct dnode) + n->num_child * sizeof(struct dnode_ptr))) {
        const uint32_t *widpos = (const uint32_t *)((const uint8_t *)n + sizeof(struct dnode) + n->num_child * sizeof(struct dnode_ptr));
        if (o->n + 1 >= o->wcap) {
            o->wcap = o->wcap ? o->wcap * 2 : 4096;
            o->off = realloc(o->off, sizeof(uint32_t) * ((size_t)o->wcap + 1)); o->wid = realloc(o->wid, sizeof(uint32_t) * o->wcap);
        }
        if (o->nchars + depth > o->ccap) {
            while (o->nchars + depth > o->ccap) o->ccap = o->ccap ? o->ccap * 2 : 65536;
            o->chars = realloc(o->chars, sizeof(uint32_t) * o->ccap);
        }
        if (!o->off || !o->wid || !o->chars) { o->oom = 1; return; }
        o->off[o->n] = o->nchars; o->wid[o->n] = *widpos; o->n++;
        for (uint32_t i = 0; i < depth; i++) o->chars[o->nchars++] = (uint32_t)o->path[i];
    }
    for (uint32_t c = 0; c < n->num_child; c++) {
        if (depth + 1 > o->pcap) { o->pcap = o->pcap ? o->pcap * 2 : 256; o->path = realloc(o->path, sizeof(chr_t) * o->pcap); if (!o->path) { o->oom = 1; return; } }
        o->path[depth] = n->child[c].c;
        dict_walk(dt, (const struct dnode *)(dt->map + n->child[c].nid.offset), depth + 1, o);
    }
}
static int attach_dict(struct dtrie *dt, int device, mgpu_dict_t **out) {
    struct dict_out o;
    memset(&o, 0, sizeof(o));
    pthread_rwlock_rdlock(&dt->trie_lock);
    dict_walk(dt, dt->root, 0, &o);
    pthread_rwlock_unlock(&dt->trie_lock);
    int rc = MGPU_E_LIMIT;
    if (!o.oom) {
        if (!o.off) o.off = calloc(1, sizeof(uint32_t));
        if (o.off) { o.off[o.n] = o.nchars; rc = mgpu_dict_create(device, o.chars, o.off, o.wid ? o.wid : o.off, o.n, out); }
    }
    free(o.chars); free(o.off); free(o.wid); free(o.path);
    return rc;
}

/* Every posting, docdata blob and num_dbi pair of a shard's LMDB environment, read inside one read transaction (a
 * consistent snapshot against the single writer, sindex.c:184-254) into the host-side source of a mirror. */
static int shim_read_shard(struct shard *s, mshard_t **src_out, size_t *txnid_out) {
    struct sindex *si = s->sindex;
    const struct mapping *map = si->map;
    if (!map) return MGPU_E_ARG;
    MDB_txn *txn;
    if (mdb_txn_begin(si->env, NULL, MDB_RDONLY, &txn) != 0) return MGPU_E_ARG;
    const size_t snap_txnid = mdb_txn_id(txn);                      /* the commit this snapshot reflects */
    pthread_rwlock_rdlock(&si->trie->trie_lock);               /* the writer bumps the counts under the write lock */
    const uint32_t word_count = si->trie->meta->word_count, tword_count = si->trie->meta->tword_count;
    pthread_rwlock_unlock(&si->trie->trie_lock);
    mbridge_t *mb = mbr_new(map->num_numbers, map->num_bools, map->num_facets, word_count, tword_count);
    int rc = mb ? 0 : MGPU_E_LIMIT;
    if (!rc) rc = bridge_posting_dbi(mb, txn, si->wid2bmap_dbi, MBR_WID);
    if (!rc) rc = bridge_posting_dbi(mb, txn, si->twid2bmap_dbi, MBR_TWID);
    if (!rc) rc = bridge_posting_dbi(mb, txn, si->twid2widbmap_dbi, MBR_TWID_WIDS);
    if (!rc) rc = bridge_posting_dbi(mb, txn, si->facetid2bmap_dbi, MBR_FACET);
    if (!rc) rc = bridge_posting_dbi(mb, txn, si->boolid2bmap_dbi, MBR_BOOL);
    if (!rc) {                                                      /* docid2data: key u32 docid (sindex.c:288-291) */
        MDB_cursor *cur; MDB_val key, data;
        if (mdb_cursor_open(txn, si->docid2data_dbi, &cur) == 0) {
            for (MDB_cursor_op op = MDB_FIRST; !rc && mdb_cursor_get(cur, &key, &data, op) == 0; op = MDB_NEXT) {
                uint32_t docid;
                if (key.mv_size != sizeof(docid)) continue;
                memcpy(&docid, key.mv_data, sizeof(docid));
                rc = mbr_add_docdata(mb, docid, data.mv_data, data.mv_size);
            }
            mdb_cursor_close(cur);
        }
    }
    for (int i = 0; !rc && i < map->num_numbers; i++) {             /* num_dbi[i]: f64 keys descending, dup-sorted u32 docids */
        MDB_cursor *cur; MDB_val key, data;
        if (mdb_cursor_open(txn, si->num_dbi[i], &cur) != 0) continue;
        for (MDB_cursor_op op = MDB_FIRST; !rc && mdb_cursor_get(cur, &key, &data, op) == 0; op = MDB_NEXT) {
            double v; uint32_t docid;
            if (key.mv_size != sizeof(v) || data.mv_size < sizeof(docid)) continue;
            memcpy(&v, key.mv_data, sizeof(v)); memcpy(&docid, data.mv_data, sizeof(docid));
            rc = mbr_add_num(mb, i, v, docid);
        }
        mdb_cursor_close(cur);
    }
    if (!rc) {                                                      /* shard_get_all_docids (shard.c:51-53) */
        struct used_iter u = { NULL, 0, 0 };
        bmap_iterate(s->sdata->used_bmap, collect_used, &u);
        rc = mbr_set_used(mb, u.v, u.n);
        free(u.v);
    }
    mdb_txn_abort(txn);
    if (rc) { mbr_free(mb); M_ERR("gpu squery: mirroring shard %d failed (%d)", s->shard_id, rc); return rc; }
    mshard_t *src = mbr_finish(mb);
    if (!src) return MGPU_E_LIMIT;
    *src_out = src; *txnid_out = snap_txnid;
    return 0;
}

/* Mirrors one shard (shim_read_shard + mmirror_create + the dictionary).  Replaces an earlier mirror of the same shard:
 * the new one is built NEXT to the live one and swapped in.  Returns 0 or an MGPU_E_* code. */
int mgpu_shim_attach_shard(struct shard *s, int device) {
    struct sindex *si = s->sindex;
    if (device < 0 || device >= 64) return MGPU_E_ARG;
    mshard_t *src = NULL; size_t snap_txnid = 0;
    int rc = shim_read_shard(s, &src, &snap_txnid);
    if (rc) return rc;
    mmirror_t *mirror = NULL;
    if ((rc = mmirror_create(src, device, &mirror))) {
        M_ERR("gpu squery: mmirror_create for shard %d: %s", s->shard_id, mgpu_last_error());
        mshard_free(src);
        return rc;
    }
    /* the shard's dictionary for the device-side typo walk (the bridged mshard_t carries no dictionary of its own) */
    if ((rc = attach_dict(si->trie, device, &mirror->gd))) {
        M_ERR("gpu squery: dictionary of shard %d: %s", s->shard_id, mgpu_last_error());
        mmirror_destroy(mirror); mshard_free(src);
        return rc;
    }
    struct shim_mirror *nm = calloc(1, sizeof(*nm));
    if (!nm) { mmirror_destroy(mirror); mshard_free(src); return MGPU_E_LIMIT; }
    nm->src = src; nm->mirror = mirror; nm->device = device; nm->txnid = snap_txnid;
    /* ONE write-locked section replaces the entry (or takes a free slot): a query either finds the old generation or
     * the new one, never none; the old one is retired after the lock is dropped and dies with its last query */
    struct shim_mirror *old = NULL;
    int at = -1;
    pthread_rwlock_wrlock(&g_shards_lock);
    for (int i = 0; i < g_nshards; i++) if (g_shards[i].shard == s) { at = i; break; }
    if (at < 0) for (int i = 0; i < SHIM_MAX_SHARDS; i++) if (!g_shards[i].shard) { at = i; break; }
    if (at >= 0) {
        old = g_shards[at].shard == s ? g_shards[at].m : NULL;
        g_shards[at].shard = s; g_shards[at].m = nm;
        if (at >= g_nshards) g_nshards = at + 1;
    }
    pthread_rwlock_unlock(&g_shards_lock);
    if (at < 0) { mirror_destroy(nm); return MGPU_E_LIMIT; }     /* table full: the shard keeps the mirror it had */
    mirror_retire(old);
    return 0;
}

/* ------------------------------------------------------------------ struct query -> m_query_t */
/* filter tree (filter.h:24-38) -> m_filter_t; leaves keep what filter_apply.c reads: type, field type, the schema's
 * priorities, numval / numval2 / numcmp1 / numcmp2, and farmhash32(strval) for facet strings (filter_apply.c:25) */
static m_filter_t *lower_filter_tree(const struct filter *f) {
    m_filter_t *o = mf_new((int)f->type);
    switch (f->field_type) {
        case F_NUMBER: case F_NUMLIST: o->field_type = MT_NUMBER; break;
        case F_STRING: case F_STRLIST: o->field_type = MT_STRING; break;
        case F_BOOLEAN: o->field_type = MT_BOOLEAN; break;
        default: o->field_type = MT_OTHER; break;
    }
    /* filter_apply's callbacks switch on F_NUMBER / F_STRING / F_BOOLEAN only (filter_apply.c:58-75) */
    if (f->field_type == F_NUMLIST || f->field_type == F_STRLIST) o->field_type = MT_OTHER;
    o->i_priority = f->s ? f->s->i_priority : -1;
    o->f_priority = f->s ? f->s->f_priority : -1;
    o->is_facet = f->s ? f->s->is_facet : 0;
    if (f->strval) o->facet_id = farmhash32(f->strval, strlen(f->strval));
    o->numval = f->numval; o->numval2 = f->numval2;
    o->numcmp1 = (int)f->numcmp1; o->numcmp2 = (int)f->numcmp2;
    for (size_t i = 0; i < kv_size(f->children); i++) mf_add_child(o, lower_filter_tree(kv_A(f->children, i)));
    return o;
}

/* the query's aggs (root with metric children, aggs.c:56-90) -> flat m_agg_t list in child order */
static int lower_aggs(const struct agg *root, m_query_t *mq) {
    if (!root) return 0;
    for (size_t i = 0; i < kv_size(root->children); i++) {
        const struct agg *c = kv_A(root->children, i);
        int field;
        switch (c->type) {
            case AGG_MAX: field = ((const struct agg_max *)c)->max_field; break;
            case AGG_MIN: field = ((const struct agg_min *)c)->min_field; break;
            case AGG_AVG: field = ((const struct agg_avg *)c)->field; break;
            case AGG_STATS: field = ((const struct agg_stats *)c)->field; break;
            case AGG_CARDINALITY: field = ((const struct agg_card *)c)->field; break;
            default: return MGPU_E_UNSUPPORTED;
        }
        if (mq_add_agg(mq, (int)c->type, field)) return MGPU_E_LIMIT;
    }
    return 0;
}

static m_query_t *lower_query(const struct query *q, int num_facets) {
    m_query_t *mq = mq_new(0);
    mq->num_words = q->num_words;
    mq->num_terms = (int)kv_size(q->terms);
    mq->cfg.prefix = (int)q->cfg.prefix; mq->cfg.typos = (int)q->cfg.typos;
    mq->cfg.hits_per_page = q->cfg.hits_per_page; mq->cfg.max_hits = q->cfg.max_hits;
    mq->cfg.max_facet_results = q->cfg.max_facet_results;
    mq->cfg.num_rules = q->cfg.num_rules;
    mq->cfg.rank_by = q->cfg.rank_by; mq->cfg.rank_sort = q->cfg.rank_sort; mq->cfg.rank_asc = q->cfg.rank_asc;
    mq->cfg.full_scan = q->cfg.full_scan; mq->cfg.full_scan_threshold = q->cfg.full_scan_threshold;
    for (int i = 0; i <= R_MAX && i < M_MAX_RULES; i++) { mq->cfg.rank_algo[i] = (int)q->cfg.rank_algo[i]; mq->rank_rule[i] = (int)q->rank_rule[i]; }
    for (int f = 0; f < num_facets && f < M_MAX_FIELDS; f++) mq->cfg.facet_enabled[f] = q->cfg.facet_enabled[f];
    mq->page_num = q->page_num;
    if (q->filter) mq->filter = lower_filter_tree(q->filter);
    return mq;
}

/* ------------------------------------------------------------------ m_sqresult_t -> struct squery_result */
static void copy_aggs(struct agg *root, const m_sqresult_t *r) {
    for (size_t i = 0; i < kv_size(root->children) && (int)i < r->num_aggs; i++) {
        struct agg *c = kv_A(root->children, i);
        const m_agg_t *a = &r->aggs[i];
        switch (c->type) {
            case AGG_MAX: ((struct agg_max *)c)->value = a->value; break;
            case AGG_MIN: ((struct agg_min *)c)->value = a->value; break;
            case AGG_AVG: ((struct agg_avg *)c)->sum = a->sum; ((struct agg_avg *)c)->count = a->count; break;
            case AGG_STATS: {
                struct agg_stats *st = (struct agg_stats *)c;
                st->sum = a->sum; st->count = a->count; st->min = a->min; st->max = a->max;
            } break;
            case AGG_CARDINALITY: {                 /* consume_card_agg: bmap_add per distinct facet id (metric-aggs.c:23-45) */
                struct agg_card *ac = (struct agg_card *)c;
                for (uint32_t x = 0; x < a->card_n; x++) bmap_add(ac->bmap, a->card_ids[x]);
            } break;
            default: break;
        }
    }
}

static void finish(struct squery *sq) {
    if (sq->worker) worker_done(sq->worker);
}

/* One (query, shard) pair on its way through the seam.  execute_squery is a batch of one; a replacement execute_query
 * (shim/query_gpu.c) hands over every shard of a query -- or of all the queries of a "requests" array -- at once, and
 * they go to the device as ONE batch of launches per GPU. */
struct sq_job {
    struct squery *sq;
    struct shim_mirror *gen;
    m_query_t *mq;
    m_termresult_t tr[M_MAX_QTERMS];
    uint32_t exact_wid[M_MAX_QWORDS];
    int num_terms, first_lookup, num_lookups, lookup_term[M_MAX_QTERMS];
    mplan_t plan;
    m_sqresult_t out;
    int rc, planned;
};

/* squery_result_new + init_facet_hash + aggs_dup (squery.c:20-37, 494-498), the mirror, and the host half of
 * lookup_terms (squery.c:144-167): short prefixes and terms without typos are exact descents of the dtrie
 * (dtrie.c:805-823) and run here; the typo walks are collected for the device */
static void job_begin(struct sq_job *j, struct squery *sq, mgpu_term_lookup_t **lk, int *nlk, int *caplk) {
    struct query *q = sq->q;
    const int num_facets = q->in->mapping->num_facets;
    memset(j, 0, sizeof(*j));
    j->sq = sq;
    j->num_terms = (int)kv_size(q->terms);
    struct squery_result *res = calloc(1, sizeof(*res));
    sq->sqres = res;
    sq->fast_rank = false;
    res->exact_docid_map = calloc(q->num_words ? q->num_words : 1, sizeof(struct bmap *));
    res->all_wordids = kh_init(WID2TYPOS);
    res->fh = calloc(num_facets ? num_facets : 1, sizeof(struct facet_hash));      /* .h stays NULL: no hashtable is built */
    res->fc = calloc(num_facets ? num_facets : 1, sizeof(struct facet_count *));
    if (q->agg) res->agg = aggs_dup(q->agg);

    j->gen = mirror_acquire(sq->shard);
    if (!j->gen) {
        M_ERR("gpu squery: shard %d has no mirror (mgpu_shim_attach_shard was not called)", sq->shard_idx);
        j->rc = MGPU_E_ARG;
        return;
    }
    if (j->num_terms > M_MAX_QTERMS || q->num_words > M_MAX_QWORDS) {
        M_ERR("gpu squery: query has %d words, limit %d", q->num_words, M_MAX_QWORDS);
        j->rc = MGPU_E_LIMIT;
        return;
    }
    struct sindex *si = sq->shard->sindex;
    mmirror_t *mirror = j->gen->mirror;
    j->first_lookup = *nlk;
    for (int t = 0; t < j->num_terms; t++) {
        term_t *tm = kv_A(q->terms, t);
        j->tr[t].twid = 0;
        m_u32map_init(&j->tr[t].wordids);
        const int walk = tm->typos && !(tm->prefix && tm->word->length <= LEVLIMIT) && mirror->gd &&
                         tm->word->length >= 1 && tm->word->length <= MGPU_MAX_TERM_LEN;
        if (walk) {
            if (*nlk == *caplk) {
                *caplk = *caplk ? *caplk * 2 : 64;
                *lk = realloc(*lk, sizeof(**lk) * (size_t)*caplk);
            }
            mgpu_term_lookup_t *l = &(*lk)[(*nlk)++];
            l->dict = mirror->gd; l->chars = (const uint32_t *)tm->word->chars; l->len = tm->word->length; l->prefix = tm->prefix;
            j->lookup_term[j->num_lookups++] = t;
            continue;
        }
        termresult_t *r = dtrie_lookup_term(si->trie, tm);
        j->tr[t].twid = r->twid;
        uint32_t wid; int dist;
        kh_foreach(r->wordids, wid, dist, { m_u32map_set(&j->tr[t].wordids, wid, dist); });
        termresult_free(r);
    }
    for (int i = 0; i < q->num_words; i++) j->exact_wid[i] = dtrie_lookup_exact(si->trie, kv_A(q->words, i));
}

/* struct query -> plan, once the device's term hits are in */
static void job_plan(struct sq_job *j) {
    struct query *q = j->sq->q;
    j->mq = lower_query(q, q->in->mapping->num_facets);
    if (!j->rc) j->rc = lower_aggs(q->agg, j->mq);
    if (!j->rc) { j->rc = mplan_from_terms(j->gen->mirror, j->mq, j->tr, j->exact_wid, 0, &j->plan); j->planned = 1; }
}

/* m_sqresult_t -> struct squery_result (what form_result and sqresult_free read), release, worker_done */
static void job_end(struct sq_job *j) {
    struct squery *sq = j->sq;
    struct query *q = sq->q;
    struct squery_result *res = sq->sqres;
    const int num_facets = q->in->mapping->num_facets;
    khiter_t k;                                              /* (kh_set assigns to a variable of this name) */
    if (j->rc) {
        M_ERR("gpu squery: shard %d: error %d %s", sq->shard_idx, j->rc, mgpu_last_error());
    } else {
        /* all_wordids as process_termresult leaves it (squery.c:56-60,100-105): wid -> distance of every matching word */
        for (uint32_t i = 0; i < j->plan.p.num_wids; i++)
            if (j->plan.wids[i].dist != 0xFF) kh_set(WID2TYPOS, res->all_wordids, j->plan.wids[i].wid, j->plan.wids[i].dist);
        res->num_hits = (int)j->out.num_hits;
        res->rank_count = j->out.rank_count;
        res->ranks = (struct docrank *)j->out.ranks; j->out.ranks = NULL;          /* identical layout, libc malloc */
        for (int f = 0; f < num_facets && f < M_MAX_FIELDS; f++) {
            if (!q->cfg.facet_enabled[f]) continue;
            res->fh[f].rcount = j->out.rcount[f];
            res->fc[f] = (struct facet_count *)j->out.fc[f]; j->out.fc[f] = NULL;  /* shard_id already stamped */
        }
        if (res->agg) copy_aggs(res->agg, &j->out);
        sq->fast_rank = j->out.fast_rank;
    }
    for (int t = 0; t < j->num_terms && t < M_MAX_QTERMS; t++) m_termresult_free(&j->tr[t]);
    m_sqresult_free(&j->out);
    if (j->planned) mplan_free(&j->plan);
    if (j->mq) mq_free(j->mq);
    mirror_release(j->gen);
    finish(sq);
}

/* n (query, shard) pairs through the seam together: the host lookups of every pair, ONE mgpu_lookup_terms and ONE
 * mgpu_execute per device for all of them (everything from get_matching_docids to the top-2N cut, squery.c:480-517),
 * then every squery_result filled.  The pairs may belong to one query (execute_query's fan-out, query.c:435-452) or to
 * several (the "requests" array, index.c:1187-1209). */
void mgpu_shim_execute_squeries(struct squery *sq, int n) {
    if (n <= 0) return;
    struct sq_job *jobs = calloc((size_t)n, sizeof(*jobs));
    mgpu_term_lookup_t *lk = NULL; int nlk = 0, caplk = 0;
    if (!jobs) { for (int i = 0; i < n; i++) finish(&sq[i]); return; }
    for (int i = 0; i < n; i++) job_begin(&jobs[i], &sq[i], &lk, &nlk, &caplk);
    /* devices in use; pairs of one device form one batch */
    for (int dev = 0; dev < 64; dev++) {
        int cnt = 0;
        for (int i = 0; i < n; i++) if (!jobs[i].rc && jobs[i].gen->device == dev) cnt++;
        if (!cnt) continue;
        int slot = 0, rc = 0;
        mgpu_ctx_t *ctx = ctx_acquire(dev, &slot);
        if (!ctx) rc = MGPU_E_HIP;
        /* the typo walks of this device's pairs: one call (the lookups of a pair are contiguous in lk) */
        if (!rc) {
            int tot = 0;
            for (int i = 0; i < n; i++) if (!jobs[i].rc && jobs[i].gen->device == dev) tot += jobs[i].num_lookups;
            if (tot) {
                mgpu_term_lookup_t *dl = malloc(sizeof(*dl) * (size_t)tot);
                int *owner = malloc(sizeof(int) * (size_t)tot), *term = malloc(sizeof(int) * (size_t)tot);
                int at = 0;
                for (int i = 0; i < n; i++) {
                    if (jobs[i].rc || jobs[i].gen->device != dev) continue;
                    for (int x = 0; x < jobs[i].num_lookups; x++) { dl[at] = lk[jobs[i].first_lookup + x]; owner[at] = i; term[at] = jobs[i].lookup_term[x]; at++; }
                }
                mgpu_term_hit_t *hits = NULL; uint32_t nh = 0;
                rc = mgpu_lookup_terms(ctx, dl, tot, &hits, &nh);
                for (uint32_t h = 0; !rc && h < nh; h++) {
                    if (hits[h].lookup >= (uint32_t)tot) { rc = MGPU_E_HIP; break; }
                    m_u32map_t *m = &jobs[owner[hits[h].lookup]].tr[term[hits[h].lookup]].wordids;
                    /* add_wordid_to_result (dtrie.c:633-640): least distance */
                    if ((int)hits[h].dist < m_u32map_get(m, hits[h].wid, 0xFF)) m_u32map_set(m, hits[h].wid, (int32_t)hits[h].dist);
                }
                mgpu_term_hits_free(hits);
                free(dl); free(owner); free(term);
            }
        }
        /* plans, then the batch */
        const mmirror_t **ms = malloc(sizeof(*ms) * (size_t)cnt);
        const m_query_t **qs = malloc(sizeof(*qs) * (size_t)cnt);
        const mplan_t **pls = malloc(sizeof(*pls) * (size_t)cnt);
        int *sidx = malloc(sizeof(int) * (size_t)cnt), *who = malloc(sizeof(int) * (size_t)cnt);
        m_sqresult_t *outs = calloc((size_t)cnt, sizeof(*outs));
        int m = 0;
        for (int i = 0; i < n; i++) {
            if (jobs[i].rc || jobs[i].gen->device != dev) continue;
            if (rc) { jobs[i].rc = rc; continue; }
            job_plan(&jobs[i]);
            if (jobs[i].rc) continue;
            ms[m] = jobs[i].gen->mirror; qs[m] = jobs[i].mq; pls[m] = &jobs[i].plan; sidx[m] = jobs[i].sq->shard_idx; who[m] = i; m++;
        }
        if (m) {
            rc = msq_execute_plans(ctx, ms, qs, pls, sidx, m, outs);
            for (int x = 0; x < m; x++) { if (rc) jobs[who[x]].rc = rc; else jobs[who[x]].out = outs[x]; }
        }
        if (ctx) ctx_release(dev, slot);
        free(ms); free(qs); free(pls); free(sidx); free(who); free(outs);
    }
    for (int i = 0; i < n; i++) job_end(&jobs[i]);
    free(lk);
    free(jobs);
}

void execute_squery(void *w) {
    mgpu_shim_execute_squeries((struct squery *)w, 1);
}

/* Row f-2: refresh keyed on the commit.  Call where a bulk write ends (sindex.c:246).  When the environment's last
 * committed transaction is newer than the one the mirror reflects, the shard is read again from a fresh snapshot and the
 * live mirror takes the DIFFERENCE as a delta (mmirror_apply_shard -> mgpu_shard_apply_delta: the new documents' records
 * appended into the mirror's headroom, the posting array re-laid and its window index recomputed on the device; the
 * dictionary mirror exported again only when the trie gained words).  The delta changes the mirror in place, so it runs
 * with the shard's entry write-locked once the queries that hold the mirror have finished: queries wait for milliseconds
 * instead of the device holding two mirrors for the seconds of a rebuild.  When the write was not an append (documents
 * deleted or replaced) or the headroom is used up, the mirror is rebuilt next to the live one and swapped in
 * (mgpu_shim_attach_shard), as before.  MGPU_SHIM_NO_DELTA=1 forces that path.
 * Returns 0 (up to date, or refreshed by a delta), 2 (refreshed by a rebuild), 1 when the shard was never attached, or
 * an MGPU_E_* code. */
int mgpu_shim_refresh_shard(struct shard *s) {
    struct shim_mirror *gen = mirror_acquire(s);
    if (!gen) return 1;
    const int device = gen->device; const size_t have = gen->txnid;
    mirror_release(gen);
    MDB_envinfo info;
    if (mdb_env_info(s->sindex->env, &info) != 0) return MGPU_E_ARG;
    if (info.me_last_txnid <= have) return 0;
    if (getenv("MGPU_SHIM_NO_DELTA")) { const int rc = mgpu_shim_attach_shard(s, device); return rc ? rc : 2; }
    mshard_t *src = NULL; size_t snap_txnid = 0;
    int rc = shim_read_shard(s, &src, &snap_txnid);
    if (rc) return rc;
    mgpu_dict_t *gd = NULL;
    int applied = 0;
    mshard_t *old_src = NULL;
    pthread_rwlock_wrlock(&g_shards_lock);                          /* no query can pick the mirror up from here on */
    struct shim_mirror *m = NULL;
    for (int i = 0; i < g_nshards; i++) if (g_shards[i].shard == s) { m = g_shards[i].m; break; }
    if (m) {
        for (;;) {                                                  /* ... and those that hold it run to their end */
            pthread_mutex_lock(&g_ref_lock);
            const int busy = m->refs;
            pthread_mutex_unlock(&g_ref_lock);
            if (!busy) break;
            struct timespec ts = { 0, 200000 };
            nanosleep(&ts, NULL);
        }
        const uint32_t words_before = m->src->word_count;
        rc = mmirror_apply_shard(m->mirror, src);
        if (!rc) {
            applied = 1;
            old_src = m->src; m->src = src; m->txnid = snap_txnid;
            if (src->word_count != words_before) {
                /* the bridged shard carries no dictionary: the trie's word list is exported again */
                if (!attach_dict(s->sindex->trie, device, &gd)) { mgpu_dict_destroy(m->mirror->gd); m->mirror->gd = gd; }
                else M_ERR("gpu squery: dictionary of shard %d after a delta: %s", s->shard_id, mgpu_last_error());
            }
        }
    }
    pthread_rwlock_unlock(&g_shards_lock);
    if (applied) { mshard_free(old_src); return 0; }
    mshard_free(src);
    if (!m) return 1;
    /* no headroom left / not an append are the expected reasons; after any other failure the live mirror may be
     * half-updated (include/marlin_gpu.h), so it is replaced all the same */
    if (rc != MGPU_E_LIMIT && rc != MGPU_E_UNSUPPORTED) {
        M_ERR("gpu squery: delta for shard %d failed (%d: %s): rebuilding the mirror", s->shard_id, rc, mgpu_last_error());
        mgpu_shim_detach_shard(s);             /* queries fail (and say so) until the new mirror is in, rather than read this one */
    }
    rc = mgpu_shim_attach_shard(s, device);
    return rc ? rc : 2;
}

void sqresult_free(struct query *q, struct squery_result *sqres) {
    if (!sqres) return;
    for (int i = 0; i < q->num_words; i++) if (sqres->exact_docid_map[i]) bmap_free(sqres->exact_docid_map[i]);
    kh_destroy(WID2TYPOS, sqres->all_wordids);
    free(sqres->exact_docid_map);
    free(sqres->ranks);
    for (int i = 0; i < q->in->mapping->num_facets; i++) {
        if (sqres->fh[i].h) hashtable_free(sqres->fh[i].h);
        free(sqres->fc[i]);
    }
    if (sqres->agg) sqres->agg->free(sqres->agg);
    free(sqres->fc);
    free(sqres->fh);
    free(sqres);
}
