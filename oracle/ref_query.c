/* ref_query.c -- the reference's OWN shard query path, run on a shard the reference's own indexing code built.
 * TEST INFRASTRUCTURE ONLY (see orc.h).  Built by `make ref-query` into oracle/_ref/ref_query (and, with the shim's
 * object in place of the reference's squery.c, into oracle/_ref/ref_query_gpu), only where /root/reference exists.
 *
 * Nothing of the reference is restated here.  This file includes src/main/sindex.c, aggs.c and metric-aggs.c from where
 * they lie (the functions wanted are static there) and links the reference's squery.c, docrank.c, filter_apply.c,
 * sort.c, hashtable.c, bmap.c, cont.c, mbmap.c, dtrie.c, query.c, the analyzer, utf8proc and LMDB as they are.  The one
 * thing it cannot use is the JSON front end (jansson needs a generated header this image lacks: DESIGN.md (c)), so a
 * document arrives as the sequence of leaf calls parse_index_document would make for it (sindex.c:792-893:
 * index_string, index_string_facet, index_number, index_number_facet, index_boolean) and a query as the fields
 * index_json_query fills in (index.c:1072-1156).  Unresolved symbols of code that never runs stay unresolved.
 *
 * stdin, one command a line (strings hex-encoded):
 *   I <dir> <shards> <strings> <numbers> <bools> <facets>   index + shards (sindex_new, sindex_set_mapping)
 *   W <shard>                        si_write_start
 *   D <docid>                        document: docid, sindex_docdata_init
 *   s <i_priority> <hex>             index_string            f <f_priority> <hex>    index_string_facet
 *   n <i_priority> <indexed> <val>   num_data + index_number g <f_priority> <val>    index_number_facet
 *   b <i_priority> <0|1>             index_boolean
 *   E                                sindex_store_docdata, docid into the shard's used bitmap
 *   C                                si_write_end
 *   Q                                new query;  t <hex> text (analyzed for search, generate_query_terms)
 *   c <hpp> <page> <maxhits> <maxfacets> <prefix> <typos> <rank_by> <rank_sort> <rank_asc> <full_scan> <threshold>
 *   r <n> <rule> ...                 rank_algo         e <hex mask>   facet_enabled
 *   F <depth> <type> <field_type> <i_priority> <f_priority> <is_facet> <numval> <numval2> <cmp1> <cmp2> <hex|->   filter node, preorder
 *   A <type> <field>                 aggregation
 *   R                                execute_squery on every shard, print, sqresult_free
 *   X                                (ref_query_gpu) mgpu_shim_attach_shard for every shard
 *   Y                                (ref_query_gpu) mgpu_shim_refresh_shard for every shard (after further W .. C blocks)
 * stdout per shard of a run:
 *   shard <i> hits <n> rank_count <n> fast <0|1>
 *   rank <docid> <typos> <proximity> <field> <position> <exact> <compare bits>      rank_count lines, in result order
 *   facet <f> <rcount> then  fc <facet_id> <count>  lines;   agg <i> <type> <value|sum> <count> <min> <max> <card>
 *   end
 */
#include "sindex.c"
#include "aggs.c"
#include "metric-aggs.c"
#include "analyzer.h"
#include "filter.h"
#include "filter_apply.h"
#include <sys/stat.h>

#ifdef REF_QUERY_GPU
int mgpu_shim_attach_shard(struct shard *s, int device);
void mgpu_shim_execute_squeries(struct squery *sq, int n);
int mgpu_shim_refresh_shard(struct shard *s);
#endif

static struct app g_app;
static struct index g_in;
static struct mapping g_map;
static struct query_cfg g_qcfg;
static struct shard *g_shard[64];
static int g_facet_enabled[MAX_FIELDS];

static char *unhex(const char *h) {
    size_t n = strlen(h) / 2;
    char *s = malloc(n + 1);
    for (size_t i = 0; i < n; i++) { unsigned v; sscanf(h + 2 * i, "%2x", &v); s[i] = (char)v; }
    s[n] = 0;
    return s;
}

static void word_cb(word_pos_t *wp, void *data) {          /* query_string_word_cb, index.c:909-917 */
    struct query *q = data;
    word_t *word = malloc(sizeof(word_t));
    word->length = wp->word.length;
    word->chars = malloc(sizeof(chr_t) * word->length);
    memcpy(word->chars, wp->word.chars, sizeof(chr_t) * word->length);
    q->num_words++;
    kv_push(word_t *, q->words, word);
}

static struct agg *metric(AGG_TYPE type, int field) {      /* what parse_*_agg leave behind (metric-aggs.c) */
    struct agg *a = NULL;
    switch (type) {
        case AGG_MAX: { struct agg_max *m = calloc(1, sizeof(*m)); a = &m->a; m->max_field = field; m->value = -DBL_MAX;
                        a->consume = consume_max_agg; a->as_json = max_agg_as_json; a->dup = max_agg_dup; a->merge = max_agg_merge; a->free = metric_agg_free; } break;
        case AGG_MIN: { struct agg_min *m = calloc(1, sizeof(*m)); a = &m->a; m->min_field = field; m->value = DBL_MAX;
                        a->consume = consume_min_agg; a->as_json = min_agg_as_json; a->dup = min_agg_dup; a->merge = min_agg_merge; a->free = metric_agg_free; } break;
        case AGG_AVG: { struct agg_avg *m = calloc(1, sizeof(*m)); a = &m->a; m->field = field;
                        a->consume = consume_avg_agg; a->as_json = avg_agg_as_json; a->dup = avg_agg_dup; a->merge = avg_agg_merge; a->free = metric_agg_free; } break;
        case AGG_STATS: { struct agg_stats *m = calloc(1, sizeof(*m)); a = &m->a; m->field = field; m->min = DBL_MAX; m->max = -DBL_MAX;
                        a->consume = consume_stats_agg; a->as_json = stats_agg_as_json; a->dup = stats_agg_dup; a->merge = stats_agg_merge; a->free = metric_agg_free; } break;
        case AGG_CARDINALITY: { struct agg_card *m = calloc(1, sizeof(*m)); a = &m->a; m->field = field; m->bmap = bmap_new();
                        a->consume = consume_card_agg; a->as_json = card_agg_as_json; a->dup = card_agg_dup; a->merge = card_agg_merge; a->free = card_agg_free; } break;
        default: return NULL;
    }
    a->kind = AGGK_METRIC; a->type = type;
    return a;
}

static void print_agg(int i, const struct agg *c) {
    unsigned long long bits[4] = {0, 0, 0, 0}; long count = 0; unsigned card = 0;
    double v[4] = {0, 0, 0, 0};
    switch (c->type) {
        case AGG_MAX: v[0] = ((const struct agg_max *)c)->value; break;
        case AGG_MIN: v[0] = ((const struct agg_min *)c)->value; break;
        case AGG_AVG: v[0] = ((const struct agg_avg *)c)->sum; count = ((const struct agg_avg *)c)->count; break;
        case AGG_STATS: { const struct agg_stats *s = (const struct agg_stats *)c; v[0] = s->sum; count = s->count; v[1] = s->min; v[2] = s->max; } break;
        case AGG_CARDINALITY: card = bmap_cardinality(((const struct agg_card *)c)->bmap); break;
        default: break;
    }
    for (int k = 0; k < 3; k++) memcpy(&bits[k], &v[k], 8);
    printf("agg %d %d %016llx %ld %016llx %016llx %u\n", i, (int)c->type, bits[0], count, bits[1], bits[2], card);
}

int main(void) {
    char *line = malloc(1 << 20);
    struct query *q = NULL;
    struct filter *fstack[16];
    int cur = -1;
    init_analyzers();
    init_filter_callbacks();                                 /* filter_apply.c's dispatch table */
    while (fgets(line, 1 << 20, stdin)) {
        size_t L = strlen(line);
        while (L && (line[L - 1] == '\n' || line[L - 1] == '\r')) line[--L] = 0;
        char *a = line + 1;
        switch (line[0]) {
        case 'I': {
            char dir[512]; int ns, a1, a2, a3, a4;
            if (sscanf(a, "%511s %d %d %d %d %d", dir, &ns, &a1, &a2, &a3, &a4) != 6 || ns > 64) return 3;
            memset(&g_in, 0, sizeof(g_in)); memset(&g_map, 0, sizeof(g_map)); memset(&g_qcfg, 0, sizeof(g_qcfg));
            snprintf(g_app.name, sizeof(g_app.name), "app"); snprintf(g_in.name, sizeof(g_in.name), "idx");
            g_in.app = &g_app; g_in.num_shards = ns; g_in.mapping = &g_map; g_in.cfg.qcfg = &g_qcfg;
            g_map.index = &g_in; g_map.num_strings = a1; g_map.num_numbers = a2; g_map.num_bools = a3; g_map.num_facets = a4;
            kv_init(g_in.shards);
            mkdir(dir, 0775);
            for (int i = 0; i < ns; i++) {
                struct shard *s = calloc(1, sizeof(*s));
                s->shard_id = (uint16_t)i; s->index = &g_in;
                snprintf(s->base_path, sizeof(s->base_path), "%s/s%d", dir, i);
                snprintf(s->idx_name, sizeof(s->idx_name), "index");
                mkdir(s->base_path, 0775);
                s->sdata = calloc(1, sizeof(struct sdata));
                s->sdata->used_bmap = bmap_new();
                s->sindex = sindex_new(s);
                sindex_set_mapping(s->sindex, &g_map);
                kv_push(struct shard *, g_in.shards, s);
                g_shard[i] = s;
            }
        } break;
        case 'W': cur = atoi(a); si_write_start(g_shard[cur]->sindex); break;
        case 'D': { struct sindex *si = g_shard[cur]->sindex; mdb_txn_renew(si->read_txn); si->wc->od.docid = (uint32_t)strtoul(a, NULL, 10); sindex_docdata_init(si); } break;
        case 's': { int p; char h[1 << 16]; if (sscanf(a, "%d %65535s", &p, h) == 2) { char *s = unhex(h); index_string(g_shard[cur]->sindex, s, p); free(s); } } break;
        case 'f': { int p; char h[1 << 16]; if (sscanf(a, "%d %65535s", &p, h) == 2) { char *s = unhex(h); index_string_facet(g_shard[cur]->sindex, s, p); free(s); } } break;
        case 'n': { int p, ix; double v; if (sscanf(a, "%d %d %lf", &p, &ix, &v) == 3) { struct sindex *si = g_shard[cur]->sindex; si->wc->od.num_data[p] = v; if (ix) index_number(si, p, v); } } break;
        case 'g': { int p; double v; if (sscanf(a, "%d %lf", &p, &v) == 2) index_number_facet(g_shard[cur]->sindex, v, p); } break;
        case 'b': { int p, v; if (sscanf(a, "%d %d", &p, &v) == 2) index_boolean(g_shard[cur]->sindex, p, v != 0); } break;
        case 'E': { struct sindex *si = g_shard[cur]->sindex; bmap_add(g_shard[cur]->sdata->used_bmap, si->wc->od.docid); sindex_store_docdata(si); mdb_txn_reset(si->read_txn); } break;
        case 'C': si_write_end(g_shard[cur]->sindex); break;
#ifdef REF_QUERY_GPU
        case 'X': for (int i = 0; i < g_in.num_shards; i++) { int rc = mgpu_shim_attach_shard(g_shard[i], 0); if (rc) { printf("attach %d failed %d\n", i, rc); fflush(stdout); return 5; } } printf("attached\n"); break;
#endif
#ifdef REF_QUERY_GPU
        case 'Y': for (int i = 0; i < g_in.num_shards; i++) { int rc = mgpu_shim_refresh_shard(g_shard[i]); printf("refresh %d %d\n", i, rc); } break;
#endif
        case 'Q':
            q = query_new(&g_in);
            q->page_num = 1;
            memset(g_facet_enabled, 0, sizeof(g_facet_enabled));
            q->cfg.facet_enabled = g_facet_enabled;
            q->cfg.rank_by = -1;
            q->text = strdup("");
            break;
        case 't': {
            free(q->text); q->text = unhex(a + 1);
            get_default_analyzer()->analyze_string_for_search(q->text, word_cb, q);        /* index.c:1150-1156 */
        } break;
        case 'c': {
            int hpp, page, mh, mfr, pre, typ, rb, rs, ra, fs; unsigned th;
            if (sscanf(a, "%d %d %d %d %d %d %d %d %d %d %u", &hpp, &page, &mh, &mfr, &pre, &typ, &rb, &rs, &ra, &fs, &th) != 11) return 3;
            q->cfg.hits_per_page = (uint16_t)hpp; q->page_num = (uint16_t)page; q->cfg.max_hits = (uint16_t)mh; q->cfg.max_facet_results = (uint16_t)mfr;
            q->cfg.prefix = (PREFIX_TYPE)pre; q->cfg.typos = (TYPO_TYPE)typ; q->cfg.rank_by = rb; q->cfg.rank_sort = rs; q->cfg.rank_asc = ra;
            q->cfg.full_scan = fs; q->cfg.full_scan_threshold = th;
        } break;
        case 'r': {
            int n = 0, off = 0, used = 0;
            sscanf(a, "%d%n", &n, &off);
            q->cfg.num_rules = n;
            for (int i = 0; i < n && i <= R_MAX; i++) { int r; if (sscanf(a + off, "%d%n", &r, &used) != 1) return 3; off += used; q->cfg.rank_algo[i] = (SORT_RULE)r; }
        } break;
        case 'e': { unsigned m = (unsigned)strtoul(a, NULL, 16); for (int f = 0; f < MAX_FIELDS && f < 32; f++) g_facet_enabled[f] = (m >> f) & 1; } break;
        case 'F': {
            int depth, type, ft, ip, fp, isf, c1, c2; double v1, v2; char h[4096];
            if (sscanf(a, "%d %d %d %d %d %d %lf %lf %d %d %4095s", &depth, &type, &ft, &ip, &fp, &isf, &v1, &v2, &c1, &c2, h) != 11 || depth > 15) return 3;
            struct filter *f = calloc(1, sizeof(*f));
            kv_init(f->children);
            struct schema *s = calloc(1, sizeof(*s));
            s->i_priority = ip; s->f_priority = fp; s->is_facet = isf; s->is_indexed = ip >= 0;
            f->s = s; f->type = (FILTER_TYPE)type; f->field_type = (F_TYPE)ft;
            f->numval = v1; f->numval2 = v2; f->numcmp1 = (FILTER_TYPE)c1; f->numcmp2 = (FILTER_TYPE)c2;
            if (strcmp(h, "-")) f->strval = unhex(h);
            fstack[depth] = f;
            if (depth == 0) q->filter = f; else kv_push(struct filter *, fstack[depth - 1]->children, f);
        } break;
        case 'A': {
            int type, field;
            if (sscanf(a, "%d %d", &type, &field) != 2) return 3;
            if (!q->agg) {                                   /* the head of parse_root_agg (aggs.c:57-65); its loop walks JSON */
                struct agg *root = calloc(1, sizeof(struct agg));
                root->kind = AGGK_ROOT; root->type = AGG_ROOT;
                root->consume = consume_root_agg; root->as_json = root_agg_as_json; root->dup = root_agg_dup;
                root->merge = root_agg_merge; root->free = root_agg_free;
                kv_init(root->children);
                q->agg = root;
            }
            struct agg *m = metric((AGG_TYPE)type, field);
            if (m) kv_push(struct agg *, q->agg->children, m);
        } break;
        case 'R': {
            /* the rank rule list as index_json_query builds it (index.c:1134-1148) */
            int rcount = 0;
            if (q->cfg.rank_by >= 0) {
                if (q->cfg.rank_sort) { q->rank_rule[0] = q->cfg.rank_asc ? R_COMP_ASC : R_COMP; q->rank_rule[q->cfg.num_rules + 1] = R_DONE; rcount++; }
                else { q->rank_rule[q->cfg.num_rules] = q->cfg.rank_asc ? R_COMP_ASC : R_COMP; q->rank_rule[q->cfg.num_rules + 1] = R_DONE; }
            } else q->rank_rule[q->cfg.num_rules] = R_DONE;
            memcpy(&q->rank_rule[rcount], q->cfg.rank_algo, q->cfg.num_rules * sizeof(SORT_RULE));
            generate_query_terms(q);
            struct squery sqs[64];
            memset(sqs, 0, sizeof(sqs));
            for (int i = 0; i < g_in.num_shards; i++) { sqs[i].q = q; sqs[i].shard_idx = i; sqs[i].shard = g_shard[i]; }
#ifdef REF_QUERY_GPU
            /* several shards: the entry a replacement execute_query uses, every shard of the query in one GPU batch */
            if (g_in.num_shards > 1) mgpu_shim_execute_squeries(sqs, g_in.num_shards);
            else execute_squery(&sqs[0]);
#else
            for (int i = 0; i < g_in.num_shards; i++) execute_squery(&sqs[i]);
#endif
            for (int i = 0; i < g_in.num_shards; i++) {
                struct squery sq = sqs[i];
                struct squery_result *r = sq.sqres;
                printf("shard %d hits %d rank_count %d fast %d\n", i, r->num_hits, r->rank_count, sq.fast_rank ? 1 : 0);
                for (int j = 0; j < r->rank_count; j++) {
                    const struct docrank *d = &r->ranks[j];
                    unsigned long long cb; memcpy(&cb, &d->compare, 8);
                    printf("rank %u %u %u %u %u %u %016llx\n", d->docid, d->typos, d->proximity, d->field, d->position, d->exact, cb);
                }
                for (int f = 0; f < g_map.num_facets; f++) {
                    if (!g_facet_enabled[f]) continue;
                    printf("facet %d %d\n", f, r->fh[f].rcount);
                    for (int x = 0; x < r->fh[f].rcount; x++) printf("fc %u %u\n", r->fc[f][x].facet_id, r->fc[f][x].count);
                }
                if (r->agg) for (size_t x = 0; x < kv_size(r->agg->children); x++) print_agg((int)x, kv_A(r->agg->children, x));
                printf("end\n");
                sqresult_free(q, r);
            }
            fflush(stdout);
            q->cfg.facet_enabled = NULL;
            /* (the query is left to the process: query_free would walk filter nodes whose strings are ours) */
            q = NULL;
        } break;
        default: break;
        }
    }
    return 0;
}
