/* query_gpu.c -- drop-in replacement for Marlin's src/main/query.c on MI355X: the same file with another execute_query.
 *
 * The reference's execute_query (query.c:426-484) hands one execute_squery job per shard to search_pool and waits; with
 * shim/squery_gpu.c behind execute_squery that is one GPU batch of ONE squery per pool thread.  This file keeps
 * everything else of query.c -- form_result, get_facet_results, hit materialisation, generate_query_terms, query_new /
 * query_free -- by COMPILING THE REFERENCE'S OWN query.c (included from where it lies, its execute_query renamed) and
 * defines execute_query so that every shard of the query goes to the device as one batch
 * (mgpu_shim_execute_squeries, squery_gpu.c): no pool threads, no worker wake-ups, one set of launches per GPU.
 * form_result is still the reference's.
 *
 * mgpu_shim_execute_queries is the same for the queries of a {"requests": [...]} body (index.c:1187-1209): all shards of
 * all queries in one batch; INTEGRATION.md shows the three-line change in index_query_callback.
 *
 * Build: `make shim` (build container only).  Link a Marlin build with query_gpu.o + squery_gpu.o in place of query.o +
 * squery.o, plus -lmarlin_host -lmarlin_gpu.
 */
#define execute_query marlin_reference_execute_query
#include "query.c"                    /* -I/root/reference/src/main: the reference's file, as it is */
#undef execute_query

void mgpu_shim_execute_squeries(struct squery *sq, int n);

static struct squery *fan_out(struct query *q) {
    struct index *in = q->in;
    struct squery *sq = calloc(in->num_shards, sizeof(struct squery));
    for (int i = 0; i < in->num_shards; i++) {
        sq[i].q = q;
        sq[i].worker = NULL;                  /* nobody waits on a condition variable: the call below returns when all are done */
        sq[i].shard_idx = i;
        sq[i].shard = kv_A(in->shards, i);
        sq[i].sqres = NULL;
    }
    return sq;
}

static char *finish_query(struct query *q, struct squery *sq, struct timeval *start) {
    struct index *in = q->in;
    json_t *j = form_result(q, sq);                           /* the reference's merge, query.c:312-422 */
    for (int i = 0; i < in->num_shards; i++) sqresult_free(q, sq[i].sqres);
    free(sq);
    struct timeval stop;
    gettimeofday(&stop, NULL);
    float took = timedifference_msec(*start, stop);
    json_object_set_new(j, J_R_TOOK, json_integer(took + 0.99));
    char *response = json_dumps(j, JSON_PRESERVE_ORDER | JSON_COMPACT);
    json_decref(j);
    return response;
}

char *execute_query(struct query *q) {
    struct timeval start;
    gettimeofday(&start, NULL);
    struct squery *sq = fan_out(q);
    mgpu_shim_execute_squeries(sq, q->in->num_shards);
    return finish_query(q, sq, &start);
}

/* n parsed queries (of one index or of several) as ONE device batch; responses[i] is what execute_query(qs[i]) returns */
void mgpu_shim_execute_queries(struct query **qs, int n, char **responses) {
    struct timeval start;
    gettimeofday(&start, NULL);
    int total = 0;
    for (int i = 0; i < n; i++) total += qs[i]->in->num_shards;
    struct squery *all = calloc(total ? total : 1, sizeof(struct squery));
    struct squery **per = calloc(n ? n : 1, sizeof(*per));
    int at = 0;
    for (int i = 0; i < n; i++) {
        per[i] = fan_out(qs[i]);
        memcpy(&all[at], per[i], sizeof(struct squery) * qs[i]->in->num_shards);
        at += qs[i]->in->num_shards;
    }
    mgpu_shim_execute_squeries(all, total);
    at = 0;
    for (int i = 0; i < n; i++) {
        memcpy(per[i], &all[at], sizeof(struct squery) * qs[i]->in->num_shards);
        at += qs[i]->in->num_shards;
        responses[i] = finish_query(qs[i], per[i], &start);
    }
    free(per);
    free(all);
}
