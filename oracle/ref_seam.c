/* ref_seam.c -- drives the reference's OWN index-writing code on the data formats either side of the query path.
 * TEST INFRASTRUCTURE ONLY (see orc.h): built by `make ref-seam` into oracle/_ref/ref_seam, only where /root/reference
 * exists.  It compiles the reference's src/main/sindex.c from where it lies by including it (the functions wanted are
 * static there): gen_wordpos_data + store_docdata (sindex.c:273-326, 409-496) write docid2data blobs into a real LMDB
 * environment, index_number (sindex.c:527-565) fills num_dbi under the reference's comparator and flags
 * (sindex.c:61-64, 1116-1118).  What LMDB then holds is printed as hex for tests/golden/make_docdata_fixtures.py.
 * Symbols of the reference this program never calls (jansson, analyzer, dtrie ...) are left unresolved by the link.
 *
 * input (stdin), one case after another:
 *   C <num_numbers> <num_facets>        start a case (one shard, fresh LMDB environment in argv[1])
 *   D <docid>                           start a document
 *   N <field> <value>                   number of the document (also indexed into num_dbi[field])
 *   F <field> <facet id>                facet id of the document
 *   W <wid> <priority> <position>       one word occurrence
 *   E                                   end of document: gen_wordpos_data + store_docdata
 *   X                                   end of case: commit, print
 * output: "blob <docid> <hex>" per document, "num <field> <f64 bits hex> <docid>" in cursor order, "end".
 */
#include "sindex.c"          /* -I/root/reference/src/main */
#include <sys/stat.h>

static void hex(const uint8_t *p, size_t n) { for (size_t i = 0; i < n; i++) printf("%02x", p[i]); }

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_seam <scratch dir> < cases\n"); return 2; }
    char line[256];
    struct sindex si; struct write_cache wc; struct mapping map;
    int ncase = 0, open_case = 0;
    uint32_t docids[4096]; int ndoc = 0;
    memset(&si, 0, sizeof(si)); memset(&wc, 0, sizeof(wc)); memset(&map, 0, sizeof(map));
    while (fgets(line, sizeof(line), stdin)) {
        if (line[0] == 'C') {
            int nn, nf;
            if (sscanf(line + 1, "%d %d", &nn, &nf) != 2 || nn > MAX_FIELDS || nf > MAX_FIELDS) return 3;
            memset(&si, 0, sizeof(si)); memset(&wc, 0, sizeof(wc)); memset(&map, 0, sizeof(map));
            map.num_numbers = nn; map.num_facets = nf;
            si.map = &map; si.wc = &wc;
            char path[512];
            snprintf(path, sizeof(path), "%s/case%d", argv[1], ncase++);
            mkdir(path, 0775);
            /* sindex_new's environment set-up (sindex.c:1225-1242) for the databases this program writes */
            if (mdb_env_create(&si.env) || mdb_env_set_mapsize(si.env, 64u << 20) || mdb_env_set_maxdbs(si.env, 64) ||
                mdb_env_open(si.env, path, MDB_NORDAHEAD | MDB_NOSYNC | MDB_NOMETASYNC, 0664)) return 4;
            if (mdb_txn_begin(si.env, NULL, 0, &si.txn)) return 4;
            if (mdb_dbi_open(si.txn, DBI_DOCID2DATA, MDB_CREATE, &si.docid2data_dbi)) return 4;
            if (mdb_dbi_open(si.txn, DBI_IDNUM2DBL, MDB_CREATE, &si.idnum2dbl_dbi)) return 4;
            for (int i = 0; i < nn; i++) {
                char dbi_name[32];
                snprintf(dbi_name, sizeof(dbi_name), "num_dbi_%d", i);
                /* flags and comparator of sindex.c:1116-1118 */
                if (mdb_dbi_open(si.txn, dbi_name, MDB_CREATE | MDB_DUPSORT | MDB_DUPFIXED | MDB_INTEGERDUP, &si.num_dbi[i])) return 4;
                mdb_set_compare(si.txn, si.num_dbi[i], double_compare);
            }
            wc.kh_idnum2dbl = kh_init(IDNUM2DBL);
            wc.od.num_data = calloc(nn ? nn : 1, sizeof(double));
            wc.od.facet_data = calloc(nf ? nf : 1, sizeof(*wc.od.facet_data));
            ndoc = 0; open_case = 1;
        } else if (line[0] == 'D' && open_case) {
            unsigned d;
            if (sscanf(line + 1, "%u", &d) != 1) return 3;
            wc.od.docid = d;
            sindex_docdata_init(&si);                        /* sindex.c:256-266 */
            if (ndoc < 4096) docids[ndoc++] = d;
        } else if (line[0] == 'N' && open_case) {
            int f; double v;
            if (sscanf(line + 1, "%d %lf", &f, &v) != 2 || f < 0 || f >= map.num_numbers) return 3;
            wc.od.num_data[f] = v;
            index_number(&si, f, v);                         /* sindex.c:527-565 */
        } else if (line[0] == 'F' && open_case) {
            int f; unsigned id;
            if (sscanf(line + 1, "%d %u", &f, &id) != 2 || f < 0 || f >= map.num_facets) return 3;
            kv_push(uint32_t, wc.od.facet_data[f], id);
        } else if (line[0] == 'W' && open_case) {
            unsigned wid, pos; int pri;
            if (sscanf(line + 1, "%u %d %u", &wid, &pri, &pos) != 3) return 3;
            /* what the analyzer callback records per occurrence (sindex.c: word position list + per-document count) */
            wid_pos_t *wp = malloc(sizeof(*wp));
            wp->wid = wid; wp->priority = pri; wp->position = pos;
            kv_push(wid_pos_t *, wc.od.kv_widpos, wp);
            int ret = 0;
            khiter_t k = kh_put(UNIQWID, wc.od.kh_uniqwid, wid, &ret);
            if (ret) kh_value(wc.od.kh_uniqwid, k) = 1; else kh_value(wc.od.kh_uniqwid, k)++;
        } else if (line[0] == 'E' && open_case) {
            sindex_store_docdata(&si);                       /* sindex.c:504-522 */
        } else if (line[0] == 'X' && open_case) {
            if (mdb_txn_commit(si.txn)) return 5;
            MDB_txn *rt;
            if (mdb_txn_begin(si.env, NULL, MDB_RDONLY, &rt)) return 5;
            for (int i = 0; i < ndoc; i++) {
                MDB_val key = { sizeof(uint32_t), &docids[i] }, data;
                if (mdb_get(rt, si.docid2data_dbi, &key, &data)) { printf("blob %u -\n", docids[i]); continue; }
                printf("blob %u ", docids[i]); hex(data.mv_data, data.mv_size); printf("\n");
            }
            for (int f = 0; f < map.num_numbers; f++) {
                mdb_set_compare(rt, si.num_dbi[f], double_compare);
                MDB_cursor *cur; MDB_val key, data;
                if (mdb_cursor_open(rt, si.num_dbi[f], &cur)) continue;
                for (MDB_cursor_op op = MDB_FIRST; mdb_cursor_get(cur, &key, &data, op) == 0; op = MDB_NEXT) {
                    uint64_t bits; uint32_t d;
                    memcpy(&bits, key.mv_data, 8); memcpy(&d, data.mv_data, 4);
                    printf("num %d %016llx %u\n", f, (unsigned long long)bits, d);
                }
                mdb_cursor_close(cur);
            }
            mdb_txn_abort(rt);
            mdb_env_close(si.env);
            printf("end\n");
            open_case = 0;
        }
    }
    return 0;
}
