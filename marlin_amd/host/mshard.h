/* mshard.h -- in-memory shard store: the host-side truth the HBM mirror is exported from.
 *
 * Stands in for one shard's LMDB environment (sindex.h:53-81).  It keeps the reference's
 * data formats where the hot path reads them:
 *   - docid2data blob, byte for byte (sindex.c:273-326 store_docdata, 409-496 gen_wordpos_data)
 *   - wid2bmap / twid2bmap / twid2widbmap / facetid2bmap / boolid2bmap as sorted docid (or wid)
 *     lists, i.e. the *sets* the roaring containers encode (mbmap.c:253-283)
 *   - num_dbi[i] as (value desc, docid asc) pairs (sindex.c:61-64, 528-535, 1116-1118)
 * The write path (sindex.c:1003-1091) is restated only as far as needed to build test and
 * benchmark corpora; it is not a storage engine.
 */
#ifndef MARLIN_AMD_MSHARD_H
#define MARLIN_AMD_MSHARD_H

#include "mcommon.h"
#include "mdict.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void (*m_word_cb)(const m_chr_t *chars, int length, int position, void *data);
void m_analyze_for_indexing(const char *str, m_word_cb cb, void *data);
void m_analyze_for_search(const char *str, m_word_cb cb, void *data);

/* Facet id hash.  The reference uses farmhash32() whose variant depends on the build
 * machine's SSE level and on NDEBUG (farmhash-c.c:1559-1573, 287-293); ids are opaque
 * 32-bit values to the hot path.  This is Google FarmHash's portable 32-bit "mk" variant
 * (farmhash-c.c:858-951) followed by the non-NDEBUG tweak the reference's release build
 * keeps (farmhash-c.c:287-293). */
uint32_t m_farmhash32_mk(const char *s, size_t len);
uint32_t m_facet_id(const char *s, size_t len);

typedef struct mshard {
    uint32_t max_docid;             /* docids are 1..max_docid (sdata.c:330) */
    int num_strings, num_numbers, num_bools, num_facets;
    mdict_t *dict;

    /* docid2data blobs */
    uint64_t *dd_off;               /* [max_docid + 2] */
    uint8_t  *dd;

    /* wid2bmap, key IDPRIORITY(wid, 0): docs holding the word in any field */
    uint32_t  word_count;
    uint64_t *wp_off;               /* [word_count + 2] */
    uint32_t *wp_doc;
    /* twid2bmap / twid2widbmap, key IDPRIORITY(twid, 0) */
    uint32_t  tword_count;
    uint64_t *tp_off;               /* [tword_count + 2] */
    uint32_t *tp_doc;
    uint64_t *tw_off;
    uint32_t *tw_wid;

    /* facetid2bmap, key IDPRIORITY(facet_id, f_priority) */
    uint32_t  nfid[M_MAX_FIELDS];
    uint32_t *fid[M_MAX_FIELDS];    /* sorted unique facet ids of field f */
    uint64_t *fp_off[M_MAX_FIELDS]; /* [nfid + 1] */
    uint32_t *fp_doc[M_MAX_FIELDS];
    /* facetid2str (sindex.c:157-169): sorted by id */
    uint32_t  nfstr;
    uint32_t *fstr_id;
    char    **fstr;

    /* boolid2bmap, key IDPRIORITY(0|1, i_priority) */
    uint64_t  bp_n[M_MAX_FIELDS][2];
    uint32_t *bp_doc[M_MAX_FIELDS][2];

    /* num_dbi[i]: (value, docid) sorted by value descending then docid ascending, no duplicates */
    uint64_t  np_n[M_MAX_FIELDS];
    double   *np_val[M_MAX_FIELDS];
    uint32_t *np_doc[M_MAX_FIELDS];

    /* sdata->used_bmap (shard.c:51-53) when the shard has holes (deleted documents): ascending docids in use;
     * has_used == 0 means every docid 1..max_docid is in use */
    int       has_used;
    uint64_t  used_n;
    uint32_t *used_doc;
} mshard_t;

typedef struct mshard_builder mshard_builder_t;

mshard_builder_t *msb_new(int num_strings, int num_numbers, int num_bools, int num_facets);
/* documents get docids 1, 2, 3 ... in call order */
uint32_t msb_begin_doc(mshard_builder_t *b);
/* index_string (sindex.c:729-736): analyze + string_new_word_pos */
void msb_add_string(mshard_builder_t *b, int i_priority, const char *str);
/* same effect as one analyzer callback (sindex.c:658-726) for an already tokenised word */
void msb_add_word(mshard_builder_t *b, int i_priority, const m_chr_t *chars, int length, int position);
/* F_NUMBER / F_NUMLIST element (sindex.c:827-859) */
void msb_add_number(mshard_builder_t *b, int i_priority, double d, int is_indexed);
/* index_string_facet / index_number_facet (sindex.c:596-649) */
void msb_add_facet_str(mshard_builder_t *b, int f_priority, const char *str);
void msb_add_facet_num(mshard_builder_t *b, int f_priority, double d);
/* index_boolean (sindex.c:581-584) */
void msb_add_bool(mshard_builder_t *b, int i_priority, int val);
/* sindex_store_docdata (sindex.c:504-522) */
/* format tests: an occurrence by word id (no dictionary entry), a facet id as it is */
void msb_add_raw_wordpos(mshard_builder_t *b, uint32_t wid, int priority, uint32_t position);
void msb_add_raw_facet(mshard_builder_t *b, int f_priority, uint32_t facet_id);
void msb_end_doc(mshard_builder_t *b);
/* builds every posting list; frees the builder */
mshard_t *msb_finalize(mshard_builder_t *b);
void mshard_free(mshard_t *s);

const char *mshard_lookup_facet(const mshard_t *s, uint32_t facet_id);   /* sindex_lookup_facet */
/* posting accessors; return length, set *docs (NULL when absent) */
uint64_t mshard_wid_posting(const mshard_t *s, uint32_t wid, const uint32_t **docs);
uint64_t mshard_twid_posting(const mshard_t *s, uint32_t twid, const uint32_t **docs);
uint64_t mshard_twid_wids(const mshard_t *s, uint32_t twid, const uint32_t **wids);
uint64_t mshard_facet_posting(const mshard_t *s, int f_priority, uint32_t facet_id, const uint32_t **docs);
uint64_t mshard_bool_posting(const mshard_t *s, int i_priority, int val, const uint32_t **docs);
const uint8_t *mshard_docdata(const mshard_t *s, uint32_t docid, uint32_t *size);

/* read_vint (sindex.c:340-351) */
const uint8_t *m_read_vint(const uint8_t *buf, int *value);

/* ---- synthetic corpora ---- */
/* data_gen.py schema (test/data_gen.py:38-81) for documents first..last of a corpus of `rend`
 * docs; `facets` = the lines of test/facet.txt.  seed drives the facet picks. */
mshard_t *msynth_datagen(uint32_t rend, uint64_t seed, const char **facets, int nfacets,
                         uint32_t *facet_counts /* [nfacets] out, may be NULL */);
/* Zipf corpus: vocabulary of `vocab` pseudo-words (4-10 letters), 8-24 tokens per doc in one
 * string field, facet fields "cat" (64 buckets, zipf) and "hc" (hc_buckets, uniform), numeric
 * fields num (unique, = global doc number) and price (integer-valued in [0,1e6)).
 * Documents global_first .. global_first+ndocs-1 that route to `shard` of `num_shards`. */
mshard_t *msynth_zipf(uint64_t total_docs, uint32_t vocab, uint32_t hc_buckets, uint64_t seed,
                      int shard, int num_shards);
mshard_t *msynth_zipf_first(uint64_t total_docs, uint32_t vocab, uint32_t hc_buckets, uint64_t seed,
                            int shard, int num_shards, uint64_t first_docs);
/* the vocabulary word with 0-based frequency rank r, written to out (returns length) */
int msynth_zipf_word(uint32_t vocab, uint64_t seed, uint32_t r, char *out);
/* shard routing used by the synthetic corpora (stand-in for get_shard_routing_id, utils.c:59-62) */
int msynth_route(uint64_t global_doc, int num_shards);

#ifdef __cplusplus
}
#endif
#endif
