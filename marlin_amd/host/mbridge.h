/* mbridge.h -- fills the host-side source of a shard mirror (mshard_t) from a FOREIGN store, record by record.
 *
 * This is what a Marlin build that links libmarlin_gpu.so does once per shard (shim/squery_gpu.c walks the shard's
 * LMDB with cursors and hands every value over unchanged): postings arrive as mbmap stores them (header value +
 * container values, mbmap.c:183-283), docid2data blobs verbatim (sindex.c:273-326), num_dbi pairs in cursor order
 * (sindex.c:61-64, 1116-1118).  No dictionary is built: term lookup stays with the caller's trie (dtrie), whose
 * wid / twid numbering the postings keep.  Plain C, no reference types.
 */
#ifndef MARLIN_AMD_MBRIDGE_H
#define MARLIN_AMD_MBRIDGE_H

#include "mshard.h"
#include "marlin_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mbridge mbridge_t;
typedef enum { MBR_WID, MBR_TWID, MBR_TWID_WIDS, MBR_FACET, MBR_BOOL } mbr_kind;

/* word_count / tword_count: dt_meta.word_count / tword_count of the shard's trie (dtrie.h:55-64) */
mbridge_t *mbr_new(int num_numbers, int num_bools, int num_facets, uint32_t word_count, uint32_t tword_count);
void mbr_free(mbridge_t *b);
/* one stored posting.  id = the key's id half (wid, twid, facet id, 0|1 for bools), priority = its priority half
 * (IDPRIORITY, common.h:23): 0 for words, f_priority for facets, i_priority for bools.  MBR_TWID_WIDS holds wids,
 * not docids (twid2widbmap).  Returns 0 or an MGPU_E_* code (malformed value, id out of range). */
int mbr_add_posting(mbridge_t *b, mbr_kind kind, uint32_t id, uint32_t priority, const mgpu_mbmap_t *stored);
/* docid2data value of one document, copied verbatim */
int mbr_add_docdata(mbridge_t *b, uint32_t docid, const void *blob, size_t len);
/* one (key, value) of num_dbi[i_priority], in cursor order (value descending, docid ascending) */
int mbr_add_num(mbridge_t *b, int i_priority, double value, uint32_t docid);
/* sdata->used_bmap (shard_get_all_docids, shard.c:51-53): ascending docids in use */
int mbr_set_used(mbridge_t *b, const uint32_t *docids, uint64_t n);
/* builds the mshard_t (caller frees with mshard_free; dict is NULL) and releases the bridge */
mshard_t *mbr_finish(mbridge_t *b);

#ifdef __cplusplus
}
#endif
#endif
