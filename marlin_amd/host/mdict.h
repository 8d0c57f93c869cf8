/* mdict.h -- host-side term dictionary (stays on the CPU, as north_star asks).
 *
 * In-memory trie with the reference dtrie's ids and lookup semantics:
 *   wid  = ++word_count on first insertion of a word          (dtrie.c:452-484)
 *   twid = ++tword_count for every new node at depth < 3      (dtrie.c:351-377)
 *   dtrie_lookup_term / dtrie_lookup_exact                    (dtrie.c:793-823)
 * The on-disk page format of dtrie.c is NOT reproduced: only the ids and the
 * {twid, wid -> distance} results, which is all the shard query consumes (dtrie.h:94-97).
 */
#ifndef MARLIN_AMD_MDICT_H
#define MARLIN_AMD_MDICT_H

#include "mcommon.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tiny u32 -> int open-addressing map (stand-in for khash WID2TYPOS, dtrie.h:18) */
typedef struct m_u32map {
    uint32_t *keys;     /* 0 = empty (wids start at 1) */
    int32_t  *vals;
    uint32_t  cap;      /* power of two */
    uint32_t  n;
} m_u32map_t;

void     m_u32map_init(m_u32map_t *m);
void     m_u32map_free(m_u32map_t *m);
void     m_u32map_clear(m_u32map_t *m);
int32_t  m_u32map_get(const m_u32map_t *m, uint32_t key, int32_t defval);
void     m_u32map_set(m_u32map_t *m, uint32_t key, int32_t val);

/* dtrie.h:94-97 */
typedef struct m_termresult {
    uint32_t   twid;
    m_u32map_t wordids;     /* wid -> least distance */
} m_termresult_t;

typedef struct mdict mdict_t;

mdict_t *mdict_new(void);
void     mdict_free(mdict_t *d);
uint32_t mdict_word_count(const mdict_t *d);
uint32_t mdict_tword_count(const mdict_t *d);

/* dtrie_insert (dtrie.c:452): returns the wid, fills twids[0..min(len,3)) */
uint32_t mdict_insert(mdict_t *d, const m_chr_t *str, int slen, uint32_t *twids);
/* dtrie_exists (dtrie.c:432) */
uint32_t mdict_exists(const mdict_t *d, const m_chr_t *str, int slen, uint32_t *twids);
/* dtrie_lookup_exact (dtrie.c:793) */
uint32_t mdict_lookup_exact(const mdict_t *d, const m_word_t *w);
/* dtrie_lookup_term (dtrie.c:805); caller owns *tr and must m_termresult_free it */
void     mdict_lookup_term(const mdict_t *d, const m_term_t *t, m_termresult_t *tr);
void     m_termresult_free(m_termresult_t *tr);
/* all words, flat, in trie order: chars[off[i] .. off[i+1]) are the characters of the word with id wid[i]; the caller frees */
int      mdict_export_words(const mdict_t *d, uint32_t **chars_out, uint32_t **off_out, uint32_t **wid_out, uint32_t *n_out);
/* which lookup mode dtrie_lookup_term takes for a term (dtrie.c:805-823): 0 = twid, 1 = no typos, 2 = typo walk */
int      mdict_lookup_mode(const m_term_t *t);

#ifdef __cplusplus
}
#endif
#endif
