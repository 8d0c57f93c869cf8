// mg_lev.h -- one dictionary word against one query term, exactly as the reference's trie walk scores it.
//
// dtrie_lookup_term's typo mode (lookup_typo / node_lev, dtrie.c:710-791; restated for the host in
// marlin_amd/host/mdict.c) walks the trie with one banded Damerau-Levenshtein row per node.  The row of a node depends
// only on the characters on the path from the root to it, so the walk's verdict on a WORD is a function of that word
// alone: follow its characters, keep the rows the walk would keep, stop where the walk would have stopped descending.
// That turns the pointer-chasing walk into an independent evaluation per dictionary word -- one lane per word on the
// GPU (k_term_scan), no trie on the device.  Everything here is the walk's arithmetic, quirks included:
//   - cells outside the band [depth - maxdist, depth + maxdist] hold 0x0F0F0F0F (the reference memsets ints with 0x0F);
//   - a node's distance is the last cell truncated to 8 bits (dtrie.c:732);
//   - the walk descends below a node only if some cell of its row is <= maxdist;
//   - prefix terms: a node within maxdist gives its distance to every word below it (least distance wins) and, from
//     depth >= term length on, ends the descent (dtrie.c:736-741).
// Compiled for the device by hipcc and for the host by the CPU test that checks it against mdict_lookup_term.
#ifndef MARLIN_AMD_MG_LEV_H
#define MARLIN_AMD_MG_LEV_H

#include <stdint.h>

#ifndef MG_HD
#ifdef __HIPCC__
#define MG_HD __host__ __device__ __forceinline__
#else
#define MG_HD static inline
#endif
#endif

#ifdef __HIPCC__
#define MG_UNROLL _Pragma("unroll")
#else
#define MG_UNROLL
#endif

#define MG_LEV_INF 0x0F0F0F0F
#define MG_LEV_MAXDIST 2                      // lookup_typo: word length > 7 -> 2, > 3 -> 1, else 0 (dtrie.c:766)
#define MG_LEV_BAND (2 * MG_LEV_MAXDIST + 1)
#define MG_LEV_NOMATCH 0xFFu

// maxdist of a term of `len` characters (dtrie.c:766)
MG_HD int mg_lev_maxdist(int len) { return len > 3 ? (len > 7 ? 2 : 1) : 0; }

// Cell i of the row at depth d; row.c[j] holds cell d - k + j.  Depth 0 is the start row (cell i = i); cell 0 of any
// row is its depth (cur[0] = prev_row[0] + 1); cells outside the band were never written: 0x0F0F0F0F.
struct MgLevRow { int c[MG_LEV_BAND]; };

MG_HD int mg_lev_cell(const MgLevRow &r, int d, int k, int size, int i) {
    if (d == 0) return i;
    if (i == 0) return d;
    int lb = d - k; if (lb < 1) lb = 1;
    int rb = d + k + 1; if (rb > size) rb = size;
    if (i < lb || i >= rb) return MG_LEV_INF;
    const int j = i - (d - k);
    int v = MG_LEV_INF;
MG_UNROLL
    for (int x = 0; x < MG_LEV_BAND; x++) if (x == j) v = r.c[x];           // static register indexing
    return v;
}

// The least distance the walk records for `word` (wlen characters) under the term `str` (tlen characters, maxdist k,
// prefix flag), or MG_LEV_NOMATCH.  `str` may live in LDS or host memory; `word` is read once per character.
template <class TermPtr, class WordPtr>
MG_HD uint32_t mg_lev_word(TermPtr str, int tlen, int k, int prefix, WordPtr word, int wlen) {
    const int size = tlen + 1;
    MgLevRow prev, pprev, cur;
MG_UNROLL
    for (int j = 0; j < MG_LEV_BAND; j++) { prev.c[j] = MG_LEV_INF; pprev.c[j] = MG_LEV_INF; cur.c[j] = MG_LEV_INF; }
    uint32_t best = MG_LEV_NOMATCH;
    uint32_t pc = 0;
    for (int d = 1; d <= wlen; d++) {
        const uint32_t c = (uint32_t)word[d - 1];
        int lb = d - k; if (lb < 1) lb = 1;
        int rb = d + k + 1; if (rb > size) rb = size;
        bool alive = d <= k;                                   // cell 0 = depth
MG_UNROLL
        for (int j = 0; j < MG_LEV_BAND; j++) {
            const int i = d - k + j;
            int v = MG_LEV_INF;
            if (j <= 2 * k && i >= lb && i < rb) {
                const int left = (j > 0 && i - 1 >= lb) ? cur.c[j - 1] : (i - 1 == 0 ? d : MG_LEV_INF);    // cur[i - 1]
                const int a = left + 1;
                const int b = mg_lev_cell(prev, d - 1, k, size, i) + 1;
                const int ins_del = a < b ? a : b;
                const uint32_t si = (uint32_t)str[i - 1];
                const int diag = mg_lev_cell(prev, d - 1, k, size, i - 1);
                const int repl = (si == c) ? diag : diag + 1;
                v = ins_del < repl ? ins_del : repl;
                if (i > 1 && d > 1) {
                    if (si == pc && (uint32_t)str[i - 2] == c) {
                        const int pp = mg_lev_cell(pprev, d - 2, k, size, i - 2);
                        const int t = (si == c) ? pp : pp + 1;
                        if (t < v) v = t;
                    }
                }
                if (v <= k) alive = true;
            }
            cur.c[j] = v;
        }
        const uint32_t dist = (uint32_t)(uint8_t)mg_lev_cell(cur, d, k, size, size - 1);
        if (dist <= (uint32_t)k) {
            if (prefix) {
                if (dist < best) best = dist;
                if (d >= size - 1) return best;                // the walk does not descend further from here
            } else if (d == wlen) {
                return dist;                                   // the word's own node
            }
        }
        if (!alive) return prefix ? best : MG_LEV_NOMATCH;     // nothing below this node is visited
        pprev = prev; prev = cur; pc = c;
    }
    return prefix ? best : MG_LEV_NOMATCH;
}

// ---- signatures: a necessary condition for a match, from 4 bytes per word.
// An edit script of cost <= k between the term and the word (or, for a prefix term, a prefix of the word) changes the
// SET of characters by at most k removals and k additions: every insertion, deletion or substitution removes at most
// one character class from the set and adds at most one, a transposition none.  The walk's distance, where it is <= k,
// is the cost of such a script (band and early stops only take matches away), so a word whose class set lacks more than
// k of the term's classes cannot match -- and, without the prefix flag, neither can one that has more than k classes the
// term lacks or whose length differs by more than k.  With the prefix flag the word must reach depth tlen - k.
// Bits 0..23: classes (code point % 24) present; bits 24..31: min(length, 255).
#define MG_LEV_SIG_CLASSES 24u
template <class Ptr>
MG_HD uint32_t mg_lev_sig(Ptr s, int len) {
    uint32_t m = 0;
    for (int i = 0; i < len; i++) m |= 1u << ((uint32_t)s[i] % MG_LEV_SIG_CLASSES);
    return m | ((uint32_t)(len > 255 ? 255 : len) << 24);
}
MG_HD int mg_lev_popc(uint32_t x) {
#ifdef __HIP_DEVICE_COMPILE__
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}
// false only if mg_lev_word(term, tlen, k, prefix, word, wlen) is MG_LEV_NOMATCH for every word with signature wsig
MG_HD bool mg_lev_sig_pass(uint32_t tsig, int k, int prefix, uint32_t wsig) {
    const int tlen = (int)(tsig >> 24), wlen = (int)(wsig >> 24);       // tlen <= 127 (MGPU_MAX_TERM_LEN); wlen 255 = "255 or more"
    const uint32_t tm = tsig & 0xFFFFFFu, wm = wsig & 0xFFFFFFu;
    // branch-free: the prefix flag only lifts the two bounds that hold for whole words
    const int revk = prefix ? 32 : k, maxlen = prefix ? 255 : tlen + k;
    return (mg_lev_popc(tm & ~wm) <= k) & (mg_lev_popc(wm & ~tm) <= revk) & (wlen >= tlen - k) & (wlen <= maxlen);
}

#endif
