// CPU check of marlin_amd/csrc/mg_lev.h (the per-word restatement of the trie walk that the GPU term scan runs):
// for every term, score EVERY exported dictionary word with mg_lev_word and compare the resulting wid -> distance set
// with mdict_lookup_term's.  Built by tests/test_term_scan.py with g++; test support, not part of the product.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
extern "C" {
#include "mdict.h"
}
#include "mg_lev.h"

extern "C" {
// words: nwords strings (UTF-32 code points) of lengths wlen[]; terms likewise + prefix flags.  Returns the number of
// terms whose sets differ (0 = all equal) and prints the first differences.
int lev_check(const uint32_t *wchars, const int *wlen, int nwords, const uint32_t *tchars, const int *tlen,
              const int *tprefix, int nterms, long *matches_out) {
    mdict_t *d = mdict_new();
    size_t at = 0;
    for (int i = 0; i < nwords; i++) { mdict_insert(d, wchars + at, wlen[i], NULL); at += (size_t)wlen[i]; }
    uint32_t *chars, *off, *wid, n;
    if (mdict_export_words(d, &chars, &off, &wid, &n)) return -1;
    if (n != mdict_word_count(d)) { printf("export holds %u of %u words\n", n, mdict_word_count(d)); return -2; }
    int bad = 0; long matches = 0, pairs = 0, passed = 0;
    at = 0;
    for (int t = 0; t < nterms; t++) {
        m_term_t term; memset(&term, 0, sizeof(term));
        term.word.length = tlen[t];
        for (int i = 0; i < tlen[t]; i++) term.word.chars[i] = tchars[at + i];
        at += (size_t)tlen[t];
        term.prefix = (uint8_t)tprefix[t]; term.typos = 1;
        if (mdict_lookup_mode(&term) != 2) continue;
        m_termresult_t tr; mdict_lookup_term(d, &term, &tr);
        std::map<uint32_t, int> want, got;
        for (uint32_t k = 0; k < tr.wordids.cap; k++) if (tr.wordids.keys[k]) want[tr.wordids.keys[k]] = tr.wordids.vals[k];
        const int k = mg_lev_maxdist(tlen[t]);
        // as the device does it: the signature filter first (k_term_filter), the walk only for the words it lets through
        // (k_term_eval) -- a match the filter dropped would show up as a missing word below
        const uint32_t tsig = mg_lev_sig(term.word.chars, tlen[t]);
        for (uint32_t w = 0; w < n; w++) {
            const int wl = (int)(off[w + 1] - off[w]);
            pairs++;
            if (!mg_lev_sig_pass(tsig, k, tprefix[t], mg_lev_sig(chars + off[w], wl))) continue;
            passed++;
            const uint32_t dist = mg_lev_word(term.word.chars, tlen[t], k, tprefix[t], chars + off[w], wl);
            if (dist != MG_LEV_NOMATCH) got[wid[w]] = (int)dist;
        }
        matches += (long)want.size();
        if (want != got) {
            if (bad < 5) {
                printf("term %d (len %d prefix %d): walk %zu words, scan %zu words\n", t, tlen[t], tprefix[t], want.size(), got.size());
                for (auto &kv : want) if (!got.count(kv.first) || got[kv.first] != kv.second) { printf("  wid %u: walk %d scan %d\n", kv.first, kv.second, got.count(kv.first) ? got[kv.first] : -1); break; }
                for (auto &kv : got) if (!want.count(kv.first)) { printf("  wid %u: scan %d, walk none\n", kv.first, kv.second); break; }
            }
            bad++;
        }
        m_termresult_free(&tr);
    }
    free(chars); free(off); free(wid);
    mdict_free(d);
    if (matches_out) *matches_out = matches;
    if (getenv("LEV_CHECK_STATS")) printf("signature filter: %ld of %ld (term, word) pairs pass, %ld matches\n", passed, pairs, matches);
    return bad;
}
}
