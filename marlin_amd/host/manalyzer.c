/* manalyzer.c -- ASCII restatement of the reference's default analyzer.
 *
 * Host-only support code (the analyzer stays on the host per north_star).  The reference
 * runs utf8proc_map (NFKC + casefold + stripmark) first (default_analyzer.c:9-13,107-111);
 * that dependency is absent here, so this restatement is exact for ASCII input only:
 * letters are lower-cased, [a-z0-9_] are token characters (utf8proc categories Ll, Nd, Pc),
 * '.' inside abbreviations and '\'' are dropped, '-' splits hyphenated words, anything else
 * ends a token.  Positions are 1-based (default_analyzer.c:61,89).
 */
#include "mshard.h"
#include <string.h>

static inline int is_token_char(unsigned char c) {
    return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_' || c >= 0x80;
}

static void analyze(const char *str, m_word_cb cb, void *data, int for_indexing) {
    size_t n = strlen(str);
    m_chr_t token[128];
    int len = 0, is_abbrev = 0, is_hyphen = 0, position = 0, hs = 0;
    for (size_t p = 0; p < n; p++) {
        unsigned char c = (unsigned char)str[p];
        if (c >= 'A' && c <= 'Z') c = (unsigned char)(c - 'A' + 'a');
        int add_word = (p == n - 1) ? 2 : 0;
        if (is_token_char(c)) {
            token[len++] = c;
        } else {
            int handled = 0;
            if (c == '.') {
                if (len == 1 || is_abbrev) { is_abbrev = 1; handled = 1; }
            }
            if (!handled && c == '\'') { is_abbrev = 1; handled = 1; }
            if (!handled && c == '-' && len > 0) {
                is_hyphen = 1;
                cb(&token[hs], len - hs, ++position, data);
                hs = len;
                handled = 1;
            }
            if (!handled) add_word = 1;
        }
        if (add_word && len > 0) {
            int skip_whole = 0;
            if (is_hyphen) {
                if ((len - hs) > 0) {
                    cb(&token[hs], len - hs, position, data);
                    position--;
                    /* the search analyzer emits only the parts (default_analyzer.c:173-189) */
                    if (!for_indexing) skip_whole = 1;
                } else {
                    skip_whole = 1;
                }
            }
            if (!skip_whole) cb(token, len, ++position, data);
            is_hyphen = 0; is_abbrev = 0; len = 0; hs = 0;
        }
        if (len >= 127) len = 0;
    }
}

void m_analyze_for_indexing(const char *str, m_word_cb cb, void *data) { analyze(str, cb, data, 1); }
void m_analyze_for_search(const char *str, m_word_cb cb, void *data) { analyze(str, cb, data, 0); }
