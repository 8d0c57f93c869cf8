/* Host code of the library under AddressSanitizer + UBSan: the sources are compiled INTO this driver (no GPU call is
 * made; the mgpu_* symbols resolve against libmarlin_gpu.so but are never reached).  Exercises what this round added on
 * the host: m_highlight (tokenizer, matcher, mark assembly) on generated text and queries, mdict_export_words on
 * generated dictionaries, the mbridge posting path.  Built and run by tests/test_host_sanitizers.py. */
#include "mshard.h"
#include "mquery.h"
#include "mdict.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned rs = 12345;
static unsigned rnd(void) { rs = rs * 1103515245u + 12345u; return (rs >> 8) & 0xFFFFFF; }

static void gen_text(char *buf, int maxlen, const char *alpha) {
    const int n = (int)(rnd() % (unsigned)maxlen);
    const int na = (int)strlen(alpha);
    for (int i = 0; i < n; i++) buf[i] = alpha[rnd() % (unsigned)na];
    buf[n] = 0;
}

/* no GPU library is linked into this driver: the one device entry point mhighlight.c names (m_highlight_batch's matcher)
 * is never reached here */
int mgpu_highlight_match(void *c, const uint32_t *a, const uint32_t *b, const uint32_t *d, uint32_t n, const void *t, uint32_t nt,
                         const uint32_t *tc, uint32_t ntc, const uint32_t *jo, uint32_t nj, int32_t *out) {
    (void)c; (void)a; (void)b; (void)d; (void)n; (void)t; (void)nt; (void)tc; (void)ntc; (void)jo; (void)nj; (void)out;
    return -1;
}

int main(void) {
    long marks = 0, strings = 0;
    const char *alpha = "aaabbcdeAB  .-'_9,!";
    for (int round = 0; round < 3000; round++) {
        char qtext[64], text[400];
        gen_text(qtext, 40, "aabbcde   ");
        m_query_t *q = mq_new(0);
        if (mq_set_text(q, qtext) != 0) { mq_free(q); continue; }
        for (int i = 0; i < 8; i++) {
            /* every fourth text is letters and hyphens only: tokens that reach the tokenizer's 127-character limit with
             * a hyphen open (the case that asked for a token of negative length) */
            gen_text(text, 399, (i & 3) == 3 ? ((round & 1) ? "ab" : "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa-") : alpha);
            char *h = m_highlight(text, q, (i & 1) ? 0 : (int)(rnd() % 5));
            strings++;
            if (h) {
                for (const char *p = h; (p = strstr(p, "<mark>")) != NULL; p += 6) marks++;
                if (strlen(h) > strlen(text) + 13 * 400) { printf("highlight grew beyond its bound\n"); return 2; }
                m_highlight_free(h);
            }
        }
        mq_free(q);
    }
    {   /* the constructed case: "ab-" + 126 letters + "-" + more letters */
        char text[400]; int n = 0;
        memcpy(text, "ab-", 3); n = 3;
        for (int i = 0; i < 126; i++) text[n++] = 'c';
        text[n++] = '-';
        for (int i = 0; i < 130; i++) text[n++] = 'd';
        memcpy(text + n, " ab", 4);
        m_query_t *q = mq_new(0);
        if (mq_set_text(q, "ab") == 0) {
            char *h = m_highlight(text, q, 0);
            if (!h || !strstr(h, "<mark>ab</mark>")) { printf("long hyphenated token: %s\n", h ? h : "(null)"); return 3; }
            m_highlight_free(h);
        }
        mq_free(q);
    }
    /* dictionaries: insert, export, check the export against lookups */
    long words = 0;
    for (int round = 0; round < 50; round++) {
        mdict_t *d = mdict_new();
        const int n = 1 + (int)(rnd() % 2000);
        for (int i = 0; i < n; i++) {
            m_chr_t w[20]; const int len = 1 + (int)(rnd() % 15);
            for (int k = 0; k < len; k++) w[k] = 'a' + rnd() % (round % 3 == 0 ? 2 : 26) + (rnd() % 50 == 0 ? 0x4e00 : 0);
            uint32_t tw[M_LEVLIMIT];
            mdict_insert(d, w, len, tw);
        }
        uint32_t *chars, *off, *wid, nw;
        if (mdict_export_words(d, &chars, &off, &wid, &nw)) return 3;
        if (nw != mdict_word_count(d)) { printf("export %u of %u\n", nw, mdict_word_count(d)); return 4; }
        for (uint32_t i = 0; i < nw; i++) {
            m_chr_t w[128];
            const uint32_t len = off[i + 1] - off[i];
            for (uint32_t k = 0; k < len; k++) w[k] = (m_chr_t)chars[off[i] + k];
            if (mdict_exists(d, w, (int)len, NULL) != wid[i]) { printf("exported word %u does not resolve to its id\n", i); return 5; }
        }
        words += nw;
        free(chars); free(off); free(wid);
        mdict_free(d);
    }
    printf("ok: %ld strings highlighted, %ld marks, %ld dictionary words exported\n", strings, marks, words);
    return 0;
}
