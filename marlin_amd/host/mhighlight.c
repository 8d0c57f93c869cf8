/* mhighlight.c -- row f-4: the string half of hit materialisation.
 *
 * m_highlight restates highlight() of the reference (highlight.c:215-345) with its tokenizer (highlight.c:28-152),
 * matcher (is_match, highlight.c:195-213) and distance function (levenshtein, highlight.c:166-192): the text is cut
 * into tokens that remember the byte range of every character, each token is tried against every query term (prefix
 * semantics: the distance function returns how many leading characters of the token match the term best), and the
 * matched leading part of a token is wrapped in <mark> ... </mark>.  Like marlin_amd/host/manalyzer.c this is exact
 * for ASCII text: the reference first runs utf8proc_map (NFKC, case folding, mark stripping), which is not available
 * here; bytes >= 0x80 are treated as token characters of one byte each.
 * Quirks kept: hyphenated words yield their parts AND the joined word as tokens; the transposition case adds `cost`,
 * not 1; text after the last token is not copied to the output; nothing matched -> NULL.
 * m_highlight is host-only code; m_highlight_batch runs the matcher for a whole batch of strings on the device
 * (mgpu_highlight_match).  The JSON side (getFields projection, which strings to highlight) is in marlin_amd/hits.py.
 */
#include "mquery.h"
#include "marlin_gpu.h"
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t start, end; } char_se_t;
typedef struct {
    m_chr_t *chars; int length;
    char_se_t *se;                 /* one per character */
    int is_match, match_len;
} hl_token_t;
typedef struct { hl_token_t *v; int n, cap; } hl_tokens_t;

static void new_token(hl_tokens_t *ts, const m_chr_t *chars, int length, const char_se_t *se_all, int se_n) {
    if (length <= 0) return;
    if (ts->n == ts->cap) { ts->cap = ts->cap ? ts->cap * 2 : 32; ts->v = realloc(ts->v, sizeof(hl_token_t) * (size_t)ts->cap); }
    hl_token_t *t = &ts->v[ts->n++];
    t->chars = malloc(sizeof(m_chr_t) * (size_t)length);
    memcpy(t->chars, chars, sizeof(m_chr_t) * (size_t)length);
    t->length = length;
    t->se = malloc(sizeof(char_se_t) * (size_t)length);
    /* the ranges of the last `length` characters seen (highlight.c:24-27) */
    memcpy(t->se, se_all + (se_n - length), sizeof(char_se_t) * (size_t)length);
    t->is_match = 0; t->match_len = 0;
}

static int is_token_char(unsigned char c) {
    return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_' || c >= 0x80;
}

/* highlight.c:28-152 */
static void tokenize(const char *str, hl_tokens_t *ts) {
    const size_t n = strlen(str);
    char_se_t *se = malloc(sizeof(char_se_t) * (n ? n : 1));
    int se_n = 0;
    m_chr_t token[128];
    int len = 0, is_abbrev = 0, is_hyphen = 0, hs = 0;
    for (size_t p = 0; p < n; p++) {
        unsigned char c = (unsigned char)str[p];
        if (c >= 'A' && c <= 'Z') c = (unsigned char)(c - 'A' + 'a');
        int add_word = (p == n - 1) ? 2 : 0;
        if (is_token_char(c)) {
            token[len++] = c;
            se[se_n].start = (uint32_t)p; se[se_n].end = (uint32_t)p + 1; se_n++;
        } else {
            int handled = 0;
            if (c == '.') { if (len == 1 || is_abbrev) { is_abbrev = 1; handled = 1; } }
            if (!handled && c == '\'') { is_abbrev = 1; handled = 1; }
            if (!handled && c == '-' && len > 0) {
                is_hyphen = 1;
                new_token(ts, &token[hs], len - hs, se, se_n);
                hs = len;
                handled = 1;
            }
            if (!handled) add_word = 1;
        }
        if (add_word && len > 0) {
            int whole = 1;
            if (is_hyphen) {
                if (len - hs > 0) new_token(ts, &token[hs], len - hs, se, se_n);
                else whole = 0;                               /* "word already added" */
            }
            if (whole) new_token(ts, token, len, se, se_n);
            is_hyphen = 0; is_abbrev = 0; len = 0; hs = 0;
        }
        /* a token of 127 characters is dropped (highlight.c:141); unlike there, the hyphen state goes with it: a stale
         * hyphen start past the new length made the next '-' ask for a token of negative length */
        if (len >= 127) { len = 0; hs = 0; is_hyphen = 0; }
    }
    free(se);
}

#define MIN3(a, b, c) ((a) < (b) ? ((a) < (c) ? (a) : (c)) : ((b) < (c) ? (b) : (c)))

/* highlight.c:166-192: s1 = token, s2 = term; returns the length of the token prefix closest to the term, or -1 */
static int hl_levenshtein(const m_chr_t *s1, int s1len, const m_chr_t *s2, int s2len, int max_dist) {
    unsigned int *m = malloc(sizeof(unsigned int) * (size_t)(s2len + 1) * (size_t)(s1len + 1));
#define M(x, y) m[(size_t)(x) * (size_t)(s1len + 1) + (size_t)(y)]
    M(0, 0) = 0;
    for (int x = 1; x <= s2len; x++) M(x, 0) = M(x - 1, 0) + 1;
    for (int y = 1; y <= s1len; y++) M(0, y) = M(0, y - 1) + 1;
    int best_pos = 0xFF;
    for (int x = 1; x <= s2len; x++) {
        int best = 0xFF;
        for (int y = 1; y <= s1len; y++) {
            const unsigned int cost = (s1[y - 1] == s2[x - 1]) ? 0 : 1;
            M(x, y) = MIN3(M(x - 1, y) + 1, M(x, y - 1) + 1, M(x - 1, y - 1) + cost);
            if (x > 1 && y > 1) {
                if (s1[y - 1] == s2[x - 2] && s1[y - 2] == s2[x - 1]) {
                    const unsigned int t = M(x - 2, y - 2) + cost;
                    if (t < M(x, y)) M(x, y) = t;
                }
            }
            if ((int)M(x, y) < best) { best = (int)M(x, y); best_pos = y; }
        }
        if (best > max_dist) { free(m); return -1; }
    }
#undef M
    free(m);
    return best_pos;
}

/* highlight.c:195-213 */
static int is_match(hl_token_t *token, const m_term_t *t) {
    int max_dist = 0;
    if (t->typos) max_dist = t->word.length > 7 ? 2 : (t->word.length > 3 ? 1 : 0);
    if (!t->prefix) {
        if (token->length < t->word.length - max_dist || token->length > t->word.length + max_dist) return 0;
    }
    const int len = hl_levenshtein(token->chars, token->length, t->word.chars, t->word.length, max_dist);
    if (len >= 0 && len <= token->length) {          /* (a term of length 0 would return 0xFF: no such term exists) */
        token->is_match = 1;
        token->match_len = len;
    }
    return token->is_match;
}

/* highlight.c:215-345 after the tokenizer.  pre == NULL: every token that the overlap rule lets through is tried against
 * the terms here (is_match); otherwise pre[i] is token i's verdict, computed ahead for all tokens (the device matcher):
 * the match length, or -1.  A token's verdict depends on the token and the terms alone, so both give the same marks. */
static char *highlight_tokens(const char *str, const m_query_t *q, int snip_num_words, hl_tokens_t *tsp, const int32_t *pre) {
    hl_tokens_t ts = *tsp;
    const int num_terms = q->num_terms, num_tokens = ts.n;
    int num_matches = 0;
    uint32_t last_match_end = 0;
    int snip_end = 0, best_start = 0, best_end = 0, best_matches = 0;
    for (int i = 0; i < num_tokens; i++) {
        hl_token_t *t = &ts.v[i];
        if (t->se[0].start < last_match_end) continue;                     /* no overlapping marks */
        if (pre) {
            if (pre[i] >= 1 && pre[i] <= t->length) {
                t->is_match = 1; t->match_len = pre[i];
                num_matches++;
                last_match_end = t->se[t->match_len - 1].end;
            }
        } else for (int j = 0; j < num_terms; j++) {
            if (is_match(t, &q->terms[j])) {
                num_matches++;
                last_match_end = t->se[t->match_len - 1].end;
                break;
            }
        }
        if (snip_num_words) {
            if (snip_end < snip_num_words) { best_matches = num_matches; best_end++; snip_end++; }
            else {
                num_matches -= ts.v[snip_end - snip_num_words].is_match;
                if (num_matches > best_matches) { best_start = snip_end - snip_num_words; best_end = snip_end; best_matches = num_matches; }
                snip_end++;
            }
        }
    }
    char *resp = NULL;
    if (last_match_end != 0) {
        /* every marked token adds 13 bytes; count the marked tokens, not num_matches (snipping subtracts from it) */
        size_t marked = 0;
        for (int i = 0; i < num_tokens; i++) marked += ts.v[i].is_match ? 1 : 0;
        const size_t slen = strlen(str);
        resp = malloc(slen + marked * strlen("<mark></mark>") + 1);
        size_t pos = 0;
        uint32_t start = 0;
        int end = num_tokens, i = 0;
        if (snip_num_words) { start = ts.v[best_start].se[0].start; i = best_start; end = best_end; }
        for (; i < end; i++) {
            hl_token_t *t = &ts.v[i];
            if (t->is_match) {
                const char_se_t *c = &t->se[0];
                if (c->start < start) start = c->start;                    /* split words */
                memcpy(&resp[pos], &str[start], c->start - start); pos += c->start - start;
                memcpy(&resp[pos], "<mark>", 6); pos += 6;
                const char_se_t *c2 = &t->se[t->match_len - 1];
                memcpy(&resp[pos], &str[c->start], c2->end - c->start); pos += c2->end - c->start;
                start = c2->end;
                memcpy(&resp[pos], "</mark>", 7); pos += 7;
            }
            if (i == num_tokens - 1) {
                const char_se_t *c = &t->se[t->length - 1];
                if (c->end > start) { memcpy(&resp[pos], &str[start], c->end - start); pos += c->end - start; }
            }
        }
        resp[pos] = '\0';
    }
    return resp;
}
static void tokens_free(hl_tokens_t *ts) {
    for (int i = 0; i < ts->n; i++) { free(ts->v[i].chars); free(ts->v[i].se); }
    free(ts->v);
    ts->v = NULL; ts->n = ts->cap = 0;
}

/* the result is malloc'd (m_highlight_free), NULL when nothing matched */
char *m_highlight(const char *str, const m_query_t *q, int snip_num_words) {
    hl_tokens_t ts = { NULL, 0, 0 };
    tokenize(str, &ts);
    char *resp = highlight_tokens(str, q, snip_num_words, &ts, NULL);
    tokens_free(&ts);
    return resp;
}

/* highlight() for n strings at once -- every string of every hit on the pages of a batch of queries (form_result's loop,
 * query.c:385-401 -> hit_query_processing): strs[i] is highlighted for qs[i] (the same query object may repeat) with
 * snip[i] (or 0 when snip is NULL).  The strings are cut into tokens here, ALL tokens are matched against their queries'
 * terms in one device launch (mgpu_highlight_match on ctx, which must be idle), and the marks are assembled here.
 * out[i] = malloc'd result (m_highlight_free) or NULL when nothing matched.  Returns 0 or an MGPU_E_* code. */
int m_highlight_batch(mgpu_ctx_t *ctx, const char *const *strs, const m_query_t *const *qs, const int *snip, int n, char **out) {
    if (n < 0 || (n && (!strs || !qs || !out))) return MGPU_E_ARG;
    for (int i = 0; i < n; i++) out[i] = NULL;
    if (!n) return 0;
    hl_tokens_t *ts = calloc((size_t)n, sizeof(*ts));
    uint32_t *job_of = malloc(sizeof(uint32_t) * (size_t)n);             /* string -> job (distinct query) */
    const m_query_t **jobs = malloc(sizeof(*jobs) * (size_t)n);
    int rc = (ts && job_of && jobs) ? 0 : MGPU_E_LIMIT;
    uint32_t njobs = 0, ntok = 0, nterms = 0;
    size_t nchars = 0, ntch = 0;
    if (!rc) {
        for (int i = 0; i < n; i++) {
            /* the pages of one query come together: looking at the last job is enough to fold them */
            if (njobs && jobs[njobs - 1] == qs[i]) job_of[i] = njobs - 1;
            else { jobs[njobs] = qs[i]; job_of[i] = njobs++; }
            tokenize(strs[i], &ts[i]);
            ntok += (uint32_t)ts[i].n;
            for (int k = 0; k < ts[i].n; k++) nchars += (size_t)ts[i].v[k].length;
        }
        for (uint32_t j = 0; j < njobs; j++) { nterms += (uint32_t)jobs[j]->num_terms; for (int t = 0; t < jobs[j]->num_terms; t++) ntch += (size_t)jobs[j]->terms[t].word.length; }
    }
    int32_t *pre = NULL;
    if (!rc && ntok) {
        uint32_t *tok_chars = malloc(sizeof(uint32_t) * (nchars ? nchars : 1)), *tok_off = malloc(sizeof(uint32_t) * ((size_t)ntok + 1)),
                 *tok_job = malloc(sizeof(uint32_t) * (size_t)ntok), *term_chars = malloc(sizeof(uint32_t) * (ntch ? ntch : 1)),
                 *jto = malloc(sizeof(uint32_t) * ((size_t)njobs + 1));
        mgpu_hl_term_t *terms = malloc(sizeof(*terms) * (nterms ? nterms : 1));
        pre = malloc(sizeof(int32_t) * (size_t)ntok);
        if (!tok_chars || !tok_off || !tok_job || !term_chars || !jto || !terms || !pre) rc = MGPU_E_LIMIT;
        else {
            uint32_t at = 0, k = 0;
            for (int i = 0; i < n; i++)
                for (int x = 0; x < ts[i].n; x++) {
                    tok_off[k] = at; tok_job[k] = job_of[i]; k++;
                    memcpy(tok_chars + at, ts[i].v[x].chars, sizeof(uint32_t) * (size_t)ts[i].v[x].length); at += (uint32_t)ts[i].v[x].length;
                }
            tok_off[k] = at;
            uint32_t ta = 0, tn = 0;
            for (uint32_t j = 0; j < njobs; j++) {
                jto[j] = tn;
                for (int t = 0; t < jobs[j]->num_terms; t++) {
                    const m_term_t *tm = &jobs[j]->terms[t];
                    terms[tn].off = ta; terms[tn].len = (uint32_t)tm->word.length; terms[tn].prefix = tm->prefix ? 1 : 0;
                    terms[tn].max_dist = tm->typos ? (tm->word.length > 7 ? 2 : (tm->word.length > 3 ? 1 : 0)) : 0;     /* is_match, highlight.c:196-198 */
                    memcpy(term_chars + ta, tm->word.chars, sizeof(uint32_t) * (size_t)tm->word.length); ta += (uint32_t)tm->word.length;
                    tn++;
                }
            }
            jto[njobs] = tn;
            rc = mgpu_highlight_match(ctx, tok_chars, tok_off, tok_job, ntok, terms, nterms, term_chars, (uint32_t)ntch, jto, njobs, pre);
        }
        free(tok_chars); free(tok_off); free(tok_job); free(term_chars); free(jto); free(terms);
    }
    if (!rc) {
        uint32_t k = 0;
        for (int i = 0; i < n; i++) {
            if (ts[i].n) out[i] = highlight_tokens(strs[i], qs[i], snip ? snip[i] : 0, &ts[i], pre + k);
            k += (uint32_t)ts[i].n;
        }
    }
    if (ts) for (int i = 0; i < n; i++) tokens_free(&ts[i]);
    free(ts); free(job_of); free(jobs); free(pre);
    return rc;
}

void m_highlight_free(char *s) { free(s); }
