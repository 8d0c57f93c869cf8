/* mdict.c -- see mdict.h.  Host-only; never on the GPU path's critical kernels. */
#include "mdict.h"
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- u32 map */

static inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}

void m_u32map_init(m_u32map_t *m) { memset(m, 0, sizeof(*m)); }

void m_u32map_free(m_u32map_t *m) {
    free(m->keys); free(m->vals);
    memset(m, 0, sizeof(*m));
}

void m_u32map_clear(m_u32map_t *m) {
    if (m->cap) memset(m->keys, 0, sizeof(uint32_t) * m->cap);
    m->n = 0;
}

int32_t m_u32map_get(const m_u32map_t *m, uint32_t key, int32_t defval) {
    if (!m->cap) return defval;
    uint32_t mask = m->cap - 1;
    for (uint32_t i = mix32(key) & mask;; i = (i + 1) & mask) {
        if (m->keys[i] == key) return m->vals[i];
        if (m->keys[i] == 0) return defval;
    }
}

static void u32map_grow(m_u32map_t *m) {
    uint32_t ncap = m->cap ? m->cap * 2 : 64;
    uint32_t *ok = m->keys; int32_t *ov = m->vals; uint32_t ocap = m->cap;
    m->keys = calloc(ncap, sizeof(uint32_t));
    m->vals = malloc(ncap * sizeof(int32_t));
    m->cap = ncap;
    uint32_t mask = ncap - 1;
    for (uint32_t j = 0; j < ocap; j++) {
        if (!ok[j]) continue;
        uint32_t i = mix32(ok[j]) & mask;
        while (m->keys[i]) i = (i + 1) & mask;
        m->keys[i] = ok[j]; m->vals[i] = ov[j];
    }
    free(ok); free(ov);
}

void m_u32map_set(m_u32map_t *m, uint32_t key, int32_t val) {
    if ((m->n + 1) * 2 > m->cap) u32map_grow(m);
    uint32_t mask = m->cap - 1;
    for (uint32_t i = mix32(key) & mask;; i = (i + 1) & mask) {
        if (m->keys[i] == key) { m->vals[i] = val; return; }
        if (m->keys[i] == 0) { m->keys[i] = key; m->vals[i] = val; m->n++; return; }
    }
}

/* ---------------------------------------------------------------- trie */

struct mdict {
    /* node arrays; node 0 is the root */
    m_chr_t    *ch;
    uint32_t *first_child;
    uint32_t *next_sib;
    uint32_t *wid;
    uint32_t *twid;
    uint32_t  num_nodes, cap_nodes;
    uint32_t  word_count, tword_count;
};

mdict_t *mdict_new(void) {
    mdict_t *d = calloc(1, sizeof(*d));
    d->cap_nodes = 1024;
    d->ch = calloc(d->cap_nodes, sizeof(m_chr_t));
    d->first_child = calloc(d->cap_nodes, sizeof(uint32_t));
    d->next_sib = calloc(d->cap_nodes, sizeof(uint32_t));
    d->wid = calloc(d->cap_nodes, sizeof(uint32_t));
    d->twid = calloc(d->cap_nodes, sizeof(uint32_t));
    d->num_nodes = 1;
    return d;
}

void mdict_free(mdict_t *d) {
    if (!d) return;
    free(d->ch); free(d->first_child); free(d->next_sib); free(d->wid); free(d->twid);
    free(d);
}

uint32_t mdict_word_count(const mdict_t *d) { return d->word_count; }
uint32_t mdict_tword_count(const mdict_t *d) { return d->tword_count; }

static inline uint32_t get_child(const mdict_t *d, uint32_t n, m_chr_t c) {
    for (uint32_t k = d->first_child[n]; k; k = d->next_sib[k]) {
        if (d->ch[k] == c) return k;
    }
    return 0;
}

static uint32_t add_child(mdict_t *d, uint32_t parent, m_chr_t c) {
    if (d->num_nodes == d->cap_nodes) {
        uint32_t nc = d->cap_nodes * 2;
        d->ch = realloc(d->ch, nc * sizeof(m_chr_t));
        d->first_child = realloc(d->first_child, nc * sizeof(uint32_t));
        d->next_sib = realloc(d->next_sib, nc * sizeof(uint32_t));
        d->wid = realloc(d->wid, nc * sizeof(uint32_t));
        d->twid = realloc(d->twid, nc * sizeof(uint32_t));
        d->cap_nodes = nc;
    }
    uint32_t k = d->num_nodes++;
    d->ch[k] = c;
    d->first_child[k] = 0;
    d->wid[k] = 0;
    d->twid[k] = 0;
    d->next_sib[k] = d->first_child[parent];
    d->first_child[parent] = k;
    return k;
}

/* add_path_nodes + dtrie_insert (dtrie.c:351-377, 452-484) */
uint32_t mdict_insert(mdict_t *d, const m_chr_t *str, int slen, uint32_t *twids) {
    uint32_t cur = 0;
    for (int i = 0; i < slen; i++) {
        uint32_t next = get_child(d, cur, str[i]);
        if (!next) {
            next = add_child(d, cur, str[i]);
            if (i < M_LEVLIMIT) d->twid[next] = ++d->tword_count;
        }
        cur = next;
        if (i < M_LEVLIMIT && twids) twids[i] = d->twid[cur];
    }
    if (!d->wid[cur]) d->wid[cur] = ++d->word_count;
    return d->wid[cur];
}

/* word_exists_under_root (dtrie.c:406-421) */
uint32_t mdict_exists(const mdict_t *d, const m_chr_t *str, int slen, uint32_t *twids) {
    uint32_t cur = 0;
    for (int i = 0; i < slen; i++) {
        uint32_t next = get_child(d, cur, str[i]);
        if (!next) return 0;
        cur = next;
        if (i < M_LEVLIMIT && twids) twids[i] = d->twid[cur];
    }
    return d->wid[cur];
}

/* lookup_word_node (dtrie.c:392-404); returns node index or UINT32_MAX */
static uint32_t lookup_word_node(const mdict_t *d, const m_chr_t *str, int slen) {
    uint32_t cur = 0;
    for (int i = 0; i < slen; i++) {
        uint32_t next = get_child(d, cur, str[i]);
        if (!next) return UINT32_MAX;
        cur = next;
    }
    return cur;
}

uint32_t mdict_lookup_exact(const mdict_t *d, const m_word_t *w) {
    uint32_t n = lookup_word_node(d, w->chars, w->length);
    return n == UINT32_MAX ? 0 : d->wid[n];
}

/* add_wordid_to_result (dtrie.c:633-640): keep the least distance */
static inline void add_wordid(m_termresult_t *tr, uint32_t wid, int distance) {
    int cur = m_u32map_get(&tr->wordids, wid, 0xFF);
    if (distance < cur) m_u32map_set(&tr->wordids, wid, distance);
}

/* node_walk (dtrie.c:642-658): every word in the subtree gets `distance` */
static void node_walk(const mdict_t *d, uint32_t n, m_termresult_t *tr, int distance) {
    /* explicit stack: subtrees can be deep for long words */
    uint32_t stack_local[256];
    uint32_t *stack = stack_local;
    size_t cap = 256, top = 0;
    stack[top++] = n;
    while (top) {
        uint32_t x = stack[--top];
        if (d->wid[x]) add_wordid(tr, d->wid[x], distance);
        for (uint32_t k = d->first_child[x]; k; k = d->next_sib[k]) {
            if (top == cap) {
                size_t ncap = cap * 2;
                uint32_t *ns = malloc(ncap * sizeof(uint32_t));
                memcpy(ns, stack, top * sizeof(uint32_t));
                if (stack != stack_local) free(stack);
                stack = ns; cap = ncap;
            }
            stack[top++] = k;
        }
    }
    if (stack != stack_local) free(stack);
}

struct lev_data {
    const mdict_t *d;
    const m_word_t *word;
    int prefix;
    m_termresult_t *tr;
    int maxdist;
};

#define LEV_INF 0x0F0F0F0F   /* memset(..., 0x0F, ...) of ints, dtrie.c:718 */

/* node_lev (dtrie.c:710-760): banded Damerau-Levenshtein row per trie node.
 * rows live in one buffer, row r at rows + r*size. */
static void node_lev(struct lev_data *ld, uint32_t n, int *prev_row, int *pprev_row,
                     m_chr_t c, m_chr_t pc, int depth) {
    const int size = ld->word->length + 1;
    int *cur = prev_row + size;
    const m_chr_t *str = ld->word->chars;

    cur[0] = prev_row[0] + 1;
    for (int i = 1; i < size; i++) cur[i] = LEV_INF;
    int lb = depth - ld->maxdist; if (lb < 1) lb = 1;
    int rb = depth + ld->maxdist + 1; if (rb > size) rb = size;
    for (int i = lb; i < rb; i++) {
        int a = cur[i - 1] + 1, b = prev_row[i] + 1;
        int ins_del = a < b ? a : b;
        int repl = (str[i - 1] == c) ? prev_row[i - 1] : prev_row[i - 1] + 1;
        cur[i] = ins_del < repl ? ins_del : repl;
        /* transposition */
        if (i > 1 && depth > 1) {
            if (str[i - 1] == pc && str[i - 2] == c) {
                int t = (str[i - 1] == c) ? pprev_row[i - 2] : pprev_row[i - 2] + 1;
                if (t < cur[i]) cur[i] = t;
            }
        }
    }
    uint8_t dist = (uint8_t)cur[size - 1];       /* the reference truncates to u8, dtrie.c:732 */
    if (dist <= ld->maxdist) {
        if (ld->prefix) {
            node_walk(ld->d, n, ld->tr, dist);
            if (depth >= size - 1) return;
        } else if (ld->d->wid[n]) {
            add_wordid(ld->tr, ld->d->wid[n], dist);
        }
    }
    for (int i = 0; i < size; i++) {
        if (cur[i] <= ld->maxdist) {
            for (uint32_t k = ld->d->first_child[n]; k; k = ld->d->next_sib[k]) {
                node_lev(ld, k, cur, prev_row, ld->d->ch[k], c, depth + 1);
            }
            break;
        }
    }
}

/* lookup_typo (dtrie.c:763-791) */
static void lookup_typo(const mdict_t *d, const m_term_t *t, m_termresult_t *tr) {
    int wlen = t->word.length;
    int maxdist = (wlen > 3) ? ((wlen > 7) ? 2 : 1) : 0;
    struct lev_data ld = { d, &t->word, t->prefix, tr, maxdist };
    int *rows = malloc(sizeof(int) * (size_t)((wlen + 1) * (wlen + 2) * 2 + 8));
    for (int i = 0; i <= wlen; i++) rows[i] = i;
    for (uint32_t k = d->first_child[0]; k; k = d->next_sib[k]) {
        node_lev(&ld, k, rows, NULL, d->ch[k], 0, 1);
    }
    free(rows);
}

int mdict_lookup_mode(const m_term_t *t) {
    if (t->prefix && t->word.length <= M_LEVLIMIT) return 0;
    return t->typos ? 2 : 1;
}

void mdict_lookup_term(const mdict_t *d, const m_term_t *t, m_termresult_t *tr) {
    tr->twid = 0;
    m_u32map_init(&tr->wordids);
    const m_word_t *w = &t->word;
    /* lookup_twid (dtrie.c:664-682) */
    if (t->prefix && w->length <= M_LEVLIMIT) {
        uint32_t n = lookup_word_node(d, w->chars, w->length);
        if (n != UINT32_MAX) tr->twid = d->twid[n];
        return;
    }
    /* lookup_notypo (dtrie.c:684-700) */
    if (!t->typos) {
        uint32_t n = lookup_word_node(d, w->chars, w->length);
        if (n != UINT32_MAX) {
            if (t->prefix) node_walk(d, n, tr, 0);
            else if (d->wid[n]) add_wordid(tr, d->wid[n], 0);
        }
        return;
    }
    lookup_typo(d, t, tr);
}

void m_termresult_free(m_termresult_t *tr) { m_u32map_free(&tr->wordids); }

/* Every word of the dictionary as a flat character array in trie (depth-first) order, so that neighbours share
 * prefixes: the layout the device-side term scan reads (marlin_amd/csrc/mg_lev.h).  off has n + 1 entries. */
int mdict_export_words(const mdict_t *d, uint32_t **chars_out, uint32_t **off_out, uint32_t **wid_out, uint32_t *n_out) {
    uint32_t n = d->word_count, cap = 1024, nchars = 0, nw = 0;
    uint32_t *chars = malloc(sizeof(uint32_t) * cap);
    uint32_t *off = malloc(sizeof(uint32_t) * ((size_t)n + 1));
    uint32_t *wid = malloc(sizeof(uint32_t) * ((size_t)n + 1));
    /* explicit DFS: stack of (node, depth); path[] holds the characters of the current path */
    uint32_t scap = 1024, top = 0;
    uint32_t *st_node = malloc(sizeof(uint32_t) * scap), *st_depth = malloc(sizeof(uint32_t) * scap);
    uint32_t pcap = 256;
    m_chr_t *path = malloc(sizeof(m_chr_t) * pcap);
    if (!chars || !off || !wid || !st_node || !st_depth || !path) goto oom;
    for (uint32_t k = d->first_child[0]; k; k = d->next_sib[k]) {
        if (top == scap) { scap *= 2; st_node = realloc(st_node, sizeof(uint32_t) * scap); st_depth = realloc(st_depth, sizeof(uint32_t) * scap); if (!st_node || !st_depth) goto oom; }
        st_node[top] = k; st_depth[top++] = 1;
    }
    while (top) {
        uint32_t x = st_node[--top], dep = st_depth[top];
        if (dep > pcap) { pcap *= 2; path = realloc(path, sizeof(m_chr_t) * pcap); if (!path) goto oom; }
        path[dep - 1] = d->ch[x];
        if (d->wid[x] && nw < n) {
            if (nchars + dep > cap) { while (nchars + dep > cap) cap *= 2; chars = realloc(chars, sizeof(uint32_t) * cap); if (!chars) goto oom; }
            off[nw] = nchars; wid[nw] = d->wid[x];
            for (uint32_t i = 0; i < dep; i++) chars[nchars++] = (uint32_t)path[i];
            nw++;
        }
        for (uint32_t k = d->first_child[x]; k; k = d->next_sib[k]) {
            if (top == scap) { scap *= 2; st_node = realloc(st_node, sizeof(uint32_t) * scap); st_depth = realloc(st_depth, sizeof(uint32_t) * scap); if (!st_node || !st_depth) goto oom; }
            st_node[top] = k; st_depth[top++] = dep + 1;
        }
    }
    off[nw] = nchars;
    free(st_node); free(st_depth); free(path);
    *chars_out = chars; *off_out = off; *wid_out = wid; *n_out = nw;
    return 0;
oom:
    free(chars); free(off); free(wid); free(st_node); free(st_depth); free(path);
    return -1;
}
