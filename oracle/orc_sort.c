/* orc_sort.c -- ORACLE (test infrastructure): the reference's sorting routines restated.
 *
 * The reference sorts rank records with a rule-driven comparator that has no docid
 * tie-break (sort.c:46-100), so which of several equal records lands where depends on the
 * exact sequence of swaps.  To make the oracle reproduce the reference's order even inside
 * tie classes, the three algorithms are restated step for step (index based):
 *   - rank_sort          sort.c:134-183  (klib introsort: median-of-3 quicksort, combsort when
 *                                         the depth budget runs out, final insertion sort)
 *   - rank_partial_sort  sort.c:185-214  (quickselect, Lomuto partition around a median of 3)
 *   - ks_partialsort     inc/ksort.h:260-285 as instantiated for facets in squery.c:17-18
 *   - ks_introsort       inc/ksort.h:180-230 as instantiated for facets in query.c:11-12
 * klib's ksort.h is (c) Attractive Chaos, MIT licence; the algorithm is public.
 * Cross-checked against the reference's own sort.c built into oracle/_ref (tests/test_ref_pin.py).
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

#define LT(x, y) m_rank_lt(&(x), &(y), rules)
#define SWAP(T, x, y) do { T t_ = (x); (x) = (y); (y) = t_; } while (0)

static void insertsort_d(m_docrank_t *a, size_t n, const int *rules) {   /* sort.c:103-110 */
    for (size_t i = 1; i < n; i++)
        for (size_t j = i; j > 0 && LT(a[j], a[j - 1]); j--) SWAP(m_docrank_t, a[j], a[j - 1]);
}

static void combsort_d(m_docrank_t *a, size_t n, const int *rules) {     /* sort.c:112-132 */
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap;
    size_t gap = n;
    do {
        if (gap > 2) {
            gap = (size_t)(gap / shrink);
            if (gap == 9 || gap == 10) gap = 11;
        }
        do_swap = 0;
        for (size_t i = 0; i + gap < n; i++) {
            size_t j = i + gap;
            if (LT(a[j], a[i])) { SWAP(m_docrank_t, a[i], a[j]); do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) insertsort_d(a, n, rules);
}

typedef struct { ptrdiff_t left, right; int depth; } isort_frame_t;

void orc_rank_sort(size_t n, m_docrank_t *a, const int *rules) {
    if (n < 1) return;
    if (n == 2) { if (LT(a[1], a[0])) SWAP(m_docrank_t, a[0], a[1]); return; }
    int d;
    for (d = 2; (1ul << d) < n; ++d);
    isort_frame_t *stack = malloc(sizeof(isort_frame_t) * ((sizeof(size_t) * (size_t)d) + 2));
    isort_frame_t *top = stack;
    ptrdiff_t s = 0, t = (ptrdiff_t)n - 1;
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { combsort_d(a + s, (size_t)(t - s + 1), rules); t = s; continue; }
            ptrdiff_t i = s, j = t, k = i + ((j - i) >> 1) + 1;
            if (LT(a[k], a[i])) { if (LT(a[k], a[j])) k = j; }
            else k = LT(a[j], a[i]) ? i : j;
            m_docrank_t rp = a[k];
            if (k != t) SWAP(m_docrank_t, a[k], a[t]);
            for (;;) {
                do ++i; while (LT(a[i], rp));
                do --j; while (i <= j && LT(rp, a[j]));
                if (j <= i) break;
                SWAP(m_docrank_t, a[i], a[j]);
            }
            SWAP(m_docrank_t, a[i], a[t]);
            if (i - s > t - i) {
                if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == stack) { free(stack); insertsort_d(a, n, rules); return; }
            --top; s = top->left; t = top->right; d = top->depth;
        }
    }
}

static size_t qspartition_d(m_docrank_t *a, size_t s, size_t e, size_t p, const int *rules) {
    m_docrank_t pv = a[p];
    size_t fi = s;
    SWAP(m_docrank_t, a[e], a[p]);
    for (size_t i = s; i < e; ++i) {
        if (LT(a[i], pv)) { SWAP(m_docrank_t, a[i], a[fi]); ++fi; }
    }
    SWAP(m_docrank_t, a[fi], a[e]);
    return fi;
}

static size_t medianofthree_d(m_docrank_t *a, size_t s, size_t e, const int *rules) {
    const size_t m = (s + e) / 2;
    if (!LT(a[s], a[m])) SWAP(m_docrank_t, a[m], a[s]);
    if (!LT(a[m], a[e])) SWAP(m_docrank_t, a[m], a[e]);
    if (!LT(a[s], a[e])) SWAP(m_docrank_t, a[s], a[e]);
    return m;
}

void orc_rank_partial_sort(m_docrank_t *a, size_t l, size_t r, size_t m, const int *rules) {
    for (;;) {                                   /* the reference recurses in tail position */
        if (l == r) return;
        const size_t p = qspartition_d(a, l, r, medianofthree_d(a, l, r, rules), rules);
        if (p == m) return;
        if (m < p) r = p - 1; else l = p + 1;
    }
}

/* ---- facets: comparator (a).count > (b).count, squery.c:17, query.c:11 ---- */
#define FGT(x, y) ((x).count > (y).count)

static size_t facet_partition(m_facet_count_t *a, size_t l, size_t r) {   /* ksort.h:260-274 */
    ptrdiff_t lpos = (ptrdiff_t)l, ll = (ptrdiff_t)l - 1, hh = (ptrdiff_t)r, low = (ptrdiff_t)r;
    for (;;) {
        do ++ll; while (FGT(a[ll], a[low]));
        do --hh; while (FGT(a[low], a[hh]) && (hh != lpos));
        if (ll >= hh) break;
        SWAP(m_facet_count_t, a[ll], a[hh]);
    }
    SWAP(m_facet_count_t, a[ll], a[low]);
    return (size_t)ll;
}

void orc_facet_partialsort(m_facet_count_t *a, size_t l, size_t r, size_t m) {   /* ksort.h:275-285 */
    for (;;) {
        if ((r == m - 1) || (l == m + 1)) return;
        int i = (int)facet_partition(a, l, r);
        if ((size_t)i < m) l = (size_t)i + 1; else r = (size_t)i - 1;
    }
}

static void facet_insertsort(m_facet_count_t *a, size_t n) {
    for (size_t i = 1; i < n; i++)
        for (size_t j = i; j > 0 && FGT(a[j], a[j - 1]); j--) SWAP(m_facet_count_t, a[j], a[j - 1]);
}

static void facet_combsort(m_facet_count_t *a, size_t n) {
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap;
    size_t gap = n;
    do {
        if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        do_swap = 0;
        for (size_t i = 0; i + gap < n; i++) {
            size_t j = i + gap;
            if (FGT(a[j], a[i])) { SWAP(m_facet_count_t, a[i], a[j]); do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) facet_insertsort(a, n);
}

void orc_facet_introsort(size_t n, m_facet_count_t *a) {                   /* ksort.h:180-230 */
    if (n < 1) return;
    if (n == 2) { if (FGT(a[1], a[0])) SWAP(m_facet_count_t, a[0], a[1]); return; }
    int d;
    for (d = 2; (1ul << d) < n; ++d);
    isort_frame_t *stack = malloc(sizeof(isort_frame_t) * ((sizeof(size_t) * (size_t)d) + 2));
    isort_frame_t *top = stack;
    ptrdiff_t s = 0, t = (ptrdiff_t)n - 1;
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { facet_combsort(a + s, (size_t)(t - s + 1)); t = s; continue; }
            ptrdiff_t i = s, j = t, k = i + ((j - i) >> 1) + 1;
            if (FGT(a[k], a[i])) { if (FGT(a[k], a[j])) k = j; }
            else k = FGT(a[j], a[i]) ? i : j;
            m_facet_count_t rp = a[k];
            if (k != t) SWAP(m_facet_count_t, a[k], a[t]);
            for (;;) {
                do ++i; while (FGT(a[i], rp));
                do --j; while (i <= j && FGT(rp, a[j]));
                if (j <= i) break;
                SWAP(m_facet_count_t, a[i], a[j]);
            }
            SWAP(m_facet_count_t, a[i], a[t]);
            if (i - s > t - i) {
                if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == stack) { free(stack); facet_insertsort(a, n); return; }
            --top; s = top->left; t = top->right; d = top->depth;
        }
    }
}
