"""The oracle's restated primitives against the reference's own sources, compiled unmodified
into oracle/_ref/libmarlin_ref.so (`make ref`; bmap.c, cont.c, sort.c, hashtable.c, farmhash-c.c).

The library only exists where it was built from /root/reference (the build container); it travels
to the GPU box with the tree.  If it is absent these tests are skipped, and the oracle stays
pinned by the robot / pytests known answers (test_oracle_golden.py, test_oracle_datagen.py)."""
import ctypes as C
import os
import random

import pytest

from marlin_amd import ffi
from oracle import orc_py
from util import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "libmarlin_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def libs():
    ref = C.CDLL(REF)
    orc = orc_py.load()
    host = ffi.load()
    vp, u32 = C.c_void_p, C.c_uint32
    ref.bmap_new.restype = vp
    ref.bmap_add.argtypes = [vp, u32]
    for f in ("bmap_and", "bmap_andnot"):
        getattr(ref, f).restype = vp; getattr(ref, f).argtypes = [vp, vp]
    ref.bmap_cardinality.restype = u32; ref.bmap_cardinality.argtypes = [vp]
    ref.bmap_and_cardinality.restype = u32; ref.bmap_and_cardinality.argtypes = [vp, vp]
    ref.bmap_exists.restype = C.c_bool; ref.bmap_exists.argtypes = [vp, u32]
    ref.bmap_free.argtypes = [vp]
    ref.bmap_iterate.argtypes = [vp, vp, vp]
    ref.oper_new.restype = vp
    ref.oper_add.argtypes = [vp, vp]
    ref.oper_or.restype = vp; ref.oper_or.argtypes = [vp]
    ref.oper_and.restype = vp; ref.oper_and.argtypes = [vp]
    ref.oper_free.argtypes = [vp]
    ref.rank_sort.argtypes = [C.c_size_t, C.POINTER(ffi.DocRank), C.POINTER(C.c_int)]
    ref.rank_partial_sort.argtypes = [C.POINTER(ffi.DocRank), C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    ref.hashtable_new.restype = vp; ref.hashtable_new.argtypes = [u32]
    ref.hashtable_insert.restype = C.POINTER(u32 * 2); ref.hashtable_insert.argtypes = [vp, u32]
    ref.hashtable_lookup.restype = C.POINTER(u32 * 2); ref.hashtable_lookup.argtypes = [vp, u32]
    ref.hashtable_free.argtypes = [vp]
    ref.farmhash32_mk.restype = u32; ref.farmhash32_mk.argtypes = [C.c_char_p, C.c_size_t]
    orc.orc_rank_sort.argtypes = [C.c_size_t, C.POINTER(ffi.DocRank), C.POINTER(C.c_int)]
    orc.orc_rank_partial_sort.argtypes = [C.POINTER(ffi.DocRank), C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    orc.orc_hashtable_new.restype = vp; orc.orc_hashtable_new.argtypes = [u32]
    orc.orc_hashtable_insert.restype = C.POINTER(u32 * 2); orc.orc_hashtable_insert.argtypes = [vp, u32]
    orc.orc_hashtable_lookup.restype = C.POINTER(u32 * 2); orc.orc_hashtable_lookup.argtypes = [vp, u32]
    orc.orc_hashtable_free.argtypes = [vp]
    for f in ("orc_bm_or", "orc_bm_and", "orc_bm_andnot"):
        getattr(orc, f).argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t]
    orc.orc_bm_cardinality.restype = C.c_uint64
    orc.orc_bm_cardinality.argtypes = [C.POINTER(C.c_uint64), C.c_size_t]
    orc.orc_bm_from_list.argtypes = [C.POINTER(C.c_uint64), C.POINTER(u32), C.c_uint64]
    return ref, orc, host


ITER = C.CFUNCTYPE(None, C.c_uint32, C.c_void_p)


def ref_bmap(ref, items):
    b = ref.bmap_new()
    for x in items:
        ref.bmap_add(b, x)
    return b


def ref_items(ref, b):
    out = []
    cb = ITER(lambda v, p: out.append(v))
    ref.bmap_iterate(b, cb, None)
    return out


def dense(orc, items, words):
    arr = (C.c_uint64 * words)()
    lst = (C.c_uint32 * max(1, len(items)))(*items)
    orc.orc_bm_from_list(arr, lst, len(items))
    return arr


def dense_items(arr):
    return [w * 64 + b for w in range(len(arr)) for b in range(64) if (arr[w] >> b) & 1]


def make_sets(rng):
    """array containers, bitset containers (> 4096 per 64k chunk), empty, and chunk-straddling sets"""
    N = 200_000
    yield N, sorted(rng.sample(range(1, N), 50)), sorted(rng.sample(range(1, N), 70))
    yield N, sorted(rng.sample(range(1, N), 60_000)), sorted(rng.sample(range(1, N), 500))
    yield N, sorted(rng.sample(range(1, N), 90_000)), sorted(rng.sample(range(1, N), 80_000))
    yield N, [], sorted(rng.sample(range(1, N), 10))
    yield N, list(range(65_000, 66_100)), list(range(65_500, 140_000, 3))


def test_bitmap_ops_match_bmap_c(libs):
    ref, orc, _ = libs
    rng = random.Random(5)
    for N, A, B in make_sets(rng):
        words = (N + 63) // 64
        ra, rb = ref_bmap(ref, A), ref_bmap(ref, B)
        for opname, rfn, ofn in (("and", ref.bmap_and, orc.orc_bm_and), ("andnot", ref.bmap_andnot, orc.orc_bm_andnot)):
            rr = rfn(ra, rb)
            da, db = dense(orc, A, words), dense(orc, B, words)
            ofn(da, db, words)
            assert ref_items(ref, rr) == dense_items(da), opname
            assert ref.bmap_cardinality(rr) == orc.orc_bm_cardinality(da, words)
            ref.bmap_free(rr)
        # oper_or (bmap.c:455-...) -- the term-union of process_termresult / get_term_docids
        op = ref.oper_new(); ref.oper_add(op, ra); ref.oper_add(op, rb)
        rr = ref.oper_or(op)
        da, db = dense(orc, A, words), dense(orc, B, words)
        orc.orc_bm_or(da, db, words)
        assert ref_items(ref, rr) == dense_items(da)
        assert ref.bmap_and_cardinality(ra, rb) == len(set(A) & set(B))
        ref.bmap_free(rr); ref.oper_free(op)
        assert ref_items(ref, ra) == A                       # bmap_iterate is ascending: the hit list order
        ref.bmap_free(ra); ref.bmap_free(rb)


def rand_ranks(rng, n, spread):
    arr = (ffi.DocRank * n)()
    for i in range(n):
        r = arr[i]
        r.docid = i + 1
        r.typos = rng.randrange(spread); r.proximity = rng.randrange(spread * 2)
        r.position = rng.randrange(spread * 3); r.exact = rng.randrange(3); r.field = rng.randrange(2)
        r.compare = float(rng.randrange(spread * 50))
    return arr


def rank_tuple(r):
    return (r.docid, r.typos, r.proximity, r.position, r.exact, r.field, r.compare)


RULESETS = [
    [ffi.MR_TYPO, ffi.MR_PROX, ffi.MR_FIELD, ffi.MR_POS, ffi.MR_EXACT, ffi.MR_DONE],
    [ffi.MR_TYPO, ffi.MR_PROX, ffi.MR_FIELD, ffi.MR_POS, ffi.MR_EXACT, ffi.MR_COMP, ffi.MR_DONE],
    [ffi.MR_COMP_ASC, ffi.MR_DONE],
    [ffi.MR_EXACT, ffi.MR_TYPO, ffi.MR_COMP, ffi.MR_DONE],
]


def test_rank_sort_matches_sort_c_including_tie_order(libs):
    """identical element order, ties included: the restated sort moves elements exactly like sort.c"""
    ref, orc, _ = libs
    rng = random.Random(11)
    for rules in RULESETS:
        ra = (C.c_int * ffi.M_MAX_RULES)(*(rules + [ffi.MR_DONE] * (ffi.M_MAX_RULES - len(rules))))
        for n, spread in ((1, 2), (2, 2), (17, 2), (100, 3), (101, 2), (1500, 3), (5000, 40)):
            a = rand_ranks(rng, n, spread)
            b = (ffi.DocRank * n)(); C.memmove(b, a, C.sizeof(a))
            ref.rank_sort(n, a, ra); orc.orc_rank_sort(n, b, ra)
            assert [rank_tuple(x) for x in a] == [rank_tuple(x) for x in b], (rules, n)
            a = rand_ranks(rng, n, spread)
            b = (ffi.DocRank * n)(); C.memmove(b, a, C.sizeof(a))
            m = min(n - 1, 40)
            ref.rank_partial_sort(a, 0, n - 1, m, ra); orc.orc_rank_partial_sort(b, 0, n - 1, m, ra)
            assert [rank_tuple(x) for x in a] == [rank_tuple(x) for x in b], (rules, n, "partial")


def test_hashtable_matches_hashtable_c(libs):
    """same cell placement and growth, so facet iteration order (squery.c:411-437) is the same"""
    ref, orc, _ = libs
    rng = random.Random(3)
    hr, ho = ref.hashtable_new(16), orc.orc_hashtable_new(16)
    keys = [rng.randrange(1, 1 << 32) for _ in range(5000)] + list(range(1, 300))
    for k in keys:
        a, b = ref.hashtable_insert(hr, k), orc.orc_hashtable_insert(ho, k)
        a.contents[1] += 1; b.contents[1] += 1
    class HT(C.Structure):
        _fields_ = [("cells", C.POINTER(C.c_uint32 * 2)), ("size", C.c_uint32), ("pop", C.c_uint32)]
    r, o = HT.from_address(hr), HT.from_address(ho)
    assert (r.size, r.pop) == (o.size, o.pop)
    assert [tuple(r.cells[i]) for i in range(r.size)] == [tuple(o.cells[i]) for i in range(o.size)]
    for k in keys[:200] + [7_777_777, 123_456_789]:
        a, b = ref.hashtable_lookup(hr, k), orc.orc_hashtable_lookup(ho, k)
        assert bool(a) == bool(b) and (not a or tuple(a.contents) == tuple(b.contents))
    ref.hashtable_free(hr); orc.orc_hashtable_free(ho)


def test_farmhash32_mk_matches_farmhash_c(libs):
    ref, _, host = libs
    rng = random.Random(9)
    for n in list(range(0, 40)) + [63, 64, 65, 100, 255, 1000]:
        s = bytes(rng.randrange(256) for _ in range(n))
        assert ref.farmhash32_mk(s, n) == host.m_farmhash32_mk(s, n), n
