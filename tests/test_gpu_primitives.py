"""Kernel-level entry points and error behaviour of the C ABI on the GPU.  pytest -m gpu."""
import ctypes as C
import random

import pytest

from marlin_amd import GpuEngine, GpuError, Index, ffi
from oracle import orc_py
from util import datagen_index

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dg():
    ix, _ = datagen_index(20000)
    eng = GpuEngine(ix)
    yield ix, eng
    eng.close(); ix.close()


def posting(lib, shard, wid):
    docs = C.POINTER(C.c_uint32)()
    n = lib.mshard_wid_posting(shard, wid, C.byref(docs))
    return set(docs[i] for i in range(n))


def test_posting_ops_match_set_algebra(dg):
    """mgpu_posting_op = bmap_and / oper_or / bmap_andnot + bmap_cardinality on mirrored postings,
    dense rows and sparse lists alike"""
    ix, eng = dg
    lib = eng.lib
    mir = ffi.Mirror.from_address(eng.mirrors[0].value)
    shard = ix.shards[0]
    rng = random.Random(4)
    wids = [w for w in range(1, mir.W + 1)]
    # a mix of frequent (dense) and rare (sparse) words
    sized = sorted(wids, key=lambda w: -len(posting(lib, shard, w)))
    picks = sized[:6] + rng.sample(sized[len(sized) // 2:], 6)
    maxdoc = max(max(posting(lib, shard, w) or {0}) for w in picks)
    nwords = maxdoc // 64 + 1
    for a in picks[:4] + picks[-4:]:
        for b in picks[2:6] + picks[-3:]:
            A, B = posting(lib, shard, a), posting(lib, shard, b)
            for op, want in ((0, A & B), (1, A | B), (2, A - B)):
                out = (C.c_uint64 * nwords)()
                card = C.c_uint64()
                rc = lib.mgpu_posting_op(eng.ctx, mir.g, op, a, b, out, nwords, C.byref(card))
                assert rc == 0, lib.mgpu_last_error()
                got = {w * 64 + i for w in range(nwords) for i in range(64) if (out[w] >> i) & 1}
                assert got == want and card.value == len(want), (a, b, op)
    bad = C.c_uint64()
    assert lib.mgpu_posting_op(eng.ctx, mir.g, 0, mir.num_postings, 1, None, 0, C.byref(bad)) == -2      # MGPU_E_ARG


def test_unsupported_mode_is_reported_not_approximated(dg):
    ix, eng = dg
    # fast-rank with a cardinality aggregation (the reference dereferences NULL there, metric-aggs.c:23-45): refused, no fallback
    pq = ix.query({"q": "", "rankBy": {"id": "asc"}, "fullScanThreshold": 5000, "aggs": {"c": {"cardinality": {"field": "facet"}}}})
    with pytest.raises(GpuError) as e:
        eng.search(pq)
    assert "-5" in str(e.value)                                     # MGPU_E_UNSUPPORTED
    # the context is still usable afterwards
    pq = ix.query({"q": "aaaaaaa", "fullScan": True})
    assert eng.search(pq)["totalHits"] == orc_py.search(ix, pq)["totalHits"]


def test_empty_and_degenerate_inputs(dg):
    ix, eng = dg
    qs = [{"q": "zzzzqqqqxxxx"},                                   # no term matches anything
          {"q": "", "filter": {"id": {"$gt": 10 ** 9}}},           # filter selects nothing
          {"q": "", "hitsPerPage": 0, "maxFacetResults": 0},       # nothing requested but the count
          {"q": "aaaaaaa", "filter": {"$and": [{"bool": True}, {"bool": False}]}}]
    pqs = [ix.query(dict(q, fullScan=True)) for q in qs]
    got = eng.search(pqs, capture=1)
    for pq, g in zip(pqs, got):
        o = orc_py.search(ix, pq, flags=1)
        assert g["totalHits"] == o["totalHits"] and g["ranks"] == o["ranks"] and g["facets"] == o["facets"], pq.jq
        assert [s["docid_map"] for s in g["_shards"]] == [s["docid_map"] for s in o["_shards"]]
