"""The bench workload at BASELINE.json's full size (10M documents, 8 shards) checked through
properties that need no CPU oracle run: partition sums, order, cross-kernel consistency,
batching independence.  pytest -m gpu (about 20 s, most of it building the corpus)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                     # corpus + query generators of the benchmark
from marlin_amd import GpuEngine, Index, ffi

pytestmark = pytest.mark.gpu
DOCS = 10_000_000


@pytest.fixture(scope="module")
def full():
    lib = ffi.load()
    shards = bench.build_shards(lib, DOCS, list(range(bench.TOTAL_SHARDS)))
    ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    eng = GpuEngine(ix)
    lib.msq_set_plan_threads(8)
    yield lib, ix, eng
    eng.close(); ix.close()


def test_full_size_properties(full):
    lib, ix, eng = full
    base = bench.make_queries(lib, 12)
    qs = []
    for jq in base:
        jq = dict(jq, maxFacetResults=64, hitsPerPage=50)
        qs.append(jq)                                                           # 0: plain
        qs.append(dict(jq, filter={"flag": True}))                              # 1: partition a
        qs.append(dict(jq, filter={"flag": False}))                             # 2: partition b
        srt = {k: v for k, v in jq.items() if k != "rankBy"}
        qs.append(dict(srt, sortBy={"num": "desc"},
                       aggs={"mx": {"max": {"field": "num"}}, "st": {"stats": {"field": "price"}}}))              # 3
    pqs = [ix.query(q) for q in qs]
    res = eng.search(pqs)
    some_hits = 0
    for i in range(0, len(res), 4):
        r, ra, rb, rg = res[i:i + 4]
        T = r["totalHits"]
        some_hits += T
        # the two filter halves partition the unfiltered hit set
        assert ra["totalHits"] + rb["totalHits"] == T
        # every document carries exactly one `cat` value and there are 64 of them: the counts add up
        assert sum(e["count"] for e in r["facets"]["cat"]) == T
        for half in (ra, rb):
            assert sum(e["count"] for e in half["facets"]["cat"]) == half["totalHits"]
        # facet lists are ordered by count
        for f in ("cat", "hc"):
            cs = [e["count"] for e in r["facets"][f]]
            assert cs == sorted(cs, reverse=True)
        # default rules then rankBy num desc (a unique field): the merged list is strictly ordered
        rk = lambda x: (x["typos"], x["proximity"], x["field"], x["position"], -x["exact"], -x["compare"])
        keys = [rk(x) for x in r["ranks"]]
        assert all(a < b for a, b in zip(keys, keys[1:]))
        assert len(keys) <= bench.TOTAL_SHARDS * 100          # per-shard top 2N, N = 50
        # sortBy num desc: strictly decreasing values, and the head equals the max aggregate
        vals = [x["compare"] for x in rg["ranks"]]
        assert all(a > b for a, b in zip(vals, vals[1:]))
        if T:
            assert rg["totalHits"] == T
            assert rg["aggs"]["mx"]["value"] == vals[0]
            assert rg["aggs"]["st"]["count"] == T
            assert rg["aggs"]["st"]["min"] >= 0 and rg["aggs"]["st"]["max"] < 1_000_000
        # filtered ranks are the unfiltered ranks restricted to the half (same order)
        key = lambda x: (x["shard"], x["docid"])
        head = [key(x) for x in r["ranks"]][:10]
        merged = sorted(ra["ranks"] + rb["ranks"], key=rk)[:10]
        assert [key(x) for x in merged] == head
    assert some_hits > 0


def test_full_size_single_equals_batched(full):
    lib, ix, eng = full
    pqs = [ix.query(q) for q in bench.make_queries(lib, 6)]
    batched = eng.search(pqs)
    for pq, b in zip(pqs, batched):
        a = eng.search(pq)
        assert a["ranks"] == b["ranks"] and a["facets"] == b["facets"] and a["totalHits"] == b["totalHits"]
