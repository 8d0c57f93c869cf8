"""BASELINE.json's other configurations as parity cases (configs[2] at full size is test_gpu_fullsize.py and the
bench itself):
  configs[0]  100k documents of the reference's test/data_gen.py schema, 1 shard, 2-term query with typo tolerance
  configs[1]  1M documents, 1 shard, 3-term AND query + facet fields, top-10
  configs[3]  10M documents: 100k-bucket facet + numeric filter + avg / stats / cardinality aggs (properties)
  configs[4]  50M documents, 8 shards, a batch of common-term queries with typos (properties; no CPU oracle run
              at this size: partition sums, order, batching independence)
pytest -m gpu."""
import ctypes as C
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from marlin_amd import GpuEngine, Index, ffi
from oracle import orc_py
from util import compare_results, datagen_index

pytestmark = pytest.mark.gpu


def _against_oracle(ix, queries):
    eng = GpuEngine(ix)
    try:
        pqs = [ix.query(dict(q, fullScan=True)) for q in queries]
        got = eng.search(pqs, capture=3)
        for pq, g in zip(pqs, got):
            compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
        return got
    finally:
        eng.close()


def test_config0_100k_two_term_typo_query():
    ix, _ = datagen_index(100_000)
    try:
        got = _against_oracle(ix, [{"q": "aaaaaab bbbbbbc"}, {"q": "aaaaaab bbbbbbc", "rankBy": {"num": "desc"}, "hitsPerPage": 10}])
        assert got[0]["totalHits"] > 10_000
    finally:
        ix.close()


def test_config1_1m_three_term_and_with_facets():
    ix, _ = datagen_index(1_000_000)
    try:
        got = _against_oracle(ix, [{"q": "aaaaaaa bbbbbbb ccccccc", "hitsPerPage": 10, "maxFacetResults": 10}])
        assert got[0]["totalHits"] > 200_000 and len(got[0]["ranks"]) >= 10
    finally:
        ix.close()


def _typo(rng, w):
    p = rng.randrange(1, len(w))
    return w[:p] + rng.choice([ch for ch in "abcdefghijklmnopqrstuvwxyz" if ch != w[p]]) + w[p + 1:]


def _property_checks(ix, eng, base):
    qs = []
    for jq in base:
        qs.append(jq)
        qs.append(dict(jq, filter={"price": {"$lt": 300000}}))
        qs.append(dict(jq, filter={"price": {"$gte": 300000}}))
        qs.append(dict(jq, aggs={"a": {"avg": {"field": "price"}}, "s": {"stats": {"field": "num"}},
                                  "c": {"cardinality": {"field": "cat"}}}, maxFacetResults=64))
    res = eng.search([ix.query(q) for q in qs])
    hits = 0
    for i in range(0, len(res), 4):
        r, lo, hi, ag = res[i:i + 4]
        T = r["totalHits"]; hits += T
        assert lo["totalHits"] + hi["totalHits"] == T                     # the numeric filter partitions the hit set
        for x in (r, lo, hi):
            assert sum(e["count"] for e in x["facets"]["cat"]) <= x["totalHits"]
            cs = [e["count"] for e in x["facets"]["hc"]]
            assert cs == sorted(cs, reverse=True) and all(c >= 1 for c in cs)
        key = lambda x: (x["typos"], x["proximity"], x["field"], x["position"], -x["exact"], -x["compare"])
        ks = [key(x) for x in r["ranks"]]
        assert all(a < b for a, b in zip(ks, ks[1:]))                       # rankBy a unique number: strict order
        assert ag["totalHits"] == T and ag["aggs"]["s"]["count"] == T
        if T:
            assert 0 <= ag["aggs"]["a"]["value"] < 1_000_000
            # 64 cat values, one per document: the distinct count equals the number of non-empty facet entries
            assert ag["aggs"]["c"]["value"] == len(ag["facets"]["cat"]) <= 64
            assert sum(e["count"] for e in ag["facets"]["cat"]) == T
    assert hits > 0
    # batching independence
    one = eng.search(ix.query(base[0]))
    assert one["ranks"] == res[0]["ranks"] and one["facets"] == res[0]["facets"]


def test_config3_high_cardinality_facet_filter_aggs():
    lib = ffi.load()
    shards = bench.build_shards(lib, 10_000_000, list(range(bench.TOTAL_SHARDS)))
    ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    eng = GpuEngine(ix)
    lib.msq_set_plan_threads(8)
    try:
        _property_checks(ix, eng, bench.make_queries(lib, 6))
    finally:
        eng.close(); ix.close()


def test_config4_50m_common_term_batch():
    import random
    lib = ffi.load()
    shards = bench.build_shards(lib, 50_000_000, list(range(bench.TOTAL_SHARDS)))
    ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    eng = GpuEngine(ix)
    lib.msq_set_plan_threads(8)
    try:
        rng = random.Random(50)
        base = []
        for _ in range(6):
            ws = [_typo(rng, bench.zipf_word(lib, rng.randrange(0, 2000))) for _ in range(2)]
            base.append({"q": " ".join(ws), "fullScan": True, "rankBy": {"num": "desc"}, "hitsPerPage": 10, "maxFacetResults": 10})
        _property_checks(ix, eng, base)
    finally:
        eng.close(); ix.close()
