"""GPU path vs CPU oracle, through the C ABI (msq_execute_batch -> mgpu_execute).

Bit-exact on: docid_map words, every per-document rank record, hit counts, aggregations,
merged rank-key order, facet counts.  Run on the GPU box: pytest -m gpu."""
import pytest

from marlin_amd import GpuEngine, Index, QueryError
from oracle import orc_py
from util import ROBOT_SUITES, compare_results, datagen_index, load_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("suite", ROBOT_SUITES)
def test_robot_suite_gpu(suite):
    fx = load_fixture(suite)
    ix = Index(fx["settings"], num_shards=fx["numShards"]).add_documents(fx["docs"])
    eng = GpuEngine(ix)
    try:
        pqs = []
        for case in fx["cases"]:
            if case["status"] == 400:
                continue
            jq = dict(case["query"]); jq["fullScan"] = True
            pqs.append(ix.query(jq))
        got = eng.search(pqs, capture=3)               # one batch for the whole suite
        for pq, g in zip(pqs, got):
            o = orc_py.search(ix, pq, flags=3)
            compare_results(g, o, ctx=(suite, pq.jq))
    finally:
        eng.close(); ix.close()


DG_QUERIES = [
    {"q": "aa"}, {"q": "bbbb"}, {"q": "aaaaaaa bbbbbbb"}, {"q": "aaaaaab bbbbbbc"},
    {"q": "aaaaaaa bbbbbbb ccccccc", "rankBy": {"num": "desc"}, "hitsPerPage": 10},
    {"q": "aaaaaaa bbbbbbb ccccccc ddddddd attr", "rankBy": {"id": "asc"}},
    {"q": "attr", "rankBy": {"id": "asc"}, "hitsPerPage": 50},
    {"q": "eeeeeee", "rankBy": {"id": "asc"}, "hitsPerPage": 50},
    {"q": "aaaaaaa attr", "sortBy": {"num": "asc"}, "hitsPerPage": 50},
    {"q": "zzzzzzz yyyyyyy", "rankBy": {"num": "desc"}, "page": 3, "hitsPerPage": 20},
    {"q": "", "rankBy": {"id": "desc"}, "filter": {"bool": {"$ne": True}}},
    {"q": "aaaaaaa", "rankBy": {"id": "asc"}, "filter": {"id": {"$gt": 100, "$lt": 105}}},
    {"q": "aaaaaaa", "rankBy": {"id": "asc"}, "filter": {"numlist": {"$gte": 25000}}},
    {"q": "", "rankBy": {"id": "asc"}, "filter": {"facetonly": {"$nin": ["threethree", "fourfour"]}}},
    {"q": "aaaaaaa", "rankBy": {"id": "asc"}, "filter": {"$or": [{"id": {"$lte": 45}}, {"facetonly": "fourfour"}]}},
    {"q": "bbbbbbb", "rankBy": {"num": "asc"}, "maxFacetResults": 15,
     "aggs": {"a": {"avg": {"field": "num"}}, "s": {"stats": {"field": "id"}}, "mx": {"max": {"field": "numlist"}},
              "mn": {"min": {"field": "num"}}, "c": {"cardinality": {"field": "facetlist"}}}},
    {"hitsPerPage": 0, "maxHits": 0, "aggs": {"a": {"avg": {"field": "num"}}}},
    {"q": "aaaaaaa", "rankAlgorithm": ["exact", "typos"], "rankBy": {"id": "asc"}},
]


@pytest.fixture(scope="module")
def dg():
    ix, _ = datagen_index(10000)
    eng = GpuEngine(ix)
    yield ix, eng
    eng.close(); ix.close()


def test_datagen_queries_gpu(dg):
    ix, eng = dg
    pqs = [ix.query(dict(jq, fullScan=True)) for jq in DG_QUERIES]
    got = eng.search(pqs, capture=3)
    for pq, g in zip(pqs, got):
        compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)


def test_facet_keys_by_separate_kernel(dg, monkeypatch):
    """facet selection keys come from k_fkeys instead of the select kernels themselves (the path a
    batch takes when a facet field or maxFacetResults is too large for the one-workgroup selections)"""
    ix, eng = dg
    monkeypatch.setenv("MGPU_NO_FUSED_FKEYS", "1")
    pqs = [ix.query(dict(jq, fullScan=True)) for jq in DG_QUERIES]
    got = eng.search(pqs, capture=3)
    for pq, g in zip(pqs, got):
        compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)


def test_facets_counted_without_the_rank_kernels_ordinals(dg, monkeypatch):
    """k_docproc looks every facet value up itself (the path of squeries whose rank stage reads no records)"""
    ix, eng = dg
    monkeypatch.setenv("MGPU_NO_FUSED_FACETS", "1")
    pqs = [ix.query(dict(jq, fullScan=True)) for jq in DG_QUERIES]
    got = eng.search(pqs, capture=3)
    for pq, g in zip(pqs, got):
        compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)


def test_sparse_facets_through_touched_lists(monkeypatch):
    """a sparse single-valued facet field counted by k_docproc with first-touch lists (the path of fields that
    also feed a cardinality aggregation, or whose documents can hold several values) instead of straight
    from the rank kernel"""
    import test_gpu_scale as sc
    lib = sc.ffi.load()
    shards = [lib.msynth_zipf(40_000, 2048, 20_000, 5, s, 2) for s in range(2)]
    ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    buf = sc.C.create_string_buffer(16)
    def zw(r):
        lib.msynth_zipf_word(2048, 5, r, buf)
        return buf.value.decode()
    qs = [{"q": zw(0), "maxFacetResults": 10}, {"q": zw(3) + " " + zw(40), "maxFacetResults": 5},
          {"q": zw(1), "aggs": {"c": {"cardinality": {"field": "hc"}}}, "maxFacetResults": 8},
          {"q": "", "sortBy": {"num": "asc"}, "maxFacetResults": 6},       # no words: the rank kernel still reads the records
          {"q": "", "maxFacetResults": 6},                                   # ... and here it does not: k_docproc looks up
          {"q": zw(2), "hitsPerPage": 0, "maxHits": 0, "maxFacetResults": 4}]   # facets only
    results = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("MGPU_NO_DIRECT_FACETS", env)
        eng = GpuEngine(ix)
        try:
            pqs = [ix.query(dict(q, fullScan=True)) for q in qs]
            got = eng.search(pqs, capture=3)
            for pq, g in zip(pqs, got):
                compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
            results.append([(g["totalHits"], g["facets"]) for g in got])
        finally:
            eng.close()
    ix.close()
    assert results[0] == results[1] and results[0][0][0] > 0


def test_many_facet_results_need_the_big_final_sort(dg):
    """2 * maxFacetResults above the one-workgroup selection's k: keys by k_fkeys, wide passes + big sort"""
    ix, eng = dg
    pqs = [ix.query({"q": "", "maxFacetResults": 300, "fullScan": True}),
           ix.query({"q": "aaaaaaa", "maxFacetResults": 200, "fullScan": True})]
    got = eng.search(pqs, capture=3)
    for pq, g in zip(pqs, got):
        compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)


def test_single_query_equals_batched(dg):
    ix, eng = dg
    pq = ix.query({"q": "aaaaaaa bbbbbbb", "rankBy": {"num": "desc"}, "fullScan": True})
    a = eng.search(pq)
    b = eng.search([pq, pq, pq])
    assert all(x["ranks"] == a["ranks"] and x["facets"] == a["facets"] for x in b)


def test_filter_rows_shared_within_a_batch_and_cardinality_sets_in_the_results(dg):
    """The squeries of one shard that name the same filter operand (a sparse posting, a numeric range) share one scattered
    row per batch; the id set of a cardinality aggregation over a small field comes back with the results download.  A batch
    that repeats and mixes filters and aggregations must answer query by query like the queries run alone, and like the
    oracle."""
    ix, eng = dg
    f1 = {"id": {"$lt": 4000}}
    f2 = {"$or": [{"id": {"$gte": 7000}}, {"facetonly": "fourfour"}]}
    f3 = {"$and": [{"id": {"$gte": 1000}}, {"id": {"$lt": 4000}}, {"facetonly": {"$ne": "threethree"}}]}
    card = {"c": {"cardinality": {"field": "facetlist"}}, "s": {"stats": {"field": "num"}}}
    jqs = [{"q": "aaaaaaa", "filter": f1}, {"q": "aaaaaab bbbbbbc", "filter": f1, "aggs": card}, {"q": "aa", "filter": f2},
           {"q": "aaaaaaa bbbbbbb", "filter": f3, "aggs": card}, {"q": "", "filter": f1, "aggs": card}, {"q": "aaaaaaa", "filter": f2},
           {"q": "bbbbbbb", "aggs": card}, {"q": "aaaaaaa", "filter": f3}]
    pqs = [ix.query(dict(jq, fullScan=True, rankBy={"id": "desc"})) for jq in jqs]
    together = eng.search(pqs, capture=3)
    for pq, g in zip(pqs, together):
        alone = eng.search(pq, capture=3)
        assert g["ranks"] == alone["ranks"] and g["facets"] == alone["facets"] and repr(g.get("aggs")) == repr(alone.get("aggs")), pq.jq
        compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)


def test_long_position_lists_take_the_overflow_path():
    """documents whose matched words or positions exceed the LDS slices of the rank kernel
    (MG_MCAP matched words, MG_PCAP positions) are ranked by the overflow kernel"""
    docs = []
    for i in range(300):
        rep = 1 + (i % 40)
        body = " ".join((["alpha"] * rep) + ["x%d" % j for j in range(i % 7)] + (["beta"] * (1 + i % 23)) + ["gamma"])
        docs.append({"id": i, "str": body, "other": "alpha beta gamma delta " * (i % 5), "num": i % 13})
    settings = {"indexedFields": ["str", "other", "num"], "facetFields": [], "hitsPerPage": 50}
    ix = Index(settings, num_shards=2).add_documents(docs)
    eng = GpuEngine(ix)
    try:
        qs = [{"q": "alpha beta"}, {"q": "alpha beta gamma"}, {"q": "beta alpha gamma delta"}, {"q": "alpha"},
              {"q": "alphx betb gamma", "rankBy": {"num": "desc"}}]
        pqs = [ix.query(dict(q, fullScan=True)) for q in qs]
        got = eng.search(pqs, capture=3)
        for pq, g in zip(pqs, got):
            compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
    finally:
        eng.close(); ix.close()
