"""f-3: the reference's default ranking mode above fullScanThreshold hits (perform_fast_ranking, docrank.c:478-560):
zero-typo documents first, then the first 1000 in num_dbi[rankBy] order, then every n-th document; facet counts of the
top 2N recounted exactly (squery.c:398-408, 439-446); aggregations over all hits through idnum2dbl.
GPU vs the oracle's restatement, plus the one known answer the reference's own suite holds for the mode
(test/robot/query.robot:37-50: 6000 documents, fullScanThreshold 5000, rankBy num asc, empty query -> totalHits 6000).
pytest -m gpu."""
import pytest

from marlin_amd import GpuEngine, GpuError, Index
from oracle import orc_py
from util import compare_results, datagen_index

pytestmark = pytest.mark.gpu


def check(ix, eng, q, expect_fast=True):
    pq = ix.query(q)
    g = eng.search(pq, capture=1)
    o = orc_py.search(ix, pq, flags=1)
    assert o["fullScan"] == (not expect_fast), q
    assert g["fullScan"] == o["fullScan"], q
    compare_results(g, o, ctx=q)
    # rankBy on a unique number: no ties, so the documents themselves must agree
    assert [(r["shard"], r["docid"]) for r in g["ranks"]] == [(r["shard"], r["docid"]) for r in o["ranks"]], q
    return g


def test_datagen_100k_default_mode():
    ix, _ = datagen_index(100000)
    eng = GpuEngine(ix)
    try:
        # maxFacetResults 500: 2N covers every value of every facet field, so no tie at a cut-off decides what is recounted
        base = {"maxFacetResults": 500}
        g = check(ix, eng, dict(base, q="aaaaaab bbbbbbc", rankBy={"num": "desc"}))
        assert g["totalHits"] == 75000 and [r["docid"] for r in g["ranks"][:3]] == [2, 3, 4]
        check(ix, eng, dict(base, q="aa", rankBy={"num": "asc"}, aggs={"s": {"stats": {"field": "num"}}, "a": {"avg": {"field": "id"}},
                                                                        "mx": {"max": {"field": "num"}}, "mn": {"min": {"field": "id"}}}))
        check(ix, eng, dict(base, q="", rankBy={"id": "desc"}, hitsPerPage=40, page=3))
        # (a filter that removes more than ~5000 matches makes the reference overrun its rank buffer: the sample is
        # drawn from the UNFILTERED zero-typo map at a rate set by the filtered count)
        check(ix, eng, dict(base, q="aaaaaaa bbbbbbb", sortBy={"num": "asc"}, filter={"id": {"$gt": 3000}}))
        # under the threshold, or fullScan asked for: the ordinary path
        check(ix, eng, dict(base, q="aaaaaaa bbbbbbb ccccccc ddddddd", rankBy={"num": "desc"}, fullScanThreshold=30000), expect_fast=False)
        check(ix, eng, dict(base, q="aa", rankBy={"num": "desc"}, fullScan=True), expect_fast=False)
        # the reference misbehaves here; refused, not approximated
        for q, code in (({"q": "aa", "rankBy": {"num": "desc"}, "aggs": {"c": {"cardinality": {"field": "facet"}}}}, "-5"),
                        ({"q": "", "filter": {"bool": True}, "rankBy": {"num": "desc"}, "fullScanThreshold": 5000}, "-3"),   # sample outgrows the rank buffer
                        ({"q": "aa", "rankBy": {"numlist": "desc"}}, "-3")):                                                 # several values per document
            with pytest.raises(GpuError) as e:
                eng.search(ix.query(q))
            assert code in str(e.value), (q, str(e.value))
        check(ix, eng, dict(base, q="aa", rankBy={"num": "desc"}))            # the context is usable afterwards
    finally:
        eng.close(); ix.close()


def test_query_robot_known_answer():
    ix, _ = datagen_index(6000)
    eng = GpuEngine(ix)
    try:
        g = check(ix, eng, {"q": "", "fullScanThreshold": 5000, "rankBy": {"num": "asc"}, "maxFacetResults": 500})
        assert g["totalHits"] == 6000                                         # query.robot:45-50
        pq = ix.query({"q": "blahblahblahblahblahblahblahblah", "fullScanThreshold": 5000, "rankBy": {"num": "asc"}})
        assert eng.search(pq)["totalHits"] == 0                               # query.robot:57-58
    finally:
        eng.close(); ix.close()


@pytest.mark.parametrize("n_exact,shards", [(0, 1), (600, 1), (1000, 1), (1001, 2), (5000, 2)])
def test_zero_typo_preference(n_exact, shards):
    """n_exact documents hold the query word itself, the others a word one edit away: the zero-typo map is empty, at most
    1000 (all of it ranked first; at exactly 1000 the ordered walk still adds one), or larger (it replaces docid_map)"""
    N = 30000
    step = N // n_exact if n_exact else 0
    docs = []
    exact = 0
    for i in range(N):
        is_exact = n_exact and i % step == 0 and exact < n_exact * 1
        exact += 1 if is_exact else 0
        docs.append({"id": i, "t": ("alphabet" if is_exact else "alphabex") + (" soup" if i % 3 == 0 else " stew"),
                     "cat": "c%d" % (i % 7), "price": float((i * 7919) % N), "flag": i % 5 != 0})
    ix = Index({"indexedFields": ["t", "price", "flag"], "facetFields": ["cat"]}, num_shards=shards).add_documents(docs)
    eng = GpuEngine(ix)
    try:
        thr = 5000 if shards > 1 else 25000
        check(ix, eng, {"q": "alphabet", "rankBy": {"price": "desc"}, "fullScanThreshold": thr})
        check(ix, eng, {"q": "alphabet", "rankBy": {"price": "asc"}, "fullScanThreshold": thr, "aggs": {"s": {"stats": {"field": "price"}}}})
        # a filter thins docid_map, not the zero-typo map the sample prefers (the reference's quirk): documents that
        # fail the filter get ranked all the same
        check(ix, eng, {"q": "alphabet", "rankBy": {"price": "desc"}, "fullScanThreshold": 5000, "filter": {"flag": True}})
    finally:
        eng.close(); ix.close()


def test_fast_rank_with_a_sparse_single_valued_facet_field(monkeypatch):
    """A field with more values than an LDS histogram holds, one value per document, is counted straight from the rank
    kernel -- or, for squeries of few hits, by hashing the hits' ordinals (k_facet_hash).  k_order files the hash jobs by
    the hit count BEFORE a sample replaces the hit list; the kernels that skip their own counting for hashed squeries look
    at the count AFTER it.  A sampled squery used to fall between the two and came back with an empty facet list (found in
    round 3): a squery whose threshold lies below the hash jobs' reach (as here: 5000 < 8192) names no hash jobs any more.  The sample's counts of a 20000-value field are ties nearly
    everywhere, so which values are listed is the canonical choice, not the reference's (ties): the lists are
    checked for what must hold whatever the choice -- every listed count is the EXACT number of matching documents with
    that value (the recount, squery.c:398-408), in descending order -- and against the other counting paths."""
    import ctypes as C
    from marlin_amd import ffi
    lib = ffi.load()
    V, S = 2048, 5
    shards = [lib.msynth_zipf(140_000, V, 20_000, S, s, 2) for s in range(2)]
    ix = Index.from_shards({"fullScanThreshold": 5000}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    eng = GpuEngine(ix)
    buf = C.create_string_buffer(16)

    def word(r):
        lib.msynth_zipf_word(V, S, r, buf)
        return buf.value.decode()
    try:
        for q in ({"q": word(0), "rankBy": {"num": "desc"}, "maxFacetResults": 10},
                  {"q": word(1) + " " + word(2)[:3], "rankBy": {"num": "asc"}, "maxFacetResults": 10},
                  {"q": "", "rankBy": {"num": "desc"}, "maxFacetResults": 5}):
            pq = ix.query(q)
            g = eng.search(pq, capture=1)
            o = orc_py.search(ix, pq, flags=1)
            assert not g["fullScan"] and not o["fullScan"] and g["totalHits"] == o["totalHits"], q
            assert g["facets"]["cat"] == o["facets"]["cat"], q          # 64 values: the list covers them all, no tie decides
            hc = g["facets"]["hc"]
            assert len(hc) == q["maxFacetResults"], (q, hc)
            assert [e["count"] for e in hc] == sorted((e["count"] for e in hc), reverse=True), q
            # per shard (the merged count of a value adds only what the shards LISTED, query.c:61-91)
            for s in range(ix.num_shards):
                lst = g["_shards"][s]["facets"][1]
                assert len(lst) == 2 * q["maxFacetResults"], (q, s)
                for fid, cnt in lst[:3]:
                    exact = eng.search(ix.query({"q": q["q"], "filter": {"hc": ix.facet_string(s, fid)}, "fullScan": True, "hitsPerPage": 0}), capture=1)
                    assert exact["_shards"][s]["num_hits"] == cnt, (q, s, fid)
            for switch in ("MGPU_NO_HASH_FACETS", "MGPU_NO_DIRECT_FACETS", "MGPU_NO_FUSED_FKEYS"):
                monkeypatch.setenv(switch, "1")
                assert eng.search(pq)["facets"] == g["facets"], (q, switch)
                monkeypatch.delenv(switch)
    finally:
        eng.close(); ix.close()


def test_sparse_facet_field_of_a_squery_with_a_large_hit_bound(monkeypatch):
    """A squery whose hit BOUND (hit_cap, from posting cardinalities: often far above the hit count) exceeds 131072 counts a
    sparse single-valued field with per-hit atomics in k_docproc and selects over the touched bins; MGPU_FDIRECT_CAP lifts the
    bound, and the rank kernel counts the field itself (k_facet_hash for few hits; for 135k hits the derived keys are
    selected by k_select_mid: nothing else can derive them).  One shard of 300k documents, both routes and the plain
    histogram route against the oracle and each other."""
    import ctypes as C
    from marlin_amd import ffi
    lib = ffi.load()
    V, S = 2048, 5
    shards = [lib.msynth_zipf(300_000, V, 20_000, S, 0, 1)]
    ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    eng = GpuEngine(ix)
    buf = C.create_string_buffer(16)

    def word(r):
        lib.msynth_zipf_word(V, S, r, buf)
        return buf.value.decode()
    try:
        sizes = []
        for q in ({"q": word(0), "rankBy": {"num": "desc"}, "maxFacetResults": 10, "fullScan": True},
                  {"q": word(0) + " " + word(1) + " " + word(900)[:2], "maxFacetResults": 10, "fullScan": True},
                  {"q": word(0)[:1] + " " + word(700), "rankBy": {"price": "asc"}, "maxFacetResults": 4, "fullScan": True},
                  {"q": word(2), "maxFacetResults": 25, "fullScan": True}):
            pq = ix.query(q)
            g = eng.search(pq, capture=1)
            o = orc_py.search(ix, pq, flags=1)
            assert g["fullScan"], q
            compare_results(g, o, ctx=q)
            sizes.append(g["_shards"][0]["num_hits"])
            for name, val in (("MGPU_FDIRECT_CAP", "8388608"), ("MGPU_NO_DIRECT_FACETS", "1"), ("MGPU_NO_HASH_FACETS", "1")):
                monkeypatch.setenv(name, val)
                g2 = eng.search(pq, capture=1)
                monkeypatch.delenv(name)
                assert [x["facets"] for x in g2["_shards"]] == [x["facets"] for x in g["_shards"]], (q, name)
        assert max(sizes) > 131072 and min(sizes) < 8192, sizes
    finally:
        eng.close(); ix.close()
