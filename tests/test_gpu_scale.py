"""GPU path vs CPU oracle on a corpus large enough to leave the small-list shortcuts:
hit lists of ~100k documents per shard (multi-workgroup radix select, global-memory histograms,
LDS-overflowing facet fields), heavy key ties, and deep pages (k = 2000).  pytest -m gpu."""
import ctypes as C

import pytest

from marlin_amd import GpuEngine, Index, ffi
from oracle import orc_py
from util import compare_results

pytestmark = pytest.mark.gpu

DOCS, VOCAB, HC, SEED, NSH = 300_000, 4096, 50_000, 77, 2


def word(lib, r):
    buf = C.create_string_buffer(16)
    lib.msynth_zipf_word(VOCAB, SEED, r, buf)
    return buf.value.decode()


@pytest.fixture(scope="module")
def zipf():
    lib = ffi.load()
    shards = [lib.msynth_zipf(DOCS, VOCAB, HC, SEED, s, NSH) for s in range(NSH)]
    ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    eng = GpuEngine(ix)
    yield lib, ix, eng
    eng.close(); ix.close()


def queries(lib):
    w0, w1, w5, w9 = word(lib, 0), word(lib, 1), word(lib, 5), word(lib, 9)
    t5 = w5[:2] + ("x" if w5[2] != "x" else "y") + w5[3:]
    return [
        {"q": w0, "rankBy": {"num": "desc"}, "hitsPerPage": 10},
        {"q": w0},                                                     # every key equal up to position: ties
        {"q": "", "sortBy": {"price": "asc"}, "hitsPerPage": 100},
        {"q": w0 + " " + w1, "rankBy": {"price": "desc"}, "hitsPerPage": 1000, "maxHits": 1000},
        {"q": t5, "filter": {"cat": "cat03"}, "rankBy": {"num": "asc"},
         "aggs": {"s": {"stats": {"field": "price"}}, "c": {"cardinality": {"field": "hc"}}}},
        {"q": "", "filter": {"flag": True}, "rankBy": {"price": "desc"}, "maxFacetResults": 50},
        {"q": w9 + " " + w0[:3], "rankBy": {"num": "desc"}, "page": 5, "hitsPerPage": 20},
        {"q": w1, "filter": {"price": {"$gte": 1000, "$lt": 500000}}, "rankBy": {"num": "desc"}},
    ]


def test_large_hit_lists_match_oracle(zipf):
    lib, ix, eng = zipf
    pqs = [ix.query(dict(jq, fullScan=True)) for jq in queries(lib)]
    got = eng.search(pqs, capture=3)
    big = 0
    for pq, g in zip(pqs, got):
        o = orc_py.search(ix, pq, flags=3)
        compare_results(g, o, ctx=pq.jq)
        big = max(big, max(s["num_hits"] for s in g["_shards"]))
    assert big > 32768, "corpus too small to reach the multi-workgroup select"


def test_deep_pages_around_the_one_workgroup_selection_bound(zipf):
    """k = page x hitsPerPage on either side of MG_SEL_SOLO_K = 512 (the largest k one workgroup selects by itself;
    256 until round 3, when a third page of 100 hits went through three wide launches), over lists of under 8192 hits
    (k_select_first), up to 131072 (k_select_mid) and beyond (the wide launches), ranked by a number and by position (ties)"""
    lib, ix, eng = zipf
    w0, w1, w40, w300 = word(lib, 0), word(lib, 1), word(lib, 40), word(lib, 300)
    qs, sizes = [], set()
    for text in (w300, w40, w0 + " " + w1, ""):
        for hpp, page in ((257, 1), (100, 3), (128, 4), (171, 3), (500, 1), (100, 6)):
            qs.append({"q": text, "hitsPerPage": hpp, "page": page, "maxHits": 1000, "rankBy": {"num": "desc"}})
        qs.append({"q": text, "hitsPerPage": 100, "page": 5, "maxHits": 1000})
    pqs = [ix.query(dict(jq, fullScan=True)) for jq in qs]
    got = eng.search(pqs, capture=3)
    for pq, g in zip(pqs, got):
        compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
        sizes.update(s["num_hits"] for s in g["_shards"])
    assert min(sizes) < 8192 and any(8192 < n <= 131072 for n in sizes) and max(sizes) > 131072, sorted(sizes)


def test_batch_of_distinct_numeric_ranges_on_two_fields(zipf, monkeypatch):
    """wide numeric ranges become filter rows built from the column (k_range_rows), eight rows of one column per workgroup
    sharing the column's loads: a batch whose ranges all differ, on two fields and in counts that leave ragged groups,
    against the oracle and against the rows scattered from the sorted order (MGPU_NO_RANGE_ROWS)"""
    lib, ix, eng = zipf
    w0 = word(lib, 0)
    for nq in (1, 5, 11):
        jqs = []
        for i in range(nq):
            f = {"price": {"$gte": 1000 + 37 * i, "$lt": 400000 + 50021 * i}} if i % 3 != 2 else {"num": {"$gt": 50 + i, "$lte": 90000 + 17001 * i}}
            if i == 4:
                f = {"$and": [f, {"num": {"$gte": 20000}}]}                  # two rows of different columns in one squery
            jqs.append({"q": w0 if i % 2 else "", "filter": f, "rankBy": {"num": "desc"}, "fullScan": True})
        pqs = [ix.query(jq) for jq in jqs]
        got = eng.search(pqs, capture=3)
        monkeypatch.setenv("MGPU_NO_RANGE_ROWS", "1")
        plain = eng.search(pqs, capture=3)
        monkeypatch.delenv("MGPU_NO_RANGE_ROWS")
        for pq, g, p2 in zip(pqs, got, plain):
            assert g["ranks"] == p2["ranks"] and g["totalHits"] == p2["totalHits"] and g["facets"] == p2["facets"], pq.jq
            compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
        assert all(g["totalHits"] > 0 for g in got)


def test_result_independent_of_batching(zipf):
    lib, ix, eng = zipf
    pqs = [ix.query(dict(jq, fullScan=True)) for jq in queries(lib)]
    batched = eng.search(pqs)
    for pq, b in zip(pqs, batched):
        a = eng.search(pq)
        assert a["ranks"] == b["ranks"] and a["facets"] == b["facets"] and a.get("aggs") == b.get("aggs"), pq.jq


def test_sparse_postings_over_windows_and_term_groups(zipf):
    """rare words: their postings stay sorted lists that k_formula ORs into LDS tiles window by window
    (long lists through the window index, short ones scanned whole); four words make 8 terms = two
    tile groups; a one-letter prefix expands to a hundred-odd lists in one term"""
    lib, ix, eng = zipf
    r = [word(lib, k) for k in (300, 700, 1500, 2500, 3500, 4000)]
    qs = [
        {"q": " ".join(r[:4]), "hitsPerPage": 50},
        {"q": r[4] + " " + r[5] + " " + r[2], "rankBy": {"num": "desc"}},
        {"q": r[1] + " " + r[3][:1], "hitsPerPage": 30},                       # broad prefix on the last word
        {"q": r[5][:-1] + ("q" if r[5][-1] != "q" else "z") + " " + r[0], "maxFacetResults": 5},     # typo on a rare word
        {"q": r[2], "filter": {"cat": "cat01"}, "rankBy": {"price": "asc"}},
    ]
    pqs = [ix.query(dict(jq, fullScan=True)) for jq in qs]
    got = eng.search(pqs, capture=3)
    for pq, g in zip(pqs, got):
        compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
    assert any(g["totalHits"] > 0 for g in got)


def test_more_sparse_lists_than_one_chunk():
    """a four-letter prefix shared by 900 words: one term unions 900 sparse postings (600 long ones that go
    through the window index, 300 short ones scanned whole), more than the 256 lists a formula workgroup
    ranges in one go, on a shard of two windows"""
    docs = []
    for i in range(40_000):
        w1 = "abcd%03dx" % (i % 600)
        extra = ("abcd%03dy" % (600 + i % 300)) if i % 13 == 0 else "pad"
        docs.append({"id": i, "str": "%s %s w%d" % (w1, extra, i % 7), "num": i % 1000})
    ix = Index({"indexedFields": ["str", "num"], "facetFields": [], "hitsPerPage": 30}, num_shards=1).add_documents(docs)
    eng = GpuEngine(ix)
    try:
        qs = [{"q": "abcd"}, {"q": "w3 abcd", "rankBy": {"num": "desc"}}, {"q": "abcd1", "hitsPerPage": 50},
              {"q": "w5 abcd", "filter": {"num": {"$lt": 500}}}]
        pqs = [ix.query(dict(jq, fullScan=True)) for jq in qs]
        got = eng.search(pqs, capture=3)
        for pq, g in zip(pqs, got):
            compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
        assert got[0]["totalHits"] > 30_000 and got[3]["totalHits"] > 0
    finally:
        eng.close(); ix.close()


def test_pipeline_with_plans_built_in_the_loop(zipf):
    """the serving loop that builds its plans per step (msq_pipeline_run_resolved: process_termresult per batch inside the
    pipeline's threads) against the batch-at-a-time path: same summed totalHits over several passes, host and device
    dictionary lookups alike; plans built from a resolved set answer like plans prepared in one go"""
    lib, ix, eng4 = zipf
    eng = GpuEngine(ix, num_ctx=4)
    try:
        base = [ix.query(dict(jq, fullScan=True)) for jq in queries(lib)]
        batches = [base[:5], base[3:], base[1:7]]
        want = [sum(r["totalHits"] for r in eng.search(b)) for b in batches]
        for dev in (False, True):
            ress = [eng.resolve(b, device_lookup=dev) for b in batches]
            eng.warm_pipeline_resolved(ress, 4)
            hits, plan_ms = eng.run_pipeline_resolved(ress, 1, 7, 4)
            assert hits == sum(want[(1 + i) % 3] for i in range(7)), dev
            assert plan_ms > 0
            # a prepared batch planned from the resolved set, full comparison with the oracle
            prep = eng.plan_resolved(ress[0], capture=3)
            outs = eng.execute_prepared(prep)
            from marlin_amd.engine import assemble, shard_results_to_py
            ns = ix.num_shards
            for i, pq in enumerate(batches[0]):
                sub = (ffi.SqResult * ns).from_address(C.addressof(outs) + i * ns * C.sizeof(ffi.SqResult))
                g = assemble(ix, pq, sub)
                g["_shards"] = shard_results_to_py(ix, pq, sub)
                compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
            eng.free_raw(outs)
            eng.free_prepared(prep)
            for r in ress:
                eng.free_resolved(r)
    finally:
        eng.close()
