"""The oracle against the REFERENCE ITSELF, run here: oracle/_ref/ref_query is the reference's own indexing code
(sindex.c leaf calls over its LMDB and dtrie) and its own execute_squery (squery.c, docrank.c, filter_apply.c, sort.c,
hashtable.c, bmap.c ...), compiled from where the sources lie (oracle/ref_query.c, `make ref-query`).  Every shard result
of the C oracle (oracle/orc_squery.c) must equal what the reference left in struct squery_result for the same documents
and query: hit count, the returned rank records IN THE REFERENCE'S ORDER (its quickselect prefix included), every facet
list in its order, aggregations, the fast-rank flag and sample.  Skipped where the binary is absent."""
import random

import pytest

from marlin_amd import Index, QueryError
from oracle import orc_py, ref_py
from util import ROBOT_SUITES, load_fixture

pytestmark = pytest.mark.skipif(not ref_py.available(), reason="oracle/_ref/ref_query not built (needs /root/reference)")


def oracle_shards(ix, pq):
    outs = orc_py.execute_raw(ix, pq, 4)                     # 4: the reference's unsorted quickselect prefix
    q = pq.q.contents
    res = []
    for s in range(ix.num_shards):
        r = outs[s]
        d = {"num_hits": r.num_hits, "rank_count": r.rank_count, "fast": int(r.fast_rank),
             "ranks": [(x.docid, x.typos, x.proximity, x.field, x.position, x.exact, x.compare) for x in (r.ranks[i] for i in range(r.rank_count))],
             "facets": {f: [(r.fc[f][x].facet_id, r.fc[f][x].count) for x in range(r.rcount[f])]
                        for f in range(r.num_facets) if q.cfg.facet_enabled[f]},
             "aggs": [r.aggs[a] for a in range(r.num_aggs)]}
        res.append(d)
    return res, outs


def check(ix, pqs, tag):
    want = ref_py.run(ix, pqs)
    n = 0
    for pq, ref in zip(pqs, want):
        got, outs = oracle_shards(ix, pq)
        q = pq.q.contents
        ctx = (tag, pq.jq)
        for s, (g, r) in enumerate(zip(got, ref)):
            assert (g["num_hits"], g["rank_count"], g["fast"]) == (r["num_hits"], r["rank_count"], r["fast"]), (ctx, s)
            strip = (lambda t: t) if q.cfg.rank_by >= 0 else (lambda t: t[:6])      # compare is never written without rankBy
            assert [strip(t) for t in g["ranks"]] == [strip(t) for t in r["ranks"]], (ctx, s, "ranks")
            assert g["facets"] == r["facets"], (ctx, s, "facets")
            assert len(g["aggs"]) == len(r["aggs"]), (ctx, s)
            for a, b in zip(g["aggs"], r["aggs"]):
                assert a.type == b["type"], (ctx, s)
                if a.type in (1, 2):
                    assert a.value == b["v0"], (ctx, s, "agg value")
                elif a.type == 3:
                    assert (a.sum, a.count) == (b["v0"], b["count"]), (ctx, s, "avg")
                elif a.type == 4:
                    assert (a.sum, a.count, a.min, a.max) == (b["v0"], b["count"], b["min"], b["max"]), (ctx, s, "stats")
                elif a.type == 5:
                    assert a.card_n == b["card"], (ctx, s, "cardinality")
            n += 1
        for s in range(ix.num_shards):
            ix.lib.m_sqresult_free(outs[s])
    return n


@pytest.mark.parametrize("suite", ROBOT_SUITES)
def test_robot_corpora_on_the_reference(suite):
    fx = load_fixture(suite)
    ix = Index(fx["settings"], num_shards=fx["numShards"]).add_documents(fx["docs"])
    try:
        pqs = []
        for case in fx["cases"]:
            try:
                pqs.append(ix.query(case["query"]))
            except QueryError:
                pass
        assert check(ix, pqs, suite) == len(pqs) * ix.num_shards
    finally:
        ix.close()


def test_generated_queries_on_the_reference():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    rng = random.Random(77)
    ix = fuzz_parity.custom_index(rng, ndocs=2500, shards=2)
    try:
        pqs = []
        for jq in fuzz_parity.custom_queries(rng, 400):
            for full in (True, False):
                try:
                    pqs.append(ix.query(dict(jq, fullScan=full)))
                except (QueryError, KeyError, ValueError):
                    pass
        assert len(pqs) > 500
        check(ix, pqs, "custom")
    finally:
        ix.close()


def test_fast_rank_sample_on_the_reference():
    """perform_fast_ranking (docrank.c:478-560): more hits than fullScanThreshold, default mode.  Which documents the
    sample picks, in what order they come back, the exact facet recount: all from the reference's own run."""
    rng = random.Random(5)
    words = ["alpha", "beta", "gamma", "delta", "alphabet", "betamax"]
    docs = [{"id": i, "title": " ".join(rng.choice(words) for _ in range(rng.randrange(1, 5))), "num": (i * 7919) % 10007,
             "brand": rng.choice(["acme", "zenith", "orbit"]), "tags": [rng.choice(["t%d" % k for k in range(12)]) for _ in range(rng.randrange(0, 3))]}
            for i in range(9000)]
    ix = Index({"indexedFields": ["title", "num"], "facetFields": ["brand", "tags"], "hitsPerPage": 20}, num_shards=1).add_documents(docs)
    try:
        qs = [{"q": "alpha", "rankBy": {"num": "desc"}}, {"q": "alpha", "rankBy": {"num": "asc"}, "fullScanThreshold": 5000},
              {"q": "alphx beta", "rankBy": {"num": "desc"}, "fullScanThreshold": 5000},            # zero-typo set smaller than the sample
              {"q": "", "sortBy": {"num": "desc"}, "fullScanThreshold": 5000, "maxFacetResults": 5},
              {"q": "alph", "rankBy": {"num": "desc"}, "fullScanThreshold": 5000, "filter": {"brand": "acme"}},
              {"q": "beta", "fullScanThreshold": 5000},                                            # no rankBy: never fast
              {"q": "alpha beta", "rankBy": {"num": "desc"}, "fullScanThreshold": 5000, "aggs": {"s": {"stats": {"field": "num"}}}}]
        pqs = [ix.query(j) for j in qs]
        want = ref_py.run(ix, pqs)
        assert any(w[0]["fast"] for w in want), "no query reached fast-rank mode"
        check(ix, pqs, "fast-rank")
    finally:
        ix.close()
