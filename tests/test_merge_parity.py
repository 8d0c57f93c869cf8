"""The product's merge (m_form_result, marlin_amd/host/mquery.c) against the oracle's literal restatement of the
reference's (oracle/orc_merge.c: query.c:312-383, 39-92, metric-aggs.c merges), on the SAME per-shard results, so that
the merged comparisons of the GPU tests do not check the merge against itself.  CPU only."""
import ctypes as C
import random

from marlin_amd import Index, ffi
from marlin_amd.engine import assemble
from oracle import orc_py
from util import RANK_FIELDS, facets_canonical, load_fixture, rank_key, _same


def both_merges(ix, pq):
    outs = orc_py.execute_raw(ix, pq)
    a = assemble(ix, pq, outs)                                           # product
    b = assemble(ix, pq, outs, form=orc_py.load().orc_form_result)       # reference restated
    for s in range(ix.num_shards):
        ix.lib.m_sqresult_free(C.byref(outs[s]))
    return a, b


def check(ix, jq):
    pq = ix.query(jq)
    a, b = both_merges(ix, pq)
    assert (a["totalHits"], a["numHits"], a["page"], a["fullScan"]) == (b["totalHits"], b["numHits"], b["page"], b["fullScan"]), jq
    ka, kb = [rank_key(r) for r in a["ranks"]], [rank_key(r) for r in b["ranks"]]
    assert ka == kb, jq                                                   # same key sequence
    for i, (x, y) in enumerate(zip(a["ranks"], b["ranks"])):
        if ka.count(ka[i]) == 1:
            assert (x["shard"], x["docid"]) == (y["shard"], y["docid"]), (jq, i)
    fa, fb = facets_canonical(a), facets_canonical(b)
    assert fa.keys() == fb.keys()
    for k in fa:
        assert [c for c, _ in fa[k]] == [c for c, _ in fb[k]], (jq, k)   # same sums, same cut
        if fa[k]:
            cut = fa[k][-1][0]
            assert [e for e in fa[k] if e[0] > cut] == [e for e in fb[k] if e[0] > cut], (jq, k)
    assert _same(a.get("aggs"), b.get("aggs")), jq


def test_merges_agree_on_the_multi_shard_robot_suites():
    for suite in ("facet", "aggs", "rank", "sort"):
        fx = load_fixture(suite)
        ix = Index(fx["settings"], num_shards=fx["numShards"]).add_documents(fx["docs"])
        try:
            for case in fx["cases"]:
                if case["status"] != 400:
                    check(ix, dict(case["query"], fullScan=True))
        finally:
            ix.close()


def test_merges_agree_on_a_generated_corpus():
    rng = random.Random(11)
    words = ["w%02d" % i for i in range(40)]
    docs = [{"t": " ".join(rng.choice(words) for _ in range(rng.randint(2, 7))), "n": rng.randint(0, 500), "m": rng.randint(0, 9),
             "c": "c%d" % rng.randint(0, 30), "tags": ["g%d" % rng.randint(0, 60) for _ in range(rng.randint(0, 3))]} for _ in range(3000)]
    for shards in (1, 3, 7):
        ix = Index({"indexedFields": ["t", "n", "m"], "facetFields": ["c", "tags"], "hitsPerPage": 20}, num_shards=shards).add_documents(docs)
        try:
            for _ in range(40):
                q = {"q": " ".join(rng.sample(words, rng.randint(0, 2))), "fullScan": True, "maxFacetResults": rng.choice([1, 3, 10, 50]),
                     "page": rng.randint(1, 3), "hitsPerPage": rng.choice([5, 20, 100])}
                if rng.random() < 0.5:
                    q["rankBy"] = {rng.choice(["n", "m"]): rng.choice(["asc", "desc"])}
                if rng.random() < 0.5:
                    q["aggs"] = {"a": {"avg": {"field": "n"}}, "s": {"stats": {"field": "m"}}, "c": {"cardinality": {"field": "tags"}},
                                 "x": {"max": {"field": "n"}}, "i": {"min": {"field": "n"}}}
                check(ix, q)
        finally:
            ix.close()
