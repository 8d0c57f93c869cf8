"""Row (b) executed: the drop-in seam on the GPU behind the REFERENCE'S OWN host code.  oracle/_ref/ref_query_gpu is the
reference's indexing code, dtrie, LMDB stores and query structures (oracle/ref_query.c) linked with shim/squery_gpu.c in
place of the reference's squery.c; oracle/_ref/ref_query is the same program with the reference's squery.c.  Both index
the same documents with the reference's sindex.c; the shim then mirrors the shard's LMDB databases and dtrie into HBM
(mgpu_shim_attach_shard: cursor walks over wid2bmap / twid2bmap / twid2widbmap / facetid2bmap / boolid2bmap / docid2data
/ num_dbi, dict_walk) and answers execute_squery through libmarlin_gpu.so.  What lands in struct squery_result must be
what the reference's own execute_squery leaves there -- hit counts, rank keys, facet counts, aggregations, the fast-rank
flag -- with ties in the canonical order the GPU path defines (ties).  pytest -m gpu."""
import os
import random
import sys

import pytest

from marlin_amd import Index, QueryError, ffi
from oracle import ref_py
from util import ROBOT_SUITES, load_fixture

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (ref_py.available() and ref_py.available(gpu=True)), reason="oracle/_ref/ref_query(_gpu) not built")]

_IDX = {ffi.MR_TYPO: 1, ffi.MR_PROX: 2, ffi.MR_FIELD: 3, ffi.MR_POS: 4, ffi.MR_EXACT: 5, ffi.MR_COMP: 6, ffi.MR_COMP_ASC: 6}


def key_of(pq):
    q = pq.q.contents
    idx = []                                                  # (tuple index, sign): the comparator's order, sort.c:46-100
    for i in range(ffi.M_MAX_RULES):
        r = q.rank_rule[i]
        if r == ffi.MR_DONE:
            break
        if r in _IDX:
            idx.append((_IDX[r], -1 if r in (ffi.MR_EXACT, ffi.MR_COMP) else 1))
    return lambda t: tuple(sg * t[i] for i, sg in idx)


def compare(ix, pqs, tag):
    want = ref_py.run(ix, pqs)
    got = ref_py.run(ix, pqs, gpu=True)
    compare_lists(pqs, want, got, tag)


def compare_lists(pqs, want, got, tag):
    for pq, ws, gs in zip(pqs, want, got):
        key = key_of(pq)
        for s, (w, g) in enumerate(zip(ws, gs)):
            ctx = (tag, pq.jq, s)
            assert (g["num_hits"], g["rank_count"], g["fast"]) == (w["num_hits"], w["rank_count"], w["fast"]), ctx
            wk, gk = sorted(key(t) for t in w["ranks"]), sorted(key(t) for t in g["ranks"])
            assert wk == gk, ctx
            cut = max(wk) if wk and w["num_hits"] > w["rank_count"] else None
            wd = {key(t): t[0] for t in w["ranks"] if wk.count(key(t)) == 1 and key(t) != cut}
            gd = {key(t): t[0] for t in g["ranks"] if gk.count(key(t)) == 1 and key(t) != cut}
            assert wd == gd, ctx
            assert w["facets"].keys() == g["facets"].keys(), ctx
            for f in w["facets"]:
                wc, gc = sorted((-c, i) for i, c in w["facets"][f]), sorted((-c, i) for i, c in g["facets"][f])
                assert [c for c, _ in wc] == [c for c, _ in gc], (ctx, f)
                if wc:
                    low = wc[-1][0]
                    assert [e for e in wc if e[0] < low] == [e for e in gc if e[0] < low], (ctx, f)
            assert w["aggs"] == g["aggs"], ctx


@pytest.mark.parametrize("suite", ROBOT_SUITES)
def test_robot_corpora_through_the_seam(suite):
    fx = load_fixture(suite)
    ix = Index(fx["settings"], num_shards=fx["numShards"]).add_documents(fx["docs"])
    try:
        pqs = []
        for case in fx["cases"]:
            try:
                pqs.append(ix.query(case["query"]))
            except QueryError:
                pass
        compare(ix, pqs, suite)
    finally:
        ix.close()


def test_generated_queries_through_the_seam():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    rng = random.Random(78)
    ix = fuzz_parity.custom_index(rng, ndocs=3000, shards=3)
    try:
        pqs = []
        for jq in fuzz_parity.custom_queries(rng, 300):
            try:
                pqs.append(ix.query(dict(jq, fullScan=rng.random() < 0.7)))
            except (QueryError, KeyError, ValueError):
                pass
        compare(ix, pqs, "custom")
    finally:
        ix.close()


def test_mirror_refresh_after_a_second_bulk_write():
    """Row f-2 behind the reference's own writer: documents indexed AFTER the mirror was attached (a second si_write_start ..
    si_write_end on the same LMDB environments and dtrie) are invisible until mgpu_shim_refresh_shard sees the environment's
    newer transaction id, reads the shard again and hands the live mirror the difference as a delta
    (mgpu_shard_apply_delta; the call reports 0) -- or, with MGPU_SHIM_NO_DELTA, rebuilds the mirror next to the live one and
    swaps it in (2); either way the answers then equal the reference's on the grown shard"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    rng = random.Random(79)
    ix_ab = fuzz_parity.custom_index(rng, ndocs=1600, shards=2)
    all_docs = [d for d in sorted((d for sh in ix_ab.docs for d in sh), key=lambda d: d["id"])]
    ix_a = Index(dict(ix_ab.settings), num_shards=2).add_documents(all_docs[:900])
    ix_ab2 = Index(dict(ix_ab.settings), num_shards=2).add_documents(all_docs)
    try:
        jqs = [dict(j, fullScan=True) for j in fuzz_parity.custom_queries(rng, 60)]
        pa, pab = [], []
        for j in jqs:
            try:
                a, b = ix_a.query(j), ix_ab2.query(j)
            except (QueryError, KeyError, ValueError):
                continue
            pa.append(a); pab.append(b)
        w1, w2 = ref_py.run_two_writes(ix_a, ix_ab2, pa, pab)
        g1, g2 = ref_py.run_two_writes(ix_a, ix_ab2, pa, pab, gpu=True)
        compare_lists(pa, w1, g1, "before the second write")
        compare_lists(pab, w2, g2, "after refresh (delta)")
        g1r, g2r = ref_py.run_two_writes(ix_a, ix_ab2, pa, pab, gpu=True, no_delta=True)
        compare_lists(pab, w2, g2r, "after refresh (rebuild)")
        assert any(x[0]["num_hits"] != y[0]["num_hits"] for x, y in zip(w1, w2)), "the second write changed nothing"
    finally:
        ix_a.close(); ix_ab.close(); ix_ab2.close()
