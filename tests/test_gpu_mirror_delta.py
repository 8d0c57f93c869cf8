"""f-2: mirror maintenance.  A committed bulk write (si_write_end, sindex.c:184-254) appends documents; the mirror takes
the difference as a delta (mgpu_shard_apply_delta: k_merge_postings + k_build_winidx on the device, dense rows' new bits by
k_scatter, the new documents appended into the mirror's headroom) instead of being rebuilt.  mirror(A) + delta(B) must
answer like the oracle over A u B -- and like a mirror built from A u B -- for generated queries (typos, prefixes,
filters on facets / numbers / bools, sortBy / rankBy, facets with values the old mirror never saw, aggregations incl.
cardinality, term lookups on the device over the re-exported dictionary), after one delta and after a second one, in
fast-rank mode (facet recount through the id -> posting table), and when the headroom is used up.  pytest -m gpu."""
import os
import random
import sys

import pytest

from marlin_amd import GpuEngine, Index, QueryError
from oracle import orc_py
from util import ROOT, compare_results

sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as fz                                             # noqa: E402

pytestmark = pytest.mark.gpu

SETTINGS = {"indexedFields": ["title", "body", "price", "stock", "instock"], "facetFields": ["brand", "tags"], "hitsPerPage": 20}


def make_docs(rng, n):
    """the fuzzer's third corpus, with drift: later documents bring words, brands and tags the earlier ones never had"""
    docs = []
    for i in range(n):
        late = i * 3 // n                                            # 0, 1, 2: which third of the corpus
        voc = fz.VOC + ["zulu%d" % k for k in range(late * 6)] + (["quasar", "quasars"] if late == 2 else [])
        n1 = rng.choice([1, 2, 3, 5, 8, 20, 45]) if rng.random() < 0.15 else rng.randrange(1, 7)
        t1 = " ".join(rng.choice(voc[:12] if rng.random() < 0.6 else voc) for _ in range(n1))
        t2 = " ".join(rng.choice(voc) for _ in range(rng.randrange(0, 4)))
        docs.append({"id": i, "title": t1, "body": t2, "price": round(rng.random() * 500, 2) if rng.random() < 0.9 else float(rng.randrange(100)),
                     "stock": rng.randrange(0, 50), "brand": rng.choice(["acme", "zenith", "orbit", "nadir", "apex"] + ["late%d" % late] * late),
                     "tags": [rng.choice(["t%02d" % k for k in range(30 + 15 * late)]) for _ in range(rng.randrange(0, 4))],
                     "instock": rng.random() < 0.6})
    return docs


def queries(rng, n):
    qs = fz.custom_queries(rng, n)
    for q in qs[::7]:                                                # words and facet values only the late documents have
        q["q"] = (q["q"] + " " + rng.choice(["zulu3", "zulu7", "quasar", "quasa", "zul"])).strip()
    for q in qs[3::11]:
        q["filter"] = rng.choice([{"brand": "late1"}, {"brand": {"$ne": "late2"}}, {"tags": {"$in": ["t33", "t47", "t58"]}}])
    return qs


def test_delta_answers_like_a_rebuilt_mirror():
    rng = random.Random(77)
    docs = make_docs(rng, 7000)
    cuts = [5000, 6200, 7000]
    ix = Index(SETTINGS, num_shards=4).add_documents(docs[:cuts[0]])
    eng = GpuEngine(ix)
    old = [ix]
    try:
        n, bad = fz.run(ix, eng, queries(rng, 60), "base")
        assert n > 30 and bad == 0
        for cut in cuts[1:]:
            ix2 = Index(SETTINGS, num_shards=4).add_documents(docs[:cut])
            cap = [eng.lib.mgpu_shard_doc_capacity(eng.lib.mmirror_gpu(m)) for m in eng.mirrors]
            how = eng.apply_index(ix2)
            assert how == ["delta"] * 4, how
            assert [eng.lib.mgpu_shard_max_docid(eng.lib.mmirror_gpu(m)) for m in eng.mirrors] == [ix2.max_docid(s) for s in range(4)]
            assert [eng.lib.mgpu_shard_doc_capacity(eng.lib.mmirror_gpu(m)) for m in eng.mirrors] == cap      # grown in place
            old.append(ix2)
            qs = queries(rng, 220)
            n, bad = fz.run(ix2, eng, qs, "delta@%d" % cut)
            assert n > 150 and bad == 0
            # and like a mirror built from the whole corpus: identical answers, ids included
            fresh = GpuEngine(ix2)
            try:
                pqs = []
                for q in qs[:80]:
                    if "rankAlgorithm" in q:
                        continue
                    try:
                        pqs.append(ix2.query(dict(q, fullScan=True)))
                    except (QueryError, KeyError, ValueError):
                        pass
                assert len(pqs) > 30
                a, b = eng.search(pqs, capture=0), fresh.search(pqs, capture=0)
                for x, y, pq in zip(a, b, pqs):
                    # facet values of equal count at the cut are a tie class: the delta mirror numbers the values that came
                    # with a delta after the old ones, a fresh one numbers all of them by id
                    fx, fy = ({f: [e["count"] for e in v] for f, v in r["facets"].items()} for r in (x, y))
                    assert x["totalHits"] == y["totalHits"] and fx == fy and repr(x.get("aggs")) == repr(y.get("aggs")), pq.jq   # (repr: an empty avg is nan)
                    assert [(r["shard"], r["docid"]) for r in x["ranks"]] == [(r["shard"], r["docid"]) for r in y["ranks"]], pq.jq
            finally:
                fresh.close()
            ix = ix2
    finally:
        eng.close()
        for i in old:
            i.close()


def test_delta_in_fast_rank_mode():
    """default mode above fullScanThreshold: the sample walks the re-sent num_dbi order, the kept facet ids are recounted
    as |posting AND docid_map| through the id -> posting table, which now holds ids in the middle of the old ones"""
    rng = random.Random(5)
    docs = make_docs(rng, 24000)
    for d in docs:
        d["price"] = float(d["id"] * 7919 % 100003)                  # unique: no ties among the sampled documents
    settings = dict(SETTINGS, fullScanThreshold=5000)
    ix = Index(settings, num_shards=2).add_documents(docs[:17000])
    ix2 = Index(settings, num_shards=2).add_documents(docs)
    eng = GpuEngine(ix)
    try:
        assert eng.apply_index(ix2) == ["delta", "delta"]
        for q, fast in (({"q": "", "rankBy": {"price": "desc"}, "maxFacetResults": 100}, True),
                        ({"q": "", "rankBy": {"price": "asc"}, "maxFacetResults": 100, "aggs": {"s": {"stats": {"field": "stock"}}}}, True),
                        ({"q": "", "sortBy": {"price": "desc"}, "maxFacetResults": 100, "hitsPerPage": 50}, True),
                        ({"q": "quasar", "rankBy": {"price": "desc"}, "maxFacetResults": 100}, False)):
            pq = ix2.query(q)
            g, o = eng.search(pq, capture=1), orc_py.search(ix2, pq, flags=1)
            assert o["fullScan"] == (not fast) and g["fullScan"] == o["fullScan"], q
            compare_results(g, o, ctx=q)
            assert [(r["shard"], r["docid"]) for r in g["ranks"]] == [(r["shard"], r["docid"]) for r in o["ranks"]], q
    finally:
        eng.close(); ix.close(); ix2.close()


def test_no_headroom_falls_back_to_a_rebuild(monkeypatch):
    rng = random.Random(9)
    docs = make_docs(rng, 3000)
    monkeypatch.setenv("MGPU_MIRROR_HEADROOM", "0")
    ix = Index(SETTINGS, num_shards=2).add_documents(docs[:2000])
    ix2 = Index(SETTINGS, num_shards=2).add_documents(docs)
    eng = GpuEngine(ix)
    try:
        # straight through the C ABI: refused with MGPU_E_LIMIT, and the mirror still answers for the old documents
        assert eng.lib.mmirror_apply_shard(eng.mirrors[0], ix2.shards[0]) == -3
        n, bad = fz.run(ix, eng, queries(rng, 40), "refused")
        assert n > 20 and bad == 0
        assert eng.apply_index(ix2) == ["rebuilt", "rebuilt"]
        n, bad = fz.run(ix2, eng, queries(rng, 80), "rebuilt")
        assert n > 50 and bad == 0
    finally:
        eng.close(); ix.close(); ix2.close()


def test_delta_edge_cases():
    """nothing new (the same shards again), a single new document, and a shard that was EMPTY before the write"""
    rng = random.Random(13)
    docs = make_docs(rng, 1200)
    ix0 = Index(SETTINGS, num_shards=3).add_documents(docs[:2])          # shard 2 holds no document yet
    eng = GpuEngine(ix0)
    keep = [ix0]
    try:
        for cut in (2, 3, 1200):
            ix = Index(SETTINGS, num_shards=3).add_documents(docs[:cut])
            keep.append(ix)
            assert eng.apply_index(ix) == ["delta"] * 3, cut
            assert [eng.lib.mgpu_shard_max_docid(eng.lib.mmirror_gpu(m)) for m in eng.mirrors] == [ix.max_docid(s) for s in range(3)]
            n, bad = fz.run(ix, eng, queries(rng, 50), "edge@%d" % cut)
            assert bad == 0 and n > 25
        same = Index(SETTINGS, num_shards=3).add_documents(docs)
        keep.append(same)
        assert eng.apply_index(same) == ["delta"] * 3                     # an empty delta
        n, bad = fz.run(same, eng, queries(rng, 50), "again")
        assert bad == 0 and n > 25
        # a write that is not an append (a document gone) is refused by the delta path and rebuilt
        fewer = Index(SETTINGS, num_shards=3).add_documents(docs[:600])
        keep.append(fewer)
        assert eng.lib.mmirror_apply_shard(eng.mirrors[0], fewer.shards[0]) == -5
    finally:
        eng.close()
        for i in keep:
            i.close()
