"""Generated queries in the reference's DEFAULT mode (no fullScan: a shard with more hits than fullScanThreshold is
ranked on a sample, perform_fast_ranking docrank.c:478-560, and its kept facet values are recounted), GPU against the CPU
oracle, on the two-shard Zipf corpus (70k documents per shard, threshold 5000: most frequent-word queries are sampled).
rankBy / sortBy a unique number, so the sampled documents themselves must agree; the 64-value facet field is listed whole
(2N covers it), so its counts must agree; the 20000-value field's sample counts tie nearly everywhere, so its list is
checked for shape only (tests/test_gpu_fastrank.py checks its counts for exactness).  Where the reference itself
misbehaves in the mode the GPU refuses (MGPU_E_LIMIT / MGPU_E_UNSUPPORTED): counted, not compared.
python tools/fuzz_default_mode.py [rounds] [seed]"""
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from marlin_amd import GpuEngine, GpuError, Index, QueryError, ffi
from oracle import orc_py
from fuzz_parity import mutate


def queries(rng, lib, n, vocab, seed):
    buf = C.create_string_buffer(16)

    def zw(r):
        lib.msynth_zipf_word(vocab, seed, r, buf)
        return buf.value.decode()
    out = []
    for _ in range(n):
        nw = rng.choice([0, 1, 1, 1, 2, 2, 3])
        ws = [zw(rng.randrange(0, 12)) if rng.random() < 0.6 else mutate(rng, zw(int(rng.paretovariate(0.7)) % vocab)) for _ in range(nw)]
        if ws and rng.random() < 0.3:
            ws[-1] = ws[-1][:3]
        q = {"q": " ".join(ws), "hitsPerPage": rng.choice([5, 10, 30]), "maxFacetResults": rng.choice([0, 40, 40]),
             rng.choice(["rankBy", "rankBy", "sortBy"]): {"num": rng.choice(["asc", "desc"])}}
        if rng.random() < 0.15:
            q["filter"] = rng.choice([{"flag": False}, {"price": {"$lt": rng.randrange(600000, 990000)}}])
        if rng.random() < 0.3:
            q["aggs"] = {"s": {"stats": {"field": "price"}}, "a": {"avg": {"field": "num"}}}
        if rng.random() < 0.1:
            q["page"] = 2
        out.append(q)
    return out


def main(rounds=200, seed=1):
    rng = random.Random(seed)
    lib = ffi.load()
    V, S = 2048, 5
    shards = [lib.msynth_zipf(140_000, V, 20_000, S, s, 2) for s in range(2)]
    ix = Index.from_shards({"fullScanThreshold": 5000}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    eng = GpuEngine(ix)
    n = fast = refused = bad = 0
    try:
        for q in queries(rng, lib, rounds, V, S):
            try:
                pq = ix.query(q)
            except (QueryError, KeyError, ValueError):
                continue
            try:
                g = eng.search(pq, capture=1)
            except GpuError as e:
                if "-3" in str(e) or "-5" in str(e):
                    refused += 1
                    continue
                raise
            o = orc_py.search(ix, pq, flags=1)
            n += 1
            fast += not g["fullScan"]
            try:
                assert g["totalHits"] == o["totalHits"] and g["fullScan"] == o["fullScan"] and g["numHits"] == o["numHits"]
                assert [(r["shard"], r["docid"]) for r in g["ranks"]] == [(r["shard"], r["docid"]) for r in o["ranks"]], "ranks"
                assert [gs["num_hits"] for gs in g["_shards"]] == [os_["num_hits"] for os_ in o["_shards"]], "shard hits"
                assert repr(g.get("aggs")) == repr(o.get("aggs")), "aggs"
                if q["maxFacetResults"]:
                    # every shard lists the field whole (2N = 80 >= 64 values), so the merged sums are exact; the merged list is
                    # cut at maxFacetResults, and values of equal count at the cut are a tie class (ties)
                    gc, oc = ([(e["count"], e["facet_id"]) for e in r["facets"]["cat"]] for r in (g, o))
                    assert [c for c, _ in gc] == [c for c, _ in oc], "cat counts"
                    low = gc[-1][0] if gc else 0
                    assert sorted(t for t in gc if t[0] > low) == sorted(t for t in oc if t[0] > low), "cat values above the cut"
                    hc = g["facets"]["hc"]
                    assert len(hc) == len(o["facets"]["hc"]) and all(e["count"] > 0 for e in hc), "hc shape"
                    assert [e["count"] for e in hc] == sorted((e["count"] for e in hc), reverse=True), "hc order"
            except AssertionError as e:
                bad += 1
                print("MISMATCH", q, str(e)[:200], "hits", g["totalHits"], o["totalHits"], "fullScan", g["fullScan"], o["fullScan"])
    finally:
        eng.close(); ix.close()
    print("default-mode fuzz: %d queries compared (%d ranked on a sample), %d refused where the reference misbehaves, %d mismatches" % (n, fast, refused, bad))
    return bad, n, fast


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:3]]
    sys.exit(1 if main(*a)[0] else 0)
