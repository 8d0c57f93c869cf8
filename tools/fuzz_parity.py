"""randomised differential run: GPU path vs CPU oracle on generated queries (diagnostic; the seeded subset
in tests/test_gpu_fuzz.py is part of the suite).  usage: fuzz_parity.py [rounds] [seed]"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from marlin_amd import GpuEngine, Index, QueryError, ffi
from oracle import orc_py
from util import compare_results, datagen_index


def mutate(rng, w):
    """one edit, or a prefix, or the word itself"""
    k = rng.randrange(6)
    if k == 0 and len(w) > 3:
        p = rng.randrange(1, len(w)); return w[:p] + rng.choice("abcdefghij") + w[p + 1:]
    if k == 1 and len(w) > 3:
        p = rng.randrange(1, len(w)); return w[:p] + w[p + 1:]
    if k == 2:
        p = rng.randrange(1, len(w) + 1); return w[:p] + rng.choice("abcdxyz") + w[p:]
    if k == 3 and len(w) > 2:
        return w[:rng.randrange(1, len(w))]
    return w


def datagen_queries(rng, n):
    words = ["aaaaaaa", "bbbbbbb", "ccccccc", "ddddddd", "eeeeeee", "eeeeeed", "attr"]
    nums = ["id", "num"]
    out = []
    for _ in range(n):
        q = {"q": " ".join(mutate(rng, rng.choice(words)) for _ in range(rng.randrange(0, 5)))}
        if rng.random() < 0.4:
            q["rankBy"] = {rng.choice(nums): rng.choice(["asc", "desc"])}
        elif rng.random() < 0.2:
            q["sortBy"] = {rng.choice(nums): rng.choice(["asc", "desc"])}
        if rng.random() < 0.35:
            f = rng.choice([{"bool": rng.random() < 0.5}, {"num": {"$gt": rng.randrange(0, 9000)}},
                            {"num": {"$gte": rng.randrange(0, 5000), "$lt": rng.randrange(5000, 10001)}},
                            {"id": {"$lte": rng.randrange(1, 9000)}}, {"bool": {"$ne": True}}])
            if rng.random() < 0.3:
                f = {"$or": [f, {"id": rng.randrange(1, 9000)}]}
            q["filter"] = f
        if rng.random() < 0.3:
            q["aggs"] = {"a": {rng.choice(["max", "min", "avg", "stats"]): {"field": rng.choice(nums)}}}
            if rng.random() < 0.4:
                q["aggs"]["c"] = {"cardinality": {"field": rng.choice(["facet", "facetlist", "facetonly"])}}
        q["hitsPerPage"] = rng.choice([0, 1, 10, 25, 100])
        if q["hitsPerPage"] == 0:
            q["maxHits"] = 0
        if rng.random() < 0.2 and q["hitsPerPage"]:
            q["page"] = rng.randrange(1, 4)
        q["maxFacetResults"] = rng.choice([0, 1, 5, 10, 50])
        out.append(q)
    return out


def zipf_queries(rng, lib, n, vocab, seed):
    buf = C.create_string_buffer(16)
    def zw(r):
        lib.msynth_zipf_word(vocab, seed, r, buf)
        return buf.value.decode()
    out = []
    for _ in range(n):
        nw = rng.randrange(0, 4)
        ws = [mutate(rng, zw(int(rng.paretovariate(0.7)) % vocab)) for _ in range(nw)]
        q = {"q": " ".join(ws), "hitsPerPage": rng.choice([0, 10, 30]), "maxFacetResults": rng.choice([0, 3, 10, 40])}
        if q["hitsPerPage"] == 0:
            q["maxHits"] = 0
        r = rng.random()
        if r < 0.4:
            q["rankBy"] = {rng.choice(["num", "price"]): rng.choice(["asc", "desc"])}
        elif r < 0.55:
            q["sortBy"] = {rng.choice(["num", "price"]): rng.choice(["asc", "desc"])}
        if rng.random() < 0.3:
            q["filter"] = rng.choice([{"flag": rng.random() < 0.5}, {"price": {"$lt": rng.randrange(1000, 900000)}},
                                      {"cat": "cat%02d" % rng.randrange(0, 64)}])
        if rng.random() < 0.25:
            q["aggs"] = {"s": {"stats": {"field": "price"}}}
            if rng.random() < 0.5:
                q["aggs"]["c"] = {"cardinality": {"field": rng.choice(["cat", "hc"])}}
        out.append(q)
    return out


def run(ix, eng, qs, tag):
    pqs, kept = [], []
    for q in qs:
        try:
            pqs.append(ix.query(dict(q, fullScan=True))); kept.append(q)
        except QueryError:
            pass
    bad = 0
    for b in range(0, len(pqs), 16):
        got = eng.search(pqs[b:b + 16], capture=3)
        for pq, g, q in zip(pqs[b:b + 16], got, kept[b:b + 16]):
            o = orc_py.search(ix, pq, flags=3)
            try:
                compare_results(g, o, ctx=q)
            except AssertionError as e:
                bad += 1
                print("MISMATCH", tag, q, str(e)[:200])
                if g.get("aggs") != o.get("aggs"):
                    print("   aggs gpu", g.get("aggs"), "\n   aggs orc", o.get("aggs"))
                for f in o["facets"]:
                    if g["facets"].get(f) != o["facets"][f]:
                        print("   facet", f, "gpu", g["facets"].get(f)[:12], "\n             orc", o["facets"][f][:12])
                print("   totalHits", g["totalHits"], o["totalHits"], "shard hits", [s["num_hits"] for s in g["_shards"]])
    return len(pqs), bad


def main(rounds=200, seed=1):
    rng = random.Random(seed)
    lib = ffi.load()
    ix1, _ = datagen_index(10000)
    e1 = GpuEngine(ix1)
    n1, b1 = run(ix1, e1, datagen_queries(rng, rounds), "datagen")
    e1.close(); ix1.close()
    V, S = 2048, 5
    shards = [lib.msynth_zipf(40_000, V, 20_000, S, s, 2) for s in range(2)]
    ix2 = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    e2 = GpuEngine(ix2)
    n2, b2 = run(ix2, e2, zipf_queries(rng, lib, rounds, V, S), "zipf")
    e2.close(); ix2.close()
    print("fuzz: %d + %d queries, %d + %d mismatches" % (n1, n2, b1, b2))
    return b1 + b2


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:3]]
    sys.exit(1 if main(*a) else 0)
