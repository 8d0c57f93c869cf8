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
import util
from util import RANK_FIELDS, compare_results, datagen_index


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


VOC = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "theta", "lambda", "sigma", "omega", "alphabet", "betamax",
       "gammaray", "deltoid", "zebra", "thermal", "lambent", "signal", "omen", "al", "be", "ga"]


def custom_index(rng, ndocs=6000, shards=4):
    """documents with repeated words (long position lists -> overflow kernel), two text fields, multi-valued facets,
    a float and an integer field, over four shards"""
    docs = []
    for i in range(ndocs):
        n1 = rng.choice([1, 2, 3, 5, 8, 20, 45]) if rng.random() < 0.15 else rng.randrange(1, 7)
        t1 = " ".join(rng.choice(VOC[:12] if rng.random() < 0.7 else VOC) for _ in range(n1))
        t2 = " ".join(rng.choice(VOC) for _ in range(rng.randrange(0, 4)))
        d = {"id": i, "title": t1, "body": t2, "price": round(rng.random() * 500, 2) if rng.random() < 0.9 else float(rng.randrange(100)),
             "stock": rng.randrange(0, 50), "brand": rng.choice(["acme", "zenith", "orbit", "nadir", "apex"]),
             "tags": [rng.choice(["t%02d" % k for k in range(30)]) for _ in range(rng.randrange(0, 4))], "instock": rng.random() < 0.6}
        docs.append(d)
    settings = {"indexedFields": ["title", "body", "price", "stock", "instock"], "facetFields": ["brand", "tags"], "hitsPerPage": 20}
    return Index(settings, num_shards=shards).add_documents(docs)


def custom_queries(rng, n):
    rules = ["typos", "proximity", "position", "exact", "field"]
    out = []
    for _ in range(n):
        q = {"q": " ".join(mutate(rng, rng.choice(VOC)) for _ in range(rng.randrange(0, 5)))}
        if rng.random() < 0.3:
            q["rankAlgorithm"] = rng.sample(rules, rng.randrange(1, 6))
        if rng.random() < 0.15:
            q["typos"] = False
        if rng.random() < 0.3:
            q["prefix"] = rng.choice(["last", "all", "none"])
        r = rng.random()
        if r < 0.35:
            q["rankBy"] = {rng.choice(["price", "stock"]): rng.choice(["asc", "desc"])}
        elif r < 0.5:
            q["sortBy"] = {rng.choice(["price", "stock"]): rng.choice(["asc", "desc"])}
        if rng.random() < 0.4:
            f = rng.choice([{"brand": rng.choice(["acme", "orbit", "nope"])}, {"instock": True}, {"stock": {"$gte": rng.randrange(0, 50)}},
                            {"price": {"$gt": rng.random() * 400, "$lte": 450.0}}, {"tags": {"$in": ["t01", "t17", "t29"]}},
                            {"brand": {"$ne": "acme"}}, {"stock": rng.randrange(0, 50)}])
            if rng.random() < 0.3:
                f = {"$and": [f, {"$or": [{"instock": False}, {"tags": "t05"}]}]}
            q["filter"] = f
        if rng.random() < 0.3:
            q["aggs"] = {"p": {rng.choice(["stats", "avg", "max", "min"]): {"field": "stock"}}}
            if rng.random() < 0.5:
                q["aggs"]["c"] = {"cardinality": {"field": rng.choice(["brand", "tags"])}}
        q["hitsPerPage"] = rng.choice([0, 5, 20, 100, 500])
        q["maxHits"] = max(q["hitsPerPage"], rng.choice([0, 100, 1000])) if q["hitsPerPage"] else 0
        if q["hitsPerPage"] and rng.random() < 0.2:
            q["page"] = rng.randrange(1, 3)
        q["maxFacetResults"] = rng.choice([0, 2, 10, 40])
        out.append(q)
    return out


def run(ix, eng, qs, tag):
    pqs, kept = [], []
    for q in qs:
        try:
            pqs.append(ix.query(dict(q, fullScan=True))); kept.append(q)
        except (QueryError, KeyError, ValueError):
            pass
    bad = 0
    for b in range(0, len(pqs), 16):
        # every other batch resolves its typo walks on the device (row f-1): same answers expected
        got = eng.search(pqs[b:b + 16], capture=3, device_lookup=bool((b // 16) & 1) and os.environ.get("FUZZ_HOST_LOOKUP") is None)
        for pq, g, q in zip(pqs[b:b + 16], got, kept[b:b + 16]):
            o = orc_py.search(ix, pq, flags=3)
            fields = RANK_FIELDS
            if "rankAlgorithm" in q:       # only the named rules (and the rankBy / sortBy value) order the hits
                fields = tuple(q["rankAlgorithm"]) + (("compare",) if ("rankBy" in q or "sortBy" in q) else ())
            try:
                compare_results(g, o, ctx=q, key_fields=fields)
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
    shards = [lib.msynth_zipf(140_000, V, 20_000, S, s, 2) for s in range(2)]      # 70k docs per shard: three formula windows
    ix2 = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    e2 = GpuEngine(ix2)
    n2, b2 = run(ix2, e2, zipf_queries(rng, lib, rounds, V, S), "zipf")
    e2.close(); ix2.close()
    ix3 = custom_index(rng)
    e3 = GpuEngine(ix3)
    n3, b3 = run(ix3, e3, custom_queries(rng, rounds), "custom")
    e3.close(); ix3.close()
    print("fuzz: %d + %d + %d queries, %d + %d + %d mismatches" % (n1, n2, n3, b1, b2, b3))
    print("merged facet lists: %d compared, %d left out because a shard's tied tail differed (DESIGN.md, ties)"
          % (util.COMPARED_MERGED_FACETS[0], util.SKIPPED_MERGED_FACETS[0]))
    return b1 + b2 + b3


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:3]]
    sys.exit(1 if main(*a) else 0)
