"""Row f-2 priced on the bench corpus: python tools/delta_probe.py [docs] [added]
mirror(docs) built, then the shards of docs + added documents applied as a delta, against building that mirror anew."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from marlin_amd import GpuEngine, Index, ffi
lib = ffi.load()
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
added = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
import threading
def index(first):
    """the bench corpus of docs + added documents, cut off after its first `first`"""
    sh = [None] * 8
    def work(s):
        sh[s] = lib.msynth_zipf_first(docs + added, bench.VOCAB, bench.HC_BUCKETS, bench.SEED, s, 8, first)
    th = [threading.Thread(target=work, args=(s,)) for s in range(8)]
    [t.start() for t in th]; [t.join() for t in th]
    return Index.from_shards({}, sh, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
ix_a, ix_b = index(docs), index(docs + added)
t0 = time.time(); eng = GpuEngine(ix_a); t_build = time.time() - t0
pqs = [ix_b.query(j) for j in bench.make_queries(lib, 64)]
t0 = time.time(); how = eng.apply_index(ix_b); t_delta = time.time() - t0
got = eng.search(pqs)
t0 = time.time(); fresh = GpuEngine(ix_b); t_fresh = time.time() - t0
want = fresh.search(pqs)
def fkey(res):
    # facet values of equal count at the cut are a tie class (the delta mirror numbers late values after the old ones)
    return {f: [e["count"] for e in v] for f, v in res["facets"].items()}
same = all(g["totalHits"] == w["totalHits"] and fkey(g) == fkey(w) and
           [(r["shard"], r["docid"]) for r in g["ranks"]] == [(r["shard"], r["docid"]) for r in w["ranks"]] for g, w in zip(got, want))
for i, (g, w) in enumerate(zip(got, want)):
    if g["totalHits"] != w["totalHits"]: print("q", i, "totalHits", g["totalHits"], w["totalHits"], [x["num_hits"] for x in g.get("_shards", [])])
    elif fkey(g) != fkey(w):
        for f in g["facets"]:
            if g["facets"][f] != w["facets"][f]: print("q", i, "facet", f, g["facets"][f][:6], w["facets"][f][:6])
    elif [(r["shard"], r["docid"]) for r in g["ranks"]] != [(r["shard"], r["docid"]) for r in w["ranks"]]: print("q", i, "ranks", g["ranks"][:3], w["ranks"][:3])
print({"docs": docs, "added": added, "how": how, "mirror_build_s": round(t_build, 2), "delta_s": round(t_delta, 3), "rebuild_s": round(t_fresh, 2),
       "answers_equal": same, "hits": sum(g["totalHits"] for g in got)})
assert same
