"""Host cost of plan building (msq_plan_resolved: process_termresult for every (query, shard) of a batch), no GPU needed:
python tools/plan_probe.py [docs] [queries]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from marlin_amd import Index, ffi
lib = ffi.load()
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 512
shards = bench.build_shards(lib, docs, list(range(8)))
ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
mirrors = [lib.mmirror_scheme(s) for s in ix.shards]
marr = (C.c_void_p * 8)(*mirrors)
pool = [ix.query(j) for j in bench.make_queries(lib, nq)]
lib.msq_set_plan_threads(8)
B = 128
res = []
t0 = time.time()
for b in range(nq // B):
    qarr = (C.POINTER(ffi.Query) * B)(*[p.q for p in pool[b * B:(b + 1) * B]])
    h = C.c_void_p()
    rc = lib.msq_resolve_batch(None, marr, 8, qarr, B, C.byref(h))
    assert rc == 0, rc
    res.append((h, qarr))
print("resolve (host trie, 8 threads): %.1f ms per batch" % ((time.time() - t0) * 1e3 / len(res)))
pp = lib.msq_prepared_new()
for rep in range(3):
    t0 = time.perf_counter()
    for h, _ in res:
        rc = lib.msq_plan_resolved(h, 0, pp)
        assert rc == 0, rc
    dt = (time.perf_counter() - t0) * 1e3 / len(res)
    print("msq_plan_resolved: %.3f ms per 128-query batch (1024 plans), one thread" % dt)
