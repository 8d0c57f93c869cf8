"""diagnostic: distribution of hits per squery on the bench workload + stage times (not part of the product)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
from marlin_amd import ffi, GpuEngine, Index
lib = ffi.load()
shards = bench.build_shards(lib, 10_000_000, list(range(8)))
ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
eng = GpuEngine(ix, device=0)
lib.msq_set_plan_threads(16)
pool = [ix.query(j) for j in bench.make_queries(lib, 4096)]
first = int(os.environ.get("PROBE_FIRST", "0"))         # which 128 of the stream's 4096 queries (0, 128, ... 3968)
pool = pool[first:first + 128]
prep = eng.prepare(pool)
outs = eng.execute_prepared(prep)
hits = sorted(outs[i].num_hits for i in range(len(outs)))
n = len(hits)
print("squeries", n, "mean", sum(hits) / n, "p50", hits[n // 2], "p90", hits[n * 9 // 10], "p99", hits[n * 99 // 100], "max", hits[-1])
print(">512:", sum(h > 512 for h in hits), ">1536:", sum(h > 1536 for h in hits), ">2048:", sum(h > 2048 for h in hits), ">8192:", sum(h > 8192 for h in hits), ">32768:", sum(h > 32768 for h in hits))
eng.free_raw(outs)
eng.set_profiling(True)
for _ in range(3):
    eng.free_raw(eng.execute_prepared(prep))
print("stage ms:", {k: round(x["ms"], 4) for k, x in eng.timing()["stages"].items()})
eng.close(); ix.close()
