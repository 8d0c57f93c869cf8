"""diagnostic: time of k_rank_tile on the bench workload with parts of it left out (results are wrong then; not part of
the product).  bit 2: phase 2 (positions + rank), 4: phase 1 probing.  Only these two: every other part feeds arrays
that later kernels index with (leaving the facet pass out faulted the GPU)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
from marlin_amd import ffi
lib = ffi.load()
gpu = C.CDLL(os.path.join(ffi.LIB_DIR, "libmarlin_gpu.so"))
import torch
torch.cuda.set_device(0)
from marlin_amd import GpuEngine, Index
shards = bench.build_shards(lib, 10_000_000, list(range(8)))
ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
eng = GpuEngine(ix, device=0)
lib.msq_set_plan_threads(16)
pool = [ix.query(j) for j in bench.make_queries(lib, 256)]
preps = [eng.prepare(pool[:128]), eng.prepare(pool[128:])]
for p in preps:
    eng.free_raw(eng.execute_prepared(p))
eng.set_profiling(True)
masks = [int(x, 0) for x in os.environ.get("RANK_SKIP", "0,2,4,6,0").split(",")]
for m in masks:
    if m & ~6:
        raise SystemExit("only bits 2 and 4 are safe")
    gpu.mgk_debug_skip(m)
    acc = 0.0
    for i in range(4):
        eng.free_raw(eng.execute_prepared(preps[i & 1]))
        acc += eng.timing()["stages"]["rank"]["ms"]
    print("skip mask %2d: rank stage %.4f ms" % (m, acc / 4), flush=True)
gpu.mgk_debug_skip(0)
for p in preps:
    eng.free_prepared(p)
eng.close(); ix.close()
