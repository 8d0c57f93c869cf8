"""diagnostic: where the host time of one bench step goes"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
from marlin_amd import ffi, GpuEngine, Index
lib = ffi.load()
import torch
torch.cuda.set_device(0)
shards = bench.build_shards(lib, 10_000_000, list(range(8)))
ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
eng = GpuEngine(ix, device=0)
lib.msq_set_plan_threads(16)
pool = [ix.query(j) for j in bench.make_queries(lib, 256)]
B = int(os.environ.get("PROBE_BATCH", "32"))
prep = eng.prepare(pool[:B])
qarr = (C.POINTER(ffi.Query) * B)(*[p.q for p in pool[:B]])
for _ in range(5):
    eng.free_raw(eng.execute_prepared(prep))
T = {"alloc": 0, "exec": 0, "merge": 0, "free": 0, "inner": 0}
N = 50
for _ in range(N):
    t0 = time.perf_counter()
    outs = (ffi.SqResult * (B * 8))()
    t1 = time.perf_counter()
    rc = lib.msq_execute_prepared(eng.ctx, prep[0], outs)
    t2 = time.perf_counter()
    lib.m_form_result_batch(qarr, outs, B, 8, None, 0)
    t3 = time.perf_counter()
    lib.m_sqresult_free_batch(outs, B * 8)
    t4 = time.perf_counter()
    tm = eng.timing()
    T["alloc"] += t1 - t0; T["exec"] += t2 - t1; T["merge"] += t3 - t2; T["free"] += t4 - t3
    T["inner"] += sum(tm["host_ms"].values()) / 1e3
print({k: round(v / N * 1e3, 3) for k, v in T.items()}, "ms per step; sizeof(SqResult)=", C.sizeof(ffi.SqResult))
