"""diagnostic: phase clocks of the rank / select kernels on the bench workload (not part of the product)"""
import ctypes as C, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
from marlin_amd import ffi
lib = ffi.load()
gpu = C.CDLL(os.path.join(ffi.LIB_DIR, "libmarlin_gpu.so"))
gpu.mgk_debug_phases.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
_orig = bench.json.dumps
def hook(line):
    return _orig(line)
import torch
torch.cuda.set_device(0)
from marlin_amd import GpuEngine, Index
shards = bench.build_shards(lib, 10_000_000, list(range(8)))
ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
eng = GpuEngine(ix, device=0)
lib.msq_set_plan_threads(16)
NB = int(os.environ.get("PROBE_BATCH", "32"))
pool = [ix.query(j) for j in bench.make_queries(lib, NB)]
only = os.environ.get("PROBE_FACET_ONLY")        # count facets of one field only (index into cat, hc)
if only is not None:
    for pq in pool:
        for f in range(2):
            pq.q.contents.cfg.facet_enabled[f] = 1 if f == int(only) else 0
prep = eng.prepare(pool[:NB])
for _ in range(3):
    eng.free_raw(eng.execute_prepared(prep))
eng.set_profiling(True)
eng.free_raw(eng.execute_prepared(prep))
print("stage ms (no phase clocks):", {k: round(x["ms"], 4) for k, x in eng.timing()["stages"].items()})
gpu.mgk_debug_phases(1, None)
for _ in range(1):
    eng.free_raw(eng.execute_prepared(prep))
print("stage ms (with phase clocks):", {k: round(x["ms"], 4) for k, x in eng.timing()["stages"].items()})
out = (C.c_ulonglong * 32)()
gpu.mgk_debug_phases(0, out)
v = list(out)
tiles = max(1, v[3]); jobs = max(1, v[8])
print("rank whole-before-phase2 per tile us: %.2f (setup + phase1 + phase1b)" % (v[16] / max(1, v[3]) / 100))
print("rank tiles/launch", tiles / 1, "per-tile us: setup %.2f phase1 %.2f phase2 %.2f" % (v[0] / tiles / 100, v[1] / tiles / 100, v[2] / tiles / 100))
print("select fused jobs/launch", jobs / 1, "us per WG (all WGs incl. idle contribute to setup): setup %.2f chunks %.2f decide %.2f final %.2f" % (v[4] / jobs / 100, v[5] / jobs / 100, v[6] / jobs / 100, v[7] / jobs / 100))
for slot, name in ((13, "ANY exit WG"), (9, "whole WG"), (10, "final"), (11, "mid whole WG"), (12, "mid final")):
    print("select max %s: %.1f us at M=%d" % (name, (v[slot] >> 32) / 100, v[slot] & 0xFFFFFFFF))
mj = max(1, v[21])
print("mid select jobs %d, passes %.2f per job, us per job: setup %.2f sweeps %.2f decide %.2f final %.2f" % (v[21], v[22] / mj, v[17] / mj / 100, v[18] / mj / 100, v[19] / mj / 100, v[20] / mj / 100))
fw = max(1, v[24])
print("formula workgroups %d: lists %.1f, elements %.1f per workgroup; us per workgroup: before first barrier %.2f, tiles + formula %.2f" % (v[24], v[25] / fw, v[26] / fw, v[27] / fw / 100, v[28] / fw / 100))
cal = (C.c_ulonglong * 3)()
gpu.mgk_debug_clock_cal.argtypes = [C.POINTER(C.c_ulonglong)]
gpu.mgk_debug_clock_cal(cal)
print("clock cal: wall ticks %d, shader cycles %d, event ns %d" % (cal[0], cal[1], cal[2]))
print("select kernel span (first WG start -> last WG end): %.1f us" % ((v[15] - ((~v[14]) & 0xFFFFFFFFFFFFFFFF)) / 100))
eng.free_prepared(prep)
eng.close(); ix.close()
