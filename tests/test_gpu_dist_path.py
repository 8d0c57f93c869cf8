"""The device path of ShardGather (pinned staging -> GPU tensors -> RCCL all_gather -> pinned -> unpack)
with a ONE-rank nccl group on the one GPU of the test box, driven from a worker thread as bench.py does
for --gpus > 1: torch's current device and stream are per thread there.  The merged answer must equal
the direct single-process merge.  (The 2-rank layout itself is covered on CPU by test_dist_gloo.py.)"""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

from util import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import ctypes as C, json, os, sys, threading
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import torch, torch.distributed as dist
from marlin_amd import GpuEngine, Index, ffi
from marlin_amd.dist import ShardGather
from test_dist_gloo import corpus, QUERIES, result_view

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lib = ffi.load()
ix = Index(corpus()[0], num_shards=2).add_documents(corpus()[1])
eng = GpuEngine(ix, device=0, num_ctx=2)
pqs = [ix.query(dict(q, fullScan=True)) for q in QUERIES]
nq = len(pqs)
prep = eng.prepare(pqs)
qarr = (C.POINTER(ffi.Query) * nq)(*[p.q for p in pqs])
# direct merge
outs = eng.execute_prepared(prep)
res = (ffi.Result * nq)()
lib.m_form_result_batch(qarr, outs, nq, 2, res, 1)
direct = [result_view(res[i], pqs[i]) for i in range(nq)]
eng.free_raw(outs)
# through the gather, twice in a row from two worker threads (slot reuse, per-thread device state)
g = ShardGather(lib, dist, 1, 0, nq, 2, 100, len(ix.facets), 40, 4, device="cuda:0", nslots=2, cardinality=True)
got = {}
lock = threading.Lock()
def work(t):
    o = eng.execute_prepared(prep, t)
    g.pack(o, t)                    # own buffer set: packing and unpacking run outside the ordering lock
    with lock:
        g.issue(t)
    sq = g.finish(t, share=True)
    r = (ffi.Result * nq)()
    lib.m_form_result_batch(qarr, sq, nq, 2, r, 1)
    got[t] = [result_view(r[i], pqs[i]) for i in range(nq)]
    g.free(sq); eng.free_raw(o)
th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
[t.start() for t in th]; [t.join() for t in th]
print("RESULT " + json.dumps({"direct": direct, "t0": got[0], "t1": got[1]}))
eng.close(); ix.close()
dist.destroy_process_group()
'''


def test_gather_on_device_from_worker_threads(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", REPO=ROOT)
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert p.returncode == 0 and line, p.stdout[-3000:]
    r = json.loads(line[0][7:])
    assert r["direct"][0]["total"] > 0
    assert r["t0"] == r["direct"] and r["t1"] == r["direct"]


def test_rccl_gather_from_the_c_pipeline():
    """DESIGN.md (e) in C: the serving loop's own threads pack the result slots, issue ncclAllGather (mgpu_comm_*, a
    ONE-rank communicator on the one GPU of the box) in step order and merge the rank's share of the queries.  Same
    hit sums as the loop without the exchange, several passes, batches of different queries in flight; a batch with a
    cardinality aggregation is refused, not mis-merged."""
    from marlin_amd import GpuEngine, GpuError, Index
    from test_dist_gloo import corpus, QUERIES
    ix = Index(corpus()[0], num_shards=2).add_documents(corpus()[1])
    eng = GpuEngine(ix, device=0, num_ctx=4)
    try:
        plain = [q for q in QUERIES if "cardinality" not in str(q)]
        pqs = [ix.query(dict(q, fullScan=True)) for q in plain]
        batches = [pqs, pqs[::-1], pqs[1:] + pqs[:1]]
        ress = [eng.resolve(b) for b in batches]
        eng.warm_pipeline_resolved(ress, 4)
        want, _ = eng.run_pipeline_resolved(ress, 0, 11, 4)
        assert want > 0
        eng.enable_gather(eng.comm_unique_id(), 0, 1, len(pqs), 100, len(ix.facets), 40, 4, nthreads=4)
        for _ in range(2):
            got, _ = eng.run_pipeline_resolved(ress, 0, 11, 4)
            assert got == want
        card = [ix.query(dict(q, fullScan=True)) for q in QUERIES if "cardinality" in str(q)][:1] * len(pqs)
        rc = eng.resolve(card)
        with pytest.raises(GpuError):
            eng.run_pipeline_resolved([rc], 0, 2, 4)
        got, _ = eng.run_pipeline_resolved(ress, 0, 5, 4)          # the loop is usable after the refusal
        assert got > 0
        for r in ress + [rc]:
            eng.free_resolved(r)
    finally:
        eng.close(); ix.close()
