"""N > 1: two ranks (gloo, CPU) each own one shard, gather their result slots to rank 0 and
merge; the answer must equal the single-process merge of both shards.  Per-shard results come
from the oracle here (no GPU in this container) -- what is under test is the slot wire format,
the all_gather layout and the merge, i.e. everything bench.py adds for --gpus > 1."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

from util import ROOT

WORKER = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import torch, torch.distributed as dist
from marlin_amd import Index, ffi
from marlin_amd.dist import ShardGather
from oracle import orc_py
from test_dist_gloo import corpus, QUERIES, result_view

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
lib = ffi.load(); orc = orc_py.load()
ix = Index(corpus()[0], num_shards=world).add_documents(corpus()[1])      # same routing on every rank
pqs = [ix.query(dict(q, fullScan=True)) for q in QUERIES]
nq = len(pqs)
# this rank's shard only
outs = (ffi.SqResult * nq)()
for i, pq in enumerate(pqs):
    assert orc.orc_execute_squery(ix.shards[rank], pq.q, rank, 0, C.byref(outs[i])) == 0
g = ShardGather(lib, dist, world, rank, nq, 1, 100, len(ix.facets), 40, 4, cardinality=True)
sq = g.gather(outs)
if rank == 0:
    qarr = (C.POINTER(ffi.Query) * nq)(*[p.q for p in pqs])
    res = (ffi.Result * nq)()
    lib.m_form_result_batch(qarr, sq, nq, world, res, 1)
    print("RESULT " + json.dumps([result_view(res[i], pqs[i]) for i in range(nq)]))
    g.free(sq)
# the shared-merge mode: every rank merges its own share of the queries
for i, pq in enumerate(pqs):
    assert orc.orc_execute_squery(ix.shards[rank], pq.q, rank, 0, C.byref(outs[i])) == 0
sq2 = g.gather(outs, share=True)
mine = g.my_queries()
qsub = (C.POINTER(ffi.Query) * len(mine))(*[pqs[i].q for i in mine])
res2 = (ffi.Result * len(mine))()
lib.m_form_result_batch(qsub, sq2, len(mine), world, res2, 1)
print("SHARE %d " % rank + json.dumps({str(q): result_view(res2[j], pqs[q]) for j, q in enumerate(mine)}))
g.free(sq2)
dist.barrier()
dist.destroy_process_group()
'''

QUERIES = [
    {"q": "red shoe", "hitsPerPage": 20},
    {"q": "blue", "rankBy": {"price": "desc"}, "hitsPerPage": 15, "aggs": {"s": {"stats": {"field": "price"}}, "m": {"max": {"field": "price"}}}},
    {"q": "", "filter": {"brand": "acme"}, "sortBy": {"price": "asc"}, "hitsPerPage": 30},
    {"q": "grn shirt", "hitsPerPage": 10},
    {"q": "shoe", "aggs": {"c": {"cardinality": {"field": "size"}}, "b": {"cardinality": {"field": "brand"}}, "a": {"avg": {"field": "price"}}}},
    {"q": "", "filter": {"size": "s1"}, "aggs": {"c": {"cardinality": {"field": "brand"}}}},
]


def corpus():
    colors = ["red", "blue", "green", "black"]
    items = ["shoe", "shirt", "hat", "sock", "coat"]
    brands = ["acme", "zenith", "orbit"]
    docs = []
    for i in range(400):
        docs.append({"id": i, "title": "%s %s %s" % (colors[i % 4], items[i % 5], "sale" if i % 7 == 0 else "new"),
                     "brand": brands[i % 3], "size": "s%d" % (i % 6), "price": float((i * 37) % 211)})
    settings = {"indexedFields": ["title", "price"], "facetFields": ["brand", "size"], "hitsPerPage": 20}
    return settings, docs


def result_view(res, pq):
    q = pq.q.contents
    v = {"total": res.total_hits, "ranks": [], "facets": {}, "aggs": []}
    for i in range(res.total_ranks):
        r = res.ranks[i]
        v["ranks"].append([r.shard_id, r.docid, r.typos, r.proximity, r.position, r.exact, r.field, r.compare])
    for f in range(res.num_facets):
        if q.cfg.facet_enabled[f]:
            v["facets"][str(f)] = sorted([res.fc[f][x].count, res.fc[f][x].facet_id] for x in range(res.fcount[f]))
    for a in range(res.num_aggs):
        ag = res.aggs[a]
        v["aggs"].append([ag.type, ag.value, ag.sum, ag.min, ag.max, ag.count, sorted(ag.card_ids[i] for i in range(ag.card_n))])
    return v


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_gather_equals_single_process_merge(tmp_path):
    from marlin_amd import Index, ffi
    from oracle import orc_py
    lib = ffi.load()
    settings, docs = corpus()
    ix = Index(settings, num_shards=2).add_documents(docs)
    pqs = [ix.query(dict(q, fullScan=True)) for q in QUERIES]
    expect = []
    for pq in pqs:
        outs = orc_py.execute_raw(ix, pq)
        res = ffi.Result()
        lib.m_form_result(pq.q, outs, 2, C.byref(res))
        expect.append(result_view(res, pq))
        lib.m_result_free(C.byref(res))
        for s in range(2):
            lib.m_sqresult_free(C.byref(outs[s]))
    assert expect[0]["total"] > 0 and expect[1]["aggs"]
    assert len(expect[4]["aggs"][0][6]) == 6 and len(expect[4]["aggs"][1][6]) == 3      # cardinality: distinct ids across both shards

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), REPO=ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT ")][0]
    got = json.loads(line[len("RESULT "):])
    want = json.loads(json.dumps(expect))
    assert got == want
    shared = {}
    for r in range(2):
        l = [l for l in outs[r].splitlines() if l.startswith("SHARE %d " % r)][0]
        shared.update(json.loads(l[len("SHARE %d " % r):]))
    assert [shared[str(i)] for i in range(len(want))] == want
