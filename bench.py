#!/usr/bin/env python3
"""bench.py -- queries/sec of the Marlin shard-query hot path on MI355X.

Workload (BASELINE.json `metric` / configs[2]): a 10M-document synthetic Zipf index in 8 shards,
a stream of 3-term queries (two typo'd frequent words + a 3-letter prefix, i.e. 6 terms with
typo / prefix expansion), facets on 2 fields (64 and 100k buckets), rankBy a unique numeric
field, top-10, fullScan.  A "step" is one batch of --batch x --sub queries (default 128 x 64 = 8192) taken through
execute_squery on every shard (plan lowering + GPU launches) and the form_result merge; the device takes it as --sub
launches of --batch queries each, --inflight of them in flight.

  N = 1 : all 8 shards on one GPU.
  N > 1 : shards split over the ranks (8/N each; the corpus is fixed but the queries per step grow with N, so a rank's work per step is constant => weak scaling); every rank runs
          the whole query batch on its shards, per-shard result slots are exchanged with ONE RCCL
          all_gather (ncclAllGather issued from the C serving loop, mgpu_comm_*; torch.distributed only hands the RCCL id round)
          and every rank merges its share of the queries.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel stage (HIP-event timed on the launch stream) and `cpu_baseline` (the C oracle = port of the
reference's CPU path run as the reference runs it: a thread pool over (query, shard) jobs on the host's cores,
bounded sample).

Side measurements, never the metric's configuration (the line's config.workload says so): --default-mode (fullScan false: shards
above fullScanThreshold hits ranked on a sample), --variant filter | filter-distinct | aggs | strfilter | facets-only | page |
words1 | words4 | sort (the stream with one ingredient changed: DESIGN.md (d)), --docs 50000000 (configs[4]).
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4D41524C494E
VOCAB = 262144
HC_BUCKETS = 100000
TOTAL_SHARDS = 8
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


def build_shards(lib, total_docs, my_shards):
    out = [None] * len(my_shards)

    def work(i, s):
        out[i] = lib.msynth_zipf(total_docs, VOCAB, HC_BUCKETS, SEED, s, TOTAL_SHARDS)
    th = [threading.Thread(target=work, args=(i, s)) for i, s in enumerate(my_shards)]
    [t.start() for t in th]
    [t.join() for t in th]
    return out


def zipf_word(lib, r):
    buf = C.create_string_buffer(16)
    lib.msynth_zipf_word(VOCAB, SEED, r, buf)
    return buf.value.decode()


def make_queries(lib, n, full_scan=True, variant=None):
    """3-word queries: two of the 24 most frequent words with one substituted letter (typo
    distance 1) + the first three letters of a frequent word (prefix -> twid posting)."""
    import random
    rng = random.Random(1234)
    qs = []
    for _ in range(n):
        ws = []
        for _ in range(2):
            w = zipf_word(lib, rng.randrange(0, 24))
            p = rng.randrange(1, len(w))
            c = rng.choice([ch for ch in "abcdefghijklmnopqrstuvwxyz" if ch != w[p]])
            ws.append(w[:p] + c + w[p + 1:])
        pre = zipf_word(lib, rng.randrange(0, 64))[:3]
        q = {"q": " ".join(ws + [pre]), "fullScan": full_scan, "rankBy": {"num": "desc"},
             "hitsPerPage": 10, "maxFacetResults": 10}
        # side measurements (configs[3]'s ingredients on the bench stream; never the metric's configuration)
        if variant == "filter":
            q["filter"] = {"price": {"$lt": 300000}} if len(qs) % 2 else {"price": {"$gte": 300000}}
        elif variant == "filter-distinct":                       # no two queries of the stream share their range
            q["filter"] = {"price": {"$lt": 200000 + 97 * len(qs)}} if len(qs) % 2 else {"price": {"$gte": 200000 + 97 * len(qs)}}
        elif variant == "strfilter":                              # a facet value as filter (cat has 64 values)
            q["filter"] = {"cat": "cat%02d" % (len(qs) % 64)}
        elif variant == "facets-only":
            q["hitsPerPage"] = 0
        elif variant == "page":                                   # third page of 100 hits
            q["hitsPerPage"] = 100; q["page"] = 3
        elif variant == "words1":
            q["q"] = ws[0]
        elif variant == "words4":
            q["q"] = " ".join(ws + [zipf_word(lib, rng.randrange(0, 24)), pre])
        elif variant == "sort":
            del q["rankBy"]; q["sortBy"] = {"price": "asc"}
        elif variant == "aggs":
            q["aggs"] = {"a": {"avg": {"field": "price"}}, "s": {"stats": {"field": "num"}}, "c": {"cardinality": {"field": "cat"}}}
        qs.append(q)
    return qs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=128, help="queries per device launch sequence (sub-batch) and GPU")
    ap.add_argument("--sub", type=int, default=64, help="sub-batches per step: a step is --batch x --sub queries per GPU")
    ap.add_argument("--cpu-queries", type=int, default=512,
                    help="bench queries the CPU baseline runs (about 25 s of CPU work at the default, spread over the cores)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = the cores this process may use, at most 16)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: gloo collectives on host tensors, every rank on cuda:0")
    ap.add_argument("--no-latency", action="store_true", help="skip the 200 single-query latency probes (cleaner kernel profiles)")
    ap.add_argument("--distinct", type=int, default=4096, help="distinct queries the stream cycles through")
    ap.add_argument("--inflight", type=int, default=None, help="batches kept in flight per GPU (= contexts; one host thread per two contexts on one GPU); "
                    "default 16, or 8 above 20M documents (a context's arena grows with the shards' hit bounds: 16 of them do not fit beside a 31 GB mirror)")
    ap.add_argument("--default-mode", action="store_true", help="NOT the metric's configuration: the reference's as-shipped mode (fullScan false): "
                    "shards with more than fullScanThreshold = 25000 hits are ranked on a sample (row f-3)")
    ap.add_argument("--variant", choices=["filter", "filter-distinct", "aggs", "strfilter", "facets-only", "page", "words1", "words4", "sort"], default=None, help="NOT the metric's configuration: the stream with a numeric range "
                    "filter, or with avg / stats / cardinality aggregations (BASELINE configs[3]'s ingredients)")
    args = ap.parse_args()
    if args.inflight is None:
        args.inflight = 16 if args.docs <= 20_000_000 else 8

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if TOTAL_SHARDS % world:
        raise SystemExit("--gpus must divide %d" % TOTAL_SHARDS)

    import torch
    dist = None
    if args.rehearse_gloo:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from marlin_amd import GpuEngine, Index, ffi
    lib = ffi.load()
    local = TOTAL_SHARDS // world
    my_shards = list(range(rank * local, (rank + 1) * local))
    t0 = time.time()
    shards = build_shards(lib, args.docs, my_shards)
    ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    t_build = time.time() - t0
    t0 = time.time()
    eng = GpuEngine(ix, device=local_rank, num_ctx=max(1, args.inflight))
    t_mirror = time.time() - t0
    lib.msq_set_plan_threads(min(16, max(1, (os.cpu_count() or 8) // max(world, 1))))
    mirror_bytes = sum(lib.mgpu_shard_bytes(lib.mmirror_gpu(m)) for m in eng.mirrors)

    # queries per step grow with the number of GPUs so that a rank's batch (queries x its shards)
    # keeps the same size: the corpus is fixed, the query stream is not: per-GPU work per step is constant (weak scaling)
    B = args.batch * world
    S = max(1, args.sub)
    nq_pool = max(B, args.distinct // B * B)   # distinct queries of the stream (rounds 1 and 2 cycled through the first 256
                                         # of the same generator; the mean hit count of a sample moves the rate: 18k there,
                                         # 28k over 4096)
    pool = [ix.query(j) for j in make_queries(lib, nq_pool, full_scan=not args.default_mode, variant=args.variant)]
    K_TOP, NF, FMAX, NAGG = 10, 2, 20, 0
    slot_b = lib.m_slot_bytes(K_TOP, NF, FMAX, NAGG)

    def batch_queries(step):
        return [pool[(step * B + i) % nq_pool] for i in range(B)]

    gather = None
    c_loop = world == 1 or not args.rehearse_gloo      # the native loop (mplan.c); the gloo rehearsal shares one GPU among the
                                                       # ranks, which RCCL cannot do: it keeps the torch.distributed gather
    if world > 1 and not c_loop:
        from marlin_amd.dist import ShardGather
        gather = ShardGather(lib, dist, world, rank, B, local, K_TOP, NF, FMAX, NAGG,
                             device=None if args.rehearse_gloo else "cuda:%d" % local_rank, nslots=max(1, args.inflight))

    # Term lookup (the DICTIONARY half of lookup_terms: dtrie_lookup_term / dtrie_lookup_exact) is outside the north-star
    # path ("per-shard query execution after term lookup"): resolved once per distinct batch, before timing, with the
    # typo walks on the device (row f-1).  Everything after it -- process_termresult (posting selection, all_wordids, zero-typo
    # sets), exact wids, filter leaves: msq_plan_resolved -- is built INSIDE the timed loop, per batch, by the pipeline's threads.
    t0 = time.time()
    nbatches = max(1, nq_pool // B)
    resolved = [eng.resolve(batch_queries(j), device_lookup=True) for j in range(nbatches)]
    t_prepare = time.time() - t0
    # prepared batches (plans built ahead) only for the N > 1 loop, the stage profile and the replay line
    nprep = nbatches        # all of them: the stage profile below averages over the same batches the timed loop cycles
    prepared = [eng.plan_resolved(resolved[j]) for j in range(nprep)]

    F = max(1, args.inflight)
    order = threading.Condition()
    gather_turn = [0]

    qsub_cache = {}

    def run_step(step, slot=0, seq=None):
        """one batch: GPU execute_squery for every (query, shard) + the form_result merge.  With
        --gpus > 1 the cross-rank gather of step `seq` waits for its turn, so every rank issues its
        collectives in the same order whatever the host threads do."""
        prep = prepared[step % nprep]
        pqs = prep[2]
        outs = eng.execute_prepared(prep, slot)
        hits = 0
        if world == 1:
            qarr = (C.POINTER(ffi.Query) * B)(*[p.q for p in pqs])
            hits = lib.m_form_result_batch(qarr, outs, B, local, None, 0)
        else:
            gather.pack(outs, slot)
            with order:
                while seq is not None and gather_turn[0] != seq:
                    if gather_turn[0] < 0:
                        raise RuntimeError("another host thread failed")
                    order.wait()
                gather.issue(slot)                       # one RCCL all_gather of fixed-size result slots, in step order
                gather_turn[0] += 1
                order.notify_all()
            sq = gather.finish(slot, share=True)         # waits for this slot's copies only
            # every rank merges its own share of the batch's queries (the gather delivered all slots everywhere)
            mine = gather.my_queries()
            if mine:
                key = step % nprep                       # the pointer array of a prepared batch is built once
                if key not in qsub_cache:
                    qsub_cache[key] = (C.POINTER(ffi.Query) * len(mine))(*[pqs[i].q for i in mine])
                hits = lib.m_form_result_batch(qsub_cache[key], sq, len(mine), TOTAL_SHARDS, None, 0)
            gather.free(sq)
        eng.free_raw(outs)
        return hits

    plan_ms = [0.0]

    def run_steps(first, count):
        """`count` steps with up to F batches in flight (one host thread and one context each)"""
        if c_loop:
            # the whole loop is native (host threads, contexts, plans, execute, [RCCL all-gather,] merge: mplan.c)
            hits, pms = eng.run_pipeline_resolved(resolved, first, count, F)
            plan_ms[0] = pms
            return hits
        gather_turn[0] = 0
        hits = [0] * F
        err = []

        def worker(t):
            try:
                torch.cuda.set_device(local_rank)        # torch's current device is per thread
                for i in range(t, count, F):
                    hits[t] += run_step(first + i, t, i)
            except BaseException as e:          # surface failures instead of hanging the other thread
                err.append(e)
                with order:
                    gather_turn[0] = -1
                    order.notify_all()
        if F == 1:
            worker(0)
        else:
            th = [threading.Thread(target=worker, args=(t,)) for t in range(F)]
            [t.start() for t in th]
            [t.join() for t in th]
        if err:
            raise err[0]
        return sum(hits)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, like the mirror upload: every context sees every distinct batch once, so that its arena, pinned
    # buffers and histogram pool have their final size (and, with several ranks, every slot's gather buffers and
    # the RCCL communicator have been used) whatever --warmup says.  The pipeline's host threads are created
    # here too and stay parked between runs: nothing of this is left for the timed region.
    t0 = time.time()
    if c_loop:
        eng.warm_pipeline_resolved(resolved, F)
        if world > 1:
            # the exchange of (e): rank 0's RCCL id goes round through torch.distributed, everything after that is C
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(eng.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            eng.enable_gather(bytes(idt.cpu().numpy().tobytes()), rank, world, B, K_TOP, NF, FMAX, NAGG, nthreads=F)
    else:
        for slot in range(F):
            for j in range(nprep):
                eng.free_raw(eng.execute_prepared(prepared[j], slot))
        gather_turn[0] = 0
        for slot in range(F):
            run_step(0, slot, slot)
    t_warm = time.time() - t0
    barrier()

    run_steps(0, args.warmup * S)
    barrier()
    t0 = time.perf_counter()
    total_hits = run_steps(args.warmup * S, args.steps * S)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        th = torch.tensor([float(total_hits)], dtype=torch.float64, device="cpu" if args.rehearse_gloo else "cuda")
        dist.all_reduce(th)
        total_hits = int(th.item())
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    qps = args.steps * S * B / dt
    plan_ms_batch = plan_ms[0] / max(1, args.steps * S)

    # the line rounds 1 and 2 reported: the first 256 queries as two batches prepared ahead and replayed (no plan building
    # in the loop) -- kept for continuity only
    qps_replay = None
    if world == 1 and nprep >= 2 and B == 128 and not args.no_latency:      # (left out of profiled runs: its two light batches would skew the per-kernel averages)
        eng.run_pipeline(prepared[:2], 0, 2 * S, F)
        torch.cuda.synchronize(); t0r = time.perf_counter()
        eng.run_pipeline(prepared[:2], 0, 10 * S, F)
        qps_replay = 10 * S * B / (time.perf_counter() - t0r)

    # ---- single-query latency (batch of 1, unloaded), p50 / p99 over 200 queries
    lat, lat_full, lat_dev = [], [], []
    if world == 1 and not args.no_latency:
        for i in range(200):
            pq = pool[i % nq_pool]
            qa = (C.POINTER(ffi.Query) * 1)(pq.q)
            t0l = time.perf_counter()
            pr = eng.prepare([pq])
            t1 = time.perf_counter()
            o = eng.execute_prepared(pr)
            lib.m_form_result_batch(qa, o, 1, local, None, 0)
            t2 = time.perf_counter()
            lat.append((t2 - t1) * 1e3)
            lat_full.append((t2 - t0l) * 1e3)
            eng.free_raw(o)
            eng.free_prepared(pr)
        lat.sort(); lat_full.sort()
        # the same with the typo walks of the term lookup on the device (row f-1)
        for i in range(200):
            pq = pool[i % nq_pool]
            qa = (C.POINTER(ffi.Query) * 1)(pq.q)
            t0l = time.perf_counter()
            pr = eng.prepare([pq], device_lookup=True)
            o = eng.execute_prepared(pr)
            lib.m_form_result_batch(qa, o, 1, local, None, 0)
            lat_dev.append((time.perf_counter() - t0l) * 1e3)
            eng.free_raw(o)
            eng.free_prepared(pr)
        lat_dev.sort()

    # ---- the same stream with the WHOLE of lookup_terms inside the loop (row f-1): every step resolves its 128 queries x
    # shards (typo walks on the device: signature filter + walk of the candidates, on the step's own context; exact
    # descents on the host), builds the plans, executes and merges -- the same native pipeline, F batches in flight
    qps_lookup, lookup_ms_batch = None, None
    if world == 1 and not args.no_latency:
        eng.run_pipeline_queries(pool, B, 0, 2 * F, F)                  # the lookup buffers of every context
        torch.cuda.synchronize(); t0b = time.perf_counter()
        nlk = 2 * S
        _, lkms, _ = eng.run_pipeline_queries(pool, B, 0, nlk, F)
        qps_lookup = nlk * B / (time.perf_counter() - t0b)
        lookup_ms_batch = lkms / nlk

    # ---- roofline of the dominant kernel stage: HIP events on the launch stream, profiled batches
    roof = None
    stage_tab = {}
    eng.set_profiling(True)
    acc = {}
    host_acc = {}
    nprof = nprep if nprep > 2 else 5
    for s in range(nprof):
        o = eng.execute_prepared(prepared[s % nprep])
        eng.free_raw(o)
        tm = eng.timing()
        for hk, hv in tm["host_ms"].items():
            host_acc[hk] = host_acc.get(hk, 0.0) + hv / nprof
        for name, v in tm["stages"].items():
            a = acc.setdefault(name, [0.0, 0])
            a[0] += v["ms"]; a[1] += v["bytes"]
    eng.set_profiling(False)
    kernel_stages = [k for k in acc if k not in ("upload+clear", "download")]
    dom = max(kernel_stages, key=lambda k: acc[k][0])
    for k in acc:
        ms, by = acc[k][0] / nprof, acc[k][1] / nprof
        stage_tab[k] = {"ms": round(ms, 4), "MB": round(by / 1e6, 3), "GBps": round(by / ms / 1e6, 1) if ms > 0 else None}
    ach = acc[dom][1] / acc[dom][0] / 1e6          # bytes / ms -> GB/s
    KERNELS = {"rank": "k_rank_tile + k_rank_overflow", "select": "k_select_first + k_select_mid (+ k_select_items, k_select_final)", "formula+compact": "k_formula + k_order + k_order_copy",
               "facets+aggs": "k_docproc + k_aggfinal", "facet-keys": "k_fkeys + k_card + k_facet_hash", "scatter": "k_scatter_list"}
    # HBM traffic of the dominant kernel comes from separate PMC passes (rocprofv3 cannot run inside the
    # timed program): a committed file, NOT a measurement of this run -- `traffic_source` says so in the line
    traffic, traffic_src = None, None
    PMC_FILE = "profiles/r03_pmc_fetch_write_per_launch.json"
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            pm = json.load(f)
        if dom == "rank" and B == 128 and world == 1 and args.docs == 10_000_000:
            kk = pm["kernels"]
            rk = [k for k in kk if "k_rank_tile" in k][0]
            traffic = kk[rk]["hbm_bytes_per_launch"] + kk["k_rank_overflow"]["hbm_bytes_per_launch"]
            traffic_src = ("constant read from the committed %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                           "workload taken earlier with tools/gpu_profile.sh), not measured in this run" % PMC_FILE)
    except (OSError, KeyError, ValueError, IndexError):
        pass
    roof = {"bound": "hbm", "kernel": KERNELS.get(dom, dom), "stage": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "launch_ms": round(acc[dom][0] / nprof, 4), "algorithmic_MB_per_launch": round(acc[dom][1] / nprof / 1e6, 3)}

    # ---- CPU baseline: the C oracle (port of the reference's execute_squery) on this host, 1 core
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import orc_py
        orc_py.load()
        n = max(1, min(args.cpu_queries, len(pool)))
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        nthr = args.cpu_threads if args.cpu_threads > 0 else max(1, min(16, avail))
        model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                for ln in f:
                    if ln.startswith("model name"):
                        model = ln.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        # the reference's shape: search_pool runs one execute_squery job per (query, shard) (query.c:435-452)
        wall, lk, busy, chk = orc_py.bench_pool(ix, pool[:n], nthr)
        frac = lk / max(1e-9, busy)                 # share of the workers' time spent in dictionary lookups
        cpu = {"value": round(n / max(1e-9, wall * (1.0 - frac)), 3), "unit": "queries/s", "cores": nthr, "kind": "port",
               "sample": "%d of the bench queries x 8 shards = %d execute_squery jobs over a pool of %d threads, "
                         "oracle/orc_squery.c; dictionary lookup time (%.1f%% of worker time) excluded as on the GPU side"
                         % (n, n * local, nthr, 100 * frac),
               "with_term_lookup": round(n / wall, 3), "one_core_equivalent": round(n / max(1e-9, busy - lk), 3),
               "cpu_seconds": round(busy, 2), "host_cpu": model, "host_cores_visible": os.cpu_count(), "host_cores_usable": avail,
               "total_hits": chk}

    if rank == 0:
        line = {
            "metric": "queries/sec, 10M-doc index, 3-term typo+prefix query w/ facets",
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[2]: %d docs Zipf corpus, %d shards, 3-word typo+prefix query, 2 facet fields, "
                                   "rankBy num desc, top-10, %s" % (args.docs, TOTAL_SHARDS, ("DEFAULT MODE (fullScan false: not the metric's configuration)" if args.default_mode else "fullScan") + (" + VARIANT %s (not the metric's configuration)" % args.variant if args.variant else "")),
                       "queries_per_step": B * S, "queries_per_device_batch": B, "device_batches_per_step": S, "batches_in_flight": F, "shards_per_gpu": local, "vocab": VOCAB,
                       "mean_hits_per_query": round(total_hits / max(1, args.steps * S * B), 1),
                       "distinct_queries": nq_pool, "plans_in_timed_region": bool(c_loop),
                       "exchange": None if world == 1 else ("ncclAllGather of result slots from the C pipeline (mgpu_comm_*)" if c_loop else "torch.distributed all_gather (gloo rehearsal)"),
                       "timed_region": "per batch: process_termresult for 128 queries x 8 shards (msq_plan_resolved) + plan lowering + "
                                       "upload + kernels + download + decode + form_result; dictionary lookups resolved before"},
            "plan_build_ms_per_batch": round(plan_ms_batch, 3),
            "replay_256_prepared_qps": round(qps_replay, 1) if qps_replay else None,
            "roofline": roof, "cpu_baseline": cpu,
            "latency_ms": {"p50": round(lat[len(lat) // 2], 3), "p99": round(lat[int(len(lat) * 0.99)], 3),
                           "with_term_lookup_p50": round(lat_full[len(lat_full) // 2], 3),
                           "with_term_lookup_p99": round(lat_full[int(len(lat_full) * 0.99)], 3),
                           "with_device_term_lookup_p50": round(lat_dev[len(lat_dev) // 2], 3),
                           "with_device_term_lookup_p99": round(lat_dev[int(len(lat_dev) * 0.99)], 3)} if lat else None,
            "with_device_term_lookup_qps": round(qps_lookup, 1) if qps_lookup else None,
            "lookup_ms_per_batch": round(lookup_ms_batch, 3) if lookup_ms_batch else None,
            "stages": stage_tab,
            "host_ms_per_step": {k: round(v, 3) for k, v in host_acc.items()},
            "setup_s": {"build_shards": round(t_build, 1), "mirror": round(t_mirror, 1), "term_lookup": round(t_prepare, 2),
                        "context_warm": round(t_warm, 2)},
            "mirror_GB": round(mirror_bytes / 1e9, 3),
        }
        print(json.dumps(line), flush=True)
    for pr in prepared:
        eng.free_prepared(pr)
    for r in resolved:
        eng.free_resolved(r)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
