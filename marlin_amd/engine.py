"""GPU execution of parsed queries and result assembly.

`GpuEngine` owns one mgpu context (= one HIP stream) and the HBM mirrors of an index's shards
on one device, and runs execute_squery for every (query, shard) pair of a batch in one set of
launches (msq_execute_batch -> mgpu_execute).  `assemble` is the merge half of form_result
(query.c:312-383) plus a dict rendering of the response fields the reference puts in JSON.
There is no CPU fallback: if libmarlin_gpu.so or a GPU is missing, construction fails.
"""
import ctypes as C

from . import ffi, hits


class GpuError(RuntimeError):
    pass


def assemble(index, pq, sqres, form=None, page_batch=None):
    """sqres: ctypes array of SqResult, one per shard.  Returns a dict shaped like the
    reference's response (query.c:312-422).
    form: the merge function (default: the product's m_form_result; the test oracle passes its own restatement).
    page_batch: a hits.PageBatch: the page's documents come from their stored (zlib) values and their strings are
    highlighted later, together with those of the batch's other queries, on the device."""
    lib = index.lib
    res = ffi.Result()
    (form or lib.m_form_result)(pq.q, sqres, index.num_shards, C.byref(res))
    q = pq.q.contents
    out = {"totalHits": res.total_hits, "numHits": res.num_hits, "page": q.page_num,
           "fullScan": bool(res.full_scan), "hits": [], "facets": {}, "ranks": []}
    for i in range(res.total_ranks):
        r = res.ranks[i]
        rec = {"docid": r.docid, "shard": r.shard_id, "typos": r.typos, "proximity": r.proximity,
               "position": r.position, "exact": r.exact, "field": r.field, "compare": r.compare}
        out["ranks"].append(rec)
        if res.page_s <= i < res.page_e:
            hit = {"_docid": r.docid, "_shard": r.shard_id}
            docs = index.docs[r.shard_id] if index.docs else None
            stored = index.stored[r.shard_id] if page_batch is not None and getattr(index, "stored", None) else None
            if stored:
                # the batch fills the marks in later: the hit must BE the object the recorder saw, not a copy of its keys
                m = hits.materialize(lib, pq, stored[r.docid - 1], batch=page_batch)
                m.update(hit)
                hit = m
            elif docs:
                hit.update(hits.materialize(lib, pq, docs[r.docid - 1]))     # getFields + highlight (row f-4)
            if pq.explain:
                hit["_explain"] = {k: rec[k] for k in ("typos", "proximity", "position", "exact", "field")}
            out["hits"].append(hit)
    for f in range(res.num_facets):
        if not q.cfg.facet_enabled[f] or f >= len(index.facets):
            continue
        lst = []
        for x in range(res.fcount[f]):
            fc = res.fc[f][x]
            lst.append({"key": index.facet_string(fc.shard_id, fc.facet_id), "count": fc.count,
                        "facet_id": fc.facet_id})
        out["facets"][index.facets[f]] = lst
    if res.num_aggs:
        aggs = {}
        for a in range(res.num_aggs):
            ag = res.aggs[a]
            name = pq.agg_names[a]
            if ag.type == ffi.MA_MAX or ag.type == ffi.MA_MIN:
                aggs[name] = {"value": ag.value}
            elif ag.type == ffi.MA_AVG:
                aggs[name] = {"value": (ag.sum / ag.count) if ag.count else float("nan"), "sum": ag.sum, "count": ag.count}
            elif ag.type == ffi.MA_STATS:
                aggs[name] = {"avg": (ag.sum / ag.count) if ag.count else float("nan"), "sum": ag.sum,
                              "min": ag.min, "max": ag.max, "count": ag.count}
            elif ag.type == ffi.MA_CARDINALITY:
                aggs[name] = {"value": ag.card_n}
        out["aggs"] = aggs
    lib.m_result_free(C.byref(res))
    return out


def shard_results_to_py(index, pq, sqres):
    """per-shard raw results (for stage-level parity tests)"""
    out = []
    q = pq.q.contents
    for s in range(index.num_shards):
        r = sqres[s]
        d = {"num_hits": r.num_hits, "rank_count": r.rank_count,
             "ranks": [(r.ranks[i].docid, r.ranks[i].typos, r.ranks[i].proximity, r.ranks[i].field,
                        r.ranks[i].position, r.ranks[i].exact, r.ranks[i].compare) for i in range(r.rank_count)],
             "facets": {}, "aggs": []}
        for f in range(r.num_facets):
            if q.cfg.facet_enabled[f]:
                d["facets"][f] = [(r.fc[f][x].facet_id, r.fc[f][x].count) for x in range(r.rcount[f])]
        for a in range(r.num_aggs):
            ag = r.aggs[a]
            d["aggs"].append({"type": ag.type, "value": ag.value, "sum": ag.sum, "min": ag.min, "max": ag.max,
                              "count": ag.count, "card": [ag.card_ids[i] for i in range(ag.card_n)]})
        if r.dbg_docid_map:
            d["docid_map"] = [r.dbg_docid_map[i] for i in range(r.dbg_words)]
        if r.dbg_all_ranks:
            d["all_ranks"] = [(x.docid, x.typos, x.proximity, x.field, x.position, x.exact, x.compare)
                              for x in (r.dbg_all_ranks[i] for i in range(r.num_hits))]
        out.append(d)
    return out


class GpuEngine:
    def __init__(self, index, device=0, scratch_bytes=0, num_ctx=1):
        """num_ctx > 1 creates further contexts (own stream, arena and staging buffers each) over the
        same shard mirrors, so that several host threads can keep batches in flight on one GPU."""
        self.lib = lib = ffi.load()
        self.index = index
        self.device = device
        if lib.mgpu_device_count() <= device:
            raise GpuError("no HIP device %d visible (mgpu_device_count=%d)" % (device, lib.mgpu_device_count()))
        self.ctxs = []
        for _ in range(max(1, num_ctx)):
            ctx = C.c_void_p()
            rc = lib.mgpu_ctx_create(device, scratch_bytes, C.byref(ctx))
            if rc:
                raise GpuError("mgpu_ctx_create: %d %s" % (rc, lib.mgpu_last_error().decode()))
            self.ctxs.append(ctx)
        self.ctx = self.ctxs[0]
        self.mirrors = []
        for s in index.shards:
            m = C.c_void_p()
            rc = lib.mmirror_create(s, device, C.byref(m))
            if rc:
                raise GpuError("mmirror_create: %d %s" % (rc, lib.mgpu_last_error().decode()))
            self.mirrors.append(m)
        self._marr = (C.c_void_p * len(self.mirrors))(*[m.value for m in self.mirrors])
        self._pipe = None
        self._pipe_n = 0

    # ---- row f-2: the index was written to
    def apply_index(self, new_index):
        """`new_index` holds the shards after an append-only bulk write (same documents first, same docids).  Every mirror
        takes the difference as a delta (mmirror_apply_shard -> mgpu_shard_apply_delta: merged posting array and window
        index rebuilt on the device, new documents appended into the mirror's headroom); a mirror without room left is
        rebuilt.  No batch may be in flight.  Returns per shard "delta" or "rebuilt"."""
        lib = self.lib
        if len(new_index.shards) != len(self.mirrors):
            raise GpuError("apply_index: shard count changed")
        how = []
        for i, s in enumerate(new_index.shards):
            rc = lib.mmirror_apply_shard(self.mirrors[i], s)
            if rc == 0:
                how.append("delta")
                continue
            # -3 (no headroom left) and -5 (not an append) are the expected reasons; after any other failure the old mirror
            # may be half-updated (include/marlin_gpu.h), so it is replaced all the same
            m = C.c_void_p()
            rc2 = lib.mmirror_create(s, self.device, C.byref(m))
            if rc2:
                raise GpuError("mmirror_create: %d %s" % (rc2, lib.mgpu_last_error().decode()))
            lib.mmirror_destroy(self.mirrors[i])
            self.mirrors[i] = m
            how.append("rebuilt")
        self._marr = (C.c_void_p * len(self.mirrors))(*[m.value for m in self.mirrors])
        self.index = new_index
        return how

    # ---- several GPUs: one process per GPU, the pipeline's batches end with an RCCL all-gather
    def comm_unique_id(self):
        """128 bytes rank 0 makes and every rank needs (hand them round with whatever the launcher has)"""
        buf = C.create_string_buffer(128)
        rc = self.lib.mgpu_comm_unique_id(buf)
        if rc:
            raise GpuError("mgpu_comm_unique_id: %d %s" % (rc, self.lib.mgpu_last_error().decode()))
        return buf.raw

    def enable_gather(self, comm_id, rank, world, queries_per_batch, k_top, num_facets, fmax, num_aggs, nthreads=None):
        """from now on run_pipeline / run_pipeline_resolved end every batch with ONE all-gather of its result slots over
        RCCL, issued by the pipeline's own threads (mplan.c:pipe_gather), and merge this rank's share of the queries:
        the returned hit sums are per rank"""
        pipe = self._pipeline(nthreads)
        pairs = queries_per_batch * self.index.num_shards
        slot_b = self.lib.m_slot_bytes(k_top, num_facets, fmax, num_aggs)
        comm = C.c_void_p()
        idb = C.create_string_buffer(bytes(comm_id), 128)
        rc = self.lib.mgpu_comm_create(self.device, idb, rank, world, self._pipe_n, pairs * slot_b, C.byref(comm))
        if rc:
            raise GpuError("mgpu_comm_create: %d %s" % (rc, self.lib.mgpu_last_error().decode()))
        self._comm = comm
        rc = self.lib.msq_pipeline_set_gather(pipe, comm, pairs, k_top, num_facets, fmax, num_aggs)
        if rc:
            raise GpuError("msq_pipeline_set_gather: %d %s" % (rc, self.lib.msq_pipeline_error(pipe).decode()))

    def close(self):
        if self._pipe is not None:
            self.lib.msq_pipeline_destroy(self._pipe)
            self._pipe = None
        if getattr(self, "_comm", None) is not None:
            self.lib.mgpu_comm_destroy(self._comm)
            self._comm = None
        for m in self.mirrors:
            self.lib.mmirror_destroy(m)
        self.mirrors = []
        for ctx in self.ctxs:
            self.lib.mgpu_ctx_destroy(ctx)
        self.ctxs = []
        self.ctx = None

    def set_profiling(self, on):
        for ctx in self.ctxs:
            self.lib.mgpu_ctx_set_profiling(ctx, 1 if on else 0)

    def execute_raw(self, pqs, capture=0):
        """returns a ctypes SqResult array of len(pqs) * num_shards (caller frees with free_raw)"""
        ns = self.index.num_shards
        nq = len(pqs)
        qarr = (C.POINTER(ffi.Query) * nq)(*[p.q for p in pqs])
        outs = (ffi.SqResult * (nq * ns))()
        rc = self.lib.msq_execute_batch(self.ctx, self._marr, ns, qarr, nq, capture, outs)
        if rc:
            self.free_raw(outs)
            raise GpuError("msq_execute_batch: %d %s" % (rc, self.lib.mgpu_last_error().decode()))
        return outs

    def prepare(self, pqs, capture=0, device_lookup=False):
        """term lookups + filter leaf resolution for a batch; returns a handle for execute_prepared.
        The parsed queries must stay alive as long as the handle.  device_lookup: the typo walks of the term lookup
        run on the GPU (msq_prepare_batch_dev, row f-1) instead of on the host trie."""
        ns = self.index.num_shards
        nq = len(pqs)
        qarr = (C.POINTER(ffi.Query) * nq)(*[p.q for p in pqs])
        h = C.c_void_p()
        if device_lookup:
            rc = self.lib.msq_prepare_batch_dev(self.ctx, self._marr, ns, qarr, nq, capture, C.byref(h))
        else:
            rc = self.lib.msq_prepare_batch(self._marr, ns, qarr, nq, capture, C.byref(h))
        if rc:
            raise GpuError("msq_prepare_batch: %d %s" % (rc, self.lib.mgpu_last_error().decode()))
        return (h, nq, pqs)

    def execute_prepared(self, prep, slot=0):
        h, nq, _ = prep
        outs = (ffi.SqResult * (nq * self.index.num_shards))()
        rc = self.lib.msq_execute_prepared(self.ctxs[slot], h, outs)
        if rc:
            self.free_raw(outs)
            raise GpuError("msq_execute_prepared: %d %s" % (rc, self.lib.mgpu_last_error().decode()))
        return outs

    def _pipeline(self, nthreads):
        """the native serving loop's thread pool (mplan.c:msq_pipeline_*), created on first use and kept"""
        nthreads = len(self.ctxs) if nthreads is None else nthreads
        if nthreads > len(self.ctxs):
            raise ValueError("more pipeline threads than contexts")
        if self._pipe is not None and self._pipe_n != nthreads:
            self.lib.msq_pipeline_destroy(self._pipe)
            self._pipe = None
        if self._pipe is None:
            carr = (C.c_void_p * nthreads)(*self.ctxs[:nthreads])
            h = C.c_void_p()
            rc = self.lib.msq_pipeline_create(carr, nthreads, C.byref(h))
            if rc:
                raise GpuError("msq_pipeline_create: %d" % rc)
            self._pipe, self._pipe_n = h, nthreads
        return self._pipe

    def warm_pipeline(self, preps, nthreads=None):
        """every prepared batch once through every context of the pipeline: arenas, pinned buffers and the
        histogram pools get their final size (set-up, like the mirror upload; not a measured step)"""
        pipe = self._pipeline(nthreads)
        parr = (C.c_void_p * len(preps))(*[p[0] for p in preps])
        rc = self.lib.msq_pipeline_warm(pipe, parr, len(preps))
        if rc:
            raise GpuError("msq_pipeline_warm: %d %s" % (rc, self.lib.msq_pipeline_error(pipe).decode()))

    def run_pipeline(self, preps, first, count, nthreads=None):
        """`count` steps of the native serving loop (mplan.c:msq_pipeline_run) over `nthreads` contexts (one host
        thread per two contexts), step i = execute_prepared + form_result of preps[(first + i) % len(preps)];
        returns the summed totalHits"""
        pipe = self._pipeline(nthreads)
        parr = (C.c_void_p * len(preps))(*[p[0] for p in preps])
        hits = C.c_long(0)
        rc = self.lib.msq_pipeline_run(pipe, parr, len(preps), first, count, C.byref(hits))
        if rc:
            raise GpuError("msq_pipeline_run: %d %s" % (rc, self.lib.msq_pipeline_error(pipe).decode()))
        return hits.value

    def resolve(self, pqs, device_lookup=False):
        """the DICTIONARY half of lookup_terms for a batch (msq_resolve_batch): dtrie_lookup_term per term and shard,
        dtrie_lookup_exact per word; device_lookup runs the typo walks on the GPU (row f-1).  Returns a handle for
        run_pipeline_resolved / plan_resolved; the parsed queries must stay alive as long as the handle."""
        ns = self.index.num_shards
        nq = len(pqs)
        qarr = (C.POINTER(ffi.Query) * nq)(*[p.q for p in pqs])
        h = C.c_void_p()
        rc = self.lib.msq_resolve_batch(self.ctx if device_lookup else None, self._marr, ns, qarr, nq, C.byref(h))
        if rc:
            raise GpuError("msq_resolve_batch: %d %s" % (rc, self.lib.mgpu_last_error().decode()))
        return (h, nq, pqs, qarr)

    def free_resolved(self, res):
        self.lib.msq_resolved_free(res[0])

    def plan_resolved(self, res, capture=0):
        """everything of lookup_terms after the dictionary (process_termresult, exact wids, filter leaves) for a
        resolved batch; returns a prepared handle like prepare()"""
        h = C.c_void_p(self.lib.msq_prepared_new())
        rc = self.lib.msq_plan_resolved(res[0], capture, h)
        if rc:
            self.lib.msq_prepared_free(h)
            raise GpuError("msq_plan_resolved: %d %s" % (rc, self.lib.mgpu_last_error().decode()))
        return (h, res[1], res[2])

    def warm_pipeline_resolved(self, ress, nthreads=None):
        pipe = self._pipeline(nthreads)
        parr = (C.c_void_p * len(ress))(*[r[0] for r in ress])
        rc = self.lib.msq_pipeline_warm_resolved(pipe, parr, len(ress))
        if rc:
            raise GpuError("msq_pipeline_warm_resolved: %d %s" % (rc, self.lib.msq_pipeline_error(pipe).decode()))

    def run_pipeline_resolved(self, ress, first, count, nthreads=None):
        """the native serving loop with the plans built inside it: step i = msq_plan_resolved + mgpu_execute_begin /
        _end + form_result of ress[(first + i) % len(ress)], all in the pipeline's own threads (two contexts per
        thread).  Returns (summed totalHits, host ms spent building plans summed over the threads)."""
        pipe = self._pipeline(nthreads)
        parr = (C.c_void_p * len(ress))(*[r[0] for r in ress])
        hits = C.c_long(0)
        rc = self.lib.msq_pipeline_run_resolved(pipe, parr, len(ress), first, count, C.byref(hits))
        if rc:
            raise GpuError("msq_pipeline_run_resolved: %d %s" % (rc, self.lib.msq_pipeline_error(pipe).decode()))
        return hits.value, self.lib.msq_pipeline_plan_ms(pipe)

    def run_pipeline_queries(self, pqs, batch, first, count, nthreads=None, warm=False):
        """the native serving loop from the queries themselves: step i = the WHOLE of lookup_terms for queries
        [(first + i) % nb * batch, + batch) -- dictionary lookups with the typo walks on the device (mgpu_lookup_terms on
        the step's own context), exact descents on the host, process_termresult -- then execute + form_result, several
        batches in flight.  Returns (summed totalHits, host ms in the dictionary half, host ms building plans)."""
        pipe = self._pipeline(nthreads)
        nq = len(pqs) // batch * batch
        qarr = (C.POINTER(ffi.Query) * nq)(*[p.q for p in pqs[:nq]])
        hits = C.c_long(0)
        rc = self.lib.msq_pipeline_run_queries(pipe, self._marr, self.index.num_shards, qarr, nq, batch, first, count, 1 if warm else 0, C.byref(hits))
        if rc:
            raise GpuError("msq_pipeline_run_queries: %d %s" % (rc, self.lib.msq_pipeline_error(pipe).decode()))
        return hits.value, self.lib.msq_pipeline_lookup_ms(pipe), self.lib.msq_pipeline_plan_ms(pipe)

    def free_prepared(self, prep):
        self.lib.msq_prepared_free(prep[0])

    def free_raw(self, outs):
        self.lib.m_sqresult_free_batch(outs, len(outs))

    def search(self, pqs, capture=0, device_lookup=False, device_highlight=False):
        """list of parsed queries -> list of assembled result dicts.  device_highlight: the hits' documents are read from
        their stored (zlib) values and the strings of all pages of the batch are highlighted with one device launch"""
        single = not isinstance(pqs, (list, tuple))
        if single:
            pqs = [pqs]
        ns = self.index.num_shards
        if device_lookup:
            prep = self.prepare(pqs, capture, device_lookup=True)
            try:
                outs = self.execute_prepared(prep)
            finally:
                self.free_prepared(prep)
        else:
            outs = self.execute_raw(pqs, capture)
        res = []
        page_batch = hits.PageBatch(self.lib, self.ctx) if device_highlight else None
        for i, pq in enumerate(pqs):
            sub = (ffi.SqResult * ns).from_address(C.addressof(outs) + i * ns * C.sizeof(ffi.SqResult))
            r = assemble(self.index, pq, sub, page_batch=page_batch)
            if capture:
                r["_shards"] = shard_results_to_py(self.index, pq, sub)
            res.append(r)
        self.free_raw(outs)
        if page_batch is not None:
            page_batch.flush()
        return res[0] if single else res

    def timing(self):
        tot = C.c_float()
        st = (C.c_float * 8)()
        self.lib.mgpu_last_timing(self.ctx, C.byref(tot), st)
        by = (C.c_uint64 * 8)()
        self.lib.mgpu_last_bytes(self.ctx, by)
        hm = (C.c_float * 3)()
        self.lib.mgpu_last_host_timing(self.ctx, hm)
        return {"total_ms": tot.value, "host_ms": {"lower": hm[0], "launch+wait": hm[1], "decode": hm[2]},
                "stages": {self.lib.mgpu_stage_name(i).decode(): {"ms": st[i], "bytes": by[i]} for i in range(8)}}
