"""Shards <-> GPUs: the cross-rank step of a sharded query.

The reference runs one execute_squery per shard on a thread pool inside one process and merges
the per-shard results in form_result (query.c:319-383).  Here a rank (= one process = one GPU)
owns a contiguous block of shards; after the GPU batch every rank packs its per-(query, shard)
results into fixed-size slots (m_slot_pack_batch), one all_gather moves them (RCCL over xGMI when
the tensors live on the GPUs, gloo on CPU in the tests) and rank 0 unpacks and merges with the
same m_form_result the single-GPU path uses.  There is no other collective on the data path.
"""
import ctypes as C

from . import ffi


class ShardGather:
    def __init__(self, lib, dist, world, rank, nq, local_shards, k_top, num_facets, fmax, num_aggs, device=None, nslots=1):
        """nslots: buffer sets, one per batch in flight (a host thread packs and unpacks in its own set, so only
        the enqueue of the collective itself has to happen in step order)"""
        import torch
        self.lib, self.dist, self.world, self.rank = lib, dist, world, rank
        self.nq, self.local = nq, local_shards
        self.shape = (k_top, num_facets, fmax, num_aggs)
        self.slot_b = lib.m_slot_bytes(*self.shape)
        n = nq * local_shards * self.slot_b
        self.on_gpu = device is not None and str(device).startswith("cuda")
        self.device = torch.device(device) if self.on_gpu else None
        self.bufs = []
        for _ in range(max(1, nslots)):
            b = {"host_send": torch.empty(n, dtype=torch.uint8), "host_recv": torch.empty(world * n, dtype=torch.uint8), "event": None}
            if self.on_gpu:
                b["host_send"] = b["host_send"].pin_memory(); b["host_recv"] = b["host_recv"].pin_memory()
                b["send"] = torch.empty(n, dtype=torch.uint8, device=device)
                b["recv"] = torch.empty(world * n, dtype=torch.uint8, device=device)
                b["event"] = torch.cuda.Event()
            self.bufs.append(b)
        self.torch = torch

    def my_queries(self):
        """indices of the batch's queries this rank merges when every rank takes a share"""
        return list(range(self.rank, self.nq, self.world))

    # ---- the three steps of a gather; only issue() has to run in the same order on every rank
    def pack(self, outs, slot=0):
        """outs: ctypes SqResult array [query][local shard] of this rank -> the slot's send buffer.  The slot's
        previous gather has been finished (finish() waits for its copies), so the buffer is free."""
        self.lib.m_slot_pack_batch(outs, self.nq * self.local, *self.shape, self.bufs[slot]["host_send"].data_ptr())

    def issue(self, slot=0):
        """enqueue copy-in, all_gather and copy-out (RCCL over xGMI when the tensors live on the GPUs)"""
        b = self.bufs[slot]
        if self.on_gpu:
            # torch's current device is per thread and the callers are worker threads: name the device
            self.torch.cuda.set_device(self.device)
            b["send"].copy_(b["host_send"], non_blocking=True)
            self.dist.all_gather_into_tensor(b["recv"], b["send"])
            b["host_recv"].copy_(b["recv"], non_blocking=True)
            b["event"].record(self.torch.cuda.current_stream(self.device))
        else:
            self.dist.all_gather_into_tensor(b["host_recv"], b["host_send"])

    def finish(self, slot=0, share=False):
        """share=False: returns, on rank 0, a SqResult array [query][global shard] (None elsewhere).
        share=True : every rank gets the [query][global shard] array of ITS share of the queries
        (my_queries()): all_gather already delivers every slot everywhere, so the merge spreads over
        the ranks with no further communication.  The caller frees the array with free()."""
        lib, b = self.lib, self.bufs[slot]
        if self.on_gpu:
            b["event"].synchronize()
        if self.rank != 0 and not share:
            return None
        total = self.world * self.local
        if share:
            mine = len(self.my_queries())
            sq = (ffi.SqResult * max(1, mine * total))()
            lib.m_slot_unpack_gathered_subset(b["host_recv"].data_ptr(), self.world, self.nq, self.local, *self.shape,
                                              self.rank, self.world, sq)
            return sq
        sq = (ffi.SqResult * (self.nq * total))()
        lib.m_slot_unpack_gathered(b["host_recv"].data_ptr(), self.world, self.nq, self.local, *self.shape, sq)
        return sq

    def gather(self, outs, share=False, slot=0):
        """pack + issue + finish in one call"""
        self.pack(outs, slot)
        self.issue(slot)
        return self.finish(slot, share)

    def free(self, sq):
        self.lib.m_sqresult_free_batch(sq, len(sq))
