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
    def __init__(self, lib, dist, world, rank, nq, local_shards, k_top, num_facets, fmax, num_aggs, device=None, nslots=1,
                 cardinality=False):
        """nslots: buffer sets, one per batch in flight (a host thread packs and unpacks in its own set, so only
        the enqueue of the collective itself has to happen in step order).
        cardinality: some query of the stream has a cardinality aggregation.  Its distinct-id sets are variable-length
        and cannot sit in the fixed slots (the reference ORs the shards' id bitmaps, metric-aggs.c:47-73): every gather
        is then followed by a sizes all_gather and one all_gather of the id buffers padded to the largest."""
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
        self.cardinality = bool(cardinality)
        self._card = {}

    def my_queries(self):
        """indices of the batch's queries this rank merges when every rank takes a share"""
        return list(range(self.rank, self.nq, self.world))

    # ---- the three steps of a gather; only issue() has to run in the same order on every rank
    def pack(self, outs, slot=0):
        """outs: ctypes SqResult array [query][local shard] of this rank -> the slot's send buffer.  The slot's
        previous gather has been finished (finish() waits for its copies), so the buffer is free."""
        self.lib.m_slot_pack_batch(outs, self.nq * self.local, *self.shape, self.bufs[slot]["host_send"].data_ptr())
        if self.cardinality:
            n = int(self.lib.m_slot_card_count(outs, self.nq * self.local, self.shape[3]))
            ids = self.torch.zeros(max(n, 1), dtype=self.torch.int32)
            if n:
                self.lib.m_slot_card_pack(outs, self.nq * self.local, self.shape[3], ids.data_ptr())
            self._card[slot] = (n, ids)

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
        if self.cardinality:
            # the variable-length part: sizes first, then the id buffers padded to the largest (small next to the slots)
            torch = self.torch
            n, ids = self._card[slot]
            dev = self.device if self.on_gpu else "cpu"
            sizes = torch.zeros(self.world, dtype=torch.int64, device=dev)
            self.dist.all_gather_into_tensor(sizes, torch.tensor([n], dtype=torch.int64, device=dev))
            sizes = sizes.cpu()
            mx = max(1, int(sizes.max()))
            send = torch.zeros(mx, dtype=torch.int32, device=dev)
            send[:max(n, 1)] = ids.to(dev)[:max(n, 1)]
            recv = torch.zeros(self.world * mx, dtype=torch.int32, device=dev)
            self.dist.all_gather_into_tensor(recv, send)
            self._card[slot] = (sizes.tolist(), mx, recv.cpu())

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
            self._attach_card(slot, sq, self.rank, self.world)
            return sq
        sq = (ffi.SqResult * (self.nq * total))()
        lib.m_slot_unpack_gathered(b["host_recv"].data_ptr(), self.world, self.nq, self.local, *self.shape, sq)
        self._attach_card(slot, sq, 0, 1)
        return sq

    def _attach_card(self, slot, sq, q_first, q_step):
        if not self.cardinality:
            return
        sizes, mx, recv = self._card[slot]
        slots = self.bufs[slot]["host_recv"]
        per_rank = self.nq * self.local * self.slot_b
        for r in range(self.world):
            rc = self.lib.m_slot_card_attach(slots.data_ptr() + r * per_rank, recv.data_ptr() + 4 * r * mx, sizes[r], r, self.world,
                                             self.nq, self.local, *self.shape, q_first, q_step, sq)
            if rc:
                raise RuntimeError("cardinality id buffer of rank %d is shorter than its slots say" % r)

    def gather(self, outs, share=False, slot=0):
        """pack + issue + finish in one call"""
        self.pack(outs, slot)
        self.issue(slot)
        return self.finish(slot, share)

    def free(self, sq):
        self.lib.m_sqresult_free_batch(sq, len(sq))
