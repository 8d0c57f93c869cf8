"""mgpu_posting_op_noalloc (include/marlin_gpu.h): the allocation-free variant of mgpu_posting_op, whose scratch
is a static device array.  Checked through the C ABI against mgpu_posting_op, bit for bit (result words and
cardinality), on two calls in a row -- the second on a smaller shard, so that whatever the first call left in the
shared scratch array would show.  pytest -m gpu; collected last (see the file name)."""
import ctypes as C
import random

import pytest

from marlin_amd import GpuEngine, ffi
from util import datagen_index

pytestmark = pytest.mark.gpu


def posting_len(lib, shard, wid):
    docs = C.POINTER(C.c_uint32)()
    return lib.mshard_wid_posting(shard, wid, C.byref(docs))


def run_pairs(lib, eng, ix, seed):
    mir = ffi.Mirror.from_address(eng.mirrors[0].value)
    shard = ix.shards[0]
    n_docs = C.cast(shard, C.POINTER(ffi.ShardStruct)).contents.max_docid
    nwords = n_docs // 64 + 1
    rng = random.Random(seed)
    sized = sorted(range(1, mir.W + 1), key=lambda w: -posting_len(lib, shard, w))
    picks = sized[:4] + rng.sample(sized[len(sized) // 2:], 4) + [0]        # dense rows, sparse lists, "all docs"
    checked = 0
    for a in picks:
        for b in picks:
            for op in (0, 1, 2):
                want = (C.c_uint64 * nwords)(); got = (C.c_uint64 * nwords)()
                wc, gc = C.c_uint64(), C.c_uint64()
                assert lib.mgpu_posting_op(eng.ctx, mir.g, op, a, b, want, nwords, C.byref(wc)) == 0, lib.mgpu_last_error()
                assert lib.mgpu_posting_op_noalloc(eng.ctx, mir.g, op, a, b, got, nwords, C.byref(gc)) == 0, lib.mgpu_last_error()
                assert gc.value == wc.value, (a, b, op, gc.value, wc.value)
                assert bytes(got) == bytes(want), (a, b, op)
                checked += 1
    # cardinality only (no output buffer), and the argument checks of the original
    wc, gc = C.c_uint64(), C.c_uint64()
    assert lib.mgpu_posting_op(eng.ctx, mir.g, 0, picks[0], picks[4], None, 0, C.byref(wc)) == 0
    assert lib.mgpu_posting_op_noalloc(eng.ctx, mir.g, 0, picks[0], picks[4], None, 0, C.byref(gc)) == 0
    assert gc.value == wc.value
    assert lib.mgpu_posting_op_noalloc(eng.ctx, mir.g, 0, mir.num_postings, 1, None, 0, C.byref(gc)) == -2      # MGPU_E_ARG
    small = (C.c_uint64 * 1)()
    if nwords > 1:
        assert lib.mgpu_posting_op_noalloc(eng.ctx, mir.g, 0, picks[0], picks[1], small, 1, C.byref(gc)) == -2
    return checked


def test_noalloc_matches_posting_op_twice_second_smaller():
    ix_big, _ = datagen_index(30000)
    ix_small, _ = datagen_index(3000)
    eng_big = GpuEngine(ix_big)
    eng_small = GpuEngine(ix_small)
    try:
        lib = eng_big.lib
        assert run_pairs(lib, eng_big, ix_big, 11) > 0
        # second call series: smaller rows inside the same static scratch array, which still holds the big shard's rows
        assert run_pairs(lib, eng_small, ix_small, 12) > 0
        # and the big one again after the small one
        assert run_pairs(lib, eng_big, ix_big, 13) > 0
    finally:
        eng_big.close(); eng_small.close(); ix_big.close(); ix_small.close()
