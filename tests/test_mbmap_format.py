"""a10: the reference's stored posting format (mbmap.c:183-283 header + cont.c containers) read as stored.

Fixtures: tests/golden/mbmap_cases.json, written by tests/golden/make_mbmap_fixtures.py from the reference's own
bmap.c / cont.c (bmap_add builds the sets, the buffers are dumped as mbmap_save writes them, bmap_iterate gives the
expected docids).  CPU part: mgpu_mbmap_decode against every case + malformed values.  GPU part: a mirror built from
the stored form (mgpu_shard_desc_t.post_mbmap) answers mgpu_posting_op like set algebra on the expected docids."""
import base64
import ctypes as C
import json
import os

import numpy as np
import pytest

from marlin_amd import ffi
from util import GOLDEN

MGPU_E_ARG, MGPU_E_LIMIT = -2, -3


def load_cases():
    with open(os.path.join(GOLDEN, "mbmap_cases.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        if "runs" in c:
            c["docids"] = [d for s, n in c["runs"] for d in range(s, s + n)]
        c["header_b"] = base64.b64decode(c["header"]) if c["header"] else None
        c["conts_b"] = [base64.b64decode(x) for x in c["containers"]]
        assert len(c["docids"]) == c["cardinality"]
    return cases


def decode(lib, header, conts, cap=None):
    m, keep = ffi.make_mbmap(header, conts)
    n = C.c_uint64()
    rc = lib.mgpu_mbmap_decode(C.byref(m), None, 0, C.byref(n))
    if rc not in (0, MGPU_E_LIMIT):
        return rc, None
    cap = n.value if cap is None else cap
    out = (C.c_uint32 * max(cap, 1))()
    rc = lib.mgpu_mbmap_decode(C.byref(m), out, cap, C.byref(n))
    return rc, list(out[:min(cap, n.value)])


def test_decode_matches_reference_iteration():
    lib = ffi.load()
    cases = load_cases()
    kinds = set()
    for c in cases:
        rc, got = decode(lib, c["header_b"], c["conts_b"])
        assert rc == 0, (c["name"], lib.mgpu_last_error())
        assert got == c["docids"], c["name"]
        for raw in c["conts_b"]:
            card = int.from_bytes(raw[2:4], "little")
            kinds.add("full" if card == 0 and len(raw) > 4 and raw[4:6] == b"\xff\xff" else "array" if len(raw) == 4 + 2 * card else "bitset")
    assert kinds == {"array", "bitset", "full"}            # every container kind of cont.c is in the fixtures
    assert any(c["header_b"] is None for c in cases)       # and the absent key


def test_python_encoder_writes_the_reference_bytes():
    """tests/util.py:mbmap_encode (used to feed the bridge in test_bridge.py) against the reference-made buffers"""
    from util import mbmap_encode
    for c in load_cases():
        header, conts = mbmap_encode(c["docids"])
        assert header == c["header_b"] and conts == c["conts_b"], c["name"]


def test_malformed_values_are_refused():
    lib = ffi.load()
    c = next(x for x in load_cases() if x["name"] == "straddle")
    hdr, conts = c["header_b"], c["conts_b"]
    assert decode(lib, hdr[:1], conts)[0] == MGPU_E_ARG                         # header shorter than its count
    assert decode(lib, hdr[:4], conts)[0] == MGPU_E_ARG                         # ... than its container list
    assert decode(lib, hdr, [conts[0][:3]] + conts[1:])[0] == MGPU_E_ARG        # container shorter than its header
    assert decode(lib, hdr, [conts[0][:-2]] + conts[1:])[0] == MGPU_E_ARG       # array shorter than its cardinality
    assert decode(lib, hdr, [conts[1]] + conts[1:])[0] == MGPU_E_ARG            # container id != key
    swapped = hdr[:2] + hdr[4:6] + hdr[2:4] + hdr[6:]
    assert decode(lib, swapped, conts)[0] == MGPU_E_ARG                         # container ids not ascending
    b = next(x for x in load_cases() if x["name"] == "bitset_4097")
    assert decode(lib, b["header_b"], [b["conts_b"][0][:5000]])[0] == MGPU_E_ARG
    bad = bytearray(b["conts_b"][0]); bad[100] ^= 0x10                          # a bit that the cardinality does not count
    assert decode(lib, b["header_b"], [bytes(bad)])[0] == MGPU_E_ARG
    # a container whose mdb_get failed is skipped, like the reference's unusable container
    rc, got = decode(lib, hdr, [None] + conts[1:])
    assert rc == 0 and got == [d for d in c["docids"] if d >= 65536]
    # short output buffer: the size is still reported
    rc, got = decode(lib, hdr, conts, cap=3)
    assert rc == MGPU_E_LIMIT and got == c["docids"][:3]


@pytest.mark.gpu
def test_mirror_from_stored_postings_answers_like_set_algebra():
    lib = ffi.load()
    cases = load_cases()
    allc = next(i for i, c in enumerate(cases) if c["name"].startswith("all_docs"))
    order = [allc] + [i for i in range(len(cases)) if i != allc]                # pid 0 = every used docid
    max_docid = max(max(c["docids"], default=0) for c in cases)
    keep, arr = [], (ffi.Mbmap * len(order))()
    for p, i in enumerate(order):
        m, k = ffi.make_mbmap(cases[i]["header_b"], cases[i]["conts_b"])
        arr[p] = m; keep.append(k)
    d = ffi.ShardDesc()
    d.max_docid = max_docid; d.num_numbers = 0; d.num_facets = 0; d.num_postings = len(order)
    d.post_mbmap = arr
    dd_off = (C.c_uint64 * (max_docid + 2))()                                    # no docdata: every blob is empty
    dd = (C.c_uint8 * 8)()
    d.dd_off = dd_off; d.dd = dd
    shard, ctx = C.c_void_p(), C.c_void_p()
    assert lib.mgpu_shard_create(0, C.byref(d), C.byref(shard)) == 0, lib.mgpu_last_error()
    assert lib.mgpu_ctx_create(0, 0, C.byref(ctx)) == 0, lib.mgpu_last_error()
    try:
        sets = [set(cases[i]["docids"]) for i in order]
        nwords = max_docid // 64 + 1
        for a in range(len(order)):
            for b in range(len(order)):
                for op, want in ((0, sets[a] & sets[b]), (1, sets[a] | sets[b]), (2, sets[a] - sets[b])):
                    if op and (a + b) % 3:                                       # every pair ANDed, a third of them ORed / ANDNOTed
                        continue
                    out = (C.c_uint64 * nwords)(); card = C.c_uint64()
                    assert lib.mgpu_posting_op(ctx, shard, op, a, b, out, nwords, C.byref(card)) == 0, lib.mgpu_last_error()
                    assert card.value == len(want), (cases[order[a]]["name"], cases[order[b]]["name"], op)
                    got = np.flatnonzero(np.unpackbits(np.frombuffer(bytes(out), dtype=np.uint8), bitorder="little"))
                    assert np.array_equal(got, np.array(sorted(want), dtype=got.dtype)), (cases[order[a]]["name"], cases[order[b]]["name"], op)
    finally:
        lib.mgpu_ctx_destroy(ctx); lib.mgpu_shard_destroy(shard)
