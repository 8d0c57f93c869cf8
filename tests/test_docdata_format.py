"""The formats on the index side of the seam, pinned on bytes the reference itself wrote: docid2data blobs (sindex.c:273-326,
409-496) and num_dbi key order (sindex.c:61-64, 527-535, 1116-1118).  tests/golden/docdata_cases.json holds inputs and
the bytes LMDB held after the reference's own gen_wordpos_data / store_docdata / index_number ran on them
(oracle/ref_seam.c, tests/golden/make_docdata_fixtures.py).  Here: marlin_amd/host/mshard.c writes the same bytes from
the same inputs, sorts numbers the same way, and the bridge that feeds the GPU mirror (mbridge.c) takes the reference's
bytes as they are and rebuilds the same shard."""
import ctypes as C
import json
import os

import pytest

from marlin_amd import ffi
from util import GOLDEN, mbmap_encode

CASES = json.load(open(os.path.join(GOLDEN, "docdata_cases.json")))["cases"]


def build_with_own_writer(lib, case):
    b = lib.msb_new(1, case["num_numbers"], 0, case["num_facets"])
    docs = {d["docid"]: d for d in case["docs"]}
    for docid in range(1, max(docs) + 1):
        assert lib.msb_begin_doc(b) == docid
        d = docs.get(docid)
        if d:
            for f, v in d["nums"]:
                lib.msb_add_number(b, f, v, 1)
            for f, i in d["facets"]:
                lib.msb_add_raw_facet(b, f, i)
            for w, p, pos in d["words"]:
                lib.msb_add_raw_wordpos(b, w, p, pos)
        lib.msb_end_doc(b)
    return lib.msb_finalize(b)


def blob_of(lib, shard, docid):
    n = C.c_uint32()
    p = lib.mshard_docdata(shard, docid, C.byref(n))
    return bytes(p[:n.value]) if p else b""


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_own_writer_matches_reference_bytes(ci):
    lib = ffi.load()
    case = CASES[ci]
    sh = lib.msb_finalize and build_with_own_writer(lib, case)
    try:
        for docid, hexblob in case["blobs"].items():
            want = bytes.fromhex(hexblob)
            got = blob_of(lib, sh, int(docid))
            # the own store pads every blob to 8 bytes (LMDB values are aligned by LMDB itself)
            assert got[:len(want)] == want, (ci, docid, got.hex(), hexblob)
            assert len(got) - len(want) < 8 and not any(got[len(want):]), (ci, docid)
        s = C.cast(sh, C.POINTER(ffi.ShardFull)).contents
        for f in range(case["num_numbers"]):
            got = [[s.np_val[f][k], s.np_doc[f][k]] for k in range(s.np_n[f])]
            assert got == case["num_order"][f], (ci, f)                # double_compare: values descending, docids ascending
    finally:
        lib.mshard_free(sh)


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_bridge_takes_reference_bytes(ci):
    """what the shim hands over after its LMDB walk -- the reference's blobs and num_dbi pairs in cursor order, postings in
    mbmap form -- rebuilds the shard the own writer builds from the same inputs: blobs, numeric index, word postings"""
    lib = ffi.load()
    case = CASES[ci]
    own = build_with_own_writer(lib, case)
    so = C.cast(own, C.POINTER(ffi.ShardFull)).contents
    mb = lib.mbr_new(case["num_numbers"], 0, case["num_facets"], so.word_count, so.tword_count)
    try:
        post, fpost = {}, {}
        for d in case["docs"]:
            for w, _, _ in d["words"]:
                post.setdefault(w, set()).add(d["docid"])
            for f, i in d["facets"]:
                fpost.setdefault((f, i), set()).add(d["docid"])
        for w, docs in post.items():
            m, keep = ffi.make_mbmap(*mbmap_encode(sorted(docs)))
            assert lib.mbr_add_posting(mb, 0, w, 0, C.byref(m)) == 0
        for (f, i), docs in fpost.items():
            m, keep = ffi.make_mbmap(*mbmap_encode(sorted(docs)))
            assert lib.mbr_add_posting(mb, 3, i, f, C.byref(m)) == 0
        for docid, hexblob in case["blobs"].items():
            raw = bytes.fromhex(hexblob)
            buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
            assert lib.mbr_add_docdata(mb, int(docid), C.cast(buf, C.c_void_p), len(raw)) == 0
        for f in range(case["num_numbers"]):
            for v, docid in case["num_order"][f]:
                assert lib.mbr_add_num(mb, f, v, docid) == 0
        ids = sorted(int(k) for k in case["blobs"])
        used = (C.c_uint32 * len(ids))(*ids)
        assert lib.mbr_set_used(mb, used, len(ids)) == 0
        br = lib.mbr_finish(mb)
        mb = None
        assert br
        sb = C.cast(br, C.POINTER(ffi.ShardFull)).contents
        for docid, hexblob in case["blobs"].items():
            want = bytes.fromhex(hexblob)
            assert blob_of(lib, br, int(docid))[:len(want)] == want
        for f in range(case["num_numbers"]):
            assert [[sb.np_val[f][k], sb.np_doc[f][k]] for k in range(sb.np_n[f])] == case["num_order"][f]
        for w in post:
            assert list(sb.wp_doc[sb.wp_off[w]:sb.wp_off[w + 1]]) == list(so.wp_doc[so.wp_off[w]:so.wp_off[w + 1]]) == sorted(post[w])
        lib.mshard_free(br)
    finally:
        if mb:
            lib.mbr_free(mb)
        lib.mshard_free(own)
