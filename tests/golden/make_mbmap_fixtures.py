#!/usr/bin/env python3
"""Generates tests/golden/mbmap_cases.json: postings in the reference's stored form (mbmap.c:183-283).

Run in the build container only (needs oracle/_ref/libmarlin_ref.so = the reference's own bmap.c / cont.c compiled
unmodified by `make ref`).  Every set is built with the reference's bmap_add; the container buffers are dumped exactly
as mbmap_save writes them (cont buffer, `len = (card > CUTOFF ? CUTOFF : buffer[1]) + 2` u16), the header value is
[num_c][cid...] as mbmap_save writes it, and the expected docids are what the reference's bmap_iterate yields.
The output is data (bytes and numbers), not source."""
import base64
import ctypes as C
import json
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmarlin_ref.so"))


class Cont(C.Structure):
    _fields_ = [("buffer", C.POINTER(C.c_uint16))]


class Bmap(C.Structure):
    _fields_ = [("c", C.POINTER(Cont)), ("num_c", C.c_uint16), ("flags", C.c_uint8)]


ref.bmap_new.restype = C.POINTER(Bmap)
ref.bmap_add.argtypes = [C.POINTER(Bmap), C.c_uint32]
ref.bmap_free.argtypes = [C.POINTER(Bmap)]
ref.bmap_cardinality.restype = C.c_uint32
ref.bmap_cardinality.argtypes = [C.POINTER(Bmap)]
ITER = C.CFUNCTYPE(None, C.c_uint32, C.c_void_p)
ref.bmap_iterate.argtypes = [C.POINTER(Bmap), ITER, C.c_void_p]
ref.cont_cardinality.restype = C.c_uint32
ref.cont_cardinality.argtypes = [C.POINTER(Cont)]

CUTOFF = 4096


def dump(name, docids):
    b = ref.bmap_new()
    for d in docids:
        ref.bmap_add(b, d)
    got = []
    ref.bmap_iterate(b, ITER(lambda v, _: got.append(v)), None)
    assert got == sorted(set(docids)), name
    nc = b.contents.num_c
    # expected docids: [start, count] runs when that is the shorter form, else the plain list
    runs = []
    for d in got:
        if runs and runs[-1][0] + runs[-1][1] == d:
            runs[-1][1] += 1
        else:
            runs.append([d, 1])
    case = {"name": name, "cardinality": ref.bmap_cardinality(b), "header": None, "containers": []}
    if 2 * len(runs) < len(got):
        case["runs"] = runs
    else:
        case["docids"] = got
    if nc:                                     # mbmap_save deletes the key of an empty set: no header value
        cids = []
        for i in range(nc):
            c = b.contents.c[i]
            card = ref.cont_cardinality(C.byref(c))
            n16 = (CUTOFF if card > CUTOFF else c.buffer[1]) + 2
            raw = bytes(C.cast(c.buffer, C.POINTER(C.c_uint8 * (2 * n16))).contents)
            cids.append(c.buffer[0])
            case["containers"].append(base64.b64encode(raw).decode())
        hdr = (C.c_uint16 * (nc + 1))(nc, *cids)
        case["header"] = base64.b64encode(bytes(hdr)).decode()
        case["cids"] = cids
    ref.bmap_free(b)
    return case


def main():
    rng = random.Random(20251111)
    cases = [
        dump("empty", []),
        dump("one", [1]),
        dump("array_small", [1, 5, 63, 64, 4095, 65535]),
        dump("array_at_cutoff_4096", list(range(7, 7 + 2 * 4096, 2))),            # card == CUTOFF stays an array
        dump("bitset_4097", list(range(3, 3 + 3 * 4097, 3))),                     # one more: becomes a bitset
        dump("full_container", list(range(65536, 131072))),                      # card wraps to 0, payload[0] = 0xFFFF
        dump("straddle", [65534, 65535, 65536, 65537, 131071, 131072, 196607, 196608, 262143]),
        dump("mixed_sparse_cids", [9] + list(range(2 * 65536, 2 * 65536 + 5000)) + list(range(7 * 65536, 8 * 65536)) + [9 * 65536 + 65535]),
        dump("random_sparse", rng.sample(range(1, 600000), 3000)),
        dump("random_dense_one_chunk", rng.sample(range(65536 * 3, 65536 * 4), 9000)),
        dump("all_docs_1_to_300000", list(range(1, 300001))),
    ]
    out = os.path.join(ROOT, "tests", "golden", "mbmap_cases.json")
    with open(out, "w") as f:
        json.dump({"source": "reference bmap.c/cont.c via oracle/_ref (make ref), tests/golden/make_mbmap_fixtures.py",
                   "cases": cases}, f, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes;", [(c["name"], c["cardinality"], len(c["containers"])) for c in cases])


if __name__ == "__main__":
    main()
