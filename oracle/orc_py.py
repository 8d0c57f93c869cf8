"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
marlin_amd/.  See oracle/orc.h for what the oracle restates and how it is pinned.
"""
import ctypes as C
import os

from marlin_amd import ffi
from marlin_amd.engine import assemble, shard_results_to_py

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def load():
    global _lib
    if _lib is None:
        ffi.load()
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liborc.so is missing: run `make oracle`")
        lib = C.CDLL(path)
        lib.orc_execute_squery.restype = C.c_int
        lib.orc_execute_squery.argtypes = [C.c_void_p, C.POINTER(ffi.Query), C.c_int, C.c_int, C.POINTER(ffi.SqResult)]
        lib.orc_lookup_time.restype = C.c_double
        lib.orc_lookup_time.argtypes = [C.c_int]
        lib.orc_bench_pool.restype = C.c_double
        lib.orc_bench_pool.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.POINTER(ffi.Query)), C.c_int, C.c_int,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        lib.orc_form_result.restype = None
        lib.orc_form_result.argtypes = [C.POINTER(ffi.Query), C.POINTER(ffi.SqResult), C.c_int, C.POINTER(ffi.Result)]
        _lib = lib
    return _lib


def execute_raw(index, pq, flags=0):
    lib = load()
    outs = (ffi.SqResult * index.num_shards)()
    for s in range(index.num_shards):
        rc = lib.orc_execute_squery(index.shards[s], pq.q, s, flags, C.byref(outs[s]))
        if rc:
            raise RuntimeError("orc_execute_squery: %d" % rc)
    return outs


def lookup_time(reset=False):
    """seconds the oracle spent in dictionary lookups since the last reset"""
    return load().orc_lookup_time(1 if reset else 0)


def search(index, pq, flags=0):
    """the oracle's answer for one parsed query, assembled like the GPU path's"""
    outs = execute_raw(index, pq, flags)
    r = assemble(index, pq, outs, form=load().orc_form_result)       # the oracle's own merge (orc_merge.c), not the product's
    if flags:
        r["_shards"] = shard_results_to_py(index, pq, outs)
    for s in range(index.num_shards):
        index.lib.m_sqresult_free(C.byref(outs[s]))
    return r


def bench_pool(index, pqs, nthreads):
    """the oracle over a thread pool, one job per (query, shard) like the reference's search_pool (query.c:435-452).
    Returns (wall seconds, lookup cpu seconds, busy cpu seconds, total hits)."""
    lib = load()
    sh = (C.c_void_p * index.num_shards)(*[s.value for s in index.shards])
    qs = (C.POINTER(ffi.Query) * len(pqs))(*[p.q for p in pqs])
    lk, busy, hits = C.c_double(), C.c_double(), C.c_uint64()
    wall = lib.orc_bench_pool(sh, index.num_shards, qs, len(pqs), nthreads, C.byref(lk), C.byref(busy), C.byref(hits))
    return wall, lk.value, busy.value, hits.value
