"""Row (b), the drop-in boundary.

1. shim/squery_gpu.c -- the replacement for the reference's squery.c -- compiles against the reference's own headers
   without a warning and exports execute_squery / sqresult_free (build container only: needs /root/reference).
2. The host side it relies on (marlin_amd/host/mbridge.c: a shard source filled record by record from a foreign store;
   mplan_from_terms: a plan from term results the caller looked up) reproduces a shard exactly: every stored set of a
   data_gen shard is re-encoded as mbmap stores it (header + containers), fed through the bridge, and the rebuilt
   shard answers like the original -- set by set on the CPU, query by query on the GPU."""
import ctypes as C
import os
import subprocess

import pytest

from marlin_amd import GpuEngine, Index, ffi
from util import ROOT, _same, datagen_index, mbmap_encode

SHIM_OBJ = os.path.join(ROOT, "shim", "_build", "squery_gpu.o")
HAVE_REF = os.path.isdir("/root/reference/src/main")


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference's headers (build container only)")
def test_shim_compiles_against_reference_headers_and_exports_the_seam():
    r = subprocess.run(["make", "-C", ROOT, "-B", "shim"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]                  # -Wall -Wextra -Werror, plus the _Static_asserts on layouts
    assert "warning" not in r.stdout
    nm = subprocess.run(["nm", SHIM_OBJ], stdout=subprocess.PIPE, text=True).stdout
    defined = {l.split()[-1] for l in nm.splitlines() if " T " in l}
    assert {"execute_squery", "sqresult_free", "mgpu_shim_execute_squeries", "mgpu_shim_attach_shard", "mgpu_shim_detach_shard",
            "mgpu_shim_refresh_shard"} <= defined
    undefined = {l.split()[-1] for l in nm.splitlines() if " U " in l}
    # host work that stays with the reference, GPU work that goes through the host library; no oracle, no CPU ranking
    assert {"dtrie_lookup_term", "dtrie_lookup_exact", "mplan_from_terms", "msq_execute_plans", "mmirror_create",
            "mgpu_dict_create", "mgpu_lookup_terms"} <= undefined
    assert not {u for u in undefined if u.startswith("orc_") or u in ("perform_ranking", "filter_apply", "rank_sort")}
    # the replacement query.c: the reference's own file with another execute_query (shim/query_gpu.c)
    nm2 = subprocess.run(["nm", os.path.join(ROOT, "shim", "_build", "query_gpu.o")], stdout=subprocess.PIPE, text=True).stdout
    defined2 = {l.split()[-1] for l in nm2.splitlines() if " T " in l}
    assert {"execute_query", "mgpu_shim_execute_queries", "generate_query_terms", "query_new", "query_free"} <= defined2
    assert "mgpu_shim_execute_squeries" in {l.split()[-1] for l in nm2.splitlines() if " U " in l}


def clone_through_bridge(lib, shard):
    """every stored set of `shard` (an mshard_t*) re-encoded in the reference's stored form and fed to the bridge"""
    s = C.cast(shard, C.POINTER(ffi.ShardFull)).contents
    mb = lib.mbr_new(s.num_numbers, s.num_bools, s.num_facets, s.word_count, s.tword_count)
    assert mb

    def feed(kind, ident, prio, members):
        if not members:
            return                                              # mbmap_save deletes the key of an empty set
        header, conts = mbmap_encode(members)
        m, keep = ffi.make_mbmap(header, conts)
        assert lib.mbr_add_posting(mb, kind, ident, prio, C.byref(m)) == 0, (kind, ident)

    MBR_WID, MBR_TWID, MBR_TWID_WIDS, MBR_FACET, MBR_BOOL = range(5)
    for w in range(1, s.word_count + 1):
        feed(MBR_WID, w, 0, s.wp_doc[s.wp_off[w]:s.wp_off[w + 1]])
    for t in range(1, s.tword_count + 1):
        feed(MBR_TWID, t, 0, s.tp_doc[s.tp_off[t]:s.tp_off[t + 1]])
        feed(MBR_TWID_WIDS, t, 0, sorted(s.tw_wid[s.tw_off[t]:s.tw_off[t + 1]]))
    for f in range(s.num_facets):
        for k in range(s.nfid[f]):
            feed(MBR_FACET, s.fid[f][k], f, s.fp_doc[f][s.fp_off[f][k]:s.fp_off[f][k + 1]])
    for i in range(s.num_bools):
        for v in (0, 1):
            feed(MBR_BOOL, v, i, s.bp_doc[i][v][:s.bp_n[i][v]])
    for d in range(1, s.max_docid + 1):
        a, b = s.dd_off[d], s.dd_off[d + 1]
        if b > a:
            assert lib.mbr_add_docdata(mb, d, C.cast(C.addressof(s.dd.contents) + a, C.c_void_p), b - a) == 0
    for i in range(s.num_numbers):
        for k in range(s.np_n[i]):
            assert lib.mbr_add_num(mb, i, s.np_val[i][k], s.np_doc[i][k]) == 0
    used = (C.c_uint32 * s.max_docid)(*range(1, s.max_docid + 1))
    assert lib.mbr_set_used(mb, used, s.max_docid) == 0
    out = lib.mbr_finish(mb)
    assert out
    return out


def same_shard(a, b):
    """the stored sets of two mshard_t, field by field"""
    assert (a.max_docid, a.num_numbers, a.num_bools, a.num_facets, a.word_count, a.tword_count) == \
           (b.max_docid, b.num_numbers, b.num_bools, b.num_facets, b.word_count, b.tword_count)
    W, T = a.word_count, a.tword_count
    assert a.wp_off[:W + 2] == b.wp_off[:W + 2] and a.wp_doc[:a.wp_off[W + 1]] == b.wp_doc[:b.wp_off[W + 1]]
    assert a.tp_off[:T + 2] == b.tp_off[:T + 2] and a.tp_doc[:a.tp_off[T + 1]] == b.tp_doc[:b.tp_off[T + 1]]
    assert a.tw_off[:T + 2] == b.tw_off[:T + 2]
    for t in range(1, T + 1):                                   # the original keeps insertion order, the bitmap sorts
        assert sorted(a.tw_wid[a.tw_off[t]:a.tw_off[t + 1]]) == b.tw_wid[b.tw_off[t]:b.tw_off[t + 1]]
    for f in range(a.num_facets):
        n = a.nfid[f]
        assert n == b.nfid[f] and a.fid[f][:n] == b.fid[f][:n] and a.fp_off[f][:n + 1] == b.fp_off[f][:n + 1]
        assert a.fp_doc[f][:a.fp_off[f][n]] == b.fp_doc[f][:b.fp_off[f][n]]
    for i in range(a.num_bools):
        for v in (0, 1):
            assert a.bp_n[i][v] == b.bp_n[i][v] and a.bp_doc[i][v][:a.bp_n[i][v]] == b.bp_doc[i][v][:b.bp_n[i][v]]
    for i in range(a.num_numbers):
        n = a.np_n[i]
        assert n == b.np_n[i] and a.np_val[i][:n] == b.np_val[i][:n] and a.np_doc[i][:n] == b.np_doc[i][:n]
    M = a.max_docid
    assert a.dd_off[:M + 2] == b.dd_off[:M + 2] and bytes(a.dd[:a.dd_off[M + 1]]) == bytes(b.dd[:b.dd_off[M + 1]])


def test_bridge_rebuilds_a_shard_from_its_stored_form():
    lib = ffi.load()
    ix, _ = datagen_index(3000)
    try:
        clone = lib.mbr_finish and clone_through_bridge(lib, ix.shards[0])
        try:
            a = C.cast(ix.shards[0], C.POINTER(ffi.ShardFull)).contents
            b = C.cast(clone, C.POINTER(ffi.ShardFull)).contents
            same_shard(a, b)
            assert b.has_used == 1 and b.used_n == a.max_docid and not b.dict
        finally:
            lib.mshard_free(clone)
        # values in the wrong order / ids out of range are refused, not stored
        mb = lib.mbr_new(1, 0, 1, 5, 5)
        assert lib.mbr_add_num(mb, 0, 5.0, 7) == 0 and lib.mbr_add_num(mb, 0, 6.0, 8) == -2 and lib.mbr_add_num(mb, 0, 5.0, 7) == -2
        m, keep = ffi.make_mbmap(*mbmap_encode([1, 2, 3]))
        assert lib.mbr_add_posting(mb, 0, 6, 0, C.byref(m)) == -2 and lib.mbr_add_posting(mb, 3, 99, 1, C.byref(m)) == -2
        assert lib.mbr_add_posting(mb, 0, 5, 0, C.byref(m)) == 0
        lib.mbr_free(mb)
    finally:
        ix.close()


@pytest.mark.gpu
def test_bridged_shard_answers_like_the_original_on_the_gpu():
    """the whole path the shim takes after its trie walk: bridge -> mmirror_create -> plan -> GPU -> result, against the
    engine on the original shard (which the rest of the suite pins to the oracle)"""
    lib = ffi.load()
    ix, _ = datagen_index(12000)
    clone = clone_through_bridge(lib, ix.shards[0])
    a = C.cast(ix.shards[0], C.POINTER(ffi.ShardFull)).contents
    b = C.cast(clone, C.POINTER(ffi.ShardFull)).contents
    b.dict = a.dict                                     # the caller's dictionary (the shim uses the reference's dtrie)
    ix2 = Index.from_shards({"hitsPerPage": 25}, [clone], ["str", "strlist", "facet", "facetlist", "spell"],
                            ["id", "num", "numlist"], ["bool"], ["facet", "facetlist", "facetonly"])
    ix2.fields["numlist"].type = "numlist"
    eng, eng2 = GpuEngine(ix), GpuEngine(ix2)
    try:
        qs = [{"q": "aaaaaab bbbbbbc", "rankBy": {"num": "desc"}},
              {"q": "aaaaaaa", "filter": {"bool": True, "num": {"$gt": 100, "$lte": 9000}}},
              {"q": "", "filter": {"facet": {"$ne": "zzz"}}, "sortBy": {"id": "asc"}, "hitsPerPage": 7, "page": 3},
              {"q": "aaaaaaa bbbbbbb ccccccc", "aggs": {"s": {"stats": {"field": "num"}}, "c": {"cardinality": {"field": "facetlist"}}}},
              {"q": "eeeeeed", "maxFacetResults": 3}, {"q": ""}]
        for q in qs:
            g1 = eng.search(ix.query(dict(q, fullScan=True)), capture=3)
            g2 = eng2.search(ix2.query(dict(q, fullScan=True)), capture=3)
            for g in (g1, g2):
                for lst in g["facets"].values():
                    for e in lst:
                        e.pop("key")                            # facetid2str stays in the caller's store
                g.pop("hits")
            assert _same(g1, g2), q
            assert g1["totalHits"] > 0, q
    finally:
        eng.close(); eng2.close()
        b.dict = None
        ix2.close(); ix.close()
