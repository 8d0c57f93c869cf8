"""ctypes bindings of the host library (libmarlin_host.so) and, through it, of the GPU C ABI.

Python is plumbing here: tests, bench.py and the multi-GPU launcher drive the C entry points
declared in marlin_amd/host/*.h and include/marlin_gpu.h.  The structures below mirror
marlin_amd/host/mcommon.h field for field; sizes are verified against the library at import.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

M_MAX_WORD_LEN = 127
M_MAX_QWORDS = 16
M_MAX_QTERMS = 32
M_MAX_FIELDS = 32
M_MAX_RULES = 10
M_MAX_AGGS = 16

# enums (mcommon.h)
MR_TYPO, MR_PROX, MR_POS, MR_EXACT, MR_FIELD, MR_GEO, MR_DONE, MR_COMP, MR_COMP_ASC, MR_MAX = range(10)
MF_AND, MF_OR, MF_NIN, MF_EQ, MF_NE, MF_GT, MF_GTE, MF_LT, MF_LTE, MF_RANGE, MF_ERROR = range(11)
MT_STRING, MT_NUMBER, MT_BOOLEAN, MT_OTHER = range(4)
MA_ROOT, MA_MAX, MA_MIN, MA_AVG, MA_STATS, MA_CARDINALITY, MA_ERROR = range(7)
MPREFIX_LAST, MPREFIX_ALL, MPREFIX_NONE = range(3)
MTYPO_OK, MTYPO_NONE = range(2)

RULE_NAMES = {"typos": MR_TYPO, "proximity": MR_PROX, "position": MR_POS, "exact": MR_EXACT,
              "field": MR_FIELD, "geo": MR_GEO}          # sort.c:11-18


class DocRank(C.Structure):
    _fields_ = [("compare", C.c_double), ("docid", C.c_uint32), ("distance", C.c_uint32),
                ("position", C.c_uint16), ("proximity", C.c_uint16), ("typos", C.c_uint8),
                ("exact", C.c_uint8), ("field", C.c_uint8), ("shard_id", C.c_uint8)]


class FacetCount(C.Structure):
    _fields_ = [("shard_id", C.c_uint8), ("facet_id", C.c_uint32), ("count", C.c_uint32)]


class Word(C.Structure):
    _fields_ = [("length", C.c_int), ("chars", C.c_uint32 * (M_MAX_WORD_LEN + 1))]


class Term(C.Structure):
    _fields_ = [("word", Word), ("prefix", C.c_uint8), ("typos", C.c_uint8)]


class Filter(C.Structure):
    pass


Filter._fields_ = [("type", C.c_int), ("field_type", C.c_int), ("i_priority", C.c_int),
                   ("f_priority", C.c_int), ("is_facet", C.c_int), ("facet_id", C.c_uint32),
                   ("numval", C.c_double), ("numval2", C.c_double), ("numcmp1", C.c_int),
                   ("numcmp2", C.c_int), ("num_children", C.c_int),
                   ("children", C.POINTER(C.POINTER(Filter)))]


class Agg(C.Structure):
    _fields_ = [("type", C.c_int), ("field", C.c_int), ("value", C.c_double), ("sum", C.c_double),
                ("min", C.c_double), ("max", C.c_double), ("count", C.c_int),
                ("card_ids", C.POINTER(C.c_uint32)), ("card_n", C.c_uint32)]


class QueryCfg(C.Structure):
    _fields_ = [("prefix", C.c_int), ("typos", C.c_int), ("hits_per_page", C.c_uint16),
                ("max_hits", C.c_uint16), ("max_facet_results", C.c_uint16), ("num_rules", C.c_int),
                ("rank_by", C.c_int), ("rank_sort", C.c_bool), ("rank_asc", C.c_bool),
                ("full_scan", C.c_bool), ("full_scan_threshold", C.c_uint32),
                ("rank_algo", C.c_int * M_MAX_RULES), ("facet_enabled", C.c_int * M_MAX_FIELDS)]


class Query(C.Structure):
    _fields_ = [("text", C.c_char_p), ("num_words", C.c_int), ("words", Word * M_MAX_QWORDS),
                ("num_terms", C.c_int), ("terms", Term * M_MAX_QTERMS),
                ("filter", C.POINTER(Filter)), ("num_aggs", C.c_int), ("aggs", Agg * M_MAX_AGGS),
                ("cfg", QueryCfg), ("page_num", C.c_uint16), ("rank_rule", C.c_int * M_MAX_RULES)]


class SqResult(C.Structure):
    _fields_ = [("num_hits", C.c_uint32), ("rank_count", C.c_int), ("ranks", C.POINTER(DocRank)),
                ("num_facets", C.c_int), ("rcount", C.c_int * M_MAX_FIELDS),
                ("fc", C.POINTER(FacetCount) * M_MAX_FIELDS), ("num_aggs", C.c_int),
                ("aggs", Agg * M_MAX_AGGS), ("fast_rank", C.c_bool),
                ("dbg_docid_map", C.POINTER(C.c_uint64)), ("dbg_words", C.c_uint32),
                ("dbg_all_ranks", C.POINTER(DocRank))]


class Result(C.Structure):
    _fields_ = [("total_hits", C.c_int), ("num_hits", C.c_int), ("page_s", C.c_int), ("page_e", C.c_int),
                ("total_ranks", C.c_int), ("ranks", C.POINTER(DocRank)), ("num_facets", C.c_int),
                ("fcount", C.c_int * M_MAX_FIELDS), ("fc", C.POINTER(FacetCount) * M_MAX_FIELDS),
                ("num_aggs", C.c_int), ("aggs", Agg * M_MAX_AGGS), ("full_scan", C.c_bool)]


class Mirror(C.Structure):
    """mmirror_t (marlin_amd/host/mplan.h): the posting-id scheme of one shard mirror"""
    _fields_ = [("s", C.c_void_p), ("g", C.c_void_p), ("device", C.c_int), ("W", C.c_uint32), ("T", C.c_uint32),
                ("fbase", C.c_uint32 * M_MAX_FIELDS), ("bbase", C.c_uint32), ("num_postings", C.c_uint32),
                ("gd", C.c_void_p)]


class TermLookup(C.Structure):
    """mgpu_term_lookup_t (include/marlin_gpu.h)"""
    _fields_ = [("dict", C.c_void_p), ("chars", C.POINTER(C.c_uint32)), ("len", C.c_int32), ("prefix", C.c_int32)]


class TermHit(C.Structure):
    """mgpu_term_hit_t"""
    _fields_ = [("lookup", C.c_uint32), ("wid", C.c_uint32), ("dist", C.c_uint32)]


class WidEntry(C.Structure):
    """mgpu_wid_entry_t"""
    _fields_ = [("wid", C.c_uint32), ("dist", C.c_uint8), ("mterm", C.c_uint8), ("exact_mask", C.c_uint16)]


class Fop(C.Structure):
    """mgpu_fop_t"""
    _fields_ = [("op", C.c_int32), ("a", C.c_int32), ("lo", C.c_uint64), ("hi", C.c_uint64)]


class Plan(C.Structure):
    """mgpu_plan_t (include/marlin_gpu.h)"""
    _fields_ = [("num_terms", C.c_int32), ("num_words", C.c_int32), ("term_npost", C.c_uint32 * 32),
                ("term_post", C.POINTER(C.c_uint32) * 32), ("term_nzero", C.c_uint32 * 32), ("exact_present", C.c_uint16),
                ("num_wids", C.c_uint32), ("wids", C.POINTER(WidEntry)),
                ("has_filter", C.c_int32), ("filter_null", C.c_int32), ("num_fops", C.c_int32), ("fops", C.POINTER(Fop)),
                ("hits_per_page", C.c_int32), ("page_num", C.c_int32), ("max_facet_results", C.c_int32),
                ("facet_enabled", C.c_uint32), ("rank_by", C.c_int32), ("rank_asc", C.c_int32),
                ("rules", C.c_int32 * M_MAX_RULES), ("full_scan", C.c_int32), ("full_scan_threshold", C.c_uint32),
                ("facet_pid_base", C.c_uint32 * M_MAX_FIELDS), ("num_aggs", C.c_int32),
                ("agg_type", C.c_int32 * M_MAX_AGGS), ("agg_field", C.c_int32 * M_MAX_AGGS), ("capture", C.c_int32)]


class ShardStruct(C.Structure):
    """leading fields of mshard_t that Python reads directly"""
    _fields_ = [("max_docid", C.c_uint32), ("num_strings", C.c_int), ("num_numbers", C.c_int),
                ("num_bools", C.c_int), ("num_facets", C.c_int)]


class ShardFull(C.Structure):
    """all of mshard_t (marlin_amd/host/mshard.h), for tests that walk a shard's stored sets"""
    _P32, _P64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    _fields_ = [("max_docid", C.c_uint32), ("num_strings", C.c_int), ("num_numbers", C.c_int),
                ("num_bools", C.c_int), ("num_facets", C.c_int), ("dict", C.c_void_p),
                ("dd_off", _P64), ("dd", C.POINTER(C.c_uint8)),
                ("word_count", C.c_uint32), ("wp_off", _P64), ("wp_doc", _P32),
                ("tword_count", C.c_uint32), ("tp_off", _P64), ("tp_doc", _P32), ("tw_off", _P64), ("tw_wid", _P32),
                ("nfid", C.c_uint32 * M_MAX_FIELDS), ("fid", _P32 * M_MAX_FIELDS), ("fp_off", _P64 * M_MAX_FIELDS),
                ("fp_doc", _P32 * M_MAX_FIELDS),
                ("nfstr", C.c_uint32), ("fstr_id", _P32), ("fstr", C.POINTER(C.c_char_p)),
                ("bp_n", (C.c_uint64 * 2) * M_MAX_FIELDS), ("bp_doc", (_P32 * 2) * M_MAX_FIELDS),
                ("np_n", C.c_uint64 * M_MAX_FIELDS), ("np_val", C.POINTER(C.c_double) * M_MAX_FIELDS),
                ("np_doc", _P32 * M_MAX_FIELDS),
                ("has_used", C.c_int), ("used_n", C.c_uint64), ("used_doc", _P32)]


class Mbmap(C.Structure):
    """mgpu_mbmap_t (include/marlin_gpu.h): one posting as mbmap stores it"""
    _fields_ = [("header", C.POINTER(C.c_uint8)), ("header_len", C.c_uint32),
                ("cont", C.POINTER(C.POINTER(C.c_uint8))), ("cont_len", C.POINTER(C.c_uint32))]


class ShardDesc(C.Structure):
    """mgpu_shard_desc_t (include/marlin_gpu.h)"""
    _fields_ = [("max_docid", C.c_uint32), ("num_numbers", C.c_int), ("num_facets", C.c_int),
                ("num_postings", C.c_uint32), ("post_off", C.POINTER(C.c_uint64)), ("post_doc", C.POINTER(C.c_uint32)),
                ("post_mbmap", C.POINTER(Mbmap)),
                ("dd_off", C.POINTER(C.c_uint64)), ("dd", C.POINTER(C.c_uint8)),
                ("np_n", C.c_uint64 * M_MAX_FIELDS), ("np_doc", C.POINTER(C.c_uint32) * M_MAX_FIELDS)]


def make_mbmap(header, containers):
    """bytes (or None) + list of bytes -> (Mbmap, keepalive): the ctypes buffers must outlive the struct's use"""
    keep = []
    m = Mbmap()
    if header is None:
        return m, keep
    hb = (C.c_uint8 * len(header)).from_buffer_copy(header)
    keep.append(hb)
    m.header = C.cast(hb, C.POINTER(C.c_uint8)); m.header_len = len(header)
    n = len(containers)
    ptrs = (C.POINTER(C.c_uint8) * max(n, 1))()
    lens = (C.c_uint32 * max(n, 1))()
    for i, c in enumerate(containers):
        if c is None:
            continue
        cb = (C.c_uint8 * len(c)).from_buffer_copy(c)
        keep.append(cb)
        ptrs[i] = C.cast(cb, C.POINTER(C.c_uint8)); lens[i] = len(c)
    keep += [ptrs, lens]
    m.cont = ptrs; m.cont_len = lens
    return m, keep


_lib = None


def load():
    """Loads libmarlin_host.so (which links libmarlin_gpu.so).  Fails loudly when not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(LIB_DIR, "libmarlin_host.so")
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make` in the repo root" % path)
    # PyTorch ships ROCm runtime libraries of its own.  A process that loads this library first (against /opt/rocm) and
    # imports torch afterwards runs with parts of both and aborts when it exits ("double free or corruption"; found with
    # tools/phase_probe.py in round 3); the other order shares one runtime.  Nothing in this package calls torch -- it is
    # the N > 1 plumbing of bench.py and of tests -- so it is imported first only where it is installed and not loaded yet.
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    # the GPU library first, so the dependency resolves without an rpath lookup surprise
    C.CDLL(os.path.join(LIB_DIR, "libmarlin_gpu.so"), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, ci, cu, cd = C.c_void_p, C.c_int, C.c_uint32, C.c_double
    sigs = {
        "m_sizeof": (C.c_size_t, [ci]),
        "msb_new": (vp, [ci, ci, ci, ci]),
        "msb_begin_doc": (cu, [vp]),
        "msb_add_string": (None, [vp, ci, C.c_char_p]),
        "msb_add_number": (None, [vp, ci, cd, ci]),
        "msb_add_facet_str": (None, [vp, ci, C.c_char_p]),
        "msb_add_facet_num": (None, [vp, ci, cd]),
        "msb_add_bool": (None, [vp, ci, ci]),
        "msb_end_doc": (None, [vp]),
        "msb_add_raw_wordpos": (None, [vp, cu, ci, cu]),
        "msb_add_raw_facet": (None, [vp, ci, cu]),
        "msb_finalize": (vp, [vp]),
        "mshard_free": (None, [vp]),
        "mshard_lookup_facet": (C.c_char_p, [vp, cu]),
        "m_facet_id": (cu, [C.c_char_p, C.c_size_t]),
        "m_farmhash32_mk": (cu, [C.c_char_p, C.c_size_t]),
        "msynth_datagen": (vp, [cu, C.c_uint64, C.POINTER(C.c_char_p), ci, C.POINTER(cu)]),
        "msynth_zipf": (vp, [C.c_uint64, cu, cu, C.c_uint64, ci, ci]),
        "msynth_zipf_first": (vp, [C.c_uint64, cu, cu, C.c_uint64, ci, ci, C.c_uint64]),
        "msynth_zipf_word": (ci, [cu, C.c_uint64, cu, C.c_char_p]),
        "msynth_route": (ci, [C.c_uint64, ci]),
        "mq_new": (C.POINTER(Query), [ci]),
        "mq_free": (None, [C.POINTER(Query)]),
        "mq_set_text": (ci, [C.POINTER(Query), C.c_char_p]),
        "mq_build_rank_rule": (None, [C.POINTER(Query)]),
        "mf_new": (C.POINTER(Filter), [ci]),
        "mf_add_child": (None, [C.POINTER(Filter), C.POINTER(Filter)]),
        "mf_free": (None, [C.POINTER(Filter)]),
        "mq_add_agg": (ci, [C.POINTER(Query), ci, ci]),
        "m_highlight": (C.c_void_p, [C.c_char_p, C.POINTER(Query), ci]),
        "m_highlight_free": (None, [C.c_void_p]),
        "m_highlight_batch": (ci, [vp, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(Query)), C.POINTER(ci), ci, C.POINTER(C.c_void_p)]),
        "m_form_result": (None, [C.POINTER(Query), C.POINTER(SqResult), ci, C.POINTER(Result)]),
        "m_result_free": (None, [C.POINTER(Result)]),
        "m_sqresult_free": (None, [C.POINTER(SqResult)]),
        "m_sqresult_free_batch": (None, [C.POINTER(SqResult), ci]),
        "mmirror_create": (ci, [vp, ci, C.POINTER(vp)]),
        "mmirror_destroy": (None, [vp]),
        "msq_set_plan_threads": (None, [ci]),
        "m_slot_bytes": (C.c_size_t, [ci, ci, ci, ci]),
        "m_slot_pack_batch": (None, [C.POINTER(SqResult), ci, ci, ci, ci, ci, vp]),
        "m_slot_unpack_gathered": (None, [vp, ci, ci, ci, ci, ci, ci, ci, C.POINTER(SqResult)]),
        "m_slot_unpack_gathered_subset": (None, [vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, C.POINTER(SqResult)]),
        "m_slot_card_count": (C.c_uint64, [C.POINTER(SqResult), ci, ci]),
        "m_slot_card_pack": (None, [C.POINTER(SqResult), ci, ci, vp]),
        "m_slot_card_attach": (ci, [vp, vp, C.c_uint64, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, C.POINTER(SqResult)]),
        "m_form_result_batch": (C.c_long, [C.POINTER(C.POINTER(Query)), C.POINTER(SqResult), ci, ci, C.POINTER(Result), ci]),
        "mmirror_gpu": (vp, [vp]),
        "msq_execute_batch": (ci, [vp, C.POINTER(vp), ci, C.POINTER(C.POINTER(Query)), ci, ci, C.POINTER(SqResult)]),
        "msq_prepare_batch": (ci, [C.POINTER(vp), ci, C.POINTER(C.POINTER(Query)), ci, ci, C.POINTER(vp)]),
        "msq_prepare_batch_dev": (ci, [vp, C.POINTER(vp), ci, C.POINTER(C.POINTER(Query)), ci, ci, C.POINTER(vp)]),
        "mgpu_dict_create": (ci, [ci, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), cu, C.POINTER(vp)]),
        "mgpu_dict_destroy": (None, [vp]),
        "mgpu_dict_bytes": (C.c_uint64, [vp]),
        "mgpu_lookup_terms": (ci, [vp, C.POINTER(TermLookup), ci, C.POINTER(C.POINTER(TermHit)), C.POINTER(cu)]),
        "mgpu_term_hits_free": (None, [C.POINTER(TermHit)]),
        "mdict_export_words": (ci, [vp, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(cu)]),
        "mdict_lookup_mode": (ci, [C.POINTER(Term)]),
        "msq_execute_prepared": (ci, [vp, vp, C.POINTER(SqResult)]),
        "msq_prepared_free": (None, [vp]),
        "msq_run_pipeline": (ci, [C.POINTER(vp), ci, C.POINTER(vp), ci, ci, ci, C.POINTER(C.c_long)]),
        "msq_pipeline_create": (ci, [C.POINTER(vp), ci, C.POINTER(vp)]),
        "msq_pipeline_warm": (ci, [vp, C.POINTER(vp), ci]),
        "msq_pipeline_run": (ci, [vp, C.POINTER(vp), ci, ci, ci, C.POINTER(C.c_long)]),
        "msq_pipeline_error": (C.c_char_p, [vp]),
        "mmirror_scheme": (vp, [vp]),
        "msq_resolve_batch": (ci, [vp, C.POINTER(vp), ci, C.POINTER(C.POINTER(Query)), ci, C.POINTER(vp)]),
        "msq_resolved_free": (None, [vp]),
        "msq_resolved_count": (ci, [vp]),
        "msq_prepared_new": (vp, []),
        "msq_prepared_plan": (C.POINTER(Plan), [vp, ci]),
        "msq_resolved_term": (cu, [vp, ci, ci, C.POINTER(cu), C.POINTER(C.POINTER(cu)), C.POINTER(C.POINTER(C.c_uint8))]),
        "msq_resolved_exact": (cu, [vp, ci, ci]),
        "msq_plan_resolved": (ci, [vp, ci, vp]),
        "msq_pipeline_warm_resolved": (ci, [vp, C.POINTER(vp), ci]),
        "msq_pipeline_run_resolved": (ci, [vp, C.POINTER(vp), ci, ci, ci, C.POINTER(C.c_long)]),
        "mmirror_apply_shard": (ci, [vp, vp]),
        "mgpu_shard_max_docid": (C.c_uint32, [vp]),
        "mgpu_shard_doc_capacity": (C.c_uint32, [vp]),
        "msq_pipeline_plan_ms": (cd, [vp]),
        "msq_pipeline_lookup_ms": (cd, [vp]),
        "msq_pipeline_run_queries": (ci, [vp, C.POINTER(vp), ci, C.POINTER(C.POINTER(Query)), ci, ci, ci, ci, ci, C.POINTER(C.c_long)]),
        "msq_pipeline_set_gather": (ci, [vp, vp, ci, ci, ci, ci, ci]),
        "mgpu_comm_unique_id": (ci, [vp]),
        "mgpu_comm_create": (ci, [ci, vp, ci, ci, ci, C.c_size_t, C.POINTER(vp)]),
        "mgpu_comm_destroy": (None, [vp]),
        "mgpu_comm_rank": (ci, [vp]),
        "mgpu_comm_world": (ci, [vp]),
        "mgpu_comm_allgather_begin": (ci, [vp, ci, vp, C.c_size_t, vp]),
        "mgpu_comm_allgather_end": (ci, [vp, ci]),
        "mgpu_host_alloc": (vp, [C.c_size_t]),
        "mgpu_host_free": (None, [vp]),
        "msq_pipeline_destroy": (None, [vp]),
        # GPU C ABI (include/marlin_gpu.h), reached through the same process image
        "mgpu_last_error": (C.c_char_p, []),
        "mgpu_device_count": (ci, []),
        "mgpu_ctx_create": (ci, [ci, C.c_uint64, C.POINTER(vp)]),
        "mgpu_ctx_destroy": (None, [vp]),
        "mgpu_ctx_set_profiling": (None, [vp, ci]),
        "mgpu_last_timing": (ci, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "mgpu_last_host_timing": (ci, [vp, C.POINTER(C.c_float)]),
        "mgpu_last_bytes": (ci, [vp, C.POINTER(C.c_uint64)]),
        "mgpu_stage_name": (C.c_char_p, [ci]),
        "mgpu_shard_bytes": (C.c_uint64, [vp]),
        "mgpu_posting_op": (ci, [vp, vp, ci, cu, cu, C.POINTER(C.c_uint64), cu, C.POINTER(C.c_uint64)]),
        "mbr_new": (vp, [ci, ci, ci, cu, cu]),
        "mbr_free": (None, [vp]),
        "mbr_add_posting": (ci, [vp, ci, cu, cu, C.POINTER(Mbmap)]),
        "mbr_add_docdata": (ci, [vp, cu, vp, C.c_size_t]),
        "mbr_add_num": (ci, [vp, ci, cd, cu]),
        "mbr_set_used": (ci, [vp, C.POINTER(cu), C.c_uint64]),
        "mbr_finish": (vp, [vp]),
        "mshard_twid_posting": (C.c_uint64, [vp, cu, C.POINTER(C.POINTER(cu))]),
        "mshard_twid_wids": (C.c_uint64, [vp, cu, C.POINTER(C.POINTER(cu))]),
        "mshard_facet_posting": (C.c_uint64, [vp, ci, cu, C.POINTER(C.POINTER(cu))]),
        "mshard_bool_posting": (C.c_uint64, [vp, ci, ci, C.POINTER(C.POINTER(cu))]),
        "mshard_docdata": (C.POINTER(C.c_uint8), [vp, cu, C.POINTER(cu)]),
        "mgpu_mbmap_decode": (ci, [C.POINTER(Mbmap), C.POINTER(cu), C.c_uint64, C.POINTER(C.c_uint64)]),
        "mgpu_shard_create": (ci, [ci, C.POINTER(ShardDesc), C.POINTER(vp)]),
        "mgpu_shard_destroy": (None, [vp]),
        "mgpu_posting_op_noalloc": (ci, [vp, vp, ci, cu, cu, C.POINTER(C.c_uint64), cu, C.POINTER(C.c_uint64)]),
        "mshard_wid_posting": (C.c_uint64, [vp, cu, C.POINTER(C.POINTER(cu))]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    expect = [DocRank, FacetCount, Agg, QueryCfg, Query, SqResult, Result, Filter, Word, Term, ShardFull, Plan]
    for i, st in enumerate(expect):
        if lib.m_sizeof(i) != C.sizeof(st):
            raise RuntimeError("ctypes mirror of %s is %d bytes, library says %d"
                               % (st.__name__, C.sizeof(st), lib.m_sizeof(i)))
    _lib = lib
    return lib
