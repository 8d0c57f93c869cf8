"""f-1: term lookup on the device (mgpu_lookup_terms, k_term_scan) against the host trie walk (mdict_lookup_term =
dtrie_lookup_term restated, dtrie.c:710-823): the wid -> least-distance sets must be equal for every term, and whole
queries prepared with device lookups must give the oracle's answers.  pytest -m gpu."""
import ctypes as C
import random

import pytest

from marlin_amd import GpuEngine, Index, ffi
from oracle import orc_py
from util import compare_results, datagen_index, load_fixture

pytestmark = pytest.mark.gpu


class U32Map(C.Structure):
    _fields_ = [("keys", C.POINTER(C.c_uint32)), ("vals", C.POINTER(C.c_int32)), ("cap", C.c_uint32), ("n", C.c_uint32)]


class TermResult(C.Structure):
    _fields_ = [("twid", C.c_uint32), ("wordids", U32Map)]


def host_walk(lib, dict_ptr, word, prefix):
    t = ffi.Term()
    t.word.length = len(word)
    for i, ch in enumerate(word):
        t.word.chars[i] = ord(ch)
    t.prefix = prefix; t.typos = 1
    if lib.mdict_lookup_mode(C.byref(t)) != 2:
        return None
    tr = TermResult()
    lib.mdict_lookup_term.argtypes = [C.c_void_p, C.POINTER(ffi.Term), C.POINTER(TermResult)]
    lib.mdict_lookup_term.restype = None
    lib.m_termresult_free.argtypes = [C.POINTER(TermResult)]
    lib.m_termresult_free.restype = None
    lib.mdict_lookup_term(dict_ptr, C.byref(t), C.byref(tr))
    out = {tr.wordids.keys[k]: tr.wordids.vals[k] for k in range(tr.wordids.cap) if tr.wordids.keys[k]}
    lib.m_termresult_free(C.byref(tr))
    return out


def device_scan(eng, terms):
    """terms: list of (shard index, word, prefix) -> list of {wid: dist}"""
    lib = eng.lib
    n = len(terms)
    arr = (ffi.TermLookup * n)()
    keep = []
    for i, (s, word, prefix) in enumerate(terms):
        mir = ffi.Mirror.from_address(eng.mirrors[s].value)
        buf = (C.c_uint32 * len(word))(*[ord(c) for c in word])
        keep.append(buf)
        arr[i].dict = mir.gd; arr[i].chars = C.cast(buf, C.POINTER(C.c_uint32)); arr[i].len = len(word); arr[i].prefix = prefix
    hits = C.POINTER(ffi.TermHit)()
    nh = C.c_uint32()
    rc = lib.mgpu_lookup_terms(eng.ctx, arr, n, C.byref(hits), C.byref(nh))
    assert rc == 0, lib.mgpu_last_error()
    out = [dict() for _ in range(n)]
    for h in range(nh.value):
        x = hits[h]
        assert x.wid not in out[x.lookup], "a word reported twice for one lookup"
        out[x.lookup][x.wid] = x.dist
    lib.mgpu_term_hits_free(hits)
    return out


def shard_dict(ix, s):
    return C.cast(ix.shards[s], C.POINTER(ffi.ShardFull)).contents.dict


def check_terms(ix, eng, words):
    terms, want = [], []
    for s in range(ix.num_shards):
        for w in words:
            for prefix in (0, 1):
                hw = host_walk(eng.lib, shard_dict(ix, s), w, prefix)
                if hw is None:
                    continue
                terms.append((s, w, prefix)); want.append(hw)
    got = device_scan(eng, terms)
    assert len(got) == len(want)
    for t, g, w in zip(terms, got, want):
        assert g == w, (t, len(g), len(w))
    return sum(len(w) for w in want)


def test_scan_equals_trie_walk_levenshtein_suite():
    fx = load_fixture("levenshtein")
    ix = Index(fx["settings"], num_shards=fx["numShards"]).add_documents(fx["docs"])
    eng = GpuEngine(ix)
    try:
        words = sorted({w for c in fx["cases"] for w in c["query"].get("q", "").split()} | {"tesst", "tsetx", "atestb", "ttest"})
        assert check_terms(ix, eng, words) > 0
        # whole queries of the suite, term lookups on the device
        pqs = [ix.query(dict(c["query"], fullScan=True)) for c in fx["cases"] if c["status"] != 400]
        got = eng.search(pqs, capture=3, device_lookup=True)
        for pq, g, c in zip(pqs, got, [c for c in fx["cases"] if c["status"] != 400]):
            compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
            for path, val in c["expect"]:
                if path == "$.totalHits":
                    assert g["totalHits"] == val, pq.jq            # the reference's own known answer
    finally:
        eng.close(); ix.close()


def test_scan_equals_trie_walk_datagen_and_queries():
    ix, _ = datagen_index(20000)
    eng = GpuEngine(ix)
    try:
        rng = random.Random(5)
        base = ["aaaaaaa", "bbbbbbb", "ccccccc", "aaaaaab", "bbbbbbc", "attr", "zzzzzzz", "aaaaaaabbbbbbb", "aaaaaaabbbbbbbccccccc", "abcdefgh"]
        words = list(base)
        for w in base:
            x = list(w)
            x[rng.randrange(len(x))] = rng.choice("abcxyz")
            words.append("".join(x))
            if len(w) > 2:
                i = rng.randrange(len(w) - 1)
                words.append(w[:i] + w[i + 1] + w[i] + w[i + 2:])
        assert check_terms(ix, eng, sorted(set(words))) > 10
        qs = [{"q": "aaaaaab bbbbbbc", "rankBy": {"num": "desc"}}, {"q": "aaaaaaa bbbbbbb ccccccc", "rankBy": {"id": "asc"}},
              {"q": "aaaaaax", "rankBy": {"id": "asc"}, "hitsPerPage": 30}, {"q": "bbbbbb attr", "sortBy": {"num": "asc"}},
              {"q": "zzzzzzy yyyyyyy", "rankBy": {"num": "desc"}}, {"q": "aa"}, {"q": ""}, {"q": "aaaaaaa", "typos": False}]
        pqs = [ix.query(dict(q, fullScan=True)) for q in qs]
        for pq, g in zip(pqs, eng.search(pqs, capture=3, device_lookup=True)):
            compare_results(g, orc_py.search(ix, pq, flags=3), ctx=pq.jq)
    finally:
        eng.close(); ix.close()


def test_scan_equals_trie_walk_on_a_zipf_vocabulary():
    """a 20000-word vocabulary of the bench generator (words of 4-10 letters): typo'd words, transpositions, two-word and
    three-word concatenations (the reference's odd terms, up to two edits), prefixes of every length"""
    lib = ffi.load()
    V, S = 20000, 9
    shards = [lib.msynth_zipf(60000, V, 500, S, s, 2) for s in range(2)]
    ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    eng = GpuEngine(ix)
    try:
        rng = random.Random(3)
        buf = C.create_string_buffer(16)

        def word(r):
            lib.msynth_zipf_word(V, S, r, buf)
            return buf.value.decode()
        terms = set()
        for _ in range(40):
            w = word(rng.randrange(0, 3000))
            p = rng.randrange(len(w))
            terms.add(w[:p] + rng.choice("abcdefghijklmnopqrstuvwxyz") + w[p + 1:])          # substitution
            if len(w) > 4:
                i = rng.randrange(len(w) - 1)
                terms.add(w[:i] + w[i + 1] + w[i] + w[i + 2:])                               # transposition
                terms.add(w[:p] + w[p + 1:])                                                 # deletion
            terms.add(w + word(rng.randrange(0, 200)))                                       # two words run together
            terms.add(w[:rng.randrange(4, len(w) + 1)])                                      # a prefix
        terms.add(word(1) + word(2) + word(3))
        n = check_terms(ix, eng, sorted(terms))
        assert n > 200
    finally:
        eng.close(); ix.close()


def test_serving_loop_with_the_lookups_inside():
    """msq_pipeline_run_queries: every step resolves its own queries (typo walks on the step's context, while other
    batches are in flight on the others), builds plans, executes, merges -- the hit sums equal those of the loop that
    is handed host-resolved term sets, over several passes and with more steps than batches"""
    lib = ffi.load()
    V, S = 20000, 9
    shards = [lib.msynth_zipf(60000, V, 500, S, s, 2) for s in range(2)]
    ix = Index.from_shards({}, shards, ["str"], ["num", "price"], ["flag"], ["cat", "hc"])
    eng = GpuEngine(ix, num_ctx=4)
    try:
        rng = random.Random(11)
        buf = C.create_string_buffer(16)

        def word(r):
            lib.msynth_zipf_word(V, S, r, buf)
            return buf.value.decode()
        qs = []
        for _ in range(48):
            ws = []
            for _ in range(2):
                w = word(rng.randrange(0, 40))
                p = rng.randrange(1, len(w))
                ws.append(w[:p] + rng.choice("abcdefghij") + w[p + 1:])
            qs.append({"q": " ".join(ws + [word(rng.randrange(0, 64))[:3]]), "fullScan": True, "rankBy": {"num": "desc"}, "maxFacetResults": 5})
        pqs = [ix.query(q) for q in qs]
        B = 8
        ress = [eng.resolve(pqs[b * B:(b + 1) * B]) for b in range(len(pqs) // B)]     # host trie walk
        eng.warm_pipeline_resolved(ress, 4)
        want, _ = eng.run_pipeline_resolved(ress, 0, 15, 4)
        assert want > 0
        for _ in range(2):
            got, lk_ms, _ = eng.run_pipeline_queries(pqs, B, 0, 15, 4)
            assert got == want and lk_ms > 0
        want2, _ = eng.run_pipeline_resolved(ress, 3, 4, 4)
        got2, _, _ = eng.run_pipeline_queries(pqs, B, 3, 4, 4)
        assert got2 == want2
        for r in ress:
            eng.free_resolved(r)
    finally:
        eng.close(); ix.close()
