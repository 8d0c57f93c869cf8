"""Plan building without a GPU: msq_resolve_batch (dictionary half of lookup_terms) + msq_plan_resolved (everything after
it) against a Python restatement of process_termresult / set_wids_under_twid / the exact-wid step
(/root/reference/src/main/squery.c:48-118, 144-167) written over the shard's own stored sets."""
import ctypes as C

from marlin_amd import ffi
from util import datagen_index

QUERIES = [
    {"q": "aaaaaab bbbbbbc"},                               # two typo'd words -> 3 terms
    {"q": "aaaaaaa bbbbbbb ccccccc"},                       # 6 terms
    {"q": "aaa"},                                           # short prefix -> twid term
    {"q": "aaaaaaa bb"},
    {"q": "zzzzzzzz"},                                      # no match
    {"q": "", "filter": {"bool": True}},
    {"q": "eeeeeed aaaaaaa", "filter": {"num": {"$gte": 10, "$lt": 500}}, "aggs": {"a": {"avg": {"field": "num"}}}},
    {"q": "aaaaaaa bbbbbbb ccccccc ddddddd"},
]


def restate(lib, shard, W, res, i, q):
    """-> (term_post lists, nzero, wid table dict wid -> (dist, mterm, exact_mask), exact_present)"""
    nt, nw = q.num_terms, q.num_words
    docs = C.POINTER(C.c_uint32)()
    allw, first = {}, {}
    posts, nzero = [], []
    for t in range(nt):
        twid = C.c_uint32(); wp = C.POINTER(C.c_uint32)(); dp = C.POINTER(C.c_uint8)()
        n = lib.msq_resolved_term(res, i, t, C.byref(twid), C.byref(wp), C.byref(dp))
        if twid.value:
            if lib.mshard_twid_posting(shard, twid.value, C.byref(docs)):
                posts.append([W + twid.value]); nzero.append(1)
                ww = C.POINTER(C.c_uint32)()
                for k in range(lib.mshard_twid_wids(shard, twid.value, C.byref(ww))):
                    allw[ww[k]] = 0; first.setdefault(ww[k], t)
            else:
                posts.append([]); nzero.append(0)
            continue
        pl, z = [], 0
        for k in range(n):
            wid, d = wp[k], dp[k]
            if not wid or not lib.mshard_wid_posting(shard, wid, C.byref(docs)):
                continue
            allw[wid] = d; first.setdefault(wid, t)
            pl.append(wid); z += d == 0
        posts.append(pl); nzero.append(z)
    table = {w: [d, first[w], 0] for w, d in allw.items()}
    present = 0
    for w in range(nw):
        wid = lib.msq_resolved_exact(res, i, w)
        if not wid or not lib.mshard_wid_posting(shard, wid, C.byref(docs)):
            continue
        present |= 1 << w
        table.setdefault(wid, [0xFF, 0xFF, 0])[2] |= 1 << w
    return posts, nzero, table, present


def test_plans_match_process_termresult():
    lib = ffi.load()
    ix, _ = datagen_index(3000)
    try:
        sh = ix.shards[0]
        m = lib.mmirror_scheme(sh)
        W = ffi.Mirror.from_address(m).W
        marr = (C.c_void_p * 1)(m)
        pqs = [ix.query(j) for j in QUERIES]
        qarr = (C.POINTER(ffi.Query) * len(pqs))(*[p.q for p in pqs])
        res = C.c_void_p()
        assert lib.msq_resolve_batch(None, marr, 1, qarr, len(pqs), C.byref(res)) == 0
        assert lib.msq_resolved_count(res) == len(pqs)
        pp = C.c_void_p(lib.msq_prepared_new())
        for rnd in range(2):                                  # the second round reuses the batch's arena
            assert lib.msq_plan_resolved(res, 0, pp) == 0
            some_terms = 0
            for i, pq in enumerate(pqs):
                q = pq.q.contents
                p = lib.msq_prepared_plan(pp, i).contents
                posts, nzero, table, present = restate(lib, sh, W, res, i, q)
                assert (p.num_terms, p.num_words) == (q.num_terms, q.num_words)
                for t in range(q.num_terms):
                    got = [p.term_post[t][k] for k in range(p.term_npost[t])]
                    assert sorted(got) == sorted(posts[t]), (pq.jq, t)
                    assert p.term_nzero[t] == nzero[t], (pq.jq, t)
                    some_terms += len(got)
                assert p.exact_present == present, pq.jq
                gt = {p.wids[k].wid: [p.wids[k].dist, p.wids[k].mterm, p.wids[k].exact_mask] for k in range(p.num_wids)}
                assert len(gt) == p.num_wids and gt == table, pq.jq
                assert bool(p.has_filter) == ("filter" in pq.jq)
                assert p.num_aggs == q.num_aggs
            assert some_terms > 10
        assert lib.msq_prepared_plan(pp, len(pqs)) is None or not lib.msq_prepared_plan(pp, len(pqs))
        lib.msq_prepared_free(pp)
        lib.msq_resolved_free(res)
        C.CDLL(None).free(C.c_void_p(m))
    finally:
        ix.close()
