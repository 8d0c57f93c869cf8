"""Pins the CPU oracle against the reference's python known-answer tests
(test/py/pytests.py:330-701) on the data_gen.py corpus (10k docs, 1 shard)."""
import pytest

from oracle import orc_py
from util import datagen_index

COUNT = 10000
WORDS = ["aaaaaaa", "bbbbbbb", "ccccccc", "ddddddd", "zzzzzzz", "yyyyyyy", "xxxxxxx", "wwwwwww"]
FONLY = ["oneone", "twotwo", "threethree", "fourfour"]
RCOUNT = [COUNT, COUNT * 3 // 4, COUNT // 2, COUNT // 4]


@pytest.fixture(scope="module")
def dg():
    ix, fcount = datagen_index(COUNT)
    yield ix, fcount
    ix.close()


def q(ix, jq):
    pq = ix.query(jq)
    r = orc_py.search(ix, pq)
    pq.close()
    return r


def ids(r):
    return [h["_docid"] for h in r["hits"]]


def test_a_strings(dg):                                     # pytests.py:330-337
    ix, _ = dg
    for i, w in enumerate(WORDS):
        for j in range(2, len(w) + 1):
            assert q(ix, {"q": w[:j]})["totalHits"] == RCOUNT[i % 4], w[:j]


def test_c_order_direction(dg):                             # pytests.py:385-409
    ix, _ = dg
    assert ids(q(ix, {"q": "aa1aa", "rankBy": {"id": "asc"}}))[:9] == list(range(1, 10))
    assert ids(q(ix, {"q": "aa1aa", "rankBy": {"id": "desc"}}))[:9] == [COUNT - i + 1 for i in range(1, 10)]
    assert ids(q(ix, {"q": "aa1aa", "rankBy": {"num": "asc"}}))[:9] == [COUNT - i + 1 for i in range(1, 10)]
    assert ids(q(ix, {"q": "aa1aa", "rankBy": {"num": "desc"}}))[:9] == list(range(1, 10))


def test_d_result_bmap(dg):                                 # pytests.py:413-431
    ix, _ = dg
    fcount = [[COUNT, COUNT * .75, COUNT * .5, COUNT * .25], [COUNT * .75] * 2 + [COUNT * .5, COUNT * .25],
              [COUNT * .5] * 3 + [COUNT * .25], [COUNT * .25] * 4]
    for i, w in enumerate(WORDS):
        for j in (2, 4, len(w)):
            r = q(ix, {"q": w[:j]})
            assert r["totalHits"] == RCOUNT[i % 4]
            got = {f["key"]: f["count"] for f in r["facets"]["facetonly"]}
            for k, name in enumerate(FONLY):
                if name in got:
                    assert got[name] == fcount[i % 4][k]


def test_e_facet_counts(dg):                                # pytests.py:434-440
    ix, fcount = dg
    r = q(ix, {"q": WORDS[0]})
    n = 0
    for fv in r["facets"].values():
        for f in fv:
            if f["key"] in FONLY:
                continue
            assert fcount[f["key"]] == f["count"]
            n += 1
    assert n >= 10


def test_g_boolean_filter(dg):                              # pytests.py:460-528
    ix, _ = dg
    r = q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"bool": {"$eq": True}}})
    assert ids(r) == [10 * v for v in range(1, 26)]
    r = q(ix, {"q": WORDS[0], "rankBy": {"id": "desc"}, "filter": {"bool": {"$eq": True}}})
    assert ids(r) == [COUNT - 10 * v for v in range(25)]
    want = [v for v in range(1, 40) if v % 10][:25]
    for flt in ({"bool": {"$ne": True}}, {"bool": {"$eq": False}}):
        assert ids(q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": flt})) == want
    r = q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"bool": {"$ne": False}}})
    assert ids(r) == [10 * v for v in range(1, 26)]


def test_h_numeric_filter(dg):                              # pytests.py:531-594
    ix, _ = dg
    r = q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"id": {"$eq": 100}}})
    assert r["totalHits"] == 1 and ids(r) == [100]
    assert q(ix, {"q": WORDS[2], "rankBy": {"id": "asc"}, "filter": {"id": {"$eq": 2}}})["totalHits"] == 0
    assert q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"id": {"$ne": 100}}})["totalHits"] == COUNT - 1
    assert ids(q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"id": {"$gt": 100}}})) == list(range(101, 126))
    assert ids(q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"id": {"$gte": 100}}})) == list(range(100, 125))
    assert ids(q(ix, {"q": WORDS[0], "rankBy": {"id": "desc"}, "filter": {"id": {"$lt": 100}}})) == list(range(99, 74, -1))
    assert ids(q(ix, {"q": WORDS[0], "rankBy": {"id": "desc"}, "filter": {"id": {"$lte": 100}}})) == list(range(100, 75, -1))
    r = q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"id": {"$gt": 100, "$lt": 105}}})
    assert r["totalHits"] == 4 and ids(r) == [101, 102, 103, 104]
    r = q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"id": {"$gte": 1000, "$lte": 1005}}})
    assert r["totalHits"] == 6 and ids(r) == list(range(1000, 1006))


def test_i_facet_filter(dg):                                # pytests.py:596-623
    ix, _ = dg
    for i, f in enumerate(FONLY):
        assert q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"facetonly": f}})["totalHits"] == RCOUNT[i]
    for n in (2, 3, 4):
        r = q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"facetonly": {"$and": FONLY[:n]}}})
        assert r["totalHits"] == RCOUNT[n - 1]
    assert q(ix, {"q": "", "rankBy": {"id": "asc"}, "filter": {"facetonly": {"$in": FONLY[2:]}}})["totalHits"] == RCOUNT[2]
    assert q(ix, {"q": "", "rankBy": {"id": "asc"}, "filter": {"facetonly": {"$nin": FONLY[2:]}}})["totalHits"] == RCOUNT[2]
    assert q(ix, {"q": "", "rankBy": {"id": "asc"}, "filter": {"facetonly": {"$nin": [FONLY[0], FONLY[3]]}}})["totalHits"] == 0


def test_j_pagination(dg):                                  # pytests.py:626-651
    ix, _ = dg
    from marlin_amd import QueryError
    with pytest.raises(QueryError):
        ix.query({"q": WORDS[0], "rankBy": {"id": "asc"}, "page": 1000, "hitsPerPage": 20})
    with pytest.raises(QueryError):
        ix.query({"q": WORDS[0], "rankBy": {"id": "asc"}, "page": -1, "hitsPerPage": 20})
    for i in (1, 2, 7, 24):
        r = q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "page": i, "hitsPerPage": 20})
        assert ids(r) == list(range((i - 1) * 20 + 1, i * 20 + 1))
    with pytest.raises(QueryError):
        ix.query({"q": WORDS[0], "rankBy": {"id": "asc"}, "page": 26, "hitsPerPage": 20})
    r = q(ix, {"q": WORDS[0], "rankBy": {"id": "asc"}, "filter": {"id": {"$lte": 45}}, "page": 5, "hitsPerPage": 10})
    assert r["totalHits"] == 45 and len(r["hits"]) == 5


def test_k_emptyquery(dg):                                  # pytests.py:653-662
    ix, _ = dg
    assert q(ix, {"rankBy": {"id": "asc"}})["totalHits"] == COUNT
    assert q(ix, {"filter": {"bool": True}})["totalHits"] == COUNT // 10
    assert q(ix, {"filter": {"bool": False}})["totalHits"] == COUNT - COUNT // 10


def test_lmn_rank(dg):                                      # pytests.py:664-701
    ix, _ = dg
    r = q(ix, {"q": "attr", "rankBy": {"id": "asc"}, "hitsPerPage": 50})
    assert r["totalHits"] == COUNT and ids(r) == list(range(1, 200, 4))
    r = q(ix, {"q": "eeeeeee", "rankBy": {"id": "asc"}, "hitsPerPage": 50})
    assert r["totalHits"] == 50
    assert ids(r)[:9] == list(range(5, 50, 5))
    assert ids(r)[10:] == [v for v in range(1, 51) if v % 5]
    r = q(ix, {"q": "aaaaaaa attr", "rankBy": {"id": "asc"}, "hitsPerPage": 50})
    assert r["totalHits"] == COUNT and ids(r) == list(range(1, 200, 4))
