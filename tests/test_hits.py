"""Row f-4: hit materialisation -- getFields projection and highlighting of the stored documents of the page's hits
(query.c:155-310, 385-401; highlight.c).  The known answers are the reference's own: test/robot/limit.robot's <mark>
counts on doc.json (tests/golden/robot_limit.json, made by make_limit_fixture.py).  The documents are ranked by the CPU
oracle here so that the test needs no GPU; the GPU engine assembles hits through the same marlin_amd.hits code."""
import ctypes as C
import json

import pytest

from marlin_amd import Index, ffi, hits
from oracle import orc_py
from util import load_fixture


@pytest.fixture(scope="module")
def fx():
    return load_fixture("limit")


def search(fx, case):
    settings = {"indexedFields": ["word", "sentence", "number", "words"], "facetFields": ["word"]}
    settings.update(case["settings"])
    ix = Index(settings, num_shards=1).add_documents([fx["doc"]])
    try:
        pq = ix.query(dict(case["query"], fullScan=True))
        return orc_py.search(ix, pq)
    finally:
        ix.close()


def test_reference_known_answers(fx):
    n = 0
    for case in fx["cases"]:
        res = search(fx, case)
        for chk in case["checks"]:
            if "totalHits" in chk:
                assert res["totalHits"] == chk["totalHits"], case["query"]
            elif "marks" in chk:
                hit = {k: v for k, v in res["hits"][0].items() if not k.startswith("_") or k == "_highlight"}
                where = hit["_highlight"] if chk["where"] == "_highlight" else hit
                assert str(where).count("<mark>") == chk["marks"], (case["query"], case["settings"], where)
            elif "has_highlight_key" in chk:
                assert ("_highlight" in res["hits"][0]) == chk["has_highlight_key"], case["query"]
            n += 1
    assert n == 24


def hl(ix, q, text):
    pq = ix.query({"q": q})
    return hits.Highlighter(ix.lib, pq.q).text(text)


def test_highlight_string_cases():
    """the strings of test/robot/highlight.robot (the suite itself only asserts status 200) and the matcher's rules"""
    ix = Index({"indexedFields": ["title"]}, num_shards=1).add_documents([{"title": "amazon"}])
    try:
        s = "amazong this amaznger this bmazon this amaozn this amazon this amazion ama-xon this amazing a.m.a.z.o.n here amazos"
        out = hl(ix, "amazon", s)
        # one edit allowed (6 letters), the last word of the query is a prefix term; the marks wrap the matched PREFIX
        assert out.startswith("<mark>amazon</mark>g this <mark>amazn</mark>ger this <mark>bmazon</mark> this <mark>amaozn</mark> this <mark>amazon</mark> this")
        assert "<mark>a.m.a.z.o.n</mark>" in out                                  # abbreviation dots are inside the token
        assert "<mark>ama-xon</mark>" in out                                     # the joined word of a hyphenated pair
        assert out.endswith("here <mark>amazo</mark>s")                          # the shortest best prefix: "amazo" is one edit away too
        assert hl(ix, "amazon", "nothing to see") is None
        assert hl(ix, "amazon", "In the Amazon middle.") == "In the <mark>Amazon</mark> middle"   # text after the last token is dropped
        assert hl(ix, "have ", "havelock a bet have") == "havelock a bet <mark>have</mark>"       # trailing space: no prefix term
        assert hl(ix, "have", "havelock a bet") == "<mark>have</mark>lock a bet"
    finally:
        ix.close()


def test_projection_rules():
    doc = load_fixture("limit")["doc"]
    f = hits.parse_limit_fields(["word", "objects.word", "objects.object.iword", "object", "nosuch"])
    out = hits.limit_fields(doc, f)
    assert list(out) == ["word", "objects", "object"]
    assert out["objects"] == [{"word": "this", "object": {"iword": "this"}}] * 2
    assert out["object"] == doc["object"]                                         # no child named: the whole object
    assert hits.parse_limit_fields(["a", "a", "a.b"]) == {"a": {"b": {}}}
