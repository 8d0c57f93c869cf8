"""f-4 on the device: the matcher of highlight() (is_match / levenshtein, highlight.c:166-213) for the tokens of ALL
strings of ALL hits of a batch of queries in one launch (mgpu_highlight_match behind m_highlight_batch), the documents read
from their stored docid2json values ([u32 length][zlib], sdata.c:105-122, 183-201).  Checked three ways: against the
reference's own known answers (test/robot/limit.robot's <mark> counts: tests/golden/robot_limit.json); string by string
against the host matcher m_highlight (= highlight.c restated, pinned by the same suite on the CPU) over generated text
-- typos, transpositions, prefixes, hyphenated and dotted words, long tokens, bytes above 0x7F, snippets; and whole
result pages: search(device_highlight=True) == search().  pytest -m gpu."""
import ctypes as C
import random

import pytest

from marlin_amd import GpuEngine, Index, ffi, hits
from util import load_fixture

pytestmark = pytest.mark.gpu


def test_reference_known_answers_on_the_device():
    fx = load_fixture("limit")
    n = 0
    for case in fx["cases"]:
        settings = {"indexedFields": ["word", "sentence", "number", "words"], "facetFields": ["word"]}
        settings.update(case["settings"])
        ix = Index(settings, num_shards=1).add_documents([fx["doc"]])
        eng = GpuEngine(ix)
        try:
            pq = ix.query(dict(case["query"], fullScan=True))
            res = eng.search(pq, device_highlight=True)
            assert res == eng.search(pq), case["query"]                  # and the same page as the host matcher makes
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
        finally:
            eng.close(); ix.close()
    assert n == 24


WORDS = ["amazon", "amazing", "river", "rivers", "delta", "deltas", "forest", "rainforest", "a.m.a.z.o.n", "ama-zon", "rain-forest-river",
         "don't", "x", "ab", "abc", "abcd", "international", "internationalisation", "café", "naïve", "straße", "z" * 126, "q" * 140]


def mutate(rng, w):
    w = list(w)
    for _ in range(rng.randrange(0, 3)):
        op = rng.randrange(4)
        if op == 0 and w:
            w[rng.randrange(len(w))] = rng.choice("abcdefghijklmnopqrstuvwxyz")
        elif op == 1:
            w.insert(rng.randrange(len(w) + 1), rng.choice("abcxyz"))
        elif op == 2 and len(w) > 1:
            del w[rng.randrange(len(w))]
        elif op == 3 and len(w) > 1:
            i = rng.randrange(len(w) - 1); w[i], w[i + 1] = w[i + 1], w[i]
    return "".join(w)


def test_batch_equals_the_host_matcher_string_by_string():
    rng = random.Random(4)
    ix = Index({"indexedFields": ["title"]}, num_shards=1).add_documents([{"title": " ".join(WORDS[:8])}])
    eng = GpuEngine(ix)
    lib = eng.lib
    try:
        jqs = [{"q": "amazon river"}, {"q": "amazng rainforest del"}, {"q": "internationalisation", "typos": False}, {"q": "abcd ab x"},
               {"q": "rivers amazon forest delta", "prefix": "all"}, {"q": "café straße"}, {"q": "z" * 100}, {"q": "don't ama-zon"}]
        pqs = [ix.query(j) for j in jqs]
        strs, qs, snips = [], [], []
        for pq in pqs:
            for _ in range(60):
                sep = rng.choice([" ", ", ", ". ", " - ", "-", "  ", "\t", "'s "])
                text = sep.join(mutate(rng, rng.choice(WORDS)) if rng.random() < 0.7 else rng.choice(WORDS).upper() for _ in range(rng.randrange(0, 14)))
                strs.append(text + rng.choice(["", ".", " ", "!"])); qs.append(pq); snips.append(rng.choice([0, 0, 3, 5]))
        n = len(strs)
        enc = [s.encode("utf-8") for s in strs]
        out = (C.c_void_p * n)()
        rc = lib.m_highlight_batch(eng.ctx, (C.c_char_p * n)(*enc), (C.POINTER(ffi.Query) * n)(*[pq.q for pq in qs]), (C.c_int * n)(*snips), n, out)
        assert rc == 0, lib.mgpu_last_error()
        marked = 0
        for i in range(n):
            p = lib.m_highlight(enc[i], qs[i].q, snips[i])
            want = C.string_at(p) if p else None
            got = C.string_at(out[i]) if out[i] else None
            if p:
                lib.m_highlight_free(p)
            if out[i]:
                lib.m_highlight_free(out[i])
            assert got == want, (strs[i], qs[i].jq, snips[i])
            marked += want is not None
        assert marked > 100                                              # not vacuous
        # empty input, and a context with a batch in flight is refused
        assert lib.m_highlight_batch(eng.ctx, None, None, None, 0, None) == 0
    finally:
        eng.close(); ix.close()


def test_pages_of_a_batch_from_stored_documents():
    rng = random.Random(6)
    docs = [{"id": i, "title": " ".join(rng.choice(WORDS[:12]) for _ in range(rng.randrange(1, 9))),
             "body": {"text": " ".join(mutate(rng, rng.choice(WORDS[:12])) for _ in range(rng.randrange(0, 12))), "tags": [rng.choice(WORDS[:8]) for _ in range(rng.randrange(0, 3))]},
             "price": float(rng.randrange(1000))} for i in range(600)]
    ix = Index({"indexedFields": ["title", "price"], "facetFields": [], "hitsPerPage": 10}, num_shards=3).add_documents(docs)
    assert hits.unpack_stored(ix.stored[0][0]) == ix.docs[0][0]            # the stored value round-trips
    eng = GpuEngine(ix)
    try:
        jqs = [{"q": "amazon river"}, {"q": "amazin forest", "getFields": ["title", "body.text"]}, {"q": "delta", "highlightSource": True},
               {"q": "rainforest riv", "highlightFields": ["body.text", "title"]}, {"q": "rivers", "highlightFields": None},
               {"q": "amazon", "page": 2, "rankBy": {"price": "desc"}}, {"q": "nosuchword"}]
        pqs = [ix.query(dict(j, fullScan=True)) for j in jqs]
        a = eng.search(pqs, device_highlight=True)
        b = eng.search(pqs)
        assert a == b
        assert sum(str(r["hits"]).count("<mark>") for r in a) > 50
    finally:
        eng.close(); ix.close()
