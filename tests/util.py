"""helpers shared by the CPU and GPU test suites"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

ROBOT_SUITES = ["rank", "rank2", "rank3", "rank4", "rank5", "rank6", "rank7", "rank_terms",
                "levenshtein", "terms", "filter_apply", "aggs", "facet", "sort"]


def load_fixture(name):
    with open(os.path.join(GOLDEN, "robot_%s.json" % name)) as f:
        return json.load(f)


def jpath(obj, path):
    """evaluates the small JSONPath subset the robot suites use: $.a.b[0].c"""
    assert path.startswith("$")
    cur = obj
    for m in re.finditer(r"\.([A-Za-z_][A-Za-z0-9_]*)|\[(\d+)\]", path[1:]):
        cur = cur[m.group(1)] if m.group(1) is not None else cur[int(m.group(2))]
    return cur


RANK_FIELDS = ("typos", "proximity", "field", "position", "exact", "compare")


def rank_key(r, fields=RANK_FIELDS):
    return tuple(r[f] for f in fields)


def facets_canonical(res):
    return {k: sorted(((e["count"], e["facet_id"]) for e in v), key=lambda t: (-t[0], t[1]))
            for k, v in res["facets"].items()}


def datagen_index(rend=10000, seed=12345):
    """the reference's test corpus (test/data_gen.py) built by the C generator"""
    import ctypes as C
    from marlin_amd import Index, ffi
    lib = ffi.load()
    facets = open(os.path.join(GOLDEN, "facet.txt")).read().splitlines()
    arr = (C.c_char_p * len(facets))(*[f.encode() for f in facets])
    counts = (C.c_uint32 * len(facets))()
    sh = lib.msynth_datagen(rend, seed, arr, len(facets), counts)
    settings = {"hitsPerPage": 25}
    ix = Index.from_shards(settings, [sh], ["str", "strlist", "facet", "facetlist", "spell"],
                           ["id", "num", "numlist"], ["bool"], ["facet", "facetlist", "facetonly"])
    ix.fields["numlist"].type = "numlist"
    fcount = {}
    for f, c in zip(facets, counts):
        if c:
            fcount[f] = fcount.get(f, 0) + c
    return ix, fcount


# merged facet lists compared / left out because a shard's tied tail differed (several shards only)
COMPARED_MERGED_FACETS = [0]
SKIPPED_MERGED_FACETS = [0]


def compare_results(gpu, orc, ctx="", key_fields=RANK_FIELDS):
    """GPU result vs oracle result for one query (both assembled with capture flags 3).

    Exact where the reference defines the answer; tie-class aware where it does not
    (DESIGN.md "ties"): equal rank keys may carry different docids, facets of equal count at
    the cut may differ.  key_fields: the rank fields the query orders by (a custom rankAlgorithm leaves the
    others out of the order, so documents that differ only in those form one tie class)."""
    assert gpu["totalHits"] == orc["totalHits"], ctx
    assert gpu["numHits"] == orc["numHits"], ctx
    # stage level: bitmap words and every per-document rank record
    for s, (gs, os_) in enumerate(zip(gpu["_shards"], orc["_shards"])):
        assert gs["num_hits"] == os_["num_hits"], (ctx, s)
        if "docid_map" in gs and "docid_map" in os_:
            assert gs["docid_map"] == os_["docid_map"], (ctx, s, "docid_map")
        if "all_ranks" in gs and "all_ranks" in os_:
            assert len(gs["all_ranks"]) == len(os_["all_ranks"]), (ctx, s)
            for a, b in zip(gs["all_ranks"], os_["all_ranks"]):
                assert a == b, (ctx, s, "rank record", a, b)
        assert gs["rank_count"] == os_["rank_count"], (ctx, s)
        for a, b in zip(gs["aggs"], os_["aggs"]):
            assert a == b, (ctx, s, "agg", a, b)
    # merged order: identical key sequence; docids identical wherever the key is unique
    gk = [rank_key(r, key_fields) for r in gpu["ranks"]]
    ok = [rank_key(r, key_fields) for r in orc["ranks"]]
    assert gk == ok, ctx
    # (a key that equals the LAST key a shard kept may tie with documents that shard cut: tie class, any member)
    cut_keys = set()
    for gs in gpu["_shards"]:
        if gs["num_hits"] > gs["rank_count"] and gs["ranks"]:
            last = dict(zip(("docid",) + RANK_FIELDS, gs["ranks"][-1]))
            cut_keys.add(rank_key(last, key_fields))
    for i, (a, b) in enumerate(zip(gpu["ranks"], orc["ranks"])):
        if gk.count(gk[i]) == 1 and not _boundary_tie(gpu, i) and tuple(gk[i]) not in cut_keys:
            assert (a["shard"], a["docid"]) == (b["shard"], b["docid"]), (ctx, i)
    # per-shard facet lists (top 2N): same counts everywhere; same ids above the cut-off count.  Which of the
    # ids that TIE at the cut-off count a shard keeps is the reference's quickselect residue (DESIGN.md "ties").
    exact_lists = {}
    for s, (gs, os_) in enumerate(zip(gpu["_shards"], orc["_shards"])):
        assert gs["facets"].keys() == os_["facets"].keys(), (ctx, s)
        for f in gs["facets"]:
            ga = sorted(((c, i) for i, c in gs["facets"][f]), key=lambda t: (-t[0], t[1]))
            oa = sorted(((c, i) for i, c in os_["facets"][f]), key=lambda t: (-t[0], t[1]))
            assert [c for c, _ in ga] == [c for c, _ in oa], (ctx, s, "facet counts", f)
            if ga:
                cut = ga[-1][0]
                assert [e for e in ga if e[0] > cut] == [e for e in oa if e[0] > cut], (ctx, s, "facet ids", f)
            exact_lists[f] = exact_lists.get(f, True) and ga == oa
    # merged facets: same counts everywhere; same ids above the cut-off count.  With several shards the sums
    # themselves depend on the shards' tie choices, so the merged lists are compared only when no shard's list
    # differed in its tied tail.
    gf, of = facets_canonical(gpu), facets_canonical(orc)
    assert gf.keys() == of.keys(), ctx
    # field names appear in field order in the merged dict, field indices in the same order in the shard dicts
    by_order = dict(zip(gf.keys(), sorted(exact_lists)))
    for k in gf:
        f = by_order.get(k)
        if len(gpu["_shards"]) > 1 and f is not None and not exact_lists[f]:
            SKIPPED_MERGED_FACETS[0] += 1                     # this field only; tools/fuzz_parity.py prints the count
            continue
        COMPARED_MERGED_FACETS[0] += 1
        assert [c for c, _ in gf[k]] == [c for c, _ in of[k]], (ctx, k)
        if gf[k]:
            cut = gf[k][-1][0]
            assert [e for e in gf[k] if e[0] > cut] == [e for e in of[k] if e[0] > cut], (ctx, k)
    assert _same(gpu.get("aggs"), orc.get("aggs")), ctx


def _same(a, b):
    """equality that treats NaN == NaN (avg of an empty hit set)"""
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, float) and isinstance(b, float) and a != a and b != b:
        return True
    return a == b


def _boundary_tie(res, i):
    """the last returned key may tie with documents that were cut"""
    return i == len(res["ranks"]) - 1 and res["totalHits"] > len(res["ranks"])


def mbmap_encode(docids):
    """ascending unique docids -> (header bytes | None, [container bytes]) as mbmap_save writes a set (mbmap.c:183-251,
    cont.c): array containers up to 4096 members, bitset containers above, card stored modulo 65536.  Test support;
    tests/test_mbmap_format.py checks it byte for byte against the reference-made fixtures."""
    import struct
    if not docids:
        return None, []
    chunks = {}
    for d in docids:
        chunks.setdefault(d >> 16, []).append(d & 0xFFFF)
    cids = sorted(chunks)
    header = struct.pack("<H%dH" % len(cids), len(cids), *cids)
    conts = []
    for cid in cids:
        v = chunks[cid]
        if len(v) <= 4096:
            conts.append(struct.pack("<HH%dH" % len(v), cid, len(v), *v))
        else:
            bits = bytearray(8192)
            for x in v:
                bits[x >> 3] |= 1 << (x & 7)
            conts.append(struct.pack("<HH", cid, len(v) & 0xFFFF) + bytes(bits))
    return header, conts
