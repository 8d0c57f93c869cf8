#!/usr/bin/env python3
"""Writes tests/golden/docdata_cases.json: docid2data blobs and num_dbi orders made by the REFERENCE's own index-writing
code (oracle/_ref/ref_seam = /root/reference/src/main/sindex.c's gen_wordpos_data, store_docdata and index_number over a
real LMDB environment; `make ref-seam`).  Build container only.  A fixture is data: the inputs below and the bytes LMDB
then held.  tests/test_docdata_format.py checks that marlin_amd/host/mshard.c writes the same bytes and that the mirror
reads them."""
import json
import os
import random
import struct
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_seam")


def make_cases():
    rng = random.Random(20261111)
    cases = []
    # 1: the shapes by hand -- single occurrence (inline position), repeats in one field, repeats over fields, a vint
    # position above 127 and above 16383, a document without words, one without facets, multi-valued facets
    cases.append({"num_numbers": 2, "num_facets": 2, "docs": [
        {"docid": 1, "nums": [[0, 5.0], [1, -2.5]], "facets": [[0, 111], [1, 222]], "words": [[7, 0, 1], [9, 0, 2], [3, 1, 1]]},
        {"docid": 2, "nums": [[0, 5.0]], "facets": [[0, 111], [0, 333], [0, 444]], "words": [[7, 0, 1], [7, 0, 5], [7, 0, 200], [7, 2, 1], [8, 1, 20000]]},
        {"docid": 3, "nums": [[1, 1e12]], "facets": [], "words": []},
        {"docid": 4, "nums": [[0, 7.25], [1, 7.25]], "facets": [[1, 4000000000]], "words": [[1, 3, 65535], [2, 0, 128], [2, 0, 129], [2, 1, 127]]},
        {"docid": 1005, "nums": [[0, 5.0]], "facets": [[0, 111]], "words": [[9, 0, 1]]},
    ]})
    # 2: no numbers, no facets
    cases.append({"num_numbers": 0, "num_facets": 0, "docs": [
        {"docid": 1, "nums": [], "facets": [], "words": [[5, 0, 1], [5, 0, 2], [6, 0, 3]]},
        {"docid": 2, "nums": [], "facets": [], "words": [[6, 0, 1]]},
    ]})
    # 3-6: generated
    for c in range(4):
        nn, nf = rng.randrange(0, 4), rng.randrange(0, 4)
        docs = []
        for d in range(1, rng.randrange(5, 40)):
            words = []
            for _ in range(rng.randrange(0, 30)):
                words.append([rng.randrange(1, 25), rng.randrange(0, 4), rng.choice([rng.randrange(1, 40), rng.randrange(1, 70000)])])
            # the analyzer never reports one (word, field, position) twice
            words = [list(t) for t in sorted(set(map(tuple, words)), key=lambda t: rng.random())]
            nums = [[f, float(rng.choice([rng.randrange(-50, 50), rng.randrange(0, 10 ** 9)]))] for f in range(nn) if rng.random() < 0.8]
            if nn and rng.random() < 0.2:
                nums.append([0, float(rng.randrange(-50, 50))])           # a second value of field 0 (numlist)
            facets = [[f, rng.randrange(1, 2 ** 32)] for f in range(nf) for _ in range(rng.choice([0, 1, 1, 1, 3]))]
            docs.append({"docid": d * (1 if c % 2 else 7), "nums": nums, "facets": facets, "words": words})
        cases.append({"num_numbers": nn, "num_facets": nf, "docs": docs})
    return cases


def run(cases):
    lines = []
    for c in cases:
        lines.append("C %d %d" % (c["num_numbers"], c["num_facets"]))
        for d in c["docs"]:
            lines.append("D %d" % d["docid"])
            for f, v in d["nums"]:
                lines.append("N %d %r" % (f, v))
            for f, i in d["facets"]:
                lines.append("F %d %d" % (f, i))
            for w, p, pos in d["words"]:
                lines.append("W %d %d %d" % (w, p, pos))
            lines.append("E")
        lines.append("X")
    with tempfile.TemporaryDirectory() as tmp:
        out = subprocess.run([EXE, tmp], input="\n".join(lines) + "\n", text=True, stdout=subprocess.PIPE, check=True).stdout
    ci = 0
    for c in cases:
        c["blobs"] = {}
        c["num_order"] = [[] for _ in range(c["num_numbers"])]
    for ln in out.splitlines():
        t = ln.split()
        if t[0] == "blob":
            cases[ci]["blobs"][t[1]] = None if t[2] == "-" else t[2]
        elif t[0] == "num":
            v = struct.unpack("<d", struct.pack("<Q", int(t[2], 16)))[0]
            cases[ci]["num_order"][int(t[1])].append([v, int(t[3])])
        elif t[0] == "end":
            ci += 1
    assert ci == len(cases)
    return cases


if __name__ == "__main__":
    cases = run(make_cases())
    with open(os.path.join(ROOT, "tests", "golden", "docdata_cases.json"), "w") as f:
        json.dump({"_about": "made by tests/golden/make_docdata_fixtures.py from the reference's sindex.c (oracle/ref_seam.c)",
                   "cases": cases}, f, separators=(",", ":"))
    print(len(cases), "cases,", sum(len(c["docs"]) for c in cases), "documents")
