"""f-1 on the CPU: marlin_amd/csrc/mg_lev.h -- the per-word restatement of the reference's typo walk (dtrie.c:710-791)
that the GPU term scan runs, one lane per dictionary word -- compiled for the host and compared with mdict_lookup_term
(the trie walk) over whole dictionaries: every word is scored, the wid -> distance sets must be equal.  Small alphabets
make matches dense; edits include transpositions; both prefix modes."""
import ctypes as C
import os
import random
import subprocess

import pytest

from marlin_amd import ffi
from util import ROOT, load_fixture


@pytest.fixture(scope="module")
def lev(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("lev") / "lev_check.so")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "marlin_amd", "host"),
           "-I" + os.path.join(ROOT, "marlin_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "support", "lev_check.cpp"), "-o", out, "-L" + ffi.LIB_DIR, "-lmarlin_host",
           "-Wl,-rpath," + ffi.LIB_DIR]
    subprocess.run(cmd, check=True)
    ffi.load()
    lib = C.CDLL(out)
    lib.lev_check.restype = C.c_int
    return lib


def run(lib, words, terms):
    wch = [ord(c) for w in words for c in w]; wl = [len(w) for w in words]
    tch = [ord(c) for t, _ in terms for c in t]; tl = [len(t) for t, _ in terms]; tp = [p for _, p in terms]
    m = C.c_long()
    rc = lib.lev_check((C.c_uint32 * max(1, len(wch)))(*wch), (C.c_int * len(wl))(*wl), len(wl), (C.c_uint32 * max(1, len(tch)))(*tch),
                       (C.c_int * len(tl))(*tl), (C.c_int * len(tp))(*tp), len(tl), C.byref(m))
    return rc, m.value


def edits(rng, w, alpha, n):
    w = list(w)
    for _ in range(n):
        op = rng.randint(0, 3)
        if op == 0 and w:
            w[rng.randrange(len(w))] = rng.choice(alpha)
        elif op == 1:
            w.insert(rng.randint(0, len(w)), rng.choice(alpha))
        elif op == 2 and len(w) > 1:
            del w[rng.randrange(len(w))]
        elif op == 3 and len(w) > 1:
            i = rng.randrange(len(w) - 1); w[i], w[i + 1] = w[i + 1], w[i]
    return "".join(w) or "a"


@pytest.mark.parametrize("seed", range(6))
def test_scan_equals_walk_on_random_dictionaries(lev, seed):
    rng = random.Random(seed)
    total = 0
    for alpha_n, nwords in ((2, 400), (3, 3000), (6, 3000), (26, 5000)):
        alpha = "abcdefghijklmnopqrstuvwxyz"[:alpha_n]
        words = sorted({"".join(rng.choice(alpha) for _ in range(rng.randint(1, 14))) for _ in range(nwords)})
        terms = []
        for _ in range(120):
            t = edits(rng, rng.choice(words), alpha, rng.randint(0, 3)) if rng.random() < 0.7 else \
                "".join(rng.choice(alpha) for _ in range(rng.randint(1, 18)))
            terms.append((t, rng.randint(0, 1)))
        rc, m = run(lev, words, terms)
        assert rc == 0, (seed, alpha_n)
        total += m
    assert total > 1000                                   # the comparison is not vacuous


def test_scan_equals_walk_on_the_reference_levenshtein_suite(lev):
    """the words and queries of test/robot/levenshtein.robot (its totals are pinned end to end by test_oracle_golden)"""
    fx = load_fixture("levenshtein")
    words = sorted({w for d in fx["docs"] for w in d["str"].split()})
    terms = []
    for case in fx["cases"]:
        for w in case["query"].get("q", "").split():
            terms += [(w, 0), (w, 1)]
    # unicode code points and a word at the length limit
    words += ["\u00e9cole", "\u00e9colier", "\u6771\u4eac\u90fd\u5e81\u820e", "x" * 127]
    terms += [("\u00e9cold", 0), ("\u00e9col", 1), ("\u6771\u4eac\u90fd\u5e81\u820e", 0), ("x" * 127, 0), ("x" * 126 + "y", 1)]
    rc, m = run(lev, words, terms)
    assert rc == 0 and m > 0
