"""Seeded randomised differential test: generated queries (typos, prefixes, 0-4 words, filters, sort / rankBy,
pages, facets with small and large cut-offs, aggregations incl. cardinality, facets-only) on the reference's
test corpus (multi-valued facets), on a two-shard Zipf corpus with a sparse 20000-value facet field and on a
four-shard corpus with repeated words, floats and custom rank rule sets; the GPU
path must equal the CPU oracle per shard (bitmap words, every rank record, top-2N, facets, aggs) and merged.
tools/fuzz_parity.py runs the same generator for as long as one likes (45000 queries were clean)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [3, 2024])
def test_generated_queries_match_the_oracle(seed):
    assert fuzz_parity.main(150, seed) == 0


def test_generated_queries_in_the_default_mode():
    """the as-shipped mode: shards above fullScanThreshold hits are ranked on a sample (tools/fuzz_default_mode.py)"""
    import fuzz_default_mode
    bad, n, fast = fuzz_default_mode.main(120, 7)
    assert bad == 0 and n > 80 and fast > 20


@pytest.mark.parametrize("run", [2, 4])
def test_generated_queries_with_formula_workgroups_over_runs_of_windows(run):
    """k_formula takes runs of consecutive windows per workgroup only when a launch is large (mgk_formula); here the run
    length is forced, so that the small corpora of the generator (three windows per shard: a full run and a ragged one
    for run = 2) go through that loop as well.  A process of its own: the library reads the variable once."""
    import subprocess
    env = dict(os.environ, MGPU_FORMULA_RUN=str(run))
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    r = subprocess.run([sys.executable, tool, "100", str(50 + run)], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 + 0 + 0 mismatches" in r.stdout
