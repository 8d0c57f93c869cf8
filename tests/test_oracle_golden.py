"""Pins the CPU oracle against the reference's own known answers (test/robot/*.robot),
re-encoded as data under tests/golden/ by make_robot_fixtures.py."""
import pytest

from marlin_amd import Index, QueryError
from oracle import orc_py
from util import ROBOT_SUITES, jpath, load_fixture


@pytest.mark.parametrize("suite", ROBOT_SUITES)
def test_robot_suite(suite):
    fx = load_fixture(suite)
    ix = Index(fx["settings"], num_shards=fx["numShards"]).add_documents(fx["docs"])
    try:
        checked = 0
        for case in fx["cases"]:
            if case["status"] == 400:
                with pytest.raises(QueryError):
                    ix.query(case["query"])
                continue
            pq = ix.query(case["query"])
            res = orc_py.search(ix, pq)
            for path, want in case["expect"]:
                got = jpath(res, path)
                if isinstance(want, float):
                    assert got == pytest.approx(want), (case["query"], path)
                else:
                    assert got == want, (case["query"], path, got, want)
                checked += 1
            pq.close()
        assert checked == sum(len(c["expect"]) for c in fx["cases"] if c["status"] != 400)
    finally:
        ix.close()
