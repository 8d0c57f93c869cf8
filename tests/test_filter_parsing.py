"""parse_filter's accept / reject decisions (filter.c, reached from index_json_query) on the reference's own
test/robot/filter.robot: 27 queries with the HTTP status the reference answers (200 = parsed, 400 = rejected).
Filter parsing is host logic in front of the path (the lowered filter program is what the GPU sees); CPU only."""
import pytest

from marlin_amd import Index, QueryError
from util import load_fixture


def test_filter_robot_statuses():
    fx = load_fixture("filter")
    docs = [{"id": i, "num": 100 - i, "numlist": [i, i + 1], "bool": i % 10 == 0, "str": "aaaaaaa bbbbbbb", "strlist": ["ccccccc"],
             "facet": ["testme", "test", "news"][i % 3], "facetlist": ["one", "two"], "facetonly": ["test", "two", "three"][i % 3], "spell": "eeeeeee"}
            for i in range(1, 101)]
    ix = Index(fx["settings"], num_shards=1).add_documents(docs)
    try:
        seen = {200: 0, 400: 0}
        for case in fx["cases"]:
            if case["status"] == 200:
                ix.query(case["query"])                       # must parse
            elif case["status"] == 400:
                with pytest.raises(QueryError):
                    ix.query(case["query"])
            seen[case["status"]] = seen.get(case["status"], 0) + 1
        assert seen[200] >= 10 and seen[400] >= 10
    finally:
        ix.close()
