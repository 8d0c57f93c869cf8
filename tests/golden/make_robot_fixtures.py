#!/usr/bin/env python3
"""Re-encodes the reference's Robot Framework known-answer suites as JSON fixtures.

Reads /root/reference/test/robot/<suite>.robot (the reference's own black-box REST tests,
SURVEY.md section 4) and writes tests/golden/robot_<suite>.json holding only DATA: the index
settings, shard count, the documents posted, and each query with the values the suite asserts
on the response.  Run in the build container (the reference is not present on the GPU box):

    python tests/golden/make_robot_fixtures.py
"""
import json
import os
import re
import sys

REF = "/root/reference/test/robot"
OUT = os.path.dirname(os.path.abspath(__file__))
SUITES = ["rank", "rank2", "rank3", "rank4", "rank5", "rank6", "rank7", "rank_terms",
          "levenshtein", "terms", "filter_apply", "aggs", "facet", "sort"]

COMMON_INDEX = {"name": "testindex", "numShards": 5}
COMMON_SETTINGS = {"indexedFields": ["str", "strlist", "facet", "facetlist", "spell", "id", "num", "numlist", "bool"],
                   "facetFields": ["facet", "facetlist", "facetonly"]}


def cells(line):
    return [c for c in re.split(r"\s{2,}|\t", line.strip()) if c != ""]


def parse(path):
    settings, index = dict(COMMON_SETTINGS), dict(COMMON_INDEX)
    docs, cases = [], []
    lines = open(path).read().splitlines()
    i = 0
    cur = None
    test_name = ""
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("${settings}"):
            settings = json.loads(cells(ln)[1])
        elif ln.startswith("${index}"):
            index = json.loads(cells(ln)[1])
        elif ln and not ln[0].isspace() and not ln.startswith("*") and not ln.startswith("$"):
            test_name = ln.strip()
        elif "Set Variable" in ln and "@{json_data}" in ln:
            buf = []
            i += 1
            while i < len(lines) and lines[i].strip().startswith("..."):
                buf.append(lines[i].strip()[3:].strip())
                i += 1
            docs.extend(json.loads(" ".join(buf)))
            continue
        else:
            c = cells(ln)
            if len(c) >= 3 and c[0] == "POST" and c[1].endswith("/query") and "/indexes/" in c[1]:
                cur = {"test": test_name, "query": json.loads(c[2]), "status": None, "expect": []}
                cases.append(cur)
            elif len(c) >= 2 and c[0] in ("POST", "GET", "DELETE", "PUT", "PATCH"):
                cur = None
            elif cur is not None and len(c) >= 3 and c[0] in ("Integer", "Number", "String"):
                if c[1] == "response status":
                    cur["status"] = int(c[2])
                else:
                    val = c[2]
                    if c[0] in ("Integer", "Number"):
                        val = float(val) if "." in val else int(val)
                    cur["expect"].append([c[1], val])
        i += 1
    return {"source": "test/robot/" + os.path.basename(path), "settings": settings,
            "numShards": index.get("numShards", 5), "docs": docs, "cases": cases}


def main():
    if not os.path.isdir(REF):
        print("reference not present; fixtures are already committed", file=sys.stderr)
        return 0
    for s in SUITES:
        fx = parse(os.path.join(REF, s + ".robot"))
        with open(os.path.join(OUT, "robot_%s.json" % s), "w") as f:
            json.dump(fx, f, indent=1, sort_keys=False)
            f.write("\n")
        print(s, len(fx["docs"]), "docs", len(fx["cases"]), "cases",
              sum(len(c["expect"]) for c in fx["cases"]), "assertions")
    return 0


if __name__ == "__main__":
    sys.exit(main())
