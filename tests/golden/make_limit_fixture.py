#!/usr/bin/env python3
"""Re-encodes the reference's test/robot/limit.robot (getFields / highlightFields / highlightSource known answers) as
tests/golden/robot_limit.json: the document and index settings the suite posts (doc.json, docset.json), and per query
the <mark> counts and key checks it asserts on hits[0].  DATA only.  Run in the build container:

    python tests/golden/make_limit_fixture.py
"""
import json
import os
import re
import sys

REF = "/root/reference/test/robot"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    if not os.path.isdir(REF):
        print("reference not present; the fixture is already committed", file=sys.stderr)
        return 0
    lines = open(os.path.join(REF, "limit.robot")).read().splitlines()
    cases, settings_now, cur = [], {}, None
    for ln in lines:
        c = [x for x in re.split(r"\s{2,}|\t", ln.strip()) if x]
        if len(c) >= 3 and c[-3] == "POST" and c[-2].endswith("/settings") and c[-1].startswith("{"):
            settings_now = dict(settings_now, **json.loads(c[-1]))        # index-level query settings posted so far
        elif len(c) >= 3 and c[-3] == "POST" and c[-2].endswith("/query"):
            cur = {"query": json.loads(c[-1]), "settings": dict(settings_now), "checks": []}
            cases.append(cur)
        elif cur is not None and "Evaluate" in c:
            expr = c[c.index("Evaluate") + 1]
            m = re.match(r"str\(\$\{res\}\[body\]\[hits\]\[0\](\[_highlight\])?\)\.count\(\"<mark>\"\) == (\d+)", expr)
            if m:
                cur["checks"].append({"where": "_highlight" if m.group(1) else "hit", "marks": int(m.group(2))})
            m = re.match(r"'_highlight' in \$\{res\}\[body\]\[hits\]\[0\]", expr)
            if m:
                cur["checks"].append({"has_highlight_key": None})
        elif cur is not None and len(c) >= 3 and c[0] == "Should be Equal" and cur["checks"] and "has_highlight_key" in cur["checks"][-1]:
            cur["checks"][-1]["has_highlight_key"] = c[2] == "${TRUE}"
        elif cur is not None and len(c) >= 3 and c[0] == "Integer" and c[1] == "$.totalHits":
            cur["checks"].append({"totalHits": int(c[2])})
    fx = {"source": "test/robot/limit.robot (+ doc.json, docset.json)",
          "doc": json.load(open(os.path.join(REF, "doc.json"))),
          "index_settings": json.load(open(os.path.join(REF, "docset.json"))),
          "cases": [c for c in cases if c["checks"]]}
    with open(os.path.join(OUT, "robot_limit.json"), "w") as f:
        json.dump(fx, f, indent=1)
        f.write("\n")
    print(len(fx["cases"]), "cases,", sum(len(c["checks"]) for c in fx["cases"]), "checks")
    return 0


if __name__ == "__main__":
    sys.exit(main())
