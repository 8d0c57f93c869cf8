"""diagnostic: stage times of a 128-query batch on the reference's data_gen corpus (multi-valued facet and number fields, three
facet fields, one shard) -- BASELINE configs[1]'s query and a few neighbours (not part of the product)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import datagen_index
from marlin_amd import GpuEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ix, _ = datagen_index(n)
eng = GpuEngine(ix)
shapes = {"configs[1]": {"q": "aaaaaaa bbbbbbb ccccccc", "hitsPerPage": 10, "maxFacetResults": 10},
          "typo + rankBy": {"q": "aaaaaab bbbbbbc", "rankBy": {"num": "desc"}, "hitsPerPage": 10, "maxFacetResults": 10},
          "numlist filter": {"q": "aaaaaaa", "filter": {"numlist": {"$gte": 25000}}, "maxFacetResults": 10},
          "aggs": {"q": "aaaaaaa bbbbbbb", "aggs": {"a": {"avg": {"field": "num"}}, "mx": {"max": {"field": "numlist"}}, "c": {"cardinality": {"field": "facetlist"}}}},
          "facets only": {"q": "aa", "hitsPerPage": 0, "maxFacetResults": 50}}
for name, jq in shapes.items():
    pqs = [ix.query(dict(jq, fullScan=True)) for _ in range(128)]
    prep = eng.prepare(pqs)
    for _ in range(2):
        eng.free_raw(eng.execute_prepared(prep))
    eng.set_profiling(True)
    outs = eng.execute_prepared(prep)
    hits = outs[0].num_hits
    eng.free_raw(outs)
    t = eng.timing()
    eng.set_profiling(False)
    print("%-16s hits %8d  total %.3f ms  %s" % (name, hits, sum(x["ms"] for x in t["stages"].values()), {k: round(x["ms"], 3) for k, x in t["stages"].items() if x["ms"] > 0.02}), flush=True)
eng.close(); ix.close()
