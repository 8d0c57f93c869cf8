"""the oracle port (oracle/orc_squery.c, 1 shard, 1 thread) timed on the data_gen-100k queries that
BASELINE.md section 2 holds reference-binary timings for.  Prints a markdown table.  CPU only."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import datagen_index
from oracle import orc_py

QUERIES = [
    ({"q": "aaaaaab bbbbbbc", "fullScan": True, "rankBy": {"num": "desc"}, "hitsPerPage": 10}, 85.8),
    ({"q": "aaaaaab bbbbbbc", "rankBy": {"num": "desc"}, "hitsPerPage": 10}, 8.5),
    ({"q": "aaaaaaa bbbbbbb ccccccc", "fullScan": True, "rankBy": {"num": "desc"}, "hitsPerPage": 10}, 69.2),
    ({"q": "aa", "hitsPerPage": 10, "fullScan": True}, 102.8),
    ({"q": "", "filter": {"num": {"$gte": 1000, "$lt": 21000}}, "aggs": {"a": {"avg": {"field": "num"}}}, "fullScan": True}, 23.6),
]


def main():
    ix, _ = datagen_index(100000)
    model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    print("host: %s, %d cores visible; 1 thread, 50 runs per row after 5 warm-up runs" % (model, os.cpu_count()))
    print("| query | totalHits | port p50 (ms) | port p99 (ms) | of which term lookup (ms) | reference p50 (ms, section 2) |")
    print("|---|---|---|---|---|---|")
    for jq, ref_ms in QUERIES:
        pq = ix.query(jq)
        ts, lk = [], []
        for i in range(55):
            orc_py.lookup_time(True)
            t0 = time.perf_counter()
            r = orc_py.search(ix, pq)
            t1 = time.perf_counter()
            if i >= 5:
                ts.append((t1 - t0) * 1e3); lk.append(orc_py.lookup_time(True) * 1e3)
        ts.sort(); lk.sort()
        print("| `%s` | %d | %.1f | %.1f | %.2f | %.1f |" % (json.dumps(jq), r["totalHits"], ts[len(ts) // 2], ts[int(len(ts) * 0.99)], lk[len(lk) // 2], ref_ms))
    ix.close()


if __name__ == "__main__":
    main()
