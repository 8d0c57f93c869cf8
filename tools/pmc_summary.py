"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files -> per-kernel KB per launch (JSON).
usage: pmc_summary.py <fetch.csv> <write.csv> <out.json>
Only full-batch launches are averaged: for every kernel the launches whose grid is the most common one."""
import collections, csv, json, sys


def per_kernel(path, counter):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        by[r["Kernel_Name"].split("(")[0]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for k, v in by.items():
        grid = collections.Counter(g for g, _ in v).most_common(1)[0][0]
        vals = [x for g, x in v if g == grid]
        out[k] = round(sum(vals) / len(vals), 1)
    return out


def main():
    f, w, o = sys.argv[1:4]
    fe, wr = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
    doc = {"_about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py --no-cpu "
                     "--no-latency --inflight 1 --steps 1 --warmup 0 --sub 3`; KB per launch, averaged over the batched launches "
                     "(1024 squeries). HBM bytes = 2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md) "
                     "+ WRITE_SIZE.",
           "kernels": {k: {"FETCH_SIZE_KB_avg": fe.get(k, 0.0), "WRITE_SIZE_KB_avg": wr.get(k, 0.0)} for k in sorted(set(fe) | set(wr))}}
    for k, v in doc["kernels"].items():
        v["hbm_bytes_per_launch"] = int(2 * v["FETCH_SIZE_KB_avg"] * 1024 + v["WRITE_SIZE_KB_avg"] * 1024)
    if "k_rank_tile" in doc["kernels"]:
        doc["k_rank_tile_hbm_bytes_per_launch"] = doc["kernels"]["k_rank_tile"]["hbm_bytes_per_launch"]
    json.dump(doc, open(o, "w"), indent=1)
    print(json.dumps({k: v for k, v in doc["kernels"].items() if k.startswith("k_")}))


if __name__ == "__main__":
    main()
