"""per-kernel averages of every counter in a rocprofv3 --pmc counter_collection.csv (full-batch launches only)"""
import collections, csv, sys
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    by[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
names = sorted({c for k in by for c in by[k]})
print("%-28s" % "kernel" + "".join("%22s" % n for n in names))
for k in sorted(by):
    row = []
    for n in names:
        v = by[k].get(n, [])
        if not v:
            row.append(0.0); continue
        grid = collections.Counter(g for g, _ in v).most_common(1)[0][0]
        vals = [x for g, x in v if g == grid]
        row.append(sum(vals) / len(vals))
    print("%-28s" % k[:28] + "".join("%22.0f" % x for x in row))
