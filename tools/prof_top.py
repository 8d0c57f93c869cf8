"""prints the per-kernel averages of a rocprofv3 --kernel-trace --stats results db (diagnostic)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print("%-60s calls %5d avg_us %9.2f pct %5.1f" % (name.split("(")[0][:60], calls, avg, pct))
