"""Per-kernel statistics from a rocprofv3 rocpd database (run_results.db): name, launches, total and average time.
Usage: python scripts/rocpd_stats.py <db> [steps]   (steps: divide counts and totals by it)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("total kernel time %.3f ms, %.1f launches (per step: /%g)" % (tot / 1e6 / steps, sum(r[1] for r in rows) / steps, steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-100s %7.1f %9.3f ms  avg %8.1f us  max %8.1f us" % (r[0][:100], r[1] / steps, r[2] / 1e6 / steps, r[2] / r[1] / 1e3, r[4] / 1e3))
