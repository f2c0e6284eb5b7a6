#!/usr/bin/env python3
"""Print a per-kernel summary (calls, total, avg, min, max) from a rocprofv3 rocpd sqlite database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total kernel time %.3f ms" % tot)
print("%-90s %6s %10s %6s %10s %9s %10s" % ("kernel", "calls", "total_ms", "%", "avg_us", "min_us", "max_us"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-90s %6d %10.3f %6.1f %10.1f %9.1f %10.1f" % (r[0][:90], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5]))
