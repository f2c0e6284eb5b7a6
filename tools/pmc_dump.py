#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc results database (kernels matching a substring)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, "
                  "counter_name").fetchall()
out = {}
for k, c, n, v in rows:
    if pat in k:
        out.setdefault(k[:70], {})[c] = v
for k, d in out.items():
    print(k)
    print("   " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(d.items())))
