#!/usr/bin/env python3
"""Per-position kernel durations of the LAST training step in a rocprofv3 kernel-trace database (second stream off): the
launches between the last two k_prologue_a, one line each (name, us) -- for side-by-side diffs of two builds / modes."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
try:
    rows = db.execute("select name, start, end, grid_size_x, workgroup_size_x from kernels order by start").fetchall()
except sqlite3.OperationalError:
    rows = [r + (0, 0) for r in db.execute("select name, start, end from kernels order by start").fetchall()]
marks = [i for i, r in enumerate(rows) if "k_prologue_a" in r[0]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lo = marks[-nsteps - 1]
steps = [rows[marks[-k - 1]:marks[-k]] for k in range(nsteps, 0, -1)]
n = min(len(s) for s in steps)
tot = 0.0
for i in range(n):
    name = steps[0][i][0].replace("vp3d::", "").replace("(anonymous namespace)::", "").replace("void ", "").replace("vp3d::mma::", "").replace("(RowsGemmArgs)", "")
    us = sum((s[i][2] - s[i][1]) for s in steps) / len(steps) / 1e3
    tot += us
    g = steps[0][i][3] // max(1, steps[0][i][4]) if steps[0][i][4] else 0
    print("%3d %-70s wg %6d %9.1f" % (i, name[:70], g, us))
print("sum of kernel durations: %.1f us over %d launches" % (tot, n))
