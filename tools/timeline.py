#!/usr/bin/env python3
"""Timeline of the last training step in a rocprofv3 kernel-trace database: start offset, duration, queue, kernel."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
# a step starts at the maximum over the raw input + the fused im2row / S16 split (k_im2row on the older path)
starts = []
for i, r in enumerate(rows):
    if "k_prologue_a" in r[0]:                         # round 3: the two-launch prologue opens a step
        starts.append(i)
    elif "k_im2row" in r[0] or "k_split_t<true>" in r[0]:
        starts.append(i - 1 if (i > 0 and "k_amax(" in rows[i - 1][0] and "k_split_t<true>" in r[0]) else i)
lo = starts[-1]
t0 = rows[lo][1]
last_end = t0
busy = 0.0
for r in rows[lo:]:
    name = r[0].replace("vp3d::(anonymous namespace)::", "").replace("void ", "")[:58]
    gap = (r[1] - last_end) / 1e3
    print("%9.1f us  +%7.1f  q%-3s %8.1f us  %s" % ((r[1] - t0) / 1e3, gap if gap > 0 else 0.0, r[3], (r[2] - r[1]) / 1e3, name))
    last_end = max(last_end, r[2])
print("step span %.1f us" % ((last_end - t0) / 1e3))
