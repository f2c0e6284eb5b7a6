#!/usr/bin/env python3
"""MFMA utilisation and effective clock per kernel from one rocprofv3 PMC pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace).

  effective clock  = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time          (the counter is summed over the 8 XCDs)
  MFMA busy        = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8)
                     = share of the clock cycles in which a SIMD's matrix pipe was executing
  fp32 roofline at the measured clock = 256 CUs x 256 FLOP/clk x effective clock
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start) from counters_collection "
                  "group by kernel_name, counter_name").fetchall()
tab = {}
for name, ctr, val, n, dur in rows:
    t = tab.setdefault(name, {})
    t[ctr] = val
    t["_n"], t["_ns"] = n, dur
print("%-70s %6s %10s %9s %9s %9s %9s" % ("kernel", "calls", "total_ms", "clk_GHz", "mfma_busy", "wait_any", "wait_inst"))
for name, t in sorted(tab.items(), key=lambda kv: -kv[1]["_ns"])[:16]:
    gui = t.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui <= 0:
        continue
    clk = gui / t["_ns"]
    busy = t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / gui
    wc = t.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print("%-70s %6d %10.3f %9.3f %9.3f %9.3f %9.3f" % (
        name.replace("void ", "").replace("vp3d::(anonymous namespace)::", "")[:70], t["_n"], t["_ns"] / 1e6, clk, busy,
        t.get("SQ_WAIT_ANY", 0.0) / wc, t.get("SQ_WAIT_INST_ANY", 0.0) / wc))
