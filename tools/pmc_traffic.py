#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each collected in its own run with --kernel-trace, as
MI355X_MICROARCH.md prescribes: the two counters do not fit one pass) into a per-kernel HBM-traffic table.

    python tools/pmc_traffic.py <fetch_results.db> <write_results.db> <out.json> [<out.txt>]

Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB per dispatch.  gfx950 correction (same guide, HBM
section): FETCH_SIZE counts 128-B requests of wide coalesced reads (16 B/lane global_load and LDS-DMA alike) as
64 B -> doubled here.  WRITE_SIZE is used as reported.  bench.py reads <out.json> for `roofline.traffic`.
"""
import json
import sqlite3
import sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                      "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (int(r[1]), float(r[2])) for r in rows}


def main():
    fetch_db, write_db, out_json = sys.argv[1:4]
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    table = {}
    for name in sorted(set(f) | set(w)):
        cf, sf = f.get(name, (0, 0.0))
        cw, sw = w.get(name, (0, 0.0))
        fetch = 2.0 * sf * 1024.0 / cf if cf else None           # bytes per launch, gfx950 x2 correction
        write = sw * 1024.0 / cw if cw else None
        table[name] = dict(launches_fetch_pass=cf, launches_write_pass=cw, fetch_bytes_per_launch=fetch,
                           write_bytes_per_launch=write,
                           hbm_bytes_per_launch=(fetch or 0.0) + (write or 0.0))
    meta = dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace)",
                correction="FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported",
                unit="bytes per launch (average over the launches of the kernel in the profiled command)")
    json.dump(dict(meta=meta, kernels=table), open(out_json, "w"), indent=1)
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w") as o:
            o.write("# %s\n# %s\n" % (meta["source"], meta["correction"]))
            o.write("%-100s %7s %14s %14s %14s\n" % ("kernel", "calls", "fetch_MB(x2)", "write_MB", "total_MB"))
            for name, t in sorted(table.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * max(kv[1]["launches_fetch_pass"], 1)):
                o.write("%-100s %7d %14.2f %14.2f %14.2f\n" % (name[:100], t["launches_fetch_pass"],
                                                               (t["fetch_bytes_per_launch"] or 0) / 1e6,
                                                               (t["write_bytes_per_launch"] or 0) / 1e6,
                                                               t["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
