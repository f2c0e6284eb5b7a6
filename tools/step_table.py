#!/usr/bin/env python3
"""Per-launch evidence table of the benchmark training step (one row per GEMM launch and per large streaming kernel).

  step_table.py run  <hostlog.json> [math]      the target command of rocprofv3: 6 steps of the bench step with the second
                                                stream off (kernels start in issue order), logging (family, M, N, K, tile
                                                configuration, K-slices) of every GEMM call of the last 3 steps
  step_table.py join <hostlog.json> <kt.db> [<fetch.db> <write.db> <mfma.db>]
                                                joins the log with the rocprofv3 databases of (separate) runs of that
                                                command: kernel-trace durations, FETCH_SIZE (x2, gfx950) + WRITE_SIZE, MFMA busy

Every number bench.py's `roofline` prints (family sums, per-launch TFLOP/s, traffic) can be recomputed from the table.
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GEMM_NAMES = ("k_nt_s16", "k_tn_s16", "k_rows_gemm", "k_red_gemm", "k_expand_fwd_s16", "k_expand_bwd_p_s16", "k_head_fwd", "k_head_bwd")
STREAM_NAMES = ("k_bn_act_fwd_s16", "k_bn_bwd_apply_s16", "k_bn_bwd_reduce_bits", "k_bn_act_fwd", "k_bn_bwd_apply", "k_bn_bwd_reduce")
N_STEPS, N_KEEP = 6, 3


def run(log_path, math):
    os.environ["VP3D_OVERLAP"] = "0"
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as bench.py: the package no longer sets it at import (round 6)
    import torch
    import videopose3d_amd as V
    from videopose3d_amd import dp, ops, loss as vloss
    dev = "cuda:0"
    torch.manual_seed(0)
    x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
    tgt = torch.randn(1024, 1, 17, 3, device=dev) * 0.3
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
    m.math = math
    sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)
    steps = []
    for _ in range(N_STEPS):
        ops.launch_log = []
        sync.zero_grad()
        vloss.mpjpe(m(x), tgt).backward()
        sync.sync()
        steps.append(ops.launch_log)
    torch.cuda.synchronize()
    ops.launch_log = None
    assert all(len(s) == len(steps[-1]) for s in steps[-N_KEEP:])
    json.dump({"math": math, "steps_kept": N_KEEP, "calls": steps[-1]}, open(log_path, "w"))


def _dispatches(db_path):
    """[(name, start, end)] of the profiled process in start order."""
    db = sqlite3.connect(db_path)
    return db.execute("select name, start, end from kernels order by start").fetchall()


def _last_steps(rows):
    """Split at the first kernels of a step -- the maximum over the raw input, then the fused im2row + S16 split
    (k_split_t<true>; k_im2row on the older path) --; the last N_KEEP steps."""
    starts = []
    for i, r in enumerate(rows):
        if "k_prologue_a" in r[0]:                     # round 3: the two-launch prologue opens a step
            starts.append(i)
        elif "k_im2row" in r[0] or "k_split_t<true>" in r[0]:
            starts.append(i - 1 if (i > 0 and "k_amax(" in rows[i - 1][0] and "k_split_t<true>" in r[0]) else i)
    assert len(starts) >= N_KEEP, "fewer steps than expected in the trace"
    bounds = starts[-N_KEEP:] + [len(rows)]
    return [rows[bounds[i]:bounds[i + 1]] for i in range(N_KEEP)]


def _counter_steps(db_path):
    """Per dispatch (start order) a dict counter -> value, split into the last N_KEEP steps."""
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, start, end, counter_name, value from counters_collection order by start").fetchall()
    disp, cur = [], None
    for name, st, en, ctr, val in rows:
        if cur is None or cur[1] != st or cur[0] != name:
            cur = [name, st, en, {}]
            disp.append(cur)
        cur[3][ctr] = cur[3].get(ctr, 0.0) + float(val)
    return _last_steps([(d[0], d[1], d[2], d[3]) for d in disp])


def _pick(step_rows, names):
    return [r for r in step_rows if any(n in r[0] for n in names)]


def _fmt(v, f):
    return (f % v) if v is not None else "-"


def join(log_path, kt_db, fetch_db=None, write_db=None, mfma_db=None, json_out=None):
    log = json.load(open(log_path))
    rows_json = []
    calls = log["calls"]
    steps = _last_steps(_dispatches(kt_db))
    gem = [_pick(s, GEMM_NAMES) for s in steps]
    n_expected = sum((c[3][5] if c[3] else 1) for c in calls)
    assert all(len(g) == n_expected for g in gem), ([len(g) for g in gem], n_expected)
    fet = [_pick(s, GEMM_NAMES) for s in _counter_steps(fetch_db)] if fetch_db else None
    wri = [_pick(s, GEMM_NAMES) for s in _counter_steps(write_db)] if write_db else None
    mfm = [_pick(s, GEMM_NAMES) for s in _counter_steps(mfma_db)] if mfma_db else None
    peak = 2500.0 / 3.0 if log["math"] == "f16x3" else 157.3
    print("# one row per GEMM launch of one training step (cfg3: B = 1024, arc 3,3,3,3,3, C = 1024, math %s), second stream off;"
          % log["math"])
    print("# us = mean over the last %d steps of a rocprofv3 --kernel-trace run; TFLOP/s = algorithmic 2*M*N*K / us; frac = of the"
          % N_KEEP)
    print("# roofline %.1f TFLOP/s of algorithmic work; MB_fetch = FETCH_SIZE x 2 (gfx950 correction), MB_write = WRITE_SIZE: separate"
          % peak)
    print("# --pmc passes; MB_alg = every operand once + the result once; busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clocks)")
    print("# launches below ~100 us depend on how fast the traced host refills the queue (an empty queue drops the clocks): three\n"
          "# collections of the same build on three boxes agree within 3 % on the large launches and differ by 15-40 % on the\n"
          "# small ones; bench.py's HIP-event numbers (untraced) are the reference for those")
    print("%-12s %7s %6s %6s %4s %3s  %-34s %9s %8s %6s %9s %9s %8s %6s %6s" % (
        "family", "M", "N", "K", "cfg", "ks", "kernel", "us", "TFLOP/s", "frac", "MB_fetch", "MB_write", "MB_alg", "busy", "GHz"))
    pos, fam = 0, {}
    for family, flops, nbytes, shape in calls:
        nk = shape[5] if shape else 1
        us = sum(sum((g[pos + j][2] - g[pos + j][1]) for j in range(nk)) for g in gem) / len(gem) / 1e3
        name = gem[-1][pos][0].replace("void ", "").replace("vp3d::(anonymous namespace)::", "").split("(")[0][:34]
        fb = wb = busy = ghz = None
        if fet:
            fb = sum(sum(f[pos + j][3].get("FETCH_SIZE", 0.0) for j in range(nk)) for f in fet) / len(fet) * 1024 * 2 / 1e6
        if wri:
            wb = sum(sum(w[pos + j][3].get("WRITE_SIZE", 0.0) for j in range(nk)) for w in wri) / len(wri) * 1024 / 1e6
        if mfm:
            gui = sum(sum(q[pos + j][3].get("GRBM_GUI_ACTIVE", 0.0) for j in range(nk)) for q in mfm) / 8.0
            bsy = sum(sum(q[pos + j][3].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for j in range(nk)) for q in mfm)
            ns = sum(sum(q[pos + j][2] - q[pos + j][1] for j in range(nk)) for q in mfm)
            busy = bsy / 1024.0 / gui if gui else None
            ghz = gui / ns if ns else None
        tf = flops / us / 1e6
        m_, n_, k_, cfg, ks = (shape[0], shape[1], shape[2], shape[3], shape[4]) if shape else (0, 0, 0, "-", 0)
        print("%-12s %7d %6d %6d %4s %3d  %-34s %9.1f %8.1f %6.3f %9s %9s %8.1f %6s %6s" % (
            family, m_, n_, k_, cfg, ks, name, us, tf, tf / peak, _fmt(fb, "%.1f"), _fmt(wb, "%.1f"), nbytes / 1e6,
            _fmt(busy, "%.3f"), _fmt(ghz, "%.2f")))
        rows_json.append({"family": family, "M": m_, "N": n_, "K": k_, "cfg": cfg, "k_slices": ks, "kernel": name, "us": us,
                          "tflops": tf, "fetch_bytes": None if fb is None else fb * 1e6, "write_bytes": None if wb is None else wb * 1e6,
                          "algorithmic_bytes": nbytes, "mfma_busy": busy, "ghz": ghz})
        f = fam.setdefault(family, [0.0, 0.0, 0])
        f[0] += flops
        f[1] += us
        f[2] += 1
        pos += nk
    print("# family sums (the numbers of bench.py's `kernels`):")
    for k, (fl, us, n) in fam.items():
        print("#   %-12s %2d launches  %8.1f us  %6.1f TFLOP/s  frac %.3f" % (k, n, us, fl / us / 1e6, fl / us / 1e6 / peak))
    tot_fl, tot_us = sum(v[0] for v in fam.values()), sum(v[1] for v in fam.values())
    print("#   all GEMMs    %8.1f us  %6.1f TFLOP/s  frac %.3f   (kernel durations only: under the tracer the host cannot keep the "
          "queue full, so a step SPAN is not meaningful here -- bench.py times the step)" % (tot_us, tot_fl / tot_us / 1e6, tot_fl / tot_us / 1e6 / peak))
    if json_out:
        json.dump({"math": log["math"], "what": "one row per GEMM launch of one cfg3 training step (tools/step_table.py join): "
                   "rocprofv3 kernel-trace durations, FETCH_SIZE x2 / WRITE_SIZE and MFMA busy from separate --pmc passes",
                   "launches": rows_json}, open(json_out, "w"), indent=1)
    print("# HBM-bound producers of the same steps (mean us; GB/s against their algorithmic bytes is in bench.py's roofline.streaming):")
    agg = {}
    for s in steps:
        for r in _pick(s, STREAM_NAMES):
            nm = r[0].replace("void ", "").replace("vp3d::(anonymous namespace)::", "").split("(")[0]
            agg.setdefault(nm, []).append((r[2] - r[1]) / 1e3)
    for nm, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("#   %-40s %3d launches/step  %8.1f us/step  (largest %.1f us)" % (nm, len(v) // len(steps), sum(v) / len(steps), max(v)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "f16x3")
    else:
        join(*sys.argv[2:])
