#!/usr/bin/env python3
"""K-slice count of the rows-form weight-gradient GEMM (k_tn_s16 + vp3d_wgrad_reduce): GEMM + reduce time per slice count for
the shapes of the cfg3 step, HIP events, random operands.   python tools/wgrad_splits.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops_s16 as S  # noqa: E402

dev = "cuda:0"
C = 1024
torch.manual_seed(0)
picked = S._wgrad_rows_splits


def timed(fn, n_it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_it * 1e3


for t_o, taps in ((27, 3), (27, 1), (9, 3), (9, 1), (3, 3), (3, 1), (1, 3), (1, 1)):
    B = 1024
    m = B * t_o
    dy = S.split(torch.randn(B, t_o, C, device=dev) * 1e-3)
    x = S.split(torch.randn(B, t_o * taps, C, device=dev))
    out = torch.empty(C, C, taps, device=dev)
    cur = picked(m, C, taps * C)
    row = []
    for s in (1, 2, 3, 4, 5, 6, 8, 10, 12, 15, 16, 20, 24, 32):
        if s > 1 and ((m + 31) // 32) // s < 6:
            continue
        S._wgrad_rows_splits = lambda *_a, s=s: s
        row.append((timed(lambda: S.wgrad_rows(dy, x, C, C, taps, out=out)), s))
    S._wgrad_rows_splits = picked
    best = min(row)
    print("M %6d x N %5d: picked s=%-2d %6.1f us   best s=%-2d %6.1f us   all: %s" % (
        m, taps * C, cur, dict((s, t) for t, s in row)[cur], best[1], best[0], " ".join("%d:%.0f" % (s, t) for t, s in row)), flush=True)
