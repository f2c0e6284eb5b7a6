#!/usr/bin/env python3
"""Tile configuration x K slices on the small launches of the cfg3 step (GEMM + finishing pass, HIP events around the whole call):
128 x 128 (20), 160 x 256 (29), 224 x 256 (28) with 1 .. 8 slices, forward form (+ statistics) -- what should the planner pick?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [("3072 x 1024 x 3072", 1024, 9, ConvSpec(1024, 1024, 3, 1, 3)), ("3072 x 1024 x 1024", 1024, 3, ConvSpec(1024, 1024, 1)),
          ("1024 x 3072 x 1024", 1024, 1, ConvSpec(1024, 3072, 1)), ("1024 x 1024 x 3072", 1024, 3, ConvSpec(1024, 1024, 3, 1, 3)),
          ("1024 x 1024 x 1024", 1024, 1, ConvSpec(1024, 1024, 1)), ("3072 x 3072 x 1024", 1024, 3, ConvSpec(1024, 3072, 1)),
          ("9216 x 1024 x 1024", 1024, 9, ConvSpec(1024, 1024, 1))]
for name, b, t, spec in shapes:
    x = torch.relu(torch.randn(b, t, spec.c_in, device=dev))
    w = torch.randn(spec.c_out, spec.c_in, spec.taps, device=dev) * 0.02
    xs, ws = S.split(x), S.split(ops.pack_weight(w))
    m = b * spec.t_out(t)
    k = spec.taps * spec.c_in
    row = []
    best = (1e9, None)
    for cfg in (20, 29, 28):
        for sp in (1, 2, 3, 4, 6, 8):
            if sp > 1 and k // 32 // sp < 6:
                continue
            slab = S.stat_slab_rows(cfg, sp)
            st = ops.stat_buffers(m, spec.c_out, dev, slab)
            us = timeit(lambda: S.conv_nt(xs, ws, spec, stats=st, cfg=cfg, splits=sp, stat_slab=slab))
            row.append("%d/%d:%5.1f" % (cfg, sp, us))
            if us < best[0]:
                best = (us, (cfg, sp))
    print("%-20s best %s %.1f us | planner %s mix %s | %s" % (name, best[1], best[0], S.plan(m, spec.c_out, k),
                                                           S.plan(m, spec.c_out, k, mix=True), "  ".join(row)), flush=True)
