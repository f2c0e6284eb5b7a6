#!/usr/bin/env python3
"""Sustained rate of the big forward GEMM per tile configuration: the same launch repeated back to back for ~40 ms (the clocks
settle where the power drawn lets them), not the best of a few isolated launches.  Needs an experiments build for 26 / 24 / 25:
    python tools/ab_lib.py videopose3d_amd/libvp3d_exp.so tools/sustained_cfg.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
B, C = 1024, 1024
for taps in (3, 1):
    x = S.split(torch.randn(B, 27 * taps, C, device=dev))
    w = S.split(torch.randn(C, taps * C, device=dev) * 0.03)
    spec = ConvSpec(C, C, taps, 1, taps)
    m = B * 27
    flops = 2.0 * m * C * taps * C
    for cfg in (22, 28, 26, 24, 25, 20):
        try:
            slab = S.stat_slab_rows(cfg, 1)
            st = ops.stat_buffers(m, C, dev, slab)

            def f():
                return S.conv_nt(x, w, spec, stats=st, cfg=cfg, splits=1, stat_slab=slab)
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            res = []
            for n in (1, 100):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(n):
                    f()
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / n * 1e3)
            print("K = %4d  cfg %3d   single launch %7.1f us = %5.1f TFLOP/s   100 back to back %7.1f us = %5.1f TFLOP/s" % (
                taps * C, cfg, res[0], flops / res[0] / 1e6, res[1], flops / res[1] / 1e6), flush=True)
        except Exception as e:  # noqa: BLE001
            print("K = %4d  cfg %3d   %s" % (taps * C, cfg, repr(e)[:100]), flush=True)
