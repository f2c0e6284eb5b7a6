#!/usr/bin/env python3
"""Time of an S16 GEMM launch as a function of K at fixed M x N (whole rounds of tiles): the slope is the main-loop rate,
the intercept the per-tile overhead (row table, first DMA latency, epilogue).  usage: s16_kscan.py [cfg ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


cfgs = [int(a) for a in sys.argv[1:]] or [20, 22, 24, 25]
n = 1024
for m in (65536, 27648):
    for form in ("plain", "stats"):
        print("M = %d, N = %d, %s epilogue   (ms | TF)" % (m, n, "fp32 out + BN slab statistics" if form == "stats" else "fp32 out"))
        for k in (128, 256, 512, 1024, 2048, 3072):
            spec = ConvSpec(k, n, 1)
            xs = S.split(torch.relu(torch.randn(1, m, k, device=dev)))
            ws = S.split(torch.randn(n, k, device=dev) * 0.02)
            st = ops.stat_buffers(m, n, dev) if form == "stats" else None
            line = "  K %5d" % k
            for cfg in cfgs:
                ms = timeit(lambda: S.conv_nt(xs, ws, spec, cfg=cfg, splits=1, stats=st))
                line += " | c%d %7.4f %6.1f" % (cfg, ms, 2.0 * m * n * k / ms / 1e9)
            print(line, flush=True)
