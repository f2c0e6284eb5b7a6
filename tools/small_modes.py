#!/usr/bin/env python3
"""Train-step latency of small configurations in both arithmetics (where does the split-fp16 engine stop paying?)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402

dev = "cuda:0"


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for fw in ([3, 3, 3], [3, 3, 3, 3, 3]):
    for b in (16, 64, 128, 256, 512, 1024):
        rf = 3 ** len(fw)
        x = (torch.randn(b, rf, 17, 2, device=dev) * 0.5).clamp(-1, 1)
        tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3
        line = "arc %-10s B=%5d train:" % (",".join(map(str, fw)), b)
        for math in ("f32", "f16x3"):
            m = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.25, channels=1024).to(dev).train()
            m.math = math

            def step():
                m.zero_grad(set_to_none=True)
                torch.mean(torch.norm(m(x) - tgt, dim=3)).backward()
            line += "  %s %7.3f ms" % (math, timed(step, 10))
            del m
        print(line, flush=True)
for t in (243, 500, 2000):
    x = (torch.randn(2, t + 242, 17, 2, device=dev) * 0.5).clamp(-1, 1)       # run.py evaluation: one sequence (+ flipped copy)
    line = "arc 3,3,3,3,3 eval B=2 T_out=%5d:" % t
    for math in ("f32", "f16x3"):
        e = V.TemporalModel(17, 2, 17, [3, 3, 3, 3, 3], channels=1024).to(dev).eval()
        e.math = math
        with torch.no_grad():
            line += "  %s %7.3f ms" % (math, timed(lambda: e(x), 10))
        del e
    print(line, flush=True)
