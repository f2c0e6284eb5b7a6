#!/usr/bin/env python3
"""Small (launch-latency-bound) training steps: the exact-fp32 engine (what engine.use_s16 picks below its size threshold),
the split-fp16 engine with per-layer launches, and the split-fp16 engine with the persistent tail (VP3D_TAIL=1) -- does the
persistent kernel pay where the step is bound by launches rather than by the matrix pipes?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import engine  # noqa: E402

dev = "cuda:0"
engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})


def timed(fn, n):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for fw in ([3, 3, 3], [3, 3, 3, 3, 3]):
    for b in (16, 64, 128, 256, 512):
        rf = 3 ** len(fw)
        x = (torch.randn(b, rf, 17, 2, device=dev) * 0.5).clamp(-1, 1)
        tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3
        line = "arc %-10s B=%4d train:" % (",".join(map(str, fw)), b)
        for math, tail in (("f32", "0"), ("f16x3", "0"), ("f16x3", "1")):
            os.environ["VP3D_TAIL"] = tail
            m = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.25, channels=1024).to(dev).train()
            m.math = math

            def step():
                m.zero_grad(set_to_none=True)
                torch.mean(torch.norm(m(x) - tgt, dim=3)).backward()
            line += "  %s%s %7.3f ms" % (math, "+tail" if tail == "1" else "", min(timed(step, 20), timed(step, 20)))
            del m
        print(line, flush=True)
