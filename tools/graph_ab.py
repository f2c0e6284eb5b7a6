#!/usr/bin/env python3
"""cfg3 step: eager vs hipGraph replay, with and without the second-stream overlap (VP3D_OVERLAP), interleaved."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402
from videopose3d_amd.graph import GraphedTrainStep  # noqa: E402

dev = "cuda:0"
x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(1024, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)


def eager():
    sync.zero_grad()
    vloss.mpjpe(m(x), tgt).backward()
    sync.sync()


def timed(fn, n=25):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for ov in ("1", "0"):
    os.environ["VP3D_OVERLAP"] = ov
    step = GraphedTrainStep(m, sync)
    for rep in range(2):
        print("VP3D_OVERLAP=%s: eager %.3f ms   graph replay %.3f ms" % (ov, timed(eager), timed(lambda: step(x, tgt))), flush=True)
    del step
