#!/usr/bin/env python3
"""Persistent tail on / off (VP3D_TAIL), interleaved in one process: training-mode forward alone (torch.no_grad(): nothing
saved) and the whole step (forward + loss + backward) of the cfg3 benchmark configuration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
x = (torch.randn(B, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(B, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)


def fwd():
    with torch.no_grad():
        m(x)


def step():
    sync.zero_grad()
    vloss.mpjpe(m(x), tgt).backward()
    sync.sync()


def timed(fn, n=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, fn in (("forward only", fwd), ("whole step", step)):
    res = {"0": [], "1": []}
    for rep in range(3):
        for v in ("0", "1"):
            os.environ["VP3D_TAIL"] = v
            res[v].append(timed(fn))
    print("%-13s VP3D_TAIL=0: %s   VP3D_TAIL=1: %s   (min %.3f vs %.3f ms)" % (
        name, " ".join("%.3f" % t for t in res["0"]), " ".join("%.3f" % t for t in res["1"]), min(res["0"]), min(res["1"])), flush=True)
