#!/usr/bin/env python3
"""Same-process A/B of the training-mode FORWARD alone (and of the whole step) of the benchmark configuration under an
environment knob read at call time or an internal switch:   python tools/fwd_ab.py SW:tile_mix 0 1 [reps]   (VP3D_TILE_224 in rounds 3-4)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as bench.py: the package no longer sets it at import (round 6)
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402
from tools.env_ab_lib import set_knob  # noqa: E402

var, values = sys.argv[1], sys.argv[2:4]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = "cuda:0"
torch.manual_seed(0)
x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(1024, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)


def fwd():
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
    res = {v: [] for v in values}
    import random
    random.seed(0)
    for rep in range(reps):
        order = list(values)
        random.shuffle(order)                        # (fixed-order interleaving has position effects: DESIGN.md 4.9)
        for v in order:
            set_knob(var, v)
            res[v].append(timed(fn))
    for v in values:
        print("%-12s %s=%s: %s  -> min %.3f median %.3f ms" % (name, var, v, " ".join("%.3f" % t for t in res[v]), min(res[v]),
                                                              sorted(res[v])[len(res[v]) // 2]), flush=True)
