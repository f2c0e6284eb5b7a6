#!/usr/bin/env python3
"""Same-process A/B of the benchmark training step under an environment knob that the package reads at call time, or under one
of its internal switches (videopose3d_amd/_switches.py: `SW:tile_mix`, `SW:wgrad_rows`, ...):
    python tools/env_ab.py VP3D_SOME_KNOB 0 1 [reps] [steps]      (interleaved runs: box-to-box spread does not enter);
more than two values: python tools/env_ab.py VP3D_SOME_KNOB 0,1,2,3 - [reps] [steps]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as bench.py: the package no longer sets it at import (round 6)
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402
from tools.env_ab_lib import set_knob  # noqa: E402

var, values = sys.argv[1], (sys.argv[2].split(",") if sys.argv[3] == "-" else sys.argv[2:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 25
dev = "cuda:0"
torch.manual_seed(0)
x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(1024, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)



def step():
    sync.zero_grad()
    vloss.mpjpe(m(x), tgt).backward()
    sync.sync()


def timed(n):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


import random  # noqa: E402
random.seed(int(os.environ.get("ENV_AB_SEED", "0")))
res = {v: [] for v in values}
for rep in range(reps):
    order = list(values)
    random.shuffle(order)                            # (no value always runs behind the same neighbour)
    for v in order:
        set_knob(var, v)
        res[v].append(timed(steps))
for v in values:
    print("%s=%s: %s  -> min %.3f median %.3f ms / step" % (var, v, " ".join("%.3f" % t for t in res[v]), min(res[v]),
                                                           sorted(res[v])[len(res[v]) // 2]), flush=True)
