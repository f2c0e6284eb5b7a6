#!/usr/bin/env python3
"""Eager cfg3 training step time of this process (for A/B of knobs that are read once per process): N windows of 30 steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402

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


for _ in range(40):
    step()
res = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 30 * 1e3)
print("step: %s   (median %.3f ms)  %s" % (" ".join("%.3f" % t for t in res), sorted(res)[len(res) // 2],
                                          " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("VP3D_"))))
