#!/usr/bin/env python3
"""Eager cfg3 step time of ONE process (for A/Bs of whole libraries in alternating processes:
    python tools/ab_lib.py videopose3d_amd/libvp3d_old.so tools/step_time.py [reps] [steps])."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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


for _ in range(20):
    step()
res = []
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / steps * 1e3)
print("%s: %s  -> min %.3f median %.3f ms / step" % (os.path.basename(V._lib.LIB_PATH), " ".join("%.3f" % t for t in res), min(res),
                                                    sorted(res)[len(res) // 2]), flush=True)
