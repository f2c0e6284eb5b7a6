#!/usr/bin/env python3
"""Run N training steps (and optionally eval forwards) in one math mode: the target command of rocprofv3 --kernel-trace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as bench.py: the package no longer sets it at import (round 6)
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402

dev = "cuda:0"
math = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
what = sys.argv[2] if len(sys.argv) > 2 else "train"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
fw, c, b = [3, 3, 3, 3, 3], 1024, 1024
torch.manual_seed(0)
x = (torch.randn(b, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3
if what == "train":
    m = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.25, channels=c).to(dev).train()
    m.math = math
    from videopose3d_amd import dp, loss as vloss
    sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)       # the bench step
    for _ in range(n):
        sync.zero_grad()
        vloss.mpjpe(m(x), tgt).backward()
        sync.sync()
else:
    e = V.TemporalModel(17, 2, 17, fw, channels=c).to(dev).eval()
    e.math = math
    with torch.no_grad():
        for _ in range(n):
            e(x)
torch.cuda.synchronize()
