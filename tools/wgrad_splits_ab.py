#!/usr/bin/env python3
"""The cfg3 step with the round-2 slice-count candidates of the rows-form weight gradient against the current ones
(ops_s16.WGRAD_SPLIT_CANDIDATES), interleaved in one process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss, ops_s16 as S  # noqa: E402

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


def timed(n=30):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


new = S.WGRAD_SPLIT_CANDIDATES
old = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)
res = {"old": [], "new": []}
for rep in range(6):
    for k, cand in (("old", old), ("new", new)):
        S.WGRAD_SPLIT_CANDIDATES = cand
        res[k].append(timed())
for k in ("old", "new"):
    print("whole step, slice candidates %s: %s   (min %.3f, median %.3f ms)" % (
        k, " ".join("%.3f" % t for t in res[k]), min(res[k]), sorted(res[k])[3]), flush=True)
