#!/usr/bin/env python3
"""BatchNorm-backward column sums inside the dgrad launch (VP3D_FUSE_BN_RED=1) against the separate reduction pass (=0),
interleaved in one process on the cfg3 benchmark configuration: gradient agreement of one step (same dropout masks), then
step times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as bench.py: the package no longer sets it at import (round 6)
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
arc = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3, 3, 3, 3, 3]
rf = 1
for f in arc:
    rf *= f
torch.manual_seed(0)
x = (torch.randn(B, rf, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(B, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, arc, dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)


def step():
    sync.zero_grad()
    vloss.mpjpe(m(x), tgt).backward()
    sync.sync()


def grads(v):
    os.environ["VP3D_FUSE_BN_RED"] = v
    st = {k: t.clone() for k, t in m.state_dict().items()}
    m._drop_calls = 1000                      # the same dropout masks for both runs
    step()
    torch.cuda.synchronize()
    out = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    m.load_state_dict(st)
    return out


try:
    g0, g1 = grads("0"), grads("1")
    worst = 0.0
    for k in g0:
        d = (g0[k] - g1[k]).abs().max().item() / (g0[k].abs().max().item() + 1e-30)
        worst = max(worst, d)
        if d > 1e-5:
            print("  %-40s rel diff %.3e" % (k, d))
    print("gradient agreement fused vs separate: worst max-norm relative difference %.3e (masks %s)" % (
        worst, "identical" if worst < 1e-3 else "DIFFERENT or a bug"), flush=True)
except Exception as e:  # noqa: BLE001
    print("gradient comparison failed:", repr(e), flush=True)
    raise


def timed(n=30):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


from videopose3d_amd import engine_s16  # noqa: E402
MODES = ("0", "auto", "auto@16384", "auto@4096", "1")      # (auto@rows: the automatic rule with another row threshold)
res = {v: [] for v in MODES}
rows_default = engine_s16.FUSE_BN_RED_MIN_ROWS
import random  # noqa: E402
random.seed(0)
for rep in range(8):
    order = list(MODES)
    random.shuffle(order)                            # (fixed-order interleaving has position effects of up to 1 %: DESIGN 4.9)
    for v in order:
        os.environ["VP3D_FUSE_BN_RED"] = v.split("@")[0]
        engine_s16.FUSE_BN_RED_MIN_ROWS = int(v.split("@")[1]) if "@" in v else rows_default
        res[v].append(timed())
for v in MODES:
    print("whole step  VP3D_FUSE_BN_RED=%-10s: %s   (min %.3f, median %.3f ms)" % (
        v, " ".join("%.3f" % t for t in res[v]), min(res[v]), sorted(res[v])[len(res[v]) // 2]), flush=True)
