#!/usr/bin/env python3
"""Untraced timing of the regions of the whole cfg3 step on the MAIN stream (HIP events between the host calls, the host
running ahead as in a training loop; rocprofv3 inflates the small kernels and slows the host): forward prologue / expand
layer / each block / head + loss, backward head (shrink) / each block / expand layer."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as bench.py: the package no longer sets it at import (round 6)
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, engine, engine_s16, loss as vloss, ops, ops_s16 as S  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(1024, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


def wrap(mod, fn_name, label_fn):
    orig = getattr(mod, fn_name)

    def f(*a, **k):
        lab = label_fn(*a, **k)
        if lab:
            mark(lab)
        return orig(*a, **k)
    setattr(mod, fn_name, f)


cnt = {"conv": 0, "actb": 0, "exp": 0}
wrap(S, "expand_fwd", lambda *a, **k: ("f expand" if cnt.__setitem__("exp", cnt["exp"] + 1) or cnt["exp"] == 1 else None))
wrap(S, "conv_nt", lambda *a, **k: (cnt.__setitem__("conv", cnt["conv"] + 1) or ("f conv%d" % cnt["conv"])))
wrap(engine, "_shrink", lambda *a, **k: "f head+loss")
wrap(ops, "conv_dgrad", lambda *a, **k: "b shrink")
wrap(S, "bn_act_bwd", lambda *a, **k: (cnt.__setitem__("actb", cnt["actb"] + 1) or ("b layer-%d" % cnt["actb"])))
wrap(S, "expand_p_from_go", lambda *a, **k: "b expand")
runs = []
for it in range(20):
    marks = []
    cnt.update(conv=0, actb=0, exp=0)
    mark("f prologue")
    sync.zero_grad()
    vloss.mpjpe(m(x), tgt).backward()
    sync.sync()
    mark("end")
    runs.append(marks)
torch.cuda.synchronize()
acc, order = {}, []
for mk in runs[8:]:
    for (n0, e0), (_, e1) in zip(mk, mk[1:]):
        if n0 not in acc:
            order.append(n0)
        acc.setdefault(n0, []).append(e0.elapsed_time(e1) * 1e3)
tot = 0.0
for k in order:
    v = sum(acc[k]) / len(acc[k])
    tot += v
    print("%-14s %8.1f us" % (k, v))
print("%-14s %8.1f us" % ("total", tot))
