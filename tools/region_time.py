#!/usr/bin/env python3
"""Untraced timing of regions of the cfg3 forward (HIP events between ops_s16 calls, no profiler: rocprofv3 inflates the small
kernels): prologue (input staging, weight maxima + packs, activation bounds) = start of forward -> first expand-layer launch;
expand layer = first expand launch -> first C x C GEMM; the rest of the stack; head (shrink conv)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as bench.py: the package no longer sets it at import (round 6)
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import engine, engine_s16, ops_s16 as S  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
marks = []


def mark(name):
    global marks
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


orig_expand, orig_conv, orig_shrink = S.expand_fwd, S.conv_nt, engine._shrink
state = {"expand": 0, "conv": 0}


def expand_fwd(*a, **k):
    if state["expand"] == 0:
        mark("expand")
    state["expand"] += 1
    return orig_expand(*a, **k)


def conv_nt(*a, **k):
    if state["conv"] == 0:
        mark("stack")
    state["conv"] += 1
    return orig_conv(*a, **k)


def shrink(*a, **k):
    mark("head")
    return orig_shrink(*a, **k)


S.expand_fwd, S.conv_nt, engine._shrink = expand_fwd, conv_nt, shrink
acc = {}
runs = []
for it in range(24):                      # no synchronisation between the steps: the host runs ahead as in a training loop
    marks = []
    state.update(expand=0, conv=0)
    with torch.no_grad():
        mark("prologue")
        m(x)
        mark("end")
    runs.append(marks)
torch.cuda.synchronize()
for marks in runs[8:]:
    for (n0, e0), (_, e1) in zip(marks, marks[1:]):
        acc.setdefault(n0, []).append(e0.elapsed_time(e1) * 1e3)
for k, v in acc.items():
    print("%-9s %7.1f us (min %.1f)" % (k, sum(v) / len(v), min(v)))
print("total     %7.1f us" % sum(sum(v) / len(v) for v in acc.values()))
