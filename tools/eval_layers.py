#!/usr/bin/env python3
"""Per-launch rates of the cfg2 eval forward in situ (HIP events around every GEMM call), with the activation scale of a
default-initialised model (running statistics 0 / 1: the activations shrink layer by layer) and with per-layer BatchNorm scales
chosen so that every layer's activations keep unit scale: is the late layers' low rate a matter of the DATA?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import ops  # noqa: E402

dev = "cuda:0"
N_WARM = 2
torch.manual_seed(0)
x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)


def run(tag, prep):
    torch.manual_seed(0)
    m = V.TemporalModel(17, 2, 17, [3, 3, 3, 3, 3], channels=1024).to(dev).eval()
    prep(m)
    with torch.no_grad():
        for _ in range(N_WARM):
            m(x)
        recs = []
        ops.set_profiler(recs)
        y = m(x)
        torch.cuda.synchronize()
        ops.set_profiler(None)
    big = [(f, e0.elapsed_time(e1)) for _, f, e0, e1, _b, _s in recs if f > 1e11]
    print("%-34s %s   |y| max %.3g" % (tag, " ".join("%6.1f" % (f / ms / 1e9) for f, ms in big), float(y.abs().max())), flush=True)


def unit_scale(m):
    # running_var such that each BatchNorm roughly renormalises its input (measured with a forward in train-mode statistics)
    tr = V.TemporalModel(17, 2, 17, [3, 3, 3, 3, 3], channels=1024).to(dev)
    tr.load_state_dict(m.state_dict())
    tr.train()
    tr.drop.p = 0.0
    for bn in [tr.expand_bn] + list(tr.layers_bn):
        bn.momentum = 1.0
    with torch.no_grad():
        tr(x[:64])
    m.load_state_dict(tr.state_dict())


print("TFLOP/s of conv0 .. conv5 (K = 3072, 1024, 3072, 1024, 3072, 1024):")
for N_WARM in (2, 8, 20):
    print("%d forwards before the instrumented one" % N_WARM)
    run("default init (running stats 0 / 1)", lambda m: None)
    run("running stats of the data (unit scale)", unit_scale)
