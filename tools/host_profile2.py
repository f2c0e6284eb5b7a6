#!/usr/bin/env python3
"""cProfile of engine.forward_train / backward_train called directly (no autograd thread): own + cumulative host time of
the package's functions per step, launch-bound configuration (arc 3,3,3, B = 128)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import engine  # noqa: E402

dev = "cuda:0"
math = sys.argv[1] if len(sys.argv) > 1 else "f32"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 128
arc = [3, 3, 3] if len(sys.argv) <= 3 else [3] * int(sys.argv[3])
if math == "f16x3":
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
torch.manual_seed(0)
m = V.TemporalModelOptimized1f(17, 2, 17, arc, dropout=0.25, channels=1024).to(dev).train()
m.math = math
rf = m.receptive_field()
x3 = (torch.randn(b, rf, 34, device=dev) * 0.5).clamp(-1, 1)
g = torch.randn(b, 1, 51, device=dev) * 0.01


def step():
    out, saved = engine.forward_train(m, x3, save=True)
    engine.backward_train(m, saved, g, False)


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
