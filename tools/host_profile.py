#!/usr/bin/env python3
"""Where does the HOST time of a launch-bound step go?  cProfile of the arc 3,3,3, B = 128 training step on the exact-fp32
engine (what run.py's semi-supervised models run on) with the GPU running behind: top functions by own time."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402

dev = "cuda:0"
math = sys.argv[1] if len(sys.argv) > 1 else "f32"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.manual_seed(0)
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
m.math = math
x = (torch.randn(b, 27, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3


def step():
    m.zero_grad(set_to_none=True)
    torch.mean(torch.norm(m(x) - tgt, dim=3)).backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
t_host = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 50 * 1e3
print("host enqueue %.3f ms / step, with the GPU %.3f ms / step" % (t_host, t_all))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
