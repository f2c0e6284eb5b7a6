#!/usr/bin/env python3
"""cfg2 eval forward time of this process (TemporalModel, arc 3,3,3,3,3, C = 1024, B = 1024, T = 243): 4 windows of 5 calls --
for A/Bs of whole libraries in alternating processes."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videopose3d_amd as V
dev = "cuda:0"
torch.manual_seed(0)
x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
m = V.TemporalModel(17, 2, 17, [3, 3, 3, 3, 3], channels=1024).to(dev).eval()
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    res = []
    for rep in range(4):
        t0 = time.perf_counter()
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 5 * 1e3)
print("eval fwd ms:", " ".join("%.3f" % t for t in res))
