#!/usr/bin/env python3
"""Step latency, eager vs hipGraph replay (graph.GraphedTrainStep), small and large configurations."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp  # noqa: E402
from videopose3d_amd.graph import GraphedTrainStep  # noqa: E402

dev = "cuda:0"


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for fw, b in (([3, 3, 3], 128), ([3, 3, 3], 1024), ([3, 3, 3, 3, 3], 128), ([3, 3, 3, 3, 3], 1024)):
    rf = 3 ** len(fw)
    x = (torch.randn(b, rf, 17, 2, device=dev) * 0.5).clamp(-1, 1)
    tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3
    m = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.25, channels=1024).to(dev).train()
    sync = dp.FlatGradSync(m.parameters(), direct_module=m)

    def eager():
        sync.zero_grad()
        torch.mean(torch.norm(m(x) - tgt, dim=3)).backward()
    ms_e = timed(eager, 20)
    step = GraphedTrainStep(m, sync)
    ms_g = timed(lambda: step(x, tgt), 20)
    print("arc %-10s B=%5d  eager %7.3f ms   graph replay %7.3f ms   (%.2fx)" % (",".join(map(str, fw)), b, ms_e, ms_g, ms_e / ms_g),
          flush=True)
    del m, sync, step
    torch.cuda.empty_cache()
