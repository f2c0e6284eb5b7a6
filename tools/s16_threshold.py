#!/usr/bin/env python3
"""Where does the split-fp16 engine start to pay?  Train step / eval forward of small configurations on BOTH engines, each forced
(engine.S16_MIN_FORWARD_FLOPS = 0 for the f16x3 rows), with the forward GFLOP of the call -- the quantity the threshold
engine.S16_MIN_FORWARD_FLOPS is expressed in.  Round 6 re-derivation (the round-1 threshold of 40 / 35 GFLOP sent run.py's own
default configuration, arc 3,3,3 at B = 1024 = 36.4 GFLOP, to the fp32 engine: profiles/r06_s16_threshold.txt)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import engine  # noqa: E402

dev = "cuda:0"
engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[1]


print("# training step (run.py's own loop shape: torch mpjpe + autograd), C = 1024, dropout 0.25", flush=True)
for fw in ([3, 3, 3], [3, 3, 3, 3], [3, 3, 3, 3, 3]):
    rf = 3 ** len(fw)
    for b in (32, 64, 128, 192, 256, 384, 512, 768, 1024):
        x = (torch.randn(b, rf, 17, 2, device=dev) * 0.5).clamp(-1, 1)
        tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3
        res = {}
        gf = 0.0
        for math in ("f32", "f16x3"):
            m = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.25, channels=1024).to(dev).train()
            m.math = math
            gf = m._plan.forward_flops(b, rf) / 1e9

            def step():
                m.zero_grad(set_to_none=True)
                torch.mean(torch.norm(m(x) - tgt, dim=3)).backward()
            res[math] = timed(step)
            del m
        print("train arc %-10s B=%5d  fwd %7.1f GFLOP   f32 %7.3f ms   f16x3 %7.3f ms   f16x3/f32 %.2f" %
              (",".join(map(str, fw)), b, gf, res["f32"], res["f16x3"], res["f16x3"] / res["f32"]), flush=True)
print("# eval forward (TemporalModel, BN folded): run.py evaluates one sequence + its mirrored copy per call (B = 2)", flush=True)
for fw, ts_ in (([3, 3, 3, 3, 3], (100, 243, 400, 600, 1000, 2000, 4000)), ([3, 3, 3], (243, 1000, 2000, 4000, 8000))):
    rf = 3 ** len(fw)
    for t in ts_:
        x = (torch.randn(2, t + rf - 1, 17, 2, device=dev) * 0.5).clamp(-1, 1)
        res = {}
        for math in ("f32", "f16x3"):
            e = V.TemporalModel(17, 2, 17, fw, channels=1024).to(dev).eval()
            e.math = math
            gf = e._plan.forward_flops(2, t + rf - 1) / 1e9
            with torch.no_grad():
                res[math] = timed(lambda: e(x))
            del e
        print("eval  arc %-10s B=2 T_out=%5d  fwd %7.1f GFLOP   f32 %7.3f ms   f16x3 %7.3f ms   f16x3/f32 %.2f" %
              (",".join(map(str, fw)), t, gf, res["f32"], res["f16x3"], res["f16x3"] / res["f32"]), flush=True)
