#!/usr/bin/env python3
"""Phase timing inside the persistent tail kernels (csrc/vp3d_tail_s16.hip): workgroup 0 stamps the 100 MHz wall clock at
kernel start, after every grid barrier and at the end; this prints the phase durations of the cfg3 step's tail
(forward: GEMM | statistics | activation per layer; backward: reduce | apply | GEMMs per layer, then the final pass)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss, ops_s16 as S  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
x = (torch.randn(B, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(B, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)
S.TAIL_TRACE["fwd"] = torch.zeros(128, dtype=torch.int64, device=dev)
S.TAIL_TRACE["bwd"] = torch.zeros(128, dtype=torch.int64, device=dev)
for _ in range(6):
    sync.zero_grad()
    vloss.mpjpe(m(x), tgt).backward()
    sync.sync()
torch.cuda.synchronize()
print("tail error flag:", S.tail_error(dev), " grouped barrier:", __import__("videopose3d_amd")._lib.lib().vp3d_tail_barrier_grouped())
names = {"fwd": ["gemm", "stats", "act"], "bwd": ["reduce(+unpack)", "apply", "gemms"]}
for k in ("fwd", "bwd"):
    t = S.TAIL_TRACE[k].cpu().tolist()
    n = t[0]
    st = t[1:1 + n]
    d = [(b - a) / 100.0 for a, b in zip(st, st[1:])]
    print("%s: %d stamps, total %.1f us" % (k, n, (st[-1] - st[0]) / 100.0))
    for i, v in enumerate(d):
        lay = i // 3
        nm = names[k][i % 3] if i < len(d) - (1 if k == "bwd" else 0) else "final"
        print("   phase %2d  layer %d  %-16s %7.1f us" % (i, lay, nm, v))
