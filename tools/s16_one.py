#!/usr/bin/env python3
"""Run one S16 GEMM shape in one tile configuration a few times (target of rocprofv3 --pmc)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"
cfg = int(sys.argv[1])
b, t = int(sys.argv[2]) if len(sys.argv) > 2 else 1024, int(sys.argv[3]) if len(sys.argv) > 3 else 81
spec = ConvSpec(1024, 1024, 3, 1, 3)
x = torch.relu(torch.randn(b, t, 1024, device=dev))
w = torch.randn(1024, 1024, 3, device=dev) * 0.02
xs, ws = S.split(x), S.split(ops.pack_weight(w))
for _ in range(3):
    S.conv_nt(xs, ws, spec, cfg=cfg, splits=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    S.conv_nt(xs, ws, spec, cfg=cfg, splits=1)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("cfg %d: %.3f ms  %.1f TF-equivalent" % (cfg, ms, 2.0 * b * spec.t_out(t) * 1024 * 3072 / ms / 1e9))
