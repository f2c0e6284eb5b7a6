#!/usr/bin/env python3
"""A few launches of the largest cfg2 forward GEMMs (for PMC passes: clock, MFMA busy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videopose3d_amd import ops
from videopose3d_amd.plan import ConvSpec
C = 1024
x = torch.randn(1024, 241, C, device="cuda:0")
for spec in (ConvSpec(C, C, 3, 3, 1), ConvSpec(C, C, 1)):
    w = torch.randn(C, C, spec.taps, device="cuda:0") * 0.02
    wt = ops.pack_weight(w)
    for _ in range(4):
        ops.conv_fwd(x, wt, spec)
torch.cuda.synchronize()
