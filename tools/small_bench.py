#!/usr/bin/env python3
"""Small-M layer shapes only (for rocprof of the split-K path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videopose3d_amd import ops
from videopose3d_amd.plan import ConvSpec
from tools.gemm_bench import run
C = 1024
for b, t in ((1024, 3), (1024, 9)):
    run("conv3 s3 T_in=%d" % t, b, t, ConvSpec(C, C, 3, 1, 3))
    run("conv1x1 T=%d" % (t // 3), b, t // 3, ConvSpec(C, C, 1))
h = torch.randn(1024, 1, C, device="cuda:0"); w = torch.randn(51, C, device="cuda:0"); bias = torch.randn(51, device="cuda:0")
for _ in range(5):
    ops.skinny_fwd(h, w, bias)
torch.cuda.synchronize()
