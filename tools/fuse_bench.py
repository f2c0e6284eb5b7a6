#!/usr/bin/env python3
"""dgrad with / without the fused activation-backward epilogue on the cfg3 shapes (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videopose3d_amd import ops
from videopose3d_amd.plan import ConvSpec
dev = "cuda:0"
C = 1024


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for tag, spec, b, t_in in (("3-tap s3 T_in=81", ConvSpec(C, C, 3, 1, 3), 1024, 81), ("1x1 T=27", ConvSpec(C, C, 1), 1024, 27),
                           ("3-tap s3 T_in=27", ConvSpec(C, C, 3, 1, 3), 1024, 27), ("1x1 T=9", ConvSpec(C, C, 1), 1024, 9)):
    t_out = spec.t_out(t_in)
    dy = torch.randn(b, t_out, C, device=dev)
    wt = ops.pack_weight(torch.randn(C, C, spec.taps, device=dev) * 0.02)
    y_up = torch.randn(b, t_in, C, device=dev)
    coef = torch.rand(4, C, device=dev) + 0.5
    base = timeit(lambda: ops.conv_dgrad(dy, wt, spec, t_in))
    line = "%-18s plain %.3f ms" % (tag, base)
    for p in (0.0, 0.25):
        drop = ops.make_dropout(p, 1, 2, 3)
        for sv in (True, False):
            ms = timeit(lambda: ops.conv_dgrad(dy, wt, spec, t_in, act_bwd=(y_up, coef, drop), store_v=sv))
            line += " | p=%.2f v=%d %.3f" % (p, sv, ms)
    red = timeit(lambda: ops.bn_act_bwd(y_up, y_up, coef, ops.make_dropout(0.25, 1, 2, 3)))
    print(line + " | separate reduce+finalize+apply %.3f" % red, flush=True)
