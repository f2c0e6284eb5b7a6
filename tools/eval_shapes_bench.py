#!/usr/bin/env python3
"""The six large GEMMs of the cfg2 eval forward (TemporalModel, arc 3,3,3,3,3, B = 1024, T = 243) stand-alone, each 10 launches
back to back, in forward order and in reverse order: are the late launches of the forward slow because of their SHAPE (rows,
dilation) or because of their POSITION in a 14-ms stretch of sustained MFMA load?   python tools/eval_shapes_bench.py [cfg]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else -1
torch.manual_seed(0)
B, C = 1024, 1024
layers = [("conv0 d=3  T 241->235", 241, ConvSpec(C, C, 3, 3, 1)), ("conv1 1x1 T 235", 235, ConvSpec(C, C, 1)),
          ("conv2 d=9  T 235->217", 235, ConvSpec(C, C, 3, 9, 1)), ("conv3 1x1 T 217", 217, ConvSpec(C, C, 1)),
          ("conv4 d=27 T 217->163", 217, ConvSpec(C, C, 3, 27, 1)), ("conv5 1x1 T 163", 163, ConvSpec(C, C, 1))]
ops_ = []
for name, t, spec in layers:
    x = torch.relu(torch.randn(B, t, C, device=dev))
    w = torch.randn(C, C, spec.taps, device=dev) * 0.02
    xs, ws = S.split(x), S.split(ops.pack_weight(w))
    bias = torch.randn(C, device=dev)
    out = torch.empty(B, spec.t_out(t), C, device=dev)
    ops_.append((name, xs, ws, spec, bias, out, 2.0 * B * spec.t_out(t) * C * spec.taps * C))


def run(order, reps=10):
    res = {}
    for i in order:
        name, xs, ws, spec, bias, out, flops = ops_[i]
        for _ in range(2):
            S.conv_nt(xs, ws, spec, bias=bias, relu=True, out=out, cfg=cfg, splits=1 if cfg > 0 else 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            S.conv_nt(xs, ws, spec, bias=bias, relu=True, out=out, cfg=cfg, splits=1 if cfg > 0 else 0)
        e1.record()
        torch.cuda.synchronize()
        res[i] = flops / (e0.elapsed_time(e1) / reps) / 1e9
    return res


fwd = run(range(6))
rev = run(reversed(range(6)))
one = {}
for i in range(6):                       # one launch each in forward order, as the forward issues them
    pass
print("tile configuration:", cfg if cfg > 0 else "planned")
for i in range(6):
    print("%-24s  forward order %6.1f TFLOP/s   reverse order %6.1f TFLOP/s" % (ops_[i][0], fwd[i], rev[i]))
