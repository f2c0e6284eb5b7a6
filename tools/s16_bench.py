#!/usr/bin/env python3
"""Split-fp16 (S16) GEMM micro-benchmark + accuracy check vs the fp32-MFMA kernel and an fp64 reference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def accuracy(cfgs):
    torch.manual_seed(0)
    for (b, t, c, spec) in [(8, 27, 256, ConvSpec(256, 256, 3, 3, 1)), (5, 27, 160, ConvSpec(160, 96, 3, 1, 3))]:
        x = torch.relu(torch.randn(b, t, spec.c_in, device=dev)) * 1.3
        w = (torch.rand(spec.c_out, spec.c_in, 3, device=dev) * 2 - 1) * 0.03
        bias = torch.randn(spec.c_out, device=dev)
        wt = ops.pack_weight(w)
        st32 = ops.stat_buffers(b * spec.t_out(t), spec.c_out, dev)
        y32 = ops.conv_fwd(x, wt, spec, stats=st32, bias=bias, relu=True)
        xd, wd = x.double(), w.double()
        ref = torch.nn.functional.conv1d(xd.permute(0, 2, 1), wd, bias.double(), dilation=spec.dil, stride=spec.stride).permute(0, 2, 1)
        den = torch.nn.functional.conv1d(xd.abs().permute(0, 2, 1), wd.abs(), dilation=spec.dil, stride=spec.stride).permute(0, 2, 1)
        ref = torch.relu(ref)
        print("shape", tuple(x.shape), spec, " fp32-MFMA max err/sum|ab| %.3e" % float(((y32.double() - ref).abs() / den).max()))
        xs, ws = S.split(x), S.split(wt)
        for cfg in cfgs:
            for splits in (1, 3):
                stats = ops.stat_buffers(b * spec.t_out(t), spec.c_out, dev)
                am = S.new_bound(dev)
                y = S.conv_nt(xs, ws, spec, stats=stats, bias=bias, relu=True, amax_out=am, cfg=cfg, splits=splits)
                print("  s16 cfg %d splits %d max err/sum|ab| %.3e   max|y-y32| %.3e  stats diff %.2e %.2e  amax %.6f (true %.6f)" % (
                    cfg, splits, float(((y.double() - ref).abs() / den).max()),
                    float((y - y32).abs().max()), float((stats[0] - st32[0]).abs().max()), float((stats[1] - st32[1]).abs().max()),
                    float(am.max()), float(y.abs().max())))


def perf(cfgs):
    shapes = [("dil3 B=1024 T=241", 1024, 241, ConvSpec(1024, 1024, 3, 3, 1)),
              ("1x1 B=1024 T=235", 1024, 235, ConvSpec(1024, 1024, 1)),
              ("s3 T_in=81 (cfg3 L0)", 1024, 81, ConvSpec(1024, 1024, 3, 1, 3)),
              ("1x1 T=27 (cfg3 L1)", 1024, 27, ConvSpec(1024, 1024, 1)),
              ("s3 T_in=27 (cfg3 L2)", 1024, 27, ConvSpec(1024, 1024, 3, 1, 3)),
              ("1x1 T=9 (cfg3 L3)", 1024, 9, ConvSpec(1024, 1024, 1)),
              ("s3 T_in=9 (cfg3 L4)", 1024, 9, ConvSpec(1024, 1024, 3, 1, 3)),
              ("1x1 T=3 (cfg3 L5)", 1024, 3, ConvSpec(1024, 1024, 1)),
              ("s3 T_in=3 (cfg3 L6)", 1024, 3, ConvSpec(1024, 1024, 3, 1, 3)),
              ("1x1 T=1 (cfg3 L7)", 1024, 1, ConvSpec(1024, 1024, 1))]
    for tag, b, t, spec in shapes:
        x = torch.relu(torch.randn(b, t, spec.c_in, device=dev))
        w = torch.randn(spec.c_out, spec.c_in, spec.taps, device=dev) * 0.02
        wt = ops.pack_weight(w)
        m, k = b * spec.t_out(t), spec.c_in * spec.taps
        flops = 2.0 * m * spec.c_out * k
        ms = timeit(lambda: ops.conv_fwd(x, wt, spec))
        line = "%-22s fp32 %7.3f ms %6.1f TF |" % (tag, ms, flops / ms / 1e9)
        xs, ws = S.split(x), S.split(wt)
        for cfg in cfgs:
            ms = timeit(lambda: S.conv_nt(xs, ws, spec, cfg=cfg, splits=1))
            line += " c%d %6.3f ms %6.1f |" % (cfg, ms, flops / ms / 1e9)
        pc, ps = S.plan(m, spec.c_out, k)
        st = ops.stat_buffers(m, spec.c_out, dev)
        ms = timeit(lambda: S.conv_nt(xs, ws, spec, stats=st))
        line += " plan(c%d,s%d) %6.3f ms %6.1f" % (pc, ps, ms, flops / ms / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    cfgs = [int(a) for a in sys.argv[1:]] or [0, 4, 20, 22]
    accuracy(cfgs)
    perf(cfgs)
