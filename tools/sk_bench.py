#!/usr/bin/env python3
"""Stream-K configurations (120 / 122) vs the planner's pick and the plain tilings (20 / 22) on the GEMM shapes of the cfg3
training step: forward form with the BatchNorm-statistics epilogue and the dgrad form (HIP events, us)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd._lib import RowMap  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"
CFGS = tuple(int(a) for a in sys.argv[1:]) or (20, 120, 22, 122)


def timeit(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def fwd(b, t, taps, c=1024):
    spec = ConvSpec(c, c, taps, 1, taps) if taps > 1 else ConvSpec(c, c, 1)
    x = S.split(torch.randn(b, t, c, device=dev))
    w = S.split(ops.pack_weight(torch.randn(c, c, taps, device=dev) * 0.02))
    m, k = b * spec.t_out(t), taps * c
    st = ops.stat_buffers(m, c, dev)
    pc, ps = S.plan(m, c, k)
    line = "fwd   M=%6d N=%5d K=%5d | plan(c%d,s%d) %7.1f |" % (m, c, k, pc, ps, timeit(lambda: S.conv_nt(x, w, spec, stats=st)))
    for cfg in CFGS:
        us = timeit(lambda: S.conv_nt(x, w, spec, stats=st, cfg=cfg, splits=1))
        line += " c%d %7.1f (%5.1f TF) |" % (cfg, us, 2.0 * m * c * k / us / 1e6)
    print(line, flush=True)


def dgrad(bb, t_o, n_taps, c=1024):
    m = bb * t_o
    dy = S.split(torch.randn(bb, t_o, c, device=dev))
    wd = S.split(torch.randn(n_taps * c, c, device=dev) * 0.02)
    dx = torch.empty((bb, n_taps * t_o, c), dtype=torch.float32, device=dev)
    r = torch.randn(bb, t_o, c, device=dev)
    rm = RowMap(bb, t_o, t_o, 1, 0, 0, 1)
    am = S.new_bound(dev)
    e = ops._epi(residual=(r, 1, 0, (n_taps // 2) * c), n_cols=n_taps * c) if n_taps > 1 else None
    pc, ps = S.plan(m, n_taps * c, c)
    line = "dgrad M=%6d N=%5d K=%5d | plan(c%d,s%d) %7.1f |" % (m, n_taps * c, c, pc, ps, timeit(
        lambda: S.gemm_rows(dy, wd, rm, c, c, n_taps * c, dx, n_taps * t_o * c, n_taps * c, epi=e, amax_out=am, family="tconv_dgrad")))
    for cfg in CFGS:
        us = timeit(lambda: S.gemm_rows(dy, wd, rm, c, c, n_taps * c, dx, n_taps * t_o * c, n_taps * c, epi=e, amax_out=am,
                                        cfg=cfg, splits=1, family="tconv_dgrad"))
        line += " c%d %7.1f (%5.1f TF) |" % (cfg, us, 2.0 * m * n_taps * c * c / us / 1e6)
    print(line, flush=True)


if __name__ == "__main__":
    for t_in in (81, 27, 9, 3):
        fwd(1024, t_in, 3)
        fwd(1024, t_in // 3, 1)
    for t_o in (27, 9, 3, 1):
        dgrad(1024, t_o, 3)
        dgrad(1024, t_o, 1)
