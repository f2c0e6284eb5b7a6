#!/usr/bin/env python3
"""Where does the time of the dgrad-form S16 GEMM (M x 3072 x 1024, residual + amax epilogue) go?  Times the launch with
the epilogue features toggled, in the planner's configuration (-1) and in the 224x256 / 256x256 / 128x128 ones (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd._lib import RowMap  # noqa: E402

dev = "cuda:0"


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


def run(bb, t_o, n_taps, c=1024):
    m = bb * t_o
    dy = S.split(torch.randn(bb, t_o, c, device=dev))
    wd = S.split(torch.randn(n_taps * c, c, device=dev) * 0.02)
    dx = torch.empty((bb, n_taps * t_o, c), dtype=torch.float32, device=dev)
    r = torch.randn(bb, t_o, c, device=dev)
    rm = RowMap(bb, t_o, t_o, 1, 0, 0, 1)
    am = S.new_bound(dev)
    flops = 2.0 * m * n_taps * c * c
    for cfg in (-1, 28, 22, 20):
        line = "M=%6d N=%5d K=%d cfg %3d:" % (m, n_taps * c, c, cfg)
        for tag, use_r, use_am in (("plain", 0, 0), ("amax", 0, 1), ("res", 1, 0), ("res+amax", 1, 1)):
            e = ops._epi(residual=(r, 1, 0, (n_taps // 2) * c), n_cols=n_taps * c) if use_r else None
            us = timeit(lambda: S.gemm_rows(dy, wd, rm, c, c, n_taps * c, dx, n_taps * t_o * c, n_taps * c, epi=e,
                                            amax_out=am if use_am else None, cfg=cfg, family="tconv_dgrad", mix=True))
            line += "  %s %7.1f us %6.1f TF" % (tag, us, flops / us / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    run(1024, 27, 3)
    run(1024, 27, 1)
    run(1024, 9, 3)
    run(1024, 9, 1)
