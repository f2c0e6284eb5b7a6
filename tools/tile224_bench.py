#!/usr/bin/env python3
"""224 x 256 tiles (tile configuration 28) against 256 x 256 (22) on the launches of the benchmark step that the planner gives
to 28 -- stand-alone, HIP events, interleaved:   python tools/tile224_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [("fwd  L0  27648 x 1024 x 3072 + statistics", 1024, 81, ConvSpec(1024, 1024, 3, 1, 3), True),
          ("fwd  L1  27648 x 1024 x 1024 + statistics", 1024, 27, ConvSpec(1024, 1024, 1), True),
          ("dgrad L0 27648 x 3072 x 1024 + amax", 1024, 27, ConvSpec(1024, 3072, 1), False),
          ("fwd  9216 x 1024 x 3072 + statistics", 1024, 27, ConvSpec(1024, 1024, 3, 1, 3), True),
          ("dgrad 9216 x 3072 x 1024 + amax", 1024, 9, ConvSpec(1024, 3072, 1), False),
          ("fwd  9216 x 1024 x 1024 + statistics", 1024, 9, ConvSpec(1024, 1024, 1), True),
          ("dgrad 3072 x 3072 x 1024 + amax", 1024, 3, ConvSpec(1024, 3072, 1), False),
          ("fwd  3072 x 1024 x 3072 + statistics", 1024, 9, ConvSpec(1024, 1024, 3, 1, 3), True)]
for name, b, t, spec, with_stats in shapes:
    x = torch.relu(torch.randn(b, t, spec.c_in, device=dev))
    w = torch.randn(spec.c_out, spec.c_in, spec.taps, device=dev) * 0.02
    xs, ws = S.split(x), S.split(ops.pack_weight(w))
    m = b * spec.t_out(t)
    flops = 2.0 * m * spec.c_out * spec.taps * spec.c_in
    res = {}
    for rep in range(3):
        for cfg in (22, 28, 29, 20):
            slab = S.stat_slab_rows(cfg)
            st = ops.stat_buffers(m, spec.c_out, dev, slab) if with_stats else None
            am = None if with_stats else S.new_bound(dev)
            us = timeit(lambda: S.conv_nt(xs, ws, spec, stats=st, amax_out=am, cfg=cfg, splits=1, stat_slab=slab))
            res.setdefault(cfg, []).append(us)
    a, b_ = min(res[22]), min(res[28])
    print("%-44s 256x256: %7.1f us (%5.1f TF)  224x256: %7.1f (%5.1f)  160x256: %7.1f (%5.1f)  128x128: %7.1f (%5.1f)   planner(mix): %s" % (
        name, a, flops / a / 1e6, b_, flops / b_ / 1e6, min(res[29]), flops / min(res[29]) / 1e6, min(res[20]), flops / min(res[20]) / 1e6,
        S.plan(m, spec.c_out, spec.taps * spec.c_in, mix=True)), flush=True)
