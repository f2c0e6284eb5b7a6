#!/usr/bin/env python3
"""Per-shape timing of the three GEMM entry points on the layer shapes of cfg2 / cfg3 (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops  # noqa: E402
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


def run(tag, b, t, spec, do_bwd=True):
    x = torch.randn(b, t, spec.c_in, device=dev)
    w = torch.randn(spec.c_out, spec.c_in, spec.taps, device=dev) * 0.02
    kpad = ops.padded_k(spec)
    if kpad:                                  # expand conv: im2row staging + 1-tap GEMM over padded rows
        flops = 2.0 * b * spec.t_out(t) * spec.c_out * spec.c_in * spec.taps
        wt = ops.pack_weight(w, ld_out=kpad)
        ms0 = timeit(lambda: ops.im2row(x, spec, kpad))
        xp = ops.im2row(x, spec, kpad)
        s1 = ConvSpec(kpad, spec.c_out, 1)
        ms = timeit(lambda: ops.conv_fwd(xp, wt, s1))
        line = "%-34s M=%7d N=%5d K=%5d  im2row %6.3f ms fwd %8.3f ms %6.1f TF" % (tag, b * spec.t_out(t), spec.c_out, kpad, ms0, ms, flops / ms / 1e9)
        if do_bwd:
            g = torch.randn(b, spec.t_out(t), spec.c_out, device=dev)
            ms = timeit(lambda: ops.conv_wgrad(g, xp, spec, rows_kpad=kpad))
            line += " | wgrad %8.3f ms %6.1f TF" % (ms, flops / ms / 1e9)
        print(line, flush=True)
        return
    wt = ops.pack_weight(w)
    t_out = spec.t_out(t)
    g = torch.randn(b, t_out, spec.c_out, device=dev)
    flops = 2.0 * b * t_out * spec.c_out * spec.c_in * spec.taps
    ms = timeit(lambda: ops.conv_fwd(x, wt, spec))
    line = "%-34s M=%7d N=%5d K=%5d  fwd %8.3f ms %6.1f TF" % (tag, b * t_out, spec.c_out, spec.c_in * spec.taps, ms, flops / ms / 1e9)
    if do_bwd:
        ms = timeit(lambda: ops.conv_dgrad(g, wt, spec, t))
        line += " | dgrad %8.3f ms %6.1f TF" % (ms, flops / ms / 1e9)
        ms = timeit(lambda: ops.conv_wgrad(g, x, spec))
        line += " | wgrad %8.3f ms %6.1f TF" % (ms, flops / ms / 1e9)
    print(line, flush=True)


if __name__ == "__main__":
    C = 1024
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    print("== cfg2 (dilated eval, B=%d, T=243) forward shapes" % B)
    run("expand k3 34->1024", B, 243, ConvSpec(34, C, 3, 1, 1), do_bwd=False)
    t = 241
    for d in (3, 9, 27, 81):
        s = ConvSpec(C, C, 3, d, 1)
        run("conv3 dil=%d" % d, B, t, s, do_bwd=False)
        t = s.t_out(t)
        run("conv1x1 T=%d" % t, B, t, ConvSpec(C, C, 1), do_bwd=False)
    print("== cfg3 (strided train, B=%d, T=243) fwd/dgrad/wgrad shapes" % B)
    run("expand k3 s3 34->1024", B, 243, ConvSpec(34, C, 3, 1, 3))
    t = 81
    for _ in range(4):
        s = ConvSpec(C, C, 3, 1, 3)
        run("conv3 s3 T_in=%d" % t, B, t, s)
        t = s.t_out(t)
        run("conv1x1 T=%d" % t, B, t, ConvSpec(C, C, 1))
    run("shrink 1024->51", B, 1, ConvSpec(C, 51, 1))
