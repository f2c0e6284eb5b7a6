#!/usr/bin/env python3
"""The expand layer's BatchNorm statistics: statistics-only pass over the conv output + vp3d_bn_finalize against the centred
second-moment matrix of the input (vp3d_expand_stats_gram_s16), stand-alone at the benchmark size, + agreement of the two."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import engine_s16, ops, ops_s16 as S  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
b, t, c_in, c = 1024, 243, 34, 1024
spec = ConvSpec(c_in, c, 3, 1, 3)
kpad, kv = engine_s16.expand_kpad(spec), 3 * c_in
x = (torch.randn(b, t, c_in, device=dev) * 0.5).clamp(-1, 1)
w = (torch.rand(c, c_in, 3, device=dev) * 2 - 1) * 0.1
xb = S.amax(x, floor=1.0)
x_rows, x_t = S.im2row_split(x, spec, kpad, kv, xb, want_t=True)
w_packed = ops.pack_weight(w, ld_out=kpad)
ws_ = S.split(w_packed)
m = b * spec.t_out(t)
bn = torch.nn.BatchNorm1d(c).to(dev)


def timeit(fn, iters=20):
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


st = ops.stat_buffers(m, c, dev)


def old():
    S.expand_fwd(x_rows, ws_, stats=st)
    return ops.bn_finalize(bn, m, st)


def new():
    return S.expand_stats_gram(x_t, w_packed, bn, m, kv, kv)


a, g = old(), new()
print("agreement: scale %.2e  shift %.2e (relative to the largest)" % (float((a[0] / g[0] - 1).abs().max()),
                                                                      float((a[1] - g[1]).abs().max() / a[1].abs().max())))
print("statistics pass + bn_finalize: %.1f us     centred Gram + sum + quadratic forms: %.1f us" % (timeit(old), timeit(new)))
recs = []
ops.set_profiler(recs)
new()
torch.cuda.synchronize()
ops.set_profiler(None)
import ctypes as C  # noqa: E402
from videopose3d_amd import _lib  # noqa: E402
L = _lib.lib()
groups = int(L.vp3d_expand_stats_gram_groups(m))
print("row groups:", groups)
