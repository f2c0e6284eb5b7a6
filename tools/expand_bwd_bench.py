#!/usr/bin/env python3
"""Stand-alone times of the pieces of the expand layer's backward (M = 82,944 x C = 1024, kpad = 128): the mask producer
(rows / transposed), P = G^T X from the rows (k_tn_s16<1>) and from transposed copies (k_nt_s16), X^T X, the post kernel --
and of the three passes they replace (reduce + finalize, apply)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videopose3d_amd._switches import SW  # noqa: E402
SW["expand_rows"] = True              # (so that the rows-form P GEMM can be timed next to the default)
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402

dev = "cuda:0"
m, c, kpad = 82944, 1024, 128


def timeit(fn, iters=10, warm=3):
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


torch.manual_seed(0)
go = torch.randn(1024, 81, c, device=dev) * 1e-4
y = torch.randn(1024, 81, c, device=dev)
x = torch.randn(m, kpad, device=dev).clamp(-1, 1)
x[:, 102:] = 0
x[:, 102] = 1
bits = torch.randint(0, 256, (m * c // 8,), dtype=torch.uint8, device=dev)
gb = S.amax(go)
xb = S.amax(x)
x_rows, x_t = S.split_t(x, xb)
coef = torch.stack([torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c)]).to(dev).contiguous()
wp = torch.randn(c, kpad, device=dev) * 0.1
print("mask -> rows        %7.1f us" % timeit(lambda: S.act_mask(go, gb, bits, 0.25, transposed=False)))
print("mask -> transposed  %7.1f us" % timeit(lambda: S.act_mask(go, gb, bits, 0.25, transposed=True)))
g_rows = S.act_mask(go, gb, bits, 0.25, transposed=False)
g_t = S.act_mask(go, gb, bits, 0.25, transposed=True)
gram = S.gram(x_t)
print("X^T X (NT + sum)    %7.1f us" % timeit(lambda: S.gram(x_t)))
print("P + post, rows      %7.1f us" % timeit(lambda: S.expand_bwd(g_rows, S.S16(x_rows.data, xb), gram, wp, coef, m, 34, 3, 102, True)))
print("P + post, NT        %7.1f us" % timeit(lambda: S.expand_bwd(g_t, x_t, gram, wp, coef, m, 34, 3, 102, False)))
for sp in (16, 32, 64):
    ws = torch.empty((sp, c, kpad), device=dev)
    from videopose3d_amd import _lib
    print("  k_tn_s16<1> alone, %2d slices %7.1f us" % (sp, timeit(lambda: _lib.check(_lib.lib().vp3d_wgrad_rows_s16(
        ops._stream(), m, g_rows.data.data_ptr(), c, c, g_rows.bound_ptr(), x_rows.data.data_ptr(), kpad, 1, kpad, xb.data_ptr(), sp,
        ws.data_ptr()), "wgrad_rows"))))
print("  NT GEMM alone       %7.1f us" % timeit(lambda: S.nt_raw(g_t, x_t)))
dyb = S.new_bound(dev)
drop = ops.make_dropout(0.25, 1, 0, 0)
print("old: reduce+finalize+apply(T only) %7.1f us" % timeit(
    lambda: S.bn_act_bwd(go, gb, y, coef, drop, 0.25, S.new_bound(dev), want_rows=False, act_bits=bits)))

# ---- forward: unfused (GEMM + statistics -> y, vp3d_bn_act_fwd_s16) vs fused (statistics-only pass, fused-activation pass)
from videopose3d_amd.plan import ConvSpec  # noqa: E402
spec = ConvSpec(kpad, c, 1)
xs3 = S.S16(x_rows.data.view(1024, 81, kpad), xb)
ws_ = S.split(wp)
bn = torch.nn.BatchNorm1d(c).to(dev)
st = ops.stat_buffers(m, c, dev)
bound = S.new_bound(dev)
bound[0] = 64.0
bits_f = S.new_act_bits(m, c, dev)
for cfg in (20, 22, -1):
    yy = S.conv_nt(xs3, ws_, spec, stats=st, cfg=cfg)
    cf = ops.bn_finalize(bn, m, st)
    print("forward cfg %3d: GEMM+stats -> y %6.1f us | act kernel %6.1f us | stats-only pass %6.1f us | fused-act pass p=0.25 %6.1f us, p=0 %6.1f us" % (
        cfg, timeit(lambda: S.conv_nt(xs3, ws_, spec, stats=st, cfg=cfg)),
        timeit(lambda: S.bn_act_fwd(yy, cf, drop, None, bound, act_bits=bits_f)),
        timeit(lambda: S.conv_nt(xs3, ws_, spec, stats=st, no_output=True, cfg=cfg)),
        timeit(lambda: S.conv_nt(xs3, ws_, spec, act=(cf, drop, bound, bits_f), cfg=cfg)),
        timeit(lambda: S.conv_nt(xs3, ws_, spec, act=(cf, None, bound, bits_f), cfg=cfg))))

# ---- the dedicated expand-layer kernel (vp3d_expand_fwd_s16) ---------------------------------------------------------------
yy = S.conv_nt(xs3, ws_, spec, stats=st)
cf = ops.bn_finalize(bn, m, st)
print("dedicated kernel: statistics pass %6.1f us | activation pass p=0.25 %6.1f us, p=0 %6.1f us" % (
    timeit(lambda: S.expand_fwd(xs3, ws_, stats=st)),
    timeit(lambda: S.expand_fwd(xs3, ws_, act=(cf, drop, bound, bits_f))),
    timeit(lambda: S.expand_fwd(xs3, ws_, act=(cf, None, bound, bits_f)))))

# ---- backward: P = G^T X from go + bits (vp3d_expand_bwd_p_s16) vs act_mask + GEMM ------------------------------------------
print("P from go + bits (fused kernel)   %7.1f us | + post %7.1f us" % (
    timeit(lambda: S.expand_p_from_go(go, gb, bits, 0.25, x_t)),
    timeit(lambda: S.expand_bwd(None, x_t, gram, wp, coef, m, 34, 3, 102, False, partials=S.expand_p_from_go(go, gb, bits, 0.25, x_t)))))
print("P from go + bits WITH X^T X riding along (+ slice sum) %7.1f us" % timeit(lambda: S.expand_p_from_go(go, gb, bits, 0.25, x_t, want_gram=True)))
