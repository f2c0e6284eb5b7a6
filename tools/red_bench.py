#!/usr/bin/env python3
"""Stand-alone timing of the training-backward dgrad launches with and without the fused BatchNorm-backward column sums
(vp3d_s16_red), HIP events, random operands: the shapes of the cfg3 step that carry the fusion.
    python tools/red_bench.py            (or through tools/ab_lib.py against another build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops_s16 as S  # noqa: E402
from videopose3d_amd._lib import RowMap  # noqa: E402

dev = "cuda:0"
C = 1024
torch.manual_seed(0)
print("%-22s %9s %9s %9s   (us per launch, 20 launches; separate = the reduction pass this replaces)" % (
    "dgrad M x N x K", "plain", "fused", "separate"))
for t_o, taps in ((27, 1), (9, 3), (9, 1), (3, 3), (3, 1), (1, 3)):
    B = 1024
    m, n, k = B * t_o, taps * C, C
    t_i = t_o * taps
    dy = S.split(torch.randn(B, t_o, C, device=dev) * 1e-3)
    wd = S.split(torch.randn(n, k, device=dev) * 0.03)
    y_up = torch.randn(B, t_i, C, device=dev)
    coef = torch.randn(4, C, device=dev).abs() + 0.5
    bits = torch.randint(0, 256, (B * t_i * C // 8,), dtype=torch.uint8, device=dev)
    dx = torch.empty(B, t_i, C, device=dev)
    dgb = torch.empty(2, C, device=dev)
    rm = RowMap(B, t_o, t_o, 1, 0, 0, 1)

    def run(fused):
        gb, db = S.new_bound(dev), S.new_bound(dev)
        red, hold = (S.make_red(y_up, coef, bits, 0.25, m, n, dgb[0], dgb[1], db) if fused else (None, None))
        S.gemm_rows(dy, wd, rm, C, C, n, dx, t_i * C, n, amax_out=gb, family="tconv_dgrad", red=red)
        return gb, hold

    def sep():
        gb, db = S.new_bound(dev), S.new_bound(dev)
        S.bn_act_bwd  # (the reduction alone: vp3d_bn_bwd_reduce_fin_s16 through the library handle)
        from videopose3d_amd import _lib, ops
        import ctypes as Cc
        L = _lib.lib()
        nparts, ng, nt = Cc.c_int32(0), Cc.c_int32(0), Cc.c_int32(0)
        L.vp3d_bn_bwd_reduce_fin_s16(ops._stream(), B * t_i, C, None, None, None, None, None, 0.25, None, None, None, None, None, None,
                                     None, None, Cc.byref(nparts), Cc.byref(ng), Cc.byref(nt))
        parts = torch.empty((nparts.value, 2, C), device=dev)
        gparts = torch.empty((max(ng.value, 1), 2, C), dtype=torch.float64, device=dev)
        rc = L.vp3d_bn_bwd_reduce_fin_s16(ops._stream(), B * t_i, C, dx.data_ptr(), y_up.data_ptr(), coef[2].data_ptr(),
                                          coef[3].data_ptr(), bits.data_ptr(), 0.25, coef[0].data_ptr(), gb.data_ptr(), parts.data_ptr(),
                                          gparts.data_ptr(), S._tickets(dev, nt.value).data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(),
                                          db.data_ptr(), Cc.byref(nparts), Cc.byref(ng), Cc.byref(nt))
        assert rc == 0
        return parts, gparts

    def timed(fn, n_it=20):
        for _ in range(3):
            keep = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_it):
            keep = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n_it * 1e3

    ok = S.red_supported(m, n, k, C)
    print("%-22s %9.1f %9s %9.1f" % ("%d x %d x %d" % (m, n, k), timed(lambda: run(False)),
                                     ("%9.1f" % timed(lambda: run(True))) if ok else "   (n/a)", timed(sep)), flush=True)
