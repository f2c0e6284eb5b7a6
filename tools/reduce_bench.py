#!/usr/bin/env python3
"""BatchNorm-backward reduction alone at the step's layer sizes: the full-row kernel without finalize vs the strip-owned
kernel with the fused finalize (HIP events), then a hand-off stress check against an fp64 torch reduction."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videopose3d_amd import ops, ops_s16 as S          # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


c = 1024

# ---- the reduction kernel alone: without / with the fused finalize --------------------------------------------------
import ctypes as C  # noqa: E402
from videopose3d_amd import _lib  # noqa: E402
L = _lib.lib()
for b, t in ((1024, 27), (1024, 9), (1024, 3), (1024, 1)):
    m = b * t
    y = torch.randn(b, t, c, device=DEV)
    go = torch.randn(b, t, c, device=DEV) * 1e-4
    coef = torch.stack([1 + 0.2 * torch.randn(c), 0.1 * torch.randn(c), 0.05 * torch.randn(c), 1 + 0.1 * torch.rand(c)]).to(DEV)
    gb = S.amax(go)
    bits = torch.randint(0, 255, (m * c // 8,), dtype=torch.uint8, device=DEV)
    sc, sh, mu, inv = (coef[i].data_ptr() for i in range(4))
    for cap in (0,):
        nparts, ngroups, ntick = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        L.vp3d_bn_bwd_reduce_fin_s16(ops._stream(), m, c, None, None, None, None, None, 0.25, None, None, None, None, None,
                                     None, None, None, C.byref(nparts), C.byref(ngroups), C.byref(ntick))
        parts = torch.empty((nparts.value, 2, c), dtype=torch.float32, device=DEV)
        gparts = torch.empty((ngroups.value, 2, c), dtype=torch.float64, device=DEV)
        tick = torch.zeros(max(ntick.value, 256), dtype=torch.int32, device=DEV)
        dg = torch.empty(2, c, device=DEV)
        dyb = S.new_bound(DEV)
        np2 = C.c_int32(0)
        L.vp3d_bn_bwd_reduce_bits(ops._stream(), m, c, None, None, None, None, None, 1.0, None, C.byref(np2))
        parts2 = torch.empty((np2.value, 2, c), dtype=torch.float32, device=DEV)
        t_plain = timed(lambda: L.vp3d_bn_bwd_reduce_bits(ops._stream(), m, c, go.data_ptr(), y.data_ptr(), mu, inv, bits.data_ptr(),
                                                          1.333, parts2.data_ptr(), C.byref(np2)))
        t_fin = timed(lambda: L.vp3d_bn_bwd_reduce_fin_s16(ops._stream(), m, c, go.data_ptr(), y.data_ptr(), mu, inv, bits.data_ptr(),
                                                           0.25, sc, gb.data_ptr(), parts.data_ptr(), gparts.data_ptr(), tick.data_ptr(),
                                                           dg[0].data_ptr(), dg[1].data_ptr(), dyb.data_ptr(), C.byref(nparts),
                                                           C.byref(ngroups), C.byref(ntick)))
        print("M=%6d (%d partial rows per strip): full-row reduce alone %6.1f us (%.0f GB/s) | strip-owned reduce + finalize %6.1f us"
              % (m, nparts.value, t_plain, 2 * m * c * 4 / t_plain / 1e3, t_fin), flush=True)

# ---- hand-off check: 300 launches on fresh inputs each, dgamma / dbeta vs an fp64 torch reduction ---------------------
torch.manual_seed(1)
worst = 0.0
for it in range(300):
    b, t = 1024, (27, 9, 3, 1)[it % 4]
    m = b * t
    y = torch.randn(b, t, c, device=DEV)
    go = torch.randn(b, t, c, device=DEV)
    coef = torch.stack([1 + 0.2 * torch.randn(c), 0.1 * torch.randn(c), 0.05 * torch.randn(c), 1 + 0.1 * torch.rand(c)]).to(DEV)
    gb = S.amax(go)
    drop = ops.make_dropout(0.25, 1234 + it, 5, 2)
    bits = S.new_act_bits(m, c, DEV)
    bd = S.new_bound(DEV)
    bd[0] = 64.0
    S.bn_act_fwd(y, coef, drop, None, bd, act_bits=bits)
    dyb = S.new_bound(DEV)
    _, _, dgam, dbet = S.bn_act_bwd(go, gb, y, coef, drop, 0.25, dyb, act_bits=bits, want_t=False)
    from tests.util import unpack_act_bits
    keep = torch.from_numpy(unpack_act_bits(bits, m, c)).to(DEV).view(b, t, c)
    g = (go.double() * keep) / 0.75
    xh = (y.double() - coef[2].double()) * coef[3].double()
    rb, rg = g.sum((0, 1)), (g * xh).sum((0, 1))
    e = max(float((dbet.double() - rb).abs().max() / rb.abs().max()), float((dgam.double() - rg).abs().max() / rg.abs().max()))
    worst = max(worst, e)
print("hand-off check: 300 launches, worst relative error of dgamma / dbeta vs fp64: %.3e" % worst)

# ---- the same with the inputs rotated over 5 buffer sets (1.1 GB: nothing stays in the 256 MB Infinity Cache) -----------------
for b, t in ((1024, 27), (1024, 9)):
    m = b * t
    sets = []
    for k in range(5):
        sets.append((torch.randn(b, t, c, device=DEV), torch.randn(b, t, c, device=DEV) * 1e-4,
                     torch.randint(0, 255, (m * c // 8,), dtype=torch.uint8, device=DEV)))
    coef = torch.stack([1 + 0.2 * torch.randn(c), 0.1 * torch.randn(c), 0.05 * torch.randn(c), 1 + 0.1 * torch.rand(c)]).to(DEV)
    sc, sh, mu, inv = (coef[i].data_ptr() for i in range(4))
    gb = S.amax(sets[0][1])
    nparts, ngroups, ntick, np2 = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    L.vp3d_bn_bwd_reduce_fin_s16(ops._stream(), m, c, None, None, None, None, None, 0.25, None, None, None, None, None, None, None, None,
                                 C.byref(nparts), C.byref(ngroups), C.byref(ntick))
    L.vp3d_bn_bwd_reduce_bits(ops._stream(), m, c, None, None, None, None, None, 1.0, None, C.byref(np2))
    parts = torch.empty((nparts.value, 2, c), dtype=torch.float32, device=DEV)
    parts2 = torch.empty((np2.value, 2, c), dtype=torch.float32, device=DEV)
    tick = torch.zeros(max(ntick.value, 256), dtype=torch.int32, device=DEV)
    dg = torch.empty(2, c, device=DEV)
    dyb = S.new_bound(DEV)
    it = [0]

    def plain():
        y, go, bits = sets[it[0] % 5]
        it[0] += 1
        L.vp3d_bn_bwd_reduce_bits(ops._stream(), m, c, go.data_ptr(), y.data_ptr(), mu, inv, bits.data_ptr(), 1.333, parts2.data_ptr(),
                                  C.byref(np2))

    def fused():
        y, go, bits = sets[it[0] % 5]
        it[0] += 1
        L.vp3d_bn_bwd_reduce_fin_s16(ops._stream(), m, c, go.data_ptr(), y.data_ptr(), mu, inv, bits.data_ptr(), 0.25, sc, gb.data_ptr(),
                                     parts.data_ptr(), None, tick.data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(), dyb.data_ptr(),
                                     C.byref(nparts), C.byref(ngroups), C.byref(ntick))
    print("M=%6d rotating buffers: full-row reduce alone %6.1f us (%.0f GB/s) | strip-owned reduce + finalize %6.1f us (%.0f GB/s)"
          % (m, timed(plain), 2 * m * c * 4 / timed(plain) / 1e3, timed(fused), 2 * m * c * 4 / timed(fused) / 1e3), flush=True)
    del sets
