#!/usr/bin/env python3
"""Where does a tile's time go?  Phase time stamps of EVERY workgroup of one k_nt_s16 launch (library built with -DVP3D_TRACE:
thread 0 leaves the 100-MHz wall clock at workgroup start / row table done / K loop done / DMA drained / statistics done /
rows stored / exit).  Build first, on the build host:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVP3D_TRACE -Iinclude -Ivideopose3d_amd/csrc \\
          videopose3d_amd/csrc/*.hip -o videopose3d_amd/libvp3d_trace.so
    python tools/gemm_trace.py [fwd1024|fwd3072|dgrad3072|red1024|red3072|eval1024]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import videopose3d_amd._lib as L  # noqa: E402

L.LIB_PATH = os.path.join(ROOT, "videopose3d_amd", "libvp3d_trace.so")
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd._lib import RowMap  # noqa: E402
from videopose3d_amd.plan import ConvSpec  # noqa: E402

dev = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 else "fwd1024"
Cc, B = 1024, 1024
torch.manual_seed(0)
lib = L.lib()
lib.vp3d_debug_trace.restype = C.c_int
lib.vp3d_debug_trace.argtypes = [C.c_void_p]
NB = 8192
trace = torch.zeros(NB * 8, dtype=torch.int64, device=dev)


def launch():
    if what.startswith("fwd") or what.startswith("eval"):
        taps = 3 if what.endswith("3072") else 1
        t_o = 27 if what.startswith("fwd") else 235
        x = S.split(torch.randn(B, t_o * taps, Cc, device=dev))
        w = S.split(torch.randn(Cc, taps * Cc, device=dev) * 0.03)
        spec = ConvSpec(Cc, Cc, taps, 1, taps)
        m = B * t_o
        if what.startswith("fwd"):
            slab = S.stat_slab_rows(*S.plan(m, Cc, taps * Cc, mix=True))
            st = ops.stat_buffers(m, Cc, dev, slab)
            return lambda: S.conv_nt(x, w, spec, stats=st, mix=True, stat_slab=slab)
        am = S.new_bounds(2, dev)
        l1 = torch.tensor([1.0, 0.1], device=dev)
        bias = torch.zeros(Cc, device=dev)
        return lambda: S.conv_nt(x, w, spec, bias=bias, relu=True, amax_out=am[0], s16_out=(x.bound, l1, None))
    taps = 3 if what.endswith("3072") else 1
    t_o = 27 if what in ("dgrad3072", "red1024") else 9
    m, n, k = B * t_o, taps * Cc, Cc
    t_i = t_o * taps
    dy = S.split(torch.randn(B, t_o, Cc, device=dev) * 1e-3)
    wd = S.split(torch.randn(n, k, device=dev) * 0.03)
    dx = torch.empty(B, t_i, Cc, device=dev)
    rm = RowMap(B, t_o, t_o, 1, 0, 0, 1)
    res = torch.randn(B, t_o, Cc, device=dev)
    e = ops._epi(residual=(res, 1, 0, Cc), n_cols=n) if taps == 3 else None
    if what.startswith("red"):
        y_up = torch.randn(B, t_i, Cc, device=dev)
        coef = torch.randn(4, Cc, device=dev).abs() + 0.5
        bits = torch.randint(0, 256, (B * t_i * Cc // 8,), dtype=torch.uint8, device=dev)
        dgb = torch.empty(2, Cc, device=dev)

        def f():
            gb, db = S.new_bound(dev), S.new_bound(dev)
            red, hold = S.make_red(y_up, coef, bits, 0.25, m, n, dgb[0], dgb[1], db)
            S.gemm_rows(dy, wd, rm, Cc, Cc, n, dx, t_i * Cc, n, epi=e, amax_out=gb, family="tconv_dgrad", red=red)
            return hold
        return f

    def g():
        gb = S.new_bound(dev)
        S.gemm_rows(dy, wd, rm, Cc, Cc, n, dx, t_i * Cc, n, epi=e, amax_out=gb, family="tconv_dgrad", mix=True)
    return g


fn = launch()
for _ in range(3):
    fn()
torch.cuda.synchronize()
assert lib.vp3d_debug_trace(trace.data_ptr()) == 0
fn()
torch.cuda.synchronize()
lib.vp3d_debug_trace(None)
t = trace.view(NB, 8).cpu().numpy().astype("float64")
used = t[:, 0] > 0
t = t[used]
t0 = t[:, 0].min()
t = (t - t0) / 100.0          # 100 MHz -> us
n = t.shape[0]
names = ["start->table", "table->K loop end", "K end->DMA drained", "drain->stats", "stats->rows stored", "stored->exit"]
print("%s: %d workgroups, launch span %.1f us" % (what, n, t[:, 6].max()))
order = t[:, 0].argsort()
t = t[order]
import numpy as np  # noqa: E402
rounds = [t[i:i + 256] for i in range(0, n, 256)]
for ri, r in enumerate(rounds):
    d = np.diff(r[:, :7], axis=1)
    print("round %d (%3d workgroups): start %.1f .. %.1f us | exit %.1f .. %.1f us" % (ri, len(r), r[:, 0].min(), r[:, 0].max(), r[:, 6].min(), r[:, 6].max()))
    print("    " + "  ".join("%s %.1f (%.1f-%.1f)" % (nm, d[:, i].mean(), d[:, i].min(), d[:, i].max()) for i, nm in enumerate(names)))
