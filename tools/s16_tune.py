#!/usr/bin/env python3
"""Sweep tile configuration x split-K of the S16 GEMM over the GEMM shapes of the cfg3 training step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from videopose3d_amd import ops, ops_s16 as S  # noqa: E402
from videopose3d_amd._lib import RowMap  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=4, warm=1):
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


def sweep(tag, m, n, k, raw):
    a = S.S16(torch.randn(m, k, device=dev), None)
    a = S.split(a.data)
    b = S.split(torch.randn(n, k, device=dev) * 0.02)
    out = torch.empty(m, n, device=dev)
    rm = RowMap(1, m, m, 1, 0, 0, 1)
    flops = 2.0 * m * n * k
    res = []
    for cfg in (20, 22, 30):
        for s in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            if cfg == 30 and s > 1:
                continue
            if s > 1 and (k // 32) // s < 4:
                continue
            if s * m * n * 4 > 2e9:
                continue
            try:
                if raw:
                    ms = timeit(lambda: S._raw_gemm(a, b, m, n, k, cfg, s))
                else:
                    st = ops.stat_buffers(m, n, dev)
                    e = ops._epi(None, False, None, st, n)
                    ms = timeit(lambda: S.gemm_rows(a, b, rm, k, k, n, out, 0, n, epi=e, cfg=cfg, splits=s))
            except Exception as ex:  # noqa: BLE001
                print("   fail", cfg, s, ex)
                continue
            res.append((ms, cfg, s))
    res.sort()
    print("   all: " + " ".join("c%d/s%d=%.0f" % (c_, s_, ms_ * 1e3) for ms_, c_, s_ in sorted(res, key=lambda r: (r[1], r[2]))), flush=True)
    pc, ps = S.plan(m, n, k, raw)
    planned = [r for r in res if r[1] == pc and r[2] == ps]
    print("%-16s M=%6d N=%5d K=%6d  best c%d s%-2d %7.3f ms %6.1f TF | 2nd c%d s%-2d %7.3f | plan c%d s%-2d %s" % (
        tag, m, n, k, res[0][1], res[0][2], res[0][0], flops / res[0][0] / 1e9, res[1][1], res[1][2], res[1][0], pc, ps,
        ("%7.3f ms" % planned[0][0]) if planned else "n/a"), flush=True)


def _raw_gemm(a, b, m, n, k, cfg, s):
    import ctypes as C
    from videopose3d_amd import _lib
    o, ws = S._opts(a, b, m, n, k, dev, None, cfg, s, raw=True)
    rm = RowMap(1, m, m, 1, 0, 0, 1)
    _lib.check(_lib.lib().vp3d_tconv_nt_s16(ops._stream(), C.byref(rm), a.data.data_ptr(), k, k, b.data.data_ptr(), k, n,
                                            None, 0, n, None, ops.zeros_page(dev).data_ptr(), C.byref(o)), "raw")


S._raw_gemm = _raw_gemm

if __name__ == "__main__":
    B = 1024
    T = [81, 27, 9, 3, 1]
    sweep("expand fwd", B * 81, 1024, 128, False)
    sweep("expand wgrad", 1024, 128, B * 81 + 0, True)
    for i in range(4):
        mo = B * T[i + 1]
        sweep("L%d fwd" % (2 * i), mo, 1024, 3072, False)
        sweep("L%d dgrad" % (2 * i), mo, 3072, 1024, False)
        sweep("L%d wgrad" % (2 * i), 1024, 3072, (mo + 63) // 64 * 64, True)
        sweep("L%d fwd/dgrad" % (2 * i + 1), mo, 1024, 1024, False)
        sweep("L%d wgrad" % (2 * i + 1), 1024, 1024, (mo + 63) // 64 * 64, True)
