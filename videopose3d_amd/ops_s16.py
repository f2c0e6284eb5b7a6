"""Tensor-level wrappers over the split-fp16 ("S16") entry points of the C ABI (include/vp3d.h).

An :class:`S16` pairs a buffer in S16 form (stored in a torch.float32 tensor of the LOGICAL shape: 4 bytes per
element, every 8 consecutive elements of a row = 16 B of fp16 high parts + 16 B of fp16 low parts) with the device
float that bounds the tensor's magnitude (its exponent is derived from it inside the kernels, never on the host).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib, ops
from ._lib import RowMap, S16Opts, check
from .plan import ConvSpec, ResSpec


class S16:
    __slots__ = ("data", "bound")

    def __init__(self, data: torch.Tensor, bound: Optional[torch.Tensor]):
        self.data, self.bound = data, bound

    @property
    def shape(self):
        return self.data.shape

    def bound_ptr(self):
        return None if self.bound is None else self.bound.data_ptr()


def new_bound(device) -> torch.Tensor:
    return torch.zeros(1, dtype=torch.float32, device=device)


def amax(t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """max|t| as a device float (accumulated into ``out`` when given: out = max(out, max|t|))."""
    ops._chk(t, "t")
    if out is None:
        out = new_bound(t.device)
    check(_lib.lib().vp3d_amax(ops._stream(), t.numel(), t.data_ptr(), out.data_ptr()), "vp3d_amax")
    return out


def split(t: torch.Tensor, bound: Optional[torch.Tensor] = None, measure: bool = True) -> S16:
    """fp32 tensor [..., C] (C % 8 == 0) -> S16 with the exponent of ``bound`` (measured with vp3d_amax when absent)."""
    ops._chk(t, "t")
    c = t.shape[-1]
    if bound is None and measure:
        bound = amax(t)
    out = torch.empty_like(t)
    check(_lib.lib().vp3d_split_rows(ops._stream(), t.numel() // c, c, t.data_ptr(), c, out.data_ptr(), c,
                                     None if bound is None else bound.data_ptr()), "vp3d_split_rows")
    return S16(out, bound)


def plan(m: int, n: int, k: int) -> Tuple[int, int]:
    cfg, splits = C.c_int32(0), C.c_int32(1)
    check(_lib.lib().vp3d_nt_s16_plan(m, n, k, C.byref(cfg), C.byref(splits)), "vp3d_nt_s16_plan")
    return cfg.value, splits.value


def _opts(x: S16, w: S16, m, n, k, device, amax_out=None, cfg=-1, splits=0, raw=False):
    o = S16Opts()
    o.x_bound = x.bound_ptr()
    o.w_bound = w.bound_ptr()
    o.amax_out = None if amax_out is None else amax_out.data_ptr()
    if cfg < 0 or splits <= 0:
        pc, ps = plan(m, n, k)
        cfg = pc if cfg < 0 else cfg
        splits = ps if splits <= 0 else splits
    o.cfg, o.splits = cfg, splits
    ws = None
    if splits > 1 or raw:
        ws = torch.empty((splits, m, n), dtype=torch.float32, device=device)
        o.ws, o.ws_floats = ws.data_ptr(), ws.numel()
    o.raw_partials = 1 if raw else 0
    return o, ws


def conv_nt(x: S16, wt: S16, spec: ConvSpec, *, bias=None, relu=False,
            residual: Optional[Tuple[torch.Tensor, ResSpec]] = None, stats=None, amax_out=None,
            cfg: int = -1, splits: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y (fp32) = conv(x) with the fused epilogue of ops.conv_fwd; x [B,T_in,C_in] and wt [C_out, taps*C_in] in S16."""
    xd, wd = x.data, wt.data
    b, t_in, c_in = xd.shape
    assert c_in == spec.c_in and wd.shape == (spec.c_out, spec.taps * spec.c_in), (xd.shape, wd.shape, spec)
    t_out = spec.t_out(t_in)
    if out is None:
        out = torch.empty((b, t_out, spec.c_out), dtype=torch.float32, device=xd.device)
    if spec.dil == 1:
        rm = RowMap(b, t_out, t_in, spec.stride, 0, 0, 1)
        c_src = spec.taps * c_in
    else:
        rm = RowMap(b, t_out, t_in, spec.stride, spec.dil, 0, spec.taps)
        c_src = c_in
    res = None
    if residual is not None:
        r, rs = residual
        assert r.shape[0] == b and r.shape[2] == spec.c_out
        res = (r, rs.step, rs.start, 0)
    e = ops._epi(bias, relu, res, stats, spec.c_out)
    m, k = b * t_out, spec.taps * c_in
    o, ws = _opts(x, wt, m, spec.c_out, k, xd.device, amax_out, cfg, splits)
    ops._timed_call("tconv_fwd", 2.0 * m * spec.c_out * k, _lib.lib().vp3d_tconv_nt_s16,
                    ops._stream(), C.byref(rm), xd.data_ptr(), c_in, c_src, wd.data_ptr(), wd.shape[1], spec.c_out,
                    out.data_ptr(), t_out * spec.c_out, spec.c_out, C.byref(e) if e is not None else None,
                    ops.zeros_page(xd.device).data_ptr(), C.byref(o),
                    nbytes=4.0 * (xd.numel() + wd.numel() + out.numel() + (out.numel() if residual is not None else 0)))
    return out
