"""Tensor-level wrappers over the split-fp16 ("S16") entry points of the C ABI (include/vp3d.h).

An :class:`S16` pairs a buffer in S16 form (stored in a torch.float32 tensor of the LOGICAL shape: 4 bytes per
element, every 8 consecutive elements of a row = 16 B of fp16 high parts + 16 B of fp16 low parts) with the device
float that bounds the tensor's magnitude (its exponent is derived from it inside the kernels, never on the host).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib, ops
from ._lib import RowMap, S16Fin, S16Opts, check
from ._switches import SW
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


BOUND_SLOTS = 32        # VP3D_BOUND_SLOTS: a bound = 32 floats whose maximum is the bound


def new_bound(device) -> torch.Tensor:
    return torch.zeros(BOUND_SLOTS, dtype=torch.float32, device=device)


def new_bounds(n: int, device) -> torch.Tensor:
    """n zeroed bounds: index with [i] to get the i-th (a contiguous 32-float row)."""
    return torch.zeros((n, BOUND_SLOTS), dtype=torch.float32, device=device)


def amax(t: torch.Tensor, out: Optional[torch.Tensor] = None, floor: float = 0.0) -> torch.Tensor:
    """max(floor, max|t|) as a device float (accumulated into ``out`` when given: out = max(out, ...))."""
    ops._chk(t, "t")
    if out is None:
        out = new_bound(t.device)
    if floor > 0.0:
        check(_lib.lib().vp3d_amax_floor(ops._stream(), t.numel(), t.data_ptr(), float(floor), out.data_ptr()), "vp3d_amax_floor")
    else:
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


_plan_cache = {}
# _switches.SW["tile_mix"] = "0": the planner never picks the 224- / 160-row tilings (read per call: tools/env_ab.py SW:tile_mix 0 1)


def plan(m: int, n: int, k: int, raw: bool = False, mix: bool = False, a_numel: int = 0, b_numel: int = 0) -> Tuple[int, int]:
    """(tile configuration, K slices) of an [m, n, k] split-fp16 GEMM.  mix: configuration 28 (224 x 256 tiles: statistics in
    32-row slabs, no fused activation / BatchNorm-backward sums) may be chosen.  a_numel / b_numel: elements of the tensors the
    operands are gathered from (a dilated conv's input is larger than m x k / taps; padded leading dimensions) -- the launcher's
    2-GiB test is on those extents, and the answer here must be the one the launch will accept."""
    mode = SW["tile_mix"]                             # (2 / 3: only launches of at most / more than 16,384 rows -- A/B runs)
    mix = bool(mix and mode != "0" and not (mode == "2" and m > 16384) and not (mode == "3" and m <= 16384))
    # the mixed tilings address both operands through 32-bit buffer descriptors and have no flat-address twin: not for operands
    # of 2 GiB and more (the statistics buffers are sized from this answer, so the refusal has to happen here, not at launch)
    mix = mix and max(m * k, a_numel) * 4 < 2 ** 31 and max(n * k, b_numel) * 4 < 2 ** 31
    key = (m, n, k, raw, mix)
    hit = _plan_cache.get(key)
    if hit is None:
        cfg, splits = C.c_int32(0), C.c_int32(1)
        check(_lib.lib().vp3d_nt_s16_plan(m, n, k, (1 if raw else 0) | (2 if mix else 0), C.byref(cfg), C.byref(splits)),
              "vp3d_nt_s16_plan")
        hit = _plan_cache[key] = (cfg.value, splits.value)
    return hit


def stat_slab_rows(cfg: int, splits: int = 1) -> int:
    """Rows per BatchNorm statistics slab of a launch in tile configuration cfg with `splits` K slices: 64; 32 for the 224- /
    160-row tilings in ONE slice (a split launch's statistics come from its finishing pass: 64-row slabs whatever the tiling)."""
    return 32 if (cfg in (28, 29) and splits <= 1) else 64


def _opts(x: S16, w: S16, m, n, k, device, amax_out=None, cfg=-1, splits=0, raw=False, mix=False):
    o = S16Opts()
    o.x_bound = x.bound_ptr()
    o.w_bound = w.bound_ptr()
    o.amax_out = None if amax_out is None else amax_out.data_ptr()
    if cfg < 0 or splits <= 0:
        pc, ps = plan(m, n, k, raw, mix, a_numel=x.data.numel(), b_numel=w.data.numel())
        cfg = pc if cfg < 0 else cfg
        splits = ps if splits <= 0 else splits
    o.cfg, o.splits = cfg, splits
    ws = None
    if cfg >= 100:                                    # stream-K configuration: partial-tile slots + one ticket per shared tile
        nf, nt = C.c_int64(0), C.c_int32(0)
        check(_lib.lib().vp3d_nt_s16_workspace(m, n, k, cfg, 1, 0, C.byref(nf), C.byref(nt)), "vp3d_nt_s16_workspace")
        if nf.value:
            ws = torch.empty(nf.value, dtype=torch.float32, device=device)
            o.ws, o.ws_floats = ws.data_ptr(), ws.numel()
            o.tickets = _tickets(device, nt.value).data_ptr()
    elif splits > 1 or raw:
        ws = torch.empty((splits, m, n), dtype=torch.float32, device=device)
        o.ws, o.ws_floats = ws.data_ptr(), ws.numel()
    o.raw_partials = 1 if raw else 0
    return o, ws


def conv_nt(x: S16, wt: S16, spec: ConvSpec, *, bias=None, relu=False, residual=None, stats=None, amax_out=None,
            cfg: int = -1, splits: int = 0, out: Optional[torch.Tensor] = None, s16_out=None, no_output: bool = False,
            act=None, mix: bool = False, stat_slab: int = 64, fin=None):
    """y = conv(x) with the fused epilogue of ops.conv_fwd; x [B,T_in,C_in] and wt [C_out, taps*C_in] in S16.

    residual = (tensor, ResSpec): fp32 tensor or S16 (decoded with its own bound).
    s16_out = None: y is fp32 (returned).  s16_out = (in_amax, l1, res_amax or None): y is written as S16 whose bound
    l1[0]*max(in_amax) + l1[1] + max(res_amax) is evaluated on the device; returns S16(y, that bound).
    mix: the planner may pick the 224 x 256 tiling (configuration 28); stat_slab: slab size `stats` was sized for
    (plan(..., mix=True) + stat_slab_rows(cfg) tell before the buffers are made; a mismatch is refused by the library).
    no_output: only the BatchNorm slab statistics (`stats`) are produced (returns None).
    act = (coef, drop, out_bound, act_bits or None): fused BatchNorm + ReLU + dropout epilogue -- returns the S16 rows of
    dropout(relu(y*coef[0] + coef[1])) under out_bound (and fills act_bits) without ever storing y.
    fin = (bn, momentum_dev or None): returns (y, coef) -- when the launch runs K-sliced, its finishing pass also finalises the
    BatchNorm statistics (vp3d_s16_fin: the last workgroup of a 64-column strip merges the slabs; coef = [4, C] scale, shift,
    mean, invstd, running statistics updated: bit-identical to ops.bn_finalize); coef is None when the launch ran in one slice
    (the caller finalises `stats` itself)."""
    xd, wd = x.data, wt.data
    b, t_in, c_in = xd.shape
    assert c_in == spec.c_in and wd.shape == (spec.c_out, spec.taps * spec.c_in), (xd.shape, wd.shape, spec)
    t_out = spec.t_out(t_in)
    if no_output:
        assert stats is not None and act is None and s16_out is None and residual is None and bias is None and not relu
        out = None
    elif out is None:
        out = torch.empty((b, t_out, spec.c_out), dtype=torch.float32, device=xd.device)
    if act is not None:
        assert stats is None and s16_out is None and residual is None and bias is None and not relu
    if spec.dil == 1:
        rm = RowMap(b, t_out, t_in, spec.stride, 0, 0, 1)
        c_src = spec.taps * c_in
    else:
        rm = RowMap(b, t_out, t_in, spec.stride, spec.dil, 0, spec.taps)
        c_src = c_in
    res = None
    res_s16 = None
    if residual is not None:
        r, rs = residual
        if isinstance(r, S16):
            res_s16, r = r, r.data
        assert r.shape[0] == b and r.shape[2] == spec.c_out
        res = (r, rs.step, rs.start, 0)
    e = ops._epi(bias, relu, res, stats, spec.c_out)
    m, k = b * t_out, spec.taps * c_in
    if s16_out is not None or res_s16 is not None or no_output or act is not None:
        splits = 1
    o, ws = _opts(x, wt, m, spec.c_out, k, xd.device, amax_out, cfg, splits, mix=mix and act is None)
    o.stat_slab_rows = stat_slab
    keep = None
    coef_fin = fin_s = None
    if (fin is not None and o.splits > 1 and stats is not None and not no_output and act is None and s16_out is None and
            res_s16 is None):       # (vp3d_tconv_nt_s16 runs an S16 residual in ONE slice: no finishing pass to finalise in)
        bn, momentum_dev = fin
        track = bn.track_running_stats and bn.running_mean is not None
        if bn.momentum is not None and b * t_out > 1:
            coef_fin = torch.empty((4, spec.c_out), dtype=torch.float32, device=xd.device)
            use_dev = momentum_dev is not None and track
            fin_s = S16Fin(bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), 0.0 if use_dev else float(bn.momentum),
                           momentum_dev if use_dev else None, bn.running_mean.data_ptr() if track else None,
                           bn.running_var.data_ptr() if track else None,
                           bn.num_batches_tracked.data_ptr() if (track and bn.num_batches_tracked is not None) else None,
                           coef_fin[0].data_ptr(), coef_fin[1].data_ptr(), coef_fin[2].data_ptr(), coef_fin[3].data_ptr(),
                           _fin_tickets(xd.device, (spec.c_out + 63) // 64).data_ptr())
            o.fin = C.addressof(fin_s)
    if no_output:
        o.no_output = 1
    if act is not None:
        coef, drop, out_bound, act_bits = act
        o.act_scale, o.act_shift, o.act_bound = coef[0].data_ptr(), coef[1].data_ptr(), out_bound.data_ptr()
        o.act_bits = None if act_bits is None else act_bits.data_ptr()
        if drop is not None:
            keep = drop
            o.act_drop = C.pointer(drop)
    if res_s16 is not None:
        o.res_s16, o.res_bound = 1, res_s16.bound.data_ptr()
    wbound = None
    if s16_out is not None:
        in_amax, l1, res_amax = s16_out
        wbound = new_bound(xd.device)
        o.out_s16, o.in_amax, o.l1 = 1, in_amax.data_ptr(), l1.data_ptr()
        o.res_amax = None if res_amax is None else res_amax.data_ptr()
        o.out_wbound = wbound.data_ptr()
    ops._timed_call("tconv_fwd", 2.0 * m * spec.c_out * k, _lib.lib().vp3d_tconv_nt_s16,
                    ops._stream(), C.byref(rm), xd.data_ptr(), c_in, c_src, wd.data_ptr(), wd.shape[1], spec.c_out,
                    ops._p(out), t_out * spec.c_out, spec.c_out, C.byref(e) if e is not None else None,
                    ops.zeros_page(xd.device).data_ptr(), C.byref(o),
                    nbytes=4.0 * (xd.numel() + wd.numel() + (0 if out is None else out.numel()) +
                                  (out.numel() if residual is not None else 0)),
                    shape=(m, spec.c_out, k, o.cfg, o.splits, 2 if o.cfg == 30 else 1))
    del keep, fin_s
    if no_output:
        return None
    if act is not None:
        return S16(out, act[2])
    if fin is not None:
        return out, coef_fin
    return out if s16_out is None else S16(out, wbound)


def expand_fwd(x: S16, wt: S16, *, stats=None, act=None) -> Optional[S16]:
    """The expand layer's dedicated forward kernel (vp3d_expand_fwd_s16): x S16 im2row rows [B, T, kpad], wt S16 [N, kpad].
    stats = (sum, m2): statistics pass (returns None).  act = (coef, drop, out_bound, act_bits or None): activation pass,
    returns the S16 rows [B, T, N] of dropout(relu(x wt^T * coef[0] + coef[1])) -- bit for bit what conv_nt + bn_act_fwd make."""
    xd, wd = x.data, wt.data
    b, t, kpad = xd.shape
    n = wd.shape[0]
    assert wd.shape[1] == kpad and (stats is None) != (act is None)
    m = b * t
    L = _lib.lib()
    if stats is not None:
        ops._timed_call("tconv_fwd", 2.0 * m * n * kpad, L.vp3d_expand_fwd_s16, ops._stream(), m, n, kpad, xd.data_ptr(),
                        x.bound_ptr(), wd.data_ptr(), wt.bound_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), None, None, None, None,
                        None, None, nbytes=4.0 * (xd.numel() + wd.numel()), shape=(m, n, kpad, "ex", 1, 1))
        return None
    coef, drop, out_bound, act_bits = act
    out = torch.empty((b, t, n), dtype=torch.float32, device=xd.device)
    ops._timed_call("tconv_fwd", 2.0 * m * n * kpad, L.vp3d_expand_fwd_s16, ops._stream(), m, n, kpad, xd.data_ptr(), x.bound_ptr(),
                    wd.data_ptr(), wt.bound_ptr(), None, None, coef[0].data_ptr(), coef[1].data_ptr(),
                    C.byref(drop) if drop is not None else None, out_bound.data_ptr(), out.data_ptr(), ops._p(act_bits),
                    nbytes=4.0 * (xd.numel() + wd.numel() + out.numel()), shape=(m, n, kpad, "ex", 1, 1))
    return S16(out, out_bound)


def expand_stats_gram(x_t: S16, w_packed: torch.Tensor, bn: torch.nn.BatchNorm1d, m_rows: int, kv: int, one_col: int,
                      momentum_dev: Optional[int] = None, want_gram: bool = False, illcond: Optional[torch.Tensor] = None):
    """[4, C] = scale, shift, mean, invstd of the expand layer's training-mode BatchNorm (running statistics updated in place)
    from the centred second-moment matrix of the layer's 128-column input -- no pass over the conv output
    (vp3d_expand_stats_gram_s16; replaces expand_fwd(stats=...) + ops.bn_finalize).  x_t: the transposed S16 copy of the im2row
    rows [kpad][pitch]; w_packed: fp32 weight rows [C][kpad]; one_col: the constant-1 padding column of the rows.
    want_gram: also return the matrix itself (float64 [kpad][kpad]) -- the backward of the layer rebuilds X^T X from it
    (expand_bwd(gram_centred=x_t)) instead of forming its own.
    illcond: int32 device scalar that receives max over channels of floor(log2 kappa_n), kappa_n = sum |w_i Cov_ij w_j| /
    (var_n + eps) -- how ill-conditioned the quadratic form is (range_guard moves the layer back to the statistics pass at
    GRAM_KAPPA_LOG2_MAX)."""
    xd = x_t.data
    kpad, ld_t = xd.shape
    c = bn.num_features
    assert w_packed.shape == (c, kpad) and w_packed.dtype == torch.float32 and w_packed.is_contiguous()
    if m_rows <= 1:
        raise ValueError("Expected more than 1 value per channel when training, got input size [%d, %d]" % (m_rows, c))
    L = _lib.lib()
    groups = int(L.vp3d_expand_stats_gram_groups(m_rows))
    dev = xd.device
    part = torch.empty((groups, kpad, kpad), dtype=torch.float32, device=dev)
    gram = torch.empty((kpad, kpad), dtype=torch.float64, device=dev)
    buf = torch.empty((4, c), dtype=torch.float32, device=dev)
    track = bn.track_running_stats and bn.running_mean is not None
    assert bn.momentum is not None or not track, "cumulative-average BatchNorm: use ops.bn_finalize"
    momentum = 0.0 if bn.momentum is None else float(bn.momentum)
    use_dev = momentum_dev is not None and track
    with ops._Timed("stream_expand_stats_gram", 0.0, 4.0 * xd.numel(), (m_rows, c)):
        check(L.vp3d_expand_stats_gram_s16(ops._stream(), m_rows, c, kpad, kv, one_col, xd.data_ptr(), ld_t, x_t.bound_ptr(),
                                           w_packed.data_ptr(), part.data_ptr(), gram.data_ptr(), bn.weight.data_ptr(),
                                           bn.bias.data_ptr(), float(bn.eps), 0.0 if use_dev else momentum,
                                           momentum_dev if use_dev else None,
                                           bn.running_mean.data_ptr() if track else None,
                                           bn.running_var.data_ptr() if track else None,
                                           bn.num_batches_tracked.data_ptr() if (track and bn.num_batches_tracked is not None) else None,
                                           buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), buf[3].data_ptr(),
                                           illcond.data_ptr() if illcond is not None else None),
              "vp3d_expand_stats_gram_s16")
    return (buf, gram) if want_gram else buf


_red_ok = {}


def red_cfg(m: int, n: int, k: int, c_up: int) -> int:
    """Tile configuration in which a dgrad launch [m, n, k] can carry the BatchNorm-backward column sums of its upstream
    activation (vp3d_s16_red), or 0: one K slice on the 128 x 128 / 256 x 256 / 224 x 256 buffer-descriptor tilings (20 / 22 /
    28), whole column tiles.  (Cached: asked per launch.)"""
    key = (m, n, k, c_up, SW["tile_mix"])
    hit = _red_ok.get(key)
    if hit is None:
        hit = 0
        if c_up > 0 and c_up % 256 == 0 and n % c_up == 0:
            cfg, splits = plan(m, n, k, mix=True)
            if not (cfg == 28 and splits == 1):
                cfg, splits = plan(m, n, k)
            if cfg in (20, 22, 28) and splits == 1 and n % (128 if cfg == 20 else 256) == 0:
                hit = cfg
        _red_ok[key] = hit
    return hit


def red_supported(m: int, n: int, k: int, c_up: int) -> bool:
    return red_cfg(m, n, k, c_up) != 0


RED_CFG_KEY = 1000              # profiler rows: tile configuration + 1000 = the instance that carries the fused sums
RED_CALLS = {"n": 0}            # dgrad launches that carried the fused BatchNorm-backward sums (tests, tools)


def make_red(y_up: torch.Tensor, coef: torch.Tensor, act_bits: torch.Tensor, p: float, m: int, n: int, dgamma: torch.Tensor,
             dbeta: torch.Tensor, dy_bound: torch.Tensor):
    """vp3d_s16_red for a dgrad launch [m rows, n columns] whose result is the gradient of dropout(relu(bn(y_up))): coef =
    the upstream layer's (scale, shift, mean, invstd) rows; dgamma / dbeta / dy_bound (zeroed 32 slots) receive what
    bn_act_bwd's reduction pass would.  Returns (struct, buffers to keep alive until the launch is enqueued)."""
    bu, tu, cu = y_up.shape
    ops._chk(y_up, "y_up")
    assert act_bits.numel() * 8 == bu * tu * cu and act_bits.dtype == torch.uint8
    parts = torch.empty(((m + 127) // 128) * 2 * n, dtype=torch.float32, device=y_up.device)
    r = _lib.S16Red()
    r.y_up, r.mean, r.invstd, r.scale = y_up.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), coef[0].data_ptr()
    r.act_bits, r.rows_up, r.c_up, r.p = act_bits.data_ptr(), bu * tu, cu, float(p)
    r.partials, r.partials_floats = parts.data_ptr(), parts.numel()
    r.tickets = _tickets(y_up.device, 2 * (cu // 128)).data_ptr()
    r.dgamma, r.dbeta, r.dy_bound = dgamma.data_ptr(), dbeta.data_ptr(), dy_bound.data_ptr()
    RED_CALLS["n"] += 1
    return r, (parts,)


def gemm_rows(x: S16, wt: S16, rm: RowMap, c_in: int, c_src: int, n: int, out: torch.Tensor, y_bpitch: int, ldy: int,
              *, epi=None, amax_out=None, family="tconv_fwd", cfg: int = -1, splits: int = 0, red=None, mix: bool = False):
    """Raw form of conv_nt: out[b*y_bpitch + t*ldy + n] = sum_k x[gather] * wt[n][k] (+ epilogue).  red: a make_red()
    struct -- the launch also reduces the BatchNorm backward of the activation whose gradient it writes."""
    xd, wd = x.data, wt.data
    m, k = rm.batch * rm.t_dst, rm.taps * c_in
    if red is not None and cfg < 0:
        cfg, splits = red_cfg(m, n, k, red.c_up), 1  # (the configuration red_supported() answered for)
    o, ws = _opts(x, wt, m, n, k, xd.device, amax_out, cfg, splits, mix=mix and red is None)
    if red is not None:
        o.red = C.addressof(red)
    ops._timed_call(family, 2.0 * m * n * k, _lib.lib().vp3d_tconv_nt_s16,
                    ops._stream(), C.byref(rm), xd.data_ptr(), xd.shape[-1], c_src, wd.data_ptr(), wd.shape[-1], n,
                    out.data_ptr(), y_bpitch, ldy, C.byref(epi) if epi is not None else None,
                    ops.zeros_page(xd.device).data_ptr(), C.byref(o),
                    nbytes=4.0 * (xd.numel() + wd.numel() + m * n) + (4.125 * m * n if red is not None else 0.0),
                    shape=(m, n, k, o.cfg + (RED_CFG_KEY if red is not None else 0), o.splits, 2 if o.cfg == 30 else 1))
    return out


def wgrad(dy_t: S16, x_t: S16, c_out: int, c_in: int, taps: int, n_cols: int, out: Optional[torch.Tensor] = None,
          flops_rows: int = 0) -> torch.Tensor:
    """dW [c_out, c_in, taps] (reference layout) from the transposed operands dy_t [c_out][Mp], x_t [n_cols][Mp]
    (S16 rows along the reduction index m; n_cols >= taps*c_in): one NT GEMM with K = Mp into raw split-K partial
    matrices, summed and un-packed by vp3d_wgrad_reduce."""
    mp = dy_t.data.shape[-1]
    assert dy_t.data.shape == (c_out, mp) and x_t.data.shape == (n_cols, mp), (dy_t.data.shape, x_t.data.shape)
    dev = dy_t.data.device
    cfg, splits = plan(c_out, n_cols, mp, raw=True)
    o, ws = _opts(dy_t, x_t, c_out, n_cols, mp, dev, None, cfg, splits, raw=True)
    rm = RowMap(1, c_out, c_out, 1, 0, 0, 1)
    ops._timed_call("tconv_wgrad", 2.0 * (flops_rows or mp) * c_out * taps * c_in, _lib.lib().vp3d_tconv_nt_s16,
                    ops._stream(), C.byref(rm), dy_t.data.data_ptr(), mp, mp, x_t.data.data_ptr(), mp, n_cols,
                    None, 0, n_cols, None, ops.zeros_page(dev).data_ptr(), C.byref(o),
                    nbytes=4.0 * (dy_t.data.numel() + x_t.data.numel() + c_out * n_cols), shape=(c_out, n_cols, mp, cfg, splits, 1))
    if out is None:
        out = torch.empty((c_out, c_in, taps), dtype=torch.float32, device=dev)
    check(_lib.lib().vp3d_wgrad_reduce(ops._stream(), ws.data_ptr(), n_cols, splits, c_out, c_in, taps, out.data_ptr()),
          "vp3d_wgrad_reduce")
    return out


def gather_t(x: S16, spec: ConvSpec, t_out: int) -> S16:
    """Transposed weight-gradient operand [taps*C][roundup(B*t_out, 64)] of conv `spec` gathered from the S16 rows of its
    input x [B,T_in,C] (vp3d_gather_t_s16): any stride / dilation / taps."""
    b, t_in, c = x.data.shape
    m = b * t_out
    out = torch.empty((spec.taps * c, t_pitch(m)), dtype=torch.float32, device=x.data.device)
    rm = RowMap(b, t_out, t_in, spec.stride, spec.dil, 0, spec.taps)
    check(_lib.lib().vp3d_gather_t_s16(ops._stream(), C.byref(rm), x.data.data_ptr(), c, out.data_ptr(), out.shape[1]),
          "vp3d_gather_t_s16")
    return S16(out, x.bound)


def nt_raw(a_t: S16, b_t: S16) -> Tuple[torch.Tensor, int]:
    """Raw split-K partials [splits][NA][NB] of  a_t @ b_t^T  for two transposed S16 operands [NA][Mp], [NB][Mp] (rows
    along the reduction index): the weight-gradient GEMM without its un-pack.  Returns (partials, splits)."""
    mp = a_t.data.shape[-1]
    na, nb = a_t.data.shape[0], b_t.data.shape[0]
    assert b_t.data.shape[-1] == mp
    dev = a_t.data.device
    cfg, splits = plan(na, nb, mp, raw=True)
    o, ws = _opts(a_t, b_t, na, nb, mp, dev, None, cfg, splits, raw=True)
    rm = RowMap(1, na, na, 1, 0, 0, 1)
    ops._timed_call("tconv_wgrad", 2.0 * mp * na * nb, _lib.lib().vp3d_tconv_nt_s16,
                    ops._stream(), C.byref(rm), a_t.data.data_ptr(), mp, mp, b_t.data.data_ptr(), mp, nb,
                    None, 0, nb, None, ops.zeros_page(dev).data_ptr(), C.byref(o),
                    nbytes=4.0 * (a_t.data.numel() + b_t.data.numel() + na * nb), shape=(na, nb, mp, cfg, splits, 1))
    return ws, splits


def act_mask(go: torch.Tensor, go_bound: torch.Tensor, act_bits: torch.Tensor, p: float, transposed: bool) -> S16:
    """G = go * keep * [bn(y) > 0] (the expand layer's backward) as S16 rows [B,T,C], or as a transposed S16 operand
    [C][roundup(M, 64)]."""
    ops._chk(go, "go")
    b, t, c = go.shape
    m = b * t
    out = torch.empty((c, t_pitch(m)), dtype=torch.float32, device=go.device) if transposed else torch.empty_like(go)
    gb = new_bound(go.device)
    check(_lib.lib().vp3d_act_mask_s16(ops._stream(), m, c, go.data_ptr(), go_bound.data_ptr(), act_bits.data_ptr(), float(p),
                                       gb.data_ptr(), None if transposed else out.data_ptr(), out.data_ptr() if transposed else None,
                                       out.shape[1] if transposed else 0), "vp3d_act_mask_s16")
    return S16(out, gb)


def gram(x_t: S16) -> torch.Tensor:
    """X^T X [kpad][kpad] as doubles from the transposed S16 copy of X (one small split-K GEMM + the slice sum)."""
    ws, splits = nt_raw(x_t, x_t)
    n = x_t.data.shape[0]
    out = torch.empty((n, n), dtype=torch.float64, device=ws.device)
    check(_lib.lib().vp3d_sum_slices(ops._stream(), n * n, splits, ws.data_ptr(), out.data_ptr()), "vp3d_sum_slices")
    return out


def expand_rows_form(c_out: int, kpad: int) -> bool:
    """P = G^T X straight from the S16 rows (k_tn_s16<1>, the narrow-B form of the rows-form weight gradient) instead of
    from transposed copies.  Opt-in (_switches.SW["expand_rows"]): measured SLOWER than the NT GEMM on transposed copies for this
    shape (M 82,944 x 1024 x 128 stand-alone: 137 us vs 110 us, tools/expand_bwd_bench.py) -- with a 128-column B tile
    the kernel issues 20 transpose reads per 12 MFMAs and is LDS-bound."""
    return SW["expand_rows"] and kpad == 128 and c_out % 256 == 0


def expand_p_from_go(go: torch.Tensor, go_bound: torch.Tensor, act_bits: torch.Tensor, p: float, x_t: S16, want_gram: bool = False):
    """Raw partials [n][C][kpad] of P = G^T X straight from the incoming gradient go [B, T, C] and the activation bits
    (vp3d_expand_bwd_p_s16: G = go * keep * [bn(y) > 0] is formed in registers, go is read once).  x_t: transposed S16 X.
    Returns (partials, n) or, with want_gram, (partials, n, X^T X as doubles [kpad][kpad]) -- X^T X rides along in the same
    launch (the X^T fragments are its operands) and is folded by vp3d_sum_slices."""
    ops._chk(go, "go")
    b, t, c = go.shape
    m = b * t
    kpad, ld_t = x_t.data.shape
    n = C.c_int32(0)
    L = _lib.lib()
    check(L.vp3d_expand_bwd_p_s16(ops._stream(), m, c, kpad, None, None, None, float(p), None, ld_t, None, None, None, C.byref(n)),
          "vp3d_expand_bwd_p_s16(query)")
    ws = torch.empty((n.value, c, kpad), dtype=torch.float32, device=go.device)
    gws = torch.empty((n.value, kpad, kpad), dtype=torch.float32, device=go.device) if want_gram else None
    ops._timed_call("tconv_wgrad", 2.0 * m * c * kpad, L.vp3d_expand_bwd_p_s16, ops._stream(), m, c, kpad, go.data_ptr(),
                    go_bound.data_ptr(), act_bits.data_ptr(), float(p), x_t.data.data_ptr(), ld_t, x_t.bound_ptr(), ws.data_ptr(),
                    ops._p(gws), C.byref(n), nbytes=4.0 * (go.numel() + x_t.data.numel() + c * kpad) + act_bits.numel(),
                    shape=(c, kpad, m, "exb", n.value, 1))
    if not want_gram:
        return ws, n.value
    gram_xx = torch.empty((kpad, kpad), dtype=torch.float64, device=go.device)
    check(L.vp3d_sum_slices(ops._stream(), kpad * kpad, n.value, gws.data_ptr(), gram_xx.data_ptr()), "vp3d_sum_slices")
    return ws, n.value, gram_xx


def expand_bwd(g: Optional[S16], x: S16, gram_xx: torch.Tensor, w_packed: torch.Tensor, coef: torch.Tensor, m_rows: int, c_in: int,
               taps: int, one_col: int, rows: bool, out_dw=None, out_dgamma=None, out_dbeta=None, partials=None,
               gram_centred: Optional[S16] = None):
    """(dW [C][c_in][taps], dgamma, dbeta) of the expand layer from G^T X, X^T X and the packed weight (see include/vp3d.h).
    rows: g [.., C] and x [.., kpad] are S16 rows (vp3d_wgrad_rows_s16); else both are transposed operands [C or kpad][Mp].
    gram_centred: the transposed S16 X when ``gram_xx`` is the FORWARD's centred second-moment matrix (expand_stats_gram's
    second result) instead of X^T X -- the launch rebuilds X^T X from it in fp64 (vp3d_expand_bwd_gram_s16)."""
    dev = x.data.device
    if partials is not None:                          # (ws, splits) from expand_p_from_go
        ws, splits = partials
        c, kpad = ws.shape[1], ws.shape[2]
    elif rows:
        c, kpad = g.data.shape[-1], x.data.shape[-1]
        assert expand_rows_form(c, kpad) and g.data.numel() // c == m_rows and x.data.numel() // kpad == m_rows
        splits = max(1, min(64, ((m_rows + 31) // 32) // 6, 512 // (c // 256)))
        ws = torch.empty((splits, c, kpad), dtype=torch.float32, device=dev)
        ops._timed_call("tconv_wgrad", 2.0 * m_rows * c * kpad, _lib.lib().vp3d_wgrad_rows_s16, ops._stream(), m_rows,
                        g.data.data_ptr(), c, c, g.bound_ptr(), x.data.data_ptr(), kpad, 1, kpad, x.bound_ptr(), splits,
                        ws.data_ptr(), nbytes=4.0 * (g.data.numel() + x.data.numel() + c * kpad), shape=(c, kpad, m_rows, "tn", splits, 1))
    else:
        c, kpad = g.data.shape[0], x.data.shape[0]
        ws, splits = nt_raw(g, x)
    dw = out_dw if out_dw is not None else torch.empty((c, c_in, taps), dtype=torch.float32, device=dev)
    if out_dgamma is not None and out_dbeta is not None:
        dgam, dbet = out_dgamma, out_dbeta
    else:
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        dgam, dbet = dgb[0], dgb[1]
    assert gram_xx.dtype == torch.float64 and gram_xx.shape == (kpad, kpad)
    xt = gram_centred
    check(_lib.lib().vp3d_expand_bwd_gram_s16(ops._stream(), c, c_in, taps, kpad, one_col, m_rows, splits, ws.data_ptr(),
                                              gram_xx.data_ptr(), xt.data.data_ptr() if xt is not None else None,
                                              xt.data.shape[1] if xt is not None else 0,
                                              xt.bound_ptr() if xt is not None else None, w_packed.data_ptr(),
                                              coef[0].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), dgam.data_ptr(),
                                              dbet.data_ptr(), dw.data_ptr()),
          "vp3d_expand_bwd_gram_s16")
    return dw, dgam, dbet


def wgrad_rows_supported(c_out: int, c_in: int) -> bool:
    return c_out % 256 == 0 and (c_in % 256 == 0 or c_in == 128)


# Slice counts the rows-form weight gradient may be cut into.  5 / 10 / 15 / 20 matter for the 48 tiles of a 1024 x 3072 weight
# gradient: 48 x 5 = 240 workgroups are one (94 % full) round of the 256 CUs, where 4 slices leave a quarter of the chip idle and
# 16 slices (three exact rounds) write 201 MB of partial matrices instead of 63 (tools/wgrad_splits.py: M 27,648: 479 -> 444 us,
# M 9,216: 174 -> 162 us, GEMM + reduce)
WGRAD_SPLIT_CANDIDATES = (1, 2, 3, 4, 5, 6, 8, 10, 12, 15, 16, 20, 24, 32)


def _wgrad_rows_splits(m_rows: int, c_out: int, n_cols: int) -> int:
    """K-slices of the 256x256 rows-form wgrad GEMM: the S16 planner's 256x256 cost terms (plan_nt_s16)."""
    tiles, nkt = (c_out // 256) * max(1, n_cols // 256), (m_rows + 31) // 32
    best, best_s = None, 1
    for s in WGRAD_SPLIT_CANDIDATES:
        if s > 1 and nkt // s < 6:
            break
        wgs = tiles * s
        per_cu, nk = (wgs + 255) // 256, (nkt + s - 1) // s
        rate = 2.22 if per_cu >= 2 else 1.68 + 0.54 * min(1.0, wgs / 256.0)
        cost = per_cu * (nk * rate + 8.3) + s * c_out * n_cols * 4.0 / 7.4e6
        if best is None or cost < best * 0.98:
            best, best_s = cost, s
    return best_s


def wgrad_rows(dy: S16, x: S16, c_out: int, c_in: int, taps: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW [c_out, c_in, taps] (reference layout) from the S16 ROWS dy [.., c_out] (M rows) and x [.., c_in] (M*taps rows:
    the input rows of a conv of stride == taps): vp3d_wgrad_rows_s16 transposes on the LDS read, so no transposed copies
    of either operand are needed; split-K partials are summed and un-packed by vp3d_wgrad_reduce."""
    dd, xd = dy.data, x.data
    m = dd.numel() // c_out
    assert dd.shape[-1] == c_out and xd.shape[-1] == c_in and xd.numel() // c_in == m * taps, (dd.shape, xd.shape, taps)
    dev = dd.device
    n_cols = taps * c_in
    splits = _wgrad_rows_splits(m, c_out, n_cols)
    ws = torch.empty((splits, c_out, n_cols), dtype=torch.float32, device=dev)
    ops._timed_call("tconv_wgrad", 2.0 * m * c_out * n_cols, _lib.lib().vp3d_wgrad_rows_s16, ops._stream(), m, dd.data_ptr(),
                    c_out, c_out, dy.bound_ptr(), xd.data_ptr(), c_in, taps, c_in, x.bound_ptr(), splits, ws.data_ptr(),
                    nbytes=4.0 * (dd.numel() + xd.numel() + c_out * n_cols), shape=(c_out, n_cols, m, "tn", splits, 1))
    if out is None:
        out = torch.empty((c_out, c_in, taps), dtype=torch.float32, device=dev)
    check(_lib.lib().vp3d_wgrad_reduce(ops._stream(), ws.data_ptr(), n_cols, splits, c_out, c_in, taps, out.data_ptr()),
          "vp3d_wgrad_reduce")
    return out


def t_pitch(m_rows: int, taps: int = 1) -> int:
    return (m_rows // taps + 63) // 64 * 64


def split_t(t2d: torch.Tensor, bound: torch.Tensor, want_rows=True, want_t=True):
    """fp32 [M, C] -> (S16 rows or None, S16 transposed [C][roundup(M,64)] or None)."""
    ops._chk(t2d, "t")
    m, c = t2d.shape
    rows = torch.empty_like(t2d) if want_rows else None
    tt = torch.empty((c, t_pitch(m)), dtype=torch.float32, device=t2d.device) if want_t else None
    check(_lib.lib().vp3d_split_t(ops._stream(), m, c, t2d.data_ptr(), c, bound.data_ptr(), ops._p(rows), c, ops._p(tt),
                                  tt.shape[1] if tt is not None else 0), "vp3d_split_t")
    return (S16(rows, bound) if rows is not None else None), (S16(tt, bound) if tt is not None else None)


def im2row_split(x: torch.Tensor, spec: ConvSpec, kpad: int, one_col: int, bound: torch.Tensor, want_t=True):
    """ops.im2row + split_t in one pass (vp3d_im2row_split_s16): [B, T_in, C_in] -> (S16 im2row rows [B, T_out, kpad], their
    transposed S16 copy [kpad][roundup(M, 64)] or None); bound must cover max|x| and the bias column's 1."""
    ops._chk(x, "x")
    assert spec.dil == 1 and kpad % 64 == 0
    b, t_in, c_in = x.shape
    t_out = spec.t_out(t_in)
    m = b * t_out
    rows = torch.empty((b, t_out, kpad), dtype=torch.float32, device=x.device)
    tt = torch.empty((kpad, t_pitch(m)), dtype=torch.float32, device=x.device) if want_t else None
    rm = RowMap(b, t_out, t_in, spec.stride, 0, 0, 1)
    check(_lib.lib().vp3d_im2row_split_s16(ops._stream(), C.byref(rm), x.data_ptr(), c_in, spec.taps * c_in, kpad, one_col,
                                           bound.data_ptr(), rows.data_ptr(), ops._p(tt), tt.shape[1] if tt is not None else 0),
          "vp3d_im2row_split_s16")
    return S16(rows, bound), (S16(tt, bound) if tt is not None else None)


def pack_weight(w: torch.Tensor, bound: torch.Tensor, want_fwd=True, want_dgrad=True, dilated_form=False):
    """Conv1d.weight [C_out, C_in, taps] -> (S16 forward pack [C_out, taps*C_in], S16 dgrad pack)."""
    ops._chk(w, "weight")
    c_out, c_in, taps = w.shape
    wf = torch.empty((c_out, taps * c_in), dtype=torch.float32, device=w.device) if want_fwd else None
    wd = None
    if want_dgrad:
        wd = torch.empty((c_in, taps * c_out) if dilated_form else (taps * c_in, c_out), dtype=torch.float32, device=w.device)
    check(_lib.lib().vp3d_pack_weight_s16(ops._stream(), w.data_ptr(), c_out, c_in, taps, bound.data_ptr(), ops._p(wf),
                                          taps * c_in, ops._p(wd), wd.shape[1] if wd is not None else 0,
                                          1 if dilated_form else 0), "vp3d_pack_weight_s16")
    return (S16(wf, bound) if wf is not None else None), (S16(wd, bound) if wd is not None else None)


def act_bound(bn: torch.nn.BatchNorm1d, m_rows: int, p: float, res_bound: Optional[torch.Tensor], out: torch.Tensor):
    check(_lib.lib().vp3d_act_bound(ops._stream(), bn.num_features, m_rows, bn.weight.data_ptr(), bn.bias.data_ptr(), float(p),
                                    ops._p(res_bound), out.data_ptr()), "vp3d_act_bound")
    return out


def bn_act_fwd(y: torch.Tensor, coef: torch.Tensor, drop, residual: Optional[Tuple[S16, ResSpec]], out_bound: torch.Tensor,
               t_taps: int = 0, want_f32: bool = False, act_bits: Optional[torch.Tensor] = None):
    """a = [res +] dropout(relu(bn(y))) as S16 rows [B,T,C] (+ the transposed copy for the consuming conv's wgrad when
    t_taps > 0: [t_taps*C][roundup(M/t_taps, 64)]).  act_bits (uint8 [M*C/8], see new_act_bits) receives the
    [bn(y) > 0 and kept] bits for bn_act_bwd."""
    ops._chk(y, "y")
    b, t, c = y.shape
    m = b * t
    out = torch.empty_like(y)
    tt = None
    if t_taps:
        assert m % t_taps == 0
        tt = torch.empty((t_taps * c, t_pitch(m, t_taps)), dtype=torch.float32, device=y.device)
    if residual is not None:
        r, rs = residual
        rd = r.data
        assert rd.shape[0] == b and rd.shape[2] == c and rs.start + rs.step * (t - 1) < rd.shape[1]
        rargs = (rd.data_ptr(), r.bound.data_ptr(), t, rd.shape[1], rs.step, rs.start, c)
    else:
        rargs = (None, None, t, 0, 0, 0, c)
    f32 = torch.empty_like(y) if want_f32 else None
    nb = y.numel() * (8.0 + (4.0 if residual is not None else 0.0) + (4.0 if tt is not None else 0.0) +
                      (4.0 if want_f32 else 0.0) + (0.125 if act_bits is not None else 0.0))
    with ops._Timed("stream_bn_act_fwd", 0.0, nb, (m, c)):
        check(_lib.lib().vp3d_bn_act_fwd_s16(ops._stream(), m, c, y.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                             C.byref(drop) if drop is not None else None, *rargs, out_bound.data_ptr(),
                                             out.data_ptr(), ops._p(f32), ops._p(tt), tt.shape[1] if tt is not None else 0,
                                             max(t_taps, 1), ops._p(act_bits)), "vp3d_bn_act_fwd_s16")
    if want_f32:
        return S16(out, out_bound), (S16(tt, out_bound) if tt is not None else None), f32
    return S16(out, out_bound), (S16(tt, out_bound) if tt is not None else None)


_ticket_pool = {}
_fin_ticket_pool = {}


def _fin_tickets(device, n: int) -> torch.Tensor:
    """Zeroed int32 tickets of the finalize-in-the-finishing-pass hand-over (vp3d_s16_fin): as _tickets, its own buffer."""
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    t = _fin_ticket_pool.get(key)
    if t is None or t.numel() < n:
        t = _fin_ticket_pool[key] = torch.zeros(max(n, 64), dtype=torch.int32, device=device)
    return t



def _tickets(device, n: int) -> torch.Tensor:
    """Zeroed int32 tickets of the last-arriver reductions, one buffer per (device, stream): the kernels leave them zero,
    launches on one stream are ordered, so the buffer is allocated (and zeroed) once."""
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    t = _ticket_pool.get(key)
    if t is None or t.numel() < n:
        t = _ticket_pool[key] = torch.zeros(max(n, 256), dtype=torch.int32, device=device)
    return t


def new_act_bits(m_rows: int, c: int, device) -> torch.Tensor:
    """Buffer for the activation bits of an [m_rows, c] activation (1 bit per element)."""
    assert c % 64 == 0
    return torch.empty(m_rows * c // 8, dtype=torch.uint8, device=device)


def bn_act_bwd(go: torch.Tensor, go_bound: torch.Tensor, y: torch.Tensor, coef: torch.Tensor, drop, p: float,
               dy_bound: torch.Tensor, out_dgamma=None, out_dbeta=None, want_rows: bool = True, sync=None,
               act_bits: Optional[torch.Tensor] = None, want_t: bool = True, presummed=None):
    """Backward of a = dropout(relu(bn(y))): returns (dy S16 rows [None unless want_rows: only dgrad reads them],
    dy S16 transposed, dgamma, dbeta); dy_bound (zeroed) receives the guaranteed bound of dy.  With the forward's
    act_bits the two passes read the mask / ReLU predicate (1 bit per element) instead of regenerating them.
    presummed = (dgamma, dbeta): the dgrad launch that wrote go already reduced them and filled dy_bound (vp3d_s16_red):
    only the apply pass runs."""
    ops._chk(go, "go")
    ops._chk(y, "y")
    b, t, c = y.shape
    assert go.shape == y.shape
    m = b * t
    L = _lib.lib()
    dref = C.byref(drop) if drop is not None else None
    nparts = C.c_int32(0)
    sc, sh, mu, inv = (coef[i].data_ptr() for i in range(4))
    if presummed is not None:
        assert sync is None and act_bits is not None
        dgam, dbet = presummed
    elif out_dgamma is not None and out_dbeta is not None:
        dgam, dbet = out_dgamma, out_dbeta
    else:
        dgb = torch.empty((2, c), dtype=torch.float32, device=y.device)
        dgam, dbet = dgb[0], dgb[1]
    fused_fin = act_bits is not None and sync is None
    if presummed is not None:
        pass
    elif fused_fin:
        # reduction + finalize + bound of dy in ONE launch (last-arriver blocks fold the partial rows): no [C]-sized kernel
        # that waits for a CU slot behind the second stream's weight-gradient GEMM
        assert act_bits.numel() * 8 == m * c and act_bits.dtype == torch.uint8
        ngroups, ntick = C.c_int32(0), C.c_int32(0)
        pp = float(p) if drop is not None else 0.0
        check(L.vp3d_bn_bwd_reduce_fin_s16(ops._stream(), m, c, None, None, None, None, None, pp, None, None, None, None, None,
                                           None, None, None, C.byref(nparts), C.byref(ngroups), C.byref(ntick)),
              "vp3d_bn_bwd_reduce_fin_s16(query)")
        parts = torch.empty((nparts.value, 2, c), dtype=torch.float32, device=y.device)
        gparts = torch.empty((ngroups.value, 2, c), dtype=torch.float64, device=y.device)
        check(L.vp3d_bn_bwd_reduce_fin_s16(ops._stream(), m, c, go.data_ptr(), y.data_ptr(), mu, inv, act_bits.data_ptr(), pp, sc,
                                           go_bound.data_ptr(), parts.data_ptr(), gparts.data_ptr(),
                                           _tickets(y.device, ntick.value).data_ptr(), dgam.data_ptr(), dbet.data_ptr(),
                                           dy_bound.data_ptr(), C.byref(nparts), C.byref(ngroups), C.byref(ntick)),
              "vp3d_bn_bwd_reduce_fin_s16")
    elif act_bits is not None:
        assert act_bits.numel() * 8 == m * c and act_bits.dtype == torch.uint8
        keep_scale = 1.0 / (1.0 - p) if drop is not None else 1.0
        check(L.vp3d_bn_bwd_reduce_bits(ops._stream(), m, c, None, None, None, None, None, 1.0, None, C.byref(nparts)),
              "vp3d_bn_bwd_reduce_bits(query)")
        parts = torch.empty((nparts.value, 2, c), dtype=torch.float32, device=y.device)
        check(L.vp3d_bn_bwd_reduce_bits(ops._stream(), m, c, go.data_ptr(), y.data_ptr(), mu, inv, act_bits.data_ptr(),
                                        keep_scale, parts.data_ptr(), C.byref(nparts)), "vp3d_bn_bwd_reduce_bits")
    else:
        check(L.vp3d_bn_bwd_reduce(ops._stream(), m, c, None, None, None, None, None, None, None, None, C.byref(nparts)),
              "vp3d_bn_bwd_reduce(query)")
        parts = torch.empty((nparts.value, 2, c), dtype=torch.float32, device=y.device)
        check(L.vp3d_bn_bwd_reduce(ops._stream(), m, c, go.data_ptr(), y.data_ptr(), sc, sh, mu, inv, dref, parts.data_ptr(),
                                   C.byref(nparts)), "vp3d_bn_bwd_reduce")
    if not fused_fin and presummed is None:
        check(L.vp3d_bn_bwd_finalize_s16(ops._stream(), c, m, parts.data_ptr(), nparts.value, dgam.data_ptr(), dbet.data_ptr(), sc,
                                         go_bound.data_ptr(), float(p), dy_bound.data_ptr()), "vp3d_bn_bwd_finalize_s16")
    a_g, a_b = dgam, dbet
    if sync is not None:                 # dp.SyncBatchNorm: global sums in the apply kernel and in the bound of dy
        a_g, a_b = ops._sync_sums(dgam, dbet, sync)
        dy_bound = new_bound(y.device)
        check(L.vp3d_dy_bound(ops._stream(), c, sync.rows_total(m), sc, (a_g / sync.frac).contiguous().data_ptr(),
                              (a_b / sync.frac).contiguous().data_ptr(), go_bound.data_ptr(), float(p), dy_bound.data_ptr()),
              "vp3d_dy_bound")
    assert want_rows or want_t
    dy = torch.empty_like(y) if want_rows else None
    dyt = torch.empty((c, t_pitch(m)), dtype=torch.float32, device=y.device) if want_t else None   # (wgrad_rows needs none)
    nb = y.numel() * (8.0 + (4.0 if dy is not None else 0.0) + (4.0 if dyt is not None else 0.0) +
                      (0.125 if act_bits is not None else 0.0))
    with ops._Timed("stream_bn_bwd_apply", 0.0, nb, (m, c)):
        check(L.vp3d_bn_bwd_apply_s16(ops._stream(), m, c, go.data_ptr(), y.data_ptr(), sc, sh, mu, inv, dref, ops._p(act_bits),
                                      a_g.data_ptr(), a_b.data_ptr(), dy_bound.data_ptr(), ops._p(dy), ops._p(dyt),
                                      dyt.shape[1] if dyt is not None else 0),
              "vp3d_bn_bwd_apply_s16")
    return (S16(dy, dy_bound) if dy is not None else None), (S16(dyt, dy_bound) if dyt is not None else None), dgam, dbet


def join(x: S16) -> torch.Tensor:
    """S16 -> fp32 (small tensors only: done with torch ops on the raw halves)."""
    d = x.data
    h = d.view(torch.float16).view(*d.shape[:-1], d.shape[-1] // 8, 2, 8).to(torch.float32)
    v = (h[..., 0, :] + h[..., 1, :]).reshape(d.shape)
    if x.bound is None:
        return v
    bd = x.bound.max()
    e = torch.frexp(bd)[1].to(torch.float32) - 15.0
    e = torch.where((bd > 0) & (bd < 3.0e38), e, torch.zeros_like(e))
    return v * torch.exp2(e)


# --------------------------------------------------------------------------------------------------------
# one-launch prologue of a training step (all layers at once)
# --------------------------------------------------------------------------------------------------------
def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def amax_multi(tensors, bounds: torch.Tensor):
    """bounds[i] (rows of a new_bounds() array) = max|tensors[i]|, one launch."""
    n = (C.c_int64 * len(tensors))(*[t.numel() for t in tensors])
    for t in tensors:
        ops._chk(t, "t")
    check(_lib.lib().vp3d_amax_multi(ops._stream(), len(tensors), _ptr_array(tensors), n, bounds.data_ptr()),
          "vp3d_amax_multi")


def pack_weights_multi(weights, bounds: torch.Tensor, want_dgrad=True, launch_ctx=None):
    """[(S16 fwd pack, S16 dgrad pack or None)] for same-shaped-channel conv weights, one launch; layer i uses bounds[i].
    launch_ctx: a callable returning a context manager the LAUNCH runs under (the outputs are allocated before it is entered,
    i.e. on the caller's stream: engine_s16.forward_train hands the launch to the second stream beside the expand layer)."""
    c_out, c_in = weights[0].shape[0], weights[0].shape[1]
    dev = weights[0].device
    wfs, wds, taps = [], [], []
    for w in weights:
        ops._chk(w, "weight")
        assert w.shape[0] == c_out and w.shape[1] == c_in
        k = w.shape[2]
        taps.append(k)
        wfs.append(torch.empty((c_out, k * c_in), dtype=torch.float32, device=dev))
        wds.append(torch.empty((k * c_in, c_out), dtype=torch.float32, device=dev) if want_dgrad else None)
    def launch():
        check(_lib.lib().vp3d_pack_weight_s16_multi(ops._stream(), len(weights), _ptr_array(weights),
                                                    (C.c_int32 * len(taps))(*taps), c_out, c_in, bounds.data_ptr(),
                                                    _ptr_array(wfs), _ptr_array(wds)), "vp3d_pack_weight_s16_multi")
    if launch_ctx is None:
        launch()
    else:
        with launch_ctx():
            launch()
    return [(S16(wf, bounds[i]), S16(wd, bounds[i]) if wd is not None else None)
            for i, (wf, wd) in enumerate(zip(wfs, wds))]


def act_bounds_multi(bns, m_rows, res_from, p: float, bounds: torch.Tensor):
    """bounds[i] = guaranteed bound of layer i's activation (vp3d_act_bound for every layer, one launch)."""
    n = len(bns)
    check(_lib.lib().vp3d_act_bounds_multi(ops._stream(), n, bns[0].num_features, _ptr_array([b.weight for b in bns]),
                                           _ptr_array([b.bias for b in bns]), (C.c_int64 * n)(*m_rows),
                                           (C.c_int32 * n)(*res_from), float(p), bounds.data_ptr()),
          "vp3d_act_bounds_multi")


def prologue_a(tensors, tensor_bounds, floors, bns, m_rows, res_from, p: float, act_bounds: torch.Tensor):
    """Launch A of the fused prologue (vp3d_prologue_a_s16): tensor_bounds[i] = max(floors[i], max|tensors[i]|) for every
    tensor and act_bounds[l] = guaranteed bound of layer l's activation -- amax / amax_multi / act_bounds_multi in one launch."""
    n, nl = len(tensors), len(bns)
    for t in tensors:
        ops._chk(t, "t")
    check(_lib.lib().vp3d_prologue_a_s16(ops._stream(), n, _ptr_array(tensors), (C.c_int64 * n)(*[t.numel() for t in tensors]),
                                         _ptr_array(tensor_bounds), (C.c_float * n)(*floors), nl, bns[0].num_features,
                                         _ptr_array([b.weight for b in bns]), _ptr_array([b.bias for b in bns]),
                                         (C.c_int64 * nl)(*m_rows), (C.c_int32 * nl)(*res_from), float(p), act_bounds.data_ptr()),
          "vp3d_prologue_a_s16")


def prologue_b(x: torch.Tensor, spec: ConvSpec, kpad: int, one_col: int, x_bound: torch.Tensor, want_t: bool, w0: torch.Tensor,
               w0_bound: torch.Tensor, weights, w_bounds: torch.Tensor, want_dgrad: bool):
    """Launch B of the fused prologue (vp3d_prologue_b_s16): im2row_split(x) + pack_weight / split of the expand conv's weight
    + pack_weights_multi(weights) in one launch.  Returns (x_rows S16, x_t S16 or None, w0_packed fp32 [C][kpad], w0 S16,
    [(S16 fwd pack, S16 dgrad pack or None)] for `weights`)."""
    ops._chk(x, "x")
    assert spec.dil == 1 and kpad % 64 == 0
    b, t_in, c_in0 = x.shape
    t_out = spec.t_out(t_in)
    m = b * t_out
    dev = x.device
    rows = torch.empty((b, t_out, kpad), dtype=torch.float32, device=dev)
    tt = torch.empty((kpad, t_pitch(m)), dtype=torch.float32, device=dev) if want_t else None
    c0 = w0.shape[0]
    w0_packed = torch.empty((c0, kpad), dtype=torch.float32, device=dev)
    w0_s16 = torch.empty((c0, kpad), dtype=torch.float32, device=dev)
    wfs, wds, taps = [], [], []
    c_out = c_in = 0
    if weights:
        c_out, c_in = weights[0].shape[0], weights[0].shape[1]
        for w in weights:
            assert w.shape[0] == c_out and w.shape[1] == c_in
            k = w.shape[2]
            taps.append(k)
            wfs.append(torch.empty((c_out, k * c_in), dtype=torch.float32, device=dev))
            wds.append(torch.empty((k * c_in, c_out), dtype=torch.float32, device=dev) if want_dgrad else None)
    rm = RowMap(b, t_out, t_in, spec.stride, 0, 0, 1)
    nl = len(weights)
    check(_lib.lib().vp3d_prologue_b_s16(ops._stream(), C.byref(rm), x.data_ptr(), c_in0, spec.taps * c_in0, kpad, one_col,
                                         x_bound.data_ptr(), rows.data_ptr(), ops._p(tt), tt.shape[1] if tt is not None else 0,
                                         w0.data_ptr(), c0, w0.shape[1], w0.shape[2], w0_bound.data_ptr(), w0_packed.data_ptr(),
                                         w0_s16.data_ptr(), nl, _ptr_array(weights), (C.c_int32 * max(nl, 1))(*(taps or [1])),
                                         c_out, c_in, w_bounds.data_ptr() if nl else None, _ptr_array(wfs), _ptr_array(wds)),
          "vp3d_prologue_b_s16")
    packs = [(S16(wf, w_bounds[i]), S16(wd, w_bounds[i]) if wd is not None else None) for i, (wf, wd) in enumerate(zip(wfs, wds))]
    return S16(rows, x_bound), (S16(tt, x_bound) if tt is not None else None), w0_packed, S16(w0_s16, w0_bound), packs
