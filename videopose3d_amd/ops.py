"""Tensor-level wrappers over the C ABI (include/vp3d.h).  PyTorch is used only for device memory and the
current HIP stream; every FLOP of the path is executed by libvp3d.so.

All activation tensors are contiguous fp32 ``[B, T, C]`` (channels-last rows).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ActBwd, Dropout, Epilogue, RowMap, check
from .plan import ConvSpec, ResSpec

_zero_pages = {}

# Optional per-launch instrumentation used by bench.py (roofline): when a list is installed, every GEMM entry
# point is bracketed with HIP events on the launch stream and (family, algorithmic FLOPs, start, end) is appended.
_prof = None


def set_profiler(records):
    global _prof
    _prof = records


class _Timed:
    def __init__(self, family, flops, nbytes=0.0, shape=None):
        self.family, self.flops, self.nbytes, self.shape = family, flops, nbytes, shape

    def __enter__(self):
        if _prof is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _prof is not None:
            self.e1.record()
            _prof.append((self.family, self.flops, self.e0, self.e1, self.nbytes, self.shape))
        return False


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Raw handle of torch's current HIP stream on the current device.  Through torch's C entry points when this build has
    them: torch.cuda.current_stream() builds a Python Stream object per call (3-9 us, ~20 calls per step: 8 % of the host time
    of a launch-bound step, tools/host_profile.py)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _lib.Vp3dError("%s: the vp3d HIP path needs a CUDA/HIP tensor, got device %s (there is no CPU fallback)"
                             % (name, t.device))
    if t.dtype != torch.float32:
        raise _lib.Vp3dError("%s: expected float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise _lib.Vp3dError("%s: expected a contiguous tensor" % name)


def zeros_page(device) -> torch.Tensor:
    """A 4 KiB page of device zeros: the source for out-of-range taps / ragged tile rows."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    z = _zero_pages.get(key)
    if z is None:
        z = torch.zeros(1024, dtype=torch.float32, device=torch.device("cuda", key))
        _zero_pages[key] = z
    return z


def make_dropout(p: float, seed: int, offset: int, layer: int, offset_ptr: Optional[int] = None) -> Optional[Dropout]:
    """offset_ptr: device address of a uint64 step counter that the kernels add to `offset` (graph replays)."""
    if p <= 0.0:
        return None
    return Dropout(float(p), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_uint64(offset & 0xFFFFFFFFFFFFFFFF), int(layer),
                   offset_ptr)


def _epi(bias=None, relu=False, residual=None, stats=None, n_cols=0) -> Optional[Epilogue]:
    """residual = (tensor[B,T_r,C_r], r_stride, r_off, r_col0)."""
    if bias is None and not relu and residual is None and stats is None:
        return None
    e = Epilogue()
    e.bias = _p(bias)
    e.relu = 1 if relu else 0
    if residual is not None:
        r, r_stride, r_off, r_col0 = residual
        _chk(r, "residual")
        e.residual = r.data_ptr()
        e.r_bpitch = r.shape[1] * r.shape[2]
        e.r_ld = r.shape[2]
        e.r_t = r.shape[1]
        e.r_stride = r_stride
        e.r_off = r_off
        e.r_col0 = r_col0
        e.r_cols = r.shape[2]
    if stats is not None:
        e.stat_sum = stats[0].data_ptr()
        e.stat_m2 = stats[1].data_ptr()
    return e


def stat_buffers(m_rows: int, c: int, device, slab_rows: int = 64) -> Tuple[torch.Tensor, torch.Tensor]:
    """(sum, M2) partial rows of the GEMM epilogue's BatchNorm statistics: one row per slab of slab_rows rows (64; 32 for the
    split-fp16 GEMM's 224-row tiling, ops_s16.stat_slab_rows)."""
    slabs = (m_rows + slab_rows - 1) // slab_rows
    buf = torch.empty((2, slabs, c), dtype=torch.float32, device=device)
    return buf[0], buf[1]


# --------------------------------------------------------------------------------------------------------
# convolutions
# --------------------------------------------------------------------------------------------------------
launch_log = None        # tools/step_table.py: a list that receives (family, flops, nbytes, shape) of every GEMM call, in issue order


def _timed_call(family, flops, fn, *args, nbytes=0.0, shape=None):
    """nbytes: algorithmic HBM bytes of the launch (every operand read once, the result written once); shape: what the
    per-launch evidence tables print -- (M, N, K, tile configuration, K-slices, GEMM kernels launched)."""
    if launch_log is not None:
        launch_log.append((family, flops, nbytes, shape))
    with _Timed(family, flops, nbytes, shape):
        check(fn(*args), fn.__name__)


def pack_weight(w: torch.Tensor, scale: Optional[torch.Tensor] = None, ld_out: Optional[int] = None) -> torch.Tensor:
    """Reference Conv1d.weight [C_out, C_in, taps] -> packed [C_out, ld_out >= taps*C_in] rows
    (wt[co][k*C_in + ci] = w[co][ci][k], optionally row-scaled; padding columns are zero)."""
    _chk(w, "weight")
    c_out, c_in, taps = w.shape
    ld = taps * c_in if ld_out is None else int(ld_out)
    if taps == 1 and scale is None and ld == c_in:
        return w.view(c_out, c_in)
    out = torch.empty((c_out, ld), dtype=torch.float32, device=w.device)
    check(_lib.lib().vp3d_pack_weight(_stream(), w.data_ptr(), c_out, c_in, taps, _p(scale), out.data_ptr(), ld),
          "vp3d_pack_weight")
    return out


def padded_k(spec: ConvSpec) -> int:
    """Row width the fast (LDS-DMA) GEMM path needs for this conv: taps*C_in rounded up to 32 when the taps are
    adjacent rows (dil == 1) and C_in itself is not DMA-friendly (expand_conv: 3*34 = 102 -> 128); else 0."""
    k = spec.taps * spec.c_in
    if spec.dil == 1 and (spec.c_in % 32 != 0) and k >= 32:
        return (k + 31) // 32 * 32
    return 0


def im2row(x: torch.Tensor, spec: ConvSpec, kpad: int, one_col: int = -1) -> torch.Tensor:
    """[B,T_in,C_in] -> [B,T_out,kpad]: row (b,t) = the taps*C_in contiguous floats at x[b, t*stride], zero padded
    (one_col >= taps*C_in: that padding column holds 1)."""
    _chk(x, "x")
    assert spec.dil == 1
    b, t_in, c_in = x.shape
    t_out = spec.t_out(t_in)
    out = torch.empty((b, t_out, kpad), dtype=torch.float32, device=x.device)
    rm = RowMap(b, t_out, t_in, spec.stride, 0, 0, 1)
    check(_lib.lib().vp3d_im2row(_stream(), C.byref(rm), x.data_ptr(), c_in, spec.taps * c_in, kpad, one_col,
                                 out.data_ptr()), "vp3d_im2row")
    return out


def _splitk_ws(m, n, k, device):
    n_ws = _lib.lib().vp3d_rows_gemm_ws_floats(m, n, k)
    if n_ws <= 0:
        return None, 0
    ws = torch.empty((n_ws,), dtype=torch.float32, device=device)
    return ws, n_ws


def conv_fwd(x: torch.Tensor, wt: torch.Tensor, spec: ConvSpec, *, bias=None, relu=False,
             residual: Optional[Tuple[torch.Tensor, ResSpec]] = None, stats=None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = conv(x) with the fused epilogue.  x [B,T_in,C_in], wt packed [C_out, taps*C_in]."""
    _chk(x, "x")
    _chk(wt, "wt")
    b, t_in, c_in = x.shape
    assert c_in == spec.c_in and wt.shape == (spec.c_out, spec.taps * spec.c_in), (x.shape, wt.shape, spec)
    t_out = spec.t_out(t_in)
    if out is None:
        out = torch.empty((b, t_out, spec.c_out), dtype=torch.float32, device=x.device)
    if spec.dil == 1:
        # taps are contiguous rows: one "tap" of taps*C_in channels (pure reshape GEMM for the strided convs)
        rm = RowMap(b, t_out, t_in, spec.stride, 0, 0, 1)
        c_src = spec.taps * c_in
    else:
        rm = RowMap(b, t_out, t_in, spec.stride, spec.dil, 0, spec.taps)
        c_src = c_in
    res = None
    if residual is not None:
        r, rs = residual
        assert r.shape[0] == b and r.shape[2] == spec.c_out
        assert rs.start + rs.step * (t_out - 1) < r.shape[1], "residual slice out of range"
        res = (r, rs.step, rs.start, 0)
    e = _epi(bias, relu, res, stats, spec.c_out)
    m, k = b * t_out, spec.taps * c_in
    ws, ws_n = _splitk_ws(m, spec.c_out, k, x.device)
    _timed_call("tconv_fwd", 2.0 * m * spec.c_out * k, _lib.lib().vp3d_tconv_fwd,
                _stream(), C.byref(rm), x.data_ptr(), c_in, c_src, wt.data_ptr(), wt.shape[1], spec.c_out,
                out.data_ptr(), t_out * spec.c_out, spec.c_out, C.byref(e) if e is not None else None,
                zeros_page(x.device).data_ptr(), _p(ws), ws_n,
                nbytes=4.0 * (x.numel() + wt.numel() + out.numel() + (out.numel() if residual is not None else 0)))
    return out


def can_fuse_act_bwd(spec: ConvSpec, t_in: int) -> bool:
    """The fused activation-backward epilogue of vp3d_tconv_dgrad needs full 128-column tiles of 16-B aligned rows
    and, for the strided (plain-GEMM) form, windows that tile the input exactly (every dx row is written)."""
    if spec.c_in % 128 != 0:
        return False
    if spec.stride == spec.taps and spec.dil == 1 and spec.taps > 1:
        return spec.taps * spec.t_out(t_in) == t_in
    return spec.stride == 1


def should_fuse_act_bwd(spec: ConvSpec, batch: int, t_in: int) -> bool:
    """Measured on MI355X (tools/fuse_bench.py): the fused epilogue is ~8 us per tile instead of ~2.5 us, which is
    hidden while the co-resident workgroup is in its main loop -- but the two workgroups of a CU start together and
    run in lockstep for the first rounds, so on launches of <= 3.4 rounds of tiles (1,728 tiles: +35 us) the fusion
    costs what the separate reduction pass takes (41 us), while on the 5,184-tile dgrad that feeds the expand layer's
    activation (85 M elements) it is free and removes a 121 us pass.  Fuse from 6 rounds of 512 tiles up."""
    if not can_fuse_act_bwd(spec, t_in):
        return False
    strided = spec.stride == spec.taps and spec.dil == 1 and spec.taps > 1
    m = batch * (spec.t_out(t_in) if strided else t_in)
    n = spec.taps * spec.c_in if strided else spec.c_in
    return ((m + 127) // 128) * ((n + 127) // 128) >= 3072


class FusedActBwd:
    """Result of a dgrad whose epilogue also ran the upstream activation's backward reduction (vp3d_act_bwd)."""
    __slots__ = ("g", "parts")

    def __init__(self, g, parts):
        self.g, self.parts = g, parts


def _act_bwd_struct(act_bwd, like_shape, n_cols, m_rows, device, store_v, keep):
    """act_bwd = (y_up [B,T_in,C], coef [4,C], drop).  Returns (ActBwd struct, FusedActBwd)."""
    y_up, coef, drop = act_bwd
    _chk(y_up, "y_up")
    assert tuple(y_up.shape) == tuple(like_shape), (y_up.shape, like_shape)
    c = y_up.shape[2]
    g = torch.empty_like(y_up)
    nparts = _lib.lib().vp3d_act_bwd_parts(m_rows, n_cols, c)
    parts = torch.empty((nparts, 2, c), dtype=torch.float32, device=device)
    ab = ActBwd()
    ab.y_up = y_up.data_ptr()
    ab.scale, ab.shift, ab.mean, ab.invstd = (coef[i].data_ptr() for i in range(4))
    ab.drop = C.pointer(drop) if drop is not None else None
    ab.g_out = g.data_ptr()
    ab.partials = parts.data_ptr()
    ab.c_stat = c
    ab.store_v = 1 if store_v else 0
    keep.append(ab)
    return ab, FusedActBwd(g, parts)


def conv_dgrad(dy: torch.Tensor, wt: torch.Tensor, spec: ConvSpec, t_in: int, *,
               residual: Optional[Tuple[torch.Tensor, ResSpec]] = None, act_bwd=None, store_v: bool = True):
    """dx [B,T_in,C_in] = conv^T(dy) (+ scatter of the block's residual gradient).  wt is the forward pack.

    act_bwd = (y_up, coef, drop) of the layer whose activation produced this conv's input: the epilogue then also
    forms g = dx*keep*[bn(y_up) > 0] and the per-slab sums of g and g*xhat (what vp3d_bn_bwd_reduce would do in a
    separate pass) and the call returns (dx or None, FusedActBwd).  store_v=False: dx itself is not needed."""
    _chk(dy, "dy")
    _chk(wt, "wt")
    b, t_out, c_out = dy.shape
    assert c_out == spec.c_out and t_out == spec.t_out(t_in)
    c_in, taps = spec.c_in, spec.taps
    ldw = wt.shape[1]
    z = zeros_page(dy.device).data_ptr()
    flops = 2.0 * b * t_out * c_out * taps * c_in
    keep = []
    fused = None
    want_dx = store_v or act_bwd is None
    if spec.stride == taps and spec.dil == 1 and taps > 1:
        # windows do not overlap: dx viewed as [B*T_out, taps*C_in] = dy @ Wt   (plain GEMM)
        covered = taps * t_out
        if act_bwd is not None and covered != t_in:
            raise _lib.Vp3dError("conv_dgrad: the fused activation backward needs windows that tile the input exactly")
        dx = None
        if want_dx:
            dx = (torch.empty if covered == t_in else torch.zeros)((b, t_in, c_in), dtype=torch.float32,
                                                                    device=dy.device)
        rm = RowMap(b, t_out, t_out, 1, 0, 0, 1)
        res = None
        if residual is not None:
            r, rs = residual                      # dx[b, start + step*t] += r[b, t]   with step == taps
            assert rs.step == taps and 0 <= rs.start < taps and r.shape == (b, t_out, c_in)
            res = (r, 1, 0, rs.start * c_in)
        e = _epi(residual=res, n_cols=taps * c_in)
        if act_bwd is not None:
            e = e if e is not None else Epilogue()
            ab, fused = _act_bwd_struct(act_bwd, (b, t_in, c_in), taps * c_in, b * t_out, dy.device, want_dx, keep)
            e.act_bwd = C.pointer(ab)
        ws, ws_n = _splitk_ws(b * t_out, taps * c_in, c_out, dy.device)
        _timed_call("tconv_dgrad", flops, _lib.lib().vp3d_tconv_dgrad,
                    _stream(), C.byref(rm), dy.data_ptr(), c_out, c_out, wt.data_ptr(), ldw, 0, taps * c_in,
                    _p(dx), t_in * c_in, taps * c_in, C.byref(e) if e is not None else None, z, _p(ws), ws_n,
                    nbytes=4.0 * (dy.numel() + wt.numel() + b * t_in * c_in + (residual[0].numel() if residual else 0)))
        return dx if act_bwd is None else (dx, fused)
    if spec.stride != 1:
        raise _lib.Vp3dError("conv_dgrad: stride %d with %d taps is not a configuration of the temporal model"
                             % (spec.stride, taps))
    # gather form: dx[b,s] = sum_k dy[b, s - k*dil] @ W_k^T
    dx = torch.empty((b, t_in, c_in), dtype=torch.float32, device=dy.device) if want_dx else None
    rm = RowMap(b, t_in, t_out, 1, -spec.dil, 0, taps)
    res = None
    if residual is not None:
        r, rs = residual                          # dx[b, s] += r[b, s - start]
        assert rs.step == 1 and r.shape[0] == b and r.shape[2] == c_in
        res = (r, 1, -rs.start, 0)
    e = _epi(residual=res, n_cols=c_in)
    if act_bwd is not None:
        e = e if e is not None else Epilogue()
        ab, fused = _act_bwd_struct(act_bwd, (b, t_in, c_in), c_in, b * t_in, dy.device, want_dx, keep)
        e.act_bwd = C.pointer(ab)
    ws, ws_n = _splitk_ws(b * t_in, c_in, taps * c_out, dy.device)
    _timed_call("tconv_dgrad", flops, _lib.lib().vp3d_tconv_dgrad,
                _stream(), C.byref(rm), dy.data_ptr(), c_out, c_out, wt.data_ptr(), ldw, c_in, c_in, _p(dx),
                t_in * c_in, c_in, C.byref(e) if e is not None else None, z, _p(ws), ws_n,
                nbytes=4.0 * (dy.numel() + wt.numel() + b * t_in * c_in + (residual[0].numel() if residual else 0)))
    return dx if act_bwd is None else (dx, fused)


def conv_wgrad(dy: torch.Tensor, x: torch.Tensor, spec: ConvSpec, *, rows_kpad: int = 0,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW in the reference layout [C_out, C_in, taps].

    rows_kpad > 0: `x` is the im2row staging of the conv input ([B,T_out,rows_kpad], see ``im2row``) and the
    reduction runs as a 1-tap GEMM over those rows."""
    _chk(dy, "dy")
    _chk(x, "x")
    b, t_out, c_out = dy.shape
    c_in, taps = spec.c_in, spec.taps
    assert c_out == spec.c_out
    if rows_kpad:
        assert x.shape == (b, t_out, rows_kpad)
        rm = RowMap(b, t_out, t_out, 1, 0, 0, 1)
        ldx, c_x, n_cols = rows_kpad, rows_kpad, rows_kpad
    else:
        _, t_in, c_in_x = x.shape
        assert c_in_x == c_in and t_out == spec.t_out(t_in)
        if spec.dil == 1:
            rm = RowMap(b, t_out, t_in, spec.stride, 0, 0, 1)
            c_x = taps * c_in
        else:
            rm = RowMap(b, t_out, t_in, spec.stride, spec.dil, 0, taps)
            c_x = c_in
        ldx, n_cols = c_in, taps * c_in
    m_rows = b * t_out
    splits = _lib.lib().vp3d_wgrad_splits(m_rows, c_out, n_cols)
    if out is not None:
        assert out.shape == (c_out, c_in, taps) and out.is_contiguous() and out.dtype == torch.float32
        dw = out
    else:
        dw = torch.empty((c_out, c_in, taps), dtype=torch.float32, device=dy.device)
    direct = splits == 1 and taps == 1 and not rows_kpad
    part = dw if direct else torch.empty((splits, c_out, n_cols), dtype=torch.float32, device=dy.device)
    _timed_call("tconv_wgrad", 2.0 * m_rows * c_out * taps * c_in, _lib.lib().vp3d_tconv_wgrad,
                _stream(), C.byref(rm), dy.data_ptr(), c_out, c_out, x.data_ptr(), ldx, c_x, part.data_ptr(), splits,
                zeros_page(dy.device).data_ptr(), nbytes=4.0 * (dy.numel() + x.numel() + dw.numel()))
    if not direct:
        check(_lib.lib().vp3d_wgrad_reduce(_stream(), part.data_ptr(), n_cols, splits, c_out, c_in, taps,
                                           dw.data_ptr()), "vp3d_wgrad_reduce")
    return dw


def colsum(g2d: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(g2d, "g")
    m, n = g2d.shape
    if out is None:
        out = torch.empty((n,), dtype=torch.float32, device=g2d.device)
    assert out.shape == (n,) and out.is_contiguous() and out.dtype == torch.float32
    check(_lib.lib().vp3d_colsum(_stream(), m, n, g2d.data_ptr(), n, out.data_ptr()), "vp3d_colsum")
    return out


# --------------------------------------------------------------------------------------------------------
# the head: the 3*J_out-column shrink conv at small row counts (csrc/vp3d_head.hip; reference common/model.py:33,137,196)
# --------------------------------------------------------------------------------------------------------
def head_supported(m_rows: int, k: int, n: int) -> bool:
    """The dedicated head kernels serve this shrink conv ([m_rows, k] x [n, k]^T); else the general GEMM entry points do."""
    from ._switches import SW
    return bool(SW["head_kernels"]) and bool(_lib.lib().vp3d_head_supported(int(m_rows), int(k), int(n)))


def head_fwd(h: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """out[B, T, N] = h[B, T, K] @ w[N, K, 1]^T + bias: one launch (vp3d_head_fwd)."""
    _chk(h, "h")
    _chk(w, "weight")
    b, t, k = h.shape
    n = w.shape[0]
    assert h.is_contiguous() and w.is_contiguous() and w.shape[1] == k and w.shape[2] == 1
    out = torch.empty((b, t, n), dtype=torch.float32, device=h.device)
    m = b * t
    _timed_call("tconv_fwd", 2.0 * m * n * k, _lib.lib().vp3d_head_fwd, _stream(), m, k, n, h.data_ptr(), w.data_ptr(), _p(bias),
                out.data_ptr(), nbytes=4.0 * (m * k + n * k + m * n), shape=(m, n, k, "head", 1, 1))
    return out


def head_bwd(gout: torch.Tensor, h: torch.Tensor, w: torch.Tensor, dh_bound: Optional[torch.Tensor] = None, want_dw: bool = True):
    """(dh [B, T, K], ws) of the shrink conv: ONE launch (vp3d_head_bwd) that also leaves max|dh| in dh_bound (32 zeroed slots)
    and the row-sliced partials of dW / dbias in the workspace ws, for head_fold."""
    _chk(gout, "gout")
    _chk(h, "h")
    b, t, k = h.shape
    n = w.shape[0]
    m = b * t
    assert gout.is_contiguous() and h.is_contiguous() and w.is_contiguous() and gout.numel() == m * n
    dh = torch.empty_like(h)
    ws = None
    if want_dw:
        ws = torch.empty((int(_lib.lib().vp3d_head_bwd_ws_floats(m, k, n)),), dtype=torch.float32, device=h.device)
    _timed_call("tconv_dgrad", 2.0 * m * n * k * (2 if want_dw else 1), _lib.lib().vp3d_head_bwd, _stream(), m, k, n,
                gout.data_ptr(), h.data_ptr(), w.data_ptr(), dh.data_ptr(), _p(dh_bound), _p(ws),
                nbytes=4.0 * (2 * m * k + n * k + m * n), shape=(m, n, k, "head", 1, 1))
    return dh, ws


def head_fold(ws: torch.Tensor, m_rows: int, w: torch.Tensor, out_dw: Optional[torch.Tensor] = None,
              out_db: Optional[torch.Tensor] = None):
    """(dW [N, K, 1], dbias [N]) from head_bwd's partials, summed in slice order (vp3d_head_fold)."""
    n, k = w.shape[0], w.shape[1]
    dw = out_dw if out_dw is not None else torch.empty_like(w)
    db = out_db if out_db is not None else torch.empty((n,), dtype=torch.float32, device=w.device)
    assert dw.is_contiguous() and db.is_contiguous() and dw.numel() == n * k and db.numel() == n
    check(_lib.lib().vp3d_head_fold(_stream(), int(m_rows), k, n, ws.data_ptr(), dw.data_ptr(), db.data_ptr()), "vp3d_head_fold")
    return dw, db


# --------------------------------------------------------------------------------------------------------
# batch norm / activation
# --------------------------------------------------------------------------------------------------------
def bn_fold(bn: torch.nn.BatchNorm1d) -> Tuple[torch.Tensor, torch.Tensor]:
    c = bn.num_features
    buf = torch.empty((2, c), dtype=torch.float32, device=bn.weight.device)
    check(_lib.lib().vp3d_bn_fold(_stream(), c, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                  bn.running_var.data_ptr(), float(bn.eps), buf[0].data_ptr(), buf[1].data_ptr()),
          "vp3d_bn_fold")
    return buf[0], buf[1]


def _bn_finalize_sync(bn: torch.nn.BatchNorm1d, m_rows: int, stats, sync) -> torch.Tensor:
    """Synchronised statistics (dp.SyncBatchNorm): local finalize without the running update, fp64 merge of the per-rank
    (mean, var) over the process group, then scale / shift / running buffers from the GLOBAL statistics."""
    c = bn.num_features
    buf = torch.empty((4, c), dtype=torch.float32, device=bn.weight.device)
    check(_lib.lib().vp3d_bn_finalize(
        _stream(), c, m_rows, stats[0].data_ptr(), stats[1].data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
        float(bn.eps), 0.0, None, None, None, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), buf[3].data_ptr()),
        "vp3d_bn_finalize")
    eps = float(bn.eps)
    var_l = (1.0 / buf[3].double() ** 2 - eps).clamp_min(0.0)
    mean_g, var_g, n_tot = sync.merge_stats(buf[2], var_l, m_rows)
    invstd = torch.rsqrt(var_g + eps)
    scale = bn.weight.detach().double() * invstd
    buf[0] = scale.float()
    buf[1] = (bn.bias.detach().double() - mean_g * scale).float()
    buf[2] = mean_g.float()
    buf[3] = invstd.float()
    if bn.track_running_stats and bn.running_mean is not None:
        mom = float(bn.momentum) if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked.item()) + 1)
        with torch.no_grad():
            bn.running_mean.mul_(1.0 - mom).add_((mom * mean_g).float())
            bn.running_var.mul_(1.0 - mom).add_((mom * var_g * (n_tot / (n_tot - 1.0).clamp_min(1.0))).float())
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
    return buf


def bn_finalize(bn: torch.nn.BatchNorm1d, m_rows: int, stats, sync=None, momentum_dev: Optional[int] = None,
                slab_rows: int = 64) -> torch.Tensor:
    """Returns a [4, C] tensor: scale, shift, mean, invstd.  Updates the running buffers in place.
    sync: a dp.SyncBatchNorm -> statistics over the global batch.  momentum_dev: device address of a float the kernel reads
    the momentum from at execution time (hipGraph replays follow set_bn_momentum), instead of bn.momentum as an argument.
    slab_rows: rows per statistics slab of `stats` (stat_buffers)."""
    if sync is not None:
        assert slab_rows == 64, "synchronised BatchNorm merges 64-row slabs"
        return _bn_finalize_sync(bn, m_rows, stats, sync)
    c = bn.num_features
    if m_rows <= 1:
        # same failure as torch.nn.functional.batch_norm in training mode
        raise ValueError("Expected more than 1 value per channel when training, got input size [%d, %d]" % (m_rows, c))
    buf = torch.empty((4, c), dtype=torch.float32, device=bn.weight.device)
    track = bn.track_running_stats and bn.running_mean is not None
    momentum = 0.0 if bn.momentum is None else float(bn.momentum)
    if bn.momentum is None and track:
        momentum = 1.0 / float(int(bn.num_batches_tracked.item()) + 1)   # cumulative average (not used by run.py)
    tail = (bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
            bn.num_batches_tracked.data_ptr() if (track and bn.num_batches_tracked is not None) else None,
            buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), buf[3].data_ptr())
    head = (_stream(), c, m_rows, stats[0].data_ptr(), stats[1].data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps))
    if slab_rows != 64:
        use_dev = momentum_dev is not None and track and bn.momentum is not None
        check(_lib.lib().vp3d_bn_finalize_slab(head[0], c, m_rows, slab_rows, *head[3:], 0.0 if use_dev else momentum,
                                               momentum_dev if use_dev else None, *tail), "vp3d_bn_finalize_slab")
        return buf
    if momentum_dev is not None and track and bn.momentum is not None:
        check(_lib.lib().vp3d_bn_finalize_dm(*head, momentum_dev, *tail), "vp3d_bn_finalize_dm")
    else:
        check(_lib.lib().vp3d_bn_finalize(*head, momentum, *tail), "vp3d_bn_finalize")
    return buf


def bn_act_fwd(y: torch.Tensor, coef: torch.Tensor, drop: Optional[Dropout],
               residual: Optional[Tuple[torch.Tensor, ResSpec]] = None) -> torch.Tensor:
    _chk(y, "y")
    b, t, c = y.shape
    out = torch.empty_like(y)
    if residual is not None:
        r, rs = residual
        _chk(r, "residual")
        assert r.shape[0] == b and r.shape[2] == c and rs.start + rs.step * (t - 1) < r.shape[1]
        args = (r.data_ptr(), t, r.shape[1], rs.step, rs.start, c)
    else:
        args = (None, t, 0, 0, 0, c)
    check(_lib.lib().vp3d_bn_act_fwd(_stream(), b * t, c, y.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                     C.byref(drop) if drop is not None else None, *args, out.data_ptr()),
          "vp3d_bn_act_fwd")
    return out


def _sync_sums(dgam, dbet, sync):
    """Global (sum g*xhat, sum g) scaled by local/global rows, so that the apply kernels' 1/M_local becomes 1/M_global."""
    g = torch.stack([dgam, dbet])
    sync.sum_(g)
    g.mul_(sync.frac)
    return g[0], g[1]


def bn_eval_coef(bn: torch.nn.BatchNorm1d) -> torch.Tensor:
    """[4, C] = scale, shift, mean, invstd of an eval-mode BatchNorm (running statistics): the coefficient block of
    bn_finalize, for the differentiable eval forward (torch.nn.functional.batch_norm(training=False))."""
    with torch.no_grad():
        invstd = torch.rsqrt(bn.running_var.float() + float(bn.eps))
        scale = bn.weight.detach().float() * invstd
        shift = bn.bias.detach().float() - bn.running_mean.float() * scale
        return torch.stack([scale, shift, bn.running_mean.float(), invstd]).contiguous()


def bn_act_bwd(go: torch.Tensor, y: torch.Tensor, coef: torch.Tensor, drop: Optional[Dropout],
               out_dgamma: Optional[torch.Tensor] = None, out_dbeta: Optional[torch.Tensor] = None, sync=None,
               frozen: bool = False):
    """Returns (dy, dgamma, dbeta) for a = dropout(relu(bn(y))); dgamma / dbeta are written into the given
    contiguous fp32 [C] tensors when provided (gradient sink).  frozen: the BatchNorm ran on its running statistics
    (eval mode): mean and variance do not depend on y, so dy = scale * g without the two batch-statistic terms."""
    _chk(go, "go")
    _chk(y, "y")
    b, t, c = y.shape
    assert go.shape == y.shape
    m = b * t
    L = _lib.lib()
    dref = C.byref(drop) if drop is not None else None
    nparts = C.c_int32(0)
    check(L.vp3d_bn_bwd_reduce(_stream(), m, c, None, None, None, None, None, None, None, None, C.byref(nparts)),
          "vp3d_bn_bwd_reduce(query)")
    parts = torch.empty((nparts.value, 2, c), dtype=torch.float32, device=y.device)
    sc, sh, mu, inv = (coef[i].data_ptr() for i in range(4))
    check(L.vp3d_bn_bwd_reduce(_stream(), m, c, go.data_ptr(), y.data_ptr(), sc, sh, mu, inv, dref, parts.data_ptr(),
                               C.byref(nparts)), "vp3d_bn_bwd_reduce")
    if out_dgamma is not None and out_dbeta is not None:
        for t in (out_dgamma, out_dbeta):
            assert t.shape == (c,) and t.is_contiguous() and t.dtype == torch.float32 and t.device == y.device
        dgam, dbet = out_dgamma, out_dbeta
    else:
        dgb = torch.empty((2, c), dtype=torch.float32, device=y.device)
        dgam, dbet = dgb[0], dgb[1]
    check(L.vp3d_bn_bwd_finalize(_stream(), c, parts.data_ptr(), nparts.value, dgam.data_ptr(), dbet.data_ptr()),
          "vp3d_bn_bwd_finalize")
    a_g, a_b = (dgam, dbet) if sync is None else _sync_sums(dgam, dbet, sync)   # the parameter gradients stay local sums
    if frozen:
        a_g = a_b = torch.zeros(c, dtype=torch.float32, device=y.device)
    dy = torch.empty_like(y)
    check(L.vp3d_bn_bwd_apply(_stream(), m, c, go.data_ptr(), y.data_ptr(), sc, sh, mu, inv, dref, a_g.data_ptr(),
                              a_b.data_ptr(), dy.data_ptr()), "vp3d_bn_bwd_apply")
    return dy, dgam, dbet


def bn_act_bwd_fused(fused: "FusedActBwd", y: torch.Tensor, coef: torch.Tensor,
                     out_dgamma: Optional[torch.Tensor] = None, out_dbeta: Optional[torch.Tensor] = None):
    """Second half of bn_act_bwd when the dgrad epilogue already produced g and the partial sums: finalize
    (dgamma, dbeta) and apply dy = scale*(g - dbeta/M - xhat*dgamma/M).  Returns (dy, dgamma, dbeta)."""
    _chk(y, "y")
    b, t, c = y.shape
    m = b * t
    L = _lib.lib()
    if out_dgamma is not None and out_dbeta is not None:
        dgam, dbet = out_dgamma, out_dbeta
    else:
        dgb = torch.empty((2, c), dtype=torch.float32, device=y.device)
        dgam, dbet = dgb[0], dgb[1]
    check(L.vp3d_bn_bwd_finalize(_stream(), c, fused.parts.data_ptr(), fused.parts.shape[0], dgam.data_ptr(),
                                 dbet.data_ptr()), "vp3d_bn_bwd_finalize")
    dy = torch.empty_like(y)
    check(L.vp3d_bn_bwd_apply_g(_stream(), m, c, fused.g.data_ptr(), y.data_ptr(), coef[0].data_ptr(),
                                coef[2].data_ptr(), coef[3].data_ptr(), dgam.data_ptr(), dbet.data_ptr(),
                                dy.data_ptr()), "vp3d_bn_bwd_apply_g")
    return dy, dgam, dbet


def dropout_mask(n: int, drop: Optional[Dropout], device) -> torch.Tensor:
    out = torch.empty((n,), dtype=torch.float32, device=device)
    check(_lib.lib().vp3d_dropout_mask(_stream(), n, C.byref(drop) if drop is not None else None, out.data_ptr()),
          "vp3d_dropout_mask")
    return out
