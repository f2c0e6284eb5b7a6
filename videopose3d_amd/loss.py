"""HIP implementations of the reference's training losses (reference common/loss.py:11-25).

``mpjpe(predicted, target)`` and ``weighted_mpjpe(predicted, target, w)`` keep the reference signatures and return
a 0-dim tensor attached to autograd; forward and gradient are produced by ONE kernel (vp3d_mpjpe) instead of the
~10 small torch kernels of ``torch.mean(torch.norm(predicted - target, dim=-1))`` + its autograd graph.
The evaluation-only metrics of loss.py (p_mpjpe: numpy SVD; n_mpjpe; mean_velocity_error) are outside the hot path
and stay with the reference.  No CPU fallback: CPU tensors raise Vp3dError.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check
from .ops import _chk, _p, _stream


_ws_pool = {}


def _workspace(device, n_doubles: int) -> torch.Tensor:
    """The fold's workspace (ticket + per-block partials), one per (device, stream): zeroed ONCE -- the kernel leaves its ticket
    zero, launches on one stream are ordered."""
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    t = _ws_pool.get(key)
    if t is None or t.numel() < n_doubles:
        t = _ws_pool[key] = torch.zeros(max(n_doubles, 1100), dtype=torch.float64, device=device)
    return t


def _mpjpe_call(pred, target, w, need_grad):
    _chk(pred, "predicted")
    _chk(target, "target")
    dim = pred.shape[-1]
    n = pred.numel() // dim
    L = _lib.lib()
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if need_grad else None
    ws_bytes = L.vp3d_mpjpe_ws_bytes(n)
    ws = _workspace(pred.device, ws_bytes // 8) if ws_bytes else None
    with torch.cuda.device(pred.device):
        check(L.vp3d_mpjpe(_stream(), n, dim, pred.data_ptr(), target.data_ptr(), _p(w), loss.data_ptr(), _p(grad),
                           _p(ws)), "vp3d_mpjpe")
    return loss, grad


class _MpjpeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, w):
        need = pred.requires_grad or target.requires_grad
        loss, grad = _mpjpe_call(pred, target, w, need)
        ctx.save_for_backward(grad)
        ctx.need = (pred.requires_grad, target.requires_grad)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        g = grad * gout                               # upstream scalar (1 for loss.backward())
        return (g if ctx.need[0] else None), (-g if ctx.need[1] else None), None


def _prep(predicted, target, w=None):
    assert predicted.shape == target.shape
    if not predicted.is_cuda:
        raise _lib.Vp3dError("videopose3d_amd.loss runs on the GPU only (got %s)" % predicted.device)
    p = predicted.to(torch.float32).contiguous()
    t = target.to(torch.float32).contiguous()
    if w is not None:
        assert w.shape[0] == predicted.shape[0]
        if w.requires_grad:
            raise _lib.Vp3dError("weighted_mpjpe: a gradient w.r.t. the weights is not implemented (run.py:359 "
                                 "builds w from the ground-truth trajectory, which carries no gradient)")
        w = torch.broadcast_to(w.to(torch.float32), predicted.shape[:-1]).contiguous()
    return p, t, w


def mpjpe(predicted, target):
    """Mean per-joint position error (reference loss.py:11-17)."""
    p, t, _ = _prep(predicted, target)
    return _MpjpeFn.apply(p, t, None)


def weighted_mpjpe(predicted, target, w):
    """Weighted mean per-joint position error (reference loss.py:19-25); ``w`` broadcasts over the norm tensor."""
    p, t, w = _prep(predicted, target, w)
    return _MpjpeFn.apply(p, t, w)
