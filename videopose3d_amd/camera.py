"""Differentiable Human3.6M camera projection on the HIP kernels (BASELINE config 5: semi-supervised
back-projection loss).  Same call contract as reference common/camera.py:37-67 (``project_to_2d``) and
:69-90 (``project_to_2d_linear``): X [N, *, 3] camera-space points, camera_params [N, 9] -> [N, *, 2].
Camera parameters carry no gradient (run.py:328-331)."""
from __future__ import annotations

import torch

from . import _lib, ops
from ._lib import check


def _stream():
    return ops._stream()


class _ProjectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, cam, linear):
        n = X.shape[0]
        ppc = X.numel() // (3 * n)
        Xc = X.contiguous().float()
        camc = cam.contiguous().float()
        out = torch.empty(X.shape[:-1] + (2,), dtype=torch.float32, device=X.device)
        check(_lib.lib().vp3d_project_to_2d_fwd(_stream(), n, ppc, Xc.data_ptr(), camc.data_ptr(), int(linear),
                                                out.data_ptr()), "vp3d_project_to_2d_fwd")
        ctx.save_for_backward(Xc, camc)
        ctx.linear = int(linear)
        ctx.x_dtype = X.dtype
        return out

    @staticmethod
    def backward(ctx, gout):
        Xc, camc = ctx.saved_tensors
        n = Xc.shape[0]
        ppc = Xc.numel() // (3 * n)
        g = gout.contiguous().float()
        dX = torch.empty_like(Xc)
        check(_lib.lib().vp3d_project_to_2d_bwd(_stream(), n, ppc, Xc.data_ptr(), camc.data_ptr(), g.data_ptr(),
                                                ctx.linear, dX.data_ptr()), "vp3d_project_to_2d_bwd")
        return dX.to(ctx.x_dtype), None, None


def _project(X, camera_params, linear):
    assert X.shape[-1] == 3
    assert len(camera_params.shape) == 2
    assert camera_params.shape[-1] == 9
    assert X.shape[0] == camera_params.shape[0]
    if not X.is_cuda:
        raise _lib.Vp3dError("project_to_2d: the HIP path needs CUDA/HIP tensors (no CPU fallback)")
    if not camera_params.is_cuda or camera_params.device != X.device:
        raise _lib.Vp3dError("project_to_2d: camera_params is on %s but X is on %s (the kernel reads both by device "
                             "pointer)" % (camera_params.device, X.device))
    with torch.cuda.device(X.device):
        return _ProjectFn.apply(X, camera_params, linear)


def project_to_2d(X, camera_params):
    """Project 3D points to 2D using the Human3.6M camera projection function (with distortion)."""
    return _project(X, camera_params, False)


def project_to_2d_linear(X, camera_params):
    """Project 3D points to 2D using only linear parameters (focal length and principal point)."""
    return _project(X, camera_params, True)
