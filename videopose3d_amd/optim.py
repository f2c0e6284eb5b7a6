"""Fused Adam / AMSGrad for the training loop (reference run.py:252,264: ``optim.Adam(params, lr, amsgrad=True)``,
run.py:420 ``optimizer.step()``, run.py:583-588 per-epoch ``param_group['lr'] *= lr_decay``).

``FlatAdam`` is a ``torch.optim.Optimizer`` whose parameters, gradients and state tensors are views into flat
fp32 buffers, so that ``step()`` is ONE pass of vp3d_adam_step over 16.95 M parameters (36 B/param of HBM traffic,
~0.1 ms) instead of torch's multi-tensor kernel chain, and so that the data-parallel gradient exchange
(dp.FlatGradSync) and the optimizer share one gradient buffer.  ``state_dict()`` has the layout of
``torch.optim.Adam`` (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` / ``max_exp_avg_sq``), so reference
checkpoints (run.py:600-608) load into it and its checkpoints load into ``torch.optim.Adam``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops
from ._lib import check


def _bump_version(t: torch.Tensor):
    """The kernel writes parameters through raw pointers; tell autograd / our eval-fold cache that they changed."""
    try:
        torch.autograd.graph.increment_version(t)
    except AttributeError:                                       # older torch
        torch._C._autograd._increment_version(t) if hasattr(torch._C._autograd, "_increment_version") else t.add_(0)


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False,
                 grad_sync=None):
        """grad_sync: an existing dp.FlatGradSync over the SAME parameters (its flat gradient buffer is reused);
        otherwise a private flat gradient buffer is created and every ``p.grad`` becomes a view into it."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise _lib.Vp3dError("FlatAdam supports a single parameter group (as run.py uses)")
        ps = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if not ps:
            raise _lib.Vp3dError("FlatAdam: no trainable parameters")
        if grad_sync is not None:
            # the ONE elementwise pass needs parameters, gradients and state in the same flat order: adopt the
            # gradient buffer's (backward-completion) order
            if {id(p) for p in grad_sync.params} != {id(p) for p in ps} or len(grad_sync.params) != len(ps):
                raise _lib.Vp3dError("FlatAdam: grad_sync was built over a different parameter set")
            ps = list(grad_sync.params)
        dev = ps[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in ps):
            raise _lib.Vp3dError("FlatAdam needs fp32 parameters on one GPU (move the model with .cuda() first, "
                                 "as run.py:250-252 does); there is no CPU path")
        from .dp import FlatGradSync, flat_layout
        self._ps = ps
        # same 256-byte-aligned slots as the flat gradient buffer: parameter views stay legal operands of the
        # LDS-DMA GEMM path (1x1 conv weights are used in place) and ONE launch covers the whole range
        self._offs, self._n = flat_layout(ps)
        self._flat_p = torch.zeros(self._n, dtype=torch.float32, device=dev)
        for p, off in zip(ps, self._offs):
            n = p.numel()
            self._flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self._flat_p[off:off + n].view_as(p)
        if grad_sync is not None:
            self._sync = grad_sync
        else:
            self._sync = FlatGradSync(ps, world=1)
        self._flat_g = self._sync.flat
        self._steps = 0
        self._state_bufs = None

    # ---- state ------------------------------------------------------------------------------------------
    def _ensure_state(self):
        if self._state_bufs is not None:
            return
        ams = bool(self.param_groups[0]["amsgrad"])
        dev = self._flat_p.device
        bufs = [torch.zeros(self._n, dtype=torch.float32, device=dev) for _ in range(3 if ams else 2)]
        self._state_bufs = bufs
        # torch.optim.Adam keeps `step` as one CPU fp32 0-dim tensor PER parameter (its foreach path increments each
        # of them in place, so they must not alias); they are brought up to date in state_dict()
        self._step_ts = [torch.zeros((), dtype=torch.float32) for _ in self._ps]
        for p, off, step_t in zip(self._ps, self._offs, self._step_ts):
            n = p.numel()
            st = self.state[p]
            st["step"] = step_t
            st["exp_avg"] = bufs[0][off:off + n].view_as(p)
            st["exp_avg_sq"] = bufs[1][off:off + n].view_as(p)
            if ams:
                st["max_exp_avg_sq"] = bufs[2][off:off + n].view_as(p)

    def load_state_dict(self, state_dict):
        """Accepts a torch.optim.Adam / FlatAdam state dict; state tensors are copied into the flat buffers."""
        super().load_state_dict(state_dict)
        loaded = {p: dict(self.state[p]) for p in self._ps if p in self.state and len(self.state[p])}
        for p in self._ps:
            self.state[p].clear() if p in self.state else None
        self._state_bufs = None
        if not loaded:
            self._steps = 0
            return
        self._ensure_state()
        steps = set()
        for p, st in loaded.items():
            for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                if k in st and k in self.state[p]:
                    self.state[p][k].copy_(st[k])
            steps.add(int(float(st["step"])))
        if len(steps) != 1:
            raise _lib.Vp3dError("FlatAdam: parameters carry different step counts %s" % sorted(steps))
        self._steps = steps.pop()
        self._sync_step_tensors()

    def _sync_step_tensors(self):
        if self._state_bufs is not None:
            for t in self._step_ts:
                t.fill_(float(self._steps))

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def zero_grad(self, set_to_none: bool = True):
        """Zeroes the flat gradient buffer; the .grad views are kept (set_to_none would detach them)."""
        self._sync.zero_grad()

    @property
    def grad_sync(self):
        return self._sync

    # ---- step -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        for p in self._ps:
            if self._sync.view_for(p) is None:       # .grad is None or no longer the flat view
                raise _lib.Vp3dError("FlatAdam: a parameter's .grad is no longer the flat-buffer view (use "
                                     "optimizer.zero_grad() of this class, not set_to_none on the module)")
        self._ensure_state()
        self._steps += 1
        h = _lib.Adam(float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                      float(g["weight_decay"]), self._steps, 1 if g["amsgrad"] else 0)
        bufs = self._state_bufs
        with torch.cuda.device(self._flat_p.device):
            check(_lib.lib().vp3d_adam_step(ops._stream(), self._n, self._flat_p.data_ptr(),
                                            self._flat_g.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(),
                                            bufs[2].data_ptr() if len(bufs) > 2 else None, C.byref(h)),
                  "vp3d_adam_step")
        for p in self._ps:
            _bump_version(p)
        return loss
