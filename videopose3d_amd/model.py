"""Drop-in replacements for the reference ``common/model.py`` classes, executed on MI355X HIP kernels.

Same constructor signatures, ``forward`` contract, helper methods and -- strictly -- the same ``state_dict``
keys / shapes / dtypes as reference common/model.py:10-197, so that ``run.py`` (``from common.model import *``,
run.py:21) works unchanged: ``.cuda()``, ``.train()/.eval()``, ``parameters()`` for Adam,
``load_state_dict(model_pos_train.state_dict())`` (run.py:426), checkpoints (run.py:600-608).

The ``nn.Conv1d`` / ``nn.BatchNorm1d`` sub-modules exist ONLY as parameter / buffer containers (that is what
fixes the state_dict names and reproduces the reference's default initialisation, RNG order included).  Their
``forward`` is never called: all arithmetic runs in libvp3d.so through ``engine``.  There is no CPU path.
"""
from __future__ import annotations

import copy
import os
import warnings

import torch
import torch.nn as nn

from . import engine
from ._lib import Vp3dError
from .plan import make_plan


_default_math = [None]


def default_math() -> str:
    """Arithmetic newly constructed models use: set_default_math() if called, else VP3D_MATH, else "f16x3"."""
    m = _default_math[0] or os.environ.get("VP3D_MATH", "f16x3")
    if m not in ("f32", "f16x3"):
        raise Vp3dError("VP3D_MATH / set_default_math: expected 'f32' or 'f16x3', got %r" % (m,))
    return m


def set_default_math(math) -> None:
    _default_math[0] = math


class TemporalModelBase(nn.Module):
    """Do not instantiate this class (mirrors reference model.py:10-77)."""

    _kind = None
    _n_models = 0
    _warned_eval_grad = False

    def __init__(self, num_joints_in, in_features, num_joints_out, filter_widths, causal, dropout, channels):
        super().__init__()
        for fw in filter_widths:
            assert fw % 2 != 0, "Only odd filter widths are supported"
        self.num_joints_in = num_joints_in
        self.in_features = in_features
        self.num_joints_out = num_joints_out
        self.filter_widths = filter_widths
        # creation order == reference (model.py:28-33) so that seeded default init gives identical weights
        self.drop = nn.Dropout(dropout)
        self.relu = nn.ReLU(inplace=True)
        self.pad = [filter_widths[0] // 2]
        self.expand_bn = nn.BatchNorm1d(channels, momentum=0.1)
        self.shrink = nn.Conv1d(channels, num_joints_out * 3, 1)
        self._stats_epoch = 0
        self._drop_calls = 0
        self._drop_seed = None
        TemporalModelBase._n_models += 1
        self._ordinal = TemporalModelBase._n_models      # construction order in this process: separates the models' mask streams
        self._drop_counter = None       # optional device uint64 step counter added to the dropout offset (graph.py)
        self._bn_momentum_host = None
        self._bn_momentum_dev = None    # optional device float read by the BatchNorm finalize launches instead of their
                                        # momentum argument (graph.py: a captured step follows set_bn_momentum)
        # GEMM arithmetic (not part of the reference API / state_dict): "f16x3" = split-fp16 operands on
        # v_mfma_f32_32x32x16_f16 (fp32-class results: 22+ operand bits, exact products, fp32 accumulation; ~3x the
        # fp32 matrix rate) wherever engine_s16.supported() says so, "f32" = v_mfma_f32_32x32x2_f32 everywhere.
        self.math = default_math()

    # ---- construction helper shared by the two concrete classes -------------------------------------
    def _build(self, channels, causal, dense, strided):
        fw = self.filter_widths
        c_in = self.num_joints_in * self.in_features
        plan = make_plan("strided" if strided else "dilated", c_in, channels, self.num_joints_out * 3, fw, causal,
                         dense)
        self.expand_conv = nn.Conv1d(c_in, channels, fw[0], stride=plan.convs[0].stride, bias=False)
        layers_conv, layers_bn = [], []
        for spec in plan.convs[1:]:
            layers_conv.append(nn.Conv1d(channels, channels, spec.taps, dilation=spec.dil, stride=spec.stride,
                                         bias=False))
            layers_bn.append(nn.BatchNorm1d(channels, momentum=0.1))
        self.pad = list(plan.pad)
        self.causal_shift = list(plan.causal_shift)
        self.layers_conv = nn.ModuleList(layers_conv)
        self.layers_bn = nn.ModuleList(layers_bn)
        self._plan = plan

    # ---- reference helper API -------------------------------------------------------------------------
    def set_bn_momentum(self, momentum):
        self.expand_bn.momentum = momentum
        for bn in self.layers_bn:
            bn.momentum = momentum
        if self._bn_momentum_dev is not None:
            self._bn_momentum_dev.fill_(float(momentum))

    def _uniform_bn_momentum(self):
        """The one momentum all BatchNorm layers share (what set_bn_momentum / the constructor leave behind), else None."""
        moms = {self.expand_bn.momentum} | {bn.momentum for bn in self.layers_bn}
        mom = next(iter(moms))
        return float(mom) if len(moms) == 1 and mom is not None else None

    def _momentum_dev_ptr(self):
        """Device address the BatchNorm finalize launches read their momentum from, or None (launch argument).  Kept in
        step with the modules' ``momentum`` attributes here, so setting ``bn.momentum`` directly also reaches a replay."""
        t = self._bn_momentum_dev
        if t is None:
            return None
        mom = self._uniform_bn_momentum()
        if mom is None:
            return None                               # per-layer momenta: launch arguments (a graph re-captures on change)
        if mom != self._bn_momentum_host:
            t.fill_(mom)
            self._bn_momentum_host = mom
        return t.data_ptr()

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .cpu(): the device-side step counter and momentum are plain attributes (not state_dict entries),
        so they follow the parameters here -- a launch on cuda:1 must never be handed a cuda:0 address."""
        out = super()._apply(fn, *args, **kwargs)
        from . import range_guard
        range_guard.invalidate(self)                 # (the dynamic-range guard measures again before the next call)
        dev = self.shrink.weight.device
        for name in ("_drop_counter", "_bn_momentum_dev"):
            t = getattr(self, name, None)
            if t is not None and t.device != dev:
                setattr(self, name, t.to(dev) if dev.type == "cuda" else None)
        return out

    def load_state_dict(self, state_dict, *args, **kwargs):
        """nn.Module.load_state_dict (run.py:209-210,219,303-304,426,429); the split-fp16 engine's dynamic-range guard measures
        the new parameters before the next call chooses an arithmetic (range_guard.py)."""
        out = super().load_state_dict(state_dict, *args, **kwargs)
        from . import range_guard
        range_guard.invalidate(self)
        return out

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_range_state":                  # (device buffers + an event of the guard: the copy builds its own)
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        # a copy is a model of its own: its own dropout mask stream (the seed mixes the ordinal) and device-side counters
        TemporalModelBase._n_models += 1
        new._ordinal = TemporalModelBase._n_models
        new._drop_seed = None
        new._drop_counter = None
        new._bn_momentum_dev = None
        return new

    def receptive_field(self):
        """Total receptive field of this model as # of frames."""
        return self._plan.receptive_field()

    def total_causal_shift(self):
        """Asymmetric offset for sequence padding (value reproduced as the reference computes it)."""
        return self._plan.total_causal_shift()

    def backward_param_groups(self):
        """Parameters grouped in the order in which engine.backward_train finishes their gradients (shrink, last
        block ... first block, expand): the bucket order of dp.FlatGradSync's overlapped gradient exchange."""
        groups = [[self.shrink.weight, self.shrink.bias]]
        for i in reversed(range(len(self.layers_conv) // 2)):
            groups.append([self.layers_conv[2 * i + 1].weight, self.layers_bn[2 * i + 1].weight,
                           self.layers_bn[2 * i + 1].bias, self.layers_conv[2 * i].weight,
                           self.layers_bn[2 * i].weight, self.layers_bn[2 * i].bias])
        groups.append([self.expand_conv.weight, self.expand_bn.weight, self.expand_bn.bias])
        return groups

    # ---- dropout stream ---------------------------------------------------------------------------------
    def _dropout_counter_ptr(self):
        return None if self._drop_counter is None else self._drop_counter.data_ptr()

    def _next_dropout_state(self):
        if self._drop_seed is None:
            rank = int(os.environ.get("RANK", "0"))
            # (model ordinal, not id(self): the same script gives the same masks in every process / on every rerun)
            self._drop_seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03 + self._ordinal) \
                & 0xFFFFFFFFFFFFFFFF
        off = self._drop_calls
        self._drop_calls += 1
        return self._drop_seed, off

    # ---- forward ----------------------------------------------------------------------------------------
    def forward(self, x):
        assert len(x.shape) == 4
        assert x.shape[-2] == self.num_joints_in
        assert x.shape[-1] == self.in_features
        if not x.is_cuda:
            raise Vp3dError("videopose3d_amd runs on MI355X HIP kernels only: got a %s tensor. Move the model and "
                            "the input to the GPU (there is deliberately no CPU fallback)." % x.device)
        if self.expand_conv.weight.device != x.device:
            raise Vp3dError("model parameters are on %s but the input is on %s"
                            % (self.expand_conv.weight.device, x.device))
        b, t = x.shape[0], x.shape[1]
        x3 = x.to(torch.float32).contiguous().view(b, t, -1)     # [B,T,J*F] is already channels-last rows
        with torch.cuda.device(x.device):
            if self.training:
                if torch.is_grad_enabled():
                    out3 = engine.TemporalStackFn.apply(self, x3, *engine.param_list(self))
                else:
                    with torch.no_grad():
                        out3, _ = engine.forward_train(self, x3, save=False)
            else:
                if torch.is_grad_enabled() and (x3.requires_grad or any(p.requires_grad for p in self.parameters())):
                    # the reference's eval-mode forward is differentiable (model.py:63-77): BatchNorm on its running
                    # statistics, exact-fp32 kernels, hand-written backward without the batch-statistic terms.  Evaluation
                    # under torch.no_grad() (as run.py does it) takes the folded-BatchNorm fast path below.
                    if not x3.requires_grad and not TemporalModelBase._warned_eval_grad:
                        TemporalModelBase._warned_eval_grad = True
                        warnings.warn("videopose3d_amd: eval-mode forward with autograd enabled runs the differentiable "
                                      "exact-fp32 path and keeps every activation for backward (like the reference); wrap "
                                      "evaluation in torch.no_grad() -- as run.py does -- for the folded-BatchNorm fast path",
                                      stacklevel=3)
                    out3 = engine.FrozenStackFn.apply(self, x3, *engine.param_list(self))
                else:
                    with torch.no_grad():
                        out3 = engine.forward_eval(self, x3)
        return out3.view(b, -1, self.num_joints_out, 3)


class TemporalModel(TemporalModelBase):
    """3D pose model with dilated temporal convolutions (all use-cases; reference model.py:79-138)."""

    def __init__(self, num_joints_in, in_features, num_joints_out, filter_widths, causal=False, dropout=0.25,
                 channels=1024, dense=False):
        super().__init__(num_joints_in, in_features, num_joints_out, filter_widths, causal, dropout, channels)
        self._build(channels, causal, dense, strided=False)


class TemporalModelOptimized1f(TemporalModelBase):
    """Single-frame-batching variant with strided convolutions (training, stride 1; reference model.py:140-197).
    Weights are interchangeable with ``TemporalModel``."""

    def __init__(self, num_joints_in, in_features, num_joints_out, filter_widths, causal=False, dropout=0.25,
                 channels=1024):
        super().__init__(num_joints_in, in_features, num_joints_out, filter_widths, causal, dropout, channels)
        self._build(channels, causal, False, strided=True)
