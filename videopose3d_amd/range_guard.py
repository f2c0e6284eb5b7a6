"""Guard of the split-fp16 ("f16x3") arithmetic against adversarial intra-tensor dynamic range.

The S16 operand format (csrc/vp3d_s16.h) keeps ONE exponent per tensor, taken from a guaranteed bound of the tensor's
maximum: an element keeps its 22+ bits while it is within 2^-17 of that bound and an absolute error of 2^-40 of the bound
below.  The reference puts no constraint on the BatchNorm affine or on the conv weights (common/model.py:32,102,113-119):
one hot ``gamma_c`` / ``beta_c`` raises the exponent of the whole activation, one hot weight row that of the whole pack.
Measured on MI355X against the float64 oracle (tools/range_edges.py): with ONE hot channel in one layer the
engine stays on the exact-fp32 engine's own error up to a factor 2^20 (the hot channel dominates every contraction it enters,
in either arithmetic); with the SAME channel hot in every layer both arithmetics lose precision and f16x3 loses it ~4x
sooner from a factor 2^14 on.  The guard keeps the model out of that regime:

* statistic (device, ``vp3d_range_stats``): log2 of (largest / median) of the per-channel activation bound
  ``|gamma_c| * sqrt(M - 1) + |beta_c|`` over every BatchNorm layer, and of the per-output-row maximum of every conv weight;
* limits: ``ACT_SPREAD_MAX`` = 12 and ``W_SPREAD_MAX`` = 16 binary orders -- below them no element that is within 2^-5 of
  its channel's typical magnitude has lost a bit (activation bounds are ~2^6 loose: 12 + 6 <= 17; weight bounds are
  measured: 16 < 17);
* action: above a limit the model's calls run on the exact-fp32 engine (``engine.use_s16`` -> False, one warning), until its
  parameters are re-loaded (``load_state_dict`` / ``.to()``) and measure inside the limits again;
* no host synchronisation in the steady state: the statistic is launched every ``CHECK_EVERY`` calls behind the step's other
  work, its two integers are copied to pinned host memory and looked at by a LATER call once the copy's event has completed
  (parameters move by an optimizer step at a time; the limits sit 2^5..2^8 below the first measurable effect).  After
  ``load_state_dict`` / ``.to()`` / construction -- the abrupt changes -- the first call measures synchronously (one 8-byte
  read-back per load, not per step).

``VP3D_RANGE_GUARD=0`` disables it (tools/range_edges.py measures the raw format that way).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings

import torch

from . import _lib
from ._lib import check

ACT_SPREAD_MAX = 12
W_SPREAD_MAX = 16
CHECK_EVERY = 16
EVAL_KFAC = 4.0          # eval: |bn(y)_c| ~ |gamma_c| * |xhat| + |beta_c| with |xhat| of a few standard deviations


def enabled() -> bool:
    return os.environ.get("VP3D_RANGE_GUARD", "1") != "0"


class _State:
    __slots__ = ("device", "out", "host", "event", "ws", "pending", "calls", "epoch", "tripped", "last", "checks", "sync_checks")

    def __init__(self, device, ws_ints):
        self.device = device
        self.out = torch.zeros(2, dtype=torch.int32, device=device)
        self.host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        self.ws = torch.empty(max(1, ws_ints), dtype=torch.int32, device=device)
        self.pending = False
        self.calls = 0
        self.epoch = -1          # the model's _range_epoch at the last completed measurement
        self.tripped = False
        self.last = None         # (activation spread, weight-row spread) of the last completed measurement
        self.checks = 0
        self.sync_checks = 0


def invalidate(mod) -> None:
    """The parameters were replaced wholesale (load_state_dict, .to(), ...): the next call measures before it chooses an engine."""
    mod.__dict__["_range_epoch"] = mod.__dict__.get("_range_epoch", 0) + 1


def tripped(mod) -> bool:
    st = mod.__dict__.get("_range_state")
    return bool(st is not None and st.tripped and enabled())


def status(mod) -> dict:
    st = mod.__dict__.get("_range_state")
    if st is None:
        return dict(tripped=False, last=None, checks=0, sync_checks=0)
    return dict(tripped=st.tripped, last=st.last, checks=st.checks, sync_checks=st.sync_checks,
                limits=(ACT_SPREAD_MAX, W_SPREAD_MAX))


def _ptrs(ts):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr


def _launch(mod, st: _State, m_rows) -> None:
    """Enqueue one measurement on the current stream: zero, statistics, 8-byte copy to pinned memory, event."""
    from . import engine, ops
    bns, convs = engine._bns(mod), engine._convs(mod)
    st.out.zero_()
    lim = int(_lib.lib().vp3d_range_max_tensors())
    stream = ops._stream()
    for lo in range(0, max(len(bns), len(convs)), lim):
        bn = [b for b in bns[lo:lo + lim] if b.weight is not None and b.bias is not None]
        kf = [math.sqrt(max(1.0, float(m_rows[lo + i]) - 1.0)) if m_rows is not None else EVAL_KFAC
              for i, b in enumerate(bns[lo:lo + lim]) if b.weight is not None and b.bias is not None]
        ws_ = [c.weight.detach() for c in convs[lo:lo + lim]]
        for w in ws_:
            assert w.is_contiguous() and w.dtype == torch.float32
        rows = [w.shape[0] for w in ws_]
        rlen = [w.shape[1] * w.shape[2] for w in ws_]
        check(_lib.lib().vp3d_range_stats(stream, len(bn), bns[0].num_features, _ptrs([b.weight for b in bn]),
                                          _ptrs([b.bias for b in bn]), (C.c_float * max(1, len(kf)))(*kf), len(ws_),
                                          _ptrs(ws_), (C.c_int64 * max(1, len(rows)))(*rows),
                                          (C.c_int64 * max(1, len(rlen)))(*rlen), st.ws.data_ptr(), st.ws.numel(),
                                          st.out.data_ptr()), "vp3d_range_stats")
    st.host.copy_(st.out, non_blocking=True)
    st.event.record()
    st.pending = True
    st.checks += 1


def _consume(mod, st: _State, epoch: int) -> None:
    a, w = int(st.host[0]), int(st.host[1])
    st.last, st.pending, st.epoch = (a, w), False, epoch
    now = a > ACT_SPREAD_MAX or w > W_SPREAD_MAX
    if now and not st.tripped:
        warnings.warn("videopose3d_amd: intra-tensor dynamic range outside the split-fp16 format's lossless window "
                      "(per-channel BatchNorm bound spread 2^%d, limit 2^%d; weight-row spread 2^%d, limit 2^%d): this model "
                      "now runs on the exact-fp32 engine (math='f32' kernels) until its parameters are re-loaded"
                      % (a, ACT_SPREAD_MAX, w, W_SPREAD_MAX), RuntimeWarning, stacklevel=3)
    st.tripped = st.tripped or now                   # sticky within a parameter epoch (tick clears it on a re-load)


def tick(mod, training: bool, x3: torch.Tensor) -> None:
    """Top of every forward (and of every graph replay) of a model whose ``math`` is "f16x3": consume a finished measurement,
    measure synchronously after an abrupt parameter change, launch the periodic asynchronous measurement."""
    if not enabled() or getattr(mod, "math", "f32") != "f16x3" or not x3.is_cuda:
        return
    if torch.cuda.is_current_stream_capturing():
        return                                       # (a captured step is guarded by its replay wrapper: graph.py)
    from . import engine, engine_s16
    if (not engine_s16.supported(mod, x3.shape[1], training, batch=x3.shape[0]) or
            mod._plan.forward_flops(x3.shape[0], x3.shape[1]) < engine.S16_MIN_FORWARD_FLOPS[bool(training)]):
        return                                       # this call runs on the fp32 kernels anyway
    st: _State = mod.__dict__.get("_range_state")
    if st is None or st.device != x3.device:
        st = mod.__dict__["_range_state"] = _State(x3.device, sum(c.weight.shape[0] for c in engine._convs(mod)))
    epoch = mod.__dict__.get("_range_epoch", 0)
    m_rows = None
    if training:
        plan = mod._plan
        t_len = plan.lengths(x3.shape[1])
        b = x3.shape[0]
        m_rows = [b * t_len[0]] + [b * t_len[(idx + 1) // 2] for idx in range(1, len(plan.convs))]
    if st.epoch != epoch:
        # abrupt change (construction, load_state_dict, .to()): measure NOW, before an engine is chosen for these parameters
        st.tripped = False
        _launch(mod, st, m_rows)
        st.event.synchronize()
        st.sync_checks += 1
        _consume(mod, st, epoch)
        return
    if st.pending and st.event.query():
        _consume(mod, st, epoch)
    st.calls += 1
    if not st.tripped and not st.pending and st.calls % CHECK_EVERY == 0:
        _launch(mod, st, m_rows)
