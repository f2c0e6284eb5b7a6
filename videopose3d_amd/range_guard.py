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
* the same spread over the COLUMNS of the two tensors that enter the stack from outside -- the input batch (joints x 2 keypoint
  columns: one hot joint) at the measured forward, the loss gradient at the head (joints x 3 columns) at that step's backward
  (``vp3d_range_cols``) -- so that the guard's claim does not rest on the measurement that a hot joint happens to be harmless
  (tools/range_edges.py: "joint x 10000" stays on the fp32 engine's own error because the hot column dominates both
  arithmetics alike); limit ``IO_SPREAD_MAX`` = 12;
* limits: ``ACT_SPREAD_MAX`` = 12 and ``W_SPREAD_MAX`` = 16 binary orders -- below them no element that is within 2^-5 of
  its channel's typical magnitude has lost a bit (activation bounds are ~2^6 loose: 12 + 6 <= 17; weight bounds are
  measured: 16 < 17);
* action: above a limit the model's calls run on the exact-fp32 engine (``engine.use_s16`` -> False, one warning), until its
  parameters are re-loaded (``load_state_dict`` / ``.to()``) and measure inside the limits again;
* no host synchronisation in the steady state of a SINGLE process: the statistic is launched every ``CHECK_EVERY`` calls behind
  the step's other work, its integers are copied to pinned host memory and looked at by a LATER call: only at calls that are a
  multiple of ``CONSUME_AFTER`` behind the launch, and only when the copy's event has completed (never waited for: a host that
  enqueues ahead of the GPU would stall; round 5 measured 1.2 ms per call for a blocking wait two calls after the launch).
  After ``load_state_dict`` / ``.to()`` / construction -- the abrupt changes -- the first call measures synchronously (one read-back
  per load, not per step);
* **data parallelism (round 6)**: for the TRAINING calls of a model whose gradients are exchanged across ranks
  (``dp.FlatGradSync(direct_module=model)``, world > 1) the decision is rank-consistent by construction instead of "host timing
  permitting": the five integers are MAX-all-reduced over the gradient exchange's process group before they are copied to the
  host (the parameter statistics are identical on every rank anyway; the input / head-gradient column spreads are rank-local
  data), launches and consumption count TRAINING calls only (evaluation calls, which one rank may run alone, neither launch
  nor consume nor take part in a collective), and a measurement launched at training call n is consumed at training call
  n + ``CONSUME_AFTER`` exactly -- waiting for the copy if it has to, which at eight steps behind costs a host that runs ahead
  nothing and a host that does not run ahead nothing either.  Every rank therefore switches engines (and re-captures a hipGraph)
  on the same step; ``tests/test_dp_gloo.py`` runs two ranks of which one measures a hot input column;
* parameter trips are sticky until the parameters are re-loaded; the input / head-gradient trips are SAMPLED checks (every
  ``CHECK_EVERY``-th batch is measured, non-contiguous inputs are not) and are re-evaluated at every later measurement: one
  outlier batch moves the model to the exact-fp32 engine until the next measured batch is inside the limit again, not for good.
  Under ``graph.GraphedTrainStep`` the head gradient of a replay is measured behind the replay (the captured backward cannot
  launch it); ``graph.GraphedStep`` (arbitrary step functions) has no handle on that tensor and leaves it unmeasured;
* what it protects is "not worse than the exact-fp32 engine", not "accurate": where BOTH arithmetics lose precision (e.g. every
  layer's beta at 2^14: gmax ~ 0.1 on either engine, tools/range_edges.py) the model stays on split-fp16.

A third integer rides along: the conditioning of the expand layer's BatchNorm statistics when they are taken from the input's
second-moment matrix (``vp3d_expand_stats_gram_s16``: var_n = W[n]^T Cov W[n], accurate to ~4e-9 * kappa_n with kappa_n =
sum |w_i Cov_ij w_j| / (var_n + eps)).  Temporal-difference filters over frame-to-frame correlated keypoints have kappa ~ 1e3..1e5;
the statistics kernel writes max_n floor(log2 kappa_n) during the training forwards, and at ``GRAM_KAPPA_LOG2_MAX`` (2^16: the
matrix path's error, 4e-9 * kappa / 2 of invstd, reaches what the reference's own fp32 conv + BatchNorm leaves, 1e-6 sqrt(kappa))
the layer goes back to the statistics pass over the conv output (no such term; +70 us per step), until the parameters are re-loaded.

``VP3D_RANGE_GUARD=0`` disables it (tools/range_edges.py measures the raw format that way).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time
import warnings

import torch

from . import _lib
from ._lib import check

ACT_SPREAD_MAX = 12
W_SPREAD_MAX = 16
IO_SPREAD_MAX = 12       # columns of the input batch / of the loss gradient at the head: A operands like the activations
CHECK_EVERY = 16
CONSUME_AFTER = 8        # a measurement launched at call n is looked at at calls n + 8, n + 16, ... (once its copy has completed)
GRAM_KAPPA_LOG2_MAX = 16  # include/vp3d.h: VP3D_GRAM_KAPPA_LOG2_MAX
EVAL_KFAC = 4.0          # eval: |bn(y)_c| ~ |gamma_c| * |xhat| + |beta_c| with |xhat| of a few standard deviations


def enabled() -> bool:
    return os.environ.get("VP3D_RANGE_GUARD", "1") != "0"


class _State:
    __slots__ = ("device", "out", "host", "event", "ws", "pending", "calls", "epoch", "tripped", "last", "checks", "sync_checks",
                 "launched_at", "gram_off", "gram_last", "last_call", "tick_us", "cols_ws", "armed", "io_last", "trip_param", "trip_io",
                 "warned", "dp_calls", "dp_epoch", "exchanges")

    def __init__(self, device, ws_ints):
        self.device = device
        cuda = torch.device(device).type == "cuda"   # (CPU: the decision logic alone, tests/test_dp_gloo.py)
        # [activation spread, weight-row spread, gram log2 kappa, input-column spread, head-gradient-column spread]
        self.out = torch.zeros(5, dtype=torch.int32, device=device)
        self.host = torch.zeros(5, dtype=torch.int32).pin_memory() if cuda else torch.zeros(5, dtype=torch.int32)
        self.cols_ws = torch.zeros(1024, dtype=torch.int32, device=device)
        self.armed = False           # the backward of the measured step adds the head gradient's column spread (out[4])
        self.io_last = None          # (input, head gradient) column spreads of the last consumed measurement
        self.launched_at = 0
        self.gram_off = False        # expand-layer statistics back on the pass over the conv output
        self.gram_last = None        # max floor(log2 kappa) seen by the last consumed measurement
        self.last_call = None        # (training, batch, frames) of the last eager tick: what a graph replay ticks with
        self.tick_us = 0.0           # host time spent in tick(), accumulated (bench.py reports it per call)
        self.event = torch.cuda.Event() if cuda else None
        self.ws = torch.empty(max(1, ws_ints), dtype=torch.int32, device=device)
        self.pending = False
        self.calls = 0
        self.epoch = -1          # the model's _range_epoch at the last completed measurement
        self.tripped = False     # trip_param or trip_io
        self.trip_param = False  # BatchNorm-bound / weight-row spread outside the limits: sticky until the parameters are re-loaded
        self.trip_io = False     # input / head-gradient column spread outside the limit at the LAST consumed measurement
        self.warned = False      # one warning per parameter epoch
        self.last = None         # (activation spread, weight-row spread) of the last completed measurement
        self.checks = 0
        self.sync_checks = 0
        self.dp_calls = 0        # TRAINING calls under data parallelism (what the rank-consistent cadence counts)
        self.dp_epoch = -1       # the parameter epoch of the last measurement that was exchanged across ranks
        self.exchanges = 0       # measurements that were MAX-reduced over the gradient exchange's process group


def invalidate(mod) -> None:
    """The parameters were replaced wholesale (load_state_dict, .to(), ...): the next call measures before it chooses an engine."""
    mod.__dict__["_range_epoch"] = mod.__dict__.get("_range_epoch", 0) + 1


def tripped(mod) -> bool:
    st = mod.__dict__.get("_range_state")
    return bool(st is not None and st.tripped and enabled())


def gram_disabled(mod) -> bool:
    """The expand layer's statistics must come from the pass over the conv output (ill-conditioned quadratic forms measured)."""
    st = mod.__dict__.get("_range_state")
    return bool(st is not None and st.gram_off and enabled())


def gram_flag(mod):
    """Device int32 scalar the statistics kernel atomicMax-es floor(log2 kappa) into (None before the first guarded call)."""
    st = mod.__dict__.get("_range_state")
    return st.out[2:] if st is not None and enabled() else None


def status(mod) -> dict:
    st = mod.__dict__.get("_range_state")
    if st is None:
        return dict(tripped=False, last=None, checks=0, sync_checks=0, gram_off=False, gram_log2_kappa=None)
    return dict(tripped=st.tripped, last=st.last, checks=st.checks, sync_checks=st.sync_checks,
                limits=(ACT_SPREAD_MAX, W_SPREAD_MAX), io_last=st.io_last, gram_off=st.gram_off, gram_log2_kappa=st.gram_last,
                trip_param=st.trip_param, trip_io=st.trip_io, exchanges=st.exchanges,
                tick_us_per_call=st.tick_us / max(1, st.calls + st.dp_calls))   # steady state: the synchronous measurements are not in it


def _ptrs(ts):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr


def measure_head_gradient(mod, gout3: torch.Tensor) -> None:
    """Backward of a step whose forward launched a measurement: column spread of the loss gradient at the head (out[4]; it
    reaches the host with the NEXT measurement's copy)."""
    st = mod.__dict__.get("_range_state")
    if st is None or not st.armed or not enabled() or torch.cuda.is_current_stream_capturing():
        return
    st.armed = False
    from . import ops
    g = gout3.detach()
    if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape[-1] <= 1024):
        return
    check(_lib.lib().vp3d_range_cols(ops._stream(), g.numel() // g.shape[-1], g.shape[-1], g.data_ptr(), g.shape[-1],
                                     st.cols_ws.data_ptr(), st.out[4:].data_ptr()), "vp3d_range_cols")


def _measure(mod, st: _State, m_rows, x3=None) -> None:
    """Enqueue the statistics kernels of one measurement on the current stream (st.out[0, 1, 3]; out[2] and out[4] have been
    accumulating since the last measurement).  The only part of the guard that touches the device kernels: tests of the
    decision logic replace it."""
    from . import engine, ops
    bns, convs = engine._bns(mod), engine._convs(mod)
    st.out[:2].zero_()                               # (out[2], out[4] accumulate over the steps since the last measurement)
    st.out[3:4].zero_()
    # (the caller hands over [B, T, J * F] -- the engines -- or [B, T, J, F] -- graph.GraphedTrainStep: columns = J * F either way)
    cols = int(getattr(mod, "num_joints_in", 0)) * int(getattr(mod, "in_features", 0))
    if (x3 is not None and x3.dtype == torch.float32 and x3.is_contiguous() and 0 < cols <= 1024 and x3.numel() % cols == 0):
        check(_lib.lib().vp3d_range_cols(ops._stream(), x3.numel() // cols, cols, x3.data_ptr(), cols,
                                         st.cols_ws.data_ptr(), st.out[3:].data_ptr()), "vp3d_range_cols")
        st.armed = True
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


def dp_sink(mod):
    """The model's gradient exchange (dp.FlatGradSync(direct_module=model)) when it spans more than one rank, else None."""
    sink = mod.__dict__.get("_vp3d_grad_sink")
    if sink is None or getattr(sink, "world", 1) <= 1:
        return None
    import torch.distributed as dist
    return sink if dist.is_initialized() else None


def _launch(mod, st: _State, m_rows, x3=None, sink=None) -> None:
    """Enqueue one measurement on the current stream: zero, statistics, (data parallel: MAX over the ranks,) copy to pinned
    memory, event."""
    _measure(mod, st, m_rows, x3)
    if sink is not None:
        import torch.distributed as dist
        # every rank acts on the job-wide maximum: the parameter statistics are identical anyway, the input / head-gradient
        # column spreads (and the conditioning flag, which depends on the batch) are rank-local data
        dist.all_reduce(st.out, op=dist.ReduceOp.MAX, group=sink.group)
        st.exchanges += 1
    st.host.copy_(st.out, non_blocking=True)
    st.out[2:3].zero_()
    st.out[4:].zero_()
    if st.event is not None:
        st.event.record()
    st.pending = True
    st.launched_at = st.dp_calls if sink is not None else st.calls
    st.checks += 1


def _consume(mod, st: _State, epoch: int) -> None:
    a, w, xi, go = int(st.host[0]), int(st.host[1]), int(st.host[3]), int(st.host[4])
    st.last, st.io_last, st.pending, st.epoch = (a, w), (xi, go), False, epoch
    st.trip_param = st.trip_param or a > ACT_SPREAD_MAX or w > W_SPREAD_MAX      # sticky within a parameter epoch
    st.trip_io = xi > IO_SPREAD_MAX or go > IO_SPREAD_MAX                        # a sampled check of THIS batch: re-evaluated
    now = st.trip_param or st.trip_io
    if now and not st.warned:
        st.warned = True
        warnings.warn("videopose3d_amd: intra-tensor dynamic range outside the split-fp16 format's lossless window "
                      "(per-channel BatchNorm bound spread 2^%d, limit 2^%d; weight-row spread 2^%d, limit 2^%d; input / head-gradient "
                      "column spread 2^%d / 2^%d, limit 2^%d): this model now runs on the exact-fp32 engine (math='f32' kernels) "
                      "until its parameters are re-loaded (parameter statistics) or a later measured batch is inside the limit again "
                      "(input / head-gradient columns)"
                      % (a, ACT_SPREAD_MAX, w, W_SPREAD_MAX, xi, go, IO_SPREAD_MAX), RuntimeWarning, stacklevel=3)
    st.tripped = now
    k = int(st.host[2])
    st.gram_last = k
    if k >= GRAM_KAPPA_LOG2_MAX and not st.gram_off:
        warnings.warn("videopose3d_amd: expand-layer BatchNorm statistics are ill-conditioned as a quadratic form of the input's "
                      "second-moment matrix (kappa 2^%d, limit 2^%d: difference-type filters over correlated input columns): this "
                      "model now takes them from a pass over the conv output until its parameters are re-loaded"
                      % (k, GRAM_KAPPA_LOG2_MAX), RuntimeWarning, stacklevel=3)
        st.gram_off = True


def tick(mod, training: bool, x3: torch.Tensor) -> None:
    """Top of every forward (and of every graph replay) of a model whose ``math`` is "f16x3": consume a finished measurement,
    measure synchronously after an abrupt parameter change, launch the periodic asynchronous measurement."""
    if not enabled() or getattr(mod, "math", "f32") != "f16x3" or not x3.is_cuda:
        return
    _tick(mod, bool(training), int(x3.shape[0]), int(x3.shape[1]), x3.device, x3)


def tick_replay(mod) -> None:
    """The same for a replay of a captured step whose inputs this module never sees (graph.GraphedStep): ticks with the shape of
    the model's last eager call (the capture's warm-up)."""
    st = mod.__dict__.get("_range_state")
    if st is None or st.last_call is None or not enabled() or getattr(mod, "math", "f32") != "f16x3":
        return
    _tick(mod, *st.last_call, st.device)


def _tick(mod, training: bool, b: int, t_in: int, device, x3=None) -> None:
    if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
        return                                       # (a captured step is guarded by its replay wrapper: graph.py)
    t0 = time.perf_counter()
    from . import engine, engine_s16
    if (not engine_s16.supported(mod, t_in, training, batch=b) or
            mod._plan.forward_flops(b, t_in) < engine.S16_MIN_FORWARD_FLOPS[bool(training)]):
        return                                       # this call runs on the fp32 kernels anyway
    st: _State = mod.__dict__.get("_range_state")
    if st is None or st.device != device:
        st = mod.__dict__["_range_state"] = _State(device, sum(c.weight.shape[0] for c in engine._convs(mod)))
    st.last_call = (training, b, t_in)
    epoch = mod.__dict__.get("_range_epoch", 0)
    m_rows = None
    if training:
        plan = mod._plan
        t_len = plan.lengths(t_in)
        m_rows = [b * t_len[0]] + [b * t_len[(idx + 1) // 2] for idx in range(1, len(plan.convs))]
    dp = dp_sink(mod)
    sink = dp if training else None                  # (evaluation calls never take part in a collective: one rank may run them alone)
    if st.epoch != epoch or (sink is not None and st.dp_epoch != epoch):
        # abrupt change (construction, load_state_dict, .to()): measure NOW, before an engine is chosen for these parameters
        # (data parallel: the first TRAINING call after the change exchanges it -- every rank makes that call at the same step)
        st.tripped = st.trip_param = st.trip_io = st.gram_off = st.warned = False
        st.out[2:].zero_()
        _launch(mod, st, m_rows, x3, sink)
        if st.event is not None:
            st.event.synchronize()
        st.sync_checks += 1
        _consume(mod, st, epoch)
        if sink is not None:
            st.dp_epoch = epoch
        st.calls = st.dp_calls = 0                   # the first periodic measurement follows the first forward (its kappa)
        st.launched_at = 0
        return                                       # (not in tick_us: a one-off that waits for the device and loads the kernels)
    if dp is not None and sink is None:
        return                                       # an evaluation call of a data-parallel model: the training calls' decisions stand
    if sink is not None:
        # rank-consistent cadence: training calls only, consumption at a FIXED distance behind the launch (waiting for the copy
        # if need be: eight steps behind, a host that runs ahead of the GPU has nothing to wait for)
        st.dp_calls += 1
        if st.pending and st.dp_calls - st.launched_at == CONSUME_AFTER:
            if st.event is not None:
                st.event.synchronize()
            _consume(mod, st, epoch)
        if not st.trip_param and not st.pending and st.dp_calls % CHECK_EVERY == 1:
            _launch(mod, st, m_rows, x3, sink)
        st.tick_us += (time.perf_counter() - t0) * 1e6
        return
    st.calls += 1
    behind = st.calls - st.launched_at
    every = max(1, min(CONSUME_AFTER, CHECK_EVERY // 2))
    if st.pending and behind > 0 and behind % every == 0 and (st.event is None or st.event.query()):
        _consume(mod, st, epoch)
    if not st.trip_param and not st.pending and st.calls % CHECK_EVERY == 1:
        _launch(mod, st, m_rows, x3)
    st.tick_us += (time.perf_counter() - t0) * 1e6
