"""Execution of the temporal stack on the HIP kernels: eval forward (BN folded into the GEMM epilogue),
training forward (GEMM + fused BN-statistics epilogue -> finalize -> BN/ReLU/dropout/residual) and the
hand-written backward (BN backward -> wgrad / dgrad GEMMs with the residual-gradient scatter fused in the
dgrad epilogue), exposed to autograd through one ``torch.autograd.Function`` for the whole stack.

Mirrors the wiring of reference common/model.py:126-138 (dilated) / :187-197 (strided):
    h0 = act(conv(x; expand_conv); expand_bn)
    for each block:  u = act(conv(h; layers_conv[2i]); layers_bn[2i])
                     h = res_i(h) + act(conv(u; layers_conv[2i+1]); layers_bn[2i+1])
    out = conv(h; shrink) + shrink.bias
with act = dropout(relu(bn(.))).
"""
from __future__ import annotations

import collections
import os
from typing import List, Optional

import torch

from . import ops, range_guard
from ._switches import SW
from .plan import ConvSpec, StackPlan


def _convs(mod) -> List[torch.nn.Conv1d]:
    return [mod.expand_conv] + list(mod.layers_conv)


def _bns(mod) -> List[torch.nn.BatchNorm1d]:
    return [mod.expand_bn] + list(mod.layers_bn)


def param_list(mod) -> List[torch.nn.Parameter]:
    """Order of the parameter tensors handed to the autograd Function."""
    ps = []
    for conv, bn in zip(_convs(mod), _bns(mod)):
        ps += [conv.weight, bn.weight, bn.bias]
    ps += [mod.shrink.weight, mod.shrink.bias]
    return ps


# --------------------------------------------------------------------------------------------------------
# eval: BN folded into packed weights + bias, cached until a parameter / buffer changes
# --------------------------------------------------------------------------------------------------------
def _eval_key(mod):
    key = [mod._stats_epoch]
    for conv, bn in zip(_convs(mod), _bns(mod)):
        for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var):
            key.append((t.data_ptr(), t._version))
    key.append((mod.shrink.weight.data_ptr(), mod.shrink.weight._version))
    return tuple(key)


def folded_weights(mod):
    key = _eval_key(mod)
    cache = mod.__dict__.get("_fold_cache")
    if cache is not None and cache[0] == key:
        return cache[1]
    packs = []
    for idx, (conv, bn) in enumerate(zip(_convs(mod), _bns(mod))):
        scale, shift = ops.bn_fold(bn)
        kpad = ops.padded_k(mod._plan.convs[idx]) if idx == 0 else 0
        packs.append((ops.pack_weight(conv.weight.detach(), scale, ld_out=kpad or None), shift))
    mod.__dict__["_fold_cache"] = (key, packs)
    return packs


def _expand_input(plan: StackPlan, x3: torch.Tensor, one_col: int = -1):
    """expand_conv reads taps*C_in = 102 contiguous floats per output row: stage them as 128-wide zero-padded
    rows (one cheap pass over the 34-channel input) so that the GEMM runs on the LDS-DMA fast path.  one_col: a
    padding column set to 1 (engine_s16's expand-layer backward; the matching weight column is zero)."""
    spec = plan.convs[0]
    kpad = ops.padded_k(spec)
    if not kpad:
        return x3, spec, 0
    return ops.im2row(x3, spec, kpad, one_col), ConvSpec(kpad, spec.c_out, 1, 1, 1), kpad


def _shrink(mod, h: torch.Tensor) -> torch.Tensor:
    w = mod.shrink.weight.detach()
    if h.is_contiguous() and ops.head_supported(h.shape[0] * h.shape[1], h.shape[2], w.shape[0]):
        # training windows (T_out = 1) and short sequences: one dedicated launch, bias included (csrc/vp3d_head.hip)
        return ops.head_fwd(h, w, mod.shrink.bias.detach())
    # N = 3*J_out (51) columns: weight rows beyond N come from the zero page, K is sliced (split-K) to fill the GPU
    return ops.conv_fwd(h, ops.pack_weight(mod.shrink.weight.detach()), mod._plan.shrink,
                        bias=mod.shrink.bias.detach())


def _forward_eval_f32(mod, x3: torch.Tensor) -> torch.Tensor:
    plan: StackPlan = mod._plan
    plan.lengths(x3.shape[1])
    packs = folded_weights(mod)
    wt, bias = packs[0]
    xin, spec0, _ = _expand_input(plan, x3)
    h = ops.conv_fwd(xin, wt, spec0, bias=bias, relu=True)
    del xin
    for i in range(plan.n_blocks):
        wt, bias = packs[1 + 2 * i]
        u = ops.conv_fwd(h, wt, plan.convs[1 + 2 * i], bias=bias, relu=True)
        wt, bias = packs[2 + 2 * i]
        h = ops.conv_fwd(u, wt, plan.convs[2 + 2 * i], bias=bias, relu=True, residual=(h, plan.res[i]))
        del u
    return _shrink(mod, h)


# --------------------------------------------------------------------------------------------------------
# training
# --------------------------------------------------------------------------------------------------------
class _Saved:
    __slots__ = ("x", "y", "coef", "drop", "wt", "kpad", "t_in")

    def __init__(self, x, y, coef, drop, wt, kpad, t_in):
        self.x, self.y, self.coef, self.drop, self.wt, self.kpad, self.t_in = x, y, coef, drop, wt, kpad, t_in


def _forward_train_f32(mod, x3: torch.Tensor, save: bool, frozen: bool = False):
    """Returns (out3, saved) ; saved is None unless `save`.  frozen: the differentiable EVAL forward (reference model.py:63-77
    with the module in .eval(): BatchNorm on its running statistics, no dropout, nothing updated) -- same kernels, the
    coefficients come from the running statistics instead of the batch's."""
    plan: StackPlan = mod._plan
    plan.lengths(x3.shape[1])
    convs, bns = _convs(mod), _bns(mod)
    p = 0.0 if frozen else float(mod.drop.p)
    seed, offset = mod._next_dropout_state() if p > 0 else (0, 0)
    if not frozen:
        mod._stats_epoch += 1        # running statistics are about to change (written by raw pointer)
    saved = []
    sync = None if frozen else mod.__dict__.get("_vp3d_sync_bn")         # dp.SyncBatchNorm: statistics over the global batch
    if sync is not None:
        sync.begin_step(x3.shape[0], x3.device)

    def layer(idx, h, residual=None):
        spec = plan.convs[idx]
        b, t_in, _ = h.shape
        kpad = 0
        if idx == 0:
            h, spec, kpad = _expand_input(plan, h)
        wt = ops.pack_weight(convs[idx].weight.detach(), ld_out=kpad or None)
        m_rows = b * spec.t_out(h.shape[1])
        if frozen:
            y = ops.conv_fwd(h, wt, spec)
            coef = ops.bn_eval_coef(bns[idx])
        else:
            stats = ops.stat_buffers(m_rows, spec.c_out, h.device)
            y = ops.conv_fwd(h, wt, spec, stats=stats)
            coef = ops.bn_finalize(bns[idx], m_rows, stats, sync=sync, momentum_dev=mod._momentum_dev_ptr())
        drop = ops.make_dropout(p, seed, offset, idx, mod._dropout_counter_ptr())
        a = ops.bn_act_fwd(y, coef, drop, residual)
        if save:
            saved.append(_Saved(h, y, coef, drop, wt, kpad, t_in))
        return a

    h = layer(0, x3)
    for i in range(plan.n_blocks):
        u = layer(1 + 2 * i, h)
        h = layer(2 + 2 * i, u, residual=(h, plan.res[i]))
    out = _shrink(mod, h)
    if not save:
        return out, None
    return out, dict(layers=saved, h_last=h, wts=ops.pack_weight(mod.shrink.weight.detach()), x3=x3, frozen=frozen)


_side_streams = {}


def _wgrad_stream(device, default_on: bool = False) -> Optional[torch.cuda.Stream]:
    """Second HIP stream for the weight-gradient GEMMs of backward: opt-in (VP3D_OVERLAP=1), never under bench.py's
    per-kernel event instrumentation.  Measured on MI355X (gpurun_out b9/b10, B=1024): forking wgrad next to its own
    dgrad is neutral (10.39 vs 10.34 ms/step: two fp32-MFMA GEMMs just split the matrix pipes), forking it under the
    next layer's HBM-bound BN-backward kernels costs 1.3 % (10.47 ms) -- the streaming kernels and the GEMM's
    L2-miss traffic contend for the fabric -- so the serial order stays the default."""
    if os.environ.get("VP3D_OVERLAP", "1" if default_on else "0") != "1" or ops._prof is not None:
        return None
    key = torch.device(device).index
    st = _side_streams.get(key)
    if st is None:
        # (same priority as the caller's stream: this runtime's range is (0, -1) -- there is no level BELOW the default that
        #  would let the second stream's workgroups only fill the slots the dependent chain leaves; round 6)
        st = torch.cuda.Stream(device=device)
        _side_streams[key] = st
    return st


# graph.py's piecewise capture: while set, the main -> second-stream hand-overs of backward END the graph being captured on
# one stream and BEGIN the next one on the other (instead of an event dependency inside ONE graph, which hipGraph replays
# serially on this runtime); the replay re-creates the dependencies with events between the graph launches
_segmenter = None


def fork_to_side(main, side, ev) -> None:
    """Everything queued on `main` so far happens before what is queued on `side` from here on."""
    if _segmenter is not None:
        _segmenter.to_side()
    else:
        ev.record(main)
        side.wait_event(ev)


def fork_back() -> None:
    """The second-stream work of this hand-over has been queued (pairs with fork_to_side)."""
    if _segmenter is not None:
        _segmenter.to_main()


def join_side(main, side) -> None:
    if _segmenter is None:
        main.wait_stream(side)


def join_side_now(main, side) -> None:
    """A join in the MIDDLE of a step: what is queued on `main` from here on waits for everything `side` has been given so far
    (the forward's weight packs, engine_s16.forward_train).  Piecewise graph capture: the piece being captured ends here and
    the replay waits for the second stream before it launches the next one."""
    if _segmenter is not None:
        _segmenter.join()
    else:
        main.wait_stream(side)


_fork_events = {}


def _fork_event(device, slot: int) -> torch.cuda.Event:
    """Reusable HIP event for the main -> wgrad-stream fork of layer `slot` (a fresh torch.cuda.Event() per layer and
    step is an hipEventCreate/Destroy pair each: host time that N ranks on one node's cores do not have)."""
    key = (torch.device(device).index, slot)
    ev = _fork_events.get(key)
    if ev is None:
        ev = _fork_events[key] = torch.cuda.Event()
    return ev


def _backward_train_f32(mod, saved, gout3: torch.Tensor, need_dx: bool):
    """Gradients in the order of ``param_list`` (+ optional input gradient).

    Optional second HIP stream (VP3D_OVERLAP=1, see _wgrad_stream): the dependent chain (BN/ReLU/dropout backward ->
    dgrad -> next layer) stays on the caller's stream and every wgrad GEMM (+ its split reduction), which has no
    consumer inside backward, is forked onto the second stream after its layer's dgrad has been queued."""
    plan: StackPlan = mod._plan
    main = torch.cuda.current_stream()
    side = _wgrad_stream(gout3.device)
    keep = []                                        # operands the side stream still reads: alive until the join
    L: List[_Saved] = saved["layers"]
    h_last = saved["h_last"]
    b, t_out, _ = h_last.shape
    gout3 = gout3.contiguous()
    g2 = gout3.view(b * t_out, -1)
    # Optional gradient sink (dp.FlatGradSync(direct_module=...)): every parameter gradient is written straight
    # into its view of the flat all-reduce buffer instead of being returned to autograd and accumulated into the
    # pre-existing .grad by one extra add kernel per tensor (29 launches per step for arc 3,3,3,3,3).
    frozen = bool(saved.get("frozen"))               # eval-mode forward: BatchNorm on running statistics (ops.bn_act_bwd)
    # (the differentiable eval-mode forward hands its gradients back to autograd: it must neither overwrite the flat
    # training gradients nor start the sink's bucket collectives)
    sink = None if frozen else mod.__dict__.get("_vp3d_grad_sink")
    convs, bns = _convs(mod), _bns(mod)

    def view(p):
        return sink.view_for(p) if sink is not None else None

    def sunk(value, out):
        return None if out is not None else value

    # Optional (_switches.SW["fuse_act_bwd"] = "1" wherever legal, "auto" on launches of >= 6 rounds of tiles): the dgrad epilogue that
    # produces an activation's incoming gradient also runs that activation's backward reduction (vp3d_act_bwd):
    # g = go*keep*[z>0] and the per-slab sums of g / g*xhat come out of the GEMM and the separate HBM pass of
    # vp3d_bn_bwd_reduce disappears.  Measured on MI355X (B=1024 step): the pass it removes (0.27 ms) comes back as
    # epilogue time of the compute-bound GEMM (+0.21 ms everywhere, +0.10 / -0.11 ms in auto mode): 10.19 / 10.17 vs
    # 10.18 ms -- no gain, so the separate kernels stay the default.
    fuse_mode = SW["fuse_act_bwd"]
    sync = mod.__dict__.get("_vp3d_sync_bn")
    if sync is not None or frozen:
        fuse_mode = "0"                              # the fused epilogue reduces with per-replica batch statistics
    if frozen:
        sync = None

    def upstream(idx):
        return (L[idx].y, L[idx].coef, L[idx].drop)

    def dgrad(dy, wt, spec, t_in, up_idx, residual=None, need_raw=True):
        """Returns (dx or None, FusedActBwd or None)."""
        ok = fuse_mode != "0" and up_idx is not None and (
            ops.can_fuse_act_bwd(spec, t_in) if fuse_mode == "1" else ops.should_fuse_act_bwd(spec, dy.shape[0], t_in))
        if ok:
            return ops.conv_dgrad(dy, wt, spec, t_in, residual=residual, act_bwd=upstream(up_idx), store_v=need_raw)
        return ops.conv_dgrad(dy, wt, spec, t_in, residual=residual), None

    o_b, o_w = view(mod.shrink.bias), view(mod.shrink.weight)
    last = 2 * plan.n_blocks if plan.n_blocks else 0             # layer whose activation is the stack output
    w_sh = mod.shrink.weight.detach()
    if fuse_mode == "0" and h_last.is_contiguous() and ops.head_supported(b * t_out, h_last.shape[2], w_sh.shape[0]):
        # the whole backward of the shrink conv as one launch + the fold of its weight / bias partials (csrc/vp3d_head.hip)
        dh, ws_h = ops.head_bwd(gout3, h_last, w_sh)
        d_sw, d_sb = ops.head_fold(ws_h, b * t_out, w_sh, out_dw=o_w, out_db=o_b)
        d_sw, d_sb, f_dh = sunk(d_sw, o_w), sunk(d_sb, o_b), None
    else:
        d_sb = sunk(ops.colsum(g2, out=o_b), o_b)
        d_sw = sunk(ops.conv_wgrad(gout3, h_last, plan.shrink, out=o_w), o_w)
        dh, f_dh = dgrad(gout3, saved["wts"], plan.shrink, t_out, last, need_raw=plan.n_blocks > 0)
    grads = [None] * (3 * len(L))
    n_done = [0]

    def group_done():
        # model.backward_param_groups() order: the sink may start exchanging a finished bucket right away; the
        # group's last gradients are produced on the wgrad stream, so that is the stream the collective must follow
        if sink is not None:
            if side is not None and n_done[0] > 0:
                with torch.cuda.stream(side):
                    sink.group_done(n_done[0])
            else:
                sink.group_done(n_done[0])
        n_done[0] += 1

    group_done()                                     # shrink

    def act_bwd(idx, go, fused):
        """BN/ReLU/dropout backward of layer idx on the caller's stream; returns dy and the deferred wgrad launch."""
        s = L[idx]
        o_g, o_bt = view(bns[idx].weight), view(bns[idx].bias)
        if o_g is None or o_bt is None:
            o_g = o_bt = None
        if fused is not None:
            dy, dgam, dbet = ops.bn_act_bwd_fused(fused, s.y, s.coef, out_dgamma=o_g, out_dbeta=o_bt)
        else:
            dy, dgam, dbet = ops.bn_act_bwd(go, s.y, s.coef, s.drop, out_dgamma=o_g, out_dbeta=o_bt, sync=sync, frozen=frozen)
        grads[3 * idx + 1] = sunk(dgam, o_g)
        grads[3 * idx + 2] = sunk(dbet, o_bt)

        def wgrad():
            # (optional second stream) forked AFTER this layer's dgrad has been queued on `main`
            out = view(convs[idx].weight)
            if side is not None:
                ev = _fork_event(gout3.device, idx)
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    dw = ops.conv_wgrad(dy, s.x, plan.convs[idx], rows_kpad=s.kpad, out=out)
                if out is None:
                    dw.record_stream(main)           # allocated on the side stream, consumed by autograd on `main`
                keep.append((dy, s.x, dw))
            else:
                dw = ops.conv_wgrad(dy, s.x, plan.convs[idx], rows_kpad=s.kpad, out=out)
            grads[3 * idx] = sunk(dw, out)

        return dy, wgrad

    for i in reversed(range(plan.n_blocks)):
        i1, i2 = 1 + 2 * i, 2 + 2 * i
        dy2, wgrad2 = act_bwd(i2, dh, f_dh)
        # gradient of u = act(i1): no residual joins it and nobody else reads it -> g only
        da1, f1 = dgrad(dy2, L[i2].wt, plan.convs[i2], L[i2].t_in, i1, need_raw=False)
        wgrad2()
        del dy2
        dy1, wgrad1 = act_bwd(i1, da1, f1)
        del da1, f1
        # gradient of the block input = output of block i-1 (activation of its second conv; layer 0 for i == 0);
        # block i-1 also needs it raw, as the residual of its own dgrad
        dh, f_dh = dgrad(dy1, L[i1].wt, plan.convs[i1], L[i1].t_in, 2 * i, residual=(dh, plan.res[i]), need_raw=i > 0)
        wgrad1()
        group_done()                                 # block i: both convs' gradients are written (or queued)
        del dy1
    dy0, wgrad0 = act_bwd(0, dh, f_dh)
    dx = None
    if need_dx:
        wt0 = L[0].wt if not L[0].kpad else ops.pack_weight(mod.expand_conv.weight.detach())
        dx = ops.conv_dgrad(dy0, wt0, plan.convs[0], L[0].t_in)
    wgrad0()
    group_done()                                     # expand
    if side is not None:
        main.wait_stream(side)                       # join: every gradient is visible to the caller's stream
        keep.clear()
    return grads + [d_sw, d_sb], dx


# Below this much forward conv work per call the step is bound by launch latency / the host, not by the matrix pipes, and the
# split-fp16 engine's extra producers / bound kernels cost more than its GEMMs save.  Re-derived in round 6 on MI355X
# (tools/s16_threshold.py, profiles/r06_s16_threshold.txt: both engines forced, run.py's own loop shape) -- the round-1 values
# (40 / 35 GFLOP) dated from a slower split-fp16 engine and sent run.py's own default configuration (-arc 3,3,3 -b 1024,
# reference common/arguments.py:37,45: 36.4 GFLOP) to the fp32 kernels:
#   training, f16x3 / f32 step time:  arc 3,3,3      B = 512 / 768 / 1024  (18 / 27 / 36 GFLOP)   1.08 / 0.96 / 0.99
#                                     arc 3,3,3,3    B = 128 / 192 / 256   (15 / 22 / 29 GFLOP)   1.14 / 0.99 / 0.79
#                                     arc 3,3,3,3,3  B =  64 / 128 / 192   (23 / 45 / 68 GFLOP)   1.19 / 0.98 / 0.65
#   (below ~20 GFLOP both engines sit on their host floors, 1.0 ms for ~60 launches against 1.45 ms for ~80; the three opt-in
#    edits of INTEGRATION.md 3b or a hipGraph replay move run.py's default to 1.08 / 0.89 ms against 1.43 on the fp32 kernels)
#   evaluation (B = 2, one sequence + mirrored copy): arc 3,3,3,3,3  T_out = 243 / 400 (27 / 38 GFLOP)  1.16 / 0.67,
#                                                      arc 3,3,3      T_out = 243 / 1000 (9 / 35 GFLOP)  1.05 / 0.76
# -> 25 GFLOP (training) / 30 GFLOP (evaluation).  VP3D_S16_MIN_GFLOP overrides both (0 = always).
_min_gf = os.environ.get("VP3D_S16_MIN_GFLOP")
S16_MIN_FORWARD_FLOPS = {True: float(_min_gf or 25.0) * 1e9, False: float(_min_gf or 30.0) * 1e9}   # [training]


def use_s16(mod, t_in: int, training: bool, need_dx: bool = False, batch: Optional[int] = None) -> bool:
    """True when this call runs on the split-fp16 GEMM path (module attribute ``math`` == "f16x3", see model.py), the
    configuration is one engine_s16 implements and the call is big enough to be compute-bound; everything else runs on
    the fp32-MFMA kernels."""
    if getattr(mod, "math", "f32") != "f16x3":
        return False
    from . import engine_s16
    if range_guard.tripped(mod):                     # intra-tensor dynamic range outside the S16 format's lossless window
        return False
    if not engine_s16.supported(mod, t_in, training, need_dx, batch=batch or 0):
        return False
    return batch is None or mod._plan.forward_flops(batch, t_in) >= S16_MIN_FORWARD_FLOPS[bool(training)]


# Which GEMM engine served the calls of this process: {"s16_eval", "f32_eval", "s16_train", "f32_train"} -> count.  The
# parity suite asserts on it (a test that claims to cover the split-fp16 kernels must have executed them).
ENGINE_CALLS = collections.Counter()


def forward_eval(mod, x3: torch.Tensor) -> torch.Tensor:
    range_guard.tick(mod, False, x3)
    if use_s16(mod, x3.shape[1], False, batch=x3.shape[0]):
        from . import engine_s16
        ENGINE_CALLS["s16_eval"] += 1
        return engine_s16.forward_eval(mod, x3)
    ENGINE_CALLS["f32_eval"] += 1
    return _forward_eval_f32(mod, x3)


def forward_train(mod, x3: torch.Tensor, save: bool, need_dx: bool = False):
    """Returns (out3, saved); saved is None unless `save` (it records which arithmetic produced it)."""
    range_guard.tick(mod, True, x3)
    if use_s16(mod, x3.shape[1], True, need_dx, batch=x3.shape[0]):
        from . import engine_s16
        ENGINE_CALLS["s16_train"] += 1
        out, saved = engine_s16.forward_train(mod, x3, save, need_dx)
        if saved is not None:
            saved["s16"] = True
        return out, saved
    ENGINE_CALLS["f32_train"] += 1
    return _forward_train_f32(mod, x3, save)


def backward_train(mod, saved, gout3: torch.Tensor, need_dx: bool):
    range_guard.measure_head_gradient(mod, gout3)     # (armed by the forward of a measured step only; either engine: an
    if saved.get("s16"):                             #  input / head-gradient trip is re-evaluated from the exact-fp32 engine too)
        from . import engine_s16
        return engine_s16.backward_train(mod, saved, gout3, need_dx)
    return _backward_train_f32(mod, saved, gout3, need_dx)


class FrozenStackFn(torch.autograd.Function):
    """The differentiable eval-mode forward (module in .eval(), autograd on): the exact-fp32 kernels with the BatchNorm
    coefficients taken from the running statistics; the backward is the training backward without the batch-statistic terms."""

    @staticmethod
    def forward(ctx, mod, x3, *params):
        ENGINE_CALLS["f32_eval_grad"] += 1
        out, saved = _forward_train_f32(mod, x3, save=True, frozen=True)
        ctx.mod = mod
        ctx.saved = saved
        ctx.need_dx = x3.requires_grad
        return out

    @staticmethod
    def backward(ctx, gout):
        if ctx.saved is None:
            raise RuntimeError("vp3d: backward called twice on the same graph (activations were freed)")
        grads, dx = _backward_train_f32(ctx.mod, ctx.saved, gout, ctx.need_dx)
        ctx.saved = None
        return (None, dx) + tuple(grads)


class TemporalStackFn(torch.autograd.Function):
    """Whole-stack training step: forward saves raw conv outputs + BN coefficients; backward is hand-written."""

    @staticmethod
    def forward(ctx, mod, x3, *params):
        out, saved = forward_train(mod, x3, save=True, need_dx=x3.requires_grad)
        ctx.mod = mod
        ctx.saved = saved
        ctx.need_dx = x3.requires_grad
        return out

    @staticmethod
    def backward(ctx, gout):
        if ctx.saved is None:
            raise RuntimeError("vp3d: backward called twice on the same graph (activations were freed)")
        grads, dx = backward_train(ctx.mod, ctx.saved, gout, ctx.need_dx)
        ctx.saved = None
        return (None, dx) + tuple(grads)
