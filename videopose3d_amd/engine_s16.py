"""The temporal stack on the split-fp16 ("S16") GEMM path: every 1024-channel contraction runs on
v_mfma_f32_32x32x16_f16 with fp32-class results (see csrc/vp3d_s16.h), ~3x the fp32-MFMA rate.

Same wiring as ``engine`` (reference common/model.py:126-138 / :187-197); what changes is the form of the GEMM
operands.  GEMMs read S16, write fp32; the streaming kernels between them read fp32 and write S16:

  training forward   y = conv(a_prev; Wt)            S16 x S16 -> fp32 y + BatchNorm slab statistics
                     a = [res +] drop(relu(bn(y)))   vp3d_bn_act_fwd_s16: S16 rows (next conv, residual) and the
                                                     transposed S16 copy the next conv's wgrad reduces over
  training backward  dy = bn/relu/drop backward      vp3d_bn_bwd_apply_s16: S16 rows (dgrad) + transposed (wgrad)
                     da = dy @ Wd  (+ residual)      NT GEMM with the transposed weight pack -> fp32 + its max
                     dW = dy^T @ a_prev^T            NT GEMM over the transposed copies, K = rows, split-K partials
  eval forward       h = relu(conv(a; W*scale)+shift) [+ res]  -> fp32 + max, then vp3d_split_rows -> S16

Per-tensor exponents come from guaranteed bounds computed on the device (vp3d_act_bound / vp3d_dy_bound: Samuelson's
inequality on the batch statistics; vp3d_amax or the GEMM epilogue's amax where no bound exists), never from the host.
The 3*J-column shrink conv and its gradients stay on the fp32-MFMA kernels (0.03 % of the FLOPs).

Supported: channels % 64 == 0 and filter widths <= 3; training additionally needs the strided model on windows that
tile exactly (T == receptive field, what run.py trains on).  ``supported()`` tells; everything else runs in fp32.
"""
from __future__ import annotations

import os
from typing import List

import torch

from . import engine, ops, ops_s16 as S, range_guard
from ._switches import SW
from ._lib import RowMap
from .plan import ConvSpec, StackPlan


MAX_TRAIN_TAPS = 8      # widest conv the training path packs / gathers (dense=True models with pad >= 4 train on the fp32 kernels)


GATHER_T_MAX_ROWS = 65535 * 64      # vp3d_gather_t_s16: one grid row per 64-row tile, at most 65535 of them


def supported(mod, t_in: int, training: bool, need_dx: bool = False, batch: int = 0) -> bool:
    """What the split-fp16 engine implements: channel counts that are multiples of 64 and an expand conv of at most 128
    input columns (taps * J_in * in_features; 17 x 2 x 3 = 102).  Training covers both classes (dilated / strided), causal
    or not, any window length and filter widths up to 8 taps per conv -- a dense=True model with wider kernels (2*pad + 1
    taps: its weight-gradient operand would be taps times the activation) trains on the fp32 kernels.  Input gradients are
    produced by the fp32 kernels for the expand layer only."""
    plan: StackPlan = mod._plan
    c = plan.convs[0].c_out
    k0 = plan.convs[0].c_in * plan.convs[0].taps
    if c % 64 != 0 or k0 > 128:
        return False
    if not training:
        return k0 >= 32                              # (eval stages the expand conv through ops.padded_k rows: none below 32 columns)
    if max(spec.taps for spec in plan.convs) > MAX_TRAIN_TAPS:
        return False
    if batch:
        # convs whose weight-gradient operand is gathered in backward (dilated class, windows that do not tile): the gather
        # kernel's row limit -- a larger call trains on the fp32 kernels instead of failing in the middle of backward
        t_len = plan.lengths(t_in)
        for idx in range(1, len(plan.convs)):
            sp = plan.convs[idx]
            t_i = t_in if idx == 0 else t_len[(idx - 1) // 2]
            if not (sp.taps == 1 or _tiles(sp, t_i)) and batch * sp.t_out(t_i) > GATHER_T_MAX_ROWS:
                return False
    return True


def _tiles(spec: ConvSpec, t_in: int) -> bool:
    """The conv's windows are disjoint and cover its input exactly (what run.py trains the strided class on): the conv is
    a reshape GEMM, and the transposed weight-gradient operand is a plain re-layout the forward producer can write."""
    return spec.dil == 1 and spec.stride == spec.taps and spec.taps * spec.t_out(t_in) == t_in


# --------------------------------------------------------------------------------------------------------
# eval
# --------------------------------------------------------------------------------------------------------
def folded_weights(mod):
    """[(S16 folded weight pack, shift, l1)] per layer, cached until a parameter / buffer changes;
    l1 = device [max_n sum_k |W[n][k]|, max_n |shift[n]|]: with the measured maximum of a layer's input it bounds the
    layer's output (|y| <= l1[0] * max|x| + l1[1]), which is what lets the GEMM epilogue write S16 directly."""
    key = engine._eval_key(mod)
    cache = mod.__dict__.get("_fold_cache_s16")
    if cache is not None and cache[0] == key:
        return cache[1]
    packs = []
    for wt, shift in engine.folded_weights(mod):     # fp32 fold (cached) -> S16; weight-only work, once per weight version
        l1 = torch.stack([wt.abs().sum(dim=1).max(), shift.abs().max()]).to(torch.float32).contiguous()
        packs.append((S.split(wt), shift, l1))
    mod.__dict__["_fold_cache_s16"] = (key, packs)
    return packs


def forward_eval(mod, x3: torch.Tensor) -> torch.Tensor:
    """Eval forward with the activations kept in S16 between the GEMMs: every epilogue applies bias / ReLU / residual,
    measures the true maximum (for the next layer's bound) and writes S16 rows with a guaranteed one-layer bound; only the
    input of the 3*J-column shrink conv is written as fp32."""
    plan: StackPlan = mod._plan
    plan.lengths(x3.shape[1])
    packs = folded_weights(mod)
    dev = x3.device
    amax = S.new_bounds(1 + 2 * plan.n_blocks, dev)
    xin, spec0, _ = engine._expand_input(plan, x3)
    xs = S.split(xin)                                 # measured bound: doubles as the input's amax
    del xin
    wt, bias, l1 = packs[0]
    hs = S.conv_nt(xs, wt, spec0, bias=bias, relu=True, amax_out=amax[0], s16_out=(xs.bound, l1, None))
    h_amax = amax[0]
    del xs
    h = None
    for i in range(plan.n_blocks):
        last = i + 1 == plan.n_blocks
        wt, bias, l1 = packs[1 + 2 * i]
        us = S.conv_nt(hs, wt, plan.convs[1 + 2 * i], bias=bias, relu=True, amax_out=amax[1 + 2 * i],
                       s16_out=(h_amax, l1, None))
        wt, bias, l1 = packs[2 + 2 * i]
        nxt = S.conv_nt(us, wt, plan.convs[2 + 2 * i], bias=bias, relu=True, residual=(hs, plan.res[i]),
                        amax_out=None if last else amax[2 + 2 * i],
                        s16_out=None if last else (amax[1 + 2 * i], l1, h_amax))
        del us
        if last:
            h = nxt
        else:
            hs, h_amax = nxt, amax[2 + 2 * i]
    if h is None:                                     # no blocks (single filter width): shrink reads the expand output
        h = S.join(hs)
    return engine._shrink(mod, h)


# --------------------------------------------------------------------------------------------------------
# training (strided model)
# --------------------------------------------------------------------------------------------------------
class _Saved:
    __slots__ = ("x_t", "y", "coef", "drop", "wd", "t_in", "kpad", "bits", "x_rows", "one_col", "w_packed", "wform", "xin_f32", "gram_fwd")

    def __init__(self, x_t, y, coef, drop, wd, t_in, kpad, bits, x_rows=None):
        self.x_t, self.y, self.coef, self.drop, self.wd, self.t_in, self.kpad = x_t, y, coef, drop, wd, t_in, kpad
        self.one_col, self.w_packed = -1, None   # expand layer: bias column of the im2row rows, fp32 weight pack (shortcut)
        self.wform = "tcopy"      # weight-gradient form: "rows" (x_rows), "tcopy" (x_t from the producer), "gather" (x_rows ->
        self.xin_f32 = None       # vp3d_gather_t_s16 in backward); expand layer with an input gradient: fp32 im2row rows
        self.gram_fwd = None      # expand layer: the forward's centred second-moment matrix of X (float64), for its backward
        self.x_rows = x_rows      # rows-form wgrad (wgrad_from_rows): the layer input as S16 rows instead of x_t
        self.bits = bits          # activation bits ([bn(y) > 0 and kept], 1 bit / element): what backward reads instead
                                  # of regenerating the Philox mask


def wgrad_from_rows(c_out: int, c_in: int, in_rows: int = 0) -> bool:
    """Default since round 2 (_switches.SW["wgrad_rows"] = False restores the transposed-copy form): the C x C weight gradients read the
    S16 rows of dy and of the layer input (vp3d_wgrad_rows_s16, channels % 256 == 0) and the producers write no
    transposed copies for them; the expand conv (K-padded im2row operand) and narrower models keep the transposed form."""
    # in_rows: rows of the conv's input (B * T_in): vp3d_wgrad_rows_s16 addresses both operands with 32-bit byte offsets
    return (SW["wgrad_rows"] and S.wgrad_rows_supported(c_out, c_in) and
            in_rows * max(c_in, c_out) * 4 < 2 ** 31)


def next_is_not_tcopy(plan: StackPlan, t_in: int, batch: int = 0) -> bool:
    """The conv after the expand layer takes its weight-gradient operand from the rows (directly or gathered), i.e. the
    expand layer's producer has no transposed copy to write (the fused GEMM epilogue writes rows only)."""
    if len(plan.convs) < 2:
        return True
    sp = plan.convs[1]
    t1 = plan.convs[0].t_out(t_in)
    tiling = sp.taps == 1 or _tiles(sp, t1)
    return (not tiling) or wgrad_from_rows(sp.c_out, sp.c_in, batch * t1)


def expand_kpad(spec: ConvSpec) -> int:
    """Row width of the expand conv's im2row operand in TRAINING: taps * C_in rounded up to 64 columns (the S16 transposing
    producers work on 64-column tiles: 17 joints -> 102 -> 128, 15 joints -> 90 -> 128, 5 joints -> 30 -> 64)."""
    k = spec.taps * spec.c_in                        # (the expand conv always has dilation 1: its taps are adjacent rows)
    return (k + 63) // 64 * 64 if spec.dil == 1 else 0


def expand_shortcut_column(plan: StackPlan, sync) -> int:
    """Padding column of the expand conv's im2row rows that carries the constant 1 of the no-dy backward of the expand
    layer (expand_bwd below), or -1 when that backward is not used: no spare padding column, synchronised BatchNorm
    (the formulas need the global sums between the two reductions)."""
    spec = plan.convs[0]
    kpad, kv = expand_kpad(spec), spec.taps * spec.c_in
    if sync is not None or not kpad or kv >= kpad or kpad > 128:
        return -1
    return kv


# BatchNorm-backward column sums inside the dgrad launch that writes the activation's incoming gradient (vp3d_s16_red):
# VP3D_FUSE_BN_RED=0 never, =1 wherever the launch supports it, default: where it pays -- activations of at least
# FUSE_BN_RED_MIN_ROWS rows.  Round 4, interleaved in RANDOMISED order on one box (tools/red_ab.py, profiles/r04_bn_red_ab.txt):
# never 4.364 ms, >= 16,384 rows 4.291, >= 8,192 rows 4.301, >= 4,096 rows 4.302, everywhere 4.304 -- the 27,648-row activation
# pays (-1.7 %), the 9,216-row ones cost 10 us more inside their dgrad launch than their reduction pass did beside the
# second stream's weight-gradient GEMM, and below that the fused launch's hand-over (~14 us of dependent memory round trips at
# its tail) costs what the 17-30 us pass it replaces does (tools/red_bench.py)
FUSE_BN_RED_DEFAULT = "auto"
FUSE_BN_RED_MIN_ROWS = 16384


def _packs(ws, specs, bounds, want_dgrad: bool):
    """[(S16 forward pack, S16 dgrad pack or None)] for the C x C convs.  One launch for the stack when every conv is a
    strided one of <= 3 taps (the benchmark configuration); else per layer, with the dgrad pack in the form the conv's
    data gradient reads: [(k, ci)][co] for the reshape GEMM of a strided conv, [ci][k * C_out + co] for the gather form."""
    if all(sp.stride == sp.taps and sp.dil == 1 and sp.taps <= 3 for sp in specs):
        return S.pack_weights_multi(ws, bounds, want_dgrad=want_dgrad)
    out = []
    for i, (w, sp) in enumerate(zip(ws, specs)):
        gather = sp.stride == 1
        if sp.taps <= 3:
            out.append(S.pack_weight(w, bounds[i], want_dgrad=want_dgrad, dilated_form=gather))
            continue
        # wider filters (rare: -arc 3,5,3 ...): re-layout with torch, split with the row kernel
        c_out, c_in, taps = w.shape
        wf = S.split(ops.pack_weight(w), bounds[i])
        wd = None
        if want_dgrad:
            wd = S.split(w.permute(1, 2, 0).reshape(c_in, taps * c_out).contiguous() if gather
                         else w.permute(2, 1, 0).reshape(taps * c_in, c_out).contiguous(), bounds[i])
        out.append((wf, wd))
    return out


def forward_train(mod, x3: torch.Tensor, save: bool, need_dx: bool = False):
    plan: StackPlan = mod._plan
    plan.lengths(x3.shape[1])
    convs, bns = engine._convs(mod), engine._bns(mod)
    p = float(mod.drop.p)
    seed, offset = mod._next_dropout_state() if p > 0 else (0, 0)
    mod._stats_epoch += 1
    dev = x3.device
    n_layers = len(plan.convs)
    # [0,n): activations, [n,2n): weights, 2n: input; with `save` also the backward's 2n bounds ([0,n): go of layer i, [n,2n): dy
    # of layer i) -- one zero-fill launch in the forward instead of a second one at the head of the backward's dependent chain
    bounds_all = S.new_bounds(2 * n_layers + 1 + (2 * n_layers if save else 0), dev)
    bounds = bounds_all[:2 * n_layers + 1]
    saved: List[_Saved] = []
    b = x3.shape[0]
    sync = mod.__dict__.get("_vp3d_sync_bn")         # dp.SyncBatchNorm: statistics (and their bounds) over the global batch
    if sync is not None:
        sync.begin_step(b, dev)

    # expand conv: im2row staging (fp32, 128-wide rows) -> S16 rows + transposed copy
    use_bits = save and SW["act_bits"]
    one_col = expand_shortcut_column(plan, sync) if (use_bits and not need_dx) else -1
    fuse_expand = (SW["expand_fused"] and sync is None and not need_dx and
                   (one_col >= 0 or not save) and next_is_not_tcopy(plan, x3.shape[1], x3.shape[0]))
    kpad = expand_kpad(plan.convs[0])
    assert kpad, "the S16 path stages the expand conv through im2row"
    t_in0 = x3.shape[1]
    rows0 = x3.shape[0] * plan.convs[0].t_out(t_in0)
    one_pass_input = not (save and need_dx) and SW["expand_kernel"] and rows0 <= 65535 * 64
    # The whole prologue as TWO launches (S.prologue_a: every maximum + the activation bounds; S.prologue_b: input staging +
    # all weight packs) instead of seven dependent ones: the benchmark configuration's shape of the stack (strided convs of
    # <= 3 taps) with the one-pass input staging, per-replica BatchNorm; everything else keeps the separate launches below
    fused_prologue = (one_pass_input and SW["prologue_fused"] and sync is None and n_layers > 1 and
                      n_layers <= 15 and kpad <= 128 and
                      all(sp.stride == sp.taps and sp.dil == 1 and sp.taps <= 3 for sp in plan.convs[1:]) and
                      all(c_.weight.shape[:2] == convs[1].weight.shape[:2] for c_ in convs[1:]))
    if fused_prologue:
        spec0 = ConvSpec(kpad, plan.convs[0].c_out, 1, 1, 1)
        xin_f32 = None
    elif one_pass_input:
        # one pass: maximum over the raw input (and the bias column's 1), then im2row + S16 split fused -- the 128-wide fp32
        # staging rows are never written
        spec0 = ConvSpec(kpad, plan.convs[0].c_out, 1, 1, 1)
        xb = S.amax(x3, out=bounds[2 * n_layers], floor=1.0 if one_col >= 0 else 0.0)
        x_rows, x_t = S.im2row_split(x3, plan.convs[0], kpad, one_col, xb, want_t=save)
        m0 = x_rows.data.shape[0] * x_rows.data.shape[1]
        xin_f32 = None
    else:
        xin, spec0 = ops.im2row(x3, plan.convs[0], kpad, one_col), ConvSpec(kpad, plan.convs[0].c_out, 1, 1, 1)
        m0 = xin.shape[0] * xin.shape[1]
        xb = S.amax(xin, out=bounds[2 * n_layers])
        x_rows, x_t = S.split_t(xin.view(m0, kpad), xb, want_rows=True, want_t=save)
        x_rows = S.S16(x_rows.data.view(xin.shape), xb)
        xin_f32 = xin if (save and need_dx) else None     # the expand layer's input gradient runs on the fp32 kernels
        del xin

    # weight-gradient form of every C x C conv: from the S16 rows (k_tn_s16), from the transposed copy its producer
    # writes (strided conv whose windows tile its input), or from a copy gathered in backward (everything else)
    t_len = plan.lengths(t_in0)
    t_in_of = [t_in0] + [t_len[(idx - 1) // 2] for idx in range(1, n_layers)]      # input length of conv idx
    wform = ["tcopy"] * n_layers
    for idx in range(1, n_layers):
        sp = plan.convs[idx]
        tiling = sp.taps == 1 or _tiles(sp, t_in_of[idx])
        wform[idx] = ("rows" if wgrad_from_rows(sp.c_out, sp.c_in, b * t_in_of[idx]) else "tcopy") if tiling else "gather"

    def next_taps(idx):
        """taps of the conv that consumes the activation of layer idx when its wgrad reduces over the producer-written
        transposed copy; 0: no consumer, or one that reads the rows (directly, or gathered in backward)."""
        return plan.convs[idx + 1].taps if idx + 1 < n_layers and wform[idx + 1] == "tcopy" else 0

    # per-step prologue for ALL layers: weight maxima -> S16 weight packs; activation bounds
    ws = [c.weight.detach() for c in convs]
    m_all = [b * t_len[0]] + [b * t_len[(idx + 1) // 2] for idx in range(1, n_layers)]
    res_from = [idx - 2 if (idx >= 2 and idx % 2 == 0) else -1 for idx in range(n_layers)]
    packs_pending = None   # (main stream, second stream) while the C x C weight packs are still running on the second stream
    if fused_prologue:
        xb = bounds[2 * n_layers]
        S.prologue_a([x3] + ws, [xb] + [bounds[n_layers + i] for i in range(n_layers)],
                     [1.0 if one_col >= 0 else 0.0] + [0.0] * n_layers, bns, m_all, res_from, p, bounds)
        # The C x C weight packs (204 MB of traffic: 68 MB of weights read, forward + dgrad packs written) depend on the maxima
        # only, and nothing before the first C x C conv reads them: optionally (SW["prologue_overlap"], default off) they run
        # on the second stream beside the input staging and the expand layer's statistics, the main stream joining in front of
        # conv 1.  Built and measured in round 6: bit-identical, and worth nothing (the pack launch occupies every CU slot;
        # the statistics kernel beside it runs 2-3x longer; first C x C GEMM at +279 us either way) -- what did help was the pack
        # kernel itself (16-byte loads, four in flight: 83 -> 54 us).
        side = engine._wgrad_stream(dev, default_on=True) if SW["prologue_overlap"] else None
        if side is not None and n_layers > 1:
            main = torch.cuda.current_stream()

            class _OnSide:
                def __enter__(self_):
                    engine.fork_to_side(main, side, engine._fork_event(dev, -3))
                    self_.ctx = torch.cuda.stream(side)
                    self_.ctx.__enter__()

                def __exit__(self_, *exc):
                    self_.ctx.__exit__(*exc)
                    engine.fork_back()
                    return False

            packs_cc = S.pack_weights_multi(ws[1:], bounds[n_layers + 1:], want_dgrad=save, launch_ctx=_OnSide)
            x_rows, x_t, w0_packed, w0_s16, _ = S.prologue_b(x3, plan.convs[0], kpad, one_col, xb, save, ws[0], bounds[n_layers],
                                                             [], None, save)
            packs_pending = (main, side)
        else:
            x_rows, x_t, w0_packed, w0_s16, packs_cc = S.prologue_b(x3, plan.convs[0], kpad, one_col, xb, save, ws[0],
                                                                    bounds[n_layers], ws[1:], bounds[n_layers + 1:], save)
        packs = [(w0_s16, None)] + packs_cc
        m0 = x_rows.data.shape[0] * x_rows.data.shape[1]
    else:
        S.amax_multi(ws, bounds[n_layers:])
        w0_packed = ops.pack_weight(ws[0], ld_out=kpad)
        packs = [(S.split(w0_packed, bounds[n_layers]), None)]
        if n_layers > 1:
            packs += _packs(ws[1:], plan.convs[1:], bounds[n_layers + 1:], save)
        S.act_bounds_multi(bns, m_all if sync is None else [sync.rows_total(m_) for m_ in m_all], res_from, p, bounds)
    c_all = [plan.convs[idx].c_out for idx in range(n_layers)]
    bits_all = S.new_act_bits(sum(m_ * c_ for m_, c_ in zip(m_all, c_all)) // c_all[0], c_all[0], dev) if use_bits else None
    bits_at = 0

    h_prev = None          # S16 block input (residual source)
    a = x_rows
    a_t = x_t
    for idx in range(n_layers):
        spec = spec0 if idx == 0 else plan.convs[idx]
        if idx >= 1 and packs_pending is not None:     # the first consumer of the C x C packs: wait for the second stream
            engine.join_side_now(*packs_pending)
            packs_pending = None
        wf, wd = packs[idx]
        t_cur = a.data.shape[1]
        m_rows = b * spec.t_out(t_cur)
        assert m_rows == m_all[idx]
        # 224 x 256 tiles (tile configuration 28) where the 256-row tiling strands most of its last round: their statistics
        # come in 32-row slabs; not for the launches that carry a fused activation, nor under synchronised BatchNorm
        mix = idx > 0 and sync is None
        slab = (S.stat_slab_rows(*S.plan(m_rows, spec.c_out, spec.taps * spec.c_in, mix=True, a_numel=a.data.numel(),
                                             b_numel=wf.data.numel())) if mix else 64)
        stats = ops.stat_buffers(m_rows, spec.c_out, dev, slab)
        # expand layer, fused: its GEMM (K = 128) is cheap enough to run twice -- pass 1 only produces the BatchNorm
        # statistics, pass 2 applies BatchNorm + ReLU + dropout in its epilogue and writes the S16 activation (+ bits):
        # the conv output never goes to HBM (its backward, expand_bwd, needs no y either)
        fused0 = idx == 0 and fuse_expand and n_layers > 1
        dedicated0 = (fused0 and SW["expand_kernel"] and
                      m_rows * kpad * 4 < 2 ** 31)                                     # vp3d_expand_fwd_s16 (32-bit byte offsets)
        # expand layer statistics from the centred second-moment matrix of its 128-column input (S.expand_stats_gram: one MFMA
        # pass over the transposed copy + per-channel quadratic forms in fp64) instead of a statistics-only pass over the conv
        # output: needs the transposed copy (save) and the constant-1 padding column (one_col)
        gram0 = (dedicated0 and a_t is not None and one_col >= 0 and bns[0].momentum is not None and
                 a_t.data.numel() * 4 < 2 ** 31 and not range_guard.gram_disabled(mod))
        if gram0:
            y = None
            coef, gram_fwd = S.expand_stats_gram(a_t, w0_packed, bns[0], m_rows, plan.convs[0].taps * plan.convs[0].c_in, one_col,
                                                 mod._momentum_dev_ptr(), want_gram=True, illcond=range_guard.gram_flag(mod))
        else:
            coef = None
            if dedicated0:
                y = S.expand_fwd(a, wf, stats=stats)
            elif idx > 0 and sync is None and SW["fin_in_finish"]:
                # K-sliced launches (the M <= 3072 layers): the finishing pass finalises the statistics itself (vp3d_s16_fin)
                y, coef = S.conv_nt(a, wf, spec, stats=stats, mix=mix, stat_slab=slab, fin=(bns[idx], mod._momentum_dev_ptr()))
            else:
                y = S.conv_nt(a, wf, spec, stats=stats, no_output=fused0, mix=mix, stat_slab=slab)
            if coef is None:
                coef = ops.bn_finalize(bns[idx], m_rows, stats, sync=sync, momentum_dev=mod._momentum_dev_ptr(), slab_rows=slab)
        drop = ops.make_dropout(p, seed, offset, idx, mod._dropout_counter_ptr())
        residual = None
        if idx >= 2 and idx % 2 == 0:
            residual = (h_prev, plan.res[idx // 2 - 1])
        bits = None
        if save:
            if use_bits:
                bits = bits_all[bits_at:bits_at + m_rows * spec.c_out // 8]
                bits_at += bits.numel()
            saved.append(_Saved(a_t, y, coef, drop, wd, t_in0 if idx == 0 else t_cur, kpad if idx == 0 else 0, bits,
                                x_rows=a if wform[idx] != "tcopy" else None))
            saved[-1].wform = wform[idx]
            if idx == 0:
                saved[0].xin_f32 = xin_f32
            if idx == 0 and one_col >= 0:
                saved[0].one_col, saved[0].w_packed = one_col, w0_packed
                # (the centred second-moment matrix of X: the layer's backward rebuilds X^T X from it instead of forming its own)
                saved[0].gram_fwd = gram_fwd if gram0 else None
                if S.expand_rows_form(spec.c_out, kpad):
                    saved[0].x_rows = a                  # P = G^T X reads the rows; the transposed copy only feeds X^T X
        if fused0:
            act0 = (coef, drop, bounds[idx], bits)
            a, a_t = (S.expand_fwd(a, wf, act=act0) if dedicated0 else S.conv_nt(a, wf, spec, act=act0)), None
        elif idx == n_layers - 1:            # the stack output also in fp32: the 3*J-column shrink conv runs on the fp32 path
            a, a_t, h_last = S.bn_act_fwd(y, coef, drop, residual, bounds[idx], t_taps=0, want_f32=True, act_bits=bits)
        else:
            a, a_t = S.bn_act_fwd(y, coef, drop, residual, bounds[idx], t_taps=next_taps(idx) if save else 0,
                                  act_bits=bits)
        if idx % 2 == 0:
            h_prev = a
    if packs_pending is not None:
        engine.join_side_now(*packs_pending)
        packs_pending = None
    out = engine._shrink(mod, h_last)
    if not save:
        return out, None
    return out, dict(layers=saved, h_last=h_last, wts=ops.pack_weight(mod.shrink.weight.detach()), bounds=bounds, bwd_bounds=bounds_all[2 * n_layers + 1:])


def backward_train(mod, saved, gout3: torch.Tensor, need_dx: bool):
    plan: StackPlan = mod._plan
    L: List[_Saved] = saved["layers"]
    h_last = saved["h_last"]
    b, t_out, _ = h_last.shape
    dev = gout3.device
    gout3 = gout3.contiguous()
    g2 = gout3.view(b * t_out, -1)
    p = float(mod.drop.p)
    sink = mod.__dict__.get("_vp3d_grad_sink")
    convs, bns = engine._convs(mod), engine._bns(mod)
    n_layers = len(L)
    bounds = saved.pop("bwd_bounds", None)         # [0,n): go of layer i, [n,2n): dy of layer i (zeroed by the forward)
    if bounds is None or bounds.shape[0] != 2 * n_layers:
        bounds = S.new_bounds(2 * n_layers, dev)   # (a second backward through the same graph is refused upstream; be safe)

    def view(prm):
        return sink.view_for(prm) if sink is not None else None

    def sunk(value, out):
        return None if out is not None else value

    main = torch.cuda.current_stream()
    # Weight-gradient GEMMs on a second HIP stream (default on, VP3D_OVERLAP=0 disables): measured on MI355X 5.59 -> 5.38
    # ms / step -- unlike the fp32 engine (neutral to -1.3 %), the split-fp16 GEMM leaves HBM bandwidth and LDS room for
    # the next layer's streaming BatchNorm-backward kernels to make progress beside it.
    side = engine._wgrad_stream(dev, default_on=True)
    keep = []
    grads = [None] * (3 * n_layers)
    n_done = [0]

    def group_done(on_side=True):
        # the bucket's all-reduce follows the stream that produced the group's last gradients
        if sink is not None:
            if side is not None and on_side and engine._segmenter is None:
                with torch.cuda.stream(side):
                    sink.group_done(n_done[0])
            else:
                sink.group_done(n_done[0])
        n_done[0] += 1

    # X^T X of the expand layer's input (its no-dy backward, below): a small GEMM with nothing upstream -> second stream
    gram_xx, gram_ev = None, None
    # (with the dedicated kernel X^T X rides along in the P = G^T X launch at the end of backward: no GEMM of its own)
    # (vp3d_expand_bwd_p_s16 addresses go and the transposed X with 32-bit byte offsets: M * C * 4 and kpad * ld_t * 4 < 2 GiB)
    fused_p = (L[0].one_col >= 0 and L[0].x_rows is None and SW["expand_kernel"] and
               L[0].bits is not None and L[0].bits.numel() * 32 < 2 ** 31 and
               L[0].x_t is not None and L[0].x_t.data.numel() * 4 < 2 ** 31)
    if L[0].one_col >= 0 and not fused_p:
        if side is not None and engine._segmenter is None:        # (a mid-backward join: not cut into graph pieces)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                gram_xx = S.gram(L[0].x_t)
                gram_xx.record_stream(main)
                gram_ev = engine._fork_event(dev, -1)
                gram_ev.record(side)
        else:
            gram_xx = S.gram(L[0].x_t)

    o_b, o_w = view(mod.shrink.bias), view(mod.shrink.weight)
    last = n_layers - 1
    w_sh = mod.shrink.weight.detach()
    use_head = h_last.is_contiguous() and ops.head_supported(b * t_out, h_last.shape[2], w_sh.shape[0])
    if use_head:
        # The shrink conv's whole backward as ONE launch on the dependent chain (dh, its maximum = the bound of the first
        # BatchNorm backward's operand, and the row-sliced partials of dW / dbias: csrc/vp3d_head.hip) + the fold of the partials
        # beside it -- was dgrad + amax on the chain and colsum + wgrad + slice reduction on the second stream (round 6)
        dh, ws_h = ops.head_bwd(gout3, h_last, w_sh, dh_bound=bounds[last])
        if side is not None:
            engine.fork_to_side(main, side, engine._fork_event(dev, -2))
            with torch.cuda.stream(side):
                d_sw, d_sb = ops.head_fold(ws_h, b * t_out, w_sh, out_dw=o_w, out_db=o_b)
            engine.fork_back()
            if o_w is None:
                d_sw.record_stream(main)             # (allocated on the second stream, handed to autograd on `main`)
            if o_b is None:
                d_sb.record_stream(main)
            keep.append((gout3, h_last, ws_h))
        else:
            d_sw, d_sb = ops.head_fold(ws_h, b * t_out, w_sh, out_dw=o_w, out_db=o_b)
        d_sw, d_sb = sunk(d_sw, o_w), sunk(d_sb, o_b)
    elif side is not None:        # nothing in backward reads the shrink gradients: off the dependent chain as well (-0.7 %)
        engine.fork_to_side(main, side, engine._fork_event(dev, -2))
        with torch.cuda.stream(side):
            d_sb = sunk(ops.colsum(g2, out=o_b), o_b)
            d_sw = sunk(ops.conv_wgrad(gout3, h_last, plan.shrink, out=o_w), o_w)
        engine.fork_back()
        for t_ in (d_sb, d_sw):
            if t_ is not None:
                t_.record_stream(main)
        keep.append((gout3, h_last))
    else:
        d_sb = sunk(ops.colsum(g2, out=o_b), o_b)
        d_sw = sunk(ops.conv_wgrad(gout3, h_last, plan.shrink, out=o_w), o_w)
    if not use_head:
        dh = ops.conv_dgrad(gout3, saved["wts"], plan.shrink, t_out)
        S.amax(dh, out=bounds[last])
    group_done()                                     # shrink

    # BatchNorm-backward column sums inside the dgrad launch that writes the activation's incoming gradient (vp3d_s16_red;
    # VP3D_FUSE_BN_RED=0 keeps the separate reduction pass everywhere): presummed[idx] = (dgamma, dbeta) once done
    fuse_mode = os.environ.get("VP3D_FUSE_BN_RED", FUSE_BN_RED_DEFAULT)
    fuse_red = fuse_mode != "0" and mod.__dict__.get("_vp3d_sync_bn") is None
    presummed = {}

    def red_for(up, m, n, k, dy):
        """vp3d_s16_red of layer `up` for the dgrad launch [m, n, k] that writes its incoming gradient, or None."""
        s = L[up]
        if (not fuse_red or up <= 0 or s.bits is None or
                (fuse_mode != "1" and s.y.shape[0] * s.y.shape[1] < FUSE_BN_RED_MIN_ROWS) or
                not S.red_supported(m, n, k, s.y.shape[2]) or
                s.y.numel() * 4 >= 2 ** 31 or dy.data.numel() * 4 >= 2 ** 31 or n * k * 4 >= 2 ** 31):
            return None
        o_g, o_bt = view(bns[up].weight), view(bns[up].bias)
        if o_g is None or o_bt is None:
            dgb = torch.empty((2, s.y.shape[2]), dtype=torch.float32, device=dev)
            o_g, o_bt = dgb[0], dgb[1]
        r, hold = S.make_red(s.y, s.coef, s.bits, p if s.drop is not None else 0.0, m, n, o_g, o_bt, bounds[n_layers + up])
        presummed[up] = (o_g, o_bt)
        keep.append(hold)                            # (the partial rows: alive until the launch is enqueued)
        return r

    def act_bwd(idx, go):
        s = L[idx]
        o_g, o_bt = view(bns[idx].weight), view(bns[idx].bias)
        if o_g is None or o_bt is None:
            o_g = o_bt = None
        dy, dy_t, dgam, dbet = S.bn_act_bwd(go, bounds[idx], s.y, s.coef, s.drop, p, bounds[n_layers + idx],
                                            out_dgamma=o_g, out_dbeta=o_bt, want_rows=idx > 0,   # expand: no dgrad
                                            sync=mod.__dict__.get("_vp3d_sync_bn"), act_bits=s.bits,
                                            want_t=s.wform != "rows", presummed=presummed.get(idx))
        if s.wform == "rows":
            dy_t = dy                                # rows-form wgrad reads the rows
        grads[3 * idx + 1] = sunk(dgam, o_g)
        grads[3 * idx + 2] = sunk(dbet, o_bt)
        return dy, dy_t


    def wgrad(idx, dy_t, on_side=True):
        spec = plan.convs[idx]
        out = view(convs[idx].weight)
        n_cols = L[idx].kpad if L[idx].kpad else spec.taps * spec.c_in
        m_rows = L[idx].y.shape[0] * L[idx].y.shape[1]

        def gemm():
            if L[idx].wform == "rows":
                return S.wgrad_rows(dy_t, L[idx].x_rows, spec.c_out, spec.c_in, spec.taps, out=out)
            x_t = L[idx].x_t
            if L[idx].wform == "gather":             # dilated / ragged windows / wide filters: build the operand now
                x_t = S.gather_t(L[idx].x_rows, spec, L[idx].y.shape[1])
            return S.wgrad(dy_t, x_t, spec.c_out, spec.c_in, spec.taps, n_cols, out=out, flops_rows=m_rows)
        if side is not None and on_side:
            # nothing inside backward consumes dW: it runs beside the next layer's HBM-bound BatchNorm-backward kernels
            # (piecewise graph capture: every hand-over is worth its graph boundary -- 4.55 ms with all of them, 4.61 with the
            # four large layers' only, 4.67 with two, B = 1024)
            engine.fork_to_side(main, side, engine._fork_event(dev, idx))
            with torch.cuda.stream(side):
                dw = gemm()
            engine.fork_back()
            if out is None:
                dw.record_stream(main)
            keep.append((dy_t, L[idx].x_t, L[idx].x_rows, dw))
        else:
            dw = gemm()
        grads[3 * idx] = sunk(dw, out)

    def dgrad(idx, dy, residual, amax_out, up=-1):
        """dx of conv idx.  Strided conv (stride == taps; also every 1x1 conv): windows do not overlap, dx viewed as
        [B*T_out, taps*C_in] = dy @ Wd is a plain GEMM (rows of a ragged tail stay zero).  Stride-1 conv of several taps
        (the dilated class): gather form dx[b,s] = sum_k dy[b, s - k*dil] @ W_k^T over the [ci][k*C_out + co] pack.
        up: the layer whose activation dx is the gradient of (its BatchNorm-backward sums may ride in the launch: red_for)."""
        spec: ConvSpec = plan.convs[idx]
        bb, t_o, c_out = dy.data.shape
        taps, c_in = spec.taps, spec.c_in
        t_i = L[idx].t_in
        if spec.stride == 1 and taps > 1:
            dx = torch.empty((bb, t_i, c_in), dtype=torch.float32, device=dev)
            rm = RowMap(bb, t_i, t_o, 1, -spec.dil, 0, taps)
            e = None
            if residual is not None:
                r, rs = residual                      # dx[b, s] += r[b, s - start]
                assert rs.step == 1 and r.shape[0] == bb and r.shape[2] == c_in
                e = ops._epi(residual=(r, 1, -rs.start, 0), n_cols=c_in)
            S.gemm_rows(dy, L[idx].wd, rm, c_out, c_out, c_in, dx, t_i * c_in, c_in, epi=e, amax_out=amax_out,
                        family="tconv_dgrad", red=red_for(up, bb * t_i, c_in, taps * c_out, dy), mix=True)
            return dx
        assert spec.stride == taps and (spec.dil == 1 or taps == 1) and taps * t_o <= t_i      # (a 1-tap conv has no dilation to speak of)
        dx = (torch.empty if taps * t_o == t_i else torch.zeros)((bb, t_i, c_in), dtype=torch.float32, device=dev)
        rm = RowMap(bb, t_o, t_o, 1, 0, 0, 1)
        e = None
        if residual is not None:
            r, rs = residual                          # dx[b, start + step*t] += r[b, t],  step == taps
            assert rs.step == taps and r.shape == (bb, t_o, c_in)
            e = ops._epi(residual=(r, 1, 0, rs.start * c_in), n_cols=taps * c_in)
        S.gemm_rows(dy, L[idx].wd, rm, c_out, c_out, taps * c_in, dx, t_i * c_in, taps * c_in, epi=e, amax_out=amax_out,
                    family="tconv_dgrad", red=red_for(up, bb * t_o, taps * c_in, c_out, dy), mix=True)
        return dx

    n_head = plan.n_blocks
    for i in reversed(range(n_head)):
        i1, i2 = 1 + 2 * i, 2 + 2 * i
        dy2, dy2_t = act_bwd(i2, dh)
        da1 = dgrad(i2, dy2, None, bounds[i1], up=i1)
        wgrad(i2, dy2_t)
        del dy2, dy2_t
        dy1, dy1_t = act_bwd(i1, da1)
        del da1
        dh = dgrad(i1, dy1, (dh, plan.res[i]), bounds[2 * i], up=2 * i)
        wgrad(i1, dy1_t)
        group_done()
        del dy1, dy1_t
    dx_in = None
    if need_dx:
        # input gradient (never asked for by run.py): the expand layer's backward on the fp32 kernels -- its conv reads
        # 102 columns, so the cost is the two streaming passes either way
        s0 = L[0]
        o_g, o_bt, o_w = view(bns[0].weight), view(bns[0].bias), view(convs[0].weight)
        if o_g is None or o_bt is None:
            o_g = o_bt = None
        dy0, dg0, db0 = ops.bn_act_bwd(dh, s0.y, s0.coef, s0.drop, out_dgamma=o_g, out_dbeta=o_bt,
                                       sync=mod.__dict__.get("_vp3d_sync_bn"))
        dw0 = ops.conv_wgrad(dy0, s0.xin_f32, plan.convs[0], rows_kpad=s0.kpad, out=o_w)
        dx_in = ops.conv_dgrad(dy0, ops.pack_weight(convs[0].weight.detach()), plan.convs[0], s0.t_in)
        grads[0], grads[1], grads[2] = sunk(dw0, o_w), sunk(dg0, o_g), sunk(db0, o_bt)
    elif L[0].one_col >= 0:
        # expand layer without dy: G = dh * keep * [z > 0] (one pass over dh), P = G^T X, then dgamma / dbeta / dW from P,
        # X^T X and the weights (vp3d_expand_bwd_s16) -- no reduce / finalize / apply passes over (dh, y) for this layer
        spec0 = plan.convs[0]
        rows0 = L[0].x_rows is not None
        p0 = p if L[0].drop is not None else 0.0
        g0, part0 = None, None
        gram_c = None
        if fused_p and L[0].gram_fwd is not None:
            # X^T X from the forward's centred second-moment matrix (in the last launch, fp64): no ride-along MFMAs in the
            # P launch, no vp3d_sum_slices behind it
            ws0, n0 = S.expand_p_from_go(dh, bounds[0], L[0].bits, p0, L[0].x_t, want_gram=False)
            part0, gram_xx, gram_c = (ws0, n0), L[0].gram_fwd, L[0].x_t
        elif fused_p:                                # vp3d_expand_bwd_p_s16: G = go * keep * bits is never stored
            ws0, n0, gram_xx = S.expand_p_from_go(dh, bounds[0], L[0].bits, p0, L[0].x_t, want_gram=True)
            part0 = (ws0, n0)
        else:
            g0 = S.act_mask(dh, bounds[0], L[0].bits, p0, transposed=not rows0)
        if gram_ev is not None:
            main.wait_event(gram_ev)
        o_w, o_g, o_bt = view(convs[0].weight), view(bns[0].weight), view(bns[0].bias)
        if o_g is None or o_bt is None:
            o_g = o_bt = None
        m0 = dh.shape[0] * dh.shape[1]
        dw0, dg0, db0 = S.expand_bwd(g0, L[0].x_rows if rows0 else L[0].x_t, gram_xx, L[0].w_packed, L[0].coef, m0, spec0.c_in,
                                     spec0.taps, L[0].one_col, rows0, out_dw=o_w, out_dgamma=o_g, out_dbeta=o_bt, partials=part0,
                                     gram_centred=gram_c)
        grads[0], grads[1], grads[2] = sunk(dw0, o_w), sunk(dg0, o_g), sunk(db0, o_bt)
    else:
        dy0, dy0_t = act_bwd(0, dh)
        # the last weight gradient (expand conv) on the MAIN stream: it runs beside the tail of the first block's wgrad
        # GEMM (still on the second stream) instead of queueing behind it (-0.4 %)
        wgrad(0, dy0_t, on_side=False)
    if side is not None:
        engine.join_side(main, side)
        keep.clear()
    group_done(on_side=False)
    return grads + [d_sw, d_sb], dx_in
