"""GPU tests of the split-fp16 ("S16") kernels through the C ABI: the NT GEMM (tile configurations, split-K + finishing
pass, ragged shapes, epilogues, BatchNorm slab statistics, amax), the S16 producers (row split, transposed copies,
weight packs, activation forward / backward) and the device-side bounds -- against fp64 references built with torch on
the CPU/GPU.  Tolerance of the GEMM: |err| <= 1e-6 * sum|a||b| (22+ bit operands; the fp32-MFMA kernel sits at ~1e-7)."""
import pytest
import torch

import videopose3d_amd as V
from videopose3d_amd import ops, ops_s16 as S
from videopose3d_amd._switches import SW
from videopose3d_amd.plan import ConvSpec, ResSpec
from tests.util import unpack_act_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GEMM_TOL = 1e-6


@pytest.fixture(autouse=True)
def _no_size_threshold():
    from videopose3d_amd import engine
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    yield
    engine.S16_MIN_FORWARD_FLOPS.update(keep)


def _ref_conv(x, w, spec, bias=None):
    xd, wd = x.double().permute(0, 2, 1), w.double()
    y = torch.nn.functional.conv1d(xd, wd, None if bias is None else bias.double(), dilation=spec.dil, stride=spec.stride)
    den = torch.nn.functional.conv1d(xd.abs(), wd.abs(), dilation=spec.dil, stride=spec.stride)
    return y.permute(0, 2, 1), den.permute(0, 2, 1)


EXPERIMENT_CFGS = {10, 13, 21, 23, 24, 25, 26, 30, 120, 122}


def need_cfg(cfg):
    """Tile configurations the planner never picks are built only with -DVP3D_BUILD_EXPERIMENTS (include/vp3d.h:
    vp3d_has_experiments); the default library rejects them -- their tests then have nothing to certify."""
    from videopose3d_amd import _lib
    if cfg in EXPERIMENT_CFGS and not _lib.lib().vp3d_has_experiments():
        pytest.skip("tile configuration %d: library built without VP3D_BUILD_EXPERIMENTS" % cfg)


CASES = [  # (B, T, spec, cfg, splits)
    (8, 27, ConvSpec(256, 256, 3, 3, 1), 0, 1),
    (8, 27, ConvSpec(256, 256, 3, 3, 1), 4, 1),
    (8, 27, ConvSpec(256, 256, 3, 3, 1), 20, 1),
    (8, 27, ConvSpec(256, 256, 3, 3, 1), 22, 1),
    (8, 27, ConvSpec(256, 256, 3, 3, 1), 13, 1),
    (8, 27, ConvSpec(256, 256, 3, 3, 1), 26, 1),        # 256x256 by four waves of 128x128 (accumulators in AGPRs)
    (5, 27, ConvSpec(160, 96, 3, 1, 3), 26, 3),         # ... ragged N, split-K
    (11, 67, ConvSpec(64, 2368, 1), 26, 1),
    (5, 27, ConvSpec(160, 96, 3, 1, 3), 0, 1),          # N = 96 (ragged column tile), strided
    (5, 27, ConvSpec(160, 96, 3, 1, 3), 22, 3),         # split-K + finishing pass
    (5, 27, ConvSpec(160, 96, 3, 1, 3), 21, 1),         # register-pipelined fragments, ragged N
    (3, 31, ConvSpec(64, 200, 1), 0, 2),                # ragged M = 93, N = 200
    (7, 40, ConvSpec(128, 128, 3, 9, 1), -1, 0),        # planned
    (300, 27, ConvSpec(64, 300, 3, 1, 3), 30, 1),       # hybrid: 256x256 tiles on whole rounds + 128x128 on the rest (M = 2700)
    (1200, 81, ConvSpec(64, 512, 3, 1, 3), 30, 1),      # hybrid with a real split (M = 32400: 254 tiles + tail rows)
    (2, 300, ConvSpec(64, 64, 5, 1, 1), 0, 1),          # 5 adjacent taps ("dense"-style)
    (11, 67, ConvSpec(64, 1344, 1), 20, 1),             # 11 column tiles: a full block of 8 + a ragged block of 3 (tile order)
    (11, 67, ConvSpec(64, 2368, 1), 22, 1),             # 10 column tiles of 256 (8 + 2), 3 row tiles: 30 tiles over 8 XCDs
    (11, 67, ConvSpec(192, 2368, 1), 22, 2),            # ... with split-K (positions of a split are a multiple of 8)
    (8, 27, ConvSpec(256, 256, 3, 3, 1), 28, 1),        # 224 x 256 tiles (wave rows of 4 + 3 row blocks): one ragged tile
    (31, 81, ConvSpec(64, 512, 3, 1, 3), 28, 1),        # ... M = 837 = 3 full tiles + 165 rows (5 blocks + 5 rows), strided
    (11, 67, ConvSpec(64, 2368, 1), 28, 1),             # ... ragged N (10 column tiles of 256, the last one 64 wide)
    (7, 40, ConvSpec(128, 256, 3, 9, 1), 28, 1),        # ... dilated taps
    (8, 27, ConvSpec(256, 256, 3, 3, 1), 29, 1),        # 160 x 256 tiles (wave rows of 3 + 2 row blocks): one tile + a ragged one
    (31, 81, ConvSpec(64, 512, 3, 1, 3), 29, 1),        # ... M = 837 = 5 full tiles + 37 rows, strided
    (11, 67, ConvSpec(64, 2368, 1), 29, 1),             # ... ragged N
    (31, 81, ConvSpec(64, 512, 3, 1, 3), 29, 3),        # ... with K slices + finishing pass (64-row slabs again)
    (31, 81, ConvSpec(64, 512, 3, 1, 3), 28, 2),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_T%d_%dx%d_k%d_d%d_s%d_cfg%d_sp%d" % (
    c[0], c[1], c[2].c_in, c[2].c_out, c[2].taps, c[2].dil, c[2].stride, c[3], c[4]))
def test_nt_gemm_vs_fp64(case):
    b, t, spec, cfg, splits = case
    need_cfg(cfg)
    g = torch.Generator().manual_seed(3)
    x = (torch.relu(torch.randn(b, t, spec.c_in, generator=g)) * 1.3).to(DEV)
    w = ((torch.rand(spec.c_out, spec.c_in, spec.taps, generator=g) * 2 - 1) * 0.03).to(DEV)
    bias = torch.randn(spec.c_out, generator=g).to(DEV)
    ref, den = _ref_conv(x, w, spec, bias)
    ref = torch.relu(ref)
    xs, ws = S.split(x), S.split(ops.pack_weight(w))
    m = b * spec.t_out(t)
    slab = S.stat_slab_rows(cfg, splits)                # 64 rows; the 224- / 160-row tilings write 32-row slabs in one K slice
    stats = ops.stat_buffers(m, spec.c_out, DEV, slab)
    am = S.new_bound(DEV)
    y = S.conv_nt(xs, ws, spec, bias=bias, relu=True, stats=stats, amax_out=am, cfg=cfg, splits=splits, stat_slab=slab)
    assert float(((y.double() - ref).abs() / den).max()) < GEMM_TOL
    assert float(am.max()) == float(y.abs().max())
    # slab statistics are those of the raw conv output (before bias / ReLU)
    raw = (_ref_conv(x, w, spec)[0]).reshape(m, spec.c_out)
    assert stats[0].shape[0] == (m + slab - 1) // slab
    for s0 in range(0, m, slab):
        blk = raw[s0:s0 + slab]
        assert torch.allclose(stats[0][s0 // slab].double(), blk.sum(0), rtol=1e-4, atol=1e-4)
        assert torch.allclose(stats[1][s0 // slab].double(), ((blk - blk.mean(0)) ** 2).sum(0), rtol=1e-3, atol=1e-4)
    if cfg in (28, 29) and splits == 1:
        # the same K order per element as the 256-row tiling: bit-identical output; a statistics buffer sized for the
        # other slab size is refused, not overrun
        y22 = S.conv_nt(xs, ws, spec, bias=bias, relu=True, cfg=22, splits=1)
        assert torch.equal(y, y22)
        from videopose3d_amd._lib import Vp3dError
        with pytest.raises(Vp3dError, match="slabs"):
            S.conv_nt(xs, ws, spec, stats=ops.stat_buffers(m, spec.c_out, DEV), cfg=cfg, splits=1)


def test_nt_gemm_exponents_and_residual():
    """Operands far outside fp16's range (1e-6 gradients x 1e4 'weights'): the per-tensor exponents keep full precision;
    residual rows are added after the scaling."""
    g = torch.Generator().manual_seed(5)
    b, t, c = 6, 9, 128
    spec = ConvSpec(c, c, 1)
    x = (torch.randn(b, t, c, generator=g) * 1e-6).to(DEV)
    w = (torch.randn(c, c, 1, generator=g) * 1e4).to(DEV)
    r = torch.randn(b, 2 * t + 1, c, generator=g).to(DEV)
    ref, den = _ref_conv(x, w, spec)
    ref = ref + r[:, 1::2][:, :t].double()
    y = S.conv_nt(S.split(x), S.split(ops.pack_weight(w)), spec, residual=(r, ResSpec(1, 2)))
    assert float(((y.double() - ref).abs() / den).max()) < GEMM_TOL
    # without exponents the same data underflows fp16: the bound is what makes the format safe
    y0 = S.conv_nt(S.split(x, measure=False), S.split(ops.pack_weight(w) * 1e-4, measure=False), spec)
    assert float(((y0.double() * 1e4 - (ref - r[:, 1::2][:, :t].double())).abs() / den).max()) > 1e-3


@pytest.mark.parametrize("cfg", [20, 22, 28, 29])
@pytest.mark.parametrize("bb,t_o,with_res", [(5, 37, True), (3, 100, True), (5, 37, False)])
def test_dgrad_epilogue_residual_rows_as_a_batch(cfg, bb, t_o, with_res):
    """The dgrad-form launch of the strided conv (dx[b, 3 t + tap] = dy[b, t] @ Wd_tap, the skip connection's gradient joins
    at the centre tap: reference common/model.py:191-196 backward) in every tiling: the fp32 epilogue fetches a 32-row block's
    residual rows and row-table entries in one batch -- ragged row counts (185 / 300 rows against 128- / 160- / 224- / 256-row
    tiles), three column tiles of which one carries the residual, amax of what was stored; all tilings agree bit for bit (same K
    order per element)."""
    from videopose3d_amd._lib import RowMap
    g = torch.Generator().manual_seed(41)
    c, taps = 256, 3
    dy = (torch.randn(bb, t_o, c, generator=g) * 0.3).to(DEV)
    wd = (torch.randn(taps * c, c, generator=g) * 0.05).to(DEV)
    r = torch.randn(bb, t_o, c, generator=g).to(DEV)
    ref = (dy.double().reshape(-1, c) @ wd.double().t()).reshape(bb, t_o, taps * c)
    den = (dy.double().abs().reshape(-1, c) @ wd.double().abs().t()).reshape(bb, t_o, taps * c) + 1e-30
    if with_res:
        ref[:, :, c:2 * c] += r.double()
    rm = RowMap(bb, t_o, t_o, 1, 0, 0, 1)
    e = ops._epi(residual=(r, 1, 0, (taps // 2) * c), n_cols=taps * c) if with_res else None
    outs = {}
    for cf in (cfg, 22):
        dx = torch.full((bb, taps * t_o, c), float("nan"), dtype=torch.float32, device=DEV)
        am = S.new_bound(DEV)
        S.gemm_rows(S.split(dy), S.split(wd), rm, c, c, taps * c, dx, taps * t_o * c, taps * c, epi=e, amax_out=am, cfg=cf,
                    splits=1, family="tconv_dgrad", mix=True)
        got = dx.view(bb, t_o, taps * c)
        assert bool(torch.isfinite(got).all())
        err = (got.double() - ref).abs() - (r.double().abs().repeat(1, 1, taps) * 2.0 ** -22 if with_res else 0.0)
        assert float((err / den).max()) < GEMM_TOL
        assert float(am.max()) == float(got.abs().max())
        outs[cf] = got.clone()
    assert torch.equal(outs[cfg], outs[22])


def test_split_join_and_transposed_copy():
    g = torch.Generator().manual_seed(7)
    m, c = 200, 128                                     # ragged: 200 rows -> transposed pitch 256, zero padded
    v = (torch.randn(m, c, generator=g) * torch.exp(torch.randn(m, c, generator=g) * 3)).to(DEV)
    bd = S.amax(v)
    assert float(bd.max()) == float(v.abs().max())
    rows, tt = S.split_t(v, bd)
    back = S.join(rows)
    assert float(((back - v).abs() / float(v.abs().max())).max()) < 2.0 ** -22        # abs error vs the tensor bound
    big = v.abs() > float(v.abs().max()) * 2.0 ** -10
    assert float(((back - v).abs() / v.abs())[big].max()) < 2.0 ** -21                 # 22+ significant bits
    assert tt.data.shape == (c, 256)
    t_back = S.join(tt)
    assert torch.equal(t_back[:, :m], back.t().contiguous())
    assert float(t_back[:, m:].abs().max()) == 0.0


def test_weight_packs_match_reference_layout():
    g = torch.Generator().manual_seed(9)
    w = (torch.randn(128, 64, 3, generator=g) * 0.05).to(DEV)
    bd = S.amax(w)
    wf, wd = S.pack_weight(w, bd)
    f = S.join(wf)                                      # [co][k*C_in + ci]
    d = S.join(wd)                                      # [(k*C_in + ci)][co]
    ref = w.permute(0, 2, 1).reshape(128, 3 * 64)
    assert float((f - ref).abs().max()) < float(w.abs().max()) * 2.0 ** -21
    assert torch.equal(d, f.t().contiguous())
    _, wdd = S.pack_weight(w, bd, want_fwd=False, dilated_form=True)     # [ci][k*C_out + co]
    dd = S.join(wdd)
    assert torch.equal(dd, S.join(wf).view(128, 3, 64).permute(2, 1, 0).reshape(64, 3 * 128))


@pytest.mark.parametrize("taps,p,use_bits,c", [(1, 0.0, False, 128), (3, 0.25, False, 128), (1, 0.0, True, 128),
                                               (3, 0.25, True, 128), (3, 0.25, True, 192), (1, 0.5, True, 1024)])
def test_activation_forward_and_backward_producers(taps, p, use_bits, c):
    """vp3d_bn_act_fwd_s16 / vp3d_bn_bwd_apply_s16 against the fp32 streaming kernels (same Philox mask), including the
    transposed copies in the layout the strided conv's wgrad reduces over; with use_bits the backward reads the forward's
    activation bits (vp3d_bn_bwd_reduce_bits + the BITS apply kernel) instead of regenerating mask and ReLU predicate."""
    g = torch.Generator().manual_seed(11)
    b, t = 6, 9
    y = torch.randn(b, t, c, generator=g).to(DEV)
    coef = torch.stack([1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g),
                        0.05 * torch.randn(c, generator=g), 1 + 0.1 * torch.rand(c, generator=g)]).to(DEV)
    res = torch.randn(b, 3 * t, c, generator=g).to(DEV)
    rs = ResSpec(1, 3)
    drop = ops.make_dropout(p, 1234, 5, 2)
    a32 = ops.bn_act_fwd(y, coef, drop, (res, rs))
    bd = S.new_bound(DEV)
    bd[0] = float(a32.abs().max()) * 1.5
    m = b * t
    bits = S.new_act_bits(m, c, DEV) if use_bits else None
    a, a_t = S.bn_act_fwd(y, coef, drop, (S.split(res), rs), bd, t_taps=taps, act_bits=bits)
    assert float((S.join(a) - a32).abs().max()) < float(bd.max()) * 2.0 ** -20
    if use_bits:       # bit e of byte ((c/64)*M + m)*8 + (c%64)/8  ==  [bn(y) > 0 and kept]
        z = y.double() * coef[0].double() + coef[1].double()
        keep = ops.dropout_mask(m * c, drop, DEV).view(b, t, c) != 0 if drop is not None else torch.ones_like(z, dtype=torch.bool)
        want = ((z > 0) & keep).view(m, c // 64, 8, 8)                      # [m][tile][byte][bit]
        sure = (z.abs() > 1e-6).view(m, c // 64, 8, 8)                      # (fp32 fma rounding may flip a z ~ 0)
        got = bits.view(c // 64, m, 8)
        unpacked = torch.stack([(got >> e) & 1 for e in range(8)], dim=-1).bool()      # [tile][m][byte][bit]
        assert torch.equal(unpacked.permute(1, 0, 2, 3) & sure, want & sure)
    at = S.join(a_t)
    assert at.shape == (taps * c, S.t_pitch(m, taps))
    expect = S.join(a).reshape(m // taps, taps, c).permute(1, 2, 0).reshape(taps * c, m // taps)
    assert torch.equal(at[:, :m // taps], expect)
    assert float(at[:, m // taps:].abs().max()) == 0.0
    # backward
    go = (torch.randn(b, t, c, generator=g) * 1e-5).to(DEV)
    dy32, dg32, db32 = ops.bn_act_bwd(go, y, coef, drop)
    gb = S.amax(go)
    dyb = S.new_bound(DEV)
    dy, dy_t, dg, db = S.bn_act_bwd(go, gb, y, coef, drop, p, dyb, act_bits=bits)
    assert float(dyb.max()) >= float(dy32.abs().max())              # the Samuelson bound is a bound
    if use_bits:                                                    # other summation order than the fp32 pass
        assert torch.allclose(dg, dg32, rtol=1e-5, atol=1e-9) and torch.allclose(db, db32, rtol=1e-5, atol=1e-9)
    else:
        assert torch.equal(dg, dg32) and torch.equal(db, db32)
    assert float((S.join(dy) - dy32).abs().max()) < float(dyb.max()) * 2.0 ** -20
    assert torch.equal(S.join(dy_t)[:, :m], S.join(dy).reshape(m, c).t().contiguous())


def test_act_bound_is_a_bound_for_adversarial_statistics():
    """Samuelson's inequality is tight for a one-hot column: the normalised value reaches sqrt(M-1)."""
    m, c = 257, 64
    y = torch.zeros(1, m, c, device=DEV)
    y[0, 0, :] = 1.0                                    # one outlier row
    bn = torch.nn.BatchNorm1d(c).to(DEV)
    stats = ops.stat_buffers(m, c, DEV)
    spec = ConvSpec(64, c, 1)
    x = torch.zeros(1, m, 64, device=DEV)
    x[0, 0, :] = 1.0
    w = torch.eye(c, 64, device=DEV).reshape(c, 64, 1).contiguous()
    yy = S.conv_nt(S.split(x), S.split(ops.pack_weight(w)), spec, stats=stats)
    assert torch.equal(yy, y)
    coef = ops.bn_finalize(bn, m, stats)
    bd = S.new_bound(DEV)
    S.act_bound(bn, m, 0.0, None, bd)
    a32 = ops.bn_act_fwd(y, coef, None, None)
    assert float(a32.abs().max()) <= float(bd.max())
    assert float(a32.abs().max()) > 0.99 * float(bd.max()) - 1e-3      # and it is attained
    a, _ = S.bn_act_fwd(y, coef, None, None, bd)
    assert torch.isfinite(S.join(a)).all()
    assert float((S.join(a) - a32).abs().max()) < float(bd.max()) * 2.0 ** -20


@pytest.mark.parametrize("joints,c", [(17, 128), (15, 64), (10, 256), (5, 64)])
def test_model_level_f16x3_equals_f32_path(joints, c):
    """Whole training step and eval forward in both arithmetics on the same weights / dropout stream; the joint counts give
    expand rows of 102 -> 128, 90 -> 96, 60 -> 64 and 30 -> 32 columns (the dedicated expand kernels with partly filled /
    fewer k-steps, the one-pass input staging only for multiples of 64)."""
    import copy
    from videopose3d_amd import engine
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    torch.manual_seed(0)
    fw = [3, 3, 3]
    V.set_default_math("f32")
    try:
        m32 = V.TemporalModelOptimized1f(joints, 2, 17, fw, dropout=0.25, channels=c).to(DEV).train()
    finally:
        V.set_default_math(None)
    m16 = copy.deepcopy(m32)
    m16.math = "f16x3"
    for m in (m32, m16):
        m._drop_seed, m._drop_calls = 99, 0
    x = (torch.randn(16, 27, joints, 2, device=DEV) * 0.5).clamp(-1, 1)
    tgt = torch.randn(16, 1, 17, 3, device=DEV) * 0.3
    outs = []
    n16 = engine.ENGINE_CALLS["s16_train"]
    for m in (m32, m16):
        yv = m(x)
        torch.mean(torch.norm(yv - tgt, dim=3)).backward()
        outs.append(yv.detach())
    assert engine.ENGINE_CALLS["s16_train"] == n16 + 1          # the second model really ran the split-fp16 engine
    assert float(torch.mean(torch.norm(outs[0] - outs[1], dim=3))) < 1e-5
    for (k, a), (_, q) in zip(m16.named_parameters(), m32.named_parameters()):
        assert float((a.grad - q.grad).abs().max() / (q.grad.abs().max() + 1e-30)) < 5e-5, k
    for (k, a), (_, q) in zip(m16.named_buffers(), m32.named_buffers()):
        if a.dtype.is_floating_point:
            assert torch.allclose(a, q, rtol=1e-5, atol=1e-6), k
    # eval forward (folded BatchNorm) on the dilated class with the same weights
    e32 = V.TemporalModel(joints, 2, 17, fw, channels=c).to(DEV)
    e32.load_state_dict(m32.state_dict())
    e16 = copy.deepcopy(e32)
    e32.math, e16.math = "f32", "f16x3"
    xe = (torch.randn(4, 40, joints, 2, device=DEV) * 0.5).clamp(-1, 1)
    n16 = engine.ENGINE_CALLS["s16_eval"]
    with torch.no_grad():
        ya, yb = e32.eval()(xe), e16.eval()(xe)
    assert engine.ENGINE_CALLS["s16_eval"] == n16 + (1 if joints * 2 * 3 >= 32 else 0)     # (30 input columns: eval stays on fp32)
    assert float(torch.mean(torch.norm(ya - yb, dim=3))) < 1e-5


def test_unsupported_configurations_fall_back_to_fp32_kernels():
    from videopose3d_amd import engine_s16
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=48)        # channels % 64 != 0
    assert m.math == V.default_math() and not engine_s16.supported(m, 27, True)
    m = V.TemporalModel(17, 2, 17, [3, 3, 3], channels=128)                   # the dilated class trains on S16 as well
    assert engine_s16.supported(m, 27, True) and engine_s16.supported(m, 27, False)
    m = V.TemporalModel(17, 2, 17, [3, 3, 3], channels=128, dense=True)       # ... unless its dense kernels are > 8 taps wide
    assert not engine_s16.supported(m, 27, True) and engine_s16.supported(m, 27, False)
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=128)
    assert engine_s16.supported(m, 27, True) and engine_s16.supported(m, 28, True)
    assert engine_s16.supported(m, 27, True, need_dx=True)
    # and calls too small to be compute-bound stay on the fp32 kernels (launch-latency regime)
    from videopose3d_amd import engine
    engine.S16_MIN_FORWARD_FLOPS.update({True: 60e9, False: 35e9})
    m.math = "f16x3"
    assert not engine.use_s16(m, 27, True, batch=1024)                       # arc 3,3,3, B = 1024: 36 GFLOP forward
    big = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], channels=1024)
    big.math = "f16x3"
    assert engine.use_s16(big, 243, True, batch=1024) and not engine.use_s16(big, 243, True, batch=64)


@pytest.mark.parametrize("cfg", [20, 22, 26])
def test_s16_output_and_s16_residual_chain(cfg):
    """Eval chaining: the epilogue writes S16 rows under the device-evaluated one-layer bound l1[0]*amax(in)+l1[1]+amax(res),
    reads an S16 residual, and still measures the true maximum."""
    need_cfg(cfg)
    g = torch.Generator().manual_seed(13)
    b, t, c = 5, 40, 128
    spec = ConvSpec(c, c, 3, 3, 1)
    x = (torch.relu(torch.randn(b, t, c, generator=g)) * 2.0).to(DEV)
    w = ((torch.rand(c, c, 3, generator=g) * 2 - 1) * 0.05).to(DEV)
    bias = torch.randn(c, generator=g).to(DEV)
    r = (torch.randn(b, t, c, generator=g) * 3).to(DEV)
    rs = ResSpec(3, 1)
    wt = ops.pack_weight(w)
    ref, den = _ref_conv(x, w, spec, bias)
    ref = torch.relu(ref) + r[:, 3:3 + spec.t_out(t)].double()
    xs, ws_, r16 = S.split(x), S.split(wt), S.split(r)
    l1 = torch.stack([wt.abs().sum(dim=1).max(), bias.abs().max()]).contiguous()
    am = S.new_bound(DEV)
    y = S.conv_nt(xs, ws_, spec, bias=bias, relu=True, residual=(r16, rs), amax_out=am, cfg=cfg,
                  s16_out=(xs.bound, l1, r16.bound))
    yv = S.join(y)
    tol = GEMM_TOL * den + float(y.bound.max()) * 2.0 ** -21
    assert bool(((yv.double() - ref).abs() <= tol).all())
    assert abs(float(am.max()) - float(ref.abs().max())) < 1e-4
    assert float(y.bound.max()) >= float(ref.abs().max())                     # the published bound is a bound
    expect = float(l1[0]) * float(x.abs().max()) + float(l1[1]) + float(r.abs().max())
    assert abs(float(y.bound.max()) - expect) < 1e-3 * expect
    # fp32 output with an S16 residual (the last block of the eval stack) agrees with the fp32-residual form
    y32 = S.conv_nt(xs, ws_, spec, bias=bias, relu=True, residual=(r16, rs), cfg=cfg)
    y32b = S.conv_nt(xs, ws_, spec, bias=bias, relu=True, residual=(S.join(r16), rs), cfg=cfg)
    assert float((y32 - y32b).abs().max()) < 1e-5
    assert float((y32.double() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("scale", [1e-4, 1.0, 3e3])
def test_model_level_parity_is_scale_invariant(scale):
    """Inputs far outside fp16's comfortable range: the device-side bounds keep the split-fp16 engine on the fp32 engine's
    results (BatchNorm renormalises, so the outputs are comparable across scales)."""
    import copy
    torch.manual_seed(2)
    fw = [3, 3, 3]
    m32 = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.0, channels=128).to(DEV).train()
    m32.math = "f32"
    with torch.no_grad():
        m32.expand_conv.weight.mul_(37.0)              # and weights whose magnitudes differ by orders between layers
        m32.layers_conv[1].weight.mul_(1e-3)
    m16 = copy.deepcopy(m32)
    m16.math = "f16x3"
    x = (torch.randn(16, 27, 17, 2, device=DEV) * 0.5).clamp(-1, 1) * scale
    tgt = torch.randn(16, 1, 17, 3, device=DEV) * 0.3
    outs = []
    for m in (m32, m16):
        yv = m(x)
        torch.mean(torch.norm(yv - tgt, dim=3)).backward()
        outs.append(yv.detach())
    assert torch.isfinite(outs[1]).all()
    assert float(torch.mean(torch.norm(outs[0] - outs[1], dim=3))) < 2e-5
    for (k, a), (_, q) in zip(m16.named_parameters(), m32.named_parameters()):
        assert torch.isfinite(a.grad).all(), k
        assert float((a.grad - q.grad).abs().max() / (q.grad.abs().max() + 1e-30)) < 2e-4, k
    ev32 = V.TemporalModel(17, 2, 17, fw, channels=128).to(DEV).eval()
    ev32.load_state_dict(m32.state_dict())
    ev32.math = "f32"
    ev16 = copy.deepcopy(ev32)
    ev16.math = "f16x3"
    with torch.no_grad():
        xe = (torch.randn(3, 60, 17, 2, device=DEV) * 0.5).clamp(-1, 1) * scale
        a, q = ev16(xe), ev32(xe)
    assert float((a - q).abs().max() / (q.abs().max() + 1e-30)) < 1e-5


@pytest.mark.timeout(300)
def test_s16_backward_bucket_exchange_through_rccl_single_rank():
    """The split-fp16 backward reports finished gradient groups from its SECOND stream (where the weight-gradient GEMMs
    run; the last group from the main stream after the join): every bucket's all-reduce is launched during backward
    and the synced gradients equal those of a plain step (world 1: the all-reduce is the identity)."""
    import os
    import socket
    import torch.distributed as dist
    from videopose3d_amd import dp, engine
    from videopose3d_amd.loss import mpjpe
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(9)
        a = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.25, channels=256).to(DEV).train()
        b = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.25, channels=256).to(DEV).train()
        b.load_state_dict(a.state_dict())
        for m in (a, b):
            m.math = "f16x3"
            m._drop_seed, m._drop_calls = 77, 0
            assert engine.use_s16(m, 27, True, batch=32)
        x = torch.randn(32, 27, 17, 2, device=DEV)
        tgt = torch.randn(32, 1, 17, 3, device=DEV)
        sync = dp.FlatGradSync(b.parameters(), direct_module=b, bucket_bytes=1 << 20, always_reduce=True)
        assert len(sync.buckets) >= 2
        for _ in range(2):
            a.zero_grad(set_to_none=True)
            mpjpe(a(x), tgt).backward()
            sync.zero_grad()
            mpjpe(b(x), tgt).backward()
            assert len(sync._handles) == len(sync.buckets)       # every bucket was launched DURING backward
            sync.sync()
            torch.cuda.synchronize()
            for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
                assert torch.equal(pa.grad, pb.grad), k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("b,t,c_out,c_in,taps", [(8, 9, 256, 256, 3), (5, 13, 512, 256, 1), (64, 27, 256, 512, 3),
                                                 (7, 37, 512, 128, 1), (64, 81, 256, 128, 1)])     # narrow B tile (k_tn_s16<1>)
def test_wgrad_from_rows_vs_fp64(b, t, c_out, c_in, taps):
    """vp3d_wgrad_rows_s16 (weight gradient straight from the S16 rows of dy and of the conv input, transposing on the
    LDS read) against an fp64 reference on the decoded operands, for ragged row counts (K tail of the 32-row tiles),
    several K-slices, strided (taps = 3) and 1x1 convs."""
    g = torch.Generator().manual_seed(21)
    dy = (torch.randn(b, t, c_out, generator=g) * 3e-4).to(DEV)
    x = torch.relu(torch.randn(b, t * taps, c_in, generator=g)).to(DEV)
    dys, xs = S.split(dy), S.split(x)
    dw = S.wgrad_rows(dys, xs, c_out, c_in, taps)
    assert dw.shape == (c_out, c_in, taps)
    dd, xd = S.join(dys).double().reshape(b * t, c_out), S.join(xs).double().reshape(b * t, taps, c_in)
    ref = torch.einsum("mo,mki->oik", dd, xd)
    den = torch.einsum("mo,mki->oik", dd.abs(), xd.abs())
    assert float(((dw.double() - ref).abs() / (den + 1e-30)).max()) < GEMM_TOL


def test_model_gradients_with_rows_form_wgrad(monkeypatch):
    """The default (_switches.SW["wgrad_rows"]): the C x C weight gradients come from vp3d_wgrad_rows_s16 and no transposed copies
    are written for them; every gradient matches the transposed-copy form (switch off) up to summation order."""
    import copy
    from videopose3d_amd import engine_s16
    torch.manual_seed(4)
    m_a = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.25, channels=256).to(DEV).train()
    m_b = copy.deepcopy(m_a)
    for m in (m_a, m_b):
        m.math = "f16x3"
        m._drop_seed, m._drop_calls = 31, 0
    x = (torch.randn(48, 27, 17, 2, device=DEV) * 0.5).clamp(-1, 1)
    tgt = torch.randn(48, 1, 17, 3, device=DEV) * 0.3
    monkeypatch.setitem(SW, "wgrad_rows", False)
    assert not engine_s16.wgrad_from_rows(256, 256)
    torch.mean(torch.norm(m_a(x) - tgt, dim=3)).backward()
    monkeypatch.setitem(SW, "wgrad_rows", True)
    assert engine_s16.wgrad_from_rows(256, 256) and not engine_s16.wgrad_from_rows(128, 128)
    y_b = m_b(x)
    torch.mean(torch.norm(y_b - tgt, dim=3)).backward()
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert float((pa.grad - pb.grad).abs().max() / (pa.grad.abs().max() + 1e-30)) < 2e-5, k


@pytest.mark.parametrize("kind,fw,causal,t_in,c", [("dilated", [3, 3, 3], False, 40, 128), ("dilated", [3, 3, 3], True, 33, 64),
                                                   ("strided", [3, 3, 3], False, 29, 128), ("strided", [3, 5, 3], True, 47, 64),
                                                   ("dilated", [3, 5, 3], False, 50, 256), ("strided", [3, 3, 3], False, 27, 256)])
def test_general_training_configurations_and_input_gradient(kind, fw, causal, t_in, c):
    """Everything run.py:171-184 can construct trains on the split-fp16 engine: the dilated class (gather-form data
    gradient, weight-gradient operand gathered from the saved rows), strided windows that do not tile the input, 5-tap
    filters, causal variants, and the gradient w.r.t. the input -- all parameter gradients, the input gradient, outputs
    and running statistics against the fp32 engine on the same weights / dropout stream (which the golden tests pin)."""
    import copy
    from videopose3d_amd import engine
    torch.manual_seed(7)
    cls = V.TemporalModel if kind == "dilated" else V.TemporalModelOptimized1f
    m32 = cls(17, 2, 17, fw, causal=causal, dropout=0.25, channels=c).to(DEV).train()
    m32.math = "f32"
    m16 = copy.deepcopy(m32)
    m16.math = "f16x3"
    for m in (m32, m16):
        m._drop_seed, m._drop_calls = 123, 0
    assert engine.use_s16(m16, t_in, True, True, batch=6)
    x0 = (torch.randn(6, t_in, 17, 2, device=DEV) * 0.5).clamp(-1, 1)
    outs, dxs = [], []
    engine.ENGINE_CALLS.clear()
    for m in (m32, m16):
        x = x0.clone().requires_grad_(True)
        yv = m(x)
        tgt = torch.randn(yv.shape, generator=torch.Generator().manual_seed(3)).to(DEV) * 0.3
        torch.mean(torch.norm(yv - tgt, dim=3)).backward()
        outs.append(yv.detach())
        dxs.append(x.grad.clone())
    assert engine.ENGINE_CALLS["s16_train"] == 1 and engine.ENGINE_CALLS["f32_train"] == 1
    assert float(torch.mean(torch.norm(outs[0] - outs[1], dim=3))) < 1e-5
    assert float((dxs[0] - dxs[1]).abs().max() / (dxs[0].abs().max() + 1e-30)) < 1e-4
    for (k, a), (_, q) in zip(m16.named_parameters(), m32.named_parameters()):
        assert float((a.grad - q.grad).abs().max() / (q.grad.abs().max() + 1e-30)) < 1e-4, k
    for (k, a), (_, q) in zip(m16.named_buffers(), m32.named_buffers()):
        if a.dtype.is_floating_point:
            assert torch.allclose(a, q, rtol=1e-5, atol=1e-6), k


def test_gather_transposed_operand_vs_torch():
    """vp3d_gather_t_s16: T[(k*C + c)][b*t_out + t] = x[b][t*stride + k*dil][c] for a dilated and a ragged strided conv."""
    g = torch.Generator().manual_seed(5)
    b, t_in, c = 3, 41, 128
    x = torch.randn(b, t_in, c, generator=g).to(DEV)
    xs = S.split(x)
    xv = S.join(xs)
    for spec in (ConvSpec(c, c, 3, 9, 1), ConvSpec(c, c, 3, 1, 3), ConvSpec(c, c, 5, 1, 5)):
        t_out = spec.t_out(t_in)
        got = S.join(S.gather_t(xs, spec, t_out))
        m = b * t_out
        assert got.shape == (spec.taps * c, S.t_pitch(m))
        for k in range(spec.taps):
            rows = xv[:, k * spec.dil: k * spec.dil + (t_out - 1) * spec.stride + 1: spec.stride]      # [b, t_out, c]
            assert torch.equal(got[k * c:(k + 1) * c, :m], rows.reshape(m, c).t().contiguous()), (spec, k)
        assert float(got[:, m:].abs().max()) == 0.0


@pytest.mark.parametrize("kc", [(128, 256), (96, 64), (64, 1024)])
@pytest.mark.parametrize("p_drop,cfg", [(0.0, 20), (0.25, 22), (0.25, -1), (0.25, 26), (0.1, -1), (0.5, -1)])
def test_fused_conv_bn_relu_dropout_epilogue_equals_unfused(p_drop, cfg, kc):
    """The expand layer's fused forward: pass 1 (no_output) writes only the BatchNorm slab statistics, pass 2 applies
    BatchNorm + ReLU + dropout in the GEMM epilogue and writes S16 rows + activation bits -- bit for bit what
    vp3d_bn_act_fwd_s16 makes of the stored conv output."""
    need_cfg(cfg)
    g = torch.Generator().manual_seed(17)
    b, t = 37, 27                                       # M = 999: ragged last tile
    k, c = kc
    spec = ConvSpec(k, c, 1)
    x = (torch.randn(b, t, k, generator=g)).clamp(-1, 1).to(DEV)
    w = (torch.randn(c, k, 1, generator=g) * 0.1).to(DEV)
    xs, ws_ = S.split(x), S.split(ops.pack_weight(w))
    m = b * t
    st_a, st_b = ops.stat_buffers(m, c, DEV), ops.stat_buffers(m, c, DEV)
    y = S.conv_nt(xs, ws_, spec, stats=st_a, cfg=cfg if cfg > 0 else -1)
    assert S.conv_nt(xs, ws_, spec, stats=st_b, no_output=True, cfg=cfg if cfg > 0 else -1) is None
    assert torch.equal(st_a[0], st_b[0]) and torch.equal(st_a[1], st_b[1])
    coef = ops.bn_finalize(torch.nn.BatchNorm1d(c).to(DEV), m, st_a)
    drop = ops.make_dropout(p_drop, 99, 3, 0)
    bound = S.new_bound(DEV)
    bound[0] = 40.0
    bits_ref, bits_f = S.new_act_bits(m, c, DEV), S.new_act_bits(m, c, DEV)
    bits_ref.zero_()
    bits_f.zero_()
    a_ref, _ = S.bn_act_fwd(y, coef, drop, None, bound, act_bits=bits_ref)
    a_f = S.conv_nt(xs, ws_, spec, act=(coef, drop, bound, bits_f), cfg=cfg if cfg > 0 else -1)
    assert torch.equal(a_f.data.view(torch.int32), a_ref.data.view(torch.int32))
    assert torch.equal(bits_f, bits_ref)
    # the dedicated expand-layer kernel (W fragments in registers, X streamed through LDS): the same bits again
    st_c = ops.stat_buffers(m, c, DEV)
    assert S.expand_fwd(xs, ws_, stats=st_c) is None
    assert torch.equal(st_a[0], st_c[0]) and torch.equal(st_a[1], st_c[1])
    bits_d = S.new_act_bits(m, c, DEV)
    bits_d.zero_()
    a_d = S.expand_fwd(xs, ws_, act=(coef, drop, bound, bits_d))
    assert torch.equal(a_d.data.view(torch.int32), a_ref.data.view(torch.int32))
    assert torch.equal(bits_d, bits_ref)
    if p_drop > 0:
        kept = float((S.join(a_f) != 0).float().mean() / (S.join(S.bn_act_fwd(y, coef, None, None, bound)[0]) != 0).float().mean())
        assert abs(kept - (1.0 - p_drop)) < 0.03, kept


@pytest.mark.parametrize("cfg", [120, 122])
@pytest.mark.parametrize("shape", [(37, 27, 256, 256, 3), (64, 48, 128, 512, 1), (300, 27, 256, 768, 3)])
def test_stream_k_equals_plain_launch(cfg, shape):
    """Stream-K configurations (shared K-tiles of the last round + in-kernel fix-up by the last contributor): same result as
    the plain launch of the same tiling up to the summation order of the K ranges (fp32 accumulators: ~1e-6 of sum|a||b|),
    fused epilogue included (bias, ReLU, residual, BatchNorm slab statistics, amax) -- and bit-identical run to run."""
    need_cfg(cfg)
    b, t, c_in, c_out, taps = shape
    g = torch.Generator().manual_seed(cfg + b)
    spec = ConvSpec(c_in, c_out, taps, 1, taps) if taps > 1 else ConvSpec(c_in, c_out, 1)
    x = torch.randn(b, t, c_in, generator=g).to(DEV)
    w = (torch.randn(c_out, c_in, taps, generator=g) * 0.05).to(DEV)
    bias = torch.randn(c_out, generator=g).to(DEV)
    xs, ws_ = S.split(x), S.split(ops.pack_weight(w))
    t_out = spec.t_out(t)
    m = b * t_out
    res = torch.randn(b, t_out, c_out, generator=g).to(DEV)
    from videopose3d_amd.plan import ResSpec

    def run(c):
        st = ops.stat_buffers(m, c_out, DEV)
        am = S.new_bound(DEV)
        y = S.conv_nt(xs, ws_, spec, bias=bias, relu=True, residual=(res, ResSpec(0, 1)), stats=st, amax_out=am, cfg=c, splits=1)
        return y, st, am

    y0, st0, am0 = run(cfg - 100)
    y1, st1, am1 = run(cfg)
    y2, st2, am2 = run(cfg)
    den = torch.nn.functional.conv1d(x.abs().permute(0, 2, 1), w.abs(), stride=spec.stride).permute(0, 2, 1)
    assert float(((y1 - y0).abs() / den).max()) < 2e-6
    assert float((st1[0] - st0[0]).abs().max() / st0[0].abs().max()) < 1e-5
    assert float((st1[1] - st0[1]).abs().max() / st0[1].abs().max()) < 1e-5
    assert abs(float(am1.max()) - float(am0.max())) <= 1e-5 * float(am0.max())
    assert torch.equal(y1, y2) and torch.equal(st1[0], st2[0]) and torch.equal(st1[1], st2[1])
    ref = torch.relu(torch.nn.functional.conv1d(x.double().permute(0, 2, 1), w.double(), bias.double(),
                                                stride=spec.stride).permute(0, 2, 1)) + res.double()
    assert float(((y1.double() - ref).abs() / (den.double() + 1.0)).max()) < 2e-6


@pytest.mark.parametrize("shape", [(37, 27, 128, 256, 0.25), (64, 81, 128, 1024, 0.25), (5, 27, 64, 64, 0.0)])
def test_expand_backward_p_from_go_equals_masked_gemm(shape):
    """vp3d_expand_bwd_p_s16 (P = G^T X with G = go * keep * bits formed in registers, go read once) against an fp64 product
    of the explicitly masked gradient, and against the two-kernel path it replaces (vp3d_act_mask_s16 + split-K GEMM)."""
    b, t, kpad, c, p = shape
    g = torch.Generator().manual_seed(b + c)
    m = b * t
    go = (torch.randn(b, t, c, generator=g) * 1e-3).to(DEV)
    x = torch.randn(m, kpad, generator=g).clamp(-1, 1).to(DEV)
    bits = torch.randint(0, 256, (m * c // 8,), generator=g, dtype=torch.uint8).to(DEV)
    gb = S.amax(go)
    xb = S.amax(x)
    _, x_t = S.split_t(x, xb, want_rows=False, want_t=True)
    ws, n, gram = S.expand_p_from_go(go, gb, bits, p, x_t, want_gram=True)
    got = ws.double().sum(0)
    xx = x.double().t() @ x.double()                     # X^T X rides along in the same launch
    assert float(((gram - xx).abs() / (x.double().abs().t() @ x.double().abs() + 1e-30)).max()) < 2e-6
    assert float((gram - S.gram(x_t)).abs().max() / xx.abs().max()) < 2e-6
    keep = torch.from_numpy(unpack_act_bits(bits, m, c)).to(DEV)
    G = go.view(m, c).double() * keep / (1.0 - p)
    ref = G.t() @ x.double()
    den = G.abs().t() @ x.double().abs()
    assert float(((got - ref).abs() / (den + 1e-30)).max()) < 2e-6
    g_t = S.act_mask(go, gb, bits, p, transposed=True)
    ws2, n2 = S.nt_raw(g_t, x_t)
    old = ws2.double().sum(0)
    assert float(((got - old).abs() / (den + 1e-30)).max()) < 2e-6


@pytest.mark.parametrize("one_col", [-1, 102, 120])
def test_im2row_split_equals_two_passes(one_col):
    """vp3d_im2row_split_s16 (the expand conv's S16 operand and its transposed copy straight from the [B, T, J*F] input)
    against vp3d_im2row + vp3d_split_t: the same bits under the same bound; vp3d_amax_floor covers the bias column's 1."""
    g = torch.Generator().manual_seed(3)
    b, t, c_in = 13, 81, 34
    x = (torch.randn(b, t, c_in, generator=g) * 0.3).to(DEV)
    spec = ConvSpec(c_in, 256, 3, 1, 3)
    kpad = ops.padded_k(spec)
    assert kpad == 128
    xin = ops.im2row(x, spec, kpad, one_col)
    bound = S.amax(x, floor=1.0 if one_col >= 0 else 0.0)
    assert float(bound.max()) == max(float(x.abs().max()), 1.0 if one_col >= 0 else 0.0)
    m = xin.shape[0] * xin.shape[1]
    r_ref, t_ref = S.split_t(xin.view(m, kpad), bound)
    r, tt = S.im2row_split(x, spec, kpad, one_col, bound)
    assert torch.equal(r.data.view(m, kpad).view(torch.int32), r_ref.data.view(torch.int32))
    assert torch.equal(tt.data.view(torch.int32), t_ref.data.view(torch.int32))


def test_engines_agree_on_random_configurations():
    """tools/fuzz_engines.py as a test (30 random model / batch configurations: class, arc incl. 1-tap and single-filter
    models, causal, dense, channels, joint counts, dropout, input gradients): the split-fp16 engine against the exact-fp32
    one on the same weights and masks.  This sweep found five configuration bugs in round 2 (96-column and < 32-column
    expand rows, C_in % 32 == 0 inputs, models without residual blocks, dilated 1-tap convs in backward)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_engines.py"), "30", "7"], capture_output=True, text=True,
                       cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("joints,c,arc", [(17, 128, [3, 3, 3]), (15, 64, [3, 3, 3, 3]), (17, 256, [3, 1, 3])])
def test_fused_prologue_is_bit_identical(joints, c, arc, monkeypatch):
    """The two-launch prologue (vp3d_prologue_a_s16: every maximum + the activation bounds; vp3d_prologue_b_s16: input staging +
    all weight packs) against the seven launches it replaces: same arithmetic, so output, gradients and running statistics
    must agree bit for bit."""
    from videopose3d_amd import engine
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    try:
        torch.manual_seed(9)
        m = V.TemporalModelOptimized1f(joints, 2, 17, arc, dropout=0.25, channels=c).to(DEV).train()
        m.math = "f16x3"
        m._drop_seed = 0xF00D
        rf = m.receptive_field()
        x = (torch.randn(24, rf, joints, 2, device=DEV) * 0.5).clamp(-1, 1)
        tgt = torch.randn(24, 1, 17, 3, device=DEV) * 0.3
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        res = []
        for fused in ("1", "0"):
            monkeypatch.setitem(SW, "prologue_fused", fused == "1")
            m.load_state_dict(sd0)
            m._drop_calls = 0
            m.zero_grad(set_to_none=True)
            y = m(x)
            torch.mean(torch.norm(y - tgt, dim=3)).backward()
            res.append((y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                        {k: v.clone() for k, v in m.state_dict().items()}))
        assert torch.equal(res[0][0], res[1][0])
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], res[1][1][k]), k
        for k in res[0][2]:
            assert torch.equal(res[0][2][k], res[1][2][k]), k
    finally:
        engine.S16_MIN_FORWARD_FLOPS.update(keep)


# ---- the block-exponent format at its edges: adversarial intra-tensor dynamic range (reference: nn.BatchNorm1d's affine and
# the conv weights are unconstrained, common/model.py:32,102,113-119) -- against the float64 oracle, with the exact-fp32 engine
# beside it as the yardstick of what ANY fp32 evaluation of the same step loses (tools/range_edges.py prints the full table) ---
RANGE_EDGE_CASES = [(128, c, s, None) for c in ("gamma", "beta", "w_row", "w_col") for s in (8, 14, 20)] + \
    [(128, "w_row", 12, None), (128, "none", 0, 1e4),
     (1024, "gamma", 20, None), (1024, "beta", 20, None), (1024, "w_row", 12, None), (1024, "w_row", 20, None),
     (1024, "w_col", 20, None), (1024, "none", 0, 1e4)]


def _range_models(channels, sd):
    from videopose3d_amd import engine
    out = []
    for math in ("f32", "f16x3"):
        m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.0, channels=channels).to(DEV).train()
        m.math = math
        m.load_state_dict(sd)
        out.append(m)
    return out


@pytest.mark.timeout(600)
@pytest.mark.parametrize("channels,case,s,joint", RANGE_EDGE_CASES)
def test_dynamic_range_edges_raw_format_vs_fp64_oracle(channels, case, s, joint, monkeypatch):
    """ONE hot channel (a BatchNorm gamma_c / beta_c, a conv-weight output row, a conv-weight input column = hot dy) up to
    2^20 x the rest, or one input joint 1e4 x the others, with the guard OFF: the raw split-fp16 format stays on the
    exact-fp32 engine's own error -- output MPJPE <= 4 x the fp32 engine's + 2e-6 (and <= 1e-4 absolute wherever the fp32
    engine itself is within 2e-5 of the float64 oracle: a beta_c of 1e5 is ill-conditioned for ANY fp32 evaluation, 3e-3
    on the fp32 engine), every gradient tensor in max-norm <= 4 x the fp32 engine's + 5e-6.  (A hot channel dominates every
    contraction it enters in EITHER arithmetic: the 2^-40 * bound absolute error of the small elements stays below fp32's
    own rounding of those sums; vp3d_s16.h.)"""
    from tests import util as U
    from videopose3d_amd import engine, range_guard
    monkeypatch.setenv("VP3D_RANGE_GUARD", "0")
    sd = U.range_edge_state(channels, case, s)
    x, tgt = U.range_edge_batch(64, joint)
    yo, go = U.range_edge_oracle(sd, x, tgt)
    m32, m16 = _range_models(channels, sd)
    n16 = engine.ENGINE_CALLS["s16_train"]
    r32 = U.range_edge_errors(m32, x, tgt, yo, go)
    assert engine.ENGINE_CALLS["s16_train"] == n16
    r16 = U.range_edge_errors(m16, x, tgt, yo, go)
    assert engine.ENGINE_CALLS["s16_train"] == n16 + 1 and not range_guard.tripped(m16)      # the raw format was measured
    assert r16["finite"] and r32["finite"]
    assert r16["mpjpe"] <= 4 * r32["mpjpe"] + 2e-6, (r16, r32)
    assert r32["mpjpe"] > 2e-5 or r16["mpjpe"] <= 1e-4, (r16, r32)
    assert r16["grad_maxnorm"] <= 4 * r32["grad_maxnorm"] + 5e-6, (r16, r32)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("channels", [128, 1024])
@pytest.mark.parametrize("case,s", [("gamma", 14), ("gamma", 20), ("gamma_all_layers", 14), ("gamma_all_layers", 20),
                                    ("beta", 20), ("w_row", 20)])
def test_range_guard_routes_adversarial_parameters_to_the_fp32_engine(channels, case, s):
    """As shipped (guard on): parameters whose per-channel spread is outside the format's lossless window (activation
    bounds 2^12, weight rows 2^16) are measured on the device when they are loaded and the model's calls run on the
    exact-fp32 engine -- bit for bit what math='f32' gives; re-loading benign parameters puts it back on split-fp16."""
    import warnings
    from tests import util as U
    from videopose3d_amd import engine, range_guard
    sd = U.range_edge_state(channels, case, s)
    x, tgt = U.range_edge_batch(32)
    m32, m16 = _range_models(channels, sd)
    outs = []
    n16, n32 = engine.ENGINE_CALLS["s16_train"], engine.ENGINE_CALLS["f32_train"]
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        for m in (m32, m16):
            y = m(x.to(DEV))
            torch.mean(torch.norm(y - tgt.to(DEV), dim=3)).backward()
            outs.append(y.detach())
    st = range_guard.status(m16)
    assert st["tripped"] and st["sync_checks"] == 1 and (st["last"][0] > 12 or st["last"][1] > 16), st
    if case != "beta":                               # (a hot beta_c is measured against gamma_c * sqrt(M - 1), not against 1)
        assert st["last"][0 if case != "w_row" else 1] >= s - 1, st         # the statistic sees the planted factor
    assert any("exact-fp32 engine" in str(w.message) for w in wlist)
    assert engine.ENGINE_CALLS["s16_train"] == n16 and engine.ENGINE_CALLS["f32_train"] == n32 + 2
    assert torch.equal(outs[0], outs[1])
    for (k, a), (_, b) in zip(m16.named_parameters(), m32.named_parameters()):
        assert torch.equal(a.grad, b.grad), k
    # benign parameters again: measured at the load, back on the split-fp16 engine
    m16.load_state_dict(U.range_edge_state(channels, "none", 0))
    m16(x.to(DEV))
    st = range_guard.status(m16)
    assert not st["tripped"] and st["sync_checks"] == 2 and st["last"][0] <= 3 and st["last"][1] <= 3, st
    assert engine.ENGINE_CALLS["s16_train"] == n16 + 1


def test_range_guard_steady_state_has_no_host_synchronisation_and_catches_inplace_edits(monkeypatch):
    """Steady state: one synchronous measurement at the first call, then an asynchronous one every CHECK_EVERY calls whose
    result a LATER call reads from pinned memory.  An in-place edit that by-passes load_state_dict is caught that way."""
    from tests import util as U
    from videopose3d_amd import engine, range_guard
    monkeypatch.setattr(range_guard, "CHECK_EVERY", 4)
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.0, channels=128).to(DEV).train()
    m.math = "f16x3"
    m.load_state_dict(U.range_edge_state(128, "none", 0))
    x = U.range_edge_batch(16)[0].to(DEV)
    for _ in range(9):
        m(x)
    st = range_guard.status(m)
    assert st["sync_checks"] == 1 and st["checks"] >= 3 and not st["tripped"], st
    with torch.no_grad():
        m.layers_bn[1].weight[7] *= 2.0 ** 15          # (no load_state_dict: nothing tells the guard)
    n32 = engine.ENGINE_CALLS["f32_train"]
    for i in range(12):
        m(x)
        torch.cuda.synchronize()                       # (lets the asynchronous copy's event complete between calls)
    st = range_guard.status(m)
    assert st["tripped"] and st["sync_checks"] == 1 and st["last"][0] >= 14, st
    assert engine.ENGINE_CALLS["f32_train"] > n32


def test_range_stats_kernel_matches_numpy():
    """vp3d_range_stats: E(max) - E(lower median of the non-zero groups) of |gamma_c| k + |beta_c| per BatchNorm layer and of
    the per-row maxima per weight tensor, maximum over layers / tensors."""
    import ctypes as C
    import numpy as np
    from videopose3d_amd import _lib
    g = torch.Generator().manual_seed(5)
    c = 192
    gam = [torch.randn(c, generator=g) for _ in range(3)]
    bet = [torch.randn(c, generator=g) * 0.1 for _ in range(3)]
    gam[1][17] = 3.0e5
    gam[2][:40] = 0.0
    bet[2][:40] = 0.0                                   # dead channels: left out of the median
    kf = [31.0, 5.0, 100.0]
    ws = [torch.randn(c, 96, generator=g) * 0.05, torch.randn(64, 301, generator=g)]
    ws[0][3] *= 7.0e4
    ws[1][5] = 0.0

    def spread(v):
        v = np.asarray(v, np.float64)
        e = np.sort(np.frexp(v[v > 0])[1])
        return int(e[-1] - e[(len(e) + 1) // 2 - 1])
    want_a = max(spread(gm.abs().numpy() * k + bt.abs().numpy()) for gm, bt, k in zip(gam, bet, kf))
    want_w = max(spread(w.abs().amax(dim=1).numpy()) for w in ws)
    gd, bd, wd = [t.to(DEV) for t in gam], [t.to(DEV) for t in bet], [t.to(DEV).contiguous() for t in ws]
    out = torch.zeros(2, dtype=torch.int32, device=DEV)
    wsp = torch.empty(sum(w.shape[0] for w in ws), dtype=torch.int32, device=DEV)

    def ptrs(ts):
        arr = (C.c_void_p * len(ts))()
        for i, t in enumerate(ts):
            arr[i] = t.data_ptr()
        return arr
    _lib.check(_lib.lib().vp3d_range_stats(ops._stream(), 3, c, ptrs(gd), ptrs(bd), (C.c_float * 3)(*kf), 2, ptrs(wd),
                                           (C.c_int64 * 2)(*[w.shape[0] for w in ws]), (C.c_int64 * 2)(*[w.shape[1] for w in ws]),
                                           wsp.data_ptr(), wsp.numel(), out.data_ptr()), "vp3d_range_stats")
    got = out.cpu().tolist()
    assert got == [want_a, want_w], (got, want_a, want_w)
    assert want_a >= 17 and want_w >= 15


def test_tile_224_planned_for_the_benchmark_rows_and_finalize_agrees():
    """The planner picks the 224 x 256 tiling (configuration 28) where the 256-row tiling strands most of a round -- the
    27,648-row layers of the benchmark step -- when the caller allows it, never for raw (weight-gradient) launches; the
    BatchNorm coefficients finalised from 32-row slabs agree with those from 64-row slabs to fp32 rounding."""
    assert S.plan(27648, 1024, 3072, mix=True) == (28, 1) and S.plan(27648, 1024, 1024, mix=True) == (28, 1)
    assert S.plan(27648, 3072, 1024, mix=True) == (28, 1)
    assert S.plan(27648, 1024, 3072)[0] == 22 and S.plan(27648, 1024, 3072, raw=True, mix=True)[0] != 28
    assert S.plan(1024, 1024, 3072, mix=True)[0] == 20          # the T_out = 1 tail stays on split 128 x 128 tiles
    assert S.plan(9216, 1024, 3072, mix=True) == (29, 1)         # 232 tiles of 160 rows in one launch instead of 3 K slices of
                                                                 # 256 x 256 tiles + a finishing pass
    assert S.plan(9216, 1024, 1024, mix=True) == (29, 1)         # 58 x 4 = 232 tiles of 160 rows: 91 % of one round
    assert S.plan(3072, 1024, 3072, mix=True) == (29, 3)         # 80 tiles of 160 rows x 3 K slices = 240 workgroups
    assert S.stat_slab_rows(29, 1) == 32 and S.stat_slab_rows(29, 3) == 64 and S.stat_slab_rows(22, 3) == 64
    assert S.plan(1024, 1024, 1024, mix=True) == S.plan(1024, 1024, 1024)      # the smallest launches stay on 128 x 128 tiles
    g = torch.Generator().manual_seed(21)
    b, t, c = 40, 27, 256
    spec = ConvSpec(c, c, 3, 1, 3)
    x = torch.relu(torch.randn(b, t, c, generator=g)).to(DEV)
    w = ((torch.rand(c, c, 3, generator=g) * 2 - 1) * 0.05).to(DEV)
    xs, ws = S.split(x), S.split(ops.pack_weight(w))
    m = b * spec.t_out(t)
    bn = torch.nn.BatchNorm1d(c).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    coefs, stats_run = [], []
    for cfg in (22, 28):
        slab = S.stat_slab_rows(cfg)
        st = ops.stat_buffers(m, c, DEV, slab)
        S.conv_nt(xs, ws, spec, stats=st, cfg=cfg, splits=1, stat_slab=slab)
        bn.running_mean.zero_()
        bn.running_var.fill_(1.0)
        coefs.append(ops.bn_finalize(bn, m, st, slab_rows=slab).clone())
        stats_run.append((bn.running_mean.clone(), bn.running_var.clone()))
    assert float((coefs[0] - coefs[1]).abs().max() / coefs[0].abs().max()) < 1e-6
    assert torch.allclose(stats_run[0][0], stats_run[1][0], rtol=1e-5, atol=1e-7)
    assert torch.allclose(stats_run[0][1], stats_run[1][1], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("p_drop", [0.25, 0.5, 0.75, 0.1])
def test_dropout_mask_two_bit_mode_statistics(p_drop):
    """p = 0.25 / 0.5 / 0.75: 2 random bits per element decide exactly and ONE Philox block serves 64 elements (vp3d_dropout.h
    "two" mode; the expand layer's kernel shares the block across its lanes); any other p keeps 16 bits per element.  Either way:
    the keep rate, no correlation between neighbours inside a block (the 2-bit fields of one word), none between the 8-element
    groups of a block, none between blocks, and another (seed, offset, layer) gives another mask."""
    n = 1 << 22
    d = ops.make_dropout(p_drop, 4321, 5, 2)
    mk = (ops.dropout_mask(n, d, DEV) > 0).float()
    keep = 1.0 - p_drop
    assert abs(float(mk.mean()) - keep) < 2e-3
    var = keep * (1 - keep)
    for lag in (1, 2, 7, 8, 9, 63, 64, 65, 4096):
        cov = float((mk[:-lag] * mk[lag:]).mean()) - keep * keep
        assert abs(cov) < 4e-3 * var / 0.1875 + 1e-3, (lag, cov)
    per_pos = mk.view(-1, 64).mean(dim=0)                 # every position of a 64-element block keeps at the same rate
    assert float((per_pos - keep).abs().max()) < 8e-3
    other = (ops.dropout_mask(n, ops.make_dropout(p_drop, 4321, 6, 2), DEV) > 0).float()
    assert abs(float((mk * other).mean()) - keep * keep) < 2e-3
    assert float(ops.dropout_mask(n, d, DEV).max()) == pytest.approx(1.0 / keep, rel=1e-6)


@pytest.mark.parametrize("joints,b,shift,scale", [(17, 96, 0.0, 0.5), (17, 37, 300.0, 1.0), (15, 64, -40.0, 3e-3), (5, 50, 0.0, 1.0)])
def test_expand_statistics_from_the_centred_gram_matrix(joints, b, shift, scale):
    """vp3d_expand_stats_gram_s16 (the expand layer's training-mode BatchNorm coefficients from the centred second-moment matrix
    of its <= 128-column input, no pass over the conv output) against the statistics pass + vp3d_bn_finalize it replaces and
    against float64 -- including inputs whose mean is 300 standard deviations away from zero (the shift by the first row is what
    keeps E[x^2] - E[x]^2 out of it) and a ragged last slab."""
    from videopose3d_amd import engine_s16
    g = torch.Generator().manual_seed(31)
    c, t = 256, 27
    c_in = joints * 2
    spec = ConvSpec(c_in, c, 3, 1, 3)
    kpad = engine_s16.expand_kpad(spec)
    kv = 3 * c_in
    one_col = kv
    assert kv < kpad
    x = (torch.randn(b, t, c_in, generator=g) * scale + shift).to(DEV)
    w = ((torch.rand(c, c_in, 3, generator=g) * 2 - 1) * 0.1).to(DEV)
    xb = S.amax(x, floor=1.0)
    x_rows, x_t = S.im2row_split(x, spec, kpad, one_col, xb, want_t=True)
    w_packed = ops.pack_weight(w, ld_out=kpad)
    ws_ = S.split(w_packed)
    m = b * spec.t_out(t)
    bn_a, bn_b = torch.nn.BatchNorm1d(c).to(DEV), torch.nn.BatchNorm1d(c).to(DEV)
    with torch.no_grad():
        for bn in (bn_a, bn_b):
            bn.weight.copy_(torch.linspace(0.5, 1.5, c))
            bn.bias.copy_(torch.linspace(-1, 1, c))
    st = ops.stat_buffers(m, c, DEV)
    assert S.expand_fwd(x_rows, ws_, stats=st) is None
    coef_a = ops.bn_finalize(bn_a, m, st)
    coef_b = S.expand_stats_gram(x_t, w_packed, bn_b, m, kv, one_col)
    # float64 reference of the conv output's statistics from the S16-rounded operands' exact values
    xr = S.join(x_rows).double().reshape(m, kpad)[:, :kv]
    y = xr @ w_packed.double()[:, :kv].t()
    den = (xr.abs() @ w_packed.double()[:, :kv].abs().t()).mean(0)          # sum |w||x| per channel: what fp32 rounding of y scales with
    mean, var = y.mean(0), y.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    for name, coef, noise in (("statistics pass", coef_a, 1e-6), ("centred Gram", coef_b, 2e-7)):
        # (the pass over the conv output inherits the fp32 accumulation noise of y itself, ~1e-6 sum|w||x|; the Gram path forms its
        #  sums in fp64 from the exact S16 values)
        err_m = (coef[2].double() - mean).abs()
        assert bool((err_m <= 2e-5 * (mean.abs() + var.sqrt()) + noise * den).all()), (name, float(err_m.max()))
        # (std(y) ~ 3e-3 next to |y| ~ 20 in the third case: the fp32 noise of y is 3e-4 of its standard deviation, and the
        #  pass over y carries it into the variance; the Gram path does not)
        if name == "centred Gram" or scale >= 0.1:
            assert float((coef[3].double() / invstd - 1).abs().max()) < 2e-5, name
    if scale >= 0.1:
        assert float((coef_a[0] / coef_b[0] - 1).abs().max()) < 2e-5
        assert float((coef_a[1] - coef_b[1]).abs().max() / coef_a[1].abs().max()) < 2e-5
        assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=2e-5, atol=1e-7)
    else:
        assert torch.allclose(bn_b.running_mean.double(), 0.1 * mean, rtol=1e-5, atol=1e-6)
        unb = var * m / (m - 1)
        assert torch.allclose(bn_b.running_var.double(), 0.9 + 0.1 * unb, rtol=1e-5, atol=1e-7)
    assert int(bn_b.num_batches_tracked) == 1


@pytest.mark.parametrize("step,wscale", [(1e-1, 0.1), (1e-2, 0.1), (1e-2, 1.0), (3e-3, 1.0), (1e-3, 1.0), (1e-4, 3.0)])
def test_expand_statistics_gram_on_correlated_columns_and_difference_filters(step, wscale):
    """The ill-conditioned case of var_n = W[n]^T Cov(x) W[n] (round-4 advisor finding): real 2D-pose windows are random walks in
    time (x_{t+1} = x_t + step * noise: adjacent taps of one joint correlate to 1 - step^2 / 2) and trained filters include
    temporal differences (w = (a, -a, 0), (a, -2a, a)), so var_n is kappa_n = sum |w_i Cov_ij w_j| / (var_n + eps) times smaller
    than the terms it is the sum of, and every relative error e of the second-moment matrix arrives in invstd as e * kappa_n / 2.
    The reference (fp32 conv output, then BatchNorm) has its own floor there: the fp32 noise of y is ~6e-8 sqrt(K) of
    sum |w||x|, i.e. ~1e-6 sqrt(kappa_n) of the normalised output.  Bar: invstd within 2e-5 + 1e-6 sqrt(kappa_n) of float64 -- not
    worse than what the reference's own arithmetic leaves -- for every channel with kappa_n <= 2^16, AND the kernel must report
    floor(log2 max kappa) so that range_guard can move the layer to the statistics pass beyond that (GRAM_KAPPA_LOG2_MAX)."""
    from videopose3d_amd import engine_s16, range_guard
    g = torch.Generator().manual_seed(77)
    c, t, b, joints = 256, 27, 64, 17
    c_in = joints * 2
    spec = ConvSpec(c_in, c, 3, 1, 3)
    kpad = engine_s16.expand_kpad(spec)
    kv = 3 * c_in
    one_col = kv
    x0 = torch.randn(b, 1, c_in, generator=g) * 0.4
    x = (x0 + step * torch.randn(b, t, c_in, generator=g).cumsum(1)).to(DEV)
    w = torch.zeros(c, c_in, 3)
    a = (torch.rand(c, c_in, generator=g) * 2 - 1) * wscale
    w[0::4, :, 0], w[0::4, :, 1] = a[0::4], -a[0::4]                           # first difference
    w[1::4, :, 0], w[1::4, :, 1], w[1::4, :, 2] = a[1::4], -2 * a[1::4], a[1::4]    # second difference
    w[2::4] = (torch.rand(c // 4, c_in, 3, generator=g) * 2 - 1) * wscale          # generic
    w[3::4, :, 1] = a[3::4]                                                      # centre tap only
    w = w.to(DEV)
    xb = S.amax(x, floor=1.0)
    x_rows, x_t = S.im2row_split(x, spec, kpad, one_col, xb, want_t=True)
    w_packed = ops.pack_weight(w, ld_out=kpad)
    m = b * spec.t_out(t)
    bn = torch.nn.BatchNorm1d(c).to(DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    coef = S.expand_stats_gram(x_t, w_packed, bn, m, kv, one_col, illcond=flag)
    xr = S.join(x_rows).double().reshape(m, kpad)[:, :kv]
    wd = w_packed.double()[:, :kv]
    y = xr @ wd.t()
    var = y.var(0, unbiased=False)
    cov = torch.cov(xr.t(), correction=0)
    kappa = ((wd.abs() @ cov.abs()) * wd.abs()).sum(1) / (var + 1e-5)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    err = (coef[3].double() / invstd - 1).abs()
    bar = 2e-5 + 1e-6 * kappa.sqrt()
    ok = kappa <= 2.0 ** range_guard.GRAM_KAPPA_LOG2_MAX
    worst = int((err / bar * ok).argmax())
    print("step %g w %g: kappa max 2^%.1f (flag %d), invstd err max %.3g overall; worst guarded channel err %.3g bar %.3g kappa %.3g"
          % (step, wscale, float(kappa.max().log2()), int(flag), float(err.max()), float(err[worst]), float(bar[worst]),
             float(kappa[worst])))
    assert bool((err[ok] <= bar[ok]).all()), (float(err[worst]), float(bar[worst]), float(kappa[worst]))
    assert abs(int(flag) - int(torch.floor(kappa.max().log2()))) <= 1, (int(flag), float(kappa.max().log2()))
    err_m = (coef[2].double() - y.mean(0)).abs()
    assert bool((err_m <= 2e-5 * (y.mean(0).abs() + var.sqrt()) + 1e-6 * (xr.abs() @ wd.abs().t()).mean(0)).all())


def test_gram_statistics_fall_back_to_the_pass_over_the_conv_output_when_ill_conditioned():
    """Model level: difference-type expand filters on random-walk keypoints with kappa > 2^16 -> range_guard reads the kernel's
    flag at most 2 x CONSUME_AFTER calls after the first periodic measurement, warns once, and the expand layer's statistics come from the
    pass over the conv output from then on (output then within the fp32 engine's own distance of the float64 oracle)."""
    from videopose3d_amd import range_guard
    torch.manual_seed(3)
    fw = [3, 3, 3]
    m = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.0, channels=1024).to(DEV).train()
    m.math = "f16x3"
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(1280, 1, 17, 2, generator=g) * 0.4            # (45 GFLOP forward: above the split-fp16 engine's size threshold)
    x = (x0 + 1e-4 * torch.randn(1280, 27, 17, 2, generator=g).cumsum(1)).to(DEV)
    with torch.no_grad():
        wgt = m.expand_conv.weight
        a = (torch.rand(1024, 34, generator=g) * 2 - 1).to(DEV) * 3.0
        wgt[:, :, 0], wgt[:, :, 1], wgt[:, :, 2] = a, -a, 0.0
    range_guard.invalidate(m)
    import warnings as _w
    with _w.catch_warnings(record=True) as rec:
        _w.simplefilter("always")
        for _ in range(2 + 2 * range_guard.CONSUME_AFTER + 1):
            m(x)
            torch.cuda.synchronize()                   # (lets the asynchronous copy's event complete between calls)
    st = range_guard.status(m)
    assert st["gram_off"] and st["gram_log2_kappa"] >= range_guard.GRAM_KAPPA_LOG2_MAX, st
    assert sum("ill-conditioned" in str(r.message) for r in rec) == 1
    assert not st["tripped"]
    range_guard.invalidate(m)                                  # a re-load measures afresh: the matrix path is tried again
    m(x)
    assert not range_guard.status(m)["gram_off"]


@pytest.mark.parametrize("joints,b,shift,scale,p", [(17, 96, 0.0, 0.5, 0.25), (17, 37, 300.0, 1.0, 0.0), (15, 64, -40.0, 3e-3, 0.5)])
def test_expand_backward_rebuilds_xtx_from_the_forward_centred_gram(joints, b, shift, scale, p):
    """vp3d_expand_bwd_gram_s16: the expand layer's backward (dW, dgamma, dbeta from P = G^T X, X^T X and the weights) with X^T X
    rebuilt in fp64 from the FORWARD's centred second-moment matrix (vp3d_expand_stats_gram_s16's `gram`) + the offsets in the
    first column of the transposed X, against the same launch fed the exact float64 X^T X of the S16-rounded operands, and
    against the ride-along form (X^T X accumulated in fp32 inside the P launch + vp3d_sum_slices) -- incl. an input 300 standard
    deviations away from zero, where the uncentred fp32 products lose what the centred ones keep."""
    from videopose3d_amd import engine_s16
    g = torch.Generator().manual_seed(37)
    c, t = 256, 27
    c_in = joints * 2
    spec = ConvSpec(c_in, c, 3, 1, 3)
    kpad = engine_s16.expand_kpad(spec)
    kv = 3 * c_in
    one_col = kv
    x = (torch.randn(b, t, c_in, generator=g) * scale + shift).to(DEV)
    w = ((torch.rand(c, c_in, 3, generator=g) * 2 - 1) * 0.1).to(DEV)
    xb = S.amax(x, floor=1.0)
    x_rows, x_t = S.im2row_split(x, spec, kpad, one_col, xb, want_t=True)
    w_packed = ops.pack_weight(w, ld_out=kpad)
    m = b * spec.t_out(t)
    bn = torch.nn.BatchNorm1d(c).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, c))
        bn.bias.copy_(torch.linspace(-1, 1, c))
    coef, gram_c = S.expand_stats_gram(x_t, w_packed, bn, m, kv, one_col, want_gram=True)
    xr = S.join(x_rows).double().reshape(m, kpad)
    xx = xr.t() @ xr                                                       # exact X^T X of the S16-rounded rows (incl. the 1 column)
    # the matrix itself: G + o_i G[one][j] + o_j G[one][i] + M o_i o_j  ==  X^T X
    o = xr[0].clone()
    o[one_col] = 0.0
    g1 = gram_c[one_col]
    rebuilt = gram_c + o[:, None] * g1[None, :] + o[None, :] * g1[:, None] + float(m) * o[:, None] * o[None, :]
    assert float(gram_c[one_col, one_col]) == float(m)
    den_xx = xr.abs().t() @ xr.abs() + 1e-30
    assert float(((rebuilt - xx).abs() / den_xx)[:kv + 1, :kv + 1].max()) < 1e-6
    go = (torch.randn(b, spec.t_out(t), c, generator=g) * 1e-3).to(DEV)
    gb = S.amax(go)
    bits = torch.randint(0, 256, (m * c // 8,), generator=g, dtype=torch.uint8).to(DEV)
    ws, n, gram_ride = S.expand_p_from_go(go, gb, bits, p, x_t, want_gram=True)
    ws2, n2 = S.expand_p_from_go(go, gb, bits, p, x_t, want_gram=False)
    assert n2 == n and torch.equal(ws, ws2)                                # (P does not depend on the ride-along)
    args = (w_packed, coef, m, c_in, 3, one_col, False)
    ideal = S.expand_bwd(None, x_t, xx.contiguous(), *args, partials=(ws, n))
    ride = S.expand_bwd(None, x_t, gram_ride, *args, partials=(ws, n))
    cent = S.expand_bwd(None, x_t, gram_c, *args, partials=(ws, n), gram_centred=x_t)
    for k, name in enumerate(("dW", "dgamma", "dbeta")):
        ref = ideal[k].double()
        tol = 2e-6 * float(ref.abs().max())
        assert float((cent[k].double() - ref).abs().max()) <= tol, name
        if shift == 0.0:                                                   # (the fp32 ride-along: a sanity bound, on centred data)
            assert float((ride[k].double() - ref).abs().max()) <= 1e-3 * float(ref.abs().max()), name
    assert torch.equal(cent[1], ideal[1]) and torch.equal(cent[2], ideal[2])   # (dgamma / dbeta do not involve X^T X)


def test_range_cols_statistic_and_hot_input_joint_trips_the_guard():
    """vp3d_range_cols: spread of the per-column maxima of a row-major tensor (binary orders between the hottest column and the
    median one; zero columns left out; ws left zero), and the guard acting on it: an input batch with one joint 2^20 hotter than
    the others (and, separately, a loss gradient with one hot column) moves the model to the exact-fp32 engine."""
    import ctypes as C
    from videopose3d_amd import _lib, engine, range_guard
    x = torch.rand(4000, 34, device=DEV) + 0.5
    x[:, 7] *= 2.0 ** 9
    x[:, 3] = 0.0
    ws = torch.zeros(1024, dtype=torch.int32, device=DEV)
    out = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(_lib.lib().vp3d_range_cols(ops._stream(), 4000, 34, x.data_ptr(), 34, ws.data_ptr(), out.data_ptr()), "range_cols")
    assert int(out) in (9, 10) and int(ws.abs().sum()) == 0
    xs = torch.rand(100, 64, device=DEV)[:, :51]                                  # a strided view: ld = 64
    out.zero_()
    _lib.check(_lib.lib().vp3d_range_cols(ops._stream(), 100, 51, xs.data_ptr(), 64, ws.data_ptr(), out.data_ptr()), "range_cols")
    assert int(out) <= 1
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    try:
        import warnings as _w
        for hot_input in (True, False):
            torch.manual_seed(1)
            m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.0, channels=256).to(DEV).train()
            m.math = "f16x3"
            xin = (torch.randn(64, 27, 17, 2, device=DEV) * 0.5).clamp(-1, 1)
            tgt = torch.randn(64, 1, 17, 3, device=DEV) * 0.3
            wgt = torch.ones(1, 1, 17, 1, device=DEV)
            if hot_input:
                xin[:, :, 5] *= 2.0 ** 20
            else:
                wgt[:, :, 11] = 2.0 ** 20                                          # the loss weighs one joint 2^20 times the others
            with _w.catch_warnings(record=True) as rec:
                _w.simplefilter("always")
                for _ in range(2 * range_guard.CHECK_EVERY + 2 * range_guard.CONSUME_AFTER + 2):
                    m.zero_grad(set_to_none=True)
                    (m(xin) * wgt - tgt).abs().mean().backward()
                    torch.cuda.synchronize()
                    if range_guard.tripped(m):
                        break
            st = range_guard.status(m)
            assert st["tripped"], (hot_input, st)
            assert (st["io_last"][0] if hot_input else st["io_last"][1]) >= 19, st
            assert not engine.use_s16(m, 27, True, batch=64)
            assert sum("dynamic range" in str(r.message) for r in rec) == 1
    finally:
        engine.S16_MIN_FORWARD_FLOPS.update(keep)


@pytest.mark.parametrize("b,t_in,taps,c", [(1024, 3, 3, 1024), (1024, 1, 1, 1024), (1024, 9, 3, 1024), (37, 3, 3, 256)])
def test_bn_finalize_inside_the_split_k_finishing_pass_is_bit_identical(b, t_in, taps, c):
    """vp3d_s16_fin: a K-sliced forward launch whose finishing pass also finalises the BatchNorm statistics (the last workgroup
    of a 64-column strip merges the strip's slabs: one ticket per strip) against the same launch + the separate
    vp3d_bn_finalize it removes from the forward's dependent chain: coefficients, running statistics and num_batches_tracked
    bit-identical, the conv output too; tickets left zero; a second launch on the same tickets works."""
    g = torch.Generator().manual_seed(11)
    spec = ConvSpec(c, c, taps, 1, taps)
    x = S.split((torch.randn(b, t_in, c, generator=g) * 0.7 + 0.3).to(DEV))
    w = torch.randn(c, c, taps, generator=g) * 0.03
    wt = S.split(ops.pack_weight(w.to(DEV)))
    m = b * spec.t_out(t_in)
    cfg, splits = S.plan(m, c, taps * c, mix=True)
    if splits <= 1:
        splits = 2
    outs = []
    for use_fin in (False, True, True):
        bn = torch.nn.BatchNorm1d(c).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, c))
            bn.bias.copy_(torch.linspace(-1, 1, c))
            bn.running_mean.copy_(torch.linspace(-0.2, 0.2, c))
        slab = S.stat_slab_rows(cfg, splits)
        st = ops.stat_buffers(m, c, DEV, slab)
        if use_fin:
            y, coef = S.conv_nt(x, wt, spec, stats=st, cfg=cfg, splits=splits, stat_slab=slab, fin=(bn, None))
            assert coef is not None
        else:
            y = S.conv_nt(x, wt, spec, stats=st, cfg=cfg, splits=splits, stat_slab=slab)
            coef = ops.bn_finalize(bn, m, st, slab_rows=slab)
        outs.append((y, coef, bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)))
    assert int(S._fin_tickets(DEV, 1).abs().sum()) == 0
    for other in outs[1:]:
        assert torch.equal(outs[0][0], other[0])
        assert torch.equal(outs[0][1], other[1]), float((outs[0][1] - other[1]).abs().max())
        assert torch.equal(outs[0][2], other[2]) and torch.equal(outs[0][3], other[3]) and other[4] == 1
    # against float64 from the conv output itself
    yd = outs[0][0].double().reshape(m, c)
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    assert torch.allclose(outs[1][1][2].double(), mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(outs[1][1][3].double(), 1.0 / torch.sqrt(var + 1e-5), rtol=2e-5)


def test_model_step_with_and_without_finalize_in_the_finishing_pass(monkeypatch):
    """Model level (the benchmark's layer shapes at B = 1024: the M <= 3072 layers run K-sliced): output, every gradient and
    every buffer bit-identical with SW["fin_in_finish"] on and off."""
    torch.manual_seed(2)
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(DEV).train()
    m.math = "f16x3"
    x = (torch.randn(1024, 243, 17, 2, device=DEV) * 0.5).clamp(-1, 1)
    tgt = torch.randn(1024, 1, 17, 3, device=DEV) * 0.3
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    res = []
    for on in (True, False):
        monkeypatch.setitem(SW, "fin_in_finish", on)
        m.load_state_dict(sd0)
        m._drop_calls = 0
        m.zero_grad(set_to_none=True)
        y = m(x)
        torch.mean(torch.norm(y - tgt, dim=3)).backward()
        res.append((y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                    {k: v.clone() for k, v in m.state_dict().items()}))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
