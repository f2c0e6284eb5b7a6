"""GPU tests of the BatchNorm-backward column sums fused into the dgrad launches of the split-fp16 training backward
(vp3d_s16_red, csrc/vp3d_gemm_s16.hip k_nt_s16<.., RED>; autograd of reference common/model.py:134 / :193
drop(relu(bn(conv(x))))) against the separate reduction pass it replaces (VP3D_FUSE_BN_RED=0) and against the oracle.  The
model-level parity suite runs through the fused launches wherever engine_s16 selects them (activations of >= 16,384 rows)."""
import numpy as np
import pytest
import torch

import videopose3d_amd as V
from videopose3d_amd import engine, graph, ops_s16 as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _step(model, x, tgt, monkeypatch, mode):
    monkeypatch.setenv("VP3D_FUSE_BN_RED", mode)
    model.zero_grad(set_to_none=True)
    n0 = S.RED_CALLS["n"]
    y = model(x)
    torch.mean(torch.norm(y - tgt, dim=3)).backward()
    torch.cuda.synchronize()
    return y.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, S.RED_CALLS["n"] - n0


def _build(cls, b, fw, c, causal, p, extra_t=0):
    torch.manual_seed(3)
    m = cls(17, 2, 17, fw, causal=causal, dropout=p, channels=c).to(DEV).train()
    m.math = "f16x3"
    m._drop_seed = 0x7ED
    rf = m.receptive_field()
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(b, rf + extra_t, 17, 2, generator=gen) * 0.5).clamp(-1, 1).to(DEV)
    tgt = (torch.randn(b, 1 + extra_t, 17, 3, generator=gen) * 0.3).to(DEV)
    return m, x, tgt


CASES = [  # class, batch, arc, channels, causal, dropout, extra frames
    ("strided", 512, [3, 3, 3], 256, False, 0.25, 0),         # 128x128 tiles, whole rounds
    ("strided", 300, [3, 3, 3, 3], 256, True, 0.0, 0),        # causal residual tap, ragged row tiles (8100 / 2700 / 900 rows)
    ("strided", 1024, [3, 3, 3], 512, False, 0.1, 0),         # 256x256 tiles with the residual window in the middle tap
    ("strided", 96, [3, 1, 3], 256, False, 0.25, 0),          # a 1-tap block
    ("dilated", 24, [3, 3, 3], 256, False, 0.25, 101),        # stride-1 convs: gather-form dgrad, overlapping windows
    ("dilated", 16, [3, 3], 256, True, 0.25, 150),            # ... causal
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s_B%d_arc%s_C%d%s_p%g" % (c[0], c[1], "".join(map(str, c[2])), c[3],
                                                                                 "_causal" if c[4] else "", c[5]))
def test_fused_sums_equal_separate_pass(case, monkeypatch):
    """Same weights, batch and dropout stream: the step with every supported dgrad launch carrying the sums (=1) against the
    step with the separate reduction pass (=0).  The two differ in the order of the fp32 partial sums only."""
    kind, b, fw, c, causal, p, extra = case
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    try:
        m, x, tgt = _build(V.TemporalModelOptimized1f if kind == "strided" else V.TemporalModel, b, fw, c, causal, p, extra)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        calls0 = m._drop_calls
        y1, g1, n1 = _step(m, x, tgt, monkeypatch, "1")
        if n1 == 0:
            pytest.skip("the planner slices K (or picks another tiling) for every dgrad of this shape: nothing to fuse")
        m.load_state_dict(sd0)
        m._drop_calls = calls0
        y0, g0, n0 = _step(m, x, tgt, monkeypatch, "0")
        assert n0 == 0
        assert torch.equal(y1, y0)                       # (the forward is the same code)
        for k in g0:
            assert _rel(g1[k], g0[k]) < 2e-5, (k, _rel(g1[k], g0[k]), n1)
        # bit-reproducible: fixed summation order, no floating-point atomics
        m.load_state_dict(sd0)
        m._drop_calls = calls0
        y2, g2, n2 = _step(m, x, tgt, monkeypatch, "1")
        assert n2 == n1
        for k in g1:
            assert torch.equal(g1[k], g2[k]), k
    finally:
        engine.S16_MIN_FORWARD_FLOPS.update(keep)


def test_fused_sums_step_vs_oracle(monkeypatch):
    """One training step with the fused sums directly against the numpy oracle (output, every gradient) -- not only through
    the separate-pass path."""
    from oracle import temporal_oracle as O
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    monkeypatch.setenv("VP3D_FUSE_BN_RED", "1")
    try:
        fw, c, b = [3, 3, 3], 256, 320
        torch.manual_seed(6)
        m = V.TemporalModelOptimized1f(17, 2, 17, fw, causal=True, dropout=0.0, channels=c).train()
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        m = m.to(DEV).train()
        m.math = "f16x3"
        gen = torch.Generator().manual_seed(8)
        x = (torch.randn(b, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
        tgt = torch.randn(b, 1, 17, 3, generator=gen) * 0.3
        n0 = S.RED_CALLS["n"]
        y = m(x.to(DEV))
        torch.mean(torch.norm(y - tgt.to(DEV), dim=3)).backward()
        torch.cuda.synchronize()
        assert S.RED_CALLS["n"] > n0, "no dgrad launch carried the sums: pick a shape the planner does not slice"
        yo, cache, running = O.forward(sd, x.numpy(), fw, causal=True, kind="strided", training=True)
        go = O.backward(cache, O.mpjpe_grad(yo, tgt.numpy()))
        assert float(np.abs(y.detach().cpu().numpy() - yo).max()) < 1e-4
        for k, p in m.named_parameters():
            g = p.grad.cpu().numpy()
            assert float(np.abs(g - go[k]).max() / (np.abs(go[k]).max() + 1e-12)) < 5e-4, k
    finally:
        engine.S16_MIN_FORWARD_FLOPS.update(keep)


def test_fused_sums_under_graph_replay(monkeypatch):
    """GraphedTrainStep captures the fused launches (tickets left zero, partial rows in the graph's pool): replays equal eager."""
    from videopose3d_amd import dp, loss as vloss
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    monkeypatch.setenv("VP3D_FUSE_BN_RED", "1")
    try:
        m, x, tgt = _build(V.TemporalModelOptimized1f, 512, [3, 3, 3], 256, False, 0.0)
        sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        sync.zero_grad()
        n0 = S.RED_CALLS["n"]
        vloss.mpjpe(m(x), tgt).backward()
        sync.sync()
        torch.cuda.synchronize()
        assert S.RED_CALLS["n"] > n0
        eager = sync.flat.clone()
        m.load_state_dict(sd0)
        gs = graph.GraphedTrainStep(m, sync=sync)
        for _ in range(3):
            m.load_state_dict(sd0)
            gs(x, tgt)
        torch.cuda.synchronize()
        assert _rel(sync.flat, eager) < 1e-6
    finally:
        engine.S16_MIN_FORWARD_FLOPS.update(keep)
