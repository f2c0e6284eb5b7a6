"""GPU tests of the dedicated head kernels (csrc/vp3d_head.hip: the 3*J_out-column shrink conv of reference common/model.py:33,
applied at :137 / :196, forward and whole backward) and of the forward prologue's second-stream weight packs (round 6).

The head kernels are held to the numpy oracle's conv restatement (oracle/temporal_oracle.py tconv_fwd / tconv_dgrad /
tconv_wgrad, float64) and to the general GEMM entry points they replace; the overlapped prologue to the single-launch one, bit
for bit (same kernels, another stream)."""
import numpy as np
import pytest
import torch

import videopose3d_amd as V
from oracle import temporal_oracle as O          # checker only
from videopose3d_amd import engine, ops, ops_s16 as S
from videopose3d_amd._switches import SW

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [  # rows (B * T_out), channels, 3 * J_out
    (1024, 1024, 51),      # the benchmark's head
    (64, 1024, 51),        # cfg5's batch
    (128, 1024, 3),        # the trajectory model (J_out = 1)
    (1023, 512, 45),       # ragged row blocks, 15 joints
    (37, 36, 96),          # channels % 64 != 0 (fp32 engine only), 32 joints
    (4096, 256, 51),       # the largest row count served
    (1, 64, 51),           # a single row
]


def _case(m, k, n, seed=0):
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(m, 1, k, generator=g)
    w = torch.randn(n, k, 1, generator=g) / np.sqrt(k)
    bias = torch.randn(n, generator=g) * 0.1
    gy = torch.randn(m, 1, n, generator=g) * 0.01
    return h, w, bias, gy


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "M%d_K%d_N%d" % s)
def test_head_kernels_vs_oracle(shape):
    m, k, n = shape
    assert ops.head_supported(m, k, n)
    h, w, bias, gy = _case(m, k, n)
    hd, wd, bd, gd = h.to(DEV), w.to(DEV), bias.to(DEV), gy.to(DEV)
    out = ops.head_fwd(hd, wd, bd)
    bound = S.new_bound(DEV)
    dh, ws = ops.head_bwd(gd, hd, wd, dh_bound=bound)
    dw, db = ops.head_fold(ws, m, wd)
    torch.cuda.synchronize()
    h64, w64, g64 = h.double().numpy(), w.double().numpy(), gy.double().numpy()
    out_o = O.tconv_fwd(h64, w64, bias=bias.double().numpy())
    dh_o = O.tconv_dgrad(g64, w64, 1)
    dw_o = O.tconv_wgrad(h64, g64, 1)
    db_o = g64.sum(axis=(0, 1))

    def rel(a, b):
        return float(np.abs(a.cpu().double().numpy().reshape(b.shape) - b).max() / (np.abs(b).max() + 1e-30))
    # fp32 FMA chains of K (forward), N (dh) and M (dW) terms against float64: tolerance 2e-6 of the tensor's maximum
    assert rel(out, out_o) < 2e-6
    assert rel(dh, dh_o) < 2e-6
    assert rel(dw, dw_o) < 2e-6
    assert rel(db, db_o) < 2e-6
    assert float(bound.max()) == float(dh.abs().max())           # the bound IS the maximum of what was stored


@pytest.mark.parametrize("shape", [(1024, 1024, 51), (1023, 512, 45)], ids=lambda s: "M%d_K%d_N%d" % s)
def test_head_kernels_vs_general_gemm_path(shape):
    """Against the fp32-MFMA entry points the head kernels replace (vp3d_tconv_fwd / _dgrad / _wgrad + vp3d_colsum)."""
    from videopose3d_amd.plan import ConvSpec
    m, k, n = shape
    h, w, bias, gy = _case(m, k, n, seed=1)
    hd, wd, bd, gd = h.to(DEV), w.to(DEV), bias.to(DEV), gy.to(DEV)
    spec = ConvSpec(k, n, 1, 1, 1)
    wt = ops.pack_weight(wd)
    out_g = ops.conv_fwd(hd, wt, spec, bias=bd)
    dh_g = ops.conv_dgrad(gd, wt, spec, 1)
    dw_g = ops.conv_wgrad(gd, hd, spec)
    db_g = ops.colsum(gd.view(m, n))
    out = ops.head_fwd(hd, wd, bd)
    dh, ws = ops.head_bwd(gd, hd, wd)
    dw, db = ops.head_fold(ws, m, wd)

    def rel(a, b):
        return float((a.double() - b.double().reshape(a.shape)).abs().max() / (b.double().abs().max() + 1e-30))
    assert rel(out, out_g) < 2e-6 and rel(dh, dh_g) < 2e-6 and rel(dw, dw_g) < 2e-6 and rel(db, db_g) < 2e-6


def test_head_refuses_what_it_does_not_serve():
    assert not ops.head_supported(4097, 1024, 51)        # rows
    assert not ops.head_supported(16, 1022, 51)          # K % 4
    assert not ops.head_supported(16, 8192, 51)          # K beyond the LDS staging
    assert not ops.head_supported(16, 1024, 129)         # columns
    h = torch.zeros(4097, 1, 64, device=DEV)
    w = torch.zeros(51, 64, 1, device=DEV)
    with pytest.raises(V.Vp3dError):
        ops.head_fwd(h, w, None)


def _train_step(model, x, tgt):
    model.zero_grad(set_to_none=True)
    y = model(x)
    torch.mean(torch.norm(y - tgt, dim=3)).backward()
    torch.cuda.synchronize()
    return y.detach().clone(), {k_: p.grad.detach().clone() for k_, p in model.named_parameters()}


@pytest.mark.parametrize("switch", ["head_kernels", "prologue_overlap"])
def test_step_with_and_without_the_round6_paths(switch, math_mode, monkeypatch):
    """The whole training step with the switch on and off: the head kernels against the GEMM path they replace
    (summation order differs: 1e-5 of each tensor's maximum), the second-stream weight packs against the one-launch prologue
    (the same kernels on another stream: bit-identical)."""
    if switch == "prologue_overlap" and math_mode != "f16x3":
        pytest.skip("the fused prologue belongs to the split-fp16 engine")
    torch.manual_seed(11)
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.25, channels=256).to(DEV).train()
    m._drop_seed = 0xABC
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(96, 27, 17, 2, generator=g) * 0.5).clamp(-1, 1).to(DEV)
    tgt = (torch.randn(96, 1, 17, 3, generator=g) * 0.3).to(DEV)
    state = {k_: v.clone() for k_, v in m.state_dict().items()}
    calls = m._drop_calls
    monkeypatch.setitem(SW, switch, True)
    y1, g1 = _train_step(m, x, tgt)
    m.load_state_dict(state)
    m._drop_calls = calls                                  # the same dropout stream
    monkeypatch.setitem(SW, switch, False)
    y0, g0 = _train_step(m, x, tgt)
    tol = 0.0 if switch == "prologue_overlap" else 1e-5
    assert float((y1 - y0).abs().max()) <= tol * float(y0.abs().max())
    for k_ in g0:
        assert float((g1[k_] - g0[k_]).abs().max()) <= tol * float(g0[k_].abs().max()) + 0.0, k_
