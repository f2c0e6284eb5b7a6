"""GPU tests of the persistent small-M tail (csrc/vp3d_tail_s16.hip: the trailing blocks of the strided training stack as one
launch per direction; reference common/model.py:190-196 + autograd) against the per-layer launches it replaces (VP3D_TAIL=0)
and, through them, against everything the parity suite holds the per-layer path to.  The model-level parity tests
(tests/test_gpu_parity.py) run THROUGH the tail as well whenever engine_s16.tail_from() selects it."""
import numpy as np
import pytest
import torch

import videopose3d_amd as V
from videopose3d_amd import engine, engine_s16, ops_s16 as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _needs_the_experiments_build():
    """The persistent tail is compiled only with VP3D_BUILD_EXPERIMENTS=1 (measured slower than the per-layer launches at every
    benchmark shape: DESIGN.md 4.8); the default library has nothing here to certify."""
    from videopose3d_amd import _lib
    if not _lib.lib().vp3d_has_experiments():
        pytest.skip("library built without VP3D_BUILD_EXPERIMENTS: no persistent tail")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _step(model, x, tgt, monkeypatch, tail):
    monkeypatch.setenv("VP3D_TAIL", "1" if tail else "0")
    model.zero_grad(set_to_none=True)
    before = dict(S.TAIL_CALLS)
    y = model(x)
    torch.mean(torch.norm(y - tgt, dim=3)).backward()
    torch.cuda.synchronize()
    ran = (S.TAIL_CALLS["fwd"] - before["fwd"], S.TAIL_CALLS["bwd"] - before["bwd"])
    return y.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, \
        {k: v.detach().clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}, ran


CASES = [  # batch, arc, channels, causal, dropout
    (64, [3, 3, 3], 128, False, 0.25),            # every block in the tail (tail starts at conv 1: the expand output's transposed copy)
    (48, [3, 3, 3, 3], 64, True, 0.0),            # causal residual tap (start = 2), no dropout
    (200, [3, 3, 3, 3, 3], 128, False, 0.25),     # first block outside the tail (200 * 27 rows), ragged 128-row tiles (M = 200)
    (1024, [3, 3, 3], 256, False, 0.1),           # M = 3072 and 1024: the benchmark's tail shapes at a quarter of its width
    (40, [3, 1, 3], 64, False, 0.25),             # a 1-tap "strided" block inside the tail
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_arc%s_C%d%s_p%g" % (c[0], "".join(map(str, c[1])), c[2], "_causal" if c[3] else "", c[4]))
def test_tail_equals_per_layer_launches(case, monkeypatch):
    """Same weights, same batch, same dropout stream: the step through the persistent tail against the step through the
    per-layer kernels -- output, every gradient, running statistics.  Differences: the order of the BatchNorm sums (exact
    fp64 two-pass here, 64-row slabs + Chan merge there) and of the K-slices; both are 1e-6-class."""
    b, fw, c, causal, p = case
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    try:
        torch.manual_seed(3)
        m = V.TemporalModelOptimized1f(17, 2, 17, fw, causal=causal, dropout=p, channels=c).to(DEV).train()
        m.math = "f16x3"
        m._drop_seed = 0x7A11
        rf = m.receptive_field()
        gen = torch.Generator().manual_seed(5)
        x = (torch.randn(b, rf, 17, 2, generator=gen) * 0.5).clamp(-1, 1).to(DEV)
        tgt = (torch.randn(b, 1, 17, 3, generator=gen) * 0.3).to(DEV)
        monkeypatch.setenv("VP3D_TAIL", "1")
        assert engine_s16.tail_from(m, m._plan, rf, b, None, True) > 0
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        calls0 = m._drop_calls
        y_t, g_t, st_t, ran_t = _step(m, x, tgt, monkeypatch, True)
        assert ran_t == (1, 1), ran_t
        assert S.tail_error(DEV) == 0, "a grid barrier of the persistent kernel timed out"
        m.load_state_dict(sd0)
        m._drop_calls = calls0                         # the same masks
        y_r, g_r, st_r, ran_r = _step(m, x, tgt, monkeypatch, False)
        assert ran_r == (0, 0)
        assert float((y_t - y_r).abs().max()) < 2e-5
        for k in st_r:
            assert _rel(st_t[k].float(), st_r[k].float()) < 1e-5, k
        for k in g_r:
            assert _rel(g_t[k], g_r[k]) < 2e-4, (k, _rel(g_t[k], g_r[k]))
    finally:
        engine.S16_MIN_FORWARD_FLOPS.update(keep)


def test_tail_forward_only_and_eval_agree(monkeypatch):
    """Training-mode forward under torch.no_grad() (nothing saved: no bits, no transposed copies) through the tail."""
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    try:
        torch.manual_seed(4)
        m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.0, channels=128).to(DEV).train()
        m.math = "f16x3"
        x = (torch.randn(32, 27, 17, 2, device=DEV) * 0.5).clamp(-1, 1)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        outs = []
        for tail in ("1", "0"):
            monkeypatch.setenv("VP3D_TAIL", tail)
            m.load_state_dict(sd0)
            with torch.no_grad():
                outs.append(m(x).clone())
        assert S.tail_error(DEV) == 0
        assert float((outs[0] - outs[1]).abs().max()) < 2e-5
    finally:
        engine.S16_MIN_FORWARD_FLOPS.update(keep)


def test_tail_step_vs_oracle(monkeypatch):
    """One training step through the persistent tail directly against the numpy oracle (output, every gradient, running
    statistics) -- not only through the per-layer path."""
    from oracle import temporal_oracle as O
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    monkeypatch.setenv("VP3D_TAIL", "1")
    try:
        torch.manual_seed(6)
        fw = [3, 3, 3, 3]
        m = V.TemporalModelOptimized1f(17, 2, 17, fw, causal=True, dropout=0.0, channels=128)
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        m = m.to(DEV).train()
        m.math = "f16x3"
        gen = torch.Generator().manual_seed(8)
        x = (torch.randn(24, 81, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
        tgt = torch.randn(24, 1, 17, 3, generator=gen) * 0.3
        before = dict(S.TAIL_CALLS)
        y = m(x.to(DEV))
        torch.mean(torch.norm(y - tgt.to(DEV), dim=3)).backward()
        assert S.TAIL_CALLS["fwd"] == before["fwd"] + 1 and S.TAIL_CALLS["bwd"] == before["bwd"] + 1
        assert S.tail_error(DEV) == 0
        yo, cache, running = O.forward(sd, x.numpy(), fw, causal=True, kind="strided", training=True)
        go = O.backward(cache, O.mpjpe_grad(yo, tgt.numpy()))
        assert float(np.abs(y.detach().cpu().numpy() - yo).max()) < 1e-4
        for k, p in m.named_parameters():
            g = p.grad.cpu().numpy()
            assert float(np.abs(g - go[k]).max() / (np.abs(go[k]).max() + 1e-12)) < 5e-4, k
        sd1 = m.state_dict()
        for k, v in running.items():
            assert float(np.abs(sd1[k].cpu().numpy() - v).max() / (np.abs(v).max() + 1e-12)) < 1e-4, k
    finally:
        engine.S16_MIN_FORWARD_FLOPS.update(keep)


def test_flat_barrier_fallback_in_a_fresh_process():
    """The barrier flavour is chosen once per process and device (probe launch); VP3D_TAIL_FLAT_BARRIER=1 forces the fallback in
    which every workgroup does its own cache write-back / invalidate: one parity case through it, in a process of its own."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VP3D_TAIL_FLAT_BARRIER="1")
    code = ("import videopose3d_amd as V\n"
            "from videopose3d_amd import _lib\n"
            "import pytest, sys\n"
            "rc = pytest.main(['-q', '-x', '-m', 'gpu', 'tests/test_gpu_tail.py', '-k', 'B64_arc333 or step_vs_oracle'])\n"
            "assert _lib.lib().vp3d_tail_barrier_grouped() == 0\n"
            "sys.exit(int(rc))\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-500:]
