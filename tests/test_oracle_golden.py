"""Pin oracle/temporal_oracle.py against fixtures produced by the reference classes
(tests/golden/make_golden.py, which imports /root/reference/common/model.py)."""
import numpy as np
import pytest

from oracle import temporal_oracle as O
from tests.util import (golden_names, load_golden, mpjpe_np, rel_err, GOLDEN, kat_matrix_build, kat_matrix_cases,
                        load_kat_matrix, KAT_GRAD_TOL, kat_grad_cases, kat_grad_rows, kat_reference_relu_pos, load_kat_grads,
                        KAT_FULL_GRAD_TOL)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_eval_forward(name):
    g = load_golden(name)
    m = g["meta"]
    y, _, _ = O.forward(g["sd0"], g["x"], m["filter_widths"], causal=m["causal"], kind=m["kind"],
                        dense=m["dense"], training=False)
    assert y.shape == g["y_eval"].shape
    assert mpjpe_np(y, g["y_eval"]) < 1e-5          # north_star bar is 1e-3; oracle is pinned 100x tighter
    assert np.abs(y - g["y_eval"]).max() < 5e-5


@pytest.mark.parametrize("name", golden_names())
def test_oracle_train_forward_backward(name):
    g = load_golden(name)
    m = g["meta"]
    y, cache, new_running = O.forward(g["sd0"], g["x"], m["filter_widths"], causal=m["causal"], kind=m["kind"],
                                      dense=m["dense"], training=True, dropout_masks=g["masks"],
                                      momentum=m["momentum"])
    assert mpjpe_np(y, g["y_train"]) < 1e-5
    assert abs(O.mpjpe(y, g["target"]) - float(g["loss"])) < 1e-5
    for k, v in new_running.items():
        assert rel_err(v, g["sd1"][k]) < 1e-5, k
    grads = O.backward(cache, O.mpjpe_grad(y, g["target"]))
    assert set(grads) == set(g["grad"])
    for k, v in g["grad"].items():
        assert grads[k].shape == v.shape, k
        assert rel_err(grads[k], v) < 2e-4, (k, rel_err(grads[k], v))     # BASELINE.md parity bar: 1e-4..1e-3 rel


@pytest.mark.parametrize("name", golden_names())
def test_oracle_scalars(name):
    m = load_golden(name)["meta"]
    assert O.receptive_field(m["filter_widths"]) == m["rf"]
    assert O.total_causal_shift(m["filter_widths"], m["causal"], m["kind"]) == m["total_causal_shift"]


def test_oracle_float64_agrees():
    g = load_golden("str_333_c32")
    m = g["meta"]
    y64, _, _ = O.forward(g["sd0"], g["x"], m["filter_widths"], kind=m["kind"], training=False, dtype=np.float64)
    assert np.abs(y64 - g["y_eval"]).max() < 5e-5


@pytest.mark.parametrize("linear", [False, True])
def test_oracle_camera(linear):
    z = np.load(GOLDEN + "/camera.npz")
    nm = "linear" if linear else "full"
    y = O.project_to_2d(z["X"], z["cam"], linear=linear)
    assert np.abs(y - z["y_" + nm]).max() < 1e-5
    dX = O.project_to_2d_grad(z["X"], z["cam"], z["g_" + nm], linear=linear)
    assert rel_err(dX, z["dX_" + nm]) < 1e-5


# ---- oracle/torch_cpu_path.py: the reference's CPU execution path (ATen/oneDNN via torch.nn.functional), used
# ---- only as bench.py's cpu_baseline; pinned here against the same reference-generated fixtures ---------------
@pytest.mark.parametrize("name", [n for n in golden_names() if "drop" not in n])
def test_torch_cpu_path_matches_reference_golden(name):
    import torch
    from oracle import torch_cpu_path as T
    g = load_golden(name)
    m = g["meta"]
    sd = {k: torch.from_numpy(np.array(v)) for k, v in g["sd0"].items()}
    x, tgt = torch.from_numpy(g["x"]), torch.from_numpy(g["target"])
    kw = dict(kind=m["kind"], causal=m["causal"], dense=m["dense"])
    with torch.no_grad():
        y = T.forward(sd, x, m["filter_widths"], training=False, **kw)
    assert np.abs(y.numpy() - g["y_eval"]).max() < 1e-5
    loss, yt, grads = T.train_step(sd, x, tgt, m["filter_widths"], momentum=m["momentum"], **kw)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert np.abs(yt.numpy() - g["y_train"]).max() < 1e-5
    for k, v in g["sd1"].items():
        if "num_batches" not in k:
            assert rel_err(sd[k].numpy(), v) < 1e-5, k
    for k, v in g["grad"].items():
        assert rel_err(grads[k].numpy(), v) < 1e-4, k


def test_oracle_evaluation_protocol_golden():
    """The oracle's eval forward composed as run.py:652-705 composes the reference (edge padding of generators.py:219-224, the
    test-time-augmentation pair and its fold, mpjpe over ragged sequences) against tests/golden/eval_protocol.npz, which the
    reference's own UnchunkedGenerator / TemporalModel / mpjpe produced."""
    import os
    g = np.load(os.path.join(GOLDEN, "eval_protocol.npz"))
    n, fw = int(g["n_seq"]), [int(v) for v in g["fw"]]
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd/")}
    kl, kr = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]
    jl, jr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    pad = (O.receptive_field(fw) - 1) // 2
    total, frames = 0.0, 0
    for i in range(n):
        x = np.pad(g["p2_%d" % i], ((pad, pad), (0, 0), (0, 0)), "edge")[None]
        xf = x.copy()
        xf[..., 0] *= -1
        xf[:, :, kl + kr] = xf[:, :, kr + kl]
        y, _, _ = O.forward(sd, np.concatenate([x, xf]), fw, kind="dilated", training=False)
        y[1, :, :, 0] *= -1
        y[1, :, jl + jr] = y[1, :, jr + jl]
        pred = y.mean(0)
        assert np.abs(pred - g["tta/pred_%d" % i]).max() < 2e-5, i
        tgt = g["p3_%d" % i].copy()
        tgt[:, 0] = 0
        total += tgt.shape[0] * mpjpe_np(pred, tgt)
        frames += tgt.shape[0]
    assert frames == int(g["tta/frames"])
    assert abs(total / frames * 1000 - float(g["tta/e1_mm"])) < 1e-2


@pytest.mark.parametrize("case", [c for c in kat_matrix_cases() if c["seed"] == 0], ids=lambda c: c["name"])
def test_oracle_kat_matrix_c1024(case):
    """SURVEY 8(c)'s matrix at the benchmark width: the oracle, fed the seed-built state (the package's constructors draw
    the reference's parameters: tests/test_host_cpu.py), against what the reference classes produced from the same
    recipe -- outputs entry-wise, gradients and running statistics through norm + a seeded projection
    (tests/util.py: why that tolerance is 1e-2).  Seed 0 here (CPU time); the GPU suite runs all three seeds."""
    import torch
    import videopose3d_amd as V
    ref = load_kat_matrix()[case["name"]]
    model, x_eval, x_train, target, proj = kat_matrix_build(case, V.TemporalModel, V.TemporalModelOptimized1f)
    assert abs(float(x_eval.double().sum()) - ref["x_sums"][0]) < 1e-6          # same RNG stream as the generator run
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    kw = dict(causal=case["causal"], kind=case["kind"])
    y, _, _ = O.forward(sd, x_eval.numpy(), case["filter_widths"], training=False, **kw)
    assert y.shape == ref["y_eval"].shape and mpjpe_np(y, ref["y_eval"]) < 1e-5
    y, cache, running = O.forward(sd, x_train.numpy(), case["filter_widths"], training=True, dropout_masks=None,
                                  momentum=0.1, **kw)
    assert mpjpe_np(y, ref["y_train"]) < 1e-5
    assert abs(O.mpjpe(y, target.numpy()) - float(ref["loss"])) < 1e-5
    grads = O.backward(cache, O.mpjpe_grad(y, target.numpy()))
    for k, g in grads.items():
        norm, pr = ref["grad/" + k]
        assert abs(np.linalg.norm(g.astype(np.float64)) - norm) < KAT_GRAD_TOL * norm, k
        assert abs(float((g.astype(np.float64) * proj[k].double().numpy()).sum()) - pr) < KAT_GRAD_TOL * norm, k
    for k, v in running.items():
        norm, pr = ref["stat/" + k]
        assert abs(float((np.asarray(v, np.float64) * proj[k].double().numpy()).sum()) - pr) < 1e-5 * max(norm, 1.0), k


@pytest.mark.parametrize("case", kat_grad_cases(), ids=lambda c: c["name"])
def test_oracle_full_gradients_c1024_vs_reference(case):
    """Tensor-level gradients at the benchmark width against the reference's own run (tests/golden/kat_grads.npz: BatchNorm /
    shrink / expand-conv gradients whole, every C x C weight gradient as 64 seeded rows), at 5e-4 of each tensor's maximum --
    the bar of the small fixtures, NOT the 1e-2 norm / projection bar of the matrix.  What made that bar necessary is handled
    explicitly: the fixture also holds the reference's ReLU decisions wherever its pre-activation is within 1e-4 of zero; the
    oracle's own decisions are counted against them (flips), then pinned to the reference's for the comparison."""
    import videopose3d_amd as V
    ref = load_kat_grads()[case["name"]]
    model, _, x_train, target, _ = kat_matrix_build(case, V.TemporalModel, V.TemporalModelOptimized1f)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    y, cache, _ = O.forward(sd, x_train.numpy(), case["filter_widths"], training=True, dropout_masks=None, momentum=0.1,
                            causal=case["causal"], kind=case["kind"])
    assert abs(O.mpjpe(y, target.numpy()) - float(ref["loss"])) < 1e-5
    n_layers = len(cache["layers"])
    for i, L in enumerate(cache["layers"]):          # the oracle's pre-activations agree with the reference's where those are tiny
        z = L["z"].reshape(-1)[ref["near_idx/%d" % i]]
        assert np.abs(z - ref["near_z/%d" % i]).max() < 2e-5, i
    own = [L["z"] > 0 for L in cache["layers"]]
    pos, flips = kat_reference_relu_pos(own, ref, n_layers)
    total = sum(p.size for p in own)
    assert flips <= max(3, 2e-6 * total), (flips, total)
    grads = O.backward(cache, O.mpjpe_grad(y, target.numpy()), relu_pos=pos)
    for k, g in grads.items():
        rows = kat_grad_rows(k, g.shape)
        got = g if rows is None else g[rows.numpy()]
        assert rel_err(got, ref["grad/" + k]) < KAT_FULL_GRAD_TOL, (k, rel_err(got, ref["grad/" + k]), flips)
