"""Pin oracle/step_oracle.py (batch assembly, TTA fold, Adam) and the loss part of temporal_oracle.py against
fixtures produced by the reference's own generators.py / loss.py and by torch.optim.Adam
(tests/golden/make_golden_step.py).  CPU only."""
import numpy as np
import pytest

from oracle import step_oracle as S
from oracle import temporal_oracle as O
from tests.util import (GEN_CASES, GOLDEN, JOINTS_LEFT, JOINTS_RIGHT, KPS_LEFT, KPS_RIGHT, gen_case_meta,
                        load_step_dataset, rel_err)


def _epoch_orders(pairs, m, n_epochs, seed=1234):
    rs = np.random.RandomState(seed)
    for _ in range(n_epochs):
        yield rs.permutation(pairs) if m["shuffle"] else pairs


@pytest.mark.parametrize("name", GEN_CASES)
def test_chunked_batches_bit_exact(name):
    z, cams, p3, p2 = load_step_dataset()
    m = gen_case_meta(z, name)
    pairs = S.chunk_pairs([a.shape[0] for a in p2], m["chunk_length"], m["augment"])
    assert (len(pairs) + m["batch_size"] - 1) // m["batch_size"] == m["num_batches"]
    b = 0
    for order in _epoch_orders(pairs, m, 2):
        for k in range(m["num_batches"]):
            chunk = order[k * m["batch_size"]:(k + 1) * m["batch_size"]]
            cam, b3, b2 = S.gather_chunks(chunk, cams if m["cams"] else None, p3, p2, m["chunk_length"], m["pad"],
                                          m["causal_shift"], KPS_LEFT, KPS_RIGHT, JOINTS_LEFT, JOINTS_RIGHT)
            assert np.array_equal(b2, z["%s/b2_%d" % (name, b)])
            assert np.array_equal(b3, z["%s/b3_%d" % (name, b)])
            if m["cams"]:
                assert np.array_equal(cam, z["%s/cam_%d" % (name, b)])
            b += 1
    assert b == m["n"]


def test_unchunked_batches_bit_exact():
    z, cams, p3, p2 = load_step_dataset()
    for i in range(len(p2)):
        cam, b3, b2 = S.unchunked_batch(i, cams, p3, p2, 13, 0, True, KPS_LEFT, KPS_RIGHT, JOINTS_LEFT, JOINTS_RIGHT)
        assert np.array_equal(b2, z["unch/b2_%d" % i]) and np.array_equal(b3, z["unch/b3_%d" % i])
        assert np.array_equal(cam, z["unch/cam_%d" % i])
        _, _, b2 = S.unchunked_batch(i, None, None, p2, 3, 3, False)
        assert np.array_equal(b2, z["unch_plain/b2_%d" % i])


def test_loss_and_tta_oracle():
    z = np.load(GOLDEN + "/step_loss.npz")
    for name in ("pos", "rec2d"):
        p, t = z[name + "/p"], z[name + "/t"]
        assert abs(O.mpjpe(p, t) - float(z[name + "/loss"])) < 1e-6
        assert rel_err(O.mpjpe_grad(p, t), z[name + "/grad"]) < 1e-5
    p, t = z["big/p"].astype(np.float32), z["big/t"].astype(np.float32)
    assert abs(O.mpjpe(p, t) - float(z["big/loss"])) < 1e-5
    g = O.mpjpe_grad(p, t)
    assert rel_err(g[:8], z["big/grad_head"]) < 1e-5 and rel_err(g[-8:], z["big/grad_tail"]) < 1e-5
    p, t, w = z["traj/p"], z["traj/t"], z["traj/w"]
    assert abs(O.mpjpe(p, t, w) - float(z["traj/loss"])) < 1e-6
    assert rel_err(3.0 * O.mpjpe_grad(p, t, w), z["traj/grad3"]) < 1e-5
    assert np.allclose(S.tta_fold(z["tta/pred"], JOINTS_LEFT, JOINTS_RIGHT), z["tta/out"], rtol=0, atol=1e-7)
    assert np.allclose(S.tta_fold(z["tta/pred"][:, :, :1]), z["tta/out_traj"], rtol=0, atol=1e-7)


def test_adam_oracle_vs_torch_optim():
    z = np.load(GOLDEN + "/step_adam.npz")
    n, steps = int(z["n_params"]), int(z["n_steps"])
    lr = 1e-3
    for i in range(n):
        p = z["p0_%d" % i].copy()
        m, v, vmax = np.zeros_like(p), np.zeros_like(p), np.zeros_like(p)
        cur = lr
        for s in range(steps):
            p, m, v, vmax = S.adam_step(p, z["g%d_%d" % (s, i)], m, v, vmax, s + 1, lr=cur)
            if s == 2:
                cur *= 0.95
            assert np.abs(p - z["p%d_%d" % (s + 1, i)]).max() < 2e-7, (i, s)
        assert rel_err(m, z["m_%d" % i]) < 1e-6 and rel_err(v, z["v_%d" % i]) < 1e-6
        assert rel_err(vmax, z["vmax_%d" % i]) < 1e-6
