"""GPU parity of the step-level HIP pieces -- device batch assembly, fused mpjpe, TTA fold, fused
Adam -- against the fixtures produced by the reference's own generators.py / loss.py / torch.optim.Adam
(tests/golden/make_golden_step.py) and against oracle/step_oracle.py.  Everything goes through the C ABI.
Bit-exact for the copy / index work (batch assembly); the stated tolerance for floating point."""
import numpy as np
import pytest
import torch

from oracle import step_oracle as S
from tests.util import (GEN_CASES, GOLDEN, JOINTS_LEFT, JOINTS_RIGHT, KPS_LEFT, KPS_RIGHT, gen_case_meta,
                        load_npz_groups, load_step_dataset, mpjpe_np, rel_err)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("math_mode")]       # tests/conftest.py: once per GEMM arithmetic


ONCE = pytest.mark.single_arithmetic          # index / byte / loss / optimizer kernels: no GEMM arithmetic involved
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


def _t(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


# ---------------------------------------------------------------------------------------------------------
# batch assembly
# ---------------------------------------------------------------------------------------------------------
def _make_gen(m, cams, p3, p2, **kw):
    from videopose3d_amd.generators import ChunkedGenerator
    return ChunkedGenerator(m["batch_size"], cams if m["cams"] else None, p3, p2, m["chunk_length"], pad=m["pad"],
                            causal_shift=m["causal_shift"], shuffle=m["shuffle"], random_seed=1234,
                            augment=m["augment"], kps_left=KPS_LEFT, kps_right=KPS_RIGHT, joints_left=JOINTS_LEFT,
                            joints_right=JOINTS_RIGHT, device=DEV, **kw)


@ONCE
@pytest.mark.parametrize("name", GEN_CASES)
def test_chunked_generator_bit_exact_vs_reference(name):
    z, cams, p3, p2 = load_step_dataset()
    m = gen_case_meta(z, name)
    g = _make_gen(m, cams, p3, p2)
    assert g.num_batches == m["num_batches"] and g.num_frames() == m["num_frames"]
    assert g.augment_enabled() == m["augment"]
    b = 0
    for _ in range(2):                                   # second epoch continues the RandomState stream
        for cam, b3, b2 in g.next_epoch():
            assert b2.is_cuda and b2.dtype == torch.float32
            assert np.array_equal(_np(b2), z["%s/b2_%d" % (name, b)]), (name, b)
            assert np.array_equal(_np(b3), z["%s/b3_%d" % (name, b)])
            if m["cams"]:
                assert np.array_equal(_np(cam), z["%s/cam_%d" % (name, b)])
            else:
                assert cam is None
            b += 1
    assert b == m["n"]


@ONCE
def test_chunked_generator_endless_and_random_state():
    from videopose3d_amd.generators import ChunkedGenerator
    z, cams, p3, p2 = load_step_dataset()
    g = ChunkedGenerator(8, None, None, p2, 1, pad=2, shuffle=True, random_seed=99, augment=False, endless=True,
                         device=DEV)
    it = g.next_epoch()
    for b in range(int(z["endless/n"])):
        cam, b3, b2 = next(it)
        assert cam is None and b3 is None
        assert np.array_equal(_np(b2), z["endless/b2_%d" % b]), b
    # set_random_state / random_state (run.py:330 hands the train generator's state to the eval generator)
    a = ChunkedGenerator(8, None, None, p2, 1, pad=2, random_seed=5, device=DEV)
    c = ChunkedGenerator(8, None, None, p2, 1, pad=2, random_seed=77, device=DEV)
    c.set_random_state(np.random.RandomState(5))
    for (_, _, x), (_, _, y) in zip(a.next_epoch(), c.next_epoch()):
        assert torch.equal(x, y)
    assert c.random_state() is c.random


@ONCE
def test_chunked_generator_shards_partition_every_batch():
    from videopose3d_amd import dp
    z, cams, p3, p2 = load_step_dataset()
    m = gen_case_meta(z, "c1")
    full = [tuple(t.clone() for t in b) for b in _make_gen(m, cams, p3, p2).next_epoch()]
    world = 3
    parts = [list(_make_gen(m, cams, p3, p2, shard=(r, world)).next_epoch()) for r in range(world)]
    kept = [k for k, b in enumerate(full) if dp.shardable(b[2].shape[0], world)]
    assert len(kept) >= len(full) - 1 and all(len(p) == len(kept) for p in parts)    # only a short last batch may go
    for j_kept, k in enumerate(kept):
        cam, b3, b2 = full[k]
        for j, ref in enumerate((cam, b3, b2)):
            got = torch.cat([parts[r][j_kept][j] for r in range(world)])
            assert torch.equal(got, ref), (k, j)
        sizes = [parts[r][j_kept][2].shape[0] for r in range(world)]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == b2.shape[0] and min(sizes) >= dp.MIN_SHARD


@ONCE
def test_unchunked_generator_bit_exact_vs_reference():
    from videopose3d_amd.generators import UnchunkedGenerator
    z, cams, p3, p2 = load_step_dataset()
    u = UnchunkedGenerator(cams, p3, p2, pad=13, causal_shift=0, augment=True, kps_left=KPS_LEFT, kps_right=KPS_RIGHT,
                           joints_left=JOINTS_LEFT, joints_right=JOINTS_RIGHT, device=DEV)
    assert u.num_frames() == sum(a.shape[0] for a in p2) and u.augment_enabled()
    n = 0
    for i, (cam, b3, b2) in enumerate(u.next_epoch()):
        assert np.array_equal(_np(b2), z["unch/b2_%d" % i]) and np.array_equal(_np(b3), z["unch/b3_%d" % i])
        assert np.array_equal(_np(cam), z["unch/cam_%d" % i])
        n += 1
    assert n == len(p2)
    u = UnchunkedGenerator(None, None, p2, pad=3, causal_shift=3, augment=False, device=DEV)
    for i, (cam, b3, b2) in enumerate(u.next_epoch()):
        assert cam is None and b3 is None and np.array_equal(_np(b2), z["unch_plain/b2_%d" % i])
    u.set_augment(True)                                   # run.py toggles TTA on an existing generator
    assert next(iter(u.next_epoch()))[2].shape[0] == 2


@ONCE
def test_gather_full_size_batch_vs_oracle():
    """BASELINE-size batch (B=1024 windows of 243 frames): the device gather vs the numpy restatement."""
    from videopose3d_amd.generators import ChunkedGenerator
    rng = np.random.RandomState(0)
    lens = [300, 1500, 77, 2200, 1000]
    p2 = [rng.standard_normal((n, 17, 2)).astype(np.float32) for n in lens]
    p3 = [rng.standard_normal((n, 17, 3)).astype(np.float32) for n in lens]
    g = ChunkedGenerator(1024, None, p3, p2, 1, pad=121, shuffle=True, augment=True, kps_left=KPS_LEFT,
                         kps_right=KPS_RIGHT, joints_left=JOINTS_LEFT, joints_right=JOINTS_RIGHT, device=DEV)
    pairs = np.random.RandomState(1234).permutation(S.chunk_pairs(lens, 1, True))
    for k, (_, b3, b2) in enumerate(g.next_epoch()):
        if k in (0, g.num_batches - 1):
            _, o3, o2 = S.gather_chunks(pairs[k * 1024:(k + 1) * 1024], None, p3, p2, 1, 121, 0, KPS_LEFT, KPS_RIGHT,
                                        JOINTS_LEFT, JOINTS_RIGHT)
            assert b2.shape == (len(o2), 243, 17, 2)
            assert np.array_equal(_np(b2), o2) and np.array_equal(_np(b3), o3)


# ---------------------------------------------------------------------------------------------------------
# loss + TTA
# ---------------------------------------------------------------------------------------------------------
@ONCE
def test_mpjpe_vs_reference_golden():
    from videopose3d_amd.loss import mpjpe, weighted_mpjpe
    z = np.load(GOLDEN + "/step_loss.npz")
    for name in ("pos", "rec2d"):
        p = _t(z[name + "/p"]).requires_grad_(True)
        l = mpjpe(p, _t(z[name + "/t"]))
        l.backward()
        assert abs(float(l) - float(z[name + "/loss"])) < 1e-6 * max(1.0, abs(float(z[name + "/loss"])))
        assert rel_err(_np(p.grad), z[name + "/grad"]) < 1e-6
    assert float(_np(p.grad).max()) != 0.0
    pz = _t(z["pos/p"]).requires_grad_(True)                      # the zero-distance joint: gradient exactly 0
    mpjpe(pz, _t(z["pos/t"])).backward()
    assert np.all(_np(pz.grad)[0, 0, 0] == 0.0) and np.isfinite(_np(pz.grad)).all()
    # > 65,536 points: multi-block path + workspace
    p = _t(z["big/p"].astype(np.float32)).requires_grad_(True)
    l = mpjpe(p, _t(z["big/t"].astype(np.float32)))
    l.backward()
    g = _np(p.grad)
    assert abs(float(l) - float(z["big/loss"])) < 2e-6
    assert rel_err(g[:8], z["big/grad_head"]) < 1e-6 and rel_err(g[-8:], z["big/grad_tail"]) < 1e-6
    assert np.abs(g.astype(np.float64).sum(axis=(0, 1, 2)) - z["big/grad_sum"]).max() < 1e-6
    # weighted, J = 1, non-unit upstream gradient (run.py:358-363)
    p = _t(z["traj/p"]).requires_grad_(True)
    l = weighted_mpjpe(p, _t(z["traj/t"]), _t(z["traj/w"]))
    (3.0 * l).backward()
    assert abs(float(l) - float(z["traj/loss"])) < 1e-6
    assert rel_err(_np(p.grad), z["traj/grad3"]) < 1e-6
    # no grad requested -> loss only, no gradient buffer
    with torch.no_grad():
        assert abs(float(mpjpe(_t(z["pos/p"]), _t(z["pos/t"]))) - float(z["pos/loss"])) < 1e-6


@ONCE
def test_mpjpe_rejects_cpu_tensors():
    from videopose3d_amd import Vp3dError
    from videopose3d_amd.loss import mpjpe
    with pytest.raises(Vp3dError):
        mpjpe(torch.zeros(2, 1, 17, 3), torch.zeros(2, 1, 17, 3))


@ONCE
def test_tta_average_vs_reference_golden():
    from videopose3d_amd.generators import tta_average
    z = np.load(GOLDEN + "/step_loss.npz")
    out = tta_average(_t(z["tta/pred"]), JOINTS_LEFT, JOINTS_RIGHT)
    assert out.shape == (1, 37, 17, 3)
    assert np.abs(_np(out) - z["tta/out"]).max() < 1e-7
    assert np.array_equal(_np(out), S.tta_fold(z["tta/pred"], JOINTS_LEFT, JOINTS_RIGHT))
    out = tta_average(_t(np.ascontiguousarray(z["tta/pred"][:, :, :1])))
    assert np.abs(_np(out) - z["tta/out_traj"]).max() < 1e-7


# ---------------------------------------------------------------------------------------------------------
# Adam
# ---------------------------------------------------------------------------------------------------------
def _adam_params(z):
    return [torch.nn.Parameter(_t(z["p0_%d" % i])) for i in range(int(z["n_params"]))]


@ONCE
def test_flat_adam_vs_torch_optim_golden():
    from videopose3d_amd.optim import FlatAdam
    z = np.load(GOLDEN + "/step_adam.npz")
    params = _adam_params(z)
    opt = FlatAdam(params, lr=1e-3, amsgrad=True)
    for s in range(int(z["n_steps"])):
        opt.zero_grad()
        for i, p in enumerate(params):
            p.grad.copy_(_t(z["g%d_%d" % (s, i)]))                # grads live in the flat buffer
        v0 = params[0]._version
        opt.step()
        assert params[0]._version > v0                            # raw-pointer update is visible to version checks
        if s == 2:
            for g in opt.param_groups:
                g["lr"] *= 0.95
        for i, p in enumerate(params):
            assert np.abs(_np(p) - z["p%d_%d" % (s + 1, i)]).max() < 2e-7, (s, i)
    sd = opt.state_dict()
    for i in range(len(params)):
        st = sd["state"][i]
        assert float(st["step"]) == float(z["n_steps"])
        assert rel_err(_np(st["exp_avg"]), z["m_%d" % i]) < 1e-6
        assert rel_err(_np(st["exp_avg_sq"]), z["v_%d" % i]) < 1e-6
        assert rel_err(_np(st["max_exp_avg_sq"]), z["vmax_%d" % i]) < 1e-6


@ONCE
def test_flat_adam_checkpoint_round_trip_with_torch_adam():
    """run.py:600-608 saves optimizer.state_dict(); run.py:262-263 loads it back."""
    from videopose3d_amd.optim import FlatAdam
    z = np.load(GOLDEN + "/step_adam.npz")
    n_steps = int(z["n_steps"])

    def run(opt, params, steps):
        for s in steps:
            opt.zero_grad()
            for i, p in enumerate(params):
                if p.grad is None:
                    p.grad = _t(z["g%d_%d" % (s, i)]).clone()
                else:
                    p.grad.copy_(_t(z["g%d_%d" % (s, i)]))
            opt.step()

    # 3 steps with torch Adam -> state into FlatAdam -> 3 more; and the other way round; both == 6 torch steps (lr const)
    ref_p = _adam_params(z)
    ref = torch.optim.Adam(ref_p, lr=1e-3, amsgrad=True)
    run(ref, ref_p, range(n_steps))
    a_p = _adam_params(z)
    a = torch.optim.Adam(a_p, lr=1e-3, amsgrad=True)
    run(a, a_p, range(3))
    b_p = [torch.nn.Parameter(p.detach().clone()) for p in a_p]
    b = FlatAdam(b_p, lr=1e-3, amsgrad=True)
    b.load_state_dict(a.state_dict())
    run(b, b_p, range(3, n_steps))
    for p, q in zip(b_p, ref_p):
        assert np.abs(_np(p) - _np(q)).max() < 3e-7
    c_p = [torch.nn.Parameter(p.detach().clone()) for p in a_p]
    c = torch.optim.Adam(c_p, lr=1e-3, amsgrad=True)
    half_p = _adam_params(z)
    half = FlatAdam(half_p, lr=1e-3, amsgrad=True)
    run(half, half_p, range(3))
    c.load_state_dict(half.state_dict())
    for p, q in zip(c_p, half_p):
        p.data.copy_(q.data)
    run(c, c_p, range(3, n_steps))
    for p, q in zip(c_p, ref_p):
        assert np.abs(_np(p) - _np(q)).max() < 3e-7


def test_flat_adam_full_model_vs_oracle_and_engine_alignment():
    """arc 3,3,3,3,3 / C=1024 (16.95 M parameters): one fused step vs the numpy restatement; the re-pointed parameter
    views must keep the GEMM fast path legal (256-byte aligned slots) and the model must still run."""
    import videopose3d_amd as V
    from videopose3d_amd.optim import FlatAdam
    torch.manual_seed(0)
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.0, channels=1024).to(DEV).train()
    opt = FlatAdam(m.parameters(), lr=1e-3, amsgrad=True)
    assert all(p.data_ptr() % 256 == 0 and p.grad.data_ptr() % 256 == 0 for p in m.parameters())
    x = torch.randn(64, 243, 17, 2, device=DEV).clamp(-1, 1)
    tgt = torch.randn(64, 1, 17, 3, device=DEV) * 0.3
    p0 = {k: _np(p).copy() for k, p in m.named_parameters()}
    opt.zero_grad()
    from videopose3d_amd.loss import mpjpe
    mpjpe(m(x), tgt).backward()
    g0 = {k: _np(p.grad).copy() for k, p in m.named_parameters()}
    opt.step()
    for k, p in m.named_parameters():
        z = np.zeros_like(p0[k])
        want, _, _, _ = S.adam_step(p0[k], g0[k], z, z, z, 1)
        assert np.abs(_np(p) - want).max() < 2e-7, k
    opt.zero_grad()
    mpjpe(m(x), tgt).backward()                                   # second step runs on the updated flat views
    opt.step()
    assert all(torch.isfinite(p).all() for p in m.parameters())


def test_fused_training_loop_vs_reference_golden():
    """The run.py-style loop of tests/golden/train_loop.npz (reference model + loss.mpjpe + optim.Adam(amsgrad) with lr /
    BN-momentum decay) executed with EVERY step-level piece on the HIP path: fused mpjpe, FlatAdam, flat gradients."""
    import videopose3d_amd as V
    from videopose3d_amd.loss import mpjpe
    from videopose3d_amd.optim import FlatAdam
    g = load_npz_groups("train_loop")
    fw = [3, 3, 3]
    tr = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.0, channels=64)
    ev = V.TemporalModel(17, 2, 17, fw, dropout=0.0, channels=64)
    tr.load_state_dict({k: torch.from_numpy(v) for k, v in g["sd0"].items()})
    tr, ev = tr.to(DEV), ev.to(DEV)
    xs, ys = _t(g["xs"]), _t(g["ys"])
    lr_decay, mom0 = 0.95, 0.1
    opt = FlatAdam(tr.parameters(), lr=1e-3, amsgrad=True)
    losses = []
    tr.train()
    for i in range(6):
        opt.zero_grad()
        loss = mpjpe(tr(xs[i]), ys[i])
        loss.backward()
        opt.step()
        losses.append(float(loss))
        for grp in opt.param_groups:
            grp["lr"] *= lr_decay
        tr.set_bn_momentum(mom0 * np.exp(-(i + 1) / 6 * np.log(mom0 / 0.001)))
    assert np.abs(np.array(losses) - g["losses"]).max() < 2e-3, (losses, g["losses"])
    ev.load_state_dict(tr.state_dict())
    ev.eval()
    with torch.no_grad():
        y = ev(_t(g["x_eval"]))
    assert mpjpe_np(_np(y), g["y_eval"]) < 5e-3


# ---------------------------------------------------------------------------------------------------------
# overlapped gradient exchange: RCCL plumbing on one GPU (world 1: the all-reduce is the identity)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(300)
def test_overlapped_bucket_exchange_through_rccl_single_rank():
    import os
    import socket
    import torch.distributed as dist
    import videopose3d_amd as V
    from videopose3d_amd import dp
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
        a = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.0, channels=256).to(DEV).train()
        b = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.0, channels=256).to(DEV).train()
        b.load_state_dict(a.state_dict())
        x = torch.randn(32, 27, 17, 2, device=DEV)
        tgt = torch.randn(32, 1, 17, 3, device=DEV)
        mpjpe(a(x), tgt).backward()
        sync = dp.FlatGradSync(b.parameters(), direct_module=b, bucket_bytes=1 << 20, always_reduce=True)
        assert len(sync.buckets) >= 2
        for _ in range(2):
            sync.zero_grad()
            mpjpe(b(x), tgt).backward()
            assert len(sync._handles) == len(sync.buckets)       # every bucket was launched DURING backward
            sync.sync()
            torch.cuda.synchronize()
        for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.equal(pa.grad, pb.grad), k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("p_drop", [0.0, 0.25])
def test_graphed_train_step_equals_eager_steps(p_drop):
    """graph.GraphedTrainStep (hipGraph replay of forward + mpjpe + backward) against the same steps run eagerly: losses,
    gradients, BatchNorm running statistics, and a dropout mask that advances on every replay."""
    import copy
    import videopose3d_amd as V
    from videopose3d_amd import dp
    from videopose3d_amd import loss as vloss
    from videopose3d_amd.graph import GraphedTrainStep
    torch.manual_seed(3)
    fw = [3, 3, 3]
    m_e = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=p_drop, channels=128).to(DEV).train()
    m_g = copy.deepcopy(m_e)
    for m in (m_e, m_g):
        m._drop_seed = 4242
    m_e._drop_calls, m_g._drop_calls = 1, 0          # replay k draws offset 0 + k; the eager model starts at 1
    sync_e = dp.FlatGradSync(m_e.parameters(), direct_module=m_e)
    step = GraphedTrainStep(m_g)
    gen = torch.Generator().manual_seed(5)
    losses_e, losses_g, masks = [], [], []
    for k in range(3):
        x = (torch.randn(16, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1).to(DEV)
        tgt = (torch.randn(16, 1, 17, 3, generator=gen) * 0.3).to(DEV)
        sync_e.zero_grad()
        le = vloss.mpjpe(m_e(x), tgt)                  # the same fused loss kernel the captured step uses
        le.backward()
        lg = step(x, tgt)
        losses_e.append(float(le))
        losses_g.append(float(lg))
        assert abs(losses_e[-1] - losses_g[-1]) < 1e-6 * max(1.0, abs(losses_e[-1])), (k, losses_e, losses_g)
        for (name, pe), (_, pg) in zip(m_e.named_parameters(), m_g.named_parameters()):
            assert torch.allclose(pe.grad, pg.grad, rtol=1e-5, atol=1e-8), (k, name)
        masks.append(m_g.expand_bn.running_mean.clone())
    for (name, be), (_, bg) in zip(m_e.named_buffers(), m_g.named_buffers()):
        assert torch.allclose(be.float(), bg.float(), rtol=1e-6, atol=1e-7), name
    assert int(m_g.expand_bn.num_batches_tracked) == 3
    assert len(step._cache) == 1                       # one capture, three replays
    if p_drop > 0:
        assert len({round(v, 9) for v in losses_g}) == 3
    # a new BatchNorm momentum (run.py:590-593, every epoch) reaches the replay through device memory (vp3d_bn_finalize_dm):
    # no re-capture, no second memory pool -- whether it is set through set_bn_momentum or on the modules directly
    for k, mom in enumerate((0.05, 0.02)):
        if k == 0:
            m_g.set_bn_momentum(mom)
        else:
            for bn in [m_g.expand_bn] + list(m_g.layers_bn):
                bn.momentum = mom
        m_e.set_bn_momentum(mom)
        x = (torch.randn(16, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1).to(DEV)
        tgt = (torch.randn(16, 1, 17, 3, generator=gen) * 0.3).to(DEV)
        sync_e.zero_grad()
        vloss.mpjpe(m_e(x), tgt).backward()
        step(x, tgt)
        assert len(step._cache) == 1
        assert torch.allclose(m_e.expand_bn.running_var, m_g.expand_bn.running_var, rtol=1e-6, atol=1e-7)
        assert torch.allclose(m_e.layers_bn[3].running_mean, m_g.layers_bn[3].running_mean, rtol=1e-6, atol=1e-7)
    # per-layer DIFFERENT momenta are launch arguments again: the stale entry (and its memory pool) is dropped, one re-capture
    old_entry = next(iter(step._cache.values()))
    m_g.layers_bn[0].momentum = m_e.layers_bn[0].momentum = 0.3
    sync_e.zero_grad()
    vloss.mpjpe(m_e(x), tgt).backward()
    step(x, tgt)
    assert len(step._cache) == 1 and next(iter(step._cache.values())) is not old_entry
    assert torch.allclose(m_e.layers_bn[0].running_var, m_g.layers_bn[0].running_var, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("p_drop", [0.0, 0.25])
def test_graphed_generic_step_equals_eager_semi_supervised_step(p_drop):
    """graph.GraphedStep (hipGraph capture of an arbitrary autograd step) on run.py's semi-supervised step (run.py:322-394:
    pose + trajectory model, mpjpe + weighted mpjpe + project_to_2d back-projection + bone-length term) against the same
    steps run eagerly: losses, every gradient of both models, running statistics; masks advance on every replay."""
    import copy
    import videopose3d_amd as V
    from videopose3d_amd import loss as vloss
    from videopose3d_amd.camera import project_to_2d
    from videopose3d_amd.graph import GraphedStep
    torch.manual_seed(11)
    fw, bsz = [3, 3, 3], 8
    pos_e = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=p_drop, channels=64).to(DEV).train()
    traj_e = V.TemporalModelOptimized1f(17, 2, 1, fw, dropout=p_drop, channels=64).to(DEV).train()
    pos_g, traj_g = copy.deepcopy(pos_e), copy.deepcopy(traj_e)
    for i, (a, b) in enumerate(((pos_e, pos_g), (traj_e, traj_g))):
        a._drop_seed = b._drop_seed = 777 + i
        a._drop_calls, b._drop_calls = 1, 0            # replay k draws offset 0 + k; the eager models start at 1
    parents = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15]
    par = torch.tensor(parents[1:], device=DEV)       # (a Python-list index would be a host-to-device copy inside the capture)

    def make_fn(pos, traj):
        def fn(cat, y3, cam):
            y_traj = y3[:, :, :1]
            y_pos = y3.clone()
            y_pos[:, :, 0] = 0
            p_cat, t_cat = pos(cat), traj(cat)
            loss = vloss.mpjpe(p_cat[:bsz], y_pos) + vloss.weighted_mpjpe(t_cat[:bsz], y_traj, 1 / y_traj[:, :, :, 2])
            recon = project_to_2d(p_cat[bsz:] + t_cat[bsz:], cam)
            loss = loss + vloss.mpjpe(recon, cat[bsz:, 13:-13, :, :2].contiguous())
            dists = p_cat[:, :, 1:] - p_cat.index_select(2, par)
            bone = torch.mean(torch.norm(dists, dim=3), dim=1)
            loss = loss + torch.mean(torch.abs(torch.mean(bone[:bsz], dim=0) - torch.mean(bone[bsz:], dim=0)))
            loss.backward()
            return loss.detach()
        return fn

    eager, step = make_fn(pos_e, traj_e), GraphedStep(make_fn(pos_g, traj_g), models=(pos_g, traj_g))
    gen = torch.Generator().manual_seed(9)
    losses = []
    for k in range(3):
        cat = (torch.randn(2 * bsz, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1).to(DEV)
        y3 = (torch.randn(bsz, 1, 17, 3, generator=gen) * 0.3).to(DEV)
        y3[:, :, 0, 2] = y3[:, :, 0, 2].abs() + 3.0
        cam = torch.tensor([1.15, 1.15, 0.0, 0.0, -0.2, 0.25, 0.0, 0.0, 0.0]).repeat(bsz, 1).to(DEV)
        pos_e.zero_grad(set_to_none=True)
        traj_e.zero_grad(set_to_none=True)
        if k == 1:            # what an unmodified training loop does between steps (optimizer.zero_grad(): set_to_none=True):
            pos_g.zero_grad(set_to_none=True)      # the replay must hand the captured gradient tensors back, or
            traj_g.zero_grad(set_to_none=True)     # optimizer.step() would silently skip every parameter
        le = float(eager(cat, y3, cam))
        lg = float(step(cat, y3, cam))
        assert all(p_.grad is not None for p_ in list(pos_g.parameters()) + list(traj_g.parameters()))
        losses.append(lg)
        assert abs(le - lg) < 1e-6 * max(1.0, abs(le)), (k, le, lg)
        for me, mg in ((pos_e, pos_g), (traj_e, traj_g)):
            for (name, pe), (_, pg) in zip(me.named_parameters(), mg.named_parameters()):
                assert torch.allclose(pe.grad, pg.grad, rtol=1e-5, atol=1e-8), (k, name)
    for me, mg in ((pos_e, pos_g), (traj_e, traj_g)):
        for (name, be), (_, bg) in zip(me.named_buffers(), mg.named_buffers()):
            assert torch.allclose(be.float(), bg.float(), rtol=1e-6, atol=1e-7), name
    assert int(pos_g.expand_bn.num_batches_tracked) == 3 and len(step._cache) == 1
    if p_drop > 0:
        assert len({round(v, 9) for v in losses}) == 3


def test_batched_sequence_evaluation_equals_one_by_one():
    """generators.predict_sequences (several ragged-length videos per forward call, TTA pair folded) against the reference
    evaluation loop: one sequence + its mirrored copy per call (run.py:652-680)."""
    import videopose3d_amd as V
    from videopose3d_amd import generators as G
    rng = np.random.RandomState(3)
    lens = [40, 133, 61, 300, 29, 135, 300, 77]
    p2 = [rng.standard_normal((n, 17, 2)).astype(np.float32) * 0.5 for n in lens]
    kl, kr = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]
    jl, jr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    torch.manual_seed(1)
    m = V.TemporalModel(17, 2, 17, [3, 3, 3], channels=128).to(DEV).eval()
    pad = (m.receptive_field() - 1) // 2
    gen = G.UnchunkedGenerator(None, None, p2, pad=pad, augment=True, kps_left=kl, kps_right=kr, joints_left=jl,
                               joints_right=jr, device=DEV)
    groups = gen.length_groups(max_frames=700)
    assert sorted(i for g_ in groups for i in g_) == list(range(len(lens))) and len(groups) > 2
    got = G.predict_sequences(m, gen, max_frames=700)
    with torch.no_grad():
        for s_id, (_, _, b2) in enumerate(gen.next_epoch()):
            ref = G.tta_average(m(b2), jl, jr)[0]
            assert got[s_id].shape == ref.shape == (lens[s_id], 17, 3)
            assert float((got[s_id] - ref).abs().max()) < 2e-5, s_id
    gen.set_augment(False)
    got = G.predict_sequences(m, gen, max_frames=10 ** 6)              # everything in one call
    with torch.no_grad():
        for s_id, (_, _, b2) in enumerate(gen.next_epoch()):
            assert float((got[s_id] - m(b2)[0]).abs().max()) < 2e-5, s_id


def test_evaluation_protocol_vs_reference_golden():
    """run.py:652-705 `evaluate()` end to end -- UnchunkedGenerator (padding + test-time-augmentation pair) -> eval-mode
    TemporalModel -> un-flip / average -> mpjpe accumulated over ragged sequences -- against numbers the REFERENCE's own
    classes produced on the same model and data (tests/golden/eval_protocol.npz, make_golden_step.py eval): every
    per-sequence prediction, and Protocol #1 error within the north star's 0.1 mm, with and without TTA, one video per call
    and length-grouped batches."""
    import os
    import videopose3d_amd as V
    from videopose3d_amd import generators as G
    from videopose3d_amd import loss as vloss
    from tests.util import GOLDEN
    g = np.load(os.path.join(GOLDEN, "eval_protocol.npz"))
    n, fw = int(g["n_seq"]), [int(v) for v in g["fw"]]
    kl, kr = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]
    jl, jr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    m = V.TemporalModel(17, 2, 17, fw, channels=int(g["channels"])).to(DEV)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}, strict=True)
    m.eval()
    p2 = [g["p2_%d" % i] for i in range(n)]
    p3 = [g["p3_%d" % i] for i in range(n)]
    pad = (m.receptive_field() - 1) // 2
    for tag, augment in (("tta", True), ("plain", False)):
        gen = G.UnchunkedGenerator(None, p3, p2, pad=pad, causal_shift=0, augment=augment, kps_left=kl, kps_right=kr,
                                   joints_left=jl, joints_right=jr, device=DEV)
        total, frames = 0.0, 0
        with torch.no_grad():
            for s_id, (_, b3, b2) in enumerate(gen.next_epoch()):
                pred = m(b2)
                if gen.augment_enabled():
                    pred = G.tta_average(pred, jl, jr)
                tgt = b3.clone()
                tgt[:, :, 0] = 0
                if gen.augment_enabled():
                    tgt = tgt[:1]
                ref = torch.from_numpy(g["%s/pred_%d" % (tag, s_id)]).to(DEV)
                assert float((pred[0] - ref).abs().max()) < 2e-5, (tag, s_id)
                total += tgt.shape[0] * tgt.shape[1] * float(vloss.mpjpe(pred, tgt))
                frames += tgt.shape[0] * tgt.shape[1]
        assert frames == int(g["%s/frames" % tag])
        e1 = total / frames * 1000
        assert abs(e1 - float(g["%s/e1_mm" % tag])) < 1e-2, (tag, e1, float(g["%s/e1_mm" % tag]))      # (north star: 0.1 mm)
        batched = G.predict_sequences(m, gen, max_frames=400)                 # several videos per forward call
        for s_id in range(n):
            ref = torch.from_numpy(g["%s/pred_%d" % (tag, s_id)]).to(DEV)
            assert float((batched[s_id] - ref).abs().max()) < 3e-5, (tag, s_id)
