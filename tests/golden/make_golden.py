#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE classes (imported from /root/reference) on CPU.

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

Writes tests/golden/*.npz (small-channel full-tensor fixtures) and tests/golden/kat_c1024.json
(known-answer checksums for seed-initialised C=1024 models) and tests/golden/kat_matrix.npz (the C=1024 seed x arc x
causal x class matrix: outputs in full, gradients and running statistics as norm + seeded projection).  The fixtures pin
oracle/temporal_oracle.py (tests/test_oracle_golden.py) and, on the GPU, the HIP path
(tests/test_gpu_*.py).  Nothing here is copied from the reference: it is only *executed*.
"""
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("VP3D_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
from common.model import TemporalModel, TemporalModelOptimized1f  # noqa: E402  (reference, executed only)
from common.loss import mpjpe, weighted_mpjpe  # noqa: E402
from common.camera import project_to_2d, project_to_2d_linear  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def randomise_bn(model, gen):
    """Non-trivial BN affine + running stats so that BN folding / eval is really exercised."""
    for name, buf in model.state_dict().items():
        if name.endswith("running_mean"):
            buf.copy_(torch.randn(buf.shape, generator=gen) * 0.1)
        elif name.endswith("running_var"):
            buf.copy_(torch.rand(buf.shape, generator=gen) * 1.5 + 0.5)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "bn" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif "bn" in name and name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))


def sd_numpy(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def make_case(name, kind, fw, causal, channels, j_in=17, feat=2, j_out=17, dense=False, seed=0,
              batch=3, extra_t=0, dropout=0.0, momentum=0.1):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(1000 + seed)
    if kind == "dilated":
        model = TemporalModel(j_in, feat, j_out, fw, causal=causal, dropout=dropout, channels=channels, dense=dense)
    else:
        model = TemporalModelOptimized1f(j_in, feat, j_out, fw, causal=causal, dropout=dropout, channels=channels)
    randomise_bn(model, gen)
    model.set_bn_momentum(momentum)
    rf = model.receptive_field()
    t_in = rf + (extra_t if kind == "dilated" else 0)
    x = (torch.randn(batch, t_in, j_in, feat, generator=gen) * 0.5).clamp(-1, 1)
    out = {"meta": json.dumps(dict(name=name, kind=kind, filter_widths=list(fw), causal=bool(causal),
                                   channels=channels, j_in=j_in, feat=feat, j_out=j_out, dense=bool(dense),
                                   seed=seed, dropout=dropout, momentum=momentum, rf=rf,
                                   total_causal_shift=int(model.total_causal_shift()))),
           "x": x.numpy()}
    for k, v in sd_numpy(model).items():
        out["sd0/" + k] = v

    # eval forward
    model.eval()
    with torch.no_grad():
        y_eval = model(x)
    out["y_eval"] = y_eval.numpy().copy()

    # train forward + backward (capture dropout masks through a hook on the shared Dropout module)
    model.train()
    masks = []

    def hook(mod, inp, outp):
        i, o = inp[0].detach(), outp.detach()
        mk = torch.where(i > 0, o / torch.where(i > 0, i, torch.ones_like(i)), torch.full_like(i, float("nan")))
        masks.append(mk.permute(0, 2, 1).contiguous().numpy())   # NCL -> NLC

    hnd = model.drop.register_forward_hook(hook)
    y_train = model(x)
    hnd.remove()
    target = torch.randn(y_train.shape, generator=gen) * 0.3
    target[:, :, 0] = 0
    loss = mpjpe(y_train, target)
    loss.backward()
    out["y_train"] = y_train.detach().numpy().copy()
    out["target"] = target.numpy()
    out["loss"] = np.float64(loss.item())
    for k, v in sd_numpy(model).items():
        if "running" in k or "num_batches" in k:
            out["sd1/" + k] = v
    for k, p in model.named_parameters():
        out["grad/" + k] = p.grad.numpy().copy()
    if dropout > 0:
        for i, mk in enumerate(masks):
            # where relu output was 0 the mask is unobservable (and irrelevant): store keep=1 scale there
            scale = 1.0 / (1.0 - dropout)
            mk = np.where(np.isnan(mk), scale, mk)
            out["mask/%d" % i] = (mk > 0).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "rf", rf, "loss %.6f" % loss.item())


def make_kat():
    """Seed-only known answers at C=1024 (weights are not stored: they follow from torch.manual_seed)."""
    kats = []
    for kind, fw, causal, b in (("dilated", [3, 3, 3], False, 4), ("strided", [3, 3, 3], False, 4),
                                ("dilated", [3, 3, 3, 3, 3], False, 2), ("strided", [3, 3, 3, 3, 3], True, 2)):
        torch.manual_seed(0)
        if kind == "dilated":
            model = TemporalModel(17, 2, 17, fw, causal=causal, channels=1024)
        else:
            model = TemporalModelOptimized1f(17, 2, 17, fw, causal=causal, channels=1024)
        model.eval()
        rf = model.receptive_field()
        x = torch.randn(b, rf, 17, 2)
        with torch.no_grad():
            y = model(x)
        n_params = sum(p.numel() for p in model.parameters())
        kats.append(dict(kind=kind, filter_widths=fw, causal=causal, batch=b, rf=rf, n_params=n_params,
                         x_sum=float(x.double().sum()), y_sum=float(y.double().sum()),
                         y_abs_sum=float(y.double().abs().sum()),
                         y_first=[float(v) for v in y.flatten()[:6]],
                         expand_w_sum=float(model.expand_conv.weight.double().sum()),
                         shrink_w_sum=float(model.shrink.weight.double().sum())))
        print("kat", kind, fw, "y_sum", kats[-1]["y_sum"], "params", n_params)
    with open(os.path.join(HERE, "kat_c1024.json"), "w") as f:
        json.dump(kats, f, indent=1)


def make_kat_matrix():
    """seeds {0,1,2} x arcs {3,3,3 / 3,3,3,3,3} x causal x both classes at C = 1024 (SURVEY 8c's matrix): eval output
    (T = RF+57 for the dilated class), train-mode output + loss with dropout 0, and norm + one seeded projection of every
    parameter gradient and post-step running statistic.  tests/util.py holds the recipe both sides follow."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests.util import kat_matrix_build, kat_matrix_cases, kat_matrix_summaries
    out = {}
    for case in kat_matrix_cases():
        model, x_eval, x_train, target, proj = kat_matrix_build(case, TemporalModel, TemporalModelOptimized1f)
        n = case["name"]
        model.eval()
        with torch.no_grad():
            out[n + "|y_eval"] = model(x_eval).numpy().copy()
        model.train()
        y = model(x_train)
        loss = mpjpe(y, target)
        loss.backward()
        out[n + "|y_train"] = y.detach().numpy().copy()
        out[n + "|loss"] = np.float64(loss.item())
        out[n + "|x_sums"] = np.array([x_eval.double().sum(), x_train.double().sum(), target.double().sum()])
        for k, v in kat_matrix_summaries(model, proj).items():
            out[n + "|" + k] = np.array(v, np.float64)
        print("kat_matrix", n, "loss %.6f" % loss.item())
    np.savez_compressed(os.path.join(HERE, "kat_matrix.npz"), **out)


def make_kat_grads():
    """FULL reference gradients of four C = 1024 training steps (both arcs x causal, strided class, seed 0: the kat_matrix
    recipe of tests/util.py): BatchNorm / shrink / expand-conv gradients whole, every C x C weight gradient as 64 seeded rows
    (tests/util.kat_grad_rows), and the reference's ReLU decisions wherever its pre-activation is within 1e-4 of zero -- the
    only places where another fp32 evaluation of the same step can legitimately decide differently.  The consumers count
    such flips explicitly instead of widening a tolerance."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests.util import kat_matrix_build, kat_grad_cases, kat_grad_rows, KAT_NEAR_ZERO
    out = {}
    for case in kat_grad_cases():
        model, _, x_train, target, _ = kat_matrix_build(case, TemporalModel, TemporalModelOptimized1f)
        n = case["name"]
        bns = [model.expand_bn] + list(model.layers_bn)
        zs = [None] * len(bns)
        hooks = []
        for i, bn in enumerate(bns):     # (the hook fires before the in-place ReLU overwrites the BatchNorm output)
            hooks.append(bn.register_forward_hook(lambda mod, inp, o, i=i: zs.__setitem__(i, o.detach().permute(0, 2, 1).contiguous().clone())))
        model.train()
        y = model(x_train)
        loss = mpjpe(y, target)
        loss.backward()
        for h in hooks:
            h.remove()
        out[n + "|loss"] = np.float64(loss.item())
        for k, p_ in model.named_parameters():
            g = p_.grad.detach()
            rows = kat_grad_rows(k, tuple(g.shape))
            out[n + "|grad/" + k] = (g if rows is None else g[rows]).numpy().copy()
        n_cand = 0
        for i, z in enumerate(zs):
            flat = z.reshape(-1)
            idx = torch.nonzero(flat.abs() < KAT_NEAR_ZERO).reshape(-1)
            out[n + "|near_idx/%d" % i] = idx.numpy().astype(np.int64)
            out[n + "|near_pos/%d" % i] = (flat[idx] > 0).numpy()
            out[n + "|near_z/%d" % i] = flat[idx].numpy().copy()
            n_cand += idx.numel()
        print("kat_grads", n, "loss %.6f" % loss.item(), "near-zero pre-activations:", n_cand)
    np.savez_compressed(os.path.join(HERE, "kat_grads.npz"), **out)


def make_camera():
    gen = torch.Generator().manual_seed(7)
    n = 6
    X = torch.randn(n, 2, 17, 3, generator=gen)
    X[..., 2] = X[..., 2].abs() + 0.6     # mostly in front of the camera, some |x/z| > 1 to hit the clamp
    X[0, 0, 0, 2] = 0.2
    cam = torch.cat((torch.rand(n, 2, generator=gen) + 1.0, torch.randn(n, 2, generator=gen) * 0.05,
                     torch.randn(n, 3, generator=gen) * 0.2, torch.randn(n, 2, generator=gen) * 0.01), dim=1)
    out = {"X": X.numpy(), "cam": cam.numpy()}
    for nm, fn in (("full", project_to_2d), ("linear", project_to_2d_linear)):
        Xr = X.clone().requires_grad_(True)
        y = fn(Xr, cam)
        g = torch.randn(y.shape, generator=gen)
        (y * g).sum().backward()
        out["y_" + nm] = y.detach().numpy()
        out["g_" + nm] = g.numpy()
        out["dX_" + nm] = Xr.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "camera.npz"), **out)
    print("wrote camera")


H36M_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15]    # 17-joint skeleton (h36m_dataset.py:245-251)


def make_semi():
    """BASELINE config 5: one semi-supervised step exactly as run.py:322-394 composes it (both models, trajectory
    loss, back-projection through project_to_2d, bone-length term), reference classes on CPU."""
    torch.manual_seed(11)
    gen = torch.Generator().manual_seed(111)
    fw, C, B = [3, 3, 3], 64, 6
    pos = TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.0, channels=C)
    traj = TemporalModelOptimized1f(17, 2, 1, fw, dropout=0.0, channels=C)
    pad = (pos.receptive_field() - 1) // 2
    inputs_2d = (torch.randn(B, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
    inputs_2d_semi = (torch.randn(B, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
    inputs_3d = torch.randn(B, 1, 17, 3, generator=gen) * 0.3
    inputs_3d[:, :, 0, 2] = inputs_3d[:, :, 0, 2].abs() + 3.0            # root depth (camera space)
    cam = torch.cat((torch.rand(B, 2, generator=gen) + 1.0, torch.randn(B, 2, generator=gen) * 0.05,
                     torch.randn(B, 3, generator=gen) * 0.2, torch.randn(B, 2, generator=gen) * 0.01), dim=1)
    out = {"inputs_2d": inputs_2d.numpy(), "inputs_2d_semi": inputs_2d_semi.numpy(), "inputs_3d": inputs_3d.numpy().copy(),
           "cam": cam.numpy(), "parents": np.array(H36M_PARENTS)}
    for k, v in sd_numpy(pos).items():
        out["pos0/" + k] = v
    for k, v in sd_numpy(traj).items():
        out["traj0/" + k] = v
    pos.train(); traj.train()
    inputs_traj = inputs_3d[:, :, :1].clone()
    inputs_3d[:, :, 0] = 0
    split = B
    cat = torch.cat((inputs_2d, inputs_2d_semi), dim=0)
    p_cat = pos(cat)
    loss_3d = mpjpe(p_cat[:split], inputs_3d)
    t_cat = traj(cat)
    w = 1 / inputs_traj[:, :, :, 2]
    loss_traj = weighted_mpjpe(t_cat[:split], inputs_traj, w)
    target_semi = inputs_2d_semi[:, pad:-pad, :, :2].contiguous()
    recon = project_to_2d(p_cat[split:] + t_cat[split:], cam)
    loss_rec = mpjpe(recon, target_semi)
    dists = p_cat[:, :, 1:] - p_cat[:, :, H36M_PARENTS[1:]]
    bone = torch.mean(torch.norm(dists, dim=3), dim=1)
    penalty = torch.mean(torch.abs(torch.mean(bone[:split], dim=0) - torch.mean(bone[split:], dim=0)))
    total = loss_3d + loss_traj + loss_rec + penalty
    total.backward()
    out["losses"] = np.array([loss_3d.item(), loss_traj.item(), loss_rec.item(), penalty.item()])
    for k, p in pos.named_parameters():
        out["posgrad/" + k] = p.grad.numpy().copy()
    for k, p in traj.named_parameters():
        out["trajgrad/" + k] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "semi_step.npz"), **out)
    print("wrote semi_step losses", out["losses"])


def make_train_loop():
    """A run.py-style mini training loop (run.py:401-420 + 583-593): 6 Adam(amsgrad) steps on fixed batches, lr and
    BN-momentum decay, then the train->eval state_dict hand-off (run.py:426) and an eval forward."""
    torch.manual_seed(21)
    gen = torch.Generator().manual_seed(211)
    fw, C, B = [3, 3, 3], 64, 12
    tr = TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.0, channels=C)
    ev = TemporalModel(17, 2, 17, fw, dropout=0.0, channels=C)
    out = {}
    for k, v in sd_numpy(tr).items():
        out["sd0/" + k] = v
    xs = (torch.randn(6, B, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
    ys = torch.randn(6, B, 1, 17, 3, generator=gen) * 0.3
    ys[:, :, :, 0] = 0
    lr, lr_decay, mom0 = 1e-3, 0.95, 0.1
    opt = torch.optim.Adam(tr.parameters(), lr=lr, amsgrad=True)
    losses = []
    tr.train()
    for i in range(6):
        opt.zero_grad()
        loss = mpjpe(tr(xs[i]), ys[i])
        loss.backward()
        opt.step()
        losses.append(loss.item())
        lr *= lr_decay
        for g in opt.param_groups:
            g["lr"] *= lr_decay
        tr.set_bn_momentum(mom0 * np.exp(-(i + 1) / 6 * np.log(mom0 / 0.001)))
    ev.load_state_dict(tr.state_dict())
    ev.eval()
    x_eval = (torch.randn(2, 27 + 40, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
    with torch.no_grad():
        y_eval = ev(x_eval)
    out.update(xs=xs.numpy(), ys=ys.numpy(), losses=np.array(losses), x_eval=x_eval.numpy(), y_eval=y_eval.numpy())
    for k, v in sd_numpy(tr).items():
        out["sd1/" + k] = v
    np.savez_compressed(os.path.join(HERE, "train_loop.npz"), **out)
    print("wrote train_loop losses", losses)


CASES = [
    # name, kind, filter widths, causal, channels, extra keyword arguments of make_case
    ("dil_333_c32", "dilated", [3, 3, 3], False, 32, dict(extra_t=11)),
    ("dil_333_c32_causal", "dilated", [3, 3, 3], True, 32, dict(extra_t=5, seed=1)),
    ("dil_33333_c32", "dilated", [3, 3, 3, 3, 3], False, 32, dict(batch=2, extra_t=7, seed=2)),
    ("dil_353_c48_dense", "dilated", [3, 5, 3], False, 48, dict(dense=True, extra_t=4, seed=3)),
    ("dil_333_c32_traj", "dilated", [3, 3, 3], False, 32, dict(j_out=1, seed=4, extra_t=2)),
    ("dil_333_c32_drop", "dilated", [3, 3, 3], False, 32, dict(dropout=0.25, extra_t=3, seed=5)),
    ("str_333_c32", "strided", [3, 3, 3], False, 32, dict(batch=5)),
    ("str_333_c32_causal", "strided", [3, 3, 3], True, 32, dict(batch=4, seed=1)),
    ("str_33333_c32", "strided", [3, 3, 3, 3, 3], False, 32, dict(batch=4, seed=2, momentum=0.03)),
    ("str_353_c48_j15", "strided", [3, 5, 3], False, 48, dict(j_in=15, batch=6, seed=3)),
    ("str_333_c32_drop", "strided", [3, 3, 3], False, 32, dict(dropout=0.25, batch=8, seed=5)),
    ("str_333_c128", "strided", [3, 3, 3], False, 128, dict(batch=16, seed=6)),
    ("dil_333_c128", "dilated", [3, 3, 3], False, 128, dict(batch=2, extra_t=30, seed=6)),
    # round 2: channel counts the split-fp16 engine accepts (C % 64 == 0), so that the "f16x3" half of the GPU parity
    # suite executes k_nt_s16 / the S16 producers on reference-generated vectors: both arcs x causal x {64, 128},
    # dropout masks, the trajectory model (J_out = 1), a width-5 arc, and the dilated class (training + eval)
    ("str_333_c64", "strided", [3, 3, 3], False, 64, dict(batch=12, seed=0)),
    ("str_333_c64_causal", "strided", [3, 3, 3], True, 64, dict(batch=10, seed=1)),
    ("str_33333_c64", "strided", [3, 3, 3, 3, 3], False, 64, dict(batch=6, seed=2)),
    ("str_33333_c64_causal", "strided", [3, 3, 3, 3, 3], True, 64, dict(batch=6, seed=0, momentum=0.03)),
    ("str_333_c128_causal", "strided", [3, 3, 3], True, 128, dict(batch=16, seed=1)),
    ("str_33333_c128_causal", "strided", [3, 3, 3, 3, 3], True, 128, dict(batch=8, seed=2)),
    ("str_333_c64_drop", "strided", [3, 3, 3], False, 64, dict(dropout=0.25, batch=16, seed=5)),
    ("str_333_c64_causal_drop", "strided", [3, 3, 3], True, 64, dict(dropout=0.25, batch=16, seed=7)),
    ("str_333_c64_traj", "strided", [3, 3, 3], False, 64, dict(j_out=1, batch=8, seed=4)),
    ("str_353_c64", "strided", [3, 5, 3], False, 64, dict(batch=6, seed=3)),
    ("dil_333_c64", "dilated", [3, 3, 3], False, 64, dict(batch=3, extra_t=9, seed=0)),
    ("dil_333_c64_causal", "dilated", [3, 3, 3], True, 64, dict(batch=3, extra_t=6, seed=1)),
    ("dil_33333_c64_causal", "dilated", [3, 3, 3, 3, 3], True, 64, dict(batch=2, extra_t=5, seed=2)),
    ("dil_333_c64_drop", "dilated", [3, 3, 3], False, 64, dict(dropout=0.25, batch=3, extra_t=4, seed=5)),
    ("dil_353_c64_dense", "dilated", [3, 5, 3], False, 64, dict(dense=True, batch=2, extra_t=3, seed=3)),
]
EXTRA = {"kat": make_kat, "kat_matrix": make_kat_matrix, "kat_grads": make_kat_grads, "camera": make_camera, "semi_step": make_semi, "train_loop": make_train_loop}


if __name__ == "__main__":
    # python tests/golden/make_golden.py            -> everything
    # python tests/golden/make_golden.py NAME ...   -> only the named fixtures (case names, kat, kat_matrix, kat_grads, camera, semi_step, train_loop)
    torch.set_num_threads(4)
    want = set(sys.argv[1:])
    known = {c[0] for c in CASES} | set(EXTRA)
    assert want <= known, sorted(want - known)
    for name, kind, fw, causal, channels, kw in CASES:
        if not want or name in want:
            make_case(name, kind, fw, causal, channels, **kw)
    for name, fn in EXTRA.items():
        if not want or name in want:
            fn()
