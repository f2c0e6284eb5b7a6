#!/usr/bin/env python3
"""Golden fixtures for the step-level pieces, produced by the REFERENCE code itself (run in the build container):

  step_generators.npz  /root/reference/common/generators.py ChunkedGenerator / UnchunkedGenerator on a small synthetic
                       dataset (ragged sequence lengths incl. ones shorter than the padding; augmentation; cameras)
  step_loss.npz        /root/reference/common/loss.py mpjpe / weighted_mpjpe values + autograd gradients, and the
                       TTA fold of run.py:677-680 executed line by line
  eval_protocol.npz    run.py:652-705 evaluate(): UnchunkedGenerator -> TemporalModel.eval() -> test-time-augmentation fold -> mpjpe
                       accumulated over ragged sequences, with and without TTA (model, data, per-sequence predictions, e1)
  step_adam.npz        torch.optim.Adam(lr, amsgrad=True) (the reference's optimizer, run.py:252) over 6 steps with
                       the per-epoch lr decay of run.py:583-588

    python tests/golden/make_golden_step.py        # needs /root/reference; writes tests/golden/step_*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from common.generators import ChunkedGenerator, UnchunkedGenerator  # noqa: E402
from common.loss import mpjpe, weighted_mpjpe  # noqa: E402

KPS_LEFT, KPS_RIGHT = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]          # COCO (run.py:81)
JOINTS_LEFT, JOINTS_RIGHT = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]                 # h36m skeleton


def make_dataset(rng):
    lens = [5, 24, 9, 31, 1]
    # small-integer-valued floats: the assembly is pure copy / negate, and such arrays compress well in the fixture
    p2 = [rng.randint(-120, 121, size=(n, 17, 2)).astype(np.float32) / 8 for n in lens]
    p3 = [rng.randint(-120, 121, size=(n, 17, 3)).astype(np.float32) / 8 for n in lens]
    cams = [rng.randint(1, 100, size=9).astype(np.float32) / 4 for _ in lens]
    return cams, p3, p2


def gen_generators():
    rng = np.random.RandomState(7)
    cams, p3, p2 = make_dataset(rng)
    out = {"n_seq": len(p2)}
    for i in range(len(p2)):
        out["p2_%d" % i], out["p3_%d" % i], out["cam_%d" % i] = p2[i], p3[i], cams[i]
    cases = [
        dict(name="c1", batch_size=16, chunk_length=1, pad=13, causal_shift=0, augment=True, shuffle=True, cams=True),
        dict(name="c3", batch_size=8, chunk_length=3, pad=4, causal_shift=4, augment=True, shuffle=True, cams=False),
        dict(name="plain", batch_size=32, chunk_length=1, pad=1, causal_shift=0, augment=False, shuffle=False, cams=True),
    ]
    for c in cases:
        g = ChunkedGenerator(c["batch_size"], cams if c["cams"] else None, p3, p2, c["chunk_length"], pad=c["pad"],
                             causal_shift=c["causal_shift"], shuffle=c["shuffle"], random_seed=1234,
                             augment=c["augment"], kps_left=KPS_LEFT, kps_right=KPS_RIGHT, joints_left=JOINTS_LEFT,
                             joints_right=JOINTS_RIGHT)
        out[c["name"] + "/meta"] = np.array([c["batch_size"], c["chunk_length"], c["pad"], c["causal_shift"],
                                             int(c["augment"]), int(c["shuffle"]), int(c["cams"]), g.num_batches,
                                             g.num_frames()])
        b = 0
        for epoch in range(2):                      # second epoch: the RandomState stream continues
            for cam, b3, b2 in g.next_epoch():
                out["%s/b2_%d" % (c["name"], b)] = b2.astype("float32")
                out["%s/b3_%d" % (c["name"], b)] = b3.astype("float32")
                if cam is not None:
                    out["%s/cam_%d" % (c["name"], b)] = cam.astype("float32")
                b += 1
        out[c["name"] + "/n"] = b
    # endless generator: resumes inside an epoch (generators.py:150-166), as the semi-supervised loop uses it
    g = ChunkedGenerator(8, None, None, p2, 1, pad=2, shuffle=True, random_seed=99, augment=False, endless=True)
    it = g.next_epoch()
    for b in range(2 * g.num_batches + 3):
        _, _, b2 = next(it)
        out["endless/b2_%d" % b] = b2.astype("float32")
    out["endless/n"] = 2 * g.num_batches + 3
    # UnchunkedGenerator with TTA
    u = UnchunkedGenerator(cams, p3, p2, pad=13, causal_shift=0, augment=True, kps_left=KPS_LEFT, kps_right=KPS_RIGHT,
                           joints_left=JOINTS_LEFT, joints_right=JOINTS_RIGHT)
    for i, (cam, b3, b2) in enumerate(u.next_epoch()):
        out["unch/cam_%d" % i], out["unch/b3_%d" % i], out["unch/b2_%d" % i] = \
            cam.astype("float32"), b3.astype("float32"), b2.astype("float32")
    u = UnchunkedGenerator(None, None, p2, pad=3, causal_shift=3, augment=False)
    for i, (_, _, b2) in enumerate(u.next_epoch()):
        out["unch_plain/b2_%d" % i] = b2.astype("float32")
    np.savez_compressed(os.path.join(HERE, "step_generators.npz"), **out)


def gen_loss():
    torch.manual_seed(3)
    out = {}
    for name, shape in (("pos", (64, 1, 17, 3)), ("big", (1300, 3, 17, 3)), ("rec2d", (16, 1, 17, 2))):
        p = torch.randn(shape)
        t = torch.randn(shape)
        if name == "big":                                  # > 65,536 points: the multi-block path; fp16-exact values
            p, t = p.half().float(), t.half().float()      # so that the fixture stores them as float16
        p.requires_grad_(True)
        if name == "pos":
            with torch.no_grad():
                p[0, 0, 0] = t[0, 0, 0]                # zero distance: sub-gradient 0
        l = mpjpe(p, t)
        l.backward()
        if name == "big":
            out[name + "/p"], out[name + "/t"] = p.detach().numpy().astype(np.float16), t.numpy().astype(np.float16)
            out[name + "/grad_head"], out[name + "/grad_tail"] = p.grad.numpy()[:8], p.grad.numpy()[-8:]
            out[name + "/grad_sum"] = p.grad.double().sum(dim=(0, 1, 2)).numpy()
            out[name + "/loss"] = l.detach().numpy()
            continue
        out[name + "/p"], out[name + "/t"], out[name + "/loss"], out[name + "/grad"] = \
            p.detach().numpy(), t.numpy(), l.detach().numpy(), p.grad.numpy()
    p = torch.randn(48, 1, 1, 3, requires_grad=True)     # trajectory model: J = 1, w = 1/z   (run.py:358-360)
    t = torch.randn(48, 1, 1, 3) + 4.0
    w = 1 / t[:, :, :, 2]
    l = weighted_mpjpe(p, t, w)
    (3.0 * l).backward()                                  # non-unit upstream gradient
    out["traj/p"], out["traj/t"], out["traj/w"], out["traj/loss"], out["traj/grad3"] = \
        p.detach().numpy(), t.numpy(), w.numpy(), l.detach().numpy(), p.grad.numpy()
    # TTA fold, run.py:677-680 line by line
    pred = torch.randn(2, 37, 17, 3)
    out["tta/pred"] = pred.numpy().copy()
    q = pred.clone()
    q[1, :, :, 0] *= -1
    q[1, :, JOINTS_LEFT + JOINTS_RIGHT] = q[1, :, JOINTS_RIGHT + JOINTS_LEFT]
    out["tta/out"] = torch.mean(q, dim=0, keepdim=True).numpy()
    q = pred[:, :, :1].clone()                            # trajectory model: no joint swap
    q[1, :, :, 0] *= -1
    out["tta/out_traj"] = torch.mean(q, dim=0, keepdim=True).numpy()
    np.savez_compressed(os.path.join(HERE, "step_loss.npz"), **out)


def gen_eval_protocol():
    """run.py:652-705 `evaluate()` composed end to end by the reference's own classes: UnchunkedGenerator (padding, test-time
    augmentation pair) -> TemporalModel.eval() -> un-flip + average -> mpjpe accumulated over the sequences, with and
    without test-time augmentation.  Stores the model, the sequences, every per-sequence prediction and the e1 numbers."""
    from common.model import TemporalModel
    torch.manual_seed(21)
    fw = [3, 3, 3]
    model = TemporalModel(17, 2, 17, fw, causal=False, dropout=0.25, channels=64)
    # (run a few training-mode batches so that the BatchNorm running statistics are not the initial 0 / 1)
    model.train()
    with torch.no_grad():
        for _ in range(3):
            model(torch.randn(8, 27 + 6, 17, 2) * 0.4)
    model.eval()
    rng = np.random.RandomState(11)
    lens = [40, 97, 61, 130, 28, 75]
    p2 = [(rng.standard_normal((n, 17, 2)) * 0.4).astype(np.float32) for n in lens]
    p3 = [(rng.standard_normal((n, 17, 3)) * 0.3).astype(np.float32) for n in lens]
    pad = (model.receptive_field() - 1) // 2
    out = {"n_seq": len(lens), "fw": np.array(fw), "channels": 64}
    for k, v in model.state_dict().items():
        out["sd/" + k] = v.numpy()
    for i in range(len(lens)):
        out["p2_%d" % i], out["p3_%d" % i] = p2[i], p3[i]
    for tag, augment in (("tta", True), ("plain", False)):
        gen = UnchunkedGenerator(None, p3, p2, pad=pad, causal_shift=0, augment=augment, kps_left=KPS_LEFT,
                                 kps_right=KPS_RIGHT, joints_left=JOINTS_LEFT, joints_right=JOINTS_RIGHT)
        epoch_loss, n = 0.0, 0
        with torch.no_grad():
            for s_id, (_, batch, batch_2d) in enumerate(gen.next_epoch()):
                inputs_2d = torch.from_numpy(batch_2d.astype("float32"))
                predicted = model(inputs_2d)                                             # run.py:669
                if gen.augment_enabled():                                                # run.py:674-680
                    predicted[1, :, :, 0] *= -1
                    predicted[1, :, JOINTS_LEFT + JOINTS_RIGHT] = predicted[1, :, JOINTS_RIGHT + JOINTS_LEFT]
                    predicted = torch.mean(predicted, dim=0, keepdim=True)
                inputs_3d = torch.from_numpy(batch.astype("float32"))
                inputs_3d[:, :, 0] = 0                                                   # run.py:688
                if gen.augment_enabled():
                    inputs_3d = inputs_3d[:1]
                error = mpjpe(predicted, inputs_3d)                                      # run.py:692
                epoch_loss += inputs_3d.shape[0] * inputs_3d.shape[1] * error.item()
                n += inputs_3d.shape[0] * inputs_3d.shape[1]
                out["%s/pred_%d" % (tag, s_id)] = predicted[0].numpy()
        out["%s/e1_mm" % tag] = np.float64(epoch_loss / n * 1000)                        # run.py:711
        out["%s/frames" % tag] = n
    np.savez_compressed(os.path.join(HERE, "eval_protocol.npz"), **out)
    print("eval_protocol.npz: e1 %.6f mm (TTA) / %.6f mm (plain) over %d frames" % (out["tta/e1_mm"], out["plain/e1_mm"], n))


def gen_adam():
    torch.manual_seed(11)
    shapes = [(33,), (7, 5, 3), (51,), (64, 64, 1)]
    params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    opt = torch.optim.Adam(params, lr=1e-3, amsgrad=True)
    out = {"n_params": len(shapes)}
    for i, p in enumerate(params):
        out["p0_%d" % i] = p.detach().numpy().copy()
    n_steps = 6
    for s in range(n_steps):
        for i, p in enumerate(params):
            scale = 10.0 ** (-(s % 3))                     # gradients shrinking then growing: exercises max_exp_avg_sq
            p.grad = torch.randn(p.shape) * scale
            out["g%d_%d" % (s, i)] = p.grad.numpy().copy()
        opt.step()
        if s == 2:
            for g in opt.param_groups:                     # run.py:583-588
                g["lr"] *= 0.95
        for i, p in enumerate(params):
            out["p%d_%d" % (s + 1, i)] = p.detach().numpy().copy()
    sd = opt.state_dict()
    for i in range(len(shapes)):
        st = sd["state"][i]
        out["m_%d" % i], out["v_%d" % i], out["vmax_%d" % i] = \
            st["exp_avg"].numpy(), st["exp_avg_sq"].numpy(), st["max_exp_avg_sq"].numpy()
    out["n_steps"] = n_steps
    np.savez_compressed(os.path.join(HERE, "step_adam.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "eval":          # only the evaluation-protocol fixture (leaves the others untouched)
        gen_eval_protocol()
        sys.exit(0)
    gen_generators()
    gen_loss()
    gen_adam()
    gen_eval_protocol()
    for f in ("step_generators.npz", "step_loss.npz", "step_adam.npz", "eval_protocol.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
