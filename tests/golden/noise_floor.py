#!/usr/bin/env python3
"""Reference-vs-reference noise floor of the 6-step training loop of tests/golden/train_loop.npz (make_golden.py
make_train_loop): the REFERENCE classes, the same seeds, run under different fp32 evaluation orders (thread counts, oneDNN on /
off) and once in float64.  What two fp32 runs of the reference itself disagree by after 6 Adam(amsgrad) steps is the floor
under any tolerance of tests/test_gpu_parity.py::test_training_loop_vs_reference_golden.

    python tests/golden/noise_floor.py        (needs /root/reference; prints a table, writes train_loop_noise.json)
"""
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("VP3D_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
from common.model import TemporalModel, TemporalModelOptimized1f  # noqa: E402  (reference, executed only)
from common.loss import mpjpe  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run(threads, mkldnn=True, double=False):
    torch.set_num_threads(threads)
    torch.manual_seed(21)
    gen = torch.Generator().manual_seed(211)
    fw, C, B = [3, 3, 3], 64, 12
    tr = TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.0, channels=C)
    ev = TemporalModel(17, 2, 17, fw, dropout=0.0, channels=C)
    xs = (torch.randn(6, B, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
    ys = torch.randn(6, B, 1, 17, 3, generator=gen) * 0.3
    ys[:, :, :, 0] = 0
    if double:
        tr, ev, xs, ys = tr.double(), ev.double(), xs.double(), ys.double()
    lr_decay, mom0 = 0.95, 0.1
    opt = torch.optim.Adam(tr.parameters(), lr=1e-3, amsgrad=True)
    losses = []
    tr.train()
    with torch.backends.mkldnn.flags(enabled=mkldnn):
        for i in range(6):
            opt.zero_grad()
            loss = mpjpe(tr(xs[i]), ys[i])
            loss.backward()
            opt.step()
            losses.append(loss.item())
            for g in opt.param_groups:
                g["lr"] *= lr_decay
            tr.set_bn_momentum(mom0 * np.exp(-(i + 1) / 6 * np.log(mom0 / 0.001)))
        ev.load_state_dict(tr.state_dict())
        ev.eval()
        x_eval = (torch.randn(2, 27 + 40, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
        if double:
            x_eval = x_eval.double()
        with torch.no_grad():
            y_eval = ev(x_eval)
    sd = {k: v.detach().double().numpy().copy() for k, v in tr.state_dict().items()}
    return np.array(losses), y_eval.double().numpy(), sd


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def main():
    base = run(4)
    gold = np.load(os.path.join(HERE, "train_loop.npz"))
    assert np.abs(base[0] - gold["losses"]).max() == 0.0, "the 4-thread run is the committed fixture"
    variants = {"1 thread": run(1), "8 threads": run(8), "4 threads, oneDNN off": run(4, mkldnn=False), "float64": run(4, double=True)}
    out = {}
    print("%-24s %-12s %-12s %-14s %-14s" % ("variant vs fixture", "max|dloss|", "eval MPJPE", "running_mean", "running_var"))
    for name, (losses, y, sd) in variants.items():
        d_loss = float(np.abs(losses - base[0]).max())
        e = float(np.mean(np.linalg.norm(y - base[1], axis=-1)))
        rm = max(rel(sd[k], base[2][k]) for k in sd if k.endswith("running_mean"))
        rv = max(rel(sd[k], base[2][k]) for k in sd if k.endswith("running_var"))
        out[name] = dict(max_abs_loss_diff=d_loss, eval_mpjpe=e, running_mean_rel=rm, running_var_rel=rv)
        print("%-24s %-12.3e %-12.3e %-14.3e %-14.3e" % (name, d_loss, e, rm, rv))
    fp32 = [v for k, v in out.items() if k != "float64"]
    out["fp32_floor"] = {k: max(v[k] for v in fp32) for k in fp32[0]}
    print("fp32 floor (max over the fp32 variants):", out["fp32_floor"])
    json.dump(out, open(os.path.join(HERE, "train_loop_noise.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
