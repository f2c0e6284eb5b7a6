"""Shared helpers for the tests: golden-fixture loading and error metrics."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    special = {"camera", "semi_step", "train_loop", "step_generators", "step_loss", "step_adam", "eval_protocol"}          # fixtures with their own layout / tests
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
                  if n not in special)


def load_npz_groups(name):
    """Fixture with 'group/key' entries -> dict of dicts (plain keys stay at top level)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        if "/" in k:
            grp, key = k.split("/", 1)
            out.setdefault(grp, {})[key] = z[k]
        else:
            out[k] = z[k]
    return out


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = dict(meta=json.loads(str(z["meta"])), sd0={}, sd1={}, grad={}, mask={})
    for k in z.files:
        if "/" in k:
            grp, key = k.split("/", 1)
            g[grp][key] = z[k]
        elif k != "meta":
            g[k] = z[k]
    if g["mask"]:
        p = g["meta"]["dropout"]
        g["masks"] = [g["mask"][str(i)].astype(np.float32) / (1.0 - p) for i in range(len(g["mask"]))]
    else:
        g["masks"] = None
    return g


def load_kats():
    with open(os.path.join(GOLDEN, "kat_c1024.json")) as f:
        return json.load(f)


def mpjpe_np(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float(np.mean(np.sqrt((d ** 2).sum(-1))))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


# ---- step-level fixtures (tests/golden/make_golden_step.py) -------------------------------------------------
KPS_LEFT, KPS_RIGHT = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]
JOINTS_LEFT, JOINTS_RIGHT = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
GEN_CASES = ("c1", "c3", "plain")


def load_step_dataset():
    """(fixture dict, cameras, poses_3d, poses_2d) of step_generators.npz."""
    z = np.load(os.path.join(GOLDEN, "step_generators.npz"))
    n = int(z["n_seq"])
    return (z, [z["cam_%d" % i] for i in range(n)], [z["p3_%d" % i] for i in range(n)],
            [z["p2_%d" % i] for i in range(n)])


def gen_case_meta(z, name):
    bs, cl, pad, shift, aug, shuf, cams, nb, nf = (int(v) for v in z[name + "/meta"])
    return dict(batch_size=bs, chunk_length=cl, pad=pad, causal_shift=shift, augment=bool(aug), shuffle=bool(shuf),
                cams=bool(cams), num_batches=nb, num_frames=nf, n=int(z[name + "/n"]))


def s16_or_skip(mode, model, t_in, training, need_dx=False):
    """Under the f16x3 id a configuration the split-fp16 engine does not implement is skipped, not silently re-run on
    the fp32 kernels (it is covered under the f32 id)."""
    import pytest
    from videopose3d_amd import engine
    if mode == "f16x3" and not engine.use_s16(model, t_in, training, need_dx):
        pytest.skip("engine_s16.supported() rejects this configuration: covered under the f32 id")


def unpack_act_bits(bits, m_rows, c):
    """Activation bits of an [m_rows, c] activation (videopose3d_amd: byte of (row m, channels c0..c0+7) at
    ((c0 // 64) * M + m) * 8 + (c0 % 64) // 8, bit e = channel c0 + e) -> bool array [m_rows, c]."""
    b = np.asarray(bits.detach().cpu().numpy() if hasattr(bits, "detach") else bits, np.uint8).reshape(c // 64, m_rows, 8)
    u = np.unpackbits(b[..., None], axis=-1, bitorder="little")            # [tile][m][byte][bit]
    return u.transpose(1, 0, 2, 3).reshape(m_rows, c).astype(bool)
