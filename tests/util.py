"""Shared helpers for the tests: golden-fixture loading and error metrics."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    special = {"camera", "semi_step", "train_loop", "step_generators", "step_loss", "step_adam", "eval_protocol", "kat_matrix", "kat_grads"}          # fixtures with their own layout / tests
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


# ---- C = 1024 known-answer matrix (tests/golden/make_golden.py kat_matrix -> tests/golden/kat_matrix.npz) -----
# Weights are not stored: they follow from torch.manual_seed(seed) + the constructor (tests/test_host_cpu.py pins that the
# package's classes draw the same parameters as the reference's); BatchNorm state, inputs, targets and the projection
# vectors follow from a second seeded generator.  The generator script runs this recipe with the REFERENCE classes, the
# tests with the package's (or feed the state to the oracle).
KAT_GRAD_TOL = 1e-2


def kat_matrix_cases():
    out = []
    for seed in (0, 1, 2):
        for fw in ((3, 3, 3), (3, 3, 3, 3, 3)):
            for causal in (False, True):
                for kind in ("dilated", "strided"):
                    out.append(dict(name="%s_%s_s%d%s" % (kind[:3], "".join(map(str, fw)), seed, "_causal" if causal else ""),
                                    kind=kind, filter_widths=list(fw), causal=causal, seed=seed))
    return out


def kat_matrix_build(case, dilated_cls, strided_cls, channels=1024):
    """(model on CPU in train mode with dropout 0, x_eval, x_train, target, {param name: projection vector})."""
    import torch
    torch.manual_seed(case["seed"])
    cls = dilated_cls if case["kind"] == "dilated" else strided_cls
    model = cls(17, 2, 17, case["filter_widths"], causal=case["causal"], dropout=0.0, channels=channels)
    gen = torch.Generator().manual_seed(4242 + case["seed"])
    sd = model.state_dict()
    with torch.no_grad():
        for k in sorted(sd):
            if k.endswith("running_mean"):
                sd[k].copy_(torch.randn(sd[k].shape, generator=gen) * 0.1)
            elif k.endswith("running_var"):
                sd[k].copy_(torch.rand(sd[k].shape, generator=gen) * 1.5 + 0.5)
            elif "bn" in k and k.endswith("weight"):
                sd[k].copy_(1.0 + 0.2 * torch.randn(sd[k].shape, generator=gen))
            elif "bn" in k and k.endswith("bias"):
                sd[k].copy_(0.1 * torch.randn(sd[k].shape, generator=gen))
    rf = model.receptive_field()
    dil = case["kind"] == "dilated"
    x_eval = (torch.randn(2, rf + (57 if dil else 0), 17, 2, generator=gen) * 0.5).clamp(-1, 1)
    # 256 rows reach the last BatchNorm layers.  Gradient tolerance of this matrix (KAT_GRAD_TOL): any two fp32 evaluations
    # of one step may disagree on the sign of a pre-activation that is zero to rounding; one such ReLU flip in a late
    # layer moves every upstream gradient by ~ 1 / sqrt(rows x C) = 2e-3 of its norm (measured against the fp64 oracle:
    # oracle-fp32, torch-fp32 and both GPU arithmetics each show 3e-4 .. 8e-3 on some case at 16 rows, 1e-6 on the
    # others), and the bias gradients of inner BatchNorm layers are small differences of large sums (the next
    # BatchNorm removes a per-channel shift).  Outputs, loss and running statistics are held to the usual bars.
    x_train = (torch.randn(8 if dil else 256, rf + (31 if dil else 0), 17, 2, generator=gen) * 0.5).clamp(-1, 1)
    target = torch.randn(x_train.shape[0], x_train.shape[1] - rf + 1, 17, 3, generator=gen) * 0.3
    target[:, :, 0] = 0
    proj = {k: torch.randn(p.shape, generator=gen) for k, p in sorted(model.named_parameters())}
    for k in sorted(sd):
        if "running" in k:
            proj[k] = torch.randn(sd[k].shape, generator=gen)
    return model, x_eval, x_train, target, proj


def kat_matrix_summaries(model, proj):
    """After loss.backward() on `model` (any device): {key: [norm, projection]} for every parameter gradient and
    every BatchNorm running statistic, in float64."""
    out = {}
    for k, p in model.named_parameters():
        g = p.grad.detach().double().cpu()
        out["grad/" + k] = [float(g.norm()), float((g * proj[k].double()).sum())]
    for k, v in model.state_dict().items():
        if "running" in k:
            v = v.detach().double().cpu()
            out["stat/" + k] = [float(v.norm()), float((v * proj[k].double()).sum())]
    return out


# ---- full reference gradients for four of the matrix's cases (tests/golden/make_golden.py kat_grads -> kat_grads.npz) ----
KAT_NEAR_ZERO = 1e-4          # |pre-activation| below which the reference's ReLU decision is stored (flip candidates)
KAT_FULL_GRAD_TOL = 5e-4      # max-norm relative, per tensor -- with the ReLU decisions accounted for, not absorbed
KAT_GRAD_ROWS = 64


def kat_grad_cases():
    return [c for c in kat_matrix_cases() if c["kind"] == "strided" and c["seed"] == 0]


def kat_grad_rows(name, shape):
    """Rows (output channels) of a C x C conv weight gradient that the fixture stores: 64 seeded ones; None = whole tensor."""
    import torch
    if not (name.startswith("layers_conv") and len(shape) == 3 and shape[0] > KAT_GRAD_ROWS and shape[1] > KAT_GRAD_ROWS):
        return None
    gen = torch.Generator().manual_seed(97 + sum(ord(ch) for ch in name))
    return torch.sort(torch.randperm(shape[0], generator=gen)[:KAT_GRAD_ROWS]).values


def load_kat_grads():
    z = np.load(os.path.join(GOLDEN, "kat_grads.npz"))
    out = {}
    for k in z.files:
        name, key = k.split("|", 1)
        out.setdefault(name, {})[key] = z[k]
    return out


def kat_reference_relu_pos(own_pos, ref, n_layers):
    """ReLU decisions of the reference for one kat_grads case, given another fp32 evaluation's decisions `own_pos` (list of
    bool arrays [B,T,C]): identical except where the reference stored a near-zero pre-activation with the other sign.
    Returns (pos as the reference decided, number of flips).  Outside the stored candidates |z_ref| >= KAT_NEAR_ZERO, two
    fp32 evaluations agreeing to ~1e-6 cannot differ in sign there."""
    pos, flips = [], 0
    for i in range(n_layers):
        p = np.array(own_pos[i], dtype=bool, copy=True)
        flat = p.reshape(-1)
        idx, rp = ref["near_idx/%d" % i], ref["near_pos/%d" % i].astype(bool)
        flips += int((flat[idx] != rp).sum())
        flat[idx] = rp
        pos.append(p)
    return pos, flips


def load_kat_matrix():
    z = np.load(os.path.join(GOLDEN, "kat_matrix.npz"))
    out = {}
    for k in z.files:
        name, key = k.split("|", 1)
        out.setdefault(name, {})[key] = z[k]
    return out


# ---- adversarial intra-tensor dynamic range (tests/test_gpu_s16.py, tools/range_edges.py) --------------------------------
RANGE_FW = [3, 3, 3]
RANGE_CASES = ("none", "gamma", "gamma_all_layers", "beta", "w_row", "w_col")


def range_edge_state(channels, case, s, ch=5, seed=0):
    """state_dict (CPU tensors) of a strided arc-3,3,3 model with a trained-looking BatchNorm affine and ONE adversarial
    edit of factor 2^s: `gamma` / `beta`: channel `ch` of the first C x C block's BatchNorm; `gamma_all_layers`: the same
    channel hot in every BatchNorm (compounds along the residual chain); `w_row`: output row `ch` of the first 3-tap C x C
    conv; `w_col`: input column `ch` of the 1x1 conv (makes that channel's incoming gradient, hence dy, hot)."""
    import torch
    import videopose3d_amd as V
    torch.manual_seed(seed)
    m = V.TemporalModelOptimized1f(17, 2, 17, RANGE_FW, dropout=0.0, channels=channels)
    f = float(2.0 ** s)
    with torch.no_grad():
        for bn in [m.expand_bn] + list(m.layers_bn):
            bn.weight.copy_(1.0 + 0.2 * torch.randn_like(bn.weight))
            bn.bias.copy_(0.1 * torch.randn_like(bn.bias))
        if case == "gamma":
            m.layers_bn[0].weight[ch] *= f
        elif case == "gamma_all_layers":
            for bn in [m.expand_bn] + list(m.layers_bn):
                bn.weight[ch] *= f
        elif case == "beta":
            m.layers_bn[0].bias[ch] = 0.1 * f
        elif case == "w_row":
            m.layers_conv[0].weight[ch] *= f
        elif case == "w_col":
            m.layers_conv[1].weight[:, ch] *= f
        elif case != "none":
            raise ValueError(case)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def range_edge_batch(batch=64, joint_scale=None):
    import torch
    g = torch.Generator().manual_seed(1234)
    x = (torch.randn(batch, 27, 17, 2, generator=g) * 0.5).clamp(-1, 1)
    if joint_scale is not None:
        x[:, :, 3, :] *= joint_scale                    # one joint far outside the screen-normalised range of the others
    tgt = torch.randn(batch, 1, 17, 3, generator=g) * 0.3
    return x, tgt


def range_edge_oracle(sd, x, tgt):
    """float64 oracle: (output, {name: gradient}) of one mpjpe training step (dropout 0)."""
    from oracle import temporal_oracle as O
    sdn = {k: v.numpy() for k, v in sd.items()}
    yo, cache, _ = O.forward(sdn, x.numpy(), RANGE_FW, kind="strided", training=True, dtype=np.float64)
    return yo, O.backward(cache, O.mpjpe_grad(yo, tgt.numpy().astype(np.float64)))


def range_edge_errors(model, x, tgt, yo, go):
    """One step of `model` (on the GPU) against the oracle's (yo, go): dict(mpjpe, grad_maxnorm [worst tensor, |d| / max|ref|],
    grad_rowrel [worst output row / input column relative to ITS OWN reference maximum], finite, names)."""
    import torch
    dev = next(model.parameters()).device
    model.zero_grad(set_to_none=True)
    y = model(x.to(dev))
    torch.mean(torch.norm(y - tgt.to(dev), dim=3)).backward()
    torch.cuda.synchronize()
    yv = y.detach().cpu().numpy().astype(np.float64)
    r = dict(mpjpe=float(np.mean(np.sqrt(((yv - yo) ** 2).sum(-1)))), finite=bool(np.isfinite(yv).all()))
    worst_t, worst_r, wt_name, wr_name = 0.0, 0.0, "", ""
    for k, p in model.named_parameters():
        a = p.grad.detach().cpu().numpy().astype(np.float64)
        ref = go[k].astype(np.float64)
        r["finite"] = r["finite"] and bool(np.isfinite(a).all())
        d = np.abs(a - ref)
        et = float(d.max() / (np.abs(ref).max() + 1e-300))
        d2, r2 = d.reshape(d.shape[0], -1), np.abs(ref).reshape(ref.shape[0], -1)
        rmax = r2.max(axis=1)
        ok = rmax > 0
        er = float((d2.max(axis=1)[ok] / rmax[ok]).max()) if ok.any() else 0.0
        if a.ndim == 3 and a.shape[1] > 1:               # conv weights: also per INPUT channel
            d3 = d.transpose(1, 0, 2).reshape(a.shape[1], -1)
            cmax = np.abs(ref).transpose(1, 0, 2).reshape(a.shape[1], -1).max(axis=1)
            okc = cmax > 0
            er = max(er, float((d3.max(axis=1)[okc] / cmax[okc]).max()))
        if et > worst_t:
            worst_t, wt_name = et, k
        if er > worst_r:
            worst_r, wr_name = er, k
    r.update(grad_maxnorm=worst_t, grad_maxnorm_at=wt_name, grad_rowrel=worst_r, grad_rowrel_at=wr_name)
    return r
