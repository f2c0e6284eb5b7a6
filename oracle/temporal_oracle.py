"""CPU oracle for the VideoPose3D temporal-model hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The shipped path
(``videopose3d_amd``) never imports anything from ``oracle/`` and raises if its HIP
library is missing.

It is an independent numpy restatement (channels-last "NLC" implicit-GEMM form, explicit
hand-derived backward, no torch.nn, no autograd) of the arithmetic that the reference
delegates to ``torch.nn.Conv1d / BatchNorm1d / ReLU / Dropout``:

  reference common/model.py:63-77    forward() reshape wrapper           -> ``forward``
  reference common/model.py:102-121  TemporalModel layer hyper-params    -> ``layer_plan(kind="dilated")``
  reference common/model.py:167-182  Optimized1f layer hyper-params      -> ``layer_plan(kind="strided")``
  reference common/model.py:126-138  dilated block wiring + residual     -> ``forward`` / ``backward``
  reference common/model.py:187-197  strided block wiring + residual     -> ``forward`` / ``backward``
  reference common/loss.py:11-17     mpjpe                               -> ``mpjpe`` / ``mpjpe_grad``
  reference common/camera.py:37-90   project_to_2d[_linear]              -> ``project_to_2d`` (+ ``_grad``)

The arithmetic itself lives in a third-party dependency of the reference that is not vendored
in /root/reference: PyTorch (README.md:31 pins only ">= 0.4.0"; this container has
torch 2.10.0+rocm7.0).  The published semantics restated here: cross-correlation Conv1d,
BatchNorm1d (biased batch variance for normalisation, unbiased for the running update,
eps=1e-5), ReLU, inverted dropout (keep-prob 1-p, scale 1/(1-p)).

Pinning: the reference has no tests or golden vectors for this path ("parity unpinned" by
the reference itself).  The oracle is therefore pinned against outputs of the reference
classes run in the build container: ``tests/golden/make_golden.py`` (committed) imports
/root/reference/common/model.py, and ``tests/test_oracle_golden.py`` checks this file
against the resulting fixtures (forward eval/train, running stats, all parameter grads).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5


# --------------------------------------------------------------------------------------
# layer hyper-parameters
# --------------------------------------------------------------------------------------
def layer_plan(filter_widths, causal=False, kind="dilated", dense=False):
    """Per-conv hyper-parameters, following model.py:102-121 (dilated) / 167-182 (strided).

    Returns dict(pad=[...], causal_shift=[...], convs=[dict(taps, dil, stride)], res=[...]).
    convs[0] is expand_conv, convs[1+2i] / convs[2+2i] are layers_conv[2i] / [2i+1].
    res[i] describes the residual slice of block i (model.py:132 / 191).
    """
    fw = list(filter_widths)
    for f in fw:
        assert f % 2 != 0, "Only odd filter widths are supported"  # model.py:20-21
    strided = kind == "strided"
    assert not (strided and dense)
    pad = [fw[0] // 2]
    causal_shift = [fw[0] // 2 if causal else 0]
    convs = [dict(taps=fw[0], dil=1, stride=fw[0] if strided else 1)]
    res = []
    next_dil = fw[0]
    for i in range(1, len(fw)):
        pad.append((fw[i] - 1) * next_dil // 2)
        if strided:
            causal_shift.append(fw[i] // 2 if causal else 0)
            convs.append(dict(taps=fw[i], dil=1, stride=fw[i]))
            res.append(dict(kind="strided", start=causal_shift[-1] + fw[i] // 2, step=fw[i]))
        else:
            causal_shift.append((fw[i] // 2 * next_dil) if causal else 0)
            if dense:
                convs.append(dict(taps=2 * pad[-1] + 1, dil=1, stride=1))
            else:
                convs.append(dict(taps=fw[i], dil=next_dil, stride=1))
            res.append(dict(kind="crop", start=pad[-1] + causal_shift[-1], pad=pad[-1]))
        convs.append(dict(taps=1, dil=1, stride=1))
        next_dil *= fw[i]
    return dict(pad=pad, causal_shift=causal_shift, convs=convs, res=res)


def receptive_field(filter_widths):
    """model.py:41-48."""
    return 1 + 2 * sum(layer_plan(filter_widths)["pad"])


def total_causal_shift(filter_widths, causal, kind):
    """model.py:50-61, reproduced as written (including its double counting for the dilated class)."""
    plan = layer_plan(filter_widths, causal, kind)
    frames = plan["causal_shift"][0]
    next_dil = filter_widths[0]
    for i in range(1, len(filter_widths)):
        frames += plan["causal_shift"][i] * next_dil
        next_dil *= filter_widths[i]
    return frames


# --------------------------------------------------------------------------------------
# primitives (NLC layout: x[B, T, C])
# --------------------------------------------------------------------------------------
def tconv_out_len(t_in, taps, dil, stride):
    return (t_in - dil * (taps - 1) - 1) // stride + 1


def tconv_fwd(x, w, dil=1, stride=1, bias=None):
    """y[b,t,:] = sum_k x[b, t*stride + k*dil, :] @ w[:,:,k].T   (torch Conv1d semantics).

    x: [B, T, C_in]; w: [C_out, C_in, taps] (reference Conv1d.weight layout)."""
    b, t_in, c_in = x.shape
    c_out, c_in2, taps = w.shape
    assert c_in == c_in2
    t_out = tconv_out_len(t_in, taps, dil, stride)
    assert t_out >= 1
    y = np.zeros((b, t_out, c_out), dtype=x.dtype)
    for k in range(taps):
        xs = x[:, k * dil: k * dil + stride * (t_out - 1) + 1: stride, :]
        y += (xs.reshape(-1, c_in) @ np.ascontiguousarray(w[:, :, k].T)).reshape(b, t_out, c_out)
    if bias is not None:
        y = y + bias
    return y


def tconv_wgrad(x, g, taps, dil=1, stride=1):
    """dW[:,:,k] = sum_{b,t} g[b,t,:]^T (x) x[b, t*stride + k*dil, :]."""
    b, t_out, c_out = g.shape
    c_in = x.shape[2]
    dw = np.zeros((c_out, c_in, taps), dtype=x.dtype)
    g2 = g.reshape(-1, c_out)
    for k in range(taps):
        xs = x[:, k * dil: k * dil + stride * (t_out - 1) + 1: stride, :].reshape(-1, c_in)
        dw[:, :, k] = g2.T @ xs
    return dw


def tconv_dgrad(g, w, t_in, dil=1, stride=1):
    """dx[b, t*stride + k*dil, :] += g[b,t,:] @ w[:,:,k]."""
    b, t_out, c_out = g.shape
    _, c_in, taps = w.shape
    dx = np.zeros((b, t_in, c_in), dtype=g.dtype)
    g2 = g.reshape(-1, c_out)
    for k in range(taps):
        contrib = (g2 @ np.ascontiguousarray(w[:, :, k])).reshape(b, t_out, c_in)
        dx[:, k * dil: k * dil + stride * (t_out - 1) + 1: stride, :] += contrib
    return dx


def bn_train_fwd(y, gamma, beta, eps=BN_EPS):
    """Batch statistics over all rows (B*T) per channel.  Returns z, xhat, mean, invstd, biased var."""
    m = y.shape[0] * y.shape[1]
    y2 = y.reshape(m, -1)
    mean = y2.mean(axis=0, dtype=np.float64)
    var = ((y2.astype(np.float64) - mean) ** 2).mean(axis=0)
    mean = mean.astype(y.dtype)
    var = var.astype(y.dtype)
    invstd = (1.0 / np.sqrt(var.astype(np.float64) + eps)).astype(y.dtype)
    xhat = (y - mean) * invstd
    z = xhat * gamma + beta
    return z, xhat, mean, invstd, var


def bn_running_update(running_mean, running_var, mean, var, m_rows, momentum):
    """running <- (1-m) running + m stat, unbiased variance for the running buffer."""
    unbiased = var * (m_rows / max(m_rows - 1, 1))
    return ((1 - momentum) * running_mean + momentum * mean,
            (1 - momentum) * running_var + momentum * unbiased)


def bn_eval(y, gamma, beta, running_mean, running_var, eps=BN_EPS):
    s = gamma / np.sqrt(running_var + eps)
    return y * s + (beta - running_mean * s)


def bn_act_bwd(go, z, xhat, gamma, invstd, mk, pos=None):
    """Backward of a = dropout(relu(bn(y))).  go = dL/da, mk = keep-mask*scale (or None).

    g = go*mk*[z>0]; dbeta = sum g; dgamma = sum g*xhat;
    dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M).
    pos (optional bool array) pins the ReLU decisions [z>0]: two fp32 implementations legitimately disagree on
    the sign of |z| ~ 1e-7 elements, and one such flip moves a gradient entry by a whole term."""
    g = go * ((z > 0) if pos is None else pos)
    if mk is not None:
        g = g * mk
    m = g.shape[0] * g.shape[1]
    g2 = g.reshape(m, -1)
    dbeta = g2.sum(axis=0, dtype=np.float64).astype(go.dtype)
    dgamma = (g2.astype(np.float64) * xhat.reshape(m, -1)).sum(axis=0).astype(go.dtype)
    dy = (gamma * invstd) * (g - dbeta / m - xhat * (dgamma / m))
    return dy, dgamma, dbeta


# --------------------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------------------
def _conv_w(params, idx):
    return params["expand_conv.weight"] if idx == 0 else params["layers_conv.%d.weight" % (idx - 1)]


def _bn_prefix(idx):
    return "expand_bn" if idx == 0 else "layers_bn.%d" % (idx - 1)


def _res_slice(h, r):
    if r["kind"] == "crop":
        t = h.shape[1]
        return h[:, r["start"]: t - 2 * r["pad"] + r["start"], :]
    return h[:, r["start"]:: r["step"], :]


def _res_unslice_add(dh, g, r):
    if r["kind"] == "crop":
        t = dh.shape[1]
        dh[:, r["start"]: t - 2 * r["pad"] + r["start"], :] += g
    else:
        dh[:, r["start"]:: r["step"], :][:, : g.shape[1]] += g


def forward(params, x, filter_widths, *, causal=False, kind="dilated", dense=False, training=False,
            dropout_masks=None, momentum=0.1, eps=BN_EPS, dtype=np.float32):
    """Whole-model forward.  x: [B, T, J, F] -> out [B, T_out, J_out, 3]  (model.py:63-77).

    params: dict name -> ndarray with the reference state_dict names/shapes.
    dropout_masks: None (identity / eval) or a list with one entry per BN layer (9 for 4 blocks),
        each an array [B, T_l, C] with values in {0, 1/(1-p)} (or None for that layer).
    Returns (out, cache, new_running) where new_running maps '<bn>.running_mean|running_var' to the
    post-step buffers (training only) and cache feeds ``backward``.
    """
    plan = layer_plan(filter_widths, causal, kind, dense)
    p = {k: np.asarray(v).astype(dtype) if np.asarray(v).dtype.kind == "f" else np.asarray(v)
         for k, v in params.items()}
    b, t = x.shape[0], x.shape[1]
    h = np.asarray(x, dtype=dtype).reshape(b, t, -1)  # [B,T,J*F] is already NLC
    n_bn = len(plan["convs"])
    cache = dict(plan=plan, layers=[], x_in=h, p=p)
    new_running = {}

    def act(y, idx):
        pre = _bn_prefix(idx)
        gamma, beta = p[pre + ".weight"], p[pre + ".bias"]
        mk = None
        if training:
            z, xhat, mean, invstd, var = bn_train_fwd(y, gamma, beta, eps)
            rm, rv = bn_running_update(p[pre + ".running_mean"], p[pre + ".running_var"], mean, var,
                                       y.shape[0] * y.shape[1], momentum)
            new_running[pre + ".running_mean"] = rm.astype(dtype)
            new_running[pre + ".running_var"] = rv.astype(dtype)
            if dropout_masks is not None and dropout_masks[idx] is not None:
                mk = np.asarray(dropout_masks[idx], dtype=dtype)
        else:
            z = bn_eval(y, gamma, beta, p[pre + ".running_mean"], p[pre + ".running_var"], eps)
            xhat = invstd = None
        a = np.maximum(z, 0)
        if mk is not None:
            a = a * mk
        return a, dict(z=z, xhat=xhat, invstd=invstd, mk=mk)

    # expand: drop(relu(expand_bn(expand_conv(x))))         model.py:127 / 188
    c = plan["convs"][0]
    y = tconv_fwd(h, _conv_w(p, 0), c["dil"], c["stride"])
    a, st = act(y, 0)
    cache["layers"].append(dict(x=h, **st))
    h = a
    for i, r in enumerate(plan["res"]):
        res = _res_slice(h, r)                               # model.py:132 / 191
        c = plan["convs"][1 + 2 * i]
        y = tconv_fwd(h, _conv_w(p, 1 + 2 * i), c["dil"], c["stride"])
        a1, st = act(y, 1 + 2 * i)                           # model.py:134 / 193
        cache["layers"].append(dict(x=h, **st))
        y2 = tconv_fwd(a1, _conv_w(p, 2 + 2 * i), 1, 1)
        a2, st = act(y2, 2 + 2 * i)                          # model.py:135 / 194
        cache["layers"].append(dict(x=a1, **st))
        assert res.shape == a2.shape, (res.shape, a2.shape)
        h = res + a2
    cache["h_last"] = h
    out = tconv_fwd(h, p["shrink.weight"], 1, 1, bias=p["shrink.bias"])   # model.py:137 / 196
    assert n_bn == len(cache["layers"])
    j_out3 = out.shape[2]
    return out.reshape(b, -1, j_out3 // 3, 3), cache, new_running


def backward(cache, gout, trace=None, relu_pos=None):
    """Gradients of all parameters given gout = dL/d(out) [B,T_out,J_out,3] (training-mode cache).
    If `trace` is a dict it receives the intermediate activation gradients (dy per BN layer, dh per block)."""
    plan, p = cache["plan"], cache["p"]
    grads = {}
    h_last = cache["h_last"]
    b, t_out, c = h_last.shape
    g = np.asarray(gout, dtype=h_last.dtype).reshape(b, t_out, -1)
    grads["shrink.bias"] = g.reshape(-1, g.shape[2]).sum(axis=0)
    grads["shrink.weight"] = tconv_wgrad(h_last, g, 1)
    dh = tconv_dgrad(g, p["shrink.weight"], t_out)

    def act_bwd(go, idx):
        L = cache["layers"][idx]
        pre = _bn_prefix(idx)
        dy, dgam, dbet = bn_act_bwd(go, L["z"], L["xhat"], p[pre + ".weight"], L["invstd"], L["mk"],
                                    None if relu_pos is None else relu_pos[idx])
        grads[pre + ".weight"] = dgam
        grads[pre + ".bias"] = dbet
        if trace is not None:
            trace["go%d" % idx] = go
            trace["dy%d" % idx] = dy
        return dy, L["x"]

    for i in reversed(range(len(plan["res"]))):
        r = plan["res"][i]
        # h_{i+1} = res + a2
        dy2, a1 = act_bwd(dh, 2 + 2 * i)
        grads["layers_conv.%d.weight" % (1 + 2 * i)] = tconv_wgrad(a1, dy2, 1)
        da1 = tconv_dgrad(dy2, _conv_w(p, 2 + 2 * i), a1.shape[1])
        dy1, h_in = act_bwd(da1, 1 + 2 * i)
        cc = plan["convs"][1 + 2 * i]
        grads["layers_conv.%d.weight" % (2 * i)] = tconv_wgrad(h_in, dy1, cc["taps"], cc["dil"], cc["stride"])
        dh_in = tconv_dgrad(dy1, _conv_w(p, 1 + 2 * i), h_in.shape[1], cc["dil"], cc["stride"])
        _res_unslice_add(dh_in, dh, r)
        dh = dh_in
    dy0, x_in = act_bwd(dh, 0)
    cc = plan["convs"][0]
    grads["expand_conv.weight"] = tconv_wgrad(x_in, dy0, cc["taps"], cc["dil"], cc["stride"])
    return grads


# --------------------------------------------------------------------------------------
# loss (reference common/loss.py:11-17) -- the parity metric
# --------------------------------------------------------------------------------------
def mpjpe(pred, target, w=None):
    """loss.py:11-17 (w is None) / :19-25 weighted_mpjpe (w broadcasts over the per-joint norm tensor)."""
    assert pred.shape == target.shape
    d = np.asarray(pred, dtype=np.float64) - np.asarray(target, dtype=np.float64)
    n = np.sqrt((d ** 2).sum(axis=-1))
    if w is not None:
        assert w.shape[0] == pred.shape[0]
        n = np.asarray(w, dtype=np.float64) * n
    return float(np.mean(n))


def mpjpe_grad(pred, target, w=None):
    """d mpjpe / d pred = w * (p-q)/||p-q|| / (B*T*J)   (0 where p == q, torch's sub-gradient of the norm)."""
    d = pred - target
    n = np.sqrt((d.astype(np.float64) ** 2).sum(axis=-1, keepdims=True))
    cnt = d.size // d.shape[-1]
    g = np.where(n > 0, d / np.maximum(n, 1e-300), 0.0) / cnt
    if w is not None:
        g = g * np.broadcast_to(np.asarray(w, dtype=np.float64), d.shape[:-1])[..., None]
    return g.astype(pred.dtype)


# --------------------------------------------------------------------------------------
# camera projection (reference common/camera.py:37-67, 69-90), BASELINE config 5
# --------------------------------------------------------------------------------------
def project_to_2d(X, cam, linear=False):
    """X [N, *, 3] camera-space points, cam [N, 9] = (f2, c2, k3, p2)."""
    assert X.shape[-1] == 3 and cam.ndim == 2 and cam.shape[-1] == 9 and X.shape[0] == cam.shape[0]
    cp = cam.reshape((cam.shape[0],) + (1,) * (X.ndim - 2) + (9,))
    f, c, k, p = cp[..., :2], cp[..., 2:4], cp[..., 4:7], cp[..., 7:]
    XX = np.clip(X[..., :2] / X[..., 2:], -1, 1)
    if linear:
        return f * XX + c
    r2 = (XX ** 2).sum(axis=-1, keepdims=True)
    radial = 1 + (k * np.concatenate((r2, r2 ** 2, r2 ** 3), axis=-1)).sum(axis=-1, keepdims=True)
    tan = (p * XX).sum(axis=-1, keepdims=True)
    return f * (XX * (radial + tan) + p * r2) + c


def project_to_2d_grad(X, cam, gout, linear=False):
    """dL/dX for project_to_2d (camera params carry no grad, run.py:328-331)."""
    cp = cam.reshape((cam.shape[0],) + (1,) * (X.ndim - 2) + (9,))
    fx, fy = cp[..., 0], cp[..., 1]
    k1, k2, k3, p1, p2 = (cp[..., 4], cp[..., 5], cp[..., 6], cp[..., 7], cp[..., 8])
    if linear:
        k1 = k2 = k3 = p1 = p2 = np.zeros_like(fx)
    x, y, z = X[..., 0], X[..., 1], X[..., 2]
    u, v = x / z, y / z
    a, bq = np.clip(u, -1, 1), np.clip(v, -1, 1)
    r2 = a * a + bq * bq
    s = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3 + p1 * a + p2 * bq
    hx, hy = fx * gout[..., 0], fy * gout[..., 1]
    ds = hx * a + hy * bq
    dr2 = hx * p1 + hy * p2 + ds * (k1 + 2 * k2 * r2 + 3 * k3 * r2 ** 2)
    da = hx * s + ds * p1 + 2 * a * dr2
    db = hy * s + ds * p2 + 2 * bq * dr2
    du = da * ((u >= -1) & (u <= 1))
    dv = db * ((v >= -1) & (v <= 1))
    return np.stack((du / z, dv / z, -(du * u + dv * v) / z), axis=-1).astype(X.dtype)
