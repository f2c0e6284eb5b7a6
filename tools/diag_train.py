#!/usr/bin/env python3
"""Layer-by-layer comparison of the HIP training step with the oracle (debug aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import engine, ops  # noqa: E402
from oracle import temporal_oracle as O  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def run(fw, batch, seed=2, dtype=np.float32):
    gen = torch.Generator().manual_seed(12)
    torch.manual_seed(seed)
    m = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.0, channels=1024)
    sd = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    rf = m.receptive_field()
    x = (torch.randn(batch, rf, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
    tgt = torch.randn(batch, 1, 17, 3, generator=gen) * 0.3
    m = m.to(DEV).train()
    x3 = x.to(DEV).view(batch, rf, -1)
    out, saved = engine.forward_train(m, x3, save=True)
    yo, cache, _ = O.forward(sd, x.numpy(), fw, kind="strided", training=True, dtype=dtype)
    print("== fw", fw, "B", batch, "oracle dtype", dtype.__name__)
    print("out rel", rel(out.cpu().numpy().reshape(yo.shape), yo))
    for i, (s, L) in enumerate(zip(saved["layers"], cache["layers"])):
        print(" layer %d  invstd rel %.2e  M=%d" % (i, rel(s.coef[3].cpu().numpy(), L["invstd"]), s.y.shape[0] * s.y.shape[1]))
    gout = O.mpjpe_grad(yo, tgt.numpy().astype(dtype))
    trace = {}
    go = O.backward(cache, gout, trace)
    g3 = torch.from_numpy(gout.astype(np.float32)).to(DEV).view(out.shape)
    # replicate engine.backward_train with comparisons
    plan = m._plan
    L = saved["layers"]
    h_last = saved["h_last"]
    b, t_out, _ = h_last.shape
    dh = ops.conv_dgrad(g3, saved["wts"], plan.shrink, t_out)
    print(" shrink.weight grad rel %.2e" % rel(ops.conv_wgrad(g3, h_last, plan.shrink).cpu().numpy(), go["shrink.weight"]))

    def act_bwd(idx, gin):
        s = L[idx]
        print("   go%d rel %.2e" % (idx, rel(gin.cpu().numpy(), trace["go%d" % idx])))
        dy, dgam, dbet = ops.bn_act_bwd(gin, s.y, s.coef, s.drop)
        pre = "expand_bn" if idx == 0 else "layers_bn.%d" % (idx - 1)
        cn = "expand_conv.weight" if idx == 0 else "layers_conv.%d.weight" % (idx - 1)
        dw = ops.conv_wgrad(dy, s.x, plan.convs[idx], rows_kpad=s.kpad)
        dw2 = ops.conv_wgrad(dy, s.x, plan.convs[idx], rows_kpad=s.kpad)
        # wgrad from the ORACLE's dy (isolates the wgrad kernel) and determinism
        dyo = torch.from_numpy(trace["dy%d" % idx].astype(np.float32)).to(DEV)
        dwo = ops.conv_wgrad(dyo, s.x, plan.convs[idx], rows_kpad=s.kpad)
        print("   layer %d: dy rel %.2e dgamma %.2e dbeta %.2e | dW rel %.2e (kernel-only %.2e, rerun-equal %s)" % (
            idx, rel(dy.cpu().numpy(), trace["dy%d" % idx]), rel(dgam.cpu().numpy(), go[pre + ".weight"]),
            rel(dbet.cpu().numpy(), go[pre + ".bias"]), rel(dw.cpu().numpy(), go[cn]), rel(dwo.cpu().numpy(), go[cn]),
            bool(torch.equal(dw, dw2))))
        return dy

    for i in reversed(range(plan.n_blocks)):
        i1, i2 = 1 + 2 * i, 2 + 2 * i
        dy2 = act_bwd(i2, dh)
        da1 = ops.conv_dgrad(dy2, L[i2].wt, plan.convs[i2], L[i2].t_in)
        dy1 = act_bwd(i1, da1)
        dh = ops.conv_dgrad(dy1, L[i1].wt, plan.convs[i1], L[i1].t_in, residual=(dh, plan.res[i]))
    act_bwd(0, dh)


if __name__ == "__main__":
    run([3, 3, 3], 48)
    run([3, 3, 3], 48, dtype=np.float64)
    run([3, 3, 3, 3, 3], 1024)
