#!/usr/bin/env python3
"""Randomised cross-check of the two GEMM engines on whole training steps and eval forwards: model class, arc, causal,
dense, channels, joint counts, batch / window sizes, dropout, input gradients.  Same weights and dropout stream in both;
outputs, every gradient and the running statistics must agree (the exact-fp32 engine is held to the oracle by the parity
suite).  python tools/fuzz_engines.py [n_cases] [seed]"""
import copy
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import engine  # noqa: E402

dev = "cuda:0"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
bad = 0
for case in range(n_cases):
    strided = rnd.random() < 0.6
    fw = rnd.choice([[3], [3, 3], [3, 3, 3], [3, 5], [5, 3], [3, 3, 3, 3], [1, 3], [3, 1, 3]])
    causal = rnd.random() < 0.3
    dense = (not strided) and rnd.random() < 0.2
    big = os.environ.get("VP3D_FUZZ_BIG", "0") == "1"      # larger shapes: reach the launches that carry the fused sums
    c = rnd.choice([256, 512] if big else [64, 128, 192, 256])
    j_in, j_out = rnd.choice([17, 15, 10, 5, 16]), rnd.choice([17, 1, 15])
    p = rnd.choice([0.0, 0.25, 0.5])
    b = rnd.choice([130, 300, 512, 1000] if big else [2, 3, 16, 33, 128])
    rf = 1
    for f in fw:
        rf *= f
    t = rf if strided else rf + rnd.choice([0, 1, 7, 20])
    need_dx = rnd.random() < 0.25
    tag = "%s fw=%s causal=%d dense=%d C=%d J=%d->%d p=%.2f B=%d T=%d dx=%d" % (
        "strided" if strided else "dilated", fw, causal, dense, c, j_in, j_out, p, b, t, need_dx)
    try:
        torch.manual_seed(case)
        V.set_default_math("f32")
        try:
            if strided:
                m32 = V.TemporalModelOptimized1f(j_in, 2, j_out, fw, causal=causal, dropout=p, channels=c)
            else:
                m32 = V.TemporalModel(j_in, 2, j_out, fw, causal=causal, dropout=p, channels=c, dense=dense)
        finally:
            V.set_default_math(None)
        m32 = m32.to(dev).train()
        m16 = copy.deepcopy(m32)
        m16.math = "f16x3"
        for m in (m32, m16):
            m._drop_seed, m._drop_calls = 1234 + case, 0
        x = (torch.randn(b, t, j_in, 2, device=dev) * 0.5).clamp(-1, 1)
        outs, dxs = [], []
        n16 = engine.ENGINE_CALLS["s16_train"]
        for m in (m32, m16):
            xi = x.clone().requires_grad_(need_dx)
            y = m(xi)
            tgt = torch.zeros_like(y)
            torch.mean(torch.norm(y - tgt + 0.1, dim=3)).backward()
            outs.append(y.detach())
            dxs.append(xi.grad)
        ran16 = engine.ENGINE_CALLS["s16_train"] > n16
        err = float(torch.mean(torch.norm(outs[0] - outs[1], dim=3)))
        # gradients: relative L2 error per tensor.  (Entry-wise maxima are not comparable across arithmetics: a ReLU
        # pre-activation within rounding of zero may fall on different sides and move a whole row of a weight gradient by one
        # term -- DESIGN.md 2; the parity suite pins those decisions against the oracle, here they only add ~1e-3 of L2.)
        gerr, flips = 0.0, 0
        for a, q in zip(m16.parameters(), m32.parameters()):
            e = (a.grad - q.grad).abs()
            flips += int((e > 1e-3 * (float(q.grad.abs().max()) + 1e-30)).sum())
            gerr = max(gerr, float(e.norm() / (q.grad.norm() + 1e-30)))
        berr = max(float((a.float() - q.float()).abs().max() / (q.float().abs().max() + 1e-30))
                   for a, q in zip(m16.buffers(), m32.buffers()))
        dxerr = float((dxs[0] - dxs[1]).norm() / (dxs[0].norm() + 1e-30)) if need_dx else 0.0
        m32.eval(), m16.eval()
        te = t + rnd.choice([0, 5, 40])
        xe = (torch.randn(2, te, j_in, 2, device=dev) * 0.5).clamp(-1, 1)
        with torch.no_grad():
            eerr = float(torch.mean(torch.norm(m32(xe) - m16(xe), dim=3)))
        gtol = 5e-3 if b >= 8 else 5e-2              # (BatchNorm over 2-3 samples: x-hat = +-1, every difference is amplified)
        ok = err < (2e-5 if b >= 8 else 2e-4) and gerr < gtol and berr < 1e-4 and dxerr < gtol and eerr < 2e-5
        print("%s %-86s s16=%d out %.1e grad %.1e (%d entries off by > 1e-3) stats %.1e dx %.1e eval %.1e" % (
            "ok  " if ok else "FAIL", tag, ran16, err, gerr, flips, berr, dxerr, eerr), flush=True)
        bad += 0 if ok else 1
    except Exception as e:                               # noqa: BLE001  (report and go on: this is a bug hunt)
        bad += 1
        print("EXC  %-86s %s" % (tag, repr(e)[:300]), flush=True)
from videopose3d_amd import ops_s16 as _S  # noqa: E402
print("dgrad launches with the fused BatchNorm-backward sums: %d" % _S.RED_CALLS["n"])
print("%d / %d cases failed" % (bad, n_cases))
sys.exit(1 if bad else 0)
