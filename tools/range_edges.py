#!/usr/bin/env python3
"""The split-fp16 ("f16x3") arithmetic at the edges of its block-exponent format: ONE exponent per tensor, taken from a
guaranteed bound; elements below 2^-17 of the bound lose bits (csrc/vp3d_s16.h).  This tool makes the intra-tensor dynamic
range adversarial -- the reference's BatchNorm affine is unconstrained (common/model.py:32,117-119) -- and reports, for both
GEMM arithmetics against the float64 oracle on the same weights / inputs (dropout 0), with the guard of
videopose3d_amd/range_guard.py OFF (the raw format is what is measured; --guard runs it as shipped):

  * output MPJPE,
  * per gradient tensor: max-norm error (|d| / max|ref|) and the WORST ROW's own relative error (|d|_row / max|ref|_row) --
    the per-tensor max-norm hides small channels next to a hot one.  (The row-relative figure is ill-conditioned where a
    row's reference gradient is a difference of large sums -- an inner BatchNorm's bias behind another BatchNorm --: read it
    against the fp32 engine's figure beside it, not against 0.)

Cases (tests/util.py: range_edge_state; strided class, arc 3,3,3, B = 64): gamma_c / beta_c of one BatchNorm x 2^s, the same
channel hot in EVERY BatchNorm, one conv-weight output row x 2^s, one conv-weight input column x 2^s (hot dy), one input
joint x 1e4.

    python tools/range_edges.py [--channels 128 1024] [--json out.json] [--guard]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videopose3d_amd as V                      # noqa: E402
from videopose3d_amd import engine, range_guard  # noqa: E402
from tests import util as U                      # noqa: E402  (case recipes shared with tests/test_gpu_s16.py; oracle = checker)

DEV = torch.device("cuda:0")


def run_case(channels, case, s, joint_scale=None):
    sd = U.range_edge_state(channels, case, s)
    x, tgt = U.range_edge_batch(64, joint_scale)
    yo, go = U.range_edge_oracle(sd, x, tgt)
    out = {}
    for math in ("f32", "f16x3"):
        m = V.TemporalModelOptimized1f(17, 2, 17, U.RANGE_FW, dropout=0.0, channels=channels).to(DEV).train()
        m.math = math
        m.load_state_dict(sd)
        n16 = engine.ENGINE_CALLS["s16_train"]
        r = U.range_edge_errors(m, x, tgt, yo, go)
        r["ran_on_s16"] = engine.ENGINE_CALLS["s16_train"] > n16
        r["guard"] = range_guard.status(m)["last"]
        out[math] = r
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="*", default=[128, 1024])
    ap.add_argument("--json", default="")
    ap.add_argument("--guard", action="store_true", help="leave the dynamic-range guard on (default: off, the raw format)")
    a = ap.parse_args()
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    os.environ["VP3D_RANGE_GUARD"] = "1" if a.guard else "0"
    rows = []
    cases = [("none", 0, None)]
    for case in ("gamma", "gamma_all_layers", "beta", "w_row", "w_col"):
        for s in (8, 14, 20):
            cases.append((case, s, None))
    cases.append(("w_row", 12, None))
    cases.append(("none", 0, 1e4))
    print("guard %s" % ("ON (as shipped)" if a.guard else "OFF (raw format)"))
    print("%-5s %-18s %4s | %-9s %-9s %-9s | %-9s %-9s %-9s | f16x3 ran on | worst row-relative gradient (f16x3)" %
          ("C", "case", "2^s", "mpjpe32", "gmax32", "grow32", "mpjpe16", "gmax16", "grow16"))
    for c in a.channels:
        for case, s, xj in cases:
            r = run_case(c, case, s, joint_scale=xj)
            name = case if xj is None else "joint x %g" % xj
            print("%-5d %-18s %4d | %.3e %.3e %.3e | %.3e %.3e %.3e | %-12s | %s%s" %
                  (c, name, s, r["f32"]["mpjpe"], r["f32"]["grad_maxnorm"], r["f32"]["grad_rowrel"],
                   r["f16x3"]["mpjpe"], r["f16x3"]["grad_maxnorm"], r["f16x3"]["grad_rowrel"],
                   "split-fp16" if r["f16x3"]["ran_on_s16"] else "fp32 (guard)", r["f16x3"]["grad_rowrel_at"],
                   "" if r["f16x3"]["finite"] else "  NON-FINITE"), flush=True)
            rows.append(dict(channels=c, case=name, log2_factor=s, **{"f32_" + k: v for k, v in r["f32"].items()},
                             **{"f16x3_" + k: v for k, v in r["f16x3"].items()}))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
