"""Time the HBM-bound producers of the split-fp16 engine at the training step's layer sizes (run on the GPU box):
forward activation producer, BatchNorm-backward reduce + apply, with the dropout mask regenerated (Philox) or read from
the forward's activation bits.  GB/s = algorithmic bytes (every operand once) / time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videopose3d_amd import ops, ops_s16 as S          # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def main():
    c = 1024
    for b, t, taps in ((1024, 81, 3), (1024, 27, 3), (1024, 27, 1), (1024, 9, 3)):
        m = b * t
        y = torch.randn(b, t, c, device=DEV)
        go = torch.randn(b, t, c, device=DEV) * 1e-4
        coef = torch.stack([1 + 0.2 * torch.randn(c), 0.1 * torch.randn(c), 0.05 * torch.randn(c), 1 + 0.1 * torch.rand(c)]).to(DEV)
        bd = S.new_bound(DEV)
        bd[0] = 64.0
        gb = S.amax(go)
        mb = m * c * 4 / 1e6
        for p in (0.0, 0.25):
            drop = ops.make_dropout(p, 1234, 5, 2)
            bits = S.new_act_bits(m, c, DEV)
            f_plain = timed(lambda: S.bn_act_fwd(y, coef, drop, None, bd, t_taps=taps))
            f_bits = timed(lambda: S.bn_act_fwd(y, coef, drop, None, bd, t_taps=taps, act_bits=bits))
            for use in (None, bits):
                dyb = S.new_bound(DEV)
                tot = timed(lambda: S.bn_act_bwd(go, gb, y, coef, drop, p, dyb, act_bits=use))
                tot_nr = timed(lambda: S.bn_act_bwd(go, gb, y, coef, drop, p, dyb, act_bits=use, want_rows=False))
                print("M=%6d taps=%d p=%.2f bits=%d | fwd %.0f us (%.0f GB/s) fwd+bits %.0f us | bwd reduce+finalize+apply: "
                      "%.0f us (%.0f GB/s), T only %.0f us (%.0f GB/s)"
                      % (m, taps, p, use is not None, f_plain, 3 * mb / f_plain * 1e3, f_bits, tot, 7 * mb / tot * 1e3, tot_nr,
                         6 * mb / tot_nr * 1e3), flush=True)


if __name__ == "__main__":
    main()
