"""CPU oracle for the step-level callers around the temporal stack.  TEST INFRASTRUCTURE ONLY
(same rules as temporal_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it).

Independent numpy restatements, each following the reference lines it cites:

  chunk_pairs / gather_chunks     reference common/generators.py:39-48 (pair list), :105-149 (batch assembly)
  unchunked_batch                 reference common/generators.py:216-239
  tta_fold                        reference run.py:677-680
  adam_step                       third-party: torch.optim.Adam (torch/optim/adam.py `_single_tensor_adam`, torch
                                  2.10.0 in this container; the reference pins only "PyTorch >= 0.4.0") as called at
                                  reference run.py:252,264,420 with amsgrad=True
  (mpjpe / weighted mpjpe + gradients live in temporal_oracle.py: mpjpe, mpjpe_grad)

Pinned against fixtures produced by the reference's own generators.py / loss.py and by torch.optim.Adam run in the
build container: tests/golden/make_golden_step.py -> tests/golden/step_*.npz, checked by
tests/test_oracle_golden.py.
"""
from __future__ import annotations

import numpy as np


def chunk_pairs(seq_lens, chunk_length, augment):
    """generators.py:39-48: (seq, start, end, flip) rows; per sequence all plain chunks, then all mirrored ones."""
    rows = []
    for i, n in enumerate(seq_lens):
        n_chunks = (n + chunk_length - 1) // chunk_length
        offset = (n_chunks * chunk_length - n) // 2
        for flip in ([0, 1] if augment else [0]):
            for k in range(n_chunks):
                s = k * chunk_length - offset
                rows.append((i, s, s + chunk_length, flip))
    return np.asarray(rows, dtype=np.int64).reshape(-1, 4)


def _flip(a, left, right):
    a = a.copy()
    a[..., 0] *= -1
    if left is not None:
        a[:, list(left) + list(right)] = a[:, list(right) + list(left)]
    return a


def gather_chunks(chunks, cameras, poses_3d, poses_2d, chunk_length, pad, causal_shift, kps_left=None,
                  kps_right=None, joints_left=None, joints_right=None):
    """generators.py:105-149 for a list of (seq, start_3d, end_3d, flip) rows -> (cam, batch_3d, batch_2d), float32."""
    b2, b3, bc = [], [], []
    for seq, start_3d, end_3d, flip in chunks:
        start_2d = start_3d - pad - causal_shift
        end_2d = end_3d + pad - causal_shift
        s2 = np.asarray(poses_2d[seq], dtype=np.float32)
        idx = np.clip(np.arange(start_2d, end_2d), 0, s2.shape[0] - 1)         # np.pad(..., 'edge')
        x = s2[idx]
        b2.append(_flip(x, kps_left, kps_right) if flip else x)
        if poses_3d is not None:
            s3 = np.asarray(poses_3d[seq], dtype=np.float32)
            idx = np.clip(np.arange(start_3d, end_3d), 0, s3.shape[0] - 1)
            y = s3[idx]
            b3.append(_flip(y, joints_left, joints_right) if flip else y)
        if cameras is not None:
            c = np.asarray(cameras[seq], dtype=np.float32).copy()
            if flip:
                c[2] *= -1
                c[7] *= -1
            bc.append(c)
    return (np.stack(bc) if bc else None, np.stack(b3) if b3 else None, np.stack(b2))


def unchunked_batch(seq, cameras, poses_3d, poses_2d, pad, causal_shift, augment, kps_left=None, kps_right=None,
                    joints_left=None, joints_right=None):
    """generators.py:216-239 for one sequence."""
    n = poses_2d[seq].shape[0]
    rows = [(seq, 0, n, 0)] + ([(seq, 0, n, 1)] if augment else [])
    return gather_chunks(rows, cameras, poses_3d, poses_2d, n, pad, causal_shift, kps_left, kps_right, joints_left,
                         joints_right)


def tta_fold(pred, joints_left=None, joints_right=None):
    """run.py:677-680: pred [2,T,J,3] -> [1,T,J,3]."""
    p = np.array(pred, dtype=np.float32, copy=True)
    p[1, :, :, 0] *= -1
    if joints_left is not None:
        p[1][:, list(joints_left) + list(joints_right)] = p[1][:, list(joints_right) + list(joints_left)]
    return ((p[0] + p[1]) / np.float32(2.0))[None]


def adam_step(p, g, m, v, vmax, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=True):
    """One torch.optim.Adam update (single-tensor path) in float32 numpy; `step` is the 1-based update index.
    Returns new (p, m, v, vmax)."""
    f = np.float32
    b1, b2 = betas
    g = g.astype(f)
    if weight_decay != 0:
        g = g + f(weight_decay) * p
    m = m + f(1.0 - b1) * (g - m)                              # exp_avg.lerp_(grad, 1 - beta1)
    v = v * f(b2) + (f(1.0 - b2) * g) * g                      # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    step_size = f(lr / bc1)
    bc2_sqrt = f(bc2 ** 0.5)
    if amsgrad:
        vmax = np.maximum(vmax, v)
        denom = np.sqrt(vmax) / bc2_sqrt + f(eps)
    else:
        denom = np.sqrt(v) / bc2_sqrt + f(eps)
    p = p - step_size * (m / denom)                            # param.addcdiv_(exp_avg, denom, value=-step_size)
    return p.astype(f), m.astype(f), v.astype(f), (vmax.astype(f) if vmax is not None else None)
