#!/usr/bin/env python3
"""A synthetic Human3.6M-shaped dataset, so that the reference's UNMODIFIED run.py can be executed end to end.

run.py needs exactly two files (run.py:38-70, common/h36m_dataset.py:233-246):

  data/data_3d_h36m.npz        positions_3d = {subject: {action: float32 [N, 32, 3]}}   world coordinates, metres
  data/data_2d_h36m_<k>.npz    positions_2d = {subject: {action: [4 x float32 [N, 17, 2]]}}  pixel coordinates per camera,
                               metadata = {layout_name, num_joints, keypoints_symmetry}

The real archives cannot be downloaded here.  This tool writes both from a procedurally animated 32-joint figure (the
skeleton topology and the four cameras per subject are taken from the reference checkout given by --reference: the dataset
class itself is what run.py will load the files with).  The 2D file is made the way data/prepare_data_h36m.py makes its
ground-truth detections (world -> camera -> project_to_2d -> pixel coordinates) plus Gaussian detector noise, so the 3D
targets are a learnable function of the 2D inputs.  Every sequence of one action has the same length, so an evaluation
pass sees only --actions distinct shapes (MIOpen searches per shape when the reference classes run on ROCm).

    python tools/make_synth_h36m.py --reference /path/to/VideoPose3D [--out data] [--keypoints synth] [--frames 900,1100,1300]
"""
import argparse
import os
import sys

import numpy as np

SUBJECTS = ["S1", "S5", "S6", "S7", "S8", "S9", "S11"]           # run.py's default train / test split (arguments.py:16-18)
STATIC = [4, 5, 9, 10, 11, 16, 20, 21, 22, 23, 24, 28, 29, 30, 31]       # h36m_dataset.py:250 (removed -> 17 joints)


def animate(parents, n_frames, rng, fps=50.0):
    """[n_frames, J, 3] world positions in metres: a root that wanders inside a 2 m box at hip height and, down the kinematic
    tree, bones of fixed length whose directions swing smoothly (two sinusoids per bone)."""
    n_j = len(parents)
    t = np.arange(n_frames, dtype=np.float64)[:, None] / fps
    pos = np.zeros((n_frames, n_j, 3))
    f = rng.uniform(0.05, 0.25, size=(3, 2))
    ph = rng.uniform(0, 2 * np.pi, size=(3, 2))
    root = np.stack([0.8 * np.sin(2 * np.pi * f[0, 0] * t[:, 0] + ph[0, 0]) + 0.2 * np.sin(2 * np.pi * 3 * f[0, 1] * t[:, 0] + ph[0, 1]),
                     0.8 * np.sin(2 * np.pi * f[1, 0] * t[:, 0] + ph[1, 0]) + 0.2 * np.sin(2 * np.pi * 3 * f[1, 1] * t[:, 0] + ph[1, 1]),
                     0.92 + 0.05 * np.sin(2 * np.pi * 4 * f[2, 0] * t[:, 0] + ph[2, 0])], axis=1)
    pos[:, 0] = root
    rest = rng.standard_normal((n_j, 3))
    rest[:, 2] *= 1.5                                              # mostly vertical bones
    length = rng.uniform(0.10, 0.45, size=n_j)
    for j in range(1, n_j):
        p = parents[j]
        assert 0 <= p < j
        w = rng.uniform(0.2, 1.5, size=2)
        ax = rng.standard_normal((2, 3))
        phj = rng.uniform(0, 2 * np.pi, size=2)
        d = rest[j][None, :] + 0.6 * np.sin(2 * np.pi * w[0] * t + phj[0]) * ax[0][None, :] \
            + 0.3 * np.sin(2 * np.pi * w[1] * t + phj[1]) * ax[1][None, :]
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pos[:, j] = pos[:, p] + length[j] * d
    return pos.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="VideoPose3D checkout (its common/ package provides skeleton + cameras)")
    ap.add_argument("--out", default=None, help="output directory (default: <reference>/data)")
    ap.add_argument("--keypoints", default="synth", help="name k of data_2d_h36m_<k>.npz (run.py -k <k>)")
    ap.add_argument("--frames", default="900,1100,1300", help="frames per action (one action per entry)")
    ap.add_argument("--noise-px", type=float, default=2.0, help="std of the detector noise added to the 2D keypoints, pixels")
    ap.add_argument("--seed", type=int, default=20260922)
    args = ap.parse_args()

    ref = os.path.abspath(args.reference)
    out = os.path.abspath(args.out or os.path.join(ref, "data"))
    os.makedirs(out, exist_ok=True)
    sys.path.insert(0, ref)
    import torch  # noqa: F401  (common.camera imports it)
    from common.camera import image_coordinates, project_to_2d, world_to_camera
    from common.h36m_dataset import Human36mDataset, h36m_skeleton
    from common.utils import wrap

    parents = [int(p) for p in h36m_skeleton.parents()]
    assert len(parents) == 32
    rng = np.random.RandomState(args.seed)
    actions = ["Walking", "Sitting", "Directions", "Eating", "Greeting", "Posing", "Waiting", "Phoning"]
    frames = [int(v) for v in args.frames.split(",")]
    assert len(frames) <= len(actions)
    pos3d = {s: {"%s%s" % (actions[a], suffix): animate(parents, n + 0, rng)
                 for a, n in enumerate(frames) for suffix in ("", " 1")} for s in SUBJECTS}
    path3d = os.path.join(out, "data_3d_h36m.npz")
    np.savez_compressed(path3d, positions_3d=pos3d)

    # the 2D "detections": what data/prepare_data_h36m.py:142-160 computes as ground-truth 2D, + noise
    dataset = Human36mDataset(path3d)                               # 17 joints, cameras with normalised intrinsics
    pos2d = {}
    for s in dataset.subjects():
        pos2d[s] = {}
        for a in dataset[s].keys():
            anim = dataset[s][a]
            views = []
            for cam in anim["cameras"]:
                p3 = world_to_camera(anim["positions"], R=cam["orientation"], t=cam["translation"])
                p2 = wrap(project_to_2d, p3, cam["intrinsic"], unsqueeze=True)
                px = image_coordinates(p2, w=cam["res_w"], h=cam["res_h"])
                px = px + rng.standard_normal(px.shape) * args.noise_px
                views.append(px.astype(np.float32))
            pos2d[s][a] = views
    meta = {"layout_name": "h36m", "num_joints": 17,
            "keypoints_symmetry": [list(dataset.skeleton().joints_left()), list(dataset.skeleton().joints_right())]}
    path2d = os.path.join(out, "data_2d_h36m_%s.npz" % args.keypoints)
    np.savez_compressed(path2d, positions_2d=pos2d, metadata=meta)
    n_seq = sum(len(v) * 4 for v in pos2d.values())
    n_fr = sum(x.shape[0] for v in pos2d.values() for views in v.values() for x in views)
    print("wrote %s and %s: %d subjects, %d camera sequences, %d frames" % (path3d, path2d, len(pos2d), n_seq, n_fr))


if __name__ == "__main__":
    main()
