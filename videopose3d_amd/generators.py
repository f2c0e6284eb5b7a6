"""Device-resident batch generators: same constructor arguments, helper API and iteration order as reference
common/generators.py (``ChunkedGenerator`` :11-166, ``UnchunkedGenerator`` :168-239), but the dataset is uploaded to
HBM once and every batch is assembled by one HIP gather kernel (vp3d_gather_chunks) instead of the reference's
per-sample numpy loop over float64 staging buffers followed by a 34 MB host->device copy per step.

What stays on the host (bit-for-bit the reference's logic): the (seq, start, end, flip) pair list
(generators.py:39-48), the ``np.random.RandomState(random_seed).permutation`` of it (:89-97), the ``endless`` state
machine (:150-166).  What moves to the GPU: edge padding (:105-118, 126-135), mirroring (:120-123, 137-141) and the
camera-parameter flip (:144-149).

Differences a caller sees (INTEGRATION.md shows the two-line run.py change):
  * ``next_epoch()`` yields ``torch.float32`` CUDA tensors (fresh per batch) instead of float64 numpy views, so
    run.py's ``torch.from_numpy(batch.astype('float32')).cuda()`` lines become no-ops to delete;
  * ``shard=(rank, world)`` yields only this rank's contiguous slice of every batch (data parallelism:
    every rank builds the identical generator -> identical permutation -> no communication).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import check
from .dp import shard_bounds, shardable


def flip_permutation(n_joints: int, left: Optional[Sequence[int]], right: Optional[Sequence[int]]) -> np.ndarray:
    """The reference's mirrored assignment ``a[:, left + right] = a[:, right + left]`` as a gather map:
    perm[dst] = src (identity elsewhere; later duplicates win, as numpy's fancy assignment does)."""
    perm = np.arange(n_joints, dtype=np.int32)
    if left is not None and right is not None:
        for dst, src in zip(list(left) + list(right), list(right) + list(left)):
            perm[dst] = src
    return perm


class _Resident:
    """Concatenated fp32 copies of the per-video arrays in HBM + frame offsets."""

    def __init__(self, cameras, poses_3d, poses_2d, device):
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _lib.Vp3dError("device generators keep the dataset in GPU memory; got device %s" % self.device)
        lens = [int(p.shape[0]) for p in poses_2d]
        self.lens = lens
        off = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        self.seq_off_host = off
        self.seq_off = torch.from_numpy(off).to(self.device)

        def cat(arrs):
            a = np.concatenate([np.asarray(x, dtype=np.float32) for x in arrs], axis=0)
            return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

        self.p2 = cat(poses_2d)
        self.j2, self.f2 = int(poses_2d[0].shape[-2]), int(poses_2d[0].shape[-1])
        self.p3 = None
        self.j3 = self.f3 = 0
        if poses_3d is not None and len(poses_3d):
            for a, b in zip(poses_3d, poses_2d):
                assert a.shape[0] == b.shape[0], "3D and 2D sequences must have the same length"
            self.p3 = cat(poses_3d)
            self.j3, self.f3 = int(poses_3d[0].shape[-2]), int(poses_3d[0].shape[-1])
        self.cams = None
        self.cam_dim = 0
        if cameras is not None and len(cameras):
            self.cams = torch.from_numpy(np.stack([np.asarray(c, dtype=np.float32) for c in cameras])).to(self.device)
            self.cam_dim = int(self.cams.shape[-1])

    def gather(self, chunks_dev: torch.Tensor, n: int, chunk_length: int, pad: int, causal_shift: int,
               perm2: Optional[torch.Tensor], perm3: Optional[torch.Tensor], want_3d=True, want_cam=True):
        """chunks_dev: device int32 [n,3] (seq, start, flip).  Returns (cam, batch_3d, batch_2d) device tensors."""
        dev = self.device
        o2 = torch.empty((n, chunk_length + 2 * pad, self.j2, self.f2), dtype=torch.float32, device=dev)
        o3 = torch.empty((n, chunk_length, self.j3, self.f3), dtype=torch.float32, device=dev) \
            if (self.p3 is not None and want_3d) else None
        oc = torch.empty((n, self.cam_dim), dtype=torch.float32, device=dev) \
            if (self.cams is not None and want_cam) else None
        g = _lib.Gather()
        g.n_chunks = n
        g.chunks = chunks_dev.data_ptr()
        g.seq_off = self.seq_off.data_ptr()
        g.poses_2d = self.p2.data_ptr()
        g.j2, g.f2 = self.j2, self.f2
        g.kps_perm = perm2.data_ptr() if perm2 is not None else None
        g.poses_3d = self.p3.data_ptr() if o3 is not None else None
        g.j3, g.f3 = self.j3, self.f3
        g.joints_perm = perm3.data_ptr() if perm3 is not None else None
        g.cameras = self.cams.data_ptr() if oc is not None else None
        g.cam_dim = self.cam_dim
        g.chunk_length, g.pad, g.causal_shift = chunk_length, pad, causal_shift
        g.out_2d = o2.data_ptr()
        g.out_3d = o3.data_ptr() if o3 is not None else None
        g.out_cam = oc.data_ptr() if oc is not None else None
        with torch.cuda.device(dev):
            check(_lib.lib().vp3d_gather_chunks(ops._stream(), C.byref(g)),
                  "vp3d_gather_chunks")
        return oc, o3, o2


def _perm_dev(n, left, right, device):
    if left is None or right is None:
        return None
    return torch.from_numpy(flip_permutation(n, left, right)).to(device)


class ChunkedGenerator:
    """Batched data generator used for training (reference generators.py:11-166): sequences are split into
    equal-length chunks and padded as necessary; batches are assembled on the GPU."""

    def __init__(self, batch_size, cameras, poses_3d, poses_2d, chunk_length, pad=0, causal_shift=0, shuffle=True,
                 random_seed=1234, augment=False, kps_left=None, kps_right=None, joints_left=None, joints_right=None,
                 endless=False, device=None, shard: Optional[Tuple[int, int]] = None):
        assert poses_3d is None or len(poses_3d) == len(poses_2d), (len(poses_3d), len(poses_2d))
        assert cameras is None or len(cameras) == len(poses_2d)

        # Build lineage info (generators.py:39-48): (seq_idx, start_frame, end_frame, flip)
        rows = []
        for i in range(len(poses_2d)):
            n_frames = poses_2d[i].shape[0]
            n_chunks = (n_frames + chunk_length - 1) // chunk_length
            offset = (n_chunks * chunk_length - n_frames) // 2
            bounds = np.arange(n_chunks + 1) * chunk_length - offset
            blk = np.stack([np.full(n_chunks, i, dtype=np.int64), bounds[:-1], bounds[1:],
                            np.zeros(n_chunks, dtype=np.int64)], axis=1)
            rows.append(blk)
            if augment:
                flipped = blk.copy()
                flipped[:, 3] = 1
                rows.append(flipped)
        # same dtype / shape as np.array(list_of_tuples) in the reference, so RandomState.permutation takes the
        # same code path and consumes the same random stream
        self.pairs = np.concatenate(rows, axis=0) if rows else np.zeros((0, 4), dtype=np.int64)

        self.num_batches = (len(self.pairs) + batch_size - 1) // batch_size
        self.batch_size = batch_size
        self.random = np.random.RandomState(random_seed)
        self.shuffle = shuffle
        self.pad = pad
        self.causal_shift = causal_shift
        self.endless = endless
        self.state = None
        self.chunk_length = chunk_length

        self.cameras = cameras
        self.poses_3d = poses_3d
        self.poses_2d = poses_2d

        self.augment = augment
        self.kps_left = kps_left
        self.kps_right = kps_right
        self.joints_left = joints_left
        self.joints_right = joints_right

        self.shard = shard
        self._res = _Resident(cameras, poses_3d, poses_2d, device)
        self._perm2 = _perm_dev(self._res.j2, kps_left, kps_right, self._res.device) if augment else None
        self._perm3 = _perm_dev(self._res.j3, joints_left, joints_right, self._res.device) \
            if (augment and self._res.p3 is not None) else None
        self._table = None          # (id of the host pairs array, device int32 [N,3])

    def num_frames(self):
        return self.num_batches * self.batch_size

    def random_state(self):
        return self.random

    def set_random_state(self, random):
        self.random = random

    def augment_enabled(self):
        return self.augment

    def next_pairs(self):
        """(first batch index, chunk table of the epoch to run): the parked position of an endless generator if there
        is one, else a new epoch -- shuffled with the generator's RandomState, which therefore advances exactly once per
        epoch (generators.py:89-97; the bit-exact batch tests pin the stream)."""
        parked = self.state
        if parked is not None:
            return parked
        return 0, (self.random.permutation(self.pairs) if self.shuffle else self.pairs)

    def _device_table(self, pairs: np.ndarray) -> torch.Tensor:
        """One upload of the epoch's chunk table (12 B per chunk); batches then only take slices of it."""
        if self._table is None or self._table[0] is not pairs:
            tab = np.ascontiguousarray(pairs[:, [0, 1, 3]].astype(np.int32))
            self._table = (pairs, torch.from_numpy(tab).to(self._res.device))
        return self._table[1]

    def _batch_rows(self, b_i: int, n_pairs: int):
        """Row range [lo, hi) of batch b_i in the epoch's chunk table for this generator (its shard of the batch when
        sharded), or None when a sharded generator has to drop the batch (short last batch: see dp.shardable)."""
        lo = b_i * self.batch_size
        hi = min(lo + self.batch_size, n_pairs)
        if self.shard is None:
            return lo, hi
        rank, world = self.shard
        if not shardable(hi - lo, world):
            return None
        s_lo, s_hi = shard_bounds(hi - lo, rank, world)
        return lo + s_lo, lo + s_hi

    def next_epoch(self):
        """Yields (cam, batch_3d, batch_2d) device tensors batch by batch; an endless generator keeps going over new
        epochs and parks its position after every batch (generators.py:99-166 semantics, run.py:330-343 usage)."""
        while True:
            first, order = self.next_pairs()
            table = self._device_table(order)
            if self.endless and first == 0 and self.shard is not None and all(
                    self._batch_rows(b_i, len(order)) is None for b_i in range(self.num_batches)):
                # (an endless generator would otherwise spin here forever without ever yielding; a one-epoch generator
                # just yields its empty epoch, as a tiny sharded validation set always did)
                raise ValueError("sharded ChunkedGenerator: no batch of this epoch can be split over %d ranks "
                                 "(%d chunks, batch size %d: every rank needs at least 2 samples per batch)"
                                 % (self.shard[1], len(order), self.batch_size))
            for b_i in range(first, self.num_batches):
                rows = self._batch_rows(b_i, len(order))
                if self.endless:
                    self.state = (b_i + 1, order)
                if rows is None:
                    continue
                yield self._res.gather(table[rows[0]:rows[1]], rows[1] - rows[0], self.chunk_length, self.pad,
                                       self.causal_shift, self._perm2, self._perm3)
            self.state = None
            if not self.endless:
                return


class UnchunkedGenerator:
    """Non-batched data generator used for testing (reference generators.py:168-239): one sequence per batch,
    plus its mirrored copy when augmentation (test-time augmentation) is enabled."""

    def __init__(self, cameras, poses_3d, poses_2d, pad=0, causal_shift=0, augment=False, kps_left=None,
                 kps_right=None, joints_left=None, joints_right=None, device=None):
        assert poses_3d is None or len(poses_3d) == len(poses_2d)
        assert cameras is None or len(cameras) == len(poses_2d)

        self.augment = augment
        self.kps_left = kps_left
        self.kps_right = kps_right
        self.joints_left = joints_left
        self.joints_right = joints_right

        self.pad = pad
        self.causal_shift = causal_shift
        self.cameras = [] if cameras is None else cameras
        self.poses_3d = [] if poses_3d is None else poses_3d
        self.poses_2d = poses_2d

        self._res = _Resident(cameras, poses_3d, poses_2d, device)
        dev = self._res.device
        self._perm2 = _perm_dev(self._res.j2, kps_left, kps_right, dev)
        self._perm3 = _perm_dev(self._res.j3, joints_left, joints_right, dev) if self._res.p3 is not None else None
        n = len(poses_2d)
        tab = np.zeros((n, 2, 3), dtype=np.int32)              # per sequence: (seq, 0, flip=0), (seq, 0, flip=1)
        tab[:, :, 0] = np.arange(n)[:, None]
        tab[:, 1, 2] = 1
        self._table = torch.from_numpy(tab).to(dev)

    def num_frames(self):
        count = 0
        for p in self.poses_2d:
            count += p.shape[0]
        return count

    def augment_enabled(self):
        return self.augment

    def set_augment(self, augment):
        self.augment = augment

    def next_epoch(self):
        for s in range(len(self.poses_2d)):
            n = 2 if self.augment else 1
            yield self._res.gather(self._table[s, :n], n, self._res.lens[s], self.pad, self.causal_shift,
                                   self._perm2 if self.augment else None, self._perm3 if self.augment else None)


    # ---- many sequences per call (not in the reference) ------------------------------------
    def length_groups(self, max_frames: int = 32768):
        """Sequence indices grouped by similar length so that (sequences x longest length) stays <= max_frames per group
        (a sequence longer than that forms its own group)."""
        order = sorted(range(len(self.poses_2d)), key=lambda i: self._res.lens[i])
        groups, cur = [], []
        for i in order:
            if cur and (len(cur) + 1) * self._res.lens[i] > max_frames:
                groups.append(cur)
                cur = []
            cur.append(i)
        if cur:
            groups.append(cur)
        return groups

    def next_epoch_batched(self, max_frames: int = 32768):
        """Like next_epoch(), but several sequences per batch: yields (seq_ids, lengths, cam, batch_3d, batch_2d) where
        batch_2d is [n, T_max + 2*pad, J, F] with rows ordered [all sequences of the group..., then their mirrored copies
        when augmentation is on].  Shorter sequences are extended by edge replication (exactly the padding rule of
        generators.py:216-239 continued to T_max): the model is convolutional along time, so the first len(s) output
        frames of row s are what a call on that sequence alone produces."""
        aug = self.augment
        for grp in self.length_groups(max_frames):
            k = len(grp)
            n = 2 * k if aug else k
            tab = np.zeros((n, 3), dtype=np.int32)
            tab[:k, 0] = grp
            if aug:
                tab[k:, 0] = grp
                tab[k:, 2] = 1
            lens = [self._res.lens[i] for i in grp]
            cam, b3, b2 = self._res.gather(torch.from_numpy(tab).to(self._res.device), n, max(lens), self.pad,
                                           self.causal_shift, self._perm2 if aug else None, self._perm3 if aug else None)
            yield grp, lens, cam, b3, b2


def predict_sequences(model, generator: "UnchunkedGenerator", max_frames: int = 32768):
    """Predictions of ``model`` (eval mode) for every sequence of an UnchunkedGenerator, many sequences per forward call:
    the evaluation loop of run.py:652-680 (forward, un-flip + average of the test-time-augmentation pair) with one
    launch set per length group instead of one per video.  Returns a list of [T_i, J_out, 3] tensors in sequence order."""
    out = [None] * len(generator.poses_2d)
    traj = model.num_joints_out == 1                       # run.py:676: the trajectory model has no joints to swap
    with torch.no_grad():
        for grp, lens, _, _, b2 in generator.next_epoch_batched(max_frames):
            pred = model(b2)                                # [n, T_max (- causal trim), J_out, 3]
            k = len(grp)
            if generator.augment_enabled():
                t = pred.shape[1]
                pair = pred.reshape(2, k * t, pred.shape[2], pred.shape[3])     # rows [originals | mirrored]
                pred = tta_average(pair, None if traj else generator.joints_left,
                                   None if traj else generator.joints_right).reshape(k, t, pred.shape[2], pred.shape[3])
            for j, s_id in enumerate(grp):
                out[s_id] = pred[j, :lens[j]].clone()
    return out


def tta_average(predicted: torch.Tensor, joints_left=None, joints_right=None) -> torch.Tensor:
    """run.py:677-680: ``predicted`` [2,T,J,3] holds the prediction for a sequence and for its mirrored copy; undo
    the mirroring of copy 1 (negate x, swap left/right joints unless both lists are None, as for the trajectory
    model) and average -> [1,T,J,3].  One kernel instead of five indexing / mean kernels."""
    assert predicted.dim() == 4 and predicted.shape[0] == 2
    if not predicted.is_cuda:
        raise _lib.Vp3dError("tta_average runs on the GPU only")
    p = predicted.to(torch.float32).contiguous()
    _, t, j, d = p.shape
    perm = _perm_dev(j, joints_left, joints_right, p.device)
    out = torch.empty((1, t, j, d), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        check(_lib.lib().vp3d_tta_fold(ops._stream(), t, j, d, p.data_ptr(),
                                       perm.data_ptr() if perm is not None else None, out.data_ptr()), "vp3d_tta_fold")
    return out
