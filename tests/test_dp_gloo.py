"""Data-parallel path on CPU: world_size-2 gloo process group (the N>1 path of bench.py / videopose3d_amd.dp)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import videopose3d_amd as V
from videopose3d_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # different initial weights per rank on purpose
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=32)
    sync = dp.FlatGradSync(m.parameters())
    n_par = sum(p.numel() for p in m.parameters())
    assert sync.world == world and n_par <= sync.numel < n_par + 64 * len(list(m.parameters()))
    assert all(off % 64 == 0 for off in sync.offsets)
    sync.broadcast_parameters(m.buffers())
    flat_w = torch.cat([p.detach().flatten() for p in m.parameters()])
    gathered = [torch.empty_like(flat_w) for _ in range(world)]
    dist.all_gather(gathered, flat_w)
    assert all(torch.equal(gathered[0], g) for g in gathered)            # replicas start identical

    # gradients: every p.grad is a view of the flat buffer; autograd-style in-place accumulation lands in it
    sync.zero_grad()
    for i, p in enumerate(m.parameters()):
        assert p.grad.data_ptr() >= sync.flat.data_ptr()
        p.grad.add_(torch.full_like(p, float(rank + 1) * (i + 1)))
    sync.sync()
    for i, p in enumerate(m.parameters()):
        expect = (i + 1) * sum(range(1, world + 1)) / world
        assert torch.allclose(p.grad, torch.full_like(p, expect)), (i, float(p.grad.flatten()[0]), expect)

    # short last batch: per-rank mean gradients re-weighted by their sample counts == global mean
    sync.zero_grad()
    counts = [5, 3]
    for p in m.parameters():
        p.grad.add_(float(rank + 1))
    sync.sync(local_count=counts[rank], global_count=sum(counts))
    expect = sum((r_ + 1) * c for r_, c in enumerate(counts)) / sum(counts)
    assert torch.allclose(next(iter(m.parameters())).grad, torch.full_like(next(iter(m.parameters())), expect))

    # zero_grad re-attaches views dropped by optimizer.zero_grad(set_to_none=True)
    for p in m.parameters():
        p.grad = None
    sync.zero_grad()
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in m.parameters())

    # overlapped exchange: flat layout in backward-completion order, buckets all-reduced as the engine reports groups
    big = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=64)
    bs = dp.FlatGradSync(big.parameters(), direct_module=big, bucket_bytes=40000)
    groups = big.backward_param_groups()
    assert [id(p) for p in bs.params] == [id(p) for g in groups for p in g]
    assert len(bs.buckets) >= 2 and bs.buckets[-1][2] == bs.numel and bs.buckets[0][1] == 0
    assert all(a[2] == b[1] for a, b in zip(bs.buckets, bs.buckets[1:]))           # contiguous cover
    assert big.__dict__["_vp3d_grad_sink"] is bs
    for weighted in (False, True):
        counts = [5, 3]
        if weighted:
            bs.zero_grad(local_count=counts[rank], global_count=sum(counts))
        else:
            bs.zero_grad()
        for k, g in enumerate(groups):                  # what engine.backward_train does: write, then report
            for p in g:
                bs.view_for(p).add_(float(rank + 1) * (k + 1))
            launched_before = bs._launched
            bs.group_done(k)
            assert bs._launched >= launched_before
        assert bs._launched == len(bs.buckets) and len(bs._handles) == len(bs.buckets)
        bs.sync()
        assert not bs._handles
        for k, g in enumerate(groups):
            if weighted:
                expect = (k + 1) * sum((r_ + 1) * c for r_, c in enumerate(counts)) / sum(counts)
            else:
                expect = (k + 1) * sum(range(1, world + 1)) / world
            for p in g:
                assert torch.allclose(p.grad, torch.full_like(p, expect)), (weighted, k)
    # a backward that never reports groups (plain autograd accumulation) is exchanged by sync(), bucket by bucket
    bs.zero_grad()
    for p in big.parameters():
        p.grad.add_(float(rank + 1))
    bs.sync()
    assert torch.allclose(bs.flat[:10], torch.full((10,), sum(range(1, world + 1)) / world))
    # ranks that disagree on what backward reported (rank 1 skipped its backward altogether, rank 0 reported two
    # groups) still issue the same collectives in the same order: no hang, correct sums
    bs.zero_grad()
    for p in big.parameters():
        p.grad.add_(float(rank + 1))
    if rank == 0:
        bs.group_done(0)
        bs.group_done(1)
    bs.sync()
    assert torch.allclose(bs.flat[:10], torch.full((10,), sum(range(1, world + 1)) / world))
    assert torch.allclose(bs.flat[-70:-64], torch.full((6,), sum(range(1, world + 1)) / world))

    # batch sharding of an identically-seeded generator permutation (generators.py:89-104)
    perm = np.random.RandomState(1234).permutation(1000)
    batch = perm[:257]                                                   # odd-sized global batch
    mine, none_kept = dp.shard_batch([batch, None], rank, world)
    assert none_kept is None
    np.save(os.path.join(out_dir, "shard%d.npy" % rank), mine)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(str(tmp_path / ("shard%d.npy" % r))) for r in range(world)]
    full = np.random.RandomState(1234).permutation(1000)[:257]
    assert np.array_equal(np.concatenate(parts), full)                   # disjoint, ordered, complete
    assert abs(len(parts[0]) - len(parts[1])) <= 1


def test_shard_bounds_cover_everything():
    for n in (1, 7, 8, 1024, 1031):
        for world in (1, 2, 4, 8):
            spans = [dp.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
