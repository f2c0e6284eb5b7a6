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


def _guard_worker(rank, world, port, out_dir):
    """The dynamic-range guard's decisions under data parallelism (videopose3d_amd/range_guard.py, round 6): two ranks whose
    MEASUREMENTS differ -- the device kernels are replaced by a per-rank script, everything else (MAX-exchange over the gradient
    exchange's group, training-call cadence, fixed-distance consumption, sticky parameter trips / re-evaluated input trips,
    engine choice) is the shipped code on CPU tensors."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dp.init_from_env("gloo")
    from videopose3d_amd import engine, range_guard as G
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    cpu = torch.device("cpu")
    torch.manual_seed(7)
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=64).train()
    m.math = "f16x3"
    sync = dp.FlatGradSync(m.parameters(), direct_module=m)
    assert G.dp_sink(m) is sync
    script = {}            # measurement index of THIS rank -> (act, weight, kappa, input, head-gradient) spreads
    n_meas = [0]

    def fake_measure(mod, st, m_rows, x3=None):
        v = script.get(n_meas[0], (3, 4, 0, 1, 1))
        n_meas[0] += 1
        st.out.copy_(torch.tensor(v, dtype=torch.int32))

    G._measure = fake_measure
    x = torch.zeros(4, 27, 34)

    def run(n_calls, batches, eval_extra=()):
        trace = []
        for i in range(n_calls):
            if rank == 0 and i in eval_extra:          # evaluation calls that only ONE rank makes: no collective, no cadence shift
                G._tick(m, False, 2, 27, cpu, x)
            G._tick(m, True, batches[rank], 27, cpu, x)
            trace.append((G.tripped(m), engine.use_s16(m, 27, True, batch=batches[rank])))
        return trace

    # (A) rank 1 alone measures a hot input column in the second periodic measurement, and a normal one in the third
    # measurement 0 = the synchronous one at the first call, 1.. = the periodic ones launched at training calls 1, 17, 33, ...
    if rank == 1:
        script[2] = (3, 4, 0, 20, 1)
    trace = run(60, (4, 4), eval_extra=(3, 18, 19, 40))
    st = G.status(m)
    assert st["exchanges"] == n_meas[0] == 5, (st, n_meas)              # call 0 (sync) + training calls 1, 17, 33, 49
    gathered = [None] * world
    dist.all_gather_object(gathered, trace)
    assert gathered[0] == gathered[1], "the ranks switched engines on different steps"
    # call index i = training call i (the synchronous call 0 resets the counter): launched at 17, consumed at 17 + 8; re-evaluated
    # from the measurement launched at 33, consumed at 41
    expect = [(False, True)] * 25 + [(True, False)] * 16 + [(False, True)] * 19
    assert trace == expect, [i for i, (a, b) in enumerate(zip(trace, expect)) if a != b]
    assert st["io_last"] == (1, 1) and not st["trip_param"]

    # (B) the short last batch (ranks hold 5 and 3 samples: their BatchNorm-bound statistics differ through sqrt(M - 1)); rank 0
    # alone measures a parameter spread beyond the limit: both ranks leave the split-fp16 engine on the same step, for good
    G.invalidate(m)
    n_meas[0] = 0
    script.clear()
    if rank == 0:
        script[1] = (13, 4, 0, 1, 1)
    trace = run(30, (5, 3))
    dist.all_gather_object(gathered, trace)
    assert gathered[0] == gathered[1]
    assert trace == [(False, True)] * 9 + [(True, False)] * 21          # launched at training call 1, consumed at 9, sticky
    st = G.status(m)
    assert st["trip_param"] and st["checks"] >= 2
    n_before = n_meas[0]
    run(20, (5, 3))
    assert n_meas[0] == n_before                                         # a parameter trip stops the periodic measurements
    dist.barrier()
    dist.destroy_process_group()


def test_range_guard_switches_every_rank_on_the_same_step(tmp_path):
    world = 2
    mp.spawn(_guard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)


def test_shard_bounds_cover_everything():
    for n in (1, 7, 8, 1024, 1031):
        for world in (1, 2, 4, 8):
            spans = [dp.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
