"""hipGraph capture of the training step (forward + mpjpe loss + backward) of a videopose3d_amd model.

The step of run.py:401-420 is ~130 kernel launches; for the large benchmark configuration the GPU is the bottleneck,
but for the small ones (the semi-supervised setup of BASELINE configs[4]: arc 3,3,3, 64 + 64 samples; short last
batches; small ``--batch-size``) the step is bound by launch latency.  ``GraphedTrainStep`` records the launches once per
input shape into a ``torch.cuda.CUDAGraph`` (every libvp3d.so entry point only enqueues on the stream it is given and
never allocates or synchronises: include/vp3d.h) and replays them with ONE host call per step.

What makes a replay a new step although the launch arguments are frozen:
  * the batch is copied into the graph's static input buffers;
  * dropout masks: the kernels add a device-side step counter to the (frozen) Philox offset (``vp3d_dropout.offset_ptr``)
    and the graph itself bumps the counter;
  * BatchNorm running statistics / ``num_batches_tracked`` live in device memory and are updated in place;
  * the BatchNorm momentum (run.py:590-593 changes it every epoch) is read from device memory by the captured finalize
    launches (vp3d_bn_finalize_dm; ``model.set_bn_momentum`` / a changed ``bn.momentum`` refreshes that float before the
    replay) -- no re-capture per epoch.  Only per-layer DIFFERENT momenta fall back to launch arguments + re-capture.
A captured graph owns a private memory pool that holds every activation and gradient of the step (several GB at B = 1024):
there is ONE live entry per (shapes, device, arithmetic); when parameter addresses (optim.FlatAdam adoption, ``.to()``),
dropout probability or per-layer momenta change, the stale entry is dropped -- pool and all -- before the new capture.
Gradients land in the flat buffer of a ``dp.FlatGradSync`` (the parameters' ``.grad`` are views of it), so
``optimizer.step()`` -- torch Adam or ``optim.FlatAdam`` -- follows unchanged; with more than one rank the gradient
exchange runs after the replay (``sync.sync()``: one all-reduce of the flat buffer).

    step = GraphedTrainStep(model_pos_train)            # once
    for _, batch_3d, batch_2d in train_generator.next_epoch():
        ...
        loss_3d_pos = step(inputs_2d, inputs_3d)        # replaces zero_grad / forward / mpjpe / backward (run.py:410-418)
        optimizer.step()
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch

from . import dp, engine, loss as vloss, range_guard
from ._lib import Vp3dError


DEBUG_DOT_PATH = None      # tools/graph_branches.py: write hipGraphDebugDotPrint's .dot of the next capture here


class _Entry:
    __slots__ = ("graph", "x", "t", "loss", "keep", "guard", "grads", "segments", "fork_event", "gout")


def _momentum_guard(m):
    """None when the model's BatchNorm layers share one momentum (then read from device memory: any value replays), else the
    tuple of launch-argument momenta the capture froze."""
    if m._uniform_bn_momentum() is not None:
        return None
    return (m.expand_bn.momentum,) + tuple(bn.momentum for bn in m.layers_bn)


def _arm_device_momentum(m, dev):
    if m._uniform_bn_momentum() is not None and m._bn_momentum_dev is None:
        m._bn_momentum_host = m._uniform_bn_momentum()
        m._bn_momentum_dev = torch.full((1,), m._bn_momentum_host, dtype=torch.float32, device=dev)
    m._momentum_dev_ptr()                              # value in step with the modules BEFORE anything is captured


class _Segments:
    """Piecewise capture of a step whose backward hands work to the second stream: graph k on the main stream ends at a
    hand-over, graph k' on the second stream holds the handed-over work, the next main graph follows (engine.fork_to_side /
    fork_back call to_side / to_main).  ``replay`` launches them in capture order, each main -> side hand-over as an event
    between two graph launches, and joins at the end: the two streams overlap as they do in the eager step, which ONE graph
    with fork / join edges does not on this runtime (hipGraph replays its branches serially: DESIGN.md 4.8)."""

    def __init__(self, pool, main, side):
        # Two memory pools: the pieces of the main stream share one, the pieces of the second stream the other.  torch only
        # documents a shared pool as safe for graphs that replay in capture order and never concurrently; the side pieces replay
        # BESIDE the main ones, so a block that a main piece frees must never be handed to a side piece that is still running
        # (that safety was implicit in the engine's `keep` list; now it does not depend on it).
        self.pool, self.side_pool, self.main, self.side = pool, torch.cuda.graph_pool_handle(), main, side
        self.graphs = []
        self.joins = set()        # indices (in self.graphs) of main pieces that wait for the second stream before they launch
        self._cur = None

    def _begin(self, kind):
        ctx = torch.cuda.stream(self.main if kind == "main" else self.side)
        ctx.__enter__()
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self.pool if kind == "main" else self.side_pool, capture_error_mode="thread_local")
        self._cur = (kind, g, ctx)

    def _end(self):
        kind, g, ctx = self._cur
        self._cur = None
        try:
            g.capture_end()
        finally:
            ctx.__exit__(None, None, None)            # also when capture_end raises (abort): never leave the stream entered
        self.graphs.append((kind, g))

    def begin(self):
        self._begin("main")

    def end(self):
        self._end()

    def abort(self):
        if self._cur is not None:
            try:
                self._end()
            except Exception:  # noqa: BLE001
                pass

    def to_side(self):
        self._end()
        self._begin("side")

    def to_main(self):
        self._end()
        self._begin("main")

    def join(self):
        """Mid-step join (engine.join_side_now): the main piece that starts here waits for the second stream's pieces so far."""
        self._end()
        self.joins.add(len(self.graphs))
        self._begin("main")

    def replay(self, fork_event):
        main = torch.cuda.current_stream()
        for i, (kind, g) in enumerate(self.graphs):
            if kind == "main":
                if i in self.joins:
                    main.wait_stream(self.side)
                g.replay()
            else:
                fork_event.record(main)
                self.side.wait_event(fork_event)
                with torch.cuda.stream(self.side):
                    g.replay()
        main.wait_stream(self.side)


def _new_graph():
    g = torch.cuda.CUDAGraph()
    if DEBUG_DOT_PATH:
        g.enable_debug_mode()
    return g


def _dump_dot(g):
    if DEBUG_DOT_PATH:
        g.debug_dump(DEBUG_DOT_PATH)


class GraphedTrainStep:
    def __init__(self, model, sync: Optional[dp.FlatGradSync] = None, piecewise: Optional[bool] = None):
        # piecewise (default on, VP3D_GRAPH_PIECEWISE=0 / piecewise=False: one graph): capture the step as graph pieces per
        # stream so that the replay keeps the eager backward's two-stream overlap
        self.piecewise = (os.environ.get("VP3D_GRAPH_PIECEWISE", "1") != "0") if piecewise is None else bool(piecewise)
        self.model = model
        self.sync = sync if sync is not None else dp.FlatGradSync(model.parameters(), direct_module=model)
        if model.__dict__.get("_vp3d_grad_sink") is not self.sync:
            raise Vp3dError("GraphedTrainStep needs a FlatGradSync created with direct_module=model (the captured backward "
                            "writes the gradients straight into its flat buffer)")
        self._cache: Dict[Tuple, _Entry] = {}

    # one training step, eagerly (this is also what gets captured)
    def _step(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        m = self.model
        b, tt = x.shape[0], x.shape[1]
        if m._drop_counter is not None:
            m._drop_counter.add_(1)
        out3, saved = engine.forward_train(m, x.view(b, tt, -1), save=True)
        pred = out3.view(b, -1, m.num_joints_out, 3)
        lval, gout = vloss._mpjpe_call(pred, t, None, True)
        self._last_gout3 = gout.view_as(out3)         # (the captured step's head gradient: measured behind a replay by the guard)
        self.sync._seen.clear()                       # every (captured) step overwrites the flat gradients: its own zero_grad
        reduce = self.sync._reduce
        self.sync._reduce = False                     # no collectives inside the step: sync.sync() exchanges afterwards
        try:
            grads, _ = engine.backward_train(m, saved, gout.view_as(out3), False)
        finally:
            self.sync._reduce = reduce
        assert all(g is None for g in grads), "the gradient sink did not take every gradient"
        return lval

    def _key(self, x, t):
        """What selects an entry (one live graph per key) ..."""
        m = self.model
        return (tuple(x.shape), tuple(t.shape), x.device.index, m.math,
                engine.use_s16(m, x.shape[1], True, batch=x.shape[0]), range_guard.gram_disabled(m))

    def _guard(self):
        """... and what invalidates it: the captured launches hold raw device addresses and launch-argument scalars.
        Parameters re-pointed after a capture (optim.FlatAdam adopting them into its flat buffer, model.to(), a new
        FlatGradSync), another dropout probability or per-layer BatchNorm momenta must re-capture, not replay stale values.
        The address TUPLE itself is compared (a hash could collide and replay on stale pointers)."""
        m = self.model
        addrs = tuple(p.data_ptr() for p in m.parameters()) + tuple(b.data_ptr() for b in m.buffers()) + \
            (self.sync.flat.data_ptr(),)
        return (addrs, float(m.drop.p), _momentum_guard(m))

    def _capture(self, x, t) -> _Entry:
        m = self.model
        if not m.training:
            raise Vp3dError("GraphedTrainStep: call model.train() first")
        if m.__dict__.get("_vp3d_sync_bn") is not None:
            raise Vp3dError("GraphedTrainStep: synchronised BatchNorm puts collectives inside the step; not captured")
        if m._drop_counter is None:
            m._drop_counter = torch.zeros(1, dtype=torch.int64, device=x.device)
        _arm_device_momentum(m, x.device)
        e = _Entry()
        e.guard = self._guard()
        e.grads = None
        e.x = x.detach().to(torch.float32).contiguous().clone()
        e.t = t.detach().to(torch.float32).contiguous().clone()
        # one eager warm-up step (creates the lazily built helpers, loads the code objects, sizes the allocator) with the
        # model's state put back afterwards: capture itself executes nothing
        state = [b_.clone() for b_ in m.buffers()]
        counters = (m._drop_calls, m._stats_epoch, m._drop_counter.clone())
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._step(e.x, e.t)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b_, s_ in zip(m.buffers(), state):
                b_.copy_(s_)
            m._drop_counter.copy_(counters[2])
        m._drop_calls, m._stats_epoch = counters[0], counters[1]
        # The captured backward forks its weight-gradient GEMMs onto the engine's second stream.  Capture with a FRESH one:
        # a stream that RCCL collectives have run on (the overlapped bucket exchange of eager steps) made hipGraphLaunch
        # crash on ROCm 7.0 when it was pulled into a capture.
        key = x.device.index
        cached = engine._side_streams.get(key)
        engine._side_streams[key] = torch.cuda.Stream(device=x.device)
        e.segments, e.fork_event, e.graph = None, None, None
        piecewise = (self.piecewise and DEBUG_DOT_PATH is None and engine.use_s16(m, x.shape[1], True, batch=x.shape[0]) and
                     os.environ.get("VP3D_OVERLAP", "1") == "1")
        try:
            if piecewise:
                # graph pieces per stream (see _Segments): the replay keeps the two-stream overlap of the eager backward
                torch.cuda.synchronize()
                cap_main = torch.cuda.Stream(device=x.device)
                seg = _Segments(torch.cuda.graph_pool_handle(), cap_main, engine._side_streams[key])
                cap_main.wait_stream(torch.cuda.current_stream())
                engine._segmenter = seg
                try:
                    with torch.cuda.stream(cap_main):
                        seg.begin()
                        e.loss = self._step(e.x, e.t)
                        seg.end()
                except BaseException:
                    seg.abort()
                    raise
                finally:
                    engine._segmenter = None
                torch.cuda.current_stream().wait_stream(cap_main)
                e.segments, e.fork_event = seg, torch.cuda.Event()
            else:
                e.graph = _new_graph()
                with torch.cuda.graph(e.graph, capture_error_mode="thread_local"):   # (an RCCL watchdog thread may be alive)
                    e.loss = self._step(e.x, e.t)
        finally:
            e.keep = engine._side_streams.pop(key)
            if cached is not None:
                engine._side_streams[key] = cached
        if e.graph is not None:
            _dump_dot(e.graph)
        m._drop_calls, m._stats_epoch = counters[0], counters[1]   # capture executes nothing: host-side counters as before
        e.gout, self._last_gout3 = self._last_gout3, None         # (no second reference into the capture's memory pool)
        return e

    def __call__(self, inputs_2d: torch.Tensor, inputs_3d: torch.Tensor) -> torch.Tensor:
        """One training step on the batch; returns the mpjpe loss (a 0-dim device tensor that the next call overwrites)."""
        if not inputs_2d.is_cuda:
            raise Vp3dError("GraphedTrainStep runs on the GPU only")
        # the split-fp16 engine's dynamic-range guard ticks per replay (the captured forward is never re-executed by Python);
        # a trip changes use_s16 and with it the key: the step is re-captured on the exact-fp32 kernels
        range_guard.tick(self.model, True, inputs_2d)
        key = self._key(inputs_2d, inputs_3d)
        e = self._cache.get(key)
        if e is not None and e.guard != self._guard():
            del self._cache[key]                      # stale addresses / scalars: release the graph AND its memory pool first
            e = None
        if e is None:
            e = self._capture(inputs_2d, inputs_3d)
            self._cache[key] = e
        self.model._momentum_dev_ptr()                # set_bn_momentum / bn.momentum since the last step -> device float
        e.x.copy_(inputs_2d.reshape(e.x.shape))
        e.t.copy_(inputs_3d.reshape(e.t.shape))
        self.model._stats_epoch += 1                  # the running statistics are about to change (eval-fold cache key)
        if e.segments is not None:
            e.segments.replay(e.fork_event)
        else:
            e.graph.replay()
        range_guard.measure_head_gradient(self.model, e.gout)    # (a no-op unless this replay's tick launched a measurement)
        if self.sync._reduce:
            self.sync.sync()
        return e.loss


class GraphedStep:
    """hipGraph capture of an ARBITRARY training step built from videopose3d_amd models, their autograd nodes and torch ops --
    e.g. run.py's semi-supervised step (run.py:322-394: pose + trajectory model, four loss terms, project_to_2d), which is
    ~260 launches for 0.07 TFLOP and bound by launch latency when run eagerly.

        def step_fn(inputs_2d_cat, inputs_3d, cam):        # forward, losses AND backward; returns tensors (e.g. the loss)
            ...
            loss_total.backward()
            return loss_total
        step = GraphedStep(step_fn, models=(model_pos_train, model_traj_train))
        for ...:
            loss = step(inputs_2d_cat, inputs_3d, cam)      # copies the arguments into the static buffers, replays
            optimizer.step()                                # (zeroing the gradients in between is unnecessary: every replay overwrites them)

    ``fn`` runs under autograd inside the capture: the parameters' ``.grad`` are set to None before it, so the gradients that
    backward produces BECOME the ``.grad`` tensors (no accumulate kernel) and live in the graph's memory pool -- a replay
    overwrites them in place, which is why ``zero_grad(set_to_none=True)`` between replays must be left out.  Dropout masks
    advance through the models' device-side step counters, BatchNorm statistics are updated in place (as GraphedTrainStep).
    ``fn`` must be capturable: no host<->device copies or synchronisation inside (build index tensors beforehand, no
    ``.item()`` / Python-list indexing).  Re-captures when an argument's shape, a model's arithmetic / dropout probability or a parameter
    address changes (the stale graph and its memory pool are released first); ``zero_grad(set_to_none=True)`` between replays
    is harmless: every call re-attaches the captured gradient tensors."""

    def __init__(self, fn: Callable, models: Sequence, warmup: int = 2):
        self.fn, self.models, self.warmup = fn, list(models), warmup
        self._cache: Dict[Tuple, _Entry] = {}

    def _params(self):
        return [p for m in self.models for p in m.parameters()]

    def _run(self, static):
        for m in self.models:
            m._drop_counter.add_(1)
        return self.fn(*static)

    def _key(self, args):
        k = tuple((tuple(a.shape), a.dtype, a.device.index) if torch.is_tensor(a) else a for a in args)
        # (the guard's decisions are part of the key: a trip -- or the expand layer's statistics going back to the pass over the
        #  conv output -- re-captures the step on the other kernels, as in GraphedTrainStep)
        return k + tuple((m.math, m.training, range_guard.tripped(m), range_guard.gram_disabled(m)) for m in self.models)

    def _guard(self):
        g = ()
        for m in self.models:
            addrs = tuple(p.data_ptr() for p in m.parameters()) + tuple(b.data_ptr() for b in m.buffers())
            g += (addrs, float(m.drop.p), _momentum_guard(m))
        return g

    def _capture(self, args) -> _Entry:
        dev = next(self.models[0].parameters()).device
        for m in self.models:
            if m.__dict__.get("_vp3d_sync_bn") is not None or m.__dict__.get("_vp3d_grad_sink") is not None:
                raise Vp3dError("GraphedStep: models with synchronised BatchNorm or a gradient sink put collectives / foreign "
                                "buffers inside the step; use GraphedTrainStep for the data-parallel step")
            if m._drop_counter is None:
                m._drop_counter = torch.zeros(1, dtype=torch.int64, device=dev)
            _arm_device_momentum(m, dev)
        e = _Entry()
        e.guard = self._guard()
        e.x = [a.detach().clone() if torch.is_tensor(a) else a for a in args]
        e.t = None
        state = [[b_.clone() for b_ in m.buffers()] for m in self.models]
        counters = [(m._drop_calls, m._stats_epoch, m._drop_counter.clone()) for m in self.models]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):              # lazily built helpers, code objects, allocator sizes, autograd buffers
                for p in self._params():
                    p.grad = None
                self._run(e.x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for m, st, cn in zip(self.models, state, counters):
                for b_, s_ in zip(m.buffers(), st):
                    b_.copy_(s_)
                m._drop_counter.copy_(cn[2])
                m._drop_calls, m._stats_epoch = cn[0], cn[1]
        for p in self._params():
            p.grad = None
        key = dev.index
        cached = engine._side_streams.get(key)
        engine._side_streams[key] = torch.cuda.Stream(device=dev)      # (a fresh second stream: see GraphedTrainStep._capture)
        e.graph = _new_graph()
        try:
            with torch.cuda.graph(e.graph, capture_error_mode="thread_local"):
                e.loss = self._run(e.x)
        finally:
            e.keep = engine._side_streams.pop(key)
            if cached is not None:
                engine._side_streams[key] = cached
        _dump_dot(e.graph)
        for m, cn in zip(self.models, counters):
            m._drop_calls, m._stats_epoch = cn[0], cn[1]
        # the gradients the captured backward produced ARE the graph's output buffers: a replay overwrites them in place
        e.grads = [p.grad for p in self._params()]
        return e

    def __call__(self, *args):
        """One step on the arguments; returns what ``fn`` returned (static tensors that the next call overwrites)."""
        if any(torch.is_tensor(a) and not a.is_cuda for a in args) or not next(self.models[0].parameters()).is_cuda:
            raise Vp3dError("GraphedStep runs on the GPU only")
        for m in self.models:                         # the dynamic-range guard ticks per replay (shape of the capture's warm-up)
            range_guard.tick_replay(m)
        key = self._key(args)
        e = self._cache.get(key)
        if e is not None and e.guard != self._guard():
            del self._cache[key]                      # (graph + its memory pool go before the new capture allocates)
            e = None
        if e is None:
            e = self._capture(args)
            self._cache[key] = e
        for dst, src in zip(e.x, args):
            if torch.is_tensor(dst):
                dst.copy_(src)
        for m in self.models:
            m._stats_epoch += 1
            m._momentum_dev_ptr()
        e.graph.replay()
        # optimizer.zero_grad() defaults to set_to_none=True: without this the next optimizer.step() would silently skip
        # every parameter.  The replay has just rewritten the captured gradient buffers: hand them back.
        for p, g in zip(self._params(), e.grads):
            if p.grad is not g:
                p.grad = g
        return e.loss
