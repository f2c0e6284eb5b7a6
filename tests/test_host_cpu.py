"""Host-side (no GPU) checks: C-ABI exports, state_dict / init parity with the reference, plan arithmetic,
loud failure without a GPU."""
import os
import re

import pytest
import torch

import videopose3d_amd as V
from videopose3d_amd import _lib, plan as P
from tests.util import golden_names, load_golden, load_kats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(meta, dropout=None):
    kw = dict(causal=meta["causal"], dropout=meta["dropout"] if dropout is None else dropout, channels=meta["channels"])
    if meta["kind"] == "dilated":
        return V.TemporalModel(meta["j_in"], meta["feat"], meta["j_out"], meta["filter_widths"], dense=meta["dense"], **kw)
    return V.TemporalModelOptimized1f(meta["j_in"], meta["feat"], meta["j_out"], meta["filter_widths"], **kw)


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vp3d.h")).read()
    declared = set(re.findall(r"\b(vp3d_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vp3d_stream_t"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    h = _lib.lib()                                   # loads libvp3d.so; raises if any symbol is missing
    for name in declared:
        assert hasattr(h, name)
    assert h.vp3d_version() == 110
    assert h.vp3d_stat_slabs(129) == 3


def test_abi_rejects_bad_arguments_without_gpu():
    h = _lib.lib()
    rc = h.vp3d_colsum(None, 0, 4, None, 4, None)    # validation happens before any launch
    assert rc == -1 and b"colsum" in h.vp3d_last_error()
    with pytest.raises(_lib.Vp3dError):
        _lib.check(rc, "vp3d_colsum")


@pytest.mark.parametrize("name", golden_names())
def test_state_dict_is_reference_compatible(name):
    g = load_golden(name)
    m = _make(g["meta"])
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["sd0"].keys())              # same names, same order as the reference emits
    for k, v in sd.items():
        assert tuple(v.shape) == g["sd0"][k].shape, k
        assert str(v.dtype).replace("torch.", "") == str(g["sd0"][k].dtype), k
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["sd0"].items()}, strict=True)
    assert m.receptive_field() == g["meta"]["rf"]
    assert m.total_causal_shift() == g["meta"]["total_causal_shift"]
    m.set_bn_momentum(0.03)
    assert m.expand_bn.momentum == 0.03 and all(bn.momentum == 0.03 for bn in m.layers_bn)


@pytest.mark.parametrize("kat", load_kats(), ids=lambda k: "%s-%d" % (k["kind"], len(k["filter_widths"])))
def test_seeded_init_matches_reference(kat):
    torch.manual_seed(0)
    cls = V.TemporalModel if kat["kind"] == "dilated" else V.TemporalModelOptimized1f
    m = cls(17, 2, 17, kat["filter_widths"], causal=kat["causal"], channels=1024)
    assert sum(p.numel() for p in m.parameters()) == kat["n_params"]
    assert abs(float(m.expand_conv.weight.double().sum()) - kat["expand_w_sum"]) < 1e-6
    assert abs(float(m.shrink.weight.double().sum()) - kat["shrink_w_sum"]) < 1e-6
    x = torch.randn(kat["batch"], kat["rf"], 17, 2)               # same RNG stream position as the reference run
    assert abs(float(x.double().sum()) - kat["x_sum"]) < 1e-6


def test_two_classes_share_state_dict():
    a = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=64)
    b = V.TemporalModel(17, 2, 17, [3, 3, 3], channels=64)
    b.load_state_dict(a.state_dict())                             # run.py:426
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_no_cpu_fallback():
    m = V.TemporalModel(17, 2, 17, [3, 3, 3], channels=32)
    with pytest.raises(V.Vp3dError):
        m(torch.zeros(2, 27, 17, 2))
    with pytest.raises(AssertionError):
        m(torch.zeros(2, 27, 16, 2))
    with pytest.raises(AssertionError):
        V.TemporalModel(17, 2, 17, [3, 4, 3])


def test_plan_shapes_cfg2_cfg3():
    p2 = P.make_plan("dilated", 34, 1024, 51, [3, 3, 3, 3, 3])
    assert p2.lengths(243) == [241, 235, 217, 163, 1] and p2.receptive_field() == 243
    p3 = P.make_plan("strided", 34, 1024, 51, [3, 3, 3, 3, 3])
    assert p3.lengths(243) == [81, 27, 9, 3, 1]
    assert [r.start for r in p3.res] == [1, 1, 1, 1] and [r.step for r in p3.res] == [3, 3, 3, 3]
    pc = P.make_plan("strided", 34, 1024, 51, [3, 3, 3], causal=True)
    assert [r.start for r in pc.res] == [2, 2] and pc.total_causal_shift() == 13
    pd = P.make_plan("dilated", 34, 1024, 51, [3, 3, 3], causal=True)
    assert pd.total_causal_shift() == 91                           # the reference's double counting, kept
    with pytest.raises(ValueError):
        p2.lengths(100)
    # algorithmic FLOPs of BASELINE.md section 2 follow from the plan
    def flops(plan, t):
        ls = [t] + plan.lengths(t)
        f = 0
        for i, c in enumerate(plan.convs):
            t_out = ls[1 + (i + 1) // 2] if i else ls[1]
            f += 2 * t_out * c.c_out * c.taps * c.c_in
        return f + 2 * ls[-1] * plan.shrink.c_out * plan.shrink.c_in
    assert flops(p2, 243) == 5217830912
    assert flops(p3, 243) == 352569344


def test_split_heuristics_bounds():
    h = _lib.lib()
    for m in (2, 31, 1024, 27648, 82944):
        s = h.vp3d_wgrad_splits(m, 1024, 3072)
        assert 1 <= s <= max(1, (m + 31) // 32)
    assert h.vp3d_wgrad_splits(27648, 1024, 3072) * 192 % 256 == 0      # whole 256-CU rounds on the big layer
    # big forward GEMMs: only the tiles that do not fill the last round of 2 x 256 workgroups are K-sliced
    assert 0 < h.vp3d_rows_gemm_ws_floats(240640, 1024, 3072) <= 640 * 128 * 128
    assert h.vp3d_rows_gemm_ws_floats(65536, 1024, 3072) == 0           # 4096 tiles = 8 whole rounds: nothing to slice
    assert h.vp3d_rows_gemm_splits(65536, 1024, 3072) == 1
    assert h.vp3d_rows_gemm_ws_floats(1024, 1024, 3072) == h.vp3d_rows_gemm_splits(1024, 1024, 3072) * 64 * 128 * 128
    assert h.vp3d_rows_gemm_splits(1024, 1024, 3072) > 1                # the T_out = 1 tail is


def test_forward_flops_match_the_survey_figures():
    """plan.forward_flops (the size measure that decides between the two GEMM engines) against SURVEY.md 8(d):
    5,217,830,912 FLOP per sample for the cfg2 eval forward, 352,569,344 for the cfg3 training forward."""
    from videopose3d_amd.plan import make_plan
    dil = make_plan("dilated", 34, 1024, 51, [3, 3, 3, 3, 3])
    stri = make_plan("strided", 34, 1024, 51, [3, 3, 3, 3, 3])
    assert dil.forward_flops(1, 243) == 5217830912
    assert stri.forward_flops(1, 243) == 352569344
    assert stri.forward_flops(1024, 243) == 1024 * 352569344


def test_engine_selection_by_arithmetic_and_size():
    """engine.use_s16: the split-fp16 engine needs math == "f16x3", a supported configuration and a call big enough to be
    compute-bound; everything else runs on the fp32-MFMA kernels."""
    import videopose3d_amd as V
    from videopose3d_amd import engine
    big = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], channels=1024)
    assert big.math == V.default_math()
    big.math = "f16x3"
    assert engine.use_s16(big, 243, True, batch=1024)
    assert engine.use_s16(big, 243, True, batch=128)                      # 45 GFLOP: already pays (tools/small_modes.py)
    assert not engine.use_s16(big, 243, True, batch=64)                   # launch-latency regime
    assert engine.use_s16(big, 243, True, need_dx=True, batch=1024)       # input gradients (expand layer on the fp32 kernels)
    assert engine.use_s16(big, 244, True, batch=1024)                     # windows that do not tile
    big.math = "f32"
    assert not engine.use_s16(big, 243, True, batch=1024)
    ev = V.TemporalModel(17, 2, 17, [3, 3, 3, 3, 3], channels=1024)
    ev.math = "f16x3"
    assert engine.use_s16(ev, 243, False, batch=1024) and not engine.use_s16(ev, 300, False, batch=2)
    assert engine.use_s16(ev, 243, True, batch=1024)                      # training of the dilated class
    dense = V.TemporalModel(17, 2, 17, [3, 3, 3], channels=1024, dense=True)
    dense.math = "f16x3"
    assert engine.use_s16(dense, 243, False, batch=1024) and not engine.use_s16(dense, 243, True, batch=1024)   # 19-tap convs
    odd = V.TemporalModel(17, 2, 17, [3, 3, 3], channels=48)
    odd.math = "f16x3"
    assert not engine.use_s16(odd, 100, False, batch=4096)                # channels % 64 != 0
    with pytest.raises(V.Vp3dError):
        V.set_default_math("bf16")
        V.default_math()
    V.set_default_math(None)


def test_s16_planner_returns_launchable_plans_and_keeps_the_tuned_picks():
    """vp3d_nt_s16_plan is a host function: every plan must be launchable (known tiling, 1 <= splits <= K/32, a split only
    when allowed and then from the measured set), and the picks the tools/s16_tune.py sweep found best on the large
    shapes of the cfg3 step are pinned (balanced tile order -> 256x256 + split-K for wgrad, hybrid for
    the first block's dgrad)."""
    import random
    from videopose3d_amd import ops_s16 as S
    rnd = random.Random(5)
    for _ in range(300):
        m = rnd.choice([1, 63, 64, 1000, 1024, 3072, 9216, 27648, 82944, 240640])
        n = rnd.choice([32, 51, 64, 128, 1024, 3072])
        k = 32 * rnd.choice([1, 2, 4, 32, 96, 288, 864, 2592])
        for raw in (False, True):
            cfg, splits = S.plan(m, n, k, raw)
            assert cfg in (20, 22, 30)
            assert 1 <= splits <= k // 32 and splits in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)
            assert cfg != 30 or splits == 1
    assert S.plan(27648, 3072, 1024, False) == (22, 1)       # first block's dgrad: plain 256x256 (the hybrid re-measured slower in round 2)
    assert S.plan(27648, 1024, 3072, False) == (22, 1)
    assert S.plan(1024, 3072, 27648, True) == (22, 16)       # wgrad: 256x256 tiles x 16 K-slices = 3 full rounds
    assert S.plan(1024, 1024, 27648, True) == (22, 16)
    assert S.plan(1024, 128, 82944, True) == (20, 32)        # expand wgrad
    assert S.plan(1024, 1024, 1024, False)[0] == 20          # the T_out = 1 tail stays on 128x128 tiles
    assert S.plan(240640, 1024, 3072, False) == (22, 1)      # cfg2 eval


def test_rows_form_wgrad_split_choice_is_launchable():
    """ops_s16._wgrad_rows_splits (host side of vp3d_wgrad_rows_s16): a candidate slice count, at least 6 K-tiles of 32 rows
    per slice when sliced, and the step's big shapes fill one (nearly) whole round of the 256 CUs."""
    from videopose3d_amd import ops_s16 as S
    for m in (1, 31, 65, 1000, 1024, 3072, 9216, 27648, 82944):
        for c_out, n_cols in ((256, 256), (256, 768), (1024, 1024), (1024, 3072), (512, 1536)):
            s = S._wgrad_rows_splits(m, c_out, n_cols)
            nkt = (m + 31) // 32
            assert s in S.WGRAD_SPLIT_CANDIDATES and 1 <= s <= max(1, nkt)
            assert s == 1 or nkt // s >= 6
    assert S._wgrad_rows_splits(27648, 1024, 3072) == 5         # 48 tiles x 5 slices = 240 workgroups: one round, 63 MB of
    assert S._wgrad_rows_splits(9216, 1024, 3072) == 5          # partials (16 slices: three rounds, 201 MB; tools/wgrad_splits.py)
    assert S._wgrad_rows_splits(27648, 1024, 1024) == 16        # 16 tiles x 16 slices = 1 round
    assert S.wgrad_rows_supported(1024, 1024) and S.wgrad_rows_supported(1024, 128) and not S.wgrad_rows_supported(1024, 96)
    assert not S.wgrad_rows_supported(128, 128) and not S.expand_rows_form(1024, 96)


def test_common_model_shadow_import_resolves_as_integration_md_says(tmp_path):
    """INTEGRATION.md section 2 (option A): with <repo>/videopose3d_amd ahead of a VideoPose3D checkout on sys.path,
    run.py:21's ``from common.model import *`` must bind OUR classes while every other ``common.*`` module still comes
    from the checkout.  The checkout is a stand-in tree here (plus the real /root/reference when it exists)."""
    import subprocess
    import sys
    fake = tmp_path / "VideoPose3D"
    (fake / "common").mkdir(parents=True)
    (fake / "common" / "generators.py").write_text("WHO = 'reference generators'\n")
    (fake / "common" / "loss.py").write_text("WHO = 'reference loss'\n")
    (fake / "common" / "model.py").write_text("raise ImportError('the reference model module must be shadowed')\n")
    code = ("from common.model import *\n"
            "import common.model, common.generators, common.loss, videopose3d_amd.model as M\n"
            "assert TemporalModel is M.TemporalModel and TemporalModelOptimized1f is M.TemporalModelOptimized1f\n"
            "assert TemporalModelBase is M.TemporalModelBase\n"
            "print(common.model.__file__); print(common.generators.__file__); print(common.loss.__file__)\n")
    trees = [str(fake)] + (["/root/reference"] if os.path.exists("/root/reference/common/generators.py") else [])
    for tree in trees:
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "videopose3d_amd"), ROOT, tree]))
        r = subprocess.run([sys.executable, "-c", code], cwd=tree, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        mod, gen, loss = r.stdout.strip().splitlines()[-3:]
        assert os.path.samefile(mod, os.path.join(ROOT, "videopose3d_amd", "common", "model.py"))
        assert os.path.samefile(gen, os.path.join(tree, "common", "generators.py"))
        assert os.path.samefile(loss, os.path.join(tree, "common", "loss.py"))


def test_sharded_generator_drops_a_last_batch_that_cannot_feed_every_rank():
    """A short last batch (generators.py:57,104) that would leave a rank with 0 or 1 samples is dropped on ALL ranks of a
    sharded generator (a rank without a batch would skip backward and with it the gradient collectives; one sample
    cannot pass training-mode BatchNorm at T_out = 1); anything larger is split into contiguous balanced slices."""
    from videopose3d_amd import dp
    from videopose3d_amd.generators import ChunkedGenerator
    assert dp.shardable(16, 8) and not dp.shardable(15, 8) and dp.shardable(2, 1) and not dp.shardable(1, 1)
    g = ChunkedGenerator.__new__(ChunkedGenerator)        # row arithmetic only: no device, no dataset
    g.batch_size = 1024
    n_pairs = 3 * 1024 + 11                               # last batch: 11 chunks
    for world in (1, 2, 4, 8):
        spans = []
        for rank in range(world):
            g.shard = (rank, world)
            assert g._batch_rows(1, n_pairs) == tuple(1024 + v for v in dp.shard_bounds(1024, rank, world))
            spans.append(g._batch_rows(3, n_pairs))
        if 11 >= 2 * world:
            assert spans[0][0] == 3072 and spans[-1][1] == n_pairs
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and min(b - a for a, b in spans) >= 2
        else:
            assert spans == [None] * world                # dropped everywhere, never on some ranks only
    g.shard = None
    assert g._batch_rows(3, n_pairs) == (3072, n_pairs)   # the unsharded generator keeps the reference's short batch


def test_direct_gradient_sink_refuses_a_second_backward_pass():
    """FlatGradSync(direct_module=...) overwrites gradients: two backward passes between zero_grad() calls must raise
    instead of silently dropping the first pass (also with world == 1, where no collective would notice)."""
    from videopose3d_amd import dp
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=32)
    sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)
    n_groups = len(m.backward_param_groups())
    for _ in range(2):
        sync.zero_grad()
        for k in range(n_groups):
            sync.group_done(k)
    with pytest.raises(V.Vp3dError):
        sync.group_done(0)
    sync.zero_grad()
    sync.group_done(0)


def test_dropout_seed_is_reproducible_across_processes():
    """The dropout stream of a model is a function of torch's seed, the rank and the model's construction ORDINAL -- not of
    id(self): two runs of the same script (or two debug reruns of a DP rank) draw the same masks."""
    import subprocess
    import sys
    code = ("import torch, videopose3d_amd as V\n"
            "torch.manual_seed(7)\n"
            "a = V.TemporalModelOptimized1f(17, 2, 17, [3, 3], channels=32)\n"
            "b = V.TemporalModelOptimized1f(17, 2, 1, [3, 3], channels=32)\n"
            "print(a._next_dropout_state()[0], b._next_dropout_state()[0])\n")
    outs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT).stdout.split() for _ in range(2)]
    assert len(outs[0]) == 2 and outs[0] == outs[1] and outs[0][0] != outs[0][1]


def test_fastdiv_is_exact(tmp_path):
    """vp3d_s16.h's FastDiv (row -> (sample, frame) split of the streaming kernels): the host-side magic numbers against
    plain division over edge cases and random (n, d) with 0 <= n < 2^31, 1 <= d < 2^31 (host build of the same header)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "fd.hip"
    src.write_text(r'''
#include "vp3d_s16.h"
#include <cstdio>
#include <cstdint>
static uint64_t rng = 88172645463325252ull;
static uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
static int check(uint32_t n, uint32_t d) {
  const vp3d::FastDiv f = vp3d::make_fastdiv(d);
  const uint32_t q = (uint32_t)(((uint64_t)n * f.mul) >> f.shift);      // = fastdiv() on the device
  if (q != n / d) { printf("n=%u d=%u got %u want %u\n", n, d, q, n / d); return 1; }
  return 0;
}
int main() {
  const uint32_t top = 0x7fffffffu;
  int bad = 0;
  const uint32_t ds[] = {1, 2, 3, 5, 7, 9, 27, 81, 243, 1024, 1025, 65535, 65536, 65537, 0x3fffffffu, 0x40000000u, 0x40000001u, top - 1, top};
  for (uint32_t d : ds) {
    const uint32_t k = top / d;
    const uint32_t ns[] = {0, 1, d - 1, d, d + 1 <= top ? d + 1 : top, k * d, k * d ? k * d - 1 : 0, top, top - 1};
    for (uint32_t n : ns) bad += check(n & top, d);
  }
  for (int i = 0; i < 2000000; ++i) {
    const uint32_t d = (uint32_t)(next() >> (33 + (next() % 31)));      // all magnitudes
    bad += check((uint32_t)(next() & top), d ? d : 1);
  }
  printf("bad=%d\n", bad);
  return bad != 0;
}
''')
    exe = tmp_path / "fd"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(root, "include"),
                    "-I", os.path.join(root, "videopose3d_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "bad=0" in out.stdout, out.stdout[-500:]


def test_engine_selection_rules(monkeypatch):
    """Which arithmetic serves a call is host logic (engine.use_s16 / engine_s16.supported): pinned here so that a GPU
    run cannot silently move a configuration to the other engine."""
    from videopose3d_amd import engine, engine_s16
    from videopose3d_amd.plan import ConvSpec
    mk = lambda cls=V.TemporalModelOptimized1f, fw=(3, 3, 3), c=128, j=17, **kw: cls(j, 2, 17, list(fw), channels=c, **kw)
    m = mk()
    assert engine_s16.supported(m, 27, True) and engine_s16.supported(m, 27, False) and engine_s16.supported(m, 27, True, True)
    assert not engine_s16.supported(mk(c=96), 27, True) and not engine_s16.supported(mk(c=96), 27, False)    # channels % 64
    assert engine_s16.supported(mk(V.TemporalModel, c=64), 40, True)                        # dilated class, any window
    assert not engine_s16.supported(mk(fw=(5, 3)), 15, True) and not engine_s16.supported(mk(fw=(5, 3)), 15, False)   # 5 x 34 = 170 columns
    assert engine_s16.supported(mk(fw=(5, 3), j=10), 15, True) and engine_s16.supported(mk(fw=(5, 3), j=10), 15, False)  # 100 columns
    assert engine_s16.supported(mk(j=5), 27, True) and not engine_s16.supported(mk(j=5), 27, False)          # 30 columns: train only
    dense = mk(V.TemporalModel, fw=(3, 5, 3), c=64, dense=True)                             # dense kernels: 3, 7, 31 taps
    assert not engine_s16.supported(dense, 50, True) and engine_s16.supported(dense, 50, False)
    # the module attribute decides first, then the size threshold
    m.math = "f32"
    assert not engine.use_s16(m, 27, True, batch=1024)
    m.math = "f16x3"
    assert engine.use_s16(m, 27, True, batch=1 << 20) and not engine.use_s16(m, 27, True, batch=1)
    # expand operand width and the 32-bit-offset guards
    assert [engine_s16.expand_kpad(ConvSpec(k, 64, 3, 1, 3)) for k in (34, 30, 10, 42)] == [128, 128, 64, 128]
    from videopose3d_amd._switches import SW
    monkeypatch.setitem(SW, "wgrad_rows", True)
    assert engine_s16.wgrad_from_rows(1024, 1024, 1024 * 243) and not engine_s16.wgrad_from_rows(1024, 1024, 8192 * 243)
    assert not engine_s16.wgrad_from_rows(128, 128, 100)
    monkeypatch.setitem(SW, "wgrad_rows", False)
    assert not engine_s16.wgrad_from_rows(1024, 1024, 1024)


def test_deepcopy_gets_its_own_dropout_stream_and_device_counters():
    """copy.deepcopy(model) is a model of its own: a new construction ordinal (hence another mask stream) and no shared
    device-side step counter / momentum scalar; its parameters are copies."""
    import copy
    torch.manual_seed(3)
    a = V.TemporalModelOptimized1f(17, 2, 17, [3, 3], channels=32)
    a._drop_counter = torch.zeros(1, dtype=torch.int64)
    a._bn_momentum_dev = torch.full((1,), 0.1)
    sa = a._next_dropout_state()[0]
    b = copy.deepcopy(a)
    assert b._ordinal != a._ordinal and b._drop_counter is None and b._bn_momentum_dev is None
    assert b._next_dropout_state()[0] != sa
    assert b.expand_conv.weight.data_ptr() != a.expand_conv.weight.data_ptr()
    assert torch.equal(b.expand_conv.weight, a.expand_conv.weight)
    assert b.receptive_field() == a.receptive_field() and list(b.state_dict()) == list(a.state_dict())


def test_bn_momentum_bookkeeping_for_captured_steps():
    """graph.py reads the BatchNorm momentum from device memory when all layers share one value (what set_bn_momentum leaves
    behind, model.py:36-39); per-layer momenta fall back to launch arguments (guarded re-capture)."""
    from videopose3d_amd import graph as G
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=32)
    assert m._uniform_bn_momentum() == 0.1 and G._momentum_guard(m) is None and m._momentum_dev_ptr() is None
    m._bn_momentum_dev = torch.full((1,), 0.1)           # (a CPU tensor stands in for the device scalar)
    m._bn_momentum_host = 0.1
    m.set_bn_momentum(0.03)
    assert abs(float(m._bn_momentum_dev) - 0.03) < 1e-9 and m._momentum_dev_ptr() == m._bn_momentum_dev.data_ptr()
    m.layers_bn[1].momentum = 0.5                         # someone sets one layer by hand: no single scalar any more
    assert m._uniform_bn_momentum() is None and m._momentum_dev_ptr() is None
    assert G._momentum_guard(m) == (0.03, 0.03, 0.5, 0.03, 0.03)
    m.set_bn_momentum(0.02)
    m.expand_bn.momentum = 0.07                           # ... or all of them, bypassing set_bn_momentum: the pointer call refreshes
    for bn in m.layers_bn:
        bn.momentum = 0.07
    m._momentum_dev_ptr()
    assert abs(float(m._bn_momentum_dev) - 0.07) < 1e-9


def test_sharded_endless_generator_raises_instead_of_spinning():
    """An epoch in which NO batch can feed every rank (fewer than 2 samples per rank) must raise: an endless generator
    (the semi-supervised one, run.py:330-343) would otherwise loop forever without yielding."""
    from videopose3d_amd.generators import ChunkedGenerator
    g = ChunkedGenerator.__new__(ChunkedGenerator)
    g.batch_size, g.num_batches, g.endless, g.shard, g.state = 4, 2, True, (0, 4), None
    g.next_pairs = lambda: (0, list(range(7)))
    g._device_table = lambda order: None
    with pytest.raises(ValueError, match="no batch of this epoch"):
        next(g.next_epoch())


def test_s16_engine_rejects_gather_form_training_beyond_the_kernel_row_limit():
    """vp3d_gather_t_s16 addresses at most 65535 row tiles: a dilated-class training call beyond that runs on the fp32
    engine instead of failing in the middle of backward (engine_s16.supported(batch=...))."""
    from videopose3d_amd import engine_s16
    m = V.TemporalModel(17, 2, 17, [3, 3, 3], channels=64)
    assert engine_s16.supported(m, 243, True, batch=64)
    assert not engine_s16.supported(m, 243, True, batch=65535 * 64 // 200)
    s = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=64)      # windows tile: no gathered operand at all
    assert engine_s16.supported(s, 27, True, batch=65535 * 64)


def test_ctypes_structs_have_the_layout_of_the_header(tmp_path):
    """The ctypes mirrors of the ABI's argument structs against include/vp3d.h compiled by the C compiler: size and the offset
    of EVERY field (a silently shifted pointer would be read as garbage by the kernels)."""
    import shutil
    import subprocess
    import ctypes as C
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = [("vp3d_dropout", _lib.Dropout), ("vp3d_rowmap", _lib.RowMap), ("vp3d_s16", _lib.S16Opts), ("vp3d_s16_red", _lib.S16Red)]
    lines = ['#include "vp3d.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for cname, cls in pairs:
        lines.append('  printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([cc, "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for (cname, cls), line in zip(pairs, out):
        got = [int(v) for v in line.split()[1:]]
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert got == want, (cname, got, want)


def test_fused_bn_backward_sums_selection_rules():
    """ops_s16.red_supported (which dgrad launches may carry the BatchNorm-backward column sums, vp3d_s16_red): one K slice on
    the 128x128 / 256x256 tilings, whole column tiles, 256-channel strips; engine_s16 adds the row threshold."""
    from videopose3d_amd import engine_s16, ops_s16 as S
    assert S.red_supported(27648, 1024, 1024, 1024)            # block 1's 1x1 dgrad: writes the gradient of a 27,648-row activation
    assert S.red_supported(9216, 3072, 1024, 1024)             # block 2's strided dgrad: three taps fold into one strip
    assert not S.red_supported(1024, 1024, 1024, 1024)         # the planner slices K: the finishing pass owns the epilogue
    assert not S.red_supported(27648, 1024, 1024, 128)         # strips are 256 channels wide
    assert not S.red_supported(27648, 1000, 1024, 1000)
    assert engine_s16.FUSE_BN_RED_DEFAULT == "auto" and engine_s16.FUSE_BN_RED_MIN_ROWS == 16384


def test_build_is_gated_by_a_content_hash_not_by_mtimes(tmp_path, monkeypatch):
    """__graft_entry__.build() compiles unless libvp3d.so was built from EXACTLY the sources in the tree: the stamp holds a sha256
    of every source / header + the flags (a fresh clone has no stamp -> compiles; touching a file without changing it does not)."""
    import importlib
    g = importlib.import_module("__graft_entry__")
    assert os.path.exists(g.LIB) and not g._stale(), "the shipped library must match the tree's sources"
    h0 = g._source_hash()
    src = os.path.join(g.CSRC, g.SOURCES[0])
    os.utime(src, None)                                  # newer mtime, same content
    assert g._source_hash() == h0 and not g._stale()
    monkeypatch.setattr(g, "STAMP", str(tmp_path / "missing.srchash"))
    assert g._stale()                                    # no stamp: a library of unknown origin is rebuilt
    stamp = tmp_path / "other.srchash"
    stamp.write_text("0" * 64 + "\n")
    monkeypatch.setattr(g, "STAMP", str(stamp))
    assert g._stale()                                    # built from other sources
    monkeypatch.setattr(g, "FLAGS", g.FLAGS + ["-DSOMETHING"])
    assert g._source_hash() != h0                        # the flags are part of the identity


def test_gemm_unit_order_covers_every_slice_and_tile_once():
    """The (K slice, tile) -> workgroup mapping of k_tn_s16 / k_nt_s16 (csrc/vp3d_gemm_s16.hip: tile_from_linear + the slice-major
    XCD shares), restated in Python: for every geometry the grid of 8 x per_xcd workgroups forms each (slice, tile) unit exactly
    once, the surplus workgroups of the last share return, an XCD's share is contiguous in slice-major order, and a share of
    whole slices contains every tile of those slices (what lets the tiles of a slice share operand panels in one L2)."""
    def tile_from_linear(L, m_tiles, n_tiles):
        gn = min(n_tiles, 8)
        full = (n_tiles // gn) * gn
        in_full = m_tiles * full
        if L < in_full:
            blk, r = divmod(L, m_tiles * gn)
            tm = r // gn
            return tm, blk * gn + (r - tm * gn)
        gl, r = n_tiles - full, L - in_full
        tm = r // gl
        return tm, full + (r - tm * gl)

    for m_tiles, n_tiles, splits in [(4, 4, 16), (4, 12, 5), (4, 12, 16), (1, 1, 1), (1, 3, 4), (124, 12, 1), (58, 4, 1), (20, 4, 3),
                                     (8, 8, 4), (3, 5, 7), (4, 1, 5), (7, 9, 2), (4, 12, 1)]:
        tiles = m_tiles * n_tiles
        units = tiles * splits
        per_xcd = (units + 7) // 8
        seen = {}
        for b in range(8 * per_xcd):
            xcd, q = b & 7, b >> 3
            u = xcd * per_xcd + q
            if u >= units:
                continue
            split, L = divmod(u, tiles)
            tm, tn = tile_from_linear(L, m_tiles, n_tiles)
            assert 0 <= tm < m_tiles and 0 <= tn < n_tiles and 0 <= split < splits
            assert (split, tm, tn) not in seen
            seen[(split, tm, tn)] = xcd
        assert len(seen) == units
        if units % 8 == 0 and per_xcd % tiles == 0:        # whole slices per XCD: all tiles of a slice sit on one XCD
            for split in range(splits):
                assert len({x for (s_, _, _), x in seen.items() if s_ == split}) == 1


def test_no_kernel_spills_inside_its_mfma_loop():
    """tools/isa_stats.py over the shipped library (no GPU needed): whatever a kernel spills, the scratch instructions lie outside
    the span between its first and last v_mfma -- the claim DESIGN.md 4.8 / 4.9 makes for the register-tight GEMM instances -- and the
    total stays small (a regression here means a K loop started paying scratch traffic per iteration)."""
    import subprocess
    import sys
    tools = "/opt/rocm/lib/llvm/bin"
    if not all(os.path.exists(os.path.join(tools, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump", "llvm-readelf")):
        pytest.skip("LLVM binutils of the ROCm image not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_stats.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    cols = lines[0].split()
    i_loop, i_ins, i_mfma = cols.index("scr_loop"), cols.index("scr_ins"), cols.index("mfma")
    rows = [ln.split(None, len(cols) - 1) for ln in lines[1:] if ln and ln[0].isdigit()]
    assert len(rows) > 40, "the kernel table looks truncated"
    gemm = [rw for rw in rows if int(rw[i_mfma]) > 0]
    assert len(gemm) >= 15
    for rw in rows:
        assert int(rw[i_loop]) == 0, "scratch instructions inside the MFMA span of %s" % rw[-1][:120]
    assert sum(int(rw[i_ins]) for rw in rows) <= 64


def test_range_guard_host_logic_without_gpu():
    """The guard never touches a CPU model, parameter loads / .to() ask for a fresh measurement, deep copies build their own state,
    and dp's launcher helpers behave (RCCL channel budget as defaults, core slices disjoint)."""
    import copy
    from videopose3d_amd import dp, engine, range_guard
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], channels=128)
    m.math = "f16x3"
    e0 = m.__dict__.get("_range_epoch", 0)
    range_guard.tick(m, True, torch.zeros(4, 27, 34))     # CPU tensor: nothing happens, no state
    assert m.__dict__.get("_range_state") is None and not range_guard.tripped(m)
    assert range_guard.status(m) == dict(tripped=False, last=None, checks=0, sync_checks=0, gram_off=False, gram_log2_kappa=None)
    assert not range_guard.gram_disabled(m) and range_guard.gram_flag(m) is None
    m.load_state_dict(m.state_dict())
    assert m.__dict__["_range_epoch"] == e0 + 1
    m.float()                                              # nn.Module._apply
    assert m.__dict__["_range_epoch"] == e0 + 2
    m.__dict__["_range_state"] = object()                  # (stands in for device buffers + an event)
    c = copy.deepcopy(m)
    assert "_range_state" not in c.__dict__
    del m.__dict__["_range_state"]
    assert engine.use_s16(m, 27, True)                     # an untripped model is not held back
    assert dp.RCCL_ENV_DEFAULTS["NCCL_MAX_NCHANNELS"] == "8"
    cores = sorted(os.sched_getaffinity(0))
    try:
        if len(cores) >= 2:
            a = dp.pin_rank_to_cores(0, 2)
            os.sched_setaffinity(0, cores)
            b = dp.pin_rank_to_cores(1, 2)
            assert a and b and not set(a) & set(b) and set(a) | set(b) <= set(cores)
        os.sched_setaffinity(0, cores)
        assert dp.pin_rank_to_cores(0, 1) is None
    finally:
        os.sched_setaffinity(0, cores)
        torch.set_num_threads(max(1, min(len(cores), 8)))


def test_mixed_tilings_are_refused_for_operands_of_2_gib_and_more():
    """ops_s16.plan(mix=True): the 224- / 160-row tilings address both operands through 32-bit buffer descriptors and have no
    flat-address twin, so the planner must not hand them out when an operand reaches 2 GiB (round-4 advisor finding: the launch
    then failed hard where configurations 20 / 22 fall back to 0 / 4) -- and the statistics buffers are sized from the same
    answer (32- vs 64-row slabs), so the refusal happens at planning time."""
    from videopose3d_amd import ops_s16 as S
    from videopose3d_amd._switches import SW
    assert SW["tile_mix"] == "1"
    cfg, splits = S.plan(27648, 1024, 3072, mix=True)
    assert cfg == 28 and splits == 1 and S.stat_slab_rows(cfg, splits) == 32
    big_m = 8 * 27648                                        # B = 8192 on one GPU: 221,184 x 3072 x 4 B = 2.7 GB
    cfg, splits = S.plan(big_m, 1024, 3072, mix=True)
    assert cfg not in (28, 29) and S.stat_slab_rows(cfg, splits) == 64
    assert S.plan(big_m, 1024, 3072, mix=True) == S.plan(big_m, 1024, 3072, mix=False)
