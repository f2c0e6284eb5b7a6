"""bench.py's launcher contract: `python bench.py --gpus N` must work bare (no external torch.distributed.run) and print
ONE JSON line from rank 0 -- on CPU through --dry-run (rendezvous, barriers, max-over-ranks timing, no GPU work), on the
GPU box with the real step on 2 ranks that share GPU 0 over gloo (VP3D_DIST_BACKEND / VP3D_BENCH_DEVICE test hooks)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus2_launches_itself_dry_run():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["metric"].startswith("frames/sec") and out["scaling"] == "weak"
    # the launcher's environment: RCCL's channel budget (dp.RCCL_ENV_DEFAULTS: measured with tools/cu_hog_ab.py) and one slice
    # of the host's cores per rank (disjoint when there are at least two cores)
    assert out["launcher"]["rccl_env"] == {"NCCL_MAX_NCHANNELS": "8", "NCCL_MIN_NCHANNELS": "4"}
    c0, c1 = out["launcher"]["cores_per_rank"]
    if len(os.sched_getaffinity(0)) >= 2:
        assert c0 and c1 and not set(c0) & set(c1)


def test_bench_single_process_dry_run():
    out = _run(["--steps", "2", "--warmup", "0", "--dry-run"])
    assert out["n_gpus"] == 1 and out["launcher"]["rccl_env"] is None


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_gpus2_self_launch_real_step_two_ranks_on_one_gpu():
    """The N > 1 path end to end without a launcher: two ranks (gloo, both on GPU 0) run the real cfg3 step with the
    bucketed gradient exchange launched from inside backward; rank 0's line carries n_gpus = 2 and the whole-job rate."""
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-eval", "--no-f32", "--no-cpu-baseline",
                "--no-rocm-ref"], {"VP3D_DIST_BACKEND": "gloo", "VP3D_BENCH_DEVICE": "0"})
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2048 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["config"]["grad_allreduce_bytes"] > 6e7
    assert out["roofline"]["frac"] > 0
    assert out["value_path"] in ("eager", "graph_replay") and out["launcher"]["rccl_env"]["NCCL_MAX_NCHANNELS"] == "8"
    if out["value_path"] == "graph_replay":
        assert out["eager"]["ms_per_step"] >= out["ms_per_step"]


def test_final_line_is_short_and_carries_the_contract(tmp_path):
    """BENCH_r04: the driver could not parse a 20 KB line.  The full-size record of a real run (profiles/r04_bench_line.json, the
    biggest line this bench ever printed) goes through the SAME final_line() the GPU run prints: the last stdout line must stay
    under 4096 bytes (bench.py's own bound is 2048), parse, and carry `roofline` and `cpu_baseline` with the contract's keys."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["VP3D_DRY_RUN_DETAIL"] = os.path.join(ROOT, "profiles", "r04_bench_line.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--dry-run"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 4096 and len(r.stdout) < 4096, (len(last), len(r.stdout))
    out = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert out["roofline"]["frac"] == pytest.approx(out["roofline"]["achieved"] / out["roofline"]["peak"], rel=1e-3)
    assert "model" not in out["config"] and "workload" in out["config"]


def test_final_line_unit():
    """final_line() on a hand-made record: bound respected even when every optional section is present and verbose."""
    sys.path.insert(0, ROOT)
    import bench
    big = {"metric": "m", "value": 1.23456789e5, "unit": "frames/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 4.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16x3 (split-fp16 operands, fp32 accumulate)",
           "math": "f16x3", "data": "synthetic", "config": {"workload": "w" * 150, "global_batch": 8192, "parallelism": "dp8"},
           "roofline": {"bound": "mfma", "achieved": 370.0, "peak": 833.3, "unit": "TFLOP/s", "frac": 0.444, "traffic": 6.9e8,
                        "algorithmic_bytes": 3.9e8, "kernel": "k_nt_s16<Cfg<2,4,4,2,2,32,0,1,3>> (" + "x" * 500 + ")",
                        "per_launch": [{"a": 1}] * 500, "note": "n" * 5000},
           "cpu_baseline": {"value": 338.2, "unit": "frames/s", "cores": 16, "kind": "port", "sample": "s" * 900},
           "mpjpe_vs_ref": {"value": 2.4e-6, "tolerance": 1e-3, "unit": "u" * 300},
           "f32_mfma": {"value": 1e5, "ms_per_step": 10.2, "step_frac_of_fp32_mfma_peak": 0.65, "roofline": {"x": "y" * 3000}},
           "cfg2_eval_fwd": {"ms": 13.8, "tflops": 386.0, "frac": 0.463, "math": "f16x3", "workload": "z" * 300},
           "launcher": {"rccl_env": {"NCCL_MAX_NCHANNELS": "8"}, "cores_of_rank0": list(range(64))},
           "timing": {"step_ms": [4.0] * 200}}
    s = bench.final_line(big)
    assert len(s) <= bench.FINAL_LINE_MAX
    o = json.loads(s)
    assert o["roofline"]["kernel"] == "k_nt_s16<Cfg<2,4,4,2,2,32,0,1,3>>" and "per_launch" not in o["roofline"]
    assert len(o["cpu_baseline"]["sample"]) <= 160 and o["value"] == pytest.approx(1.23456789e5, rel=1e-5)


def test_box_state_reads_a_sysfs_tree_and_never_raises(tmp_path):
    """bench.box_state: power / clock / temperature from the amdgpu hwmon files (here a fake sysfs tree with two cards: no PCI
    match on a CPU host -> the card with the highest power draw), missing files -> None, no tree -> an error entry, never an
    exception (the diagnostics must not cost the run its result)."""
    sys.path.insert(0, ROOT)
    import bench
    for card, power in (("card0", 241e6), ("card8", 1000.2e6)):
        hw = tmp_path / "class" / "drm" / card / "device" / "hwmon" / "hwmon4"
        hw.mkdir(parents=True)
        (hw / "power1_input").write_text("%d\n" % power)
        (hw / "power1_cap").write_text("1400000000\n")
        (hw / "freq1_input").write_text("2226900000\n")
    steps = []
    out = bench.box_state(lambda: steps.append(1), n_steps=3, samples=2, sysfs=str(tmp_path))
    assert len(steps) == 3 and out["power_w"] == 1000.2 and out["power_cap_w"] == 1400.0 and out["sclk_mhz"] == 2226.9
    assert out["temp_junction_c"] is None and out["matched_by"] == "highest power draw" and "card8" in out["hwmon"]
    assert "error" in bench.box_state(lambda: None, n_steps=1, samples=1, sysfs=str(tmp_path / "nothing"))
    line = json.loads(bench.final_line({"metric": "m", "box": out, "config": {}}))
    assert line["box"] == {"power_w": 1000.2, "power_cap_w": 1400.0, "sclk_mhz": 2226.9, "temp_junction_c": None}
