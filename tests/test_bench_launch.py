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
