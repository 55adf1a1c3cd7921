"""bench.py's multi-rank bookkeeping on CPU (gloo, world 2): `python bench.py --gpus 2` must START two ranks (not silently measure one), report
`n_gpus` 2, count the units of ALL ranks and divide by the SLOWEST rank's time.  `--stub` swaps the hot path for a sleep (rank r sleeps
(r + 1) x 20 ms per step) and labels the line as a stub; launcher, rank_env, timed_steps and whole_job are the code the GPU run uses."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def _line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_flag_launches_that_many_ranks_and_takes_the_slowest():
    steps = 4
    d = _line(_run(["--gpus", "2", "--stub", "--steps", str(steps), "--warmup", "0"]))
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["scaling"] == "weak" and d["data"] == "stub"
    # rank 1 sleeps 40 ms per step, rank 0 20 ms: max over ranks, not rank 0's own clock and not the mean
    assert 40.0 <= d["ms_per_step"] < 80.0, d["ms_per_step"]
    # whole-job aggregate: 2 ranks x 16 images x steps / that time
    assert abs(d["value"] - 2 * 16 * steps / (d["ms_per_step"] * steps / 1000.0)) / d["value"] < 1e-9


def test_single_rank_default_and_explicit():
    for args in (["--stub", "--steps", "2"], ["--gpus", "1", "--stub", "--steps", "2", "--batch", "4"]):
        d = _line(_run(args))
        assert d["n_gpus"] == 1 and 20.0 <= d["ms_per_step"] < 40.0
    assert d["config"]["batch_per_gpu"] == 4 and abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2 / 1000.0)) / d["value"] < 1e-9


def test_forced_process_group_at_world_one():
    """ECAD_FORCE_DIST=1: a single rank still creates the process group and runs barrier + MAX all-reduce through it (gloo here, RCCL in
    tests/test_gpu_bench_contract.py) -- the switch that lets a 1-GPU box exercise the collective path."""
    d = _line(_run(["--gpus", "1", "--stub", "--steps", "2"], env_extra={"ECAD_FORCE_DIST": "1"}))
    assert d["n_gpus"] == 1 and d["config"]["process_group"] == "gloo, world 1" and 20.0 <= d["ms_per_step"] < 40.0
    assert _line(_run(["--stub", "--steps", "1"]))["config"]["process_group"] is None


def test_gpus_flag_contradicting_torchrun_environment_is_refused():
    p = _run(["--gpus", "4", "--stub"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_under_torchrun_gpus_defaults_to_world_size():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29731",
                        str(ROOT / "bench.py"), "--gpus", "2", "--stub", "--steps", "2"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    d = _line(p)
    assert d["n_gpus"] == 2 and d["ms_per_step"] >= 40.0


def test_traffic_is_quoted_only_from_a_profile_with_this_runs_stamp(tmp_path, monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench

    stamp = bench.kernel_source_stamp()
    assert len(stamp) == 16 and stamp == bench.kernel_source_stamp()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    monkeypatch.setattr(bench, "kernel_source_stamp", lambda: stamp)
    (prof / "r01_x_pmc_traffic.json").write_text(json.dumps({"gemm_ff1_gelu": {"hbm_bytes_per_launch": 1.0}}))  # round-1 style: no stamp
    (prof / "r02_a_pmc_traffic.json").write_text(json.dumps({"gemm_ff1_gelu": {"hbm_bytes_per_launch": 2.0}, "_stamp": {"workload": "alpha256_b16", "dtype": "f16", "kernel_sources": "0" * 16}}))
    d, f, _ = bench.stamped_traffic("alpha256_b16", "f16")
    assert d == {} and f is None  # stale kernels / unstamped: null, never a silently old number
    (prof / "r02_b_pmc_traffic.json").write_text(json.dumps({"gemm_ff1_gelu": {"hbm_bytes_per_launch": 3.0}, "_stamp": {"workload": "alpha256_b16", "dtype": "f16", "kernel_sources": stamp}}))
    d, f, _ = bench.stamped_traffic("alpha256_b16", "f16")
    assert f == "r02_b_pmc_traffic.json" and d["gemm_ff1_gelu"]["hbm_bytes_per_launch"] == 3.0
    assert bench.stamped_traffic("sigma1024_b4", "f16")[1] is None and bench.stamped_traffic("alpha256_b16", "fp8")[1] is None
