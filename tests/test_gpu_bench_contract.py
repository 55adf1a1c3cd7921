"""-m gpu: bench.py prints ONE JSON line with the contract's keys (metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus `roofline`, `cpu_baseline`, the PNG-ready rate and the
secondary workloads, on a short run."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _bench(*args):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _bench("--steps", "2", "--warmup", "1", "--secondary", "alpha256_batch1_latency")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f16" and d["data"] == "synthetic"
    assert "ours_fast" in d["config"]["workload"] and "batch=16" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2 / 1000.0)) / d["value"] < 1e-6  # value == images / timed seconds
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["achieved"] > 100.0  # the dominant GEMM really ran on the matrix cores
    # PMC traffic is quoted only from a profile measured on exactly these kernel sources, with its file named; otherwise null
    assert (r["traffic"] is None) == (r["traffic_source"] is None) and len(r["kernel_sources_stamp"]) == 16
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "images/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # the rate comparable to the reference's own latency numbers (VAE decode + PIL inside) is reported by default, beside and below `value`
    assert 0 < d["png_ready_images_per_s"] < d["value"]
    s = d["secondary"]["alpha256_batch1_latency"]
    assert "error" not in s and s["value"] > 0 and "batch=1 " in s["workload"]


def test_gpus_1_explicit_and_secondary_workload_lines_are_labelled():
    d = _bench("--gpus", "1", "--batch", "2", "--steps", "2", "--warmup", "1", "--no-roofline", "--no-cpu-baseline", "--no-decode-tail")
    assert d["n_gpus"] == 1 and "[secondary: batch 2]" in d["metric"] and "cpu_baseline" not in d and "secondary" not in d
