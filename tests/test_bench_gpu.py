"""The measurement contract of bench.py, checked on the GPU box: one JSON line on stdout with the driver's keys, the roofline object
of the dominant kernel (+ whole-step fraction), and -- in the default form -- a cpu_baseline object.  Short run (2 timed steps)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"bench.py must print ONE line on stdout, got {len(lines)}"
    return json.loads(lines[0])


def test_bench_line_contract(cuda):
    d = _run("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--decode-steps", "2", "--sustain-seconds", "0")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["unit"] == "image-lines/s"
    assert d["config"]["workload"].startswith("c3") and d["config"]["global_batch"] == 256
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]            # whole-job throughput = lines / measured time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "step_frac", "ms_per_launch"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.05 < r["frac"] < 1.0 and 0.05 < r["step_frac"] < r["frac"] + 0.3
    assert "filter gradient" in r["kernel"] and "forward" in d["roofline_best"]["kernel"]
    assert d["decode_chars_per_s"] > 1e5 and d["families"]["_sum_ms"] > 0
    assert "cpu_baseline" not in d                                                            # --no-cpu-baseline


def test_bench_other_workloads_run(cuda):
    d = _run("--workload", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--decode-steps", "0", "--sustain-seconds", "0")
    assert d["dtype"] == "f32" and d["roofline"]["peak"] == 157.3 and d["config"]["img"] == "32x100"
    d = _run("--workload", "c4", "--steps", "24", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", "--decode-steps", "0", "--sustain-seconds", "0")
    assert d["config"]["per_gpu_batch"] == 64 and d["cluster_fallback"] is False and d["value"] > 1000
    d = _run("--dropout", "0.3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--decode-steps", "0", "--sustain-seconds", "0")
    assert d["dropout"] == 0.3 and d["cluster_fallback"] is False
