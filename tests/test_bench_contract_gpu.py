"""The driver's contract for bench.py (one JSON line on stdout, the named keys, a timed region
that does not depend on --steps): run as the driver runs it, on the headline leg only."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *args],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly ONE line on stdout"
    return json.loads(lines[0])


def test_bench_line_contract_and_steps_invariance():
    a = run_bench("--steps", "20", "--warmup", "5", "--rollout-only")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int),
                     ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool),
                     ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(a[key], typ), key
    assert a["vs_baseline"] is None and a["scaling"] == "weak" and a["n_gpus"] == 1
    assert a["steps"] == 20 and a["warmup"] == 5 and "workload" in a["config"]
    assert a["config"]["boards_per_launch"] == 4096 and a["config"]["launches_in_flight"] == 1
    r = a["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # algorithmic bytes of one launch / its duration, and the whole-job value, are consistent
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(a["value"] - 4096 * 1e3 / a["ms_per_step"]) < 1e-6 * a["value"]
    assert r["kernel_ms"] <= a["ms_per_step"] * 1.001
    assert a["timed_region_s"] >= 0.1
    # `value` does not move with --steps (3x between --steps 20 and --steps 2000)
    b = run_bench("--steps", "500", "--warmup", "5", "--rollout-only")
    assert abs(a["value"] - b["value"]) < 0.1 * a["value"], (a["value"], b["value"])
