"""The driver's contract for bench.py (one JSON line on stdout, the named keys): the nested configs[1] leg as
--rollout-only prints it (a timed region that does not depend on --steps), and the default run's headline."""
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


def test_headline_is_the_pv_mcts_leg_under_the_clock():
    """The default run's headline (VERDICT r04 task 1): BASELINE configs[2] -- whole PV-MCTS self-play games on the
    persistent search -- with K >= 3 timed batches, a fresh position table per batch, the search kernel's MFMA roofline
    from live HIP events, and the hits of the position table split into same-game and cross-game."""
    a = run_bench("--steps", "3", "--warmup", "1", "--mcts-games", "256", "--no-rollout-leg", "--mcts400-turns", "0",
                  "--nthr1-turns", "0", "--train-iters", "0", "--no-cpu-baseline")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int),
                     ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str),
                     ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("leaf_evals_per_sec", float)):
        assert isinstance(a[key], typ), key
    assert a["steps"] == 3 and a["warmup"] == 1 and a["unit"] == "games/s" and a["vs_baseline"] is None
    assert "configs[2]" in a["config"]["workload"] and a["config"]["full_games"] is True
    assert a["config"]["games_per_step"] == 256 and a["config"]["sims_per_move"] == 100
    m = a["mcts"]
    assert m["steps"] == 3 and len(m["step_seconds"]) == 3 and m["full_games"] and m["persistent"] is not None
    assert abs(a["value"] - 256 * 3 / m["seconds"]) < 1e-9 * a["value"]
    assert abs(a["ms_per_step"] * 3 - m["seconds"] * 1e3) < 1e-6 * m["seconds"] * 1e3
    assert a["step_ms_min"] <= a["step_ms_median"] <= a["step_ms_max"] <= a["ms_per_step"] * 3
    assert sum(m["step_seconds"]) <= m["seconds"] * 1.001
    # three batches of NEW games: the leaf evaluations of three different sets of games
    assert m["leaf_evals"] == a["leaf_evals_per_step"] * 3 and m["leaf_evals"] > 3 * 256 * 100 * 50
    r = a["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "executed_frac", "launches"):
        assert key in r, key
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and r["launches"] == 3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < r["executed_frac"] < 1
    assert abs(r["achieved"] - r["algorithmic_flops_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["kernel_ms"] <= a["ms_per_step"] * 1.001          # the launch is inside its step
    assert r["kernel_ms"] * 3 >= 0.8 * m["seconds"] * 1e3      # ... and is most of it
    t = a["table_hits"]
    assert t["fresh_per_step"] is True and t["hits"] == t["hits_same_game"] + t["hits_cross_game"] > 0
    assert t["hits_same_game"] > 0 and t["hits_cross_game"] > 0
    assert m["batches_replayed_turn_by_turn"] == 0
    # round 6: what the same kernel sustains outside the latency regime of BASELINE's 1024-game batch -- one batch each
    # of 2048 and 4096 games per launch (beyond 32 game workgroups the search is split by role), and the spread of
    # `value` over this round's boxes
    sat = a["mcts_saturated"]
    assert set(sat) >= {"2048", "4096"} and sat["2048"]["games_per_launch"] == 2048
    for k, cus in (("2048", (32, 0)), ("4096", (64, 0))):
        assert sat[k]["games_per_sec"] > 0 and 0 < sat[k]["frac"] < sat[k]["executed_frac"] < 1
        assert sat[k]["role_split_game_cus"] in cus and sat[k]["batches_replayed_turn_by_turn"] == 0
    assert m["persistent"]["role_split_game_cus"] == 0          # (256 games: the single launch)
    assert a["value_spread"]["min"] <= a["value_spread"]["max"] and a["value_spread"]["boxes"] >= 1
