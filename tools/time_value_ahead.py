"""Lab tool: bench.py's PV-MCTS leg (configs[2], full games) under different schedules.

    python tools/time_value_ahead.py "IAGO_VALUE_AHEAD=0" "IAGO_VALUE_AHEAD=1" "IAGO_VALUE_AHEAD=1 IAGO_ASYNC=1" ...

Each argument is a space-separated list of environment settings for one run; prints one line per run.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    sims = int(os.environ.get("SIMS", "100"))
    games = int(os.environ.get("GAMES", "1024"))
    turns = int(os.environ.get("TURNS", "-1"))
    import torch  # noqa: F401
    import bench
    out = bench.mcts_leg(games, sims, max(turns, 0), turns < 0, 1, 0, None)
    keep = {k: out.get(k) for k in ("leaf_evals_per_sec", "games_per_sec", "seconds", "policy_evals",
                                    "value_inline", "value_ahead", "async_steps", "turns_played", "persistent")}
    keep["leaf_evals_per_sec"] = round(keep["leaf_evals_per_sec"] / 1e6, 3)
    print(os.environ.get("SPEC", ""), json.dumps(keep), flush=True)


def main():
    # one child process per run (several knobs are read once per process); the parent never touches the GPU
    import subprocess
    runs = sys.argv[1:] or ["IAGO_VALUE_AHEAD=0", "IAGO_VALUE_AHEAD=1"]
    for spec in runs:
        env = dict(os.environ)
        for kv in spec.split():
            k, v = kv.split("=", 1)
            env[k] = v
        env["SPEC"] = spec
        env["IAGO_TIME_CHILD"] = "1"
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)


if __name__ == "__main__":
    if os.environ.get("IAGO_TIME_CHILD") == "1":
        one()
    else:
        main()
