"""Lab tool: what does one more dependent launch cost inside the captured playout graph?
k tiny elementwise kernels are appended to every playout (after the backup) and bench.py's PV-MCTS leg
is timed: the slope in microseconds per playout per extra launch is the price of a launch boundary on the
playouts' critical path.   python tools/exp_gap_cost.py 0 1 2 4"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(k):
    import torch  # noqa: F401
    import bench
    from iago_amd import engine
    orig = engine.BatchedMCTS._playout_lookahead

    def patched(self, *a, **kw):
        r = orig(self, *a, **kw)
        for _ in range(k):
            self.leaf_value.add_(0.0)      # a dependent launch that changes nothing
        return r
    engine.BatchedMCTS._playout_lookahead = patched
    out = bench.mcts_leg(1024, 100, 0, True, 1, 0, None)
    us = out["seconds"] / (out["leaf_evals"] / 1024) * 1e6
    print(json.dumps({"extra_launches": k, "leaf_evals_per_sec": out["leaf_evals_per_sec"], "us_per_playout": us}), flush=True)


if __name__ == "__main__":
    if os.environ.get("IAGO_GAP_CHILD"):
        one(int(os.environ["IAGO_GAP_CHILD"]))
    else:
        for k in sys.argv[1:] or ["0", "1", "2", "4"]:
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, IAGO_GAP_CHILD=k))
