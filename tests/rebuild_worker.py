"""Worker process of tests/test_bench_batch_gpu.py: rebuilds recorded games' searches with oracle/mcts_py.MCTS
(MCTS.py:105-154), the nets' outputs from the production kernels on one board (its own HIP context: the same nets,
random init seed 0).  Several of these run side by side: a whole game of 400-playout searches is ~24,000 playouts of
the Python restatement.

    python tests/rebuild_worker.py <in.npz> <out.json>

in.npz: n_sims, n_thr, games [k], n_turns [k] (turns to rebuild), compare_from [k], and per game g: pi_<g> [T][64],
move_<g> [T], zlog_<g> [playouts], game_turns_<g>.  out.json: {"compared": {g: n}, "max_path": {g: d}} or {"error": ...}.
"""
import json
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {}
    try:
        from iago_amd import ops
        from tests.bench_batch_util import Probe, make_nets, rebuild
        d = np.load(src)
        policy, value = make_nets()
        games = [int(g) for g in d["games"]]
        T = max(int(d["pi_%d" % g].shape[0]) for g in games)
        G = max(games) + 1
        B = dict(ops=ops, policy=policy, value=value, n_sims=int(d["n_sims"]),
                 pi=np.zeros((T, G, 64), np.int64), move=np.zeros((T, G), np.int64),
                 zlog=np.zeros((max(int(d["zlog_%d" % g].shape[0]) for g in games), G), np.int8),
                 zn=np.zeros(G, np.int64), game_turns=np.zeros(G, np.int64))
        for g in games:
            pi, mv, zl = d["pi_%d" % g], d["move_%d" % g], d["zlog_%d" % g]
            B["pi"][:pi.shape[0], g], B["move"][:mv.shape[0], g] = pi, mv
            B["zlog"][:zl.shape[0], g], B["zn"][g] = zl, zl.shape[0]
            B["game_turns"][g] = int(d["game_turns_%d" % g])
        probe = Probe(B)
        out["compared"], out["max_path"] = {}, {}
        for g, n_turns, start in zip(games, d["n_turns"], d["compare_from"]):
            out["compared"][str(g)] = rebuild(B, probe, g, int(n_turns), n_thr=int(d["n_thr"]), compare_from=int(start))
            out["max_path"][str(g)] = int(B["max_path"][g])
    except BaseException:   # (an assertion of rebuild(): reported to the test, which fails with it)
        out["error"] = traceback.format_exc()
    with open(dst, "w") as f:
        json.dump(out, f)
    sys.exit(1 if "error" in out else 0)


if __name__ == "__main__":
    main()
