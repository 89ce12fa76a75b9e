#!/usr/bin/env python3
"""BASELINE configs[4] as it is worded -- "self-play feeding train_rl.py REINFORCE update on gathered
(s, pi, z), 1000 iterations" -- at one GPU's share: per iteration one round of PV-MCTS self-play
(`games` lockstep games played to the end, `sims` playouts per move, the learner as the search's
policy net, a fixed random-init Value net, shipped RolloutPolicy) -> SelfPlayResult.tuples() ->
ReinforceTrainer.step_from_tuples (gather, canonical order, double-softmax REINFORCE update,
ChainerAdam + weight decay; the search engine re-captures its graph when the weights change).
One JSON line.
    python3 tools/run_reinforce_mcts.py [iters=1000] [games=64] [sims=20]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from iago_amd import engine, network, ops  # noqa: E402
from iago_amd.train_rl import ReinforceTrainer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
games = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 20
w, b = bench.shipped_rollout_weights()
torch.manual_seed(0)
tr = ReinforceTrainer(network.SLPolicy(), pool_dir=None, N=32, seed=0)
value = network.Value().cuda().eval()
m = engine.BatchedMCTS(games, tr.model1, value, ops.RolloutWeights(w, b), n_thr=15,
                       capacity=engine.suggest_capacity(sims, 15), seed=1)   # default engine: the persistent search
sp = engine.SelfPlayEngine(m)


def one():
    tr.model1.eval()
    res = sp.play(sims)
    return tr.step_from_tuples(res.tuples()), res


for _ in range(2):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
tuples = leaf0 = 0
leaf0 = m.n_leaf_evals
losses = []
for i in range(iters):
    out, res = one()
    tuples += out["n_tuples"]
    losses.append(out["loss"])
    if (i + 1) % 100 == 0:
        print("iteration %d, %.1f s" % (i + 1, time.perf_counter() - t0), file=sys.stderr, flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"config": "PV-MCTS self-play (%d games per round, %d playouts per move, n_thr 15, lmbda 0.5, the engine's default: the persistent search, one launch per round) -> tuples (own, opp, move, z) -> "
                            "REINFORCE update (ChainerAdam alpha 1e-3 + WD 5e-4), 1 x MI355X" % (games, sims),
                  "iterations": iters, "seconds": dt, "iters_per_sec": iters / dt, "games_per_sec": games * iters / dt,
                  "tuples": tuples, "leaf_evals": m.n_leaf_evals - leaf0,
                  "leaf_evals_per_sec": (m.n_leaf_evals - leaf0) / dt,
                  "loss_first": losses[0], "loss_last": losses[-1], "adam_t": int(tr.opt.t),
                  "turns_last_round": res.n_turns}))
