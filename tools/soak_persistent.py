"""Lab tool: soak of the persistent search -- whole batches of games of many sizes back to back, fresh engines, the error
flags read after every batch (a launch that gave up, a full pool) and the results of repeated batches compared."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iago_amd import engine, network, ops
w, b = bench.shipped_rollout_weights()
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(1)
ref = {}
t_all = time.perf_counter()
for i in range(rounds):
    games = int(rs.choice([1, 7, 64, 100, 256, 257, 512, 1000, 1024, 2048, 4096]))
    sims = int(rs.choice([16, 30, 100]))
    if games >= 2048:
        sims = 16
    m = engine.BatchedMCTS(games, policy, value, ops.RolloutWeights(w, b), n_thr=15, seed=3,
                           capacity=engine.suggest_capacity(sims, 15, moves=64))
    assert m.persistent
    t0 = time.perf_counter()
    r = engine.SelfPlayEngine(m).play(sims)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sig = (int(r.move.to(torch.int64).sum()), int(r.z.to(torch.int64).sum()), r.n_turns)
    key = (games, sims)
    same = ref.setdefault(key, sig) == sig
    print("%3d: %5d games x %3d playouts: %7.1f ms, %6.2f M leaf-evals/s, turns %d%s" % (
        i, games, sims, dt * 1e3, m.n_leaf_evals / dt / 1e6, r.n_turns, "" if same else "  RESULT DIFFERS FROM THE FIRST RUN"), flush=True)
    assert same
    m.close()
print("soak ok: %d batches in %.1f s" % (rounds, time.perf_counter() - t_all))
