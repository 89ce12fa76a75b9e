"""Lab: descent levels per playout over whole games vs without the last turns (400 playouts per move)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iago_amd import engine, network, ops
w, b = bench.shipped_rollout_weights()
sims = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for turns in (40, 52, 56, 60, 64):
    torch.manual_seed(0)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    m = engine.BatchedMCTS(1024, policy, value, ops.RolloutWeights(w, b), n_thr=15, seed=7, persistent=True,
                           capacity=engine.suggest_capacity(sims, 15, moves=64))
    m.enable_stats()
    eng = engine.SelfPlayEngine(m, max_turns=turns)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.play(sims, record=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lv, ch = (int(x) for x in m.stats.to(torch.int64).sum(dim=0).tolist())
    per_game = m.stats[:, 0].to(torch.float64)
    print("turns %d: %.1f ms, leaf evals %d, levels %d (%.2f per playout; per game min %.0f max %.0f), children scored %d" % (
        turns, dt * 1e3, m.n_leaf_evals, lv, lv / max(1, m.n_leaf_evals), per_game.min().item(), per_game.max().item(), ch), flush=True)
    m.close()
