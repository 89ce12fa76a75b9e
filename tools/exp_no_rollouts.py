"""Lab: upper bound on what faster rollouts could give -- the same search with lmbda = 0 (no rollouts at all)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iago_amd import engine, network, ops
w, b = bench.shipped_rollout_weights()
sims = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for lm in (0.5, 0.0, 0.5, 0.0):
    torch.manual_seed(0)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    m = engine.BatchedMCTS(1024, policy, value, ops.RolloutWeights(w, b), n_thr=15, seed=7, persistent=True, lmbda=lm,
                           capacity=engine.suggest_capacity(sims, 15, moves=64))
    eng = engine.SelfPlayEngine(m)
    eng.play(sims, record=False); torch.cuda.synchronize()
    m.tree.reset() if hasattr(m.tree, "reset") else None
    m2 = engine.BatchedMCTS(1024, policy, value, ops.RolloutWeights(w, b), n_thr=15, seed=8, persistent=True, lmbda=lm,
                            capacity=engine.suggest_capacity(sims, 15, moves=64))
    eng2 = engine.SelfPlayEngine(m2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng2.play(sims, record=False); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = m2._ps["totals"].cpu().tolist()
    print("lmbda %.1f: %.1f ms, %.2f M leaf-evals/s, value requests %d policy %d, net wait/walk %.2f" % (
        lm, dt * 1e3, m2.n_leaf_evals / dt / 1e6, t[0], t[1], t[4] / max(1, t[5])), flush=True)
    m.close(); m2.close()
