"""Lab tool: where a game workgroup's iteration goes (replies + moves, descent, rollouts, backup, end of iteration), clock
stamps of game workgroup 0.  Needs the stamped variant of the search kernel:

    python tools/build_search_variants.py
    IAGO_HIP_LIB=$PWD/tools/_build/search_phases.so python tools/exp_game_phases.py
"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iago_amd import engine, network, ops
w, b = bench.shipped_rollout_weights()
for games, turns in ((1024, 64), (1024, 12), (1, 64)):
    torch.manual_seed(0)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    m = engine.BatchedMCTS(games, policy, value, ops.RolloutWeights(w, b), n_thr=15, seed=7, persistent=True,
                           capacity=engine.suggest_capacity(100, 15, moves=64))
    eng = engine.SelfPlayEngine(m, max_turns=turns)
    eng.play(100, record=False)
    torch.cuda.synchronize()
    t = m._ps["totals"].cpu().tolist()
    it = t[2] / max(1, -(-games // 32))
    ph = [x / 100.0 for x in t[11:16]]
    print("games %d turns %d: iterations per game workgroup %.0f; workgroup 0 us: replies/moves %.0f descent %.0f rollouts %.0f backup %.0f end %.0f (sum %.0f)" % (
        games, turns, it, *ph, sum(ph)))
    print("  per iteration us: " + " ".join("%.1f" % (x / it) for x in ph))
    m.close()
