"""Lab tool: where a game workgroup's iteration goes (replies + moves, descent, rollouts, backup, end of iteration), clock
stamps of game workgroup 0.  Needs the stamped variant of the search kernel:

    python tools/build_search_variants.py
    IAGO_HIP_LIB=$PWD/tools/_build/search_phases.so python tools/exp_game_phases.py
"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iago_amd import _lib, engine, network, ops
L = _lib.lib()
L.iago_debug_game_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
w, b = bench.shipped_rollout_weights()
NAMES = ("replies + moves", "descent", "control words + packing", "rollout passes", "backup", "end of iteration")
# python tools/exp_game_phases.py [playouts per move = 100] [n_thr = 15]   (round 6: 400 / 15 = configs[3]'s share, 100 / 1)
SIMS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N_THR = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for games, turns in ((1024, 128), (1024, 12), (64, 128)):
    torch.manual_seed(0)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    m = engine.BatchedMCTS(games, policy, value, ops.RolloutWeights(w, b), n_thr=N_THR, seed=7, persistent=True, split=0,
                           capacity=engine.suggest_capacity(SIMS, N_THR, moves=64))
    eng = engine.SelfPlayEngine(m, max_turns=turns)
    L.iago_debug_game_phases(buf, 1)
    eng.play(SIMS, record=False)
    torch.cuda.synchronize()
    L.iago_debug_game_phases(buf, 1)
    t = list(buf)
    it = max(1, t[7])
    print("%d playouts per move, n_thr %d; games %d, %d turns: game workgroup 0: %d iterations, %.1f us each:" % (SIMS, N_THR, games, turns, it, sum(t[:6]) / it / 100.0))
    print("   " + ";  ".join("%s %.2f" % (n, x / it / 100.0) for n, x in zip(NAMES, t[:6])))
    m.close()
