"""Lab tool: throughput of the value-data generator (value_self_play.py:32-59: SL policy to a random ply, one random move,
RL policy to the end; one (state, z) per game) for a lockstep batch."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iago_amd import network, value_self_play
torch.manual_seed(0)
sl, rl = network.SLPolicy().cuda().eval(), network.SLPolicy().cuda().eval()
for B in (64, 1024, 4096):
    value_self_play.generate(sl, rl, B, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = value_self_play.generate(sl, rl, B, seed=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("B %5d: %.1f ms, %.0f games/s, %d turns, dropped %d" % (B, dt * 1e3, B / dt, r["n_turns"], int(r["dropped"].sum())))
