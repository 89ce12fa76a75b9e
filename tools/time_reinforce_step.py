import sys, time, torch
sys.path.insert(0, "/root/repo")
from iago_amd import network
from iago_amd.train_rl import ReinforceTrainer
import iago_amd.train_rl as T
torch.manual_seed(0)
import os
if os.environ.get("BENCHMARK") == "1":
    torch.backends.cudnn.benchmark = True
tr = ReinforceTrainer(network.SLPolicy(), pool_dir=None, N=32, seed=0)
t0 = time.perf_counter()
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
print("3 warm-up steps: %.1f s" % (time.perf_counter() - t0))
def timed(f, *a):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
for it in range(4):
    m2, t_pick = timed(tr.pick_opponent)
    (tup, res), t_play = timed(tr.play_set, m2)
    loss, t_upd = timed(tr._update, tup["own"], tup["opp"], tup["action"], tup["z"])
    print("pick %.1f ms  play_set %.1f ms  update %.1f ms  tuples %d" % (t_pick, t_play, t_upd, tup["z"].numel()))
t0 = time.perf_counter()
for _ in range(10):
    tr.step()
torch.cuda.synchronize()
print("10 steps: %.1f ms each" % ((time.perf_counter() - t0) * 100))
