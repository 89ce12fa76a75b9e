"""Where a REINFORCE set's time goes (host-side sections, each closed by a device sync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iago_amd import network
from iago_amd.train_rl import ReinforceTrainer
torch.manual_seed(0)
tr = ReinforceTrainer(network.SLPolicy(), pool_dir=None, N=32, seed=0)
for _ in range(4):
    tr.step()
torch.cuda.synchronize()
acc = {}
def sec(name, f):
    t = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    return r
K = 20
t0 = time.perf_counter()
for _ in range(K):
    m2 = sec("pick_opponent", tr.pick_opponent)
    tup, wins = sec("play_set", lambda: tr.play_set(m2))
    loss = sec("update", lambda: tr._update(tup["own"], tup["opp"], tup["action"], tup["z"]))
    sec("item", lambda: float(loss.item()))
tot = time.perf_counter() - t0
for k, v in acc.items():
    print("%-14s %.3f ms" % (k, v / K * 1e3))
print("total %.3f ms per set (with the extra syncs)" % (tot / K * 1e3))
t0 = time.perf_counter()
for _ in range(K):
    tr.step()
torch.cuda.synchronize()
print("step(): %.3f ms per set" % ((time.perf_counter() - t0) / K * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        tr.step()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25))
