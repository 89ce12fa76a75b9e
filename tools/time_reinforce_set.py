"""Lab tool: where a REINFORCE set's time goes -- the 64 policy-vs-policy games (src/rl_self_play.py) against the update
(src/train_rl.py:55-66), host timers around synchronised halves, then the kernels of each half (torch profiler)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iago_amd import network
from iago_amd.train_rl import ReinforceTrainer
torch.manual_seed(0)
tr = ReinforceTrainer(network.SLPolicy(), pool_dir=None, N=32, seed=0)
for _ in range(4):
    tr.step()
torch.cuda.synchronize()
tp = tu = 0.0
n = 20
for _ in range(n):
    t0 = time.perf_counter()
    tup, _ = tr.play_set(tr.pick_opponent())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    tr._update(tup["own"], tup["opp"], tup["action"], tup["z"])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    tp += t1 - t0; tu += t2 - t1
print("play_set %.2f ms, update %.2f ms per set (%d tuples)" % (tp / n * 1e3, tu / n * 1e3, tup["own"].numel()))
# one play_set under the profiler's eyes: count launches
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tup, _ = tr.play_set(tr.pick_opponent())
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
# ---- the update alone, then its kernels
for _ in range(3):
    t0 = time.perf_counter()
    tr._update(tup["own"], tup["opp"], tup["action"], tup["z"])
    torch.cuda.synchronize()
    print("update %.2f ms" % ((time.perf_counter() - t0) * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tr._update(tup["own"], tup["opp"], tup["action"], tup["z"])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
