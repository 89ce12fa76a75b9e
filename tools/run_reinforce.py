#!/usr/bin/env python3
"""BASELINE configs[4] at one GPU's share: the REINFORCE loop of src/train_rl.py:28-81 for
`iters` sets (default 1000, the reference's `--set 1000`): 64 SLPolicy-vs-pool games per set
(odd games with the handicap stone), the double-softmax update with ChainerAdam + weight
decay, the snapshot gating into a pool directory.  One JSON line.
    python3 tools/run_reinforce.py [iters=1000] [pool_dir=/tmp/iago_rl_pool]"""
import json, os, shutil, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iago_amd import network  # noqa: E402
from iago_amd.train_rl import ReinforceTrainer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pool = sys.argv[2] if len(sys.argv) > 2 else "/tmp/iago_rl_pool"
shutil.rmtree(pool, ignore_errors=True)
os.makedirs(pool)
torch.manual_seed(0)
model = network.SLPolicy()
model.save_npz(os.path.join(pool, "model0.npz"))      # the pool starts with the SL policy (src/train_rl.py:33)
tr = ReinforceTrainer(model, pool_dir=pool, N=32, seed=0)
for _ in range(2):
    tr.step()                                         # warm-up (MIOpen kernel selection)
torch.cuda.synchronize()
t0 = time.perf_counter()
done = 0
stopped = False
for _ in range(iters):
    r = tr.step()
    done += 1
    if r["stop"]:                                     # src/train_rl.py:80-81: rate < 0.2 ends the run
        stopped = True
        break
torch.cuda.synchronize()
dt = time.perf_counter() - t0
log = tr.log[2:]
print(json.dumps({"config": "REINFORCE, 64 games per set (N = 32, handicap on odd games), SLPolicy random init "
                            "(seed 0) vs a pool seeded with itself, ChainerAdam alpha 1e-3 + WD 5e-4, 1 x MI355X",
                  "iterations": done, "stopped_by_rate_below_0.2": stopped, "seconds": dt,
                  "iters_per_sec": done / dt, "games_per_sec": 64 * done / dt,
                  "tuples": sum(x["n_tuples"] for x in log),
                  "snapshots_saved": sum(1 for x in log if x["saved"]), "pool_models": tr.models,
                  "mean_rate_first_100": sum(x["rate"] for x in log[:100]) / max(1, len(log[:100])),
                  "mean_rate_last_100": sum(x["rate"] for x in log[-100:]) / max(1, len(log[-100:])),
                  "loss_first": log[0]["loss"], "loss_last": log[-1]["loss"],
                  "adam_t": int(tr.opt.t)}))
