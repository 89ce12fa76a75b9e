"""Lab tool: timeline of ONE persistent search in mid-game (game workgroup 0's samples: queue depth, finished
workgroups) -- where the launch is bound by the net workgroups and where by the games' own chains."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from iago_amd import engine, network, ops  # noqa: E402

w, b = bench.shipped_rollout_weights()
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
m = engine.BatchedMCTS(1024, policy, value, ops.RolloutWeights(w, b), n_thr=15, seed=7, persistent=True,
                       capacity=engine.suggest_capacity(100, 15, moves=24))
eng = engine.SelfPlayEngine(m, max_turns=int(sys.argv[1]) if len(sys.argv) > 1 else 14)
m.trace = torch.zeros((4096, 4), dtype=torch.int64, device="cuda")
eng.play(100, record=False)
torch.cuda.synchronize()
t = m.trace.cpu().numpy()
n = int((t[:, 0] > 0).sum())
t = t[:n]
print("iterations of game workgroup 0 in the last search: %d, duration %.2f ms" % (n, t[-1, 0] / 1e5))
for i in range(0, n, max(1, n // 40)):
    print("t %7.1f us  queued %6d  tickets %6d  backlog %5d  finished wgs %3d" % (
        t[i, 0] / 100.0, t[i, 1], t[i, 2], t[i, 1] - t[i, 2], t[i, 3]))
d = np.diff(t[:, 0]) / 100.0
print("iteration time us: mean %.1f median %.1f p90 %.1f" % (d.mean(), np.median(d), np.percentile(d, 90)))
