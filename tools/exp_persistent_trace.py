"""Lab tool: timeline of ONE persistent launch (game workgroup 0's samples: queue depth, finished workgroups) --
where the launch is bound by the net workgroups and where by the games' own chains.

    python tools/exp_persistent_trace.py [turns=14] [playouts per move=100] [trace rows=4096]

With 64 turns (whole games) and enough rows the tail of the batch shows: the time between the first and the last game
workgroup's end, during which the net workgroups run dry."""
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
                       capacity=engine.suggest_capacity(int(sys.argv[2]) if len(sys.argv) > 2 else 100, 15,
                                                        moves=max(24, int(sys.argv[1]) if len(sys.argv) > 1 else 14)))
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
eng = engine.SelfPlayEngine(m, max_turns=int(sys.argv[1]) if len(sys.argv) > 1 else 14)
m.trace = torch.zeros((rows, 4), dtype=torch.int64, device="cuda")
import time  # noqa: E402
torch.cuda.synchronize()
t_host = time.perf_counter()
eng.play(sims, record=False)
torch.cuda.synchronize()
print("host time of the play: %.2f ms" % ((time.perf_counter() - t_host) * 1e3))
t = m.trace.cpu().numpy()
per_game = t[-1024:][::-1]          # the kernel's per-game rows: end tick, requests sent, turns
t = t[:-1024]
n = int((t[:, 0] > 0).sum())
t = t[:n]
ends = per_game[:, 0] / 1e5
req = per_game[:, 1]
print("games' ends (ms): min %.1f p10 %.1f median %.1f p90 %.1f max %.1f;  requests per game: min %d median %d max %d;  "
      "correlation(end, requests) %.2f" % (ends.min(), np.percentile(ends, 10), np.median(ends), np.percentile(ends, 90),
                                           ends.max(), req.min(), np.median(req), req.max(), np.corrcoef(ends, req)[0, 1]))
by_wg = ends.reshape(32, 32)
print("per workgroup: first end %s" % np.round(by_wg.min(1)[:8], 0), " last end %s" % np.round(by_wg.max(1)[:8], 0))
print("iterations of game workgroup 0 in the last search: %d, duration %.2f ms" % (n, t[-1, 0] / 1e5))
playouts = t[:, 3] >> 8
t[:, 3] &= 0xFF
step = max(1, n // 40)
for i in range(0, n, step):
    j = max(0, i - step)
    rate = (playouts[i] - playouts[j]) / max(1e-9, (t[i, 0] - t[j, 0]) / 100.0)   # per us
    print("t %9.1f us  queued %7d  backlog %5d  finished wgs %3d  playouts %9d  (%.1f M/s)" % (
        t[i, 0] / 100.0, t[i, 1], t[i, 1] - t[i, 2], t[i, 3], playouts[i], rate))
d = np.diff(t[:, 0]) / 100.0
print("iteration time us: mean %.1f median %.1f p90 %.1f" % (d.mean(), np.median(d), np.percentile(d, 90)))
fin = t[:, 3]
for k in (1, 8, 16, 24, 31):
    at = np.nonzero(fin >= k)[0]
    if len(at):
        print("%2d game workgroups finished at %.2f ms" % (k, t[at[0], 0] / 1e5))
