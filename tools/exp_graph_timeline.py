#!/usr/bin/env python3
"""Timeline of the captured PV-MCTS playout graph (BASELINE configs[2]): one-thread stamp kernels
(tools/stamp_kernel.hip -> tools/_build/libstamp.so, built with
`hipcc -O2 --offload-arch=gfx950 -shared -fPIC`) are enqueued between the launches of a block of
8 playouts -- before / after the descent, after the leaf evaluation, after the backup, and around
the policy batch on the second stream -- and captured with them.  After a game the stamps give
the mean duration of every segment as the graph really runs it (the stamps add a launch each: the
absolute times are upper bounds, the proportions are the information)."""
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from iago_amd import engine, network, ops  # noqa: E402

S = C.CDLL(os.path.join(ROOT, "tools", "_build", "libstamp.so"))
S.stamp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
CAP = 1 << 20
buf = torch.zeros(2 * CAP, dtype=torch.int64, device="cuda")
idx = torch.zeros(1, dtype=torch.int32, device="cuda")


def stamp(label):
    S.stamp(buf.data_ptr(), idx.data_ptr(), CAP, label, torch.cuda.current_stream().cuda_stream)


D0, D1, V1, B1, P0, P1 = range(6)
w, b = bench.shipped_rollout_weights()
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
m = engine.BatchedMCTS(1024, policy, value, ops.RolloutWeights(w, b), lmbda=0.5, c_puct=1.0, n_thr=15,
                       capacity=engine.suggest_capacity(100, 15), seed=7, use_graph=True)
m.warmup()

orig_playout = m._playout_lookahead
orig_counted = value.forward_boards_counted
orig_flush = m._flush_lookahead


def playout(*a, **k):
    stamp(D0)
    r = orig_playout(*a, **k)
    stamp(B1)
    return r


def counted(*a, **k):
    stamp(D1)
    r = orig_counted(*a, **k)
    stamp(V1)
    return r


def flush(*a, **k):
    stamp(P0)
    r = orig_flush(*a, **k)
    stamp(P1)
    return r


m._playout_lookahead = playout
value.forward_boards_counted = counted
m._flush_lookahead = flush
res = engine.SelfPlayEngine(m, max_turns=128).play(100, record=False)
torch.cuda.synchronize()
n = min(int(idx.item()), CAP)
a = buf[:2 * n].cpu().numpy().reshape(n, 2)
lab, t = a[:, 0], a[:, 1].astype(np.float64) * 0.01   # us (100 MHz)
seg = collections.defaultdict(list)
last = {}
n_eval = 0
for L, x in zip(lab, t):
    L = int(L)
    if L == V1 and D1 in last:
        # by the playout's place in its look-ahead group of 4: the policy batch of the previous
        # group runs beside playouts 0 and 1 (lookahead_overlap = 2)
        seg["leaf evaluation, playout %d of its group" % (n_eval % 4)].append(x - last[D1])
        n_eval += 1
    if L == D1 and D0 in last:
        seg["descent"].append(x - last[D0])
    if L == V1 and D1 in last:
        seg["leaf evaluation"].append(x - last[D1])
    if L == B1 and V1 in last:
        seg["backup"].append(x - last[V1])
    if L == D0 and B1 in last:
        seg["backup end -> next descent start"].append(x - last[B1])
    if L == D0 and D0 in last:
        seg["playout period"].append(x - last[D0])
    if L == P1 and P0 in last:
        seg["policy batch"].append(x - last[P0])
    last[L] = x
print("turns %d, stamps %d" % (res.n_turns, n))
g = np.asarray(seg["backup end -> next descent start"])
big = g[g > 60.0]
print("gaps backup -> next descent above 60 us (between the searches of two moves, replay boundaries): "
      "%d, mean %.1f us, total %.1f ms of %.1f ms" % (len(big), big.mean() if len(big) else 0.0, big.sum() / 1e3,
                                                        (t.max() - t.min()) / 1e3))
for k, v in seg.items():
    v = np.asarray(v)
    v = v[v < 5000]   # (drop the gaps between moves: host work)
    print("%-34s n %6d  mean %7.1f us  median %7.1f  p10 %7.1f  p90 %7.1f" % (
        k, len(v), v.mean(), np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
