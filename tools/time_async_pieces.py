#!/usr/bin/env python3
"""The leaf-evaluation launch of a game-asynchronous step (iago_value_rollout_async) alone:
microseconds per launch with chosen rows in chosen queues (piece p of the value net runs on queue
row (step - p) mod parts), with and without rollouts, against the whole one-board walk."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from iago_amd import engine, network, ops  # noqa: E402

G = 1024
PARTS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w, b = bench.shipped_rollout_weights()
W = ops.RolloutWeights(w, b)
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
m = engine.BatchedMCTS(G, policy, value, W, n_thr=15, capacity=1024, seed=7, use_graph=False, async_steps=True,
                       async_parts=PARTS)
own, opp = bench.torch.full((G,), engine.START_OWN, dtype=torch.int64, device="cuda"), None
opp = torch.full((G,), engine.START_OPP, dtype=torch.int64, device="cuda")
m.cur_own.copy_(own)
m.cur_opp.copy_(opp)
m._a_fq_index.copy_(torch.arange(G, device="cuda").reshape(1, G).expand(PARTS, G))
ro = ops.rollout_prepare(m.cur_own, m.cur_opp, W, seed=7, stream_id=0, stream_id_dev=m._sim_dev, out=m._rollout_out)


def launch():
    with torch.no_grad():
        m.value_fn.forward_boards_async(m.cur_own, m.cur_opp, m.v, ro, C.byref(m._async))


def timed(counts, roll, reps=30):
    """counts[p] = rows of the queue piece p works on; roll: games rolled out."""
    m._a_step.zero_()
    for p in range(PARTS):
        m._a_fq_count[(0 - p) % PARTS] = counts[p]
    m._a_roll.zero_()
    m._a_roll[:roll] = 1
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()      # (a launch costs the host more than the GPU: replay a graph of them)
    with torch.cuda.graph(g):
        for _ in range(reps):
            launch()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for counts, roll in [((0,) * PARTS, 0), ((0,) * PARTS, 1024)] + \
        [(tuple(50 if q == p else 0 for q in range(PARTS)), 0) for p in range(PARTS)] + \
        [((50,) * PARTS, 0), ((50,) * PARTS, 1024), ((110,) * PARTS, 0), ((110,) * PARTS, 1024),
         ((8,) * PARTS, 1024), ((1,) * PARTS, 0)]:
    print("rows per piece %-18s rolled %4d : %7.1f us" % (counts, roll, timed(counts, roll)))
idx = torch.arange(G, device="cuda")
for n in ():
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    out = torch.zeros(G, device="cuda")
    for _ in range(3):
        value.forward_boards_counted(own, opp, idx, cnt, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        value.forward_boards_counted(own, opp, idx, cnt, out)
    e1.record()
    torch.cuda.synchronize()
    print("whole walk, %3d rows (one-board persistent launch): %7.1f us" % (n, e0.elapsed_time(e1) / 30 * 1e3))
