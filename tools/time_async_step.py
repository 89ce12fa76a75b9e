#!/usr/bin/env python3
"""Game-asynchronous steps against lockstep playouts on mid-game trees (1024 games, BASELINE
configs[2] constants): seconds per 100-playout search, steps per search, and the three launches of
a step timed one by one (eager, HIP events) with the policy batches left out."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from iago_amd import _lib, engine, network, ops  # noqa: E402
from iago_amd.engine import _p, _stream  # noqa: E402

G = 1024
w, b = bench.shipped_rollout_weights()
W = ops.RolloutWeights(w, b)
PARTS = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def make(async_steps):
    torch.manual_seed(0)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    m = engine.BatchedMCTS(G, policy, value, W, n_thr=15, capacity=engine.suggest_capacity(100, 15), seed=7,
                           use_graph=True, async_steps=async_steps, async_parts=PARTS if async_steps else None)
    own = torch.full((G,), engine.START_OWN, dtype=torch.int64, device="cuda")
    opp = torch.full((G,), engine.START_OPP, dtype=torch.int64, device="cuda")
    act = torch.ones(G, dtype=torch.uint8, device="cuda")
    m.tree.reset()
    times = []
    for t in range(14):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s0 = m.n_steps
        m.search(own, opp, act, 100)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0, m.n_steps - s0))
        mv, _ = m.best_move(act)
        ops.apply_moves(own, opp, mv)
        m.update_with_move(mv)
        own, opp = opp, own
    return m, own, opp, act, times


for mode in (False, True):
    m, own, opp, act, times = make(mode)
    print("async" if mode else "lockstep", "ms per search, moves 4..13:",
          " ".join("%.2f" % (1e3 * t) for t, _ in times[4:]), "| steps:", [s for _, s in times[4:]])
    if not mode:
        m.close()
        continue
    # a graph of 16 steps WITHOUT the policy batches, on a fresh search of these trees
    m._a_wait.zero_(); m._a_done.zero_(); m._a_roll.zero_(); m._a_fq_count.zero_(); m._a_nsims.fill_(100)
    m._g_own.copy_(own); m._g_opp.copy_(opp); m._g_active.copy_(act)
    m._la_cur = 0
    m._step_async(m._g_own, m._g_opp, m._g_active)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(16):
            m._step_async(m._g_own, m._g_opp, m._g_active)
    out = []
    for rep in range(9):
        m._la_queues[0]["count"].zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append("%.1f (rows in flight %d, rolled %d)" % (e0.elapsed_time(e1) / 16 * 1e3, int(m._a_fq_count.sum()),
                                                            int(m._a_roll.sum())))
    print("us per step, graph of 16 steps, no policy batches:", "; ".join(out))
    print("done min/mean/max after 145 steps:", int(m._a_done.min()), float(m._a_done.float().mean()), int(m._a_done.max()))
    m._la_error.zero_()
    # the launches of one step, timed one by one over a fresh search (eager, no policy batches)
    L = _lib.lib()
    m._a_wait.zero_(); m._a_done.zero_(); m._a_roll.zero_(); m._a_fq_count.zero_(); m._a_nsims.fill_(100)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = [0.0, 0.0, 0.0]
    rows = []
    N = 120
    for s in range(N):
        ev[0].record()
        L.iago_mcts_descend(m.tree.ref(), _p(own), _p(opp), _p(act), m.c_puct, m.n_thr, _p(m.cur_node), _p(m.cur_own),
                            _p(m.cur_opp), _p(m.legal), None, C.byref(m._la_async[0]), None, None,
                            _p(m._value_total), _stream())
        ev[1].record()
        ro = ops.rollout_prepare(m.cur_own, m.cur_opp, W, seed=7, stream_id=0, stream_id_dev=m._sim_dev,
                                 out=m._rollout_out)
        with torch.no_grad():
            m.value_fn.forward_boards_async(m.cur_own, m.cur_opp, m.v, ro, C.byref(m._async))
        ev[2].record()
        L.iago_mcts_mix_backup_lookahead(m.tree.ref(), _p(act), _p(m.cur_node), _p(m.cur_own), _p(m.cur_opp), _p(m.v),
                                         _p(m.z), m.lmbda, _p(m.leaf_value), None, C.byref(m._la_async[0]), _stream())
        ev[3].record()
        torch.cuda.synchronize()
        for k in range(3):
            acc[k] += ev[k].elapsed_time(ev[k + 1])
        rows.append((tuple(m._a_fq_count.tolist()), int(m._a_roll.sum().item()),
                     round(ev[1].elapsed_time(ev[2]) * 1e3, 1)))
        m._la_queues[0]["count"].zero_()   # (no flush here: the queue would overflow)
    print("per step us: descent %.1f  leaf-eval %.1f  backup %.1f" % tuple(1e3 * a / N for a in acc))
    print("(queued rows in flight, games rolled, leaf-eval us) every 6th step:", rows[::6])
    print("done min/mean/max after %d steps:" % N, int(m._a_done.min()), float(m._a_done.float().mean()),
          int(m._a_done.max()))
