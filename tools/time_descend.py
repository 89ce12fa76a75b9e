#!/usr/bin/env python3
"""Duration of iago_mcts_descend / iago_mcts_mix_backup_lookahead alone on mid-search trees
(1024 games, 60 playouts into the first move and after 10 moves), 20 calls captured in one
hipGraph; with all games active and with 1 game in 8 / 1 in 64 (is a launch bound by the slowest
wave or by the number of waves?)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from iago_amd import _lib, engine, network, ops  # noqa: E402
from iago_amd.engine import _p, _stream  # noqa: E402

w, b = bench.shipped_rollout_weights()
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
G = 1024
m = engine.BatchedMCTS(G, policy, value, ops.RolloutWeights(w, b), lmbda=0.5, c_puct=1.0, n_thr=15,
                       capacity=engine.suggest_capacity(100, 15), seed=7, use_graph=False)
L = _lib.lib()


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        s.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def measure(tag, own, opp):
    for every in (1, 8, 64):
        act = torch.zeros(G, dtype=torch.uint8, device="cuda")
        act[::every] = 1

        def descend():
            m._fresh_count.zero_()
            L.iago_mcts_descend(m.tree.ref(), _p(own), _p(opp), _p(act), m.c_puct, m.n_thr, _p(m.cur_node),
                                _p(m.cur_own), _p(m.cur_opp), _p(m.legal), None, C.byref(m._la[0]),
                                _p(m._fresh_idx), _p(m._fresh_count), None, _stream())

        def zero_only():
            m._fresh_count.zero_()

        t_d = timed(descend) - timed(zero_only)
        print("%s, %4d games active: descent %.1f us per call (a repeated descent of unchanged trees)" % (
            tag, int(act.sum().item()), t_d), flush=True)


sp = engine.SelfPlayEngine(m, max_turns=128)
own = torch.full((G,), engine.START_OWN, dtype=torch.int64, device="cuda")
opp = torch.full((G,), engine.START_OPP, dtype=torch.int64, device="cuda")
m.tree.reset()
act = torch.ones(G, dtype=torch.uint8, device="cuda")
m.search(own, opp, act, 60)
measure("first move, 60 playouts in", own, opp)
for t in range(10):
    m.search(own, opp, act, 100)
    mv, _ = m.best_move(act)
    ops.apply_moves(own, opp, mv)
    m.update_with_move(mv)
    own, opp = opp, own
m.search(own, opp, act, 60)
measure("move 11, 60 playouts in", own, opp)
m.tree.v.fill_(float("nan"))   # every leaf "fresh": 1024 appends to the list through one counter
measure("move 11, every leaf without a stored value", own, opp)
