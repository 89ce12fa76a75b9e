#!/usr/bin/env python3
"""Timing experiment (not a search): what would a playout STEP cost if the Value evaluation left the
critical chain?  Games are independent, so a game whose leaf needs the net could wait for it
(two steps) while the games whose leaves have a stored value (84 %) go on: per step the main stream
runs descent -> rollout of all leaves (the stand-alone 16-lane kernel) -> backup, the Value launch
of the step's ~165 fresh rows runs on one of two alternating side streams and must be back two
steps later, the policy batch (170 rows) runs every 4 steps on a third.  Launch sequence and sizes
as in BASELINE configs[2] in mid-game; the trees are real, the bookkeeping is not (timing only).
Prints microseconds per step for the variants."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from iago_amd import _lib, engine, network, ops  # noqa: E402
from iago_amd.engine import _p, _stream  # noqa: E402

w, b = bench.shipped_rollout_weights()
W = ops.RolloutWeights(w, b)
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
G = 1024
m = engine.BatchedMCTS(G, policy, value, W, lmbda=0.5, c_puct=1.0, n_thr=15,
                       capacity=engine.suggest_capacity(100, 15), seed=7, use_graph=False)
L = _lib.lib()
own = torch.full((G,), engine.START_OWN, dtype=torch.int64, device="cuda")
opp = torch.full((G,), engine.START_OPP, dtype=torch.int64, device="cuda")
act = torch.ones(G, dtype=torch.uint8, device="cuda")
m.tree.reset()
for t in range(10):
    m.search(own, opp, act, 100)
    mv, _ = m.best_move(act)
    ops.apply_moves(own, opp, mv)
    m.update_with_move(mv)
    own, opp = opp, own
m.search(own, opp, act, 60)
torch.cuda.synchronize()

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 165
idx = torch.randperm(G, device="cuda")[:ROWS].contiguous()
n_rows = torch.tensor([ROWS], dtype=torch.int32, device="cuda")
vbuf = [torch.zeros(G, device="cuda") for _ in range(2)]
pol_own, pol_opp = own[:170].clone(), opp[:170].clone()
ro_out = ops.RolloutResult()
sid = torch.zeros(1, dtype=torch.int32, device="cuda")


def descend():
    m._fresh_count.zero_()
    L.iago_mcts_descend(m.tree.ref(), _p(own), _p(opp), _p(act), m.c_puct, m.n_thr, _p(m.cur_node), _p(m.cur_own),
                        _p(m.cur_opp), _p(m.legal), None, C.byref(m._la[0]), _p(m._fresh_idx), _p(m._fresh_count),
                        None, _stream())


def backup():
    m._la_queues[0]["count"].zero_()
    L.iago_mcts_mix_backup_lookahead(m.tree.ref(), _p(act), _p(m.cur_node), _p(m.cur_own), _p(m.cur_opp), _p(vbuf[0]),
                                     _p(ro_out.z), 0.5, _p(m.leaf_value), None, C.byref(m._la[0]), _stream())


def rollout():
    ops.rollout(m.cur_own, m.cur_opp, W, seed=7, stream_id=0, stream_id_dev=sid, out=ro_out)


def value_rows(k):
    with torch.no_grad():
        value.forward_boards_counted(m.cur_own, m.cur_opp, idx, n_rows, vbuf[k])


def fused_leaf():
    ro = ops.rollout_prepare(m.cur_own, m.cur_opp, W, seed=7, stream_id=0, stream_id_dev=sid, out=ro_out)
    with torch.no_grad():
        value.forward_boards_counted(m.cur_own, m.cur_opp, idx, n_rows, vbuf[0], rollout=ro)


def policy_batch():
    policy.forward_boards_split3(pol_own, pol_opp)


def timed(block, steps, reps=6):
    block()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            block()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        s.synchronize()
    return e0.elapsed_time(e1) / (reps * steps) * 1e3


STEPS = 12
DEPTH = int(sys.argv[2]) if len(sys.argv) > 2 else 2    # steps after which a Value launch must be back
sv = [torch.cuda.Stream() for _ in range(DEPTH)]
vbuf = [torch.zeros(G, device="cuda") for _ in range(DEPTH)]
sp = torch.cuda.Stream()


def lockstep(with_policy):
    main = torch.cuda.current_stream()
    for i in range(STEPS):
        if with_policy and i % 4 == 0:
            sp.wait_stream(main)
            with torch.cuda.stream(sp):
                policy_batch()
        if with_policy and i % 4 == 2:
            main.wait_stream(sp)
        descend()
        fused_leaf()
        backup()
    if with_policy:
        main.wait_stream(sp)


def asynchronous(with_policy, with_value=True):
    main = torch.cuda.current_stream()
    for i in range(STEPS):
        k = i % DEPTH
        if with_policy and i % 4 == 0:
            sp.wait_stream(main)
            with torch.cuda.stream(sp):
                policy_batch()
        if with_policy and i % 4 == 2:
            main.wait_stream(sp)
        if with_value and i >= DEPTH:
            main.wait_stream(sv[k])          # the Value rows of step i - DEPTH are back
        descend()
        if with_value:
            sv[k].wait_stream(main)
            with torch.cuda.stream(sv[k]):
                value_rows(k)
        rollout()
        backup()
    for s in sv:
        main.wait_stream(s)
    if with_policy:
        main.wait_stream(sp)


print("rows per Value launch: %d, Value back after %d steps" % (ROWS, DEPTH))
print("lockstep (descent, fused leaf evaluation, backup), no policy : %6.1f us per step" % timed(lambda: lockstep(False), STEPS))
print("lockstep + policy batch every 4 steps                        : %6.1f us per step" % timed(lambda: lockstep(True), STEPS))
print("main chain only (descent, rollout, backup)                   : %6.1f us per step" % timed(lambda: asynchronous(False, False), STEPS))
print("asynchronous Value on side streams, no policy                : %6.1f us per step" % timed(lambda: asynchronous(False), STEPS))
print("asynchronous Value + policy batch every 4 steps              : %6.1f us per step" % timed(lambda: asynchronous(True), STEPS))
