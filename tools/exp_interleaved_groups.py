#!/usr/bin/env python3
"""Experiment: BASELINE configs[2]'s 1024 lockstep games as G groups (own engine, tree pools,
hipGraph and streams each), ONE host thread enqueueing the groups' graph replays alternately so
that one group's descend / backup launches run beside another group's leaf evaluation.
Prints leaf-evals/s for every G given."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from iago_amd import engine, network, ops  # noqa: E402

N_GAMES, N_SIMS = 1024, 100
w, b = bench.shipped_rollout_weights()
torch.manual_seed(0)
policy = network.SLPolicy().cuda().eval()
value = network.Value().cuda().eval()


class Group(object):
    def __init__(self, n, g):
        self.s = torch.cuda.Stream()
        with torch.cuda.stream(self.s):
            self.m = engine.BatchedMCTS(n, policy, value, ops.RolloutWeights(w, b), lmbda=0.5, c_puct=1.0, n_thr=15,
                                        capacity=engine.suggest_capacity(N_SIMS, 15), seed=7, game_id_base=g * n,
                                        use_graph=True)
            self.m.warmup()
            engine.SelfPlayEngine(self.m, max_turns=4).play(16, record=False)
            self.m.n_leaf_evals = 0
            self.m.tree.reset()
            self.own = torch.full((n,), engine.START_OWN, dtype=torch.int64, device="cuda")
            self.opp = torch.full((n,), engine.START_OPP, dtype=torch.int64, device="cuda")
            self.stones = torch.full((n,), 4, dtype=torch.int32, device="cuda")
            self.pass_flg = torch.zeros(n, dtype=torch.bool, device="cuda")
            self.done = torch.zeros(n, dtype=torch.bool, device="cuda")
        self.s.synchronize()


def play(groups):
    t = 0
    while t < 128:
        for G in groups:
            with torch.cuda.stream(G.s):
                G.active = ((ops.legal_moves(G.own, G.opp) != 0) & ~G.done).to(torch.uint8)
                G.cnt = G.active.sum()
        for G in groups:
            with torch.cuda.stream(G.s):     # .item() copies on the current stream
                G.n_active = int(G.cnt.item())
        live = [G for G in groups if G.n_active]
        for G in live:
            m = G.m
            with torch.cuda.stream(G.s):
                if m._graph is None:
                    m._capture()
                    m._graph_key = m._graph_state()
                m._g_own.copy_(G.own)
                m._g_opp.copy_(G.opp)
                m._g_active.copy_(G.active)
                m._sim_dev.fill_(m.sim_counter)
        block = 2 * live[0].m.lookahead if live else 1
        for _ in range(N_SIMS // block):
            for G in live:
                with torch.cuda.stream(G.s):
                    G.m._graph.replay()
        for G in live:
            m = G.m
            with torch.cuda.stream(G.s):
                m._lookahead_tail(m._g_own, m._g_opp, m._g_active, N_SIMS % block, None)
                m.sim_counter += N_SIMS
                m.n_leaf_evals += G.n_active * N_SIMS
        for G in groups:
            m = G.m
            with torch.cuda.stream(G.s):
                move, _ = m.best_move(G.active)
                mv = torch.where(G.active.bool(), move, torch.full_like(move, -1))
                ops.apply_moves(G.own, G.opp, mv)
                placed = G.active.bool()
                G.stones = G.stones + placed.to(torch.int32)
                passing = ~placed & ~G.done
                G.stones = torch.where(passing & G.pass_flg, torch.full_like(G.stones, 64), G.stones)
                G.pass_flg = torch.where(G.done, G.pass_flg, passing)
                m.update_with_move(mv, (~G.done).to(torch.uint8))
                G.own, G.opp = G.opp, G.own
                if t % 2 == 1:
                    G.done = G.done | (G.stones >= 64)
                    G.all_done = G.done.all()
        t += 1
        if t % 2 == 0:
            fin = []
            for G in groups:
                with torch.cuda.stream(G.s):
                    fin.append(bool(G.all_done.item()))
            if all(fin):
                break
    return t


for ng in [int(x) for x in (sys.argv[1:] or ["1", "2", "4"])]:
    groups = [Group(N_GAMES // ng, g) for g in range(ng)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    turns = play(groups)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    leaf = sum(G.m.n_leaf_evals for G in groups)
    torch.cuda.synchronize()
    ovf = sum(int(G.m.tree.overflow.sum().item()) for G in groups)
    print("groups %d x %4d games: %.3f s, %.3f M leaf-evals/s, %.0f games/s, %d turns, overflow %d" % (
        ng, N_GAMES // ng, dt, leaf / dt / 1e6, N_GAMES / dt, turns, ovf), flush=True)
    del groups
    import gc
    gc.collect()
    torch.cuda.synchronize()
