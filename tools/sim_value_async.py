"""CPU simulation (lab tool): value look-ahead + wait-on-miss schedule.

Per game the trace of oracle PV-MCTS (tools/sim_value_lookahead.TracedMCTS) is replayed against a wall
clock of steps: a game executes one playout per step unless it waits for the value of a fresh leaf.
Prefetch strategies queue children for a batched background value net (batch every KV steps, lands LV
steps after its launch).  A miss waits MISS_WAIT steps (the three-piece inline walk of the async engine)
or for the batch.  Reports, per strategy: hit rate, wasted evaluations, steps per 100-playout search
for 1024 lockstep-ended games (max of 1024 draws from the per-(game, search) distribution).
"""
import argparse
import os
import pickle
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sim_value_lookahead as S  # noqa: E402
from oracle import mcts_py, oracle as orc  # noqa: E402
from iago_amd import network  # noqa: E402


class Logged(S.TracedMCTS):
    """adds a per-playout event log: ('trig', X), ('exp', X), ('fresh', X, rank) with X = index into events
    (root / unknown parent: X = -1)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.log = []       # (playout index, kind, X, rank)
        self.trig_nodes = {}


def play_logged(n_sims, policy, value, rng, trigger=10):
    def pol(x):
        with torch.no_grad():
            return policy(torch.from_numpy(x)).numpy()

    def val(x):
        with torch.no_grad():
            return value(torch.from_numpy(x)).numpy()[0]

    def roll(state, c):
        return int(rng.integers(-1, 2))

    m = S.TracedMCTS(pol, val, roll, trigger=trigger)
    mcts_py.selfplay_game(m, n_sims)
    return m


def replay(m, n_sims, strat, KV, LV, miss_wait, KP=4, LP=2):
    """strat = (m_pri, m_exp, chase): children queued (in P order) when the priors land / at the expansion
    / per first visit of a sibling.  Returns dict."""
    m_pri, m_exp, chase = strat
    # build per-playout lists
    by_p = {}
    for xi, r in enumerate(m.events):
        by_p.setdefault(r.t_trig, []).append(("trig", xi))
        by_p.setdefault(r.t_exp, []).append(("exp", xi))
        for rank, t in enumerate(r.first):
            if t is not None:
                by_p.setdefault(t, []).append(("fresh", xi, rank))
    n_play = m.t
    wall = 0
    landed = {}   # (xi, rank) -> wall step the value is available
    queued = {}   # xi -> number of children queued so far (P order prefix)
    pri_at = {}   # xi -> wall step the priors land
    pending_pri = []  # (wall land, xi)
    hits = miss = 0
    steps_per_search = []
    search_start_wall = 0

    def vland(w):
        return (w // KV + 1) * KV + LV

    def queue_children(xi, upto, w):
        k = len(m.events[xi].order)
        q = queued.get(xi, 0)
        upto = min(upto, k)
        for rank in range(q, upto):
            landed[(xi, rank)] = vland(w)
        queued[xi] = max(q, upto)

    p = 0
    while p < n_play:
        if p % n_sims == 0 and p > 0:
            # search boundary: lockstep end, everything in flight lands (the queues are flushed)
            steps_per_search.append(wall - search_start_wall)
            for key in landed:
                landed[key] = min(landed[key], wall)
            for i, (w, xi) in enumerate(pending_pri):
                pending_pri[i] = (min(w, wall), xi)
            search_start_wall = wall
        # priors landing -> queue the first m_pri children
        for (w, xi) in list(pending_pri):
            if w <= wall:
                pending_pri.remove((w, xi))
                if m_pri:
                    queue_children(xi, m_pri, wall)
        extra = 0
        for ev in by_p.get(p, ()):
            if ev[0] == "exp":
                xi = ev[1]
                if m_exp:
                    queue_children(xi, max(queued.get(xi, 0), 1) + m_exp if chase else m_exp, wall)
        for ev in by_p.get(p, ()):
            if ev[0] == "fresh":
                xi, rank = ev[1], ev[2]
                t_l = landed.get((xi, rank))
                if t_l is not None and t_l <= wall:
                    hits += 1
                else:
                    miss += 1
                    w_batch = t_l if t_l is not None else 10 ** 9
                    extra = min(miss_wait, max(w_batch - wall, 0)) if miss_wait is not None else max(w_batch - wall, 0)
                    landed[(xi, rank)] = wall + extra
                    queued[xi] = max(queued.get(xi, 0), rank + 1)
                if chase:
                    queue_children(xi, max(queued.get(xi, 0), rank + 1) + chase, wall)
        for ev in by_p.get(p, ()):
            if ev[0] == "trig":
                xi = ev[1]
                pending_pri.append(((wall // KP + 1) * KP + LP, xi))
        wall += 1 + extra
        p += 1
    steps_per_search.append(wall - search_start_wall)
    waste = sum(1 for (xi, rank) in landed if m.events[xi].first[rank] is None)
    return dict(hits=hits, miss=miss, waste=waste, steps=steps_per_search)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=8)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--shipped", action="store_true")
    ap.add_argument("--cache", default="/tmp/sim_games_%s_%d_%d.pkl")
    args = ap.parse_args()
    path = args.cache % ("shipped" if args.shipped else "rand", args.games, args.sims)
    if os.path.exists(path):
        games = pickle.load(open(path, "rb"))
    else:
        torch.set_num_threads(4)
        torch.manual_seed(0)
        policy, value = network.SLPolicy().eval(), network.Value().eval()
        if args.shipped:
            g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
            policy.load_npz(os.path.join(g, "sl_model.npz"))
            value.load_npz(os.path.join(g, "value_model.npz"))
        rng = np.random.default_rng(0)
        games = []
        for s in range(args.games):
            m = play_logged(args.sims, policy, value, rng)
            g = S.Rec()
            g.events, g.t, g.fresh, g.root_fresh, g.n_policy_evals = m.events, m.t, m.fresh, m.root_fresh, m.n_policy_evals
            games.append(g)
        pickle.dump(games, open(path, "wb"))
    playouts = sum(g.t for g in games)
    fresh = sum(g.fresh for g in games)
    print("games %d playouts %d fresh/playout %.4f policy/playout %.4f" % (
        len(games), playouts, fresh / playouts, sum(g.n_policy_evals for g in games) / playouts))
    rs = np.random.default_rng(1)
    print("%-22s %5s %5s %6s %7s %7s %9s %9s" % ("strategy", "KV", "LV", "hit", "waste", "evals", "steps mean", "steps@1024"))
    for KV, LV in ((1, 2), (2, 2), (2, 3), (4, 3)):
        for name, strat in (("none", (0, 0, 0)), ("exp all", (0, 64, 0)), ("pri1 + exp all", (1, 64, 0)),
                            ("pri2 + exp all", (2, 64, 0)), ("pri2 + exp3 chase1", (2, 3, 1)),
                            ("pri2 + exp4 chase2", (2, 4, 2)), ("pri1 + exp3 chase2", (1, 3, 2)),
                            ("exp3 chase2", (0, 3, 2)), ("pri all", (64, 0, 0))):
            for miss_wait in (2,):
                tot = dict(hits=0, miss=0, waste=0)
                steps = []
                for g in games:
                    r = replay(g, args.sims, strat, KV, LV, miss_wait)
                    for k in tot:
                        tot[k] += r[k]
                    steps.extend(r["steps"])
                steps = np.array(steps)
                # lockstep end of every search over 1024 games: expected max of 1024 draws
                mx = np.mean([rs.choice(steps, 1024).max() for _ in range(200)])
                used = tot["hits"] + tot["miss"]
                print("%-22s %5d %5d %6.3f %7.3f %7.4f %9.1f %9.1f" % (
                    name, KV, LV, tot["hits"] / max(used, 1), tot["waste"] / max(used, 1),
                    (used + tot["waste"] + sum(g.root_fresh for g in games)) / playouts, steps.mean(), mx))


if __name__ == "__main__":
    main()
