"""CPU simulation (lab tool, not product): how often would a value look-ahead hit?

Plays PV-MCTS self-play games with oracle/mcts_py.py (random-init or shipped nets on the CPU), records
for every expanded node X the time (global playout index of the game) its priors could be known
(trigger visit), its expansion, and the first visits of its children -- which are exactly the
playouts that end on a FRESH leaf (no stored value).  Then replays prefetch strategies with a landing
latency of L playouts and reports hit rate and wasted evaluations.

    python tools/sim_value_lookahead.py --games 8 --sims 100 [--shipped]
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mcts_py, oracle as orc  # noqa: E402
from iago_amd import network  # noqa: E402


class Rec(object):
    pass


class TracedMCTS(mcts_py.MCTS):
    """mcts_py.MCTS with a per-node value cache (the product's value cache: same trees) and event
    records.  time = playouts of this game so far."""

    def __init__(self, *a, trigger=10, **kw):
        super().__init__(*a, **kw)
        self.t = 0
        self.trigger = trigger
        self.events = []   # per expanded node: Rec(t_trig, t_exp, order=[actions by P desc], first=[t of first visit or None])
        self.fresh = 0
        self.root_fresh = 0
        self.search_starts = []

    def playout(self, state, color, node):
        c = color
        while True:
            if node.is_leaf():
                if node.n_visits >= self.n_thr:
                    actions = orc.legal_actions(state, c)
                    if len(actions) < 1:
                        node.children[-1] = mcts_py.Node(node, 1)
                    if len(actions) == 1:
                        node.children[actions[0]] = mcts_py.Node(node, 1)
                    else:
                        prob = np.asarray(self.policy_fn(orc.make_state_var(state, c)), np.float32).reshape(64)
                        self.n_policy_evals += 1
                        node.expand([(a, prob[a]) for a in actions])
                    r = Rec()
                    r.t_trig = getattr(node, "t_trig", self.t)
                    r.t_exp = self.t
                    r.n_root_trig = getattr(node, "n_root_trig", 1)
                    # P order: descending P, ties lowest action (children dict is ascending action)
                    acts = list(node.children.keys())
                    r.order = sorted(acts, key=lambda a: (-float(node.children[a].P), acts.index(a)))
                    r.first = [None] * len(acts)
                    r.visits_after = 0
                    node.rec = r
                    self.events.append(r)
                    continue
                if getattr(node, "v", None) is None:
                    node.v = np.float32(self.value_fn(orc.make_state_var(state, c)))
                    self.fresh += 1
                    par = node.parent
                    if par is not None and hasattr(par, "rec"):
                        a = [k for k, ch in par.children.items() if ch is node][0]
                        rank = par.rec.order.index(a)
                        par.rec.first[rank] = self.t
                        # the structural property: children are first-visited in P order
                        assert all(x is not None for x in par.rec.first[:rank]), "P-order violated"
                    else:
                        self.root_fresh += 1
                v = node.v
                z = self.rollout_fn(state, c)
                leaf_value = (1 - self.lmbda) * v + self.lmbda * z
                node.update_recursive(leaf_value)
                if node.n_visits == self.trigger and node.is_leaf():
                    node.t_trig = self.t
                    nr = node
                    while nr.parent is not None:
                        nr = nr.parent
                    node.n_root_trig = nr.n_visits
                self.t += 1
                self.n_leaf_evals += 1
                return leaf_value
            action, node = node.select(self.c_puct)
            state = orc.place_stone(state, action, c)
            c = 3 - c


def play(seed, n_sims, policy, value, rng):
    def pol(x):
        with torch.no_grad():
            return policy(torch.from_numpy(x)).numpy()

    def val(x):
        with torch.no_grad():
            return value(torch.from_numpy(x)).numpy()[0]

    def roll(state, c):
        return int(rng.integers(-1, 2))

    m = TracedMCTS(pol, val, roll)
    mcts_py.selfplay_game(m, n_sims)
    return m


def evaluate(games, n_sims, L, strategy, K=4):
    """strategy(rec) -> initial lead m (children queued when the priors land); +1 sibling per first
    visit.  A value queued at time t lands at the next group boundary after t (multiple of K inside a
    search; a search's end flushes everything) + L playouts... modelled as t_land = boundary(t) + L,
    capped at the search's end."""
    hits = miss = waste = total_exp = 0
    for m in games:
        for r in m.events:
            k = len(r.order)
            if k == 1 and r.order[0] == -1:
                pass
            total_exp += 1

            def land(t):
                b = (t // K + 1) * K
                end = (t // n_sims + 1) * n_sims
                return min(b + L, end)
            # the priors land like a value does (policy batch), then the values of the first m children
            t_pri = land(r.t_trig)
            m0 = min(k, strategy(r))
            landed = [None] * k
            for i in range(m0):
                landed[i] = land(t_pri)
            queued = m0
            for i in range(k):
                tv = r.first[i]
                if tv is None:
                    break
                if landed[i] is not None and landed[i] <= tv:
                    hits += 1
                else:
                    miss += 1
                # this first visit queues the next sibling(s)
                if queued < k:
                    if queued <= i:   # not even queued: the visit itself evaluates it; queue the next
                        queued = i + 1
                    if queued < k:
                        landed[queued] = land(max(tv, t_pri))
                        queued += 1
            for i in range(k):
                if landed[i] is not None and r.first[i] is None:
                    waste += 1
    return hits, miss, waste, total_exp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--shipped", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(4)
    torch.manual_seed(0)
    policy, value = network.SLPolicy().eval(), network.Value().eval()
    if args.shipped:
        g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
        policy.load_npz(os.path.join(g, "sl_model.npz"))
        value.load_npz(os.path.join(g, "value_model.npz"))
    rng = np.random.default_rng(0)
    games = [play(s, args.sims, policy, value, rng) for s in range(args.games)]
    playouts = sum(m.t for m in games)
    fresh = sum(m.fresh for m in games)
    rootf = sum(m.root_fresh for m in games)
    pol = sum(m.n_policy_evals for m in games)
    print("playouts %d fresh %d (%.3f) root-fresh %d policy evals %d expansions %d" % (
        playouts, fresh, fresh / playouts, rootf, pol, sum(len(m.events) for m in games)))
    used = [sum(1 for x in r.first if x is not None) for m in games for r in m.events]
    ks = [len(r.order) for m in games for r in m.events]
    print("children per expansion %.2f, first-visited %.2f" % (np.mean(ks), np.mean(used)))
    gaps = [r.t_exp - r.t_trig for m in games for r in m.events]
    print("trigger->expansion playouts: median %d, p10 %d, min %d" % (np.median(gaps), np.percentile(gaps, 10), min(gaps)))
    out = []
    for L in (2, 4, 8):
        for name, strat in (("m1", lambda r: 1), ("m2", lambda r: 2), ("m3", lambda r: 3), ("m4", lambda r: 4),
                            ("rate", lambda r: 1 + math.ceil((L + 4) * 10.0 / max(r.n_root_trig, 10))),
                            ("all", lambda r: 64)):
            h, ms, w, _ = evaluate(games, args.sims, L, strat)
            print("L=%d %-5s hit %.3f  inline per playout %.4f  wasted/used %.3f  evals/playout %.4f" % (
                L, name, h / max(h + ms, 1), (ms + rootf) / playouts, w / max(h + ms, 1), (h + ms + w) / playouts))
            out.append(dict(L=L, strategy=name, hits=h, misses=ms, wasted=w))
    print(json.dumps(dict(playouts=playouts, fresh=fresh, results=out)))


if __name__ == "__main__":
    main()


def interval_stats(games):
    import collections
    pre, post = [], []
    for m in games:
        for r in m.events:
            ts = [t for t in r.first if t is not None]
            if len(ts) >= 3:
                iv = np.diff(ts)
                pre.append((r.t_exp - r.t_trig) / 5.0)
                post.append(float(np.mean(iv)))
    pre, post = np.array(pre), np.array(post)
    print("expansions with >=3 first visits: %d; pre-interval median %.1f, post-interval median %.1f, corr %.2f" % (
        len(pre), np.median(pre), np.median(post), np.corrcoef(pre, post)[0, 1]))
    for lo, hi in ((0, 1.5), (1.5, 3), (3, 6), (6, 1e9)):
        sel = (pre >= lo) & (pre < hi)
        if sel.any():
            print("  pre in [%.1f,%.1f): n=%d post median %.1f p25 %.1f p75 %.1f" % (
                lo, hi, sel.sum(), np.median(post[sel]), np.percentile(post[sel], 25), np.percentile(post[sel], 75)))
