#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference code.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference (shionhonda/IaGo) is pure Python on top of Chainer/cupy/numba,
none of which is installed here.  Its board / search code is nevertheless plain
numpy + Python, so this script imports the reference modules *unmodified* from
/root/reference under small stub modules for `chainer` and `numba`
 and records input/output pairs of the reference's own
functions.  Only data is written: no reference source text leaves
/root/reference.

Two reference defects have to be worked around to execute the search code; both
are applied by monkey-patching at run time and are documented in DESIGN.md:
  * MCTS.playout calls `node.copy()` (MCTS.py:106) but Node defines no copy;
    Node.copy is patched to return the node itself (a copy would make every
    playout a no-op on the real tree).
  * mcts_self_play.Simulate / MCTS.MCTS load npz weights through Chainer in
    __init__; instances are created with __new__ and given stand-in callables
    with the same call signature (`model(x).data`).
numpy.random.choice is replaced by a recorder with numpy's own algorithm
(cdf = cumsum(p)/sum; searchsorted(u, 'right')) so that the uniforms that drove
each game are part of the fixture; the script asserts that this restatement is
identical to the real numpy.random.RandomState.choice before using it.
"""
import contextlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ stubs
def install_stubs():
    ch = types.ModuleType("chainer")

    class Chain(object):
        def __init__(self, *a, **k):
            pass

        @contextlib.contextmanager
        def init_scope(self):
            yield

    class _Cfg(object):
        train = False
        enable_backprop = False

    @contextlib.contextmanager
    def using_config(name, value):
        yield

    def _noop_link(*a, **k):
        return lambda x: x

    ch.Chain = Chain
    ch.Variable = lambda x: x
    ch.config = _Cfg()
    ch.using_config = using_config
    links = types.ModuleType("chainer.links")
    links.Convolution2D = _noop_link
    links.Bias = _noop_link
    links.Linear = _noop_link
    functions = types.ModuleType("chainer.functions")
    serializers = types.ModuleType("chainer.serializers")
    serializers.load_npz = lambda *a, **k: None
    serializers.save_npz = lambda *a, **k: None
    optimizers = types.ModuleType("chainer.optimizers")
    cuda = types.ModuleType("chainer.cuda")
    cuda.cupy = np
    ch.links, ch.functions, ch.serializers, ch.optimizers, ch.cuda = (
        links, functions, serializers, optimizers, cuda)
    ch.Variable = lambda x: x
    for name, mod in [("chainer", ch), ("chainer.links", links), ("chainer.functions", functions),
                      ("chainer.serializers", serializers), ("chainer.optimizers", optimizers),
                      ("chainer.cuda", cuda)]:
        sys.modules[name] = mod
    nb = types.ModuleType("numba")
    nb.jit = lambda f=None, *a, **k: f if callable(f) else (lambda g: g)
    sys.modules["numba"] = nb


install_stubs()
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "src"))
import MCTS as ref_mcts  # noqa: E402  (must precede `game`: circular import)
import game as ref_game  # noqa: E402
import mcts_self_play as ref_sim  # noqa: E402
import rl_env as ref_env  # noqa: E402
import rl_self_play as ref_rl  # noqa: E402
import load as ref_load  # noqa: E402  (pure numpy; main() is not run)

gf = ref_game.GameFunctions


# ---------------------------------------------------------------- helpers
def to_bits(state):
    s = np.asarray(state).reshape(64)
    p1 = sum(1 << a for a in range(64) if s[a] == 1)
    p2 = sum(1 << a for a in range(64) if s[a] == 2)
    return p1, p2


def mask_of(actions):
    return sum(1 << a for a in actions)


def new_env():
    env = ref_env.GameEnv.__new__(ref_env.GameEnv)
    env.model1 = env.model2 = None
    env.reset()
    return env


class Recorder(object):
    """Stand-in for numpy.random.choice with numpy's algorithm; records u."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.us = []

    def __call__(self, n, p=None):
        u = self.rs.random_sample()
        self.us.append(u)
        cdf = np.asarray(p, dtype=np.float64).cumsum()
        cdf /= cdf[-1]
        return int(cdf.searchsorted(u, side="right"))


def check_choice_restatement():
    rs = np.random.RandomState(123)
    for t in range(2000):
        k = rs.randint(1, 20)
        p = np.zeros(64)
        idx = rs.choice(64, size=k, replace=False)
        p[idx] = rs.random_sample(k) + 1e-3
        p /= p.sum()
        seed = int(rs.randint(0, 2 ** 31 - 1))
        want = np.random.RandomState(seed).choice(64, p=p)
        rec = Recorder(seed)
        got = rec(64, p=p)
        assert want == got, (t, want, got)


class _Out(object):
    def __init__(self, data):
        self.data = data


class FakeRollout(object):
    """RolloutPolicy stand-in (network.py:49-64 semantics) in float32 numpy."""

    def __init__(self, w, b):
        self.w = np.asarray(w, np.float32).reshape(2, 3, 3)
        self.b = np.asarray(b, np.float32).reshape(64)

    def __call__(self, x):
        x = np.asarray(x, np.float32).reshape(2, 8, 8)
        h = np.zeros(64, np.float32)
        for i in range(8):
            for j in range(8):
                acc = np.float32(0)
                for c in range(2):
                    for ky in range(3):
                        for kx in range(3):
                            y, xx = i + ky - 1, j + kx - 1
                            if 0 <= y < 8 and 0 <= xx < 8:
                                acc = np.float32(acc + self.w[c, ky, kx] * x[c, y, xx])
                h[i * 8 + j] = np.float32(acc + self.b[i * 8 + j])
        e = np.exp(h - h.max())
        return _Out((e / e.sum()).astype(np.float32).reshape(1, 64))


def hash_probs(x, salt):
    """Deterministic stand-in policy: dyadic-rational probabilities that depend
    only on the (2,8,8) planes, exactly representable in float32."""
    bits = np.asarray(x, np.float32).reshape(128)
    h = np.uint64(1469598103934665603 + salt)
    for k in np.nonzero(bits)[0]:
        h = np.uint64((int(h) ^ int(k + 1)) * 1099511628211 % (1 << 64))
    vals = np.empty(64, np.float32)
    for a in range(64):
        h = np.uint64((int(h) * 6364136223846793005 + 1442695040888963407) % (1 << 64))
        vals[a] = np.float32(((int(h) >> 40) & 0x3FF) + 1)
    return vals, int(h)


class FakePolicy(object):
    def __init__(self, salt):
        self.salt = salt

    def __call__(self, x):
        vals, _ = hash_probs(x, self.salt)
        return _Out((vals / np.float32(65536.0)).astype(np.float32).reshape(1, 64))


class FakeValue(object):
    def __init__(self, salt):
        self.salt = salt

    def __call__(self, x):
        _, h = hash_probs(x, self.salt)
        v = np.float32((((h >> 20) & 0x7FF) - 1024) / 1024.0)
        return _Out(np.array([v], np.float32))


# -------------------------------------------------------------- 1. traces
def rules_traces():
    """Uniform-random games on the reference env rules (rl_env.valid_pos /
    place_stone), cross-checked against game.GameFunctions at every step."""
    recs = []  # p1,p2,color,legal,action,p1',p2'
    games = []  # z (judge), n_turns, first record index, handicap id
    handicaps = [None, (2, 4), (3, 5), (4, 2), (5, 3)]
    rs = np.random.RandomState(2024)
    for g in range(100):
        env = new_env()
        hc = 0 if g < 50 else 1 + (g % 4)
        if hc:
            env.state[handicaps[hc][0], handicaps[hc][1]] = 2  # src/train_rl.py:43-46
        first = len(recs)
        pass_flg, done, color, turns = False, False, 1, 0
        while not done:
            pos = env.valid_pos(color)
            acts = [(p[0] - 1) * 8 + (p[1] - 1) for p in pos]
            assert acts == gf.legal_actions(env.state, color)
            assert acts == sorted(acts)
            p1, p2 = to_bits(env.state)
            if acts:
                a = acts[rs.randint(len(acts))]
                st2 = gf.place_stone(env.state.copy(), a, color)
                env.place_stone([a // 8 + 1, a % 8 + 1], color)
                assert np.array_equal(st2, env.state)
                pass_flg = False
            else:
                a = -1
                if pass_flg:
                    done = True
                pass_flg = True
            q1, q2 = to_bits(env.state)
            recs.append((p1, p2, color, mask_of(acts), a, q1, q2))
            color = 3 - color
            turns += 1
            if np.sum(env.state == 0) == 0:
                done = True
        games.append((env.judge(), turns, first, hc))
    recs = np.array([[r[0], r[1], r[2], r[3], r[4] & 0xFF, r[5], r[6]] for r in recs],
                    dtype=np.uint64)
    return recs, np.array(games, dtype=np.int64)


# ------------------------------------------------------- 2./3. edge cases
def parse(rows):
    m = {".": 0, "X": 1, "O": 2}
    return np.array([[m[c] for c in r] for r in rows], dtype=np.float32)


EDGE_BOARDS = {
    # runs that reach every edge; colour 1 brackets along rows / columns / diagonals
    "edge_runs": ["XOOOOOO.", "O......O", "O......O", "O......O", "O......O", "O......O", "O......X",
                  ".OOOOOOX"],
    # A<->H file wrap trap: a stone on H3 and its "neighbour" A4 must not connect
    "wrap_trap": ["........", "........", ".......O", "X.......", "OX......", "........", "......XO",
                  "X......."],
    "wrap_trap2": ["......XO", "O......X", "X.......", "........", ".......X", "O.......", "XO......",
                   "........"],
    # all four corners capturable along diagonals
    "corners": [".......O", ".O....O.", "..X..X..", "........", "........", "..X..X..", ".O....O.",
                "........"],
    "corners2": [".OOOOOO.", "OOOOOOOO", "OOXOOXOO", "OOOOOOOO", "OOOOOOOO", "OOXOOXOO", "OOOOOOOO",
                 ".OOOOOO."],
    # nobody can move (double pass terminal with empties)
    "dead": ["XXXXXXXX", "XXXXXXXX", "XXXXXXXX", "XXXXXXX.", "XXXXXX..", "XXXXXX.O", "XXXXXXX.",
             "XXXXXXXX"],
    # colour 1 must pass, colour 2 can move
    "pass1": ["OOOOOOOO", "OOOOOOOO", "OOOOOOOO", "OOOOOOOO", "OOOOOOOO", "OOOOOOOO", "OOOOOOXX",
              "OOOOOOX."],
    "pass1b": [".XOOOOOO", "........", "........", "........", "........", "........", "........",
               "........"],
    "full": ["XOXOXOXO", "OXOXOXOX", "XOXOXOXO", "OXOXOXOX", "XOXOXOXO", "OXOXOXOX", "XOXOXOXO",
             "OXOXOXOX"],
    "empty": ["........"] * 8,
    # one move flipping in all 8 directions
    "star": ["X..X..X.", ".O.O.O..", "..OOO...", "XOO.OOOX", "..OOO...", ".O.O.O..", "X..X..X.",
             "...X...."],
    # long runs with a gap / own stone inside
    "gaps": ["XOO.OOX.", "XOOXOO..", ".OOOOOOX", "X.OOOOO.", "........", "O.XXXXX.", "OXXXXXX.",
             ".XXXXXXO"],
    "start": ["........", "........", "........", "...OX...", "...XO...", "........", "........",
              "........"],
}


def rules_edge():
    names, boards, legal, order = [], [], [], {}
    place = []  # (board idx, color, action, p1', p2')
    for name, rows in EDGE_BOARDS.items():
        st = parse(rows)
        names.append(name)
        p1, p2 = to_bits(st)
        boards.append((p1, p2))
        row = []
        for color in (1, 2):
            env = new_env()
            env.state = st.copy()
            acts = [(p[0] - 1) * 8 + (p[1] - 1) for p in env.valid_pos(color)]
            assert acts == gf.legal_actions(st.copy(), color)
            order["%s/%d" % (name, color)] = acts
            row.append(mask_of(acts))
            # every cell as a target, legal or not: the reference never checks
            # legality in place_stone (occupied targets are overwritten too)
            for a in list(range(64)) + [-1]:
                s2 = gf.place_stone(st.copy(), a, color)
                if a >= 0:
                    env2 = new_env()
                    env2.state = st.copy()
                    env2.place_stone([a // 8 + 1, a % 8 + 1], color)
                    assert np.array_equal(env2.state, s2)
                q1, q2 = to_bits(s2)
                place.append((len(names) - 1, color, a & 0xFF, q1, q2))
        legal.append(row)
    return (names, np.array(boards, np.uint64), np.array(legal, np.uint64),
            np.array(place, np.uint64), order)


# ------------------------------------------------------------- 4. planes
def planes(recs):
    idx = np.linspace(0, len(recs) - 1, 24).astype(int)
    outs = []
    for i in idx:
        p1, p2, color = int(recs[i, 0]), int(recs[i, 1]), int(recs[i, 2])
        st = np.zeros(64, np.float32)
        for a in range(64):
            st[a] = 1 if (p1 >> a) & 1 else (2 if (p2 >> a) & 1 else 0)
        st = st.reshape(8, 8)
        v1 = gf.make_state_var(st.copy(), 1)
        v2 = gf.make_state_var(st.copy(), 2)
        sim = ref_sim.Simulate.__new__(ref_sim.Simulate)
        assert np.array_equal(sim.make_state_var(st.copy(), color), v1 if color == 1 else v2)
        env = new_env()
        env.state = st.copy()
        obs = np.stack([st == 1, st == 2], axis=0).astype(np.float32).reshape(1, 2, 8, 8)
        outs.append(np.concatenate([v1, v2, obs], axis=0))
    return idx, np.stack(outs)  # (24, 3, 2, 8, 8)


# --------------------------------------------- 5. Simulate (leaf rollout)
def simulate_traces():
    rs = np.random.RandomState(7)
    w = rs.randn(18).astype(np.float32)
    b = (0.5 * rs.randn(64)).astype(np.float32)
    ref_w = np.load(os.path.join(REF, "models", "rollout_model.npz"))
    shipped_w = ref_w["conv1/W"].reshape(18).astype(np.float32)
    shipped_b = ref_w["bias2/b"].reshape(64).astype(np.float32)
    cases = []
    real_choice = np.random.choice
    try:
        for g in range(40):
            ww, bb = (w, b) if g % 2 == 0 else (shipped_w, shipped_b)
            # leaf position: play k uniform-random plies from the start first
            env = new_env()
            color, k = 1, int(rs.randint(0, 50))
            for _ in range(k):
                acts = gf.legal_actions(env.state, color)
                if acts:
                    gf.place_stone(env.state, acts[rs.randint(len(acts))], color)
                color = 3 - color
            start = env.state.copy()
            rec = Recorder(1000 + g)
            np.random.choice = rec
            sim = ref_sim.Simulate.__new__(ref_sim.Simulate)
            sim.state = start.copy()
            sim.stone_num = 64 - np.sum(sim.state == 0)
            sim.pass_flg = False
            sim.model = FakeRollout(ww, bb)
            # record per-turn actions by wrapping place_stone / turn
            trace = []
            orig_turn = sim.turn

            def turn(c, _sim=sim, _orig=orig_turn, _tr=trace):
                before = _sim.state.copy()
                _orig(c)
                diff = np.argwhere((before == 0) & (_sim.state != 0))
                _tr.append(int(diff[0][0] * 8 + diff[0][1]) if len(diff) else -1)

            sim.turn = turn
            z = sim(color)
            np.random.choice = real_choice
            # uniforms aligned per turn (a pass consumes none)
            us = np.zeros(140, np.float64)
            it = iter(rec.us)
            for t, a in enumerate(trace):
                if a >= 0:
                    us[t] = next(it)
            p1, p2 = to_bits(start)
            q1, q2 = to_bits(sim.state)
            cases.append(dict(weights=g % 2, p1=p1, p2=p2, color=color, z=int(z), q1=q1, q2=q2,
                              trace=trace, uniforms=us[:len(trace)].tolist()))
    finally:
        np.random.choice = real_choice
    return dict(w=w.tolist(), b=b.tolist(), shipped_w=shipped_w.tolist(),
                shipped_b=shipped_b.tolist(), cases=cases)


# ---------------------------------- 5b. rl_self_play.Game (policy-vs-policy)
def rl_game_traces():
    rs = np.random.RandomState(11)
    cases = []
    real_choice = np.random.choice
    handicaps = [None, (2, 4), (3, 5), (4, 2), (5, 3)]
    try:
        for g in range(12):
            w1, b1 = rs.randn(18).astype(np.float32), (0.5 * rs.randn(64)).astype(np.float32)
            w2, b2 = rs.randn(18).astype(np.float32), (0.5 * rs.randn(64)).astype(np.float32)
            rec = Recorder(5000 + g)
            np.random.choice = rec
            game = ref_rl.Game(FakeRollout(w1, b1), FakeRollout(w2, b2))
            hc = g % 5
            if hc:
                game.state[handicaps[hc][0], handicaps[hc][1]] = 2
            states, actions, z = game()
            np.random.choice = real_choice
            q1, q2 = to_bits(game.state)
            cases.append(dict(w1=w1.tolist(), b1=b1.tolist(), w2=w2.tolist(), b2=b2.tolist(),
                              handicap=hc, uniforms=[float(u) for u in rec.us],
                              states=[list(map(int, to_bits(s))) for s in states],
                              actions=[int(a) for a in actions], z=int(z), q1=q1, q2=q2))
    finally:
        np.random.choice = real_choice
    return cases


# ------------------------------------------------------- 5c. GameEnv.step
class _Pred(object):
    def __init__(self, m):
        self.predictor = m


class FakeEnvOpponent(object):
    """Stand-in for GameEnv.model2.predictor.  rl_env.get_position
    (rl_env.py:152-172) rejection-samples from `out - min(out)` until the draw
    is legal, recursing on every miss, so a stand-in with little mass on legal
    cells overflows the stack (a reference hazard, rl_env.py:8).  This one adds
    mass on colour 2's legal cells (planes are [state==1, state==2], no swap) so
    that misses are rare but still occur."""

    def __init__(self, w, b):
        self.base = FakeRollout(w, b)

    def __call__(self, x):
        x = np.asarray(x, np.float32).reshape(2, 8, 8)
        st = (x[0] + 2 * x[1]).astype(np.float32)
        out = self.base(x).data.reshape(64).copy()
        for a in gf.legal_actions(st, 2):
            out[a] += np.float32(0.25)
        return _Out(out.reshape(1, 64))


def env_traces():
    """rl_env.GameEnv.reset/step (rl_env.py:26-74) with a stand-in opponent.
    random.choice (illegal agent action fallback, rl_env.py:46-48) is seeded."""
    import random as pyrandom
    rs = np.random.RandomState(17)
    cases = []
    real_choice = np.random.choice
    try:
        for g in range(6):
            w2, b2 = rs.randn(18).astype(np.float32), (0.5 * rs.randn(64)).astype(np.float32)
            rec = Recorder(7000 + g)
            np.random.choice = rec
            pyrandom.seed(g)
            env = ref_env.GameEnv.__new__(ref_env.GameEnv)
            env.model1 = None
            env.model2 = _Pred(FakeEnvOpponent(w2, b2))
            obs = env.reset()
            steps = []
            done = False
            while not done:
                pos = env.valid_pos(1)
                acts = [(p[0] - 1) * 8 + (p[1] - 1) for p in pos]
                # legal agent actions only (the illegal fallback draws from
                # python's `random`, which no fixture can replay)
                a = acts[rs.randint(len(acts))] if acts else int(rs.randint(64))
                obs, r, done, info = env.step(a)
                p1, p2 = to_bits(env.state)
                steps.append(dict(action=int(a), p1=p1, p2=p2, done=bool(done), reward=int(r),
                                  stone_num=int(env.stone_num), pass_flg=bool(env.pass_flg),
                                  obs_ok=bool(np.array_equal(
                                      np.asarray(obs),
                                      np.stack([env.state == 1, env.state == 2], 0)
                                      .astype(np.float32).reshape(1, 2, 8, 8)))))
            np.random.choice = real_choice
            cases.append(dict(w2=w2.tolist(), b2=b2.tolist(), uniforms=[float(u) for u in rec.us],
                              steps=steps, z=int(env())))
    finally:
        np.random.choice = real_choice
    return cases


# ---------------------------------------------------------- 6. node math
def node_math():
    rs = np.random.RandomState(3)
    out = []
    for t in range(20):
        k = int(rs.randint(1, 12))
        acts = sorted(rs.choice(64, size=k, replace=False).tolist())
        priors = (rs.randint(1, 1024, size=k) / 1024.0).astype(np.float32)
        if t % 4 == 0:
            priors[:] = priors[0]  # exact ties -> first-wins
        root = ref_mcts.Node(None, 1.0)
        root.expand([(a, priors[i]) for i, a in enumerate(acts)])
        c_puct = 1 if t % 2 == 0 else 2.5
        steps = []
        for s in range(40):
            # a visit: root then the selected child get the same leaf value
            lv = np.float32(rs.randint(-64, 65) / 64.0)
            if root.n_visits == 0:
                # mirror MCTS: first visits land on the root itself
                root.update(lv, c_puct)
                steps.append(dict(action=None, lv=float(lv)))
                continue
            a, node = root.select(c_puct)
            us = [float(root.children[x].u) for x in acts]
            vals = [float(root.children[x].get_value()) for x in acts]
            node.update_recursive(lv, c_puct)
            steps.append(dict(action=int(a), lv=float(lv), u=us, value=vals))
        out.append(dict(actions=acts, priors=[float(p) for p in priors], c_puct=c_puct, steps=steps,
                        final_n=[int(root.children[x].n_visits) for x in acts],
                        final_Q=[float(root.children[x].Q) for x in acts],
                        final_P=[float(root.children[x].P) for x in acts],
                        root_n=int(root.n_visits), root_Q=float(root.Q),
                        best=int(max(root.children.items(), key=lambda an: an[1].n_visits)[0])))
    return out


# ------------------------------------------------- 6b. full MCTS playouts
def dump_tree(node, depth=0, max_depth=6):
    d = dict(n=int(node.n_visits), Q=float(node.Q), P=float(node.P), children={})
    if depth < max_depth:
        for a, ch in node.children.items():
            d["children"][str(int(a))] = dump_tree(ch, depth + 1, max_depth)
    d["order"] = [int(a) for a in node.children.keys()]
    return d


def mcts_traces():
    """Real MCTS.playout / get_move logic (MCTS.py:105-147) with stand-in nets
    and a sim-count budget instead of the 10 s wall clock."""
    ref_mcts.Node.copy = lambda self: self  # see module docstring
    rs = np.random.RandomState(5)
    w = rs.randn(18).astype(np.float32)
    b = (0.5 * rs.randn(64)).astype(np.float32)
    cases = []
    real_choice = np.random.choice
    real_sim_cls = ref_sim.Simulate
    try:
        for g in range(6):
            n_thr = [15, 15, 1, 4, 15, 2][g]
            lmbda = [0.5, 0.5, 0.5, 0.0, 1.0, 0.25][g]
            c_puct = [1, 1, 1, 2.5, 1, 1][g]
            n_sims = [60, 100, 40, 60, 50, 80][g]
            env = new_env()
            color = 1
            for _ in range(int(rs.randint(0, 40))):
                acts = gf.legal_actions(env.state, color)
                if acts:
                    gf.place_stone(env.state, acts[rs.randint(len(acts))], color)
                color = 3 - color
            if g == 4:
                env.state = parse(EDGE_BOARDS["pass1"])
                color = 1
            start = env.state.copy()
            m = ref_mcts.MCTS.__new__(ref_mcts.MCTS)
            m.root = ref_mcts.Node(None, 1.0)
            m.policy_net = FakePolicy(g)
            m.value_net = FakeValue(g)
            m.lmbda, m.c_puct, m.n_thr, m.time_limit = lmbda, c_puct, n_thr, 1e9
            rec = Recorder(9000 + g)
            np.random.choice = rec
            zs = []

            def make_sim(state, _w=w, _b=b):
                sim = real_sim_cls.__new__(real_sim_cls)
                sim.state = state.copy()
                sim.stone_num = 64 - np.sum(sim.state == 0)
                sim.pass_flg = False
                sim.model = FakeRollout(_w, _b)
                return sim

            def evaluate_rollout(state, c, _zs=zs):
                z = make_sim(state)(c)
                _zs.append(int(z))
                return z

            m.evaluate_rollout = evaluate_rollout
            for s in range(n_sims):
                m.playout(start.copy(), color, m.root)
            np.random.choice = real_choice
            move = max(m.root.children.items(), key=lambda an: an[1].n_visits)[0] \
                if m.root.children else None
            p1, p2 = to_bits(start)
            cases.append(dict(p1=p1, p2=p2, color=color, n_thr=n_thr, lmbda=lmbda, c_puct=c_puct,
                              n_sims=n_sims, salt=g, zs=zs, move=None if move is None else int(move),
                              tree=dump_tree(m.root)))
    finally:
        np.random.choice = real_choice
    return dict(w=w.tolist(), b=b.tolist(), cases=cases)


# ------------------------------------------------- 6b. game front-end
class ScriptedMCTS(object):
    """Stand-in for MCTS.MCTS behind game.Game: get_move picks a legal move from a hash
    of the position, every call is logged (the front-end's call pattern is the fixture)."""

    def __init__(self):
        self.calls = []

    def get_move(self, state, color):
        acts = gf.legal_actions(state, color)
        p1, p2 = to_bits(state)
        a = acts[(p1 * 31 + p2 * 17 + color) % len(acts)]
        self.calls.append(["get_move", p1, p2, int(color), int(a)])
        return a

    def update_with_move(self, move):
        self.calls.append(["update_with_move", int(move)])


def frontend_traces():
    """game.Game in auto mode (game.py:13-150, 236-262): ASCII board, prompts, gamelog
    text and the MCTS call pattern, with stand-ins for the Chainer model and MCTS.MCTS
    (Game.__init__ loads npz files through Chainer) and the recorded np.random.choice."""
    import io
    cases = []
    real_choice = np.random.choice
    try:
        for g, start in enumerate([None, "pass1"]):
            game = ref_game.Game.__new__(ref_game.Game)
            game.p1, game.p2 = "IaGo(SLPolicy)", "IaGo(PV-MCTS)"   # game.py:19,25
            game.model = FakePolicy(40 + g)
            game.state = np.zeros([8, 8], dtype=np.float32)
            game.state[4, 3] = game.state[3, 4] = 1
            game.state[3, 3] = game.state[4, 4] = 2
            if start is not None:
                game.state = parse(EDGE_BOARDS[start])
            game.stone_num = int(np.sum(game.state != 0))
            game.play_num = 1
            game.pass_flg = False
            game.date = "2000-01-0%d-00-00" % (g + 1)
            game.gamelog = "IaGo \n" + game.date + "\n"
            game.mcts = ScriptedMCTS()
            p1, p2 = to_bits(game.state)
            rec = Recorder(7700 + g)
            np.random.choice = rec
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                game.show()
                while game.stone_num < 64:       # game.py:249-251
                    game.turn(1, True)
                    game.turn(2, True)
                jd = game.judge()
                print(jd)
                game.gamelog += jd + "\n"
            np.random.choice = real_choice
            f1, f2 = to_bits(game.state)
            cases.append(dict(p1=p1, p2=p2, salt=40 + g, date=game.date, us=rec.us, stdout=buf.getvalue(),
                              gamelog=game.gamelog, calls=game.mcts.calls, final=[f1, f2],
                              play_num=int(game.play_num)))
    finally:
        np.random.choice = real_choice
    return dict(cases=cases)


# ----------------------------------------------------------- 7. sampling
def sampling():
    rs = np.random.RandomState(99)
    ps, us, idx = [], [], []
    for t in range(300):
        k = int(rs.randint(1, 25))
        p = np.zeros(64)
        cells = rs.choice(64, size=k, replace=False)
        p[cells] = rs.random_sample(k) ** 3 + 1e-6
        p /= p.sum()
        seed = int(rs.randint(0, 2 ** 31 - 1))
        u = np.random.RandomState(seed).random_sample()
        i = int(np.random.RandomState(seed).choice(64, p=p))
        ps.append(p)
        us.append(u)
        idx.append(i)
    return np.array(ps), np.array(us), np.array(idx, np.int64)


# ------------------------------------------------------ 8. augmentation
def augment(recs):
    """The 8-fold dihedral augmentation of load.py:56-71 on sample positions,
    with the reference's own action maps load.rotate / load.transpose and the
    numpy calls of its main()."""
    idx = np.linspace(0, len(recs) - 1, 40).astype(int)
    states = np.zeros((len(idx), 8, 8))
    actions = np.zeros(len(idx))
    for k, i in enumerate(idx):
        p1, p2, legal = int(recs[i, 0]), int(recs[i, 1]), int(recs[i, 3])
        for a in range(64):
            states[k, a // 8, a % 8] = 1 if (p1 >> a) & 1 else (2 if (p2 >> a) & 1 else 0)
        acts = [a for a in range(64) if (legal >> a) & 1]
        actions[k] = acts[k % len(acts)] if acts else 27
    S, A = states, actions
    for i in range(3):                                   # load.py:58-63
        states = np.rot90(states, k=1, axes=(1, 2))
        S = np.concatenate([S, states], axis=0)
        actions = ref_load.rotate(actions)
        A = np.concatenate([A, actions], axis=0)
    states = states.transpose(0, 2, 1)                   # load.py:64-68
    S = np.concatenate([S, states], axis=0)
    actions = ref_load.transpose(actions)
    A = np.concatenate([A, actions], axis=0)
    for i in range(3):                                   # load.py:69-74
        states = np.rot90(states, k=1, axes=(1, 2))
        S = np.concatenate([S, states], axis=0)
        actions = ref_load.rotate(actions)
        A = np.concatenate([A, actions], axis=0)
    bits = np.array([to_bits(st) for st in S], dtype=np.uint64)
    return bits.reshape(8, len(idx), 2), A.reshape(8, len(idx)).astype(np.int64)


def main():
    check_choice_restatement()
    recs, games = rules_traces()
    names, boards, legal, place, order = rules_edge()
    pidx, pl = planes(recs)
    ps, us, idx = sampling()
    aug_bits, aug_act = augment(recs)
    np.savez_compressed(os.path.join(OUT, "rules.npz"), trace=recs, games=games,
                        edge_boards=boards, edge_legal=legal, edge_place=place,
                        planes_idx=pidx, planes=pl, samp_p=ps, samp_u=us, samp_idx=idx,
                        aug_bits=aug_bits, aug_act=aug_act)
    with open(os.path.join(OUT, "order.json"), "w") as f:
        json.dump(dict(names=names, order=order), f)
    with open(os.path.join(OUT, "simulate.json"), "w") as f:
        json.dump(simulate_traces(), f)
    with open(os.path.join(OUT, "rl_game.json"), "w") as f:
        json.dump(rl_game_traces(), f)
    with open(os.path.join(OUT, "env.json"), "w") as f:
        json.dump(env_traces(), f)
    with open(os.path.join(OUT, "node_math.json"), "w") as f:
        json.dump(node_math(), f)
    with open(os.path.join(OUT, "mcts.json"), "w") as f:
        json.dump(mcts_traces(), f)
    with open(os.path.join(OUT, "frontend.json"), "w") as f:
        json.dump(frontend_traces(), f)
    print("records:", len(recs), "games:", len(games), "edge boards:", len(names))
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
