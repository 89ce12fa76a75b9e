"""Pure-Python restatement of the reference PV-MCTS (MCTS.py:10-154) on top of
the C oracle's rules.  TEST INFRASTRUCTURE ONLY (see othello_oracle.c).

Small cases only: this is the reference's recursive dict-tree algorithm, one
game, one playout at a time.  dtypes follow what the reference code yields
under numpy >= 2 (the interpreter the golden vectors were recorded with):
P and Q are float32, u and Q+u are float64.
"""
import math

import numpy as np

from . import oracle as orc


class Node(object):
    """MCTS.py:10-76."""

    def __init__(self, parent=None, prob=0):
        self.parent = parent
        self.children = {}  # insertion-ordered: ascending legal action (MCTS.py:34-36)
        self.n_visits = 0
        self.Q = 0  # python int like the reference: the dtype of Q follows the leaf values
        self.P = np.float32(np.float32(prob) + np.float32(0.1))  # MCTS.py:19
        self.u = self.P  # MCTS.py:18

    def is_leaf(self):  # MCTS.py:24-25
        return len(self.children) < 1

    def expand(self, action_probs):  # MCTS.py:27-37
        for action, prob in action_probs:
            if action not in self.children:
                self.children[action] = Node(self, prob)

    def U(self, c_puct):  # MCTS.py:48-49
        cp = np.float32(np.float32(c_puct) * self.P)
        return float(cp) * math.sqrt(self.parent.n_visits) / (0.01 + self.n_visits)

    def select(self, c_puct):  # MCTS.py:39-46, python max(): first maximum wins
        best, best_v = None, None
        for a, ch in self.children.items():
            ch.u = ch.U(c_puct)
            v = float(ch.Q) + ch.u  # MCTS.py:75-76
            if best is None or v > best_v:
                best, best_v = (a, ch), v
        return best

    def update(self, leaf_value):  # MCTS.py:51-63
        self.n_visits += 1
        # numpy>=2 scalar promotion: np.float32 leaf values (lmbda < 1) keep Q float32;
        # with lmbda == 1 the reference's leaf value is a python float and Q is float64
        self.Q += (leaf_value - self.Q) / self.n_visits

    def update_recursive(self, leaf_value):  # MCTS.py:68-72: same value, no sign flip
        node = self
        while node is not None:
            node.update(leaf_value)
            node = node.parent


class MCTS(object):
    """MCTS.py:78-154 with the wall-clock budget replaced by a sim count.

    policy_fn(planes(1,2,8,8)) -> 64 float32 probs; value_fn(planes) -> float32;
    rollout_fn(state, color) -> z in {-1,0,1}.
    """

    def __init__(self, policy_fn, value_fn, rollout_fn, lmbda=0.5, c_puct=1, n_thr=15):
        self.root = Node(None, 1.0)  # MCTS.py:81
        self.policy_fn, self.value_fn, self.rollout_fn = policy_fn, value_fn, rollout_fn
        self.lmbda, self.c_puct, self.n_thr = lmbda, c_puct, n_thr
        self.n_leaf_evals = 0
        self.n_policy_evals = 0
        self.max_path = 0  # (test diagnostic: nodes on the deepest playout's path, the root included)

    def playout(self, state, color, node):  # MCTS.py:105-133 minus the broken node.copy()
        c = color
        depth = 1
        while True:
            if node.is_leaf():
                if node.n_visits >= self.n_thr:  # MCTS.py:109
                    actions = orc.legal_actions(state, c)
                    if len(actions) < 1:
                        node.children[-1] = Node(node, 1)  # MCTS.py:112-114 pass
                    if len(actions) == 1:
                        node.children[actions[0]] = Node(node, 1)  # MCTS.py:115-117
                    else:
                        # MCTS.py:118-120; also runs for 0 legal moves (expands nothing)
                        prob = np.asarray(self.policy_fn(orc.make_state_var(state, c)),
                                          np.float32).reshape(64)
                        self.n_policy_evals += 1
                        node.expand([(a, prob[a]) for a in actions])
                    continue  # MCTS.py:121: recurse into the same, now non-leaf node
                x = orc.make_state_var(state, c)
                v = np.float32(self.value_fn(x)) if self.lmbda < 1 else 0  # MCTS.py:123
                z = self.rollout_fn(state, c) if self.lmbda > 0 else 0  # MCTS.py:124
                leaf_value = (1 - self.lmbda) * v + self.lmbda * z  # MCTS.py:125
                if isinstance(leaf_value, np.float32):
                    assert leaf_value == orc.leaf_value(self.lmbda, v, z)
                node.update_recursive(leaf_value)  # MCTS.py:127
                self.n_leaf_evals += 1
                self.max_path = max(self.max_path, depth)
                return leaf_value
            action, node = node.select(self.c_puct)  # MCTS.py:130
            state = orc.place_stone(state, action, c)  # MCTS.py:131 (-1 = pass: unchanged)
            c = 3 - c
            depth += 1

    def get_move(self, state, color, n_sims):  # MCTS.py:139-147
        for _ in range(n_sims):
            self.playout(np.array(state, dtype=np.float32), color, self.root)
        best, best_n = None, None
        for a, ch in self.root.children.items():  # first maximum wins
            if best is None or ch.n_visits > best_n:
                best, best_n = a, ch.n_visits
        return best

    def update_with_move(self, last_move):  # MCTS.py:149-154
        if last_move in self.root.children:
            self.root = self.root.children[last_move]
            self.root.parent = None
        else:
            self.root = Node(None, 1.0)


def dump_tree(node, depth=0, max_depth=6):
    d = dict(n=int(node.n_visits), Q=float(node.Q), P=float(node.P), children={},
             order=[int(a) for a in node.children.keys()])
    if depth < max_depth:
        for a, ch in node.children.items():
            d["children"][str(int(a))] = dump_tree(ch, depth + 1, max_depth)
    return d


# ---------------------------------------------------------------- drivers
def rl_game(policy1, policy2, uniforms, handicap=None):
    """src/rl_self_play.Game(model1, model2)() -- src/rl_self_play.py:10-31,111-145.
    policy(planes)->64 float32 probs; uniforms: iterator of the draws of
    np.random.choice.  Returns (states, actions, z, final_state)."""
    state = orc.initial_state(handicap)
    states, actions = [], []
    stone_num, pass_flg = 4, False
    it = iter(uniforms)
    while stone_num < 64:
        for color in (1, 2):
            acts = orc.legal_actions(state, color)
            if len(acts) > 0:
                prob = np.asarray((policy1 if color == 1 else policy2)(
                    orc.make_state_var(state, color)), np.float32).reshape(64)
                p = orc.masked_probs(prob, acts)
                a = orc.choice_cdf(p, next(it))
                if color == 1:  # src/rl_self_play.py:134-138: swapped board + action
                    states.append(state * (3 - state) * (3 - state) / 2)
                    actions.append(a)
                orc.place_stone(state, a, color)
                pass_flg = False
                stone_num += 1
            else:
                if pass_flg:
                    stone_num = 64
                pass_flg = True
    return states, actions, orc.judge(state, 1), state


class GameEnv(object):
    """rl_env.GameEnv reset/step/judge (rl_env.py:26-79,152-172); the opponent
    (colour 2) model is `opp(planes)->64 floats`, uniforms drive its sampling."""

    def __init__(self, opp, uniforms):
        self.opp = opp
        self.it = iter(uniforms)
        self.reset()

    def reset(self):
        self.state = orc.initial_state()
        self.stone_num, self.pass_flg = 4, False
        return orc.env_obs(self.state)

    def _opp_action(self, acts):  # rl_env.py:152-172: rejection sampling on out - min(out)
        while True:
            out = np.asarray(self.opp(orc.env_obs(self.state)), np.float32).reshape(64).copy()
            out -= out.min()
            a = orc.choice_cdf((out / np.sum(out)).astype(np.float64), next(self.it))
            if a in acts:
                return a

    def step(self, action):  # rl_env.py:41-74
        done = False
        acts = orc.legal_actions(self.state, 1)
        if len(acts) > 0:
            if action not in acts:
                raise ValueError("illegal agent action: the reference falls back to python's "
                                 "random.choice here (rl_env.py:46-48), which is not replayable")
            orc.place_stone(self.state, action, 1)
            self.stone_num += 1
            self.pass_flg = False
        else:
            if self.pass_flg:
                done = True
            self.pass_flg = True
        acts = orc.legal_actions(self.state, 2)
        if len(acts) > 0:
            orc.place_stone(self.state, self._opp_action(acts), 2)
            self.stone_num += 1
            self.pass_flg = False
        else:
            if self.pass_flg:
                done = True
            self.pass_flg = True
        if self.stone_num >= 64:
            done = True
        return orc.env_obs(self.state), 0, done, None

    def __call__(self):
        return orc.judge(self.state, 1)


def selfplay_game(mcts, n_sims, handicap=None):
    """One PV-MCTS self-play game in the turn structure of game.py:117-142,253-255
    with BOTH colours driven by MCTS.get_move (the reference drives colour 2
    only; colour 1 is a human or the SL policy): a side with a legal move
    searches, plays the most visited child and advances the tree
    (game.py:112-113); a side without one passes and advances the tree with -1
    (game.py:136-140).  Returns (moves per turn, z from colour 1's view, state)."""
    state = orc.initial_state(handicap)
    stone_num, pass_flg, moves = 4, False, []
    while stone_num < 64:
        for color in (1, 2):
            acts = orc.legal_actions(state, color)
            if len(acts) > 0:
                a = mcts.get_move(state, color, n_sims)
                mcts.update_with_move(a)
                orc.place_stone(state, a, color)
                stone_num += 1
                pass_flg = False
                moves.append(a)
            else:
                if pass_flg:
                    stone_num = 64
                pass_flg = True
                mcts.update_with_move(-1)
                moves.append(-1)
    return moves, orc.judge(state, 1), state


def value_self_play(policy0, policy1, stop_num, draws):
    """value_self_play.SelfPlay(stop_num)() -- value_self_play.py:32-59,133-161.
    policy(planes) -> 64 float32 numbers (whatever the net returns: the reference applies
    its own softmax on top, value_self_play.py:143,169-171); draws: iterator of the uniforms
    consumed, in order, by numpy.random.choice (inverse CDF) and random.choice
    (seq[floor(u * len)]).  Returns (own_bits, opp_bits of the recorded position with own =
    the side to move there, result from that side's view, final state)."""
    it = iter(draws)
    state = orc.initial_state()
    stone_num, pass_flg = 4, False

    def turn(color, policy):  # value_self_play.py:150-161
        nonlocal stone_num, pass_flg
        acts = orc.legal_actions(state, color)
        if len(acts) > 0:
            out = np.asarray(policy(orc.make_state_var(state, color)), np.float32).reshape(64)
            ex = np.exp(out)
            p = ex / np.sum(ex)                                   # softmax(), :169-171
            a = orc.choice_cdf(p.astype(np.float64), next(it))    # np.random.choice(64, p=...)
            if a not in acts:                                     # :145-148
                u = next(it)
                a = acts[min(int(u * len(acts)), len(acts) - 1)]
            orc.place_stone(state, a, color)
            pass_flg = False
            stone_num += 1
        else:
            if pass_flg:
                stone_num = 64
            pass_flg = True

    cl = 1
    while stone_num < stop_num:                                   # :34-36
        turn(cl, policy0)
        cl = 3 - cl
    color = cl
    p1, p2 = orc.state_to_bits(state)
    own, opp = (p1, p2) if color == 1 else (p2, p1)               # :38-43: the mover becomes "2"
    acts = orc.legal_actions(state, cl)
    if len(acts) == 0:                                            # :46-48
        return own, opp, -1, state
    u = next(it)
    orc.place_stone(state, acts[min(int(u * len(acts)), len(acts) - 1)], cl)  # :49-50
    pass_flg = False
    stone_num += 1
    cl = 3 - cl
    while stone_num < 64:                                         # :55-57
        turn(cl, policy1)
        cl = 3 - cl
    return own, opp, orc.judge(state, color), state
