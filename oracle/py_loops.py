"""Python-loop restatement of the reference's board code, for the CPU baseline.

TEST INFRASTRUCTURE ONLY (tests/ and bench.py's cpu_baseline leg): the product
never imports this.  oracle/othello_oracle.c is the fast checker; this module is
the *same algorithm in the reference's own execution model* -- interpreted Python
walking an (8,8) float32 numpy array cell by cell and direction by direction
(rl_env.py:88-138, mcts_self_play.py:25-29,100-134) -- so that bench.py can
report what the reference's code structure costs per core next to the C port
(BASELINE.md measured 73.9 games/s/core for the real
rl_env.py board loops in the survey container).  Checked against the golden
records of tests/golden/rules.npz (recorded from the reference) in
tests/test_oracle_golden.py.
"""
import numpy as np

DIRS = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1))  # rl_env.py:94-95


def _inside(y, x):
    return 0 <= y <= 7 and 0 <= x <= 7  # rl_env.py:82-83, negated


def _bracket_end(state, y, x, dy, dx, color):
    """From the neighbour (y+dy, x+dx): walk over the opponent's run (rl_env.py:97-104,
    123-131).  Returns the cell behind the run if it holds `color`, else None."""
    y, x = y + dy, x + dx
    if not _inside(y, x) or state[y, x] + color != 3:
        return None
    while state[y, x] + color == 3:
        y, x = y + dy, x + dx
        if not _inside(y, x):
            return None
    return (y, x) if state[y, x] == color else None


def legal_actions(state, color):
    """rl_env.py:114-138 / mcts_self_play.py:64-89: ascending action indices."""
    out = []
    for i in range(8):
        for j in range(8):
            if state[i, j] != 0:
                continue
            for dy, dx in DIRS:
                if _bracket_end(state, i, j, dy, dx, color) is not None:
                    out.append(i * 8 + j)
                    break
    return out


def place_stone(state, action, color):
    """rl_env.py:88-112 / mcts_self_play.py:36-62, in place; no legality check."""
    y, x = action // 8, action % 8
    state[y, x] = color
    for dy, dx in DIRS:
        end = _bracket_end(state, y, x, dy, dx, color)
        if end is None:
            continue
        ry, rx = end[0] - dy, end[1] - dx
        while state[ry, rx] + color == 3:  # rl_env.py:108-112: walk back, turning the run
            state[ry, rx] = color
            ry, rx = ry - dy, rx - dx
    return state


def judge(state, color):
    """mcts_self_play.py:113-121."""
    a, b = np.sum(state == color), np.sum(state == 3 - color)
    return 1 if a > b else (-1 if a < b else 0)


def make_state_var(state, color):
    """game.py:168-174: planes [opponent of the mover, mover]."""
    return np.stack([state == 3 - color, state == color]).astype(np.float32).reshape(1, 2, 8, 8)


def simulate(state, color, policy, rng):
    """Simulate(state)(color) (mcts_self_play.py:9-29,100-134): `policy(planes)` returns 64
    probabilities (the B = 1 RolloutPolicy call), rng.random_sample() the uniform of
    numpy.random.choice.  Returns (z from `color`'s view, board-steps played)."""
    state = np.array(state, dtype=np.float32, copy=True)
    stone_num = 64 - int(np.sum(state == 0))
    pass_flg = False
    steps = 0
    while stone_num < 64:
        for c in (color, 3 - color):
            steps += 1
            actions = legal_actions(state, c)
            if not actions:
                if pass_flg:
                    stone_num = 64
                pass_flg = True
                continue
            prob = np.asarray(policy(make_state_var(state, c))).reshape(64)
            valid = np.zeros(64)
            valid[actions] = 1
            p = prob * valid
            cdf = np.cumsum(p / np.sum(p))
            cdf /= cdf[-1]
            a = int(np.searchsorted(cdf, rng.random_sample(), side="right"))  # numpy.random.choice
            if a not in actions:
                a = actions[-1]
            place_stone(state, a, c)
            pass_flg = False
            stone_num += 1
    return judge(state, color), steps
