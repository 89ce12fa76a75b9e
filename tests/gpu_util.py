"""Helpers shared by the -m gpu parity tests."""
import numpy as np

from oracle import oracle as orc


def positions_from_trace(trace):
    """Golden trace records -> (own, opp, legal, action, own', opp') from the
    mover's point of view, as numpy uint64 / int8 arrays."""
    p1, p2, color = trace[:, 0], trace[:, 1], trace[:, 2]
    own = np.where(color == 1, p1, p2)
    opp = np.where(color == 1, p2, p1)
    q1, q2 = trace[:, 5], trace[:, 6]
    own2 = np.where(color == 1, q1, q2)
    opp2 = np.where(color == 1, q2, q1)
    action = trace[:, 4].astype(np.uint8).view(np.int8)
    return own, opp, trace[:, 3], action, own2, opp2


def random_positions(n, seed):
    """n reachable positions (own, opp) from uniform-random oracle playouts."""
    rs = np.random.RandomState(seed)
    own, opp = [], []
    while len(own) < n:
        z, final, tr = orc.random_playout(orc.initial_state(), 1, seed=seed, game_id=len(own))
        s = orc.initial_state()
        color = 1
        cut = rs.randint(0, len(tr) + 1)
        for t, a in enumerate(tr):
            if t == cut:
                break
            orc.place_stone(s, a, color)
            color = 3 - color
        p1, p2 = orc.state_to_bits(s)
        own.append(p1 if color == 1 else p2)
        opp.append(p2 if color == 1 else p1)
    return np.array(own, np.uint64), np.array(opp, np.uint64)


def state_of(own, opp):
    """(own, opp) bits -> reference-style board with own = colour 1."""
    return orc.bits_to_state(int(own), int(opp))
