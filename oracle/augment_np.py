"""numpy restatement of the reference's 8-fold dihedral data augmentation
(load.py:12-22,56-74).  TEST INFRASTRUCTURE ONLY.

Order of the 8 variants: identity; three successive np.rot90 (counter-
clockwise); the transpose of the third rotation; three more rotations of
that.  Actions follow load.rotate ((y,x) -> (7-x, y)) and load.transpose
((y,x) -> (x, y)).
"""
import numpy as np


def rotate(action):  # load.py:12-16
    y, x = action // 8, action % 8
    return (7 - x) * 8 + y


def transpose(action):  # load.py:18-22
    y, x = action // 8, action % 8
    return x * 8 + y


def augment8(states, actions):
    """states (n,8,8), actions (n,) ints -> (8,n,8,8), (8,n)."""
    states = np.asarray(states)
    actions = np.asarray(actions, dtype=np.int64)
    S, A = [states], [actions]
    for _ in range(3):                      # load.py:58-63
        states = np.rot90(states, k=1, axes=(1, 2))
        actions = rotate(actions)
        S.append(states)
        A.append(actions)
    states = states.transpose(0, 2, 1)      # load.py:64-68
    actions = transpose(actions)
    S.append(states)
    A.append(actions)
    for _ in range(3):                      # load.py:69-74
        states = np.rot90(states, k=1, axes=(1, 2))
        actions = rotate(actions)
        S.append(states)
        A.append(actions)
    return np.stack(S), np.stack(A)
