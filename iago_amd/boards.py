"""Conversions between the reference's (8,8) float32 boards (0 empty, 1, 2) and
the device bitboards (bit a = row*8+col).  Host plumbing for the B=1 mirrors of
the reference interfaces; the batched engine never leaves bitboards."""
import numpy as np
import torch

_W = (np.uint64(1) << np.arange(64, dtype=np.uint64))


def state_to_bits(state):
    s = np.asarray(state).reshape(64)
    return int(_W[s == 1].sum(dtype=np.uint64)), int(_W[s == 2].sum(dtype=np.uint64))


def bits_to_state(p1, p2, out=None):
    s = np.zeros(64, dtype=np.float32) if out is None else out.reshape(64)
    b1 = (np.uint64(p1) & _W) != 0
    b2 = (np.uint64(p2) & _W) != 0
    s[:] = 0
    s[b1] = 1
    s[b2] = 2
    return s.reshape(8, 8)


def own_opp(state, color):
    """(own, opp) int64 CUDA tensors of one board, own = stones of `color`."""
    p1, p2 = state_to_bits(state)
    a, b = (p1, p2) if color == 1 else (p2, p1)
    t = torch.from_numpy(np.array([a, b], dtype=np.uint64).view(np.int64)).cuda()
    return t[0:1], t[1:2]


def mask_to_actions(mask):
    m = int(mask) & 0xFFFFFFFFFFFFFFFF
    return [a for a in range(64) if (m >> a) & 1]


def initial_state():
    """game.py:26-30 / rl_env.py:14-18."""
    s = np.zeros((8, 8), dtype=np.float32)
    s[4, 3] = s[3, 4] = 1
    s[3, 3] = s[4, 4] = 2
    return s
