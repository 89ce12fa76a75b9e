"""ctypes front-end of the C oracle (oracle/othello_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of othello_oracle.c.  The functions
keep the reference's calling conventions: `state` is an (8,8) float32 array of
0/1/2, `color` is 1 or 2, actions are ints ``row*8+col`` and ``-1`` is a pass.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "othello_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        u8p = C.POINTER(C.c_uint8)
        L.orc_legal_actions.argtypes = [fp, C.c_int, ip]
        L.orc_legal_actions.restype = C.c_int
        L.orc_place_stone.argtypes = [fp, C.c_int, C.c_int]
        L.orc_place_stone.restype = None
        L.orc_make_state_var.argtypes = [fp, C.c_int, fp]
        L.orc_make_state_var.restype = None
        L.orc_env_obs.argtypes = [fp, fp]
        L.orc_env_obs.restype = None
        L.orc_judge.argtypes = [fp, C.c_int]
        L.orc_judge.restype = C.c_int
        L.orc_choice_cdf.argtypes = [dp, C.c_int, C.c_double]
        L.orc_choice_cdf.restype = C.c_int
        L.orc_rollout_policy.argtypes = [fp, fp, fp, fp, fp]
        L.orc_rollout_policy.restype = None
        L.orc_masked_probs.argtypes = [fp, ip, C.c_int, dp]
        L.orc_masked_probs.restype = None
        L.orc_philox.argtypes = [C.c_uint64] + [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)]
        L.orc_philox.restype = None
        L.orc_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_uniform.restype = C.c_float
        L.orc_simulate.argtypes = [fp, C.c_int, fp, fp, fp, C.c_uint64, C.c_uint32, u8p, ip]
        L.orc_simulate.restype = C.c_int
        L.orc_random_playout.argtypes = [fp, C.c_int, C.c_uint64, C.c_uint32, u8p, ip]
        L.orc_random_playout.restype = C.c_int
        L.orc_random_playout_stream.argtypes = [fp, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, u8p, ip]
        L.orc_random_playout_stream.restype = C.c_int
        L.orc_simulate_batch.argtypes = [fp, C.c_int, fp, fp, C.c_uint64, C.c_uint32, C.c_int,
                                         C.POINTER(C.c_int8)]
        L.orc_simulate_batch.restype = C.c_long
        L.orc_node_P.argtypes = [C.c_float]
        L.orc_node_P.restype = C.c_float
        L.orc_node_U.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int]
        L.orc_node_U.restype = C.c_double
        L.orc_node_update_Q.argtypes = [C.c_float, C.c_float, C.c_int]
        L.orc_node_update_Q.restype = C.c_float
        L.orc_leaf_value.argtypes = [C.c_float, C.c_float, C.c_int]
        L.orc_leaf_value.restype = C.c_float
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _state(state):
    s = np.ascontiguousarray(state, dtype=np.float32)
    assert s.shape == (8, 8)
    return s


# ---------------------------------------------------------------- rules
def initial_state(handicap=None):
    """Start position (rl_env.py:14-18, game.py:26-30); `handicap` = (row,col)
    of the extra colour-2 stone of src/train_rl.py:43-46."""
    s = np.zeros((8, 8), dtype=np.float32)
    s[4, 3] = s[3, 4] = 1
    s[3, 3] = s[4, 4] = 2
    if handicap is not None:
        s[handicap[0], handicap[1]] = 2
    return s


def legal_actions(state, color):
    s = _state(state)
    out = (C.c_int * 64)()
    n = lib().orc_legal_actions(_f(s), int(color), out)
    return [out[i] for i in range(n)]


def place_stone(state, action, color):
    """In place, like the reference; returns the same array."""
    assert state.dtype == np.float32 and state.flags.c_contiguous and state.shape == (8, 8)
    lib().orc_place_stone(_f(state), int(action), int(color))
    return state


def make_state_var(state, color):
    s = _state(state)
    out = np.empty((1, 2, 8, 8), dtype=np.float32)
    lib().orc_make_state_var(_f(s), int(color), _f(out))
    return out


def env_obs(state):
    s = _state(state)
    out = np.empty((1, 2, 8, 8), dtype=np.float32)
    lib().orc_env_obs(_f(s), _f(out))
    return out


def judge(state, color=1):
    return lib().orc_judge(_f(_state(state)), int(color))


# ------------------------------------------------------------- sampling
def choice_cdf(p, u):
    p = np.ascontiguousarray(p, dtype=np.float64)
    return lib().orc_choice_cdf(p.ctypes.data_as(C.POINTER(C.c_double)), p.size, float(u))


def rollout_policy(x, w, b):
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(2, 8, 8)
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(18)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(64)
    prob = np.empty(64, dtype=np.float32)
    logits = np.empty(64, dtype=np.float32)
    lib().orc_rollout_policy(_f(x), _f(w), _f(b), _f(prob), _f(logits))
    return prob, logits


def masked_probs(prob, actions):
    prob = np.ascontiguousarray(prob, dtype=np.float32)
    acts = (C.c_int * len(actions))(*actions)
    p = np.empty(64, dtype=np.float64)
    lib().orc_masked_probs(_f(prob), acts, len(actions), p.ctypes.data_as(C.POINTER(C.c_double)))
    return p


def philox(seed, c0, c1, c2=0, c3=0):
    out = (C.c_uint32 * 4)()
    lib().orc_philox(seed, c0, c1, c2, c3, out)
    return [out[i] for i in range(4)]


def uniform(seed, game, step, stream=0):
    return float(lib().orc_uniform(seed, game, step, stream))


# -------------------------------------------------------------- rollout
def simulate(state, color, w, b, uniforms=None, seed=0, game_id=0):
    """Simulate(state)(color).  Returns (z, final_state, trace) where trace is
    the per-turn action list (-1 = pass)."""
    s = _state(state).copy()
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(18)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(64)
    trace = (C.c_uint8 * 160)()
    nt = C.c_int(0)
    up = None
    if uniforms is not None:
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float32)
        assert uniforms.size >= 130
        up = _f(uniforms)
    z = lib().orc_simulate(_f(s), int(color), _f(w), _f(b), up, int(seed), int(game_id), trace,
                           C.byref(nt))
    tr = [(-1 if trace[i] == 0xFF else trace[i]) for i in range(nt.value)]
    return z, s, tr


def random_playout(state, color, seed=0, game_id=0, stream=0):
    """Uniformly random legal moves to the end (Simulate's turn structure, mcts_self_play.py:25-29,124-134)
    from Philox stream `stream` of (seed, game_id).  Returns (z from `color`'s view, final state, trace)."""
    s = _state(state).copy()
    trace = (C.c_uint8 * 160)()
    nt = C.c_int(0)
    z = lib().orc_random_playout_stream(_f(s), int(color), int(seed), int(game_id) & 0xFFFFFFFF,
                                        int(stream) & 0xFFFFFFFF, trace, C.byref(nt))
    tr = [(-1 if trace[i] == 0xFF else trace[i]) for i in range(nt.value)]
    return z, s, tr


def simulate_batch(state, color, w, b, seed, first_game, n):
    s = _state(state)
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(18)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(64)
    z = np.empty(n, dtype=np.int8)
    steps = lib().orc_simulate_batch(_f(s), int(color), _f(w), _f(b), int(seed), int(first_game),
                                     int(n), z.ctypes.data_as(C.POINTER(C.c_int8)))
    return z, int(steps)


# ------------------------------------------------------------ node math
def node_P(prob):
    return np.float32(lib().orc_node_P(float(np.float32(prob))))


def node_U(c_puct, P, parent_n, n):
    return float(lib().orc_node_U(float(c_puct), float(np.float32(P)), int(parent_n), int(n)))


def node_update_Q(Q, leaf_value, n_after):
    return np.float32(lib().orc_node_update_Q(float(np.float32(Q)), float(np.float32(leaf_value)),
                                              int(n_after)))


def leaf_value(lmbda, v, z):
    return np.float32(lib().orc_leaf_value(float(lmbda), float(np.float32(v)), int(z)))


# ----------------------------------------------------- bitboard bridging
def state_to_bits(state):
    """(8,8) 0/1/2 board -> (p1, p2) uint64 with bit index a = row*8+col."""
    s = np.asarray(state).reshape(64)
    p1 = 0
    p2 = 0
    for a in range(64):
        if s[a] == 1:
            p1 |= 1 << a
        elif s[a] == 2:
            p2 |= 1 << a
    return p1, p2


def bits_to_state(p1, p2):
    s = np.zeros(64, dtype=np.float32)
    for a in range(64):
        if (int(p1) >> a) & 1:
            s[a] = 1
        if (int(p2) >> a) & 1:
            s[a] = 2
    return s.reshape(8, 8)


def actions_to_mask(actions):
    m = 0
    for a in actions:
        m |= 1 << a
    return m
