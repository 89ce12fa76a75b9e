"""Simulate: the leaf rollout of the reference (mcts_self_play.py:9-139) with the
same constructor / call signature, running the fused HIP rollout kernel."""
import copy

import numpy as np
import torch

from . import boards, network, ops

_DEFAULT_WEIGHTS = None


def set_rollout_model(model_or_path):
    """The reference reloads './models/rollout_model.npz' on every construction
    (mcts_self_play.py:18-19); here the weights are staged on the device once."""
    global _DEFAULT_WEIGHTS
    model = model_or_path
    if isinstance(model_or_path, str):
        model = network.RolloutPolicy().load_npz(model_or_path)
    _DEFAULT_WEIGHTS = ops.RolloutWeights(*model.kernel_weights())
    return _DEFAULT_WEIGHTS


class Simulate(object):
    _counter = 0

    def __init__(self, state, weights=None, seed=0, uniforms=None):
        self.state = copy.deepcopy(state)                      # mcts_self_play.py:13
        self.stone_num = 64 - int(np.sum(self.state == 0))     # mcts_self_play.py:15
        self.pass_flg = False
        self.weights = weights if weights is not None else _DEFAULT_WEIGHTS
        if self.weights is None:
            raise RuntimeError("no rollout weights: call set_rollout_model(path or RolloutPolicy)")
        self.seed = seed
        # test hook: the draws of np.random.choice in the reference run (mcts_self_play.py:104-106),
        # indexed by turn (the entry of a passing turn is not consumed), instead of the Philox
        # stream -- what the parity tests replay the recorded reference games with
        self.uniforms = uniforms
        Simulate._counter += 1
        self.rollout_id = Simulate._counter

    def __call__(self, color):
        """Play to the end with the rollout policy; +1/0/-1 from `color`'s view
        (mcts_self_play.py:25-29,113-121).  self.state ends as the final board."""
        own, opp = boards.own_opp(self.state, color)
        us = None
        if self.uniforms is not None:
            from ._lib import IAGO_MAX_TURNS
            us = torch.zeros((IAGO_MAX_TURNS, 1), dtype=torch.float32, device=own.device)
            u = torch.as_tensor(np.asarray(self.uniforms, np.float32))
            us[:u.numel(), 0] = u.to(own.device)
        res = ops.rollout(own, opp, self.weights, seed=self.seed, id_base=self.rollout_id,
                          uniforms=us, want_final=True, want_turns=True)
        a, b = ops.tensor_to_bits(res.final_own)[0], ops.tensor_to_bits(res.final_opp)[0]
        boards.bits_to_state(a if color == 1 else b, b if color == 1 else a, out=self.state)
        self.stone_num = 64
        return int(res.z.item())


def is_outside(pos):  # mcts_self_play.py:137-139
    return pos[0] < 0 or pos[0] > 7 or pos[1] < 0 or pos[1] > 7
