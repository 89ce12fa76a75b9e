"""MCTS: the reference's search object (MCTS.py:78-154) -- same constructor
arguments, get_move(state, color) and update_with_move(move) -- as a B = 1 view
of the batched HIP engine (engine.BatchedMCTS).

Differences, both forced: the budget is `n_sims` playouts per move when given
(the reference's 10 s wall clock, MCTS.py:142, is kept as the fallback), and
the nets are passed in instead of being read from './models' (MCTS.py:82-85).
"""
import time

import torch

from . import boards, engine, ops


class MCTS(object):

    def __init__(self, lmbda=0.5, c_puct=1, n_thr=15, time_limit=10, policy_net=None,
                 value_net=None, rollout_weights=None, n_sims=None, capacity=65536, seed=0,
                 use_graph=False):
        if policy_net is None or (value_net is None and lmbda < 1):
            raise ValueError("policy_net / value_net are required (the reference loads "
                             "./models/sl_model.npz and ./models/value_model.npz here)")
        self.lmbda, self.c_puct, self.n_thr, self.time_limit = lmbda, c_puct, n_thr, time_limit
        self.n_sims = n_sims
        self.policy_net, self.value_net = policy_net, value_net
        self._m = engine.BatchedMCTS(1, policy_net, value_net, rollout_weights, lmbda=lmbda,
                                     c_puct=c_puct, n_thr=n_thr, capacity=capacity, seed=seed,
                                     use_graph=use_graph)
        self._one = torch.ones(1, dtype=torch.uint8, device="cuda")

    def get_move(self, state, color):
        """MCTS.py:139-147: playouts from the root, then the most visited child."""
        own, opp = boards.own_opp(state, color)
        if self.n_sims is not None:
            self._m.search(own, opp, self._one, self.n_sims)
        else:
            start = time.time()
            while time.time() - start < self.time_limit:
                self._m.search(own, opp, self._one, 8)
        move = int(self._m.best_move(self._one)[0].item())
        if move == -2:
            raise ValueError("max() arg is an empty sequence: the root has no children "
                             "(fewer than n_thr playouts)")  # what MCTS.py:147 raises
        return move

    def update_with_move(self, last_move):
        """MCTS.py:149-154."""
        self._m.update_with_move(torch.tensor([last_move], dtype=torch.int8, device="cuda"))

    @property
    def n_leaf_evals(self):
        return self._m.n_leaf_evals
