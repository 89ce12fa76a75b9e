"""GameFunctions: the functional flavour of the reference's board rules
(game.py:153-235) with the same names, arguments and in-place semantics, running
on the HIP kernels (B = 1 launches).  This is the copy MCTS.py calls
(MCTS.py:8,94,102,111,131).  The interactive CLI of game.py is out of scope.
"""
import numpy as np
import torch

from . import boards, ops


class GameFunctions(object):

    @classmethod
    def ac2pos(cls, actions):  # game.py:155-160
        return [[a // 8 + 1, a % 8 + 1] for a in actions]

    @classmethod
    def is_outside(cls, pos):  # game.py:163-165
        return pos[0] < 0 or pos[0] > 7 or pos[1] < 0 or pos[1] > 7

    @classmethod
    def make_state_var(cls, state, color):
        """game.py:168-174 -> (1,2,8,8) float32 (CUDA tensor): channel 0 = the
        opponent of `color`, channel 1 = `color`."""
        own, opp = boards.own_opp(state, color)
        return ops.encode_planes(own, opp)

    @classmethod
    def place_stone(cls, state, action, color):
        """game.py:180-207: in place, returns `state`; action == -1 is a pass;
        no legality check."""
        if action == -1:
            return state
        own, opp = boards.own_opp(state, color)
        ops.apply_moves(own, opp, torch.tensor([action], dtype=torch.int8, device="cuda"))
        a, b = ops.tensor_to_bits(own)[0], ops.tensor_to_bits(opp)[0]
        boards.bits_to_state(a if color == 1 else b, b if color == 1 else a, out=state)
        return state

    @classmethod
    def legal_actions(cls, state, color):
        """game.py:210-235: ascending list of a = row*8+col."""
        own, opp = boards.own_opp(state, color)
        return boards.mask_to_actions(ops.tensor_to_bits(ops.legal_moves(own, opp))[0])
