"""GameEnv: the gym-like environment of the reference (rl_env.py:10-185) with
the same methods and step semantics, its rules running on the HIP kernels.
The agent is colour 1; `step(action)` plays it, then lets model2 answer as
colour 2.  Observations are (1,2,8,8) planes [state==1, state==2] -- NOT
perspective-swapped, unlike make_state_var (rl_env.py:36-38,70-72).
"""
import random

import numpy as np
import torch

from . import boards, ops


class GameEnv(object):

    def __init__(self, model1, model2, choice=None, fallback_choice=None):
        """choice(n, p=...) replaces numpy.random.choice (rl_env.py:166) and
        fallback_choice(seq) python's random.choice (rl_env.py:48): tests inject
        recorded draws, exactly like game.Game's injected I/O."""
        self.model1, self.model2 = model1, model2
        self.choice = choice if choice is not None else (lambda n, p=None: np.random.choice(n, p=p))
        self.fallback_choice = fallback_choice if fallback_choice is not None else random.choice
        self.reset()

    def _obs(self):
        # [state==1, state==2]: "own" = player 2, "opp" = player 1 in encode_planes' terms
        own, opp = boards.own_opp(self.state, 2)
        return ops.encode_planes(own, opp)

    def reset(self):  # rl_env.py:26-39
        self.state = boards.initial_state()
        self.stone_num, self.pass_flg = 4, False
        return self._obs()

    def is_outside(self, pos):  # rl_env.py:82-83
        return pos[0] < 0 or pos[0] > 7 or pos[1] < 0 or pos[1] > 7

    def valid_pos(self, color):  # rl_env.py:114-138: 1-origin [y, x], row-major ascending
        own, opp = boards.own_opp(self.state, color)
        mask = ops.tensor_to_bits(ops.legal_moves(own, opp))[0]
        return [[a // 8 + 1, a % 8 + 1] for a in boards.mask_to_actions(mask)]

    def place_stone(self, position, color):  # rl_env.py:88-112
        a = (position[0] - 1) * 8 + (position[1] - 1)
        own, opp = boards.own_opp(self.state, color)
        ops.apply_moves(own, opp, torch.tensor([a], dtype=torch.int8, device="cuda"))
        x, y = ops.tensor_to_bits(own)[0], ops.tensor_to_bits(opp)[0]
        boards.bits_to_state(x if color == 1 else y, y if color == 1 else x, out=self.state)

    def judge(self):  # rl_env.py:141-149, from player 1's view
        you, ai = np.sum(self.state == 1), np.sum(self.state == 2)
        return 1 if you > ai else (-1 if you < ai else 0)

    def __call__(self):  # rl_env.py:78-79
        return self.judge()

    def get_position(self, color, positions):
        """rl_env.py:152-172: sample from `out - min(out)` (NOT masked), retry
        until the draw is legal.  Colour 1 sees the swapped board like the
        reference; the models are `.predictor`-wrapped there and plain here."""
        if color == 1:
            own, opp = boards.own_opp(self.state, 1)
            x = ops.encode_planes(own, opp)
            model = self.model1
        else:
            x = self._obs()
            model = self.model2
        model = getattr(model, "predictor", model)
        while True:
            # the reference re-evaluates the net on every retry (rl_env.py:170-171) and
            # shifts / normalises in the net's float32 (rl_env.py:165-166)
            with torch.no_grad():
                out = torch.as_tensor(model(x)).reshape(64).to(torch.float32).cpu().numpy().copy()
            out -= np.min(out)
            if not any(out[(q[0] - 1) * 8 + (q[1] - 1)] > 0.0 for q in positions):
                # the shift gives the net's lowest cell probability 0: when every legal cell sits there the reference
                # draws again for ever (rl_env.py:167-172, the same net output every time); the mirror says so instead
                raise RuntimeError("GameEnv.get_position: the shifted net output puts no mass on a legal position "
                                   "(the reference's retry loop, rl_env.py:167-172, never ends here)")
            idx = int(self.choice(64, p=out / np.sum(out)))
            position = [idx // 8 + 1, idx % 8 + 1]
            if position in positions:
                return position

    def step(self, action):  # rl_env.py:41-74
        done = False
        positions = self.valid_pos(1)
        if len(positions) > 0:
            position = [action // 8 + 1, action % 8 + 1]
            if position not in positions:
                position = self.fallback_choice(positions)  # rl_env.py:46-48
            self.place_stone(position, 1)
            self.stone_num += 1
            self.pass_flg = False
        else:
            if self.pass_flg:
                done = True
            self.pass_flg = True
        positions = self.valid_pos(2)
        if len(positions) > 0:
            self.place_stone(self.get_position(2, positions), 2)
            self.stone_num += 1
            self.pass_flg = False
        else:
            if self.pass_flg:
                done = True
            self.pass_flg = True
        if self.stone_num >= 64:
            done = True
        return self._obs(), 0, done, None
