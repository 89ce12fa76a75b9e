"""GameFunctions: the functional flavour of the reference's board rules
(game.py:153-235) with the same names, arguments and in-place semantics, running
on the HIP kernels (B = 1 launches).  This is the copy MCTS.py calls
(MCTS.py:8,94,102,111,131).

Game: the front-end object of game.py:13-150 -- the same attributes,
turn logic, ASCII board, prompts, gamelog text and MCTS call pattern (get_move +
update_with_move including -1 passes) -- around injected models instead of files read
through Chainer.  argparse / main() of the reference's CLI are not reproduced; play()
is its game loop (game.py:249-262).
"""
import os
from datetime import datetime

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


def _policy_probs(model, state_var):
    """`model(state_var).data.reshape(64)` of game.py:108 for a torch module, a Chainer-style
    callable (result has .data) or a plain function returning an array."""
    with torch.no_grad():
        out = model(state_var)
    if not isinstance(out, torch.Tensor) and hasattr(out, "data"):
        out = out.data
    if isinstance(out, torch.Tensor):
        out = out.detach().to("cpu").numpy()
    return np.asarray(out).reshape(64)


class Game(object):
    """game.py:13-150.  auto=True: player 1 samples from `model` (SLPolicy), player 2 is
    `mcts` (PV-MCTS); auto=False: player 1 types positions "row,col" (1-origin) read through
    `read`.  `out` receives every line the reference prints; `choice` stands in for
    numpy.random.choice (game.py:111)."""

    def __init__(self, auto, model=None, mcts=None, date=None, out=print, read=input, choice=None):
        if auto:
            if model is None:
                raise ValueError("auto play needs the SLPolicy model (the reference loads "
                                 "./models/sl_model.npz here, game.py:19-21)")
            self.p1 = "IaGo(SLPolicy)"
            self.model = model
        else:
            self.p1 = "You"
            self.model = None
        self.p2 = "IaGo(PV-MCTS)"
        if mcts is None:
            raise ValueError("an MCTS object is required (iago_amd.MCTS.MCTS(policy_net=..., value_net=...))")
        self.state = boards.initial_state()
        self.stone_num = 4
        self.play_num = 1
        self.pass_flg = False
        self.date = date if date is not None else datetime.now().strftime("%Y-%m-%d-%H-%M")
        self.gamelog = "IaGo \n" + self.date + "\n"
        self.mcts = mcts
        self._out, self._read = out, read
        self._choice = choice if choice is not None else np.random.choice

    def _count(self, v):
        return int(np.sum(self.state == v))

    def board_lines(self):
        """The lines show() prints (game.py:40-50)."""
        cell = {0: "   ", 1: " X ", 2: " O "}
        rule = " " + "-" * 33
        lines = ["".join("   %d" % c for c in range(1, 9)) + "   "]
        for i in range(8):
            lines.append(rule)
            lines.append(str(i + 1) + "|" + "|".join(cell[int(v)] for v in self.state[i]) + "|")
        lines.append(rule)
        lines.append("%s(X):%d, %s(O):%d, Empty:%d" % (self.p1, self._count(1), self.p2, self._count(2),
                                                     self._count(0)))
        lines.append("\n")
        return lines

    def show(self):
        for line in self.board_lines():
            self._out(line)

    def judge(self):
        """game.py:53-62."""
        n1, n2 = self._count(1), self._count(2)
        if n1 > n2:
            self._out("%s WIN!" % self.p1)
        elif n1 < n2:
            self._out("%s WIN" % self.p2)
        else:
            self._out("DRAW")
        return "%s:%d, %s:%d, Empty:%d" % (self.p1, n1, self.p2, n2, self._count(0))

    def safeinput(self):
        """game.py:65-72: "d,d" or ask again."""
        while True:
            line = self._read()
            if len(line) == 3 and line[0].isdigit() and line[1] == "," and line[2].isdigit():
                return line.split(",")
            self._out("Try again.")

    def get_action(self, color, actions):
        """game.py:75-93."""
        if color == 1:
            while True:
                self._out("Your turn. Choose a position!")
                row, col = (int(e) for e in self.safeinput())
                action = (row - 1) * 8 + (col - 1)
                if action in actions:
                    break
                self._out("This position is invalid. Choose another position")
            self.mcts.update_with_move(action)
        else:
            self._out("Thinking... Wait a second.")
            action = self.mcts.get_move(self.state, 2)
            self.mcts.update_with_move(action)
        return action

    def get_action_auto(self, color, actions):
        """game.py:96-118 (the last forced move bypasses the search tree, as there)."""
        if self.stone_num > 62 and len(actions) == 1:
            return actions[0]
        if color == 1:
            prob = _policy_probs(self.model, GameFunctions.make_state_var(self.state, color))
            valid = np.zeros(64)
            valid[actions] = 1
            while True:
                action = int(self._choice(64, p=prob * valid / np.sum(prob * valid)))
                if action in actions:
                    break
            self.mcts.update_with_move(action)
        else:
            action = self.mcts.get_move(self.state, 2)
            self.mcts.update_with_move(action)
        return action

    def turn(self, color, auto):
        """game.py:121-145."""
        name = (self.p1, self.p2)[color - 1]
        actions = GameFunctions.legal_actions(self.state, color)
        self._out("Valid choice: %s" % (GameFunctions.ac2pos(actions),))
        if len(actions) > 0:
            action = self.get_action_auto(color, actions) if auto else self.get_action(color, actions)
            position = [action // 8 + 1, action % 8 + 1]
            self._out(str(position))
            self.state = GameFunctions.place_stone(self.state, action, color)
            self.stone_num += 1
            self.show()
            self.pass_flg = False
            self.gamelog += "[%d]%s: %s\n" % (self.play_num, name, position)
        else:
            if self.pass_flg:
                self.stone_num = 64  # two passes in a row end the game
            self._out(name + " pass.")
            self.pass_flg = True
            self.mcts.update_with_move(-1)
            self.gamelog += "[%d]%s: Pass\n" % (self.play_num, name)
        self.play_num += 1

    def save_gamelog(self, directory="./gamelog"):
        """game.py:148-154: <directory>/<date>.txt."""
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, self.date + ".txt")
        with open(path, "w") as f:
            f.write(self.gamelog)
        return path


def play(game, auto):
    """The game loop of game.py:246-262 (banners excluded): returns the judge line,
    which is also appended to the gamelog."""
    game.show()
    while game.stone_num < 64:
        game.turn(1, auto)
        game.turn(2, auto)
    jd = game.judge()
    game._out(jd)
    game.gamelog += jd + "\n"
    return jd
