"""The game front-end mirror (iago_amd.game.Game) against a transcript
recorded from the reference's game.Game (game.py:13-150, 246-262): ASCII board, prompts,
gamelog text, MCTS call pattern.  Board rules run on the HIP kernels."""
import json
import os

import numpy as np
import pytest

from tests.test_oracle_golden import _hash_probs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "frontend.json")


class _Policy(object):
    def __init__(self, salt):
        self.salt = salt

    def __call__(self, planes):
        x = planes.detach().cpu().numpy()
        return _hash_probs(x, self.salt)[0] / np.float32(65536.0)


class _ScriptedMCTS(object):
    def __init__(self):
        self.calls = []

    def get_move(self, state, color):
        from iago_amd import boards
        from iago_amd.game import GameFunctions
        acts = GameFunctions.legal_actions(state, color)
        p1, p2 = boards.state_to_bits(state)
        a = acts[(p1 * 31 + p2 * 17 + color) % len(acts)]
        self.calls.append(["get_move", p1, p2, int(color), int(a)])
        return a

    def update_with_move(self, move):
        self.calls.append(["update_with_move", int(move)])


class _Replay(object):
    """numpy.random.choice restated (cdf / searchsorted right) on the recorded uniforms."""

    def __init__(self, us):
        self.us, self.k = us, 0

    def __call__(self, n, p=None):
        u = self.us[self.k]
        self.k += 1
        cdf = np.asarray(p, dtype=np.float64).cumsum()
        cdf /= cdf[-1]
        return int(cdf.searchsorted(u, side="right"))


@pytest.mark.parametrize("case", range(2))
def test_auto_game_transcript_matches_reference(case):
    from iago_amd import boards
    from iago_amd.game import Game, play
    c = json.load(open(GOLD))["cases"][case]
    lines = []
    mcts = _ScriptedMCTS()
    replay = _Replay(c["us"])
    g = Game(True, model=_Policy(c["salt"]), mcts=mcts, date=c["date"], out=lines.append, choice=replay)
    boards.bits_to_state(c["p1"], c["p2"], out=g.state)
    g.stone_num = int(np.sum(g.state != 0))
    play(g, True)
    assert "\n".join(lines) + "\n" == c["stdout"]
    assert g.gamelog == c["gamelog"]
    assert mcts.calls == c["calls"]
    assert list(boards.state_to_bits(g.state)) == c["final"]
    assert g.play_num == c["play_num"] and replay.k == len(c["us"])


def test_manual_turn_and_gamelog_file(tmp_path):
    from iago_amd.game import Game
    typed = iter(["x", "9,9", "1,1", "3,4"])
    lines = []
    mcts = _ScriptedMCTS()
    g = Game(False, mcts=mcts, date="2000-01-03-00-00", out=lines.append, read=lambda: next(typed))
    g.turn(1, False)
    assert mcts.calls == [["update_with_move", 19]]
    assert "Try again." in lines and "This position is invalid. Choose another position" in lines
    g.turn(2, False)
    assert mcts.calls[1][0] == "get_move" and mcts.calls[2][0] == "update_with_move"
    assert g.gamelog.startswith("IaGo \n2000-01-03-00-00\n[1]You: [3, 4]\n[2]IaGo(PV-MCTS): [")
    path = g.save_gamelog(str(tmp_path))
    assert open(path).read() == g.gamelog and path.endswith("2000-01-03-00-00.txt")
