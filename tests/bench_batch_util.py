"""Helpers of tests/test_bench_batch_gpu.py that its worker processes share (tests/rebuild_worker.py): the oracle's
policy_fn / value_fn as the production kernels on ONE board, and the rebuild of a recorded game's searches with
oracle/mcts_py.MCTS (MCTS.py:105-154)."""
import numpy as np
import torch

from oracle import mcts_py
from oracle import oracle as orc


def make_nets():
    """bench.mcts_leg's nets, to the letter (random init, seed 0)."""
    from iago_amd import network
    torch.manual_seed(0)
    policy = network.SLPolicy().cuda().eval()
    value = network.Value().cuda().eval()
    value.split_f16 = True
    return policy, value


class Probe(object):
    """policy_fn / value_fn of the oracle: the production kernels on ONE board, memoised."""

    def __init__(self, B):
        self.ops, self.policy, self.value = B["ops"], B["policy"], B["value"]
        self.p_cache, self.v_cache = {}, {}
        self.idx = torch.zeros(1, dtype=torch.int64, device="cuda")
        self.one = torch.ones(1, dtype=torch.int32, device="cuda")
        self.out = torch.zeros(1, dtype=torch.float32, device="cuda")

    def _boards(self, x):
        x = np.asarray(x, np.float32).reshape(2, 64)
        own = sum(1 << a for a in range(64) if x[1, a] == 1.0)   # channel 1 = side to move
        opp = sum(1 << a for a in range(64) if x[0, a] == 1.0)
        return (own, opp), self.ops.bits_to_tensor([own]), self.ops.bits_to_tensor([opp])

    def policy_fn(self, x):
        key, o, p = self._boards(x)
        if key not in self.p_cache:
            self.p_cache[key] = self.policy.forward_boards_split3(o, p).cpu().numpy().reshape(64).copy()
        return self.p_cache[key]

    def value_fn(self, x):
        key, o, p = self._boards(x)
        if key not in self.v_cache:
            with torch.no_grad():
                self.value.forward_boards_counted(o, p, self.idx, self.one, self.out)
            self.v_cache[key] = np.float32(self.out.cpu().numpy()[0])
        return self.v_cache[key]


def rebuild(B, probe, g, n_turns, n_thr=15, compare_from=0):
    """(ii): game g's first n_turns turns searched again by the oracle's MCTS.py restatement: root visit counts by
    action and moves equal the launch's records (from turn compare_from on: the searches before it are run all the
    same -- MCTS.update_with_move carries the subtree from search to search).  Returns the searches compared."""
    n_sims = B["n_sims"]
    it = iter(B["zlog"][:B["zn"][g], g])
    om = mcts_py.MCTS(probe.policy_fn, probe.value_fn, lambda s, c: int(next(it)), lmbda=0.5, c_puct=1.0, n_thr=n_thr)
    state = orc.initial_state()
    stone_num, pass_flg, t, n_cmp = 4, False, 0, 0
    while stone_num < 64 and t < n_turns:
        for color in (1, 2):
            acts = orc.legal_actions(state, color)
            if len(acts) > 0:
                a = om.get_move(state, color, n_sims)
                want = np.zeros(64, np.int64)
                for act, ch in om.root.children.items():
                    want[act] = ch.n_visits
                if t >= compare_from:
                    assert B["pi"][t, g].tolist() == want.tolist(), (g, t)
                    assert int(B["move"][t, g]) == a, (g, t)
                    n_cmp += 1
                om.update_with_move(a)
                orc.place_stone(state, a, color)
                stone_num += 1
                pass_flg = False
            else:
                if pass_flg:
                    stone_num = 64
                pass_flg = True
                om.update_with_move(-1)
            t += 1
            if t >= n_turns:
                break
    if n_turns >= B["game_turns"][g]:
        assert next(it, None) is None, g      # the oracle consumed exactly the playouts the launch ran
    B.setdefault("max_path", {})[g] = om.max_path   # the deepest descent (nodes on a playout's path, the root included)
    return n_cmp


