"""Full PV-MCTS self-play games on the GPU engine (engine.SelfPlayEngine) equal
the oracle's game-by-game: same move at every turn, same final board, same z,
same recorded visit distributions."""
import numpy as np
import pytest
import torch

from oracle import mcts_py
from oracle import oracle as orc
from tests.test_mcts_gpu import fake_nets

pytestmark = pytest.mark.gpu


def test_selfplay_games_match_oracle():
    from iago_amd import engine, ops
    B, n_sims, n_thr, lmbda = 6, 20, 2, 0.5
    policy_np, value_np, policy_t, value_t = fake_nets(11)
    m = engine.BatchedMCTS(B, policy_t, value_t, None, lmbda=lmbda, n_thr=n_thr, capacity=8192,
                           seed=1)
    zs = []  # (sim index) -> z per game, only meaningful for the games active in that sim
    acts = []
    orig = m.simulate

    def sim(own, opp, active, n_active=None):
        orig(own, opp, active, n_active)
        zs.append(m.z.cpu().numpy().copy())
        acts.append(active.cpu().numpy().copy())

    m.simulate = sim
    hc_cells = [None, (2, 4), (3, 5), (4, 2), (5, 3), None]
    hc = np.array([0 if c is None else 1 << (c[0] * 8 + c[1]) for c in hc_cells], np.uint64)
    res = engine.SelfPlayEngine(m).play(n_sims, handicap=ops.bits_to_tensor(hc))
    moves = res.move.cpu().numpy()
    z = res.z.cpu().numpy()
    f1, f2 = ops.tensor_to_bits(res.final_p1), ops.tensor_to_bits(res.final_p2)
    pi = res.pi.cpu().numpy()
    valid = res.valid.cpu().numpy()
    for g in range(B):
        it = iter([zz[g] for zz, aa in zip(zs, acts) if aa[g]])
        om = mcts_py.MCTS(policy_np, value_np, lambda s, c: int(next(it)), lmbda=lmbda, n_thr=n_thr)
        want_moves, want_z, state = mcts_py.selfplay_game(om, n_sims, hc_cells[g])
        got = moves[:len(want_moves), g].tolist()
        assert got == want_moves, g
        assert np.all(moves[len(want_moves):, g] == -1)
        assert z[g] == want_z
        assert orc.state_to_bits(state) == (int(f1[g]), int(f2[g]))
        for t, a in enumerate(want_moves):
            assert valid[t, g] == (1 if a >= 0 else 0)
            if a >= 0:
                assert pi[t, g].sum() <= n_sims + 400 and pi[t, g, a] == pi[t, g].max()
    tup = res.tuples()
    n_rows = int(valid.sum())
    assert tup["own"].numel() == n_rows and tuple(tup["pi"].shape) == (n_rows, 64)
    assert set(np.unique(tup["z"].cpu().numpy())) <= {-1, 0, 1}
    assert m.n_leaf_evals == sum(int(a.sum()) for a in acts)
