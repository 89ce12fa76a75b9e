"""The value look-ahead (iago_mcts_value_ahead, include/iago_hip_experimental.h; engine.BatchedMCTS(value_ahead=True)):
an EXPERIMENTAL schedule of the per-playout engine -- built, tree-identical, measured slower (LABNOTES.md, round 4), off by
default -- kept with a smoke test of each property.

value_func(state) (MCTS.py:97-103) is evaluated for the children of a node when the node expands, as
one batch off the playouts' critical path, instead of at each child's first visit (MCTS.py:123-124).
What must hold: (1) every value stored in a node -- whichever way it got there -- is the value net's
output for THAT node's position, bit for bit (a wrong row -> node mapping or a batch-dependent kernel
would show here); (2) the trees are the trees of the search without the look-ahead; (3) the look-ahead
does take evaluations off the critical path.  The comparison of the production path with the oracle
(tests/test_mcts_production_gpu.py) has one case with the look-ahead on.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.conftest import GOLDEN, load_json
from tests.gpu_util import random_positions, state_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets():
    from iago_amd import engine, network, ops
    assert torch.cuda.is_available()
    torch.manual_seed(3)
    policy = network.SLPolicy().cuda().eval()          # random init: broad trees, like bench.py's leg
    value = network.Value().cuda().eval()
    g = load_json("simulate.json")
    return engine, ops, policy, value, ops.RolloutWeights(g["shipped_w"], g["shipped_b"])


def _search(nets, G, n_sims, own, opp, n_sims2=0, **kw):
    engine, ops, policy, value, rw = nets
    kw.setdefault("persistent", False)   # (these tests are about the per-playout launches)
    m = engine.BatchedMCTS(G, policy, value, rw, n_thr=15, capacity=engine.suggest_capacity(n_sims + n_sims2, 15, moves=2),
                           seed=21, game_id_base=300, **kw)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    active = torch.ones(G, dtype=torch.uint8, device="cuda")
    m.search(o, p, active, n_sims)
    if n_sims2:
        mv = m.best_move(active)[0].clone()
        mv = torch.where(mv == -2, torch.full_like(mv, -1), mv)
        m.update_with_move(mv, active.clone())
        ops.apply_moves(o, p, mv)
        m.search(p, o, active, n_sims2)
    return m


def _positions(G):
    own, opp = random_positions(G, seed=77)
    own[: G // 2] = 0x0000000810000000
    opp[: G // 2] = 0x0000001008000000
    return own, opp


def _tree_arrays(m):
    t = m.tree
    return {k: getattr(t, k).cpu().numpy().copy() for k in ("n_visits", "q", "p", "first_child", "parent", "action",
                                                             "n_children", "n_nodes", "root")}


@pytest.mark.parametrize("use_graph,async_steps", [(True, False)])   # (one smoke: an engine fenced off as experimental)
def test_trees_equal_with_and_without_value_ahead(nets, use_graph, async_steps):
    G, n_sims, n_sims2 = 96, 100, 45
    own, opp = _positions(G)
    a = _search(nets, G, n_sims, own, opp, n_sims2, use_graph=use_graph, async_steps=async_steps, value_ahead=True)
    b = _search(nets, G, n_sims, own, opp, n_sims2, use_graph=use_graph, async_steps=False, value_ahead=False)
    assert a.value_ahead and not b.value_ahead
    ta, tb = _tree_arrays(a), _tree_arrays(b)
    for k in ta:
        assert np.array_equal(ta[k], tb[k]), k
    # the look-ahead evaluated rows, and fewer leaves were evaluated on the critical path
    assert a.n_value_ahead > 0 and b.n_value_ahead == 0
    assert a.n_value_inline < b.n_value_inline
    assert a.n_leaf_evals == b.n_leaf_evals == G * (n_sims + n_sims2)
    # every value either search stored is the same number
    va, vb = a.tree.v.cpu().numpy(), b.tree.v.cpu().numpy()
    both = ~np.isnan(va) & ~np.isnan(vb)
    assert both.sum() > G and np.array_equal(va[both], vb[both])
    assert (~np.isnan(vb) & np.isnan(va)).sum() == 0      # whatever the plain search evaluated, this one has too
    a.close()
    b.close()


def test_stored_values_belong_to_their_nodes(nets):
    """Walk whole trees from the root with the ORACLE's place_stone (game.py:180-207): the value in
    every node equals the one-board walk of the value net on the node's position."""
    engine, ops, policy, value, rw = nets
    G, n_sims = 24, 120
    own, opp = _positions(G)
    m = _search(nets, G, n_sims, own, opp, use_graph=True, value_ahead=True)
    t = _tree_arrays(m)
    v = m.tree.v.cpu().numpy()
    cap = m.tree.capacity
    boards_own, boards_opp, want_idx = [], [], []
    for g in range(G):
        base = g * cap
        stack = [(int(t["root"][g]), state_of(own[g], opp[g]), 1)]
        while stack:
            node, state, color = stack.pop()
            if not np.isnan(v[base + node]):
                p1, p2 = orc.state_to_bits(state)
                boards_own.append(p1 if color == 1 else p2)
                boards_opp.append(p2 if color == 1 else p1)
                want_idx.append(base + node)
            fc, k = int(t["first_child"][base + node]), int(t["n_children"][base + node])
            if fc >= 0:
                for j in range(k):
                    a = int(t["action"][base + fc + j])
                    s2 = state.copy()
                    orc.place_stone(s2, a, color)
                    stack.append((fc + j, s2, 3 - color))
    n = len(want_idx)
    assert n > 10 * G
    bo, bp = ops.bits_to_tensor(np.array(boards_own, np.uint64)), ops.bits_to_tensor(np.array(boards_opp, np.uint64))
    idx = torch.arange(n, dtype=torch.int64, device="cuda")
    cnt = torch.full((1,), n, dtype=torch.int32, device="cuda")
    out = torch.zeros(n, dtype=torch.float32, device="cuda")
    with torch.no_grad():
        value.forward_boards_counted(bo, bp, idx, cnt, out)       # the one-board walk of the leaf evaluation
    assert np.array_equal(out.cpu().numpy(), v[np.array(want_idx)])
    # values WITHOUT a visit exist: the look-ahead put them there (never by a playout)
    assert ((t["n_visits"] == 0) & ~np.isnan(v)).sum() > 0
    m.close()


@pytest.mark.parametrize("boards,grid", [(1, 64), (2, 176), (4, 7)])
def test_value_batch_equals_one_board_walks(nets, boards, grid):
    """iago_value_forward_batch (device-side count, capped grid, 1 / 2 / 4 boards per workgroup) gives
    the values of the one-board walk, bit for bit, for ragged counts; rows past the count untouched."""
    engine, ops, policy, value, rw = nets
    n = 333
    own, opp = random_positions(n, seed=5)
    bo, bp = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    idx = torch.arange(n, dtype=torch.int64, device="cuda")
    want = torch.zeros(n, dtype=torch.float32, device="cuda")
    with torch.no_grad():
        value.forward_boards_counted(bo, bp, idx, torch.full((1,), n, dtype=torch.int32, device="cuda"), want)
        for count in (0, 1, 5, 330, 333, 1000):
            out = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
            value.forward_boards_batch(bo, bp, torch.full((1,), count, dtype=torch.int32, device="cuda"), out, boards, grid)
            k = min(count, n)
            assert torch.equal(out[:k], want[:k]), (boards, grid, count)
            assert bool((out[k:] == -7.0).all())
    value.check_saturation()
