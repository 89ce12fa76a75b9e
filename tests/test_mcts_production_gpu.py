"""The PRODUCTION PV-MCTS path against the oracle's restatement of MCTS.py, directly.

tests/test_mcts_gpu.py compares the search with oracle/mcts_py.py through stand-in nets, which
select the host-counted playout (no look-ahead, no value cache), and ties the production
variants to that path with GPU-vs-GPU tree-identity tests.  Here the engine runs exactly as
bench.py runs it -- BatchedMCTS(use_graph=True) at its defaults: hipGraph replay, policy
look-ahead 4 with the batches on a second stream beside 2 playouts, value cache, one-launch
descent (descend_kernel), one-launch leaf evaluation (value_rollout_kernel: one-board and
two-board Value walks + rollout_row_kernel<false>), path backup (mix_backup_path_kernel),
policy_resident_kernel in two launches -- with the reference's SHIPPED SLPolicy / Value /
RolloutPolicy weights (MCTS.py:82-85, mcts_self_play.py:18-19), and every tree is compared
bit for bit (visit counts, float32 Q and P, child order, chosen move, also after
update_with_move) with the tree oracle/mcts_py.MCTS builds for the same game when it is fed

  * the rollout results the search itself backed up: `z_log`, a device record the backup kernel
    writes per (playout, game) -- Simulate (MCTS.py:125) draws from a Philox stream the oracle's
    float32 softmax cannot reproduce at CDF edges, so z is replayed, as test_mcts_gpu.py does;
  * the nets' float32 outputs for every position the oracle visits, computed by the SAME
    production kernels on that single board (policy_resident_kernel / the one-board Value walk:
    a board's output does not depend on its batch -- which this test therefore also proves
    inside a search: any difference between a value or prior the search used and the one the
    oracle was fed shows up in Q or P).

Reference semantics: MCTS.py:105-154, mcts_self_play.py:25-29,100-134.
"""
import os

import numpy as np
import pytest
import torch

from oracle import mcts_py
from oracle import oracle as orc
from tests.conftest import GOLDEN, load_json
from tests.gpu_util import random_positions, state_of
from tests.test_oracle_golden import _cmp_tree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shipped():
    from iago_amd import engine, network, ops
    assert torch.cuda.is_available()
    policy = network.SLPolicy().load_npz(os.path.join(GOLDEN, "sl_model.npz")).cuda().eval()
    value = network.Value().load_npz(os.path.join(GOLDEN, "value_model.npz")).cuda().eval()
    g = load_json("simulate.json")
    return engine, ops, policy, value, ops.RolloutWeights(g["shipped_w"], g["shipped_b"])


class NetProbe(object):
    """policy_fn / value_fn of the oracle: the production kernels on ONE board, memoised."""

    def __init__(self, ops, policy, value):
        self.ops, self.policy, self.value = ops, policy, value
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


def _positions(G, golden_rules):
    own, opp = random_positions(G, seed=33)
    own[: G // 2] = 0x0000000810000000     # half of the games at the start position, like bench.py
    opp[: G // 2] = 0x0000001008000000
    eb = golden_rules["edge_boards"]
    # 'pass1' (index 6): colour 1 must pass; 'dead' (5): nobody can move; 'full' (8)
    for k, e in ((1, 6), (2, 5), (3, 8)):
        own[k], opp[k] = eb[e][0], eb[e][1]
    return own, opp


# (the two engines that serve the path -- the persistent search and the lockstep per-playout launches -- at both sizes;
# the schedules that measured slower and are fenced off in include/iago_hip_experimental.h -- game-asynchronous steps,
# value look-ahead -- keep one smoke each: VERDICT r04 task 8)
@pytest.mark.parametrize("n_sims,n_sims2,G,async_steps,value_ahead,persistent", [
    (100, 60, 320, False, False, True), (400, 37, 64, False, False, True),
    (100, 60, 320, False, False, False), (400, 37, 64, False, False, False),
    (400, 37, 64, True, False, False), (400, 37, 64, True, True, False)])
def test_production_search_trees_bit_exact_vs_oracle(shipped, golden_rules, n_sims, n_sims2, G, async_steps,
                                                     value_ahead, persistent):
    """async_steps=False: lockstep playouts, the default and what bench.py times.  True: the same
    search as game-asynchronous steps (iago_mcts_async: a game with a fresh leaf waits while the value
    net walks its board in 3 pieces beside the other games' steps) -- the same trees, bit for bit.
    value_ahead=True: the children of every expanded node get their values from batches on the side
    stream (iago_mcts_value_ahead) -- again the same trees.
    persistent=True: what bench.py times since round 4 -- the whole search as ONE launch in which every
    game runs on its own clock (iago_mcts_search_persistent: game workgroups + net workgroups serving a
    queue of positions; the policy net at the expansion, as the reference) -- the same trees."""
    engine, ops, policy, value, rw = shipped
    own, opp = _positions(G, golden_rules)
    cap = engine.suggest_capacity(n_sims + n_sims2, 15, moves=2)
    m = engine.BatchedMCTS(G, policy, value, rw, lmbda=0.5, c_puct=1.0, n_thr=15, capacity=cap, seed=5,
                           game_id_base=1000, use_graph=True, z_log_rows=max(n_sims, n_sims2),
                           async_steps=async_steps, value_ahead=value_ahead, persistent=persistent)
    assert m.value_ahead == value_ahead and (not value_ahead or m.n_value_ahead == 0)
    assert m.persistent == persistent and m.value_cache and policy.split3 and value.split_f16
    if not persistent:
        # the per-playout launches at their defaults
        assert m.use_graph and m.sync_free and m.lookahead == 4 and m.lookahead_overlap == 2
        assert m.fused_descent and m.fused_leaf_eval and m._la_path is not None
        assert m.async_steps == async_steps and (not async_steps or m.async_parts == 3)
        assert policy.split3_parts == 2
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    active = torch.ones(G, dtype=torch.uint8, device="cuda")
    active[5] = 0  # an idle game must stay untouched
    m.search(o, p, active, n_sims)
    assert persistent or (m._graph is not None and (n_sims < 32 or m._graph_long is not None))   # replayed, not eager
    move = m.best_move(active)[0].cpu().numpy()
    visits = m.visits.cpu().numpy()
    zn = m.z_log_n.cpu().numpy()
    zlog = m.z_log.cpu().numpy()
    assert zn[5] == 0 and np.all(np.delete(zn, 5) == n_sims)
    assert int(m.tree.n_nodes[5].item()) == 1 and int(m.tree.n_visits[5 * cap].item()) == 0
    # the value net ran on a fraction of the leaves only (the cache), the policy a few visits ahead
    assert 0 < m.n_value_evals < 0.6 * m.n_leaf_evals and m.n_policy_evals > 0
    assert persistent or (m.n_value_ahead > 0) == value_ahead   # (the persistent search walks values ahead on idle hands)

    probe = NetProbe(ops, policy, value)
    checked = [g for g in list(range(0, 12)) + list(range(G // 2 - 2, G // 2 + 6)) if g != 5]
    oracles = {}
    for g in checked:
        it = iter(zlog[:n_sims, g])
        om = mcts_py.MCTS(probe.policy_fn, probe.value_fn, lambda s, c, it=it: int(next(it)), lmbda=0.5,
                          c_puct=1.0, n_thr=15)
        want_move = om.get_move(state_of(own[g], opp[g]), 1, n_sims)
        assert next(it, None) is None
        _cmp_tree(m.tree.dump(g, max_depth=64), mcts_py.dump_tree(om.root, max_depth=64), "g%d" % g)
        if want_move is None:
            assert move[g] == -2
        else:
            assert move[g] == want_move, g
            for a, ch in om.root.children.items():
                if a >= 0:
                    assert visits[g, a] == ch.n_visits
        oracles[g] = om
    # the values stored in the nodes ARE the probe's values (a direct look at the cache)
    v_tree = m.tree.v.cpu().numpy()
    for g in checked[:6]:
        root = int(m.tree.root[g].item())
        assert v_tree[g * cap + root] == probe.value_fn(orc.make_state_var(state_of(own[g], opp[g]), 1)), g

    # MCTS.update_with_move (MCTS.py:149-154): subtree reuse, the carried value cache, a playout
    # count that is no multiple of the look-ahead block (the un-captured tail of the search)
    mv = torch.from_numpy(np.where(move == -2, -1, move).astype(np.int8)).cuda()
    m.update_with_move(mv, active.clone())
    o2, p2 = o.clone(), p.clone()
    ops.apply_moves(o2, p2, mv)
    m.z_log_n.zero_()
    m.search(p2, o2, active, n_sims2)  # the other side is to move now
    zlog = m.z_log.cpu().numpy()
    assert np.all(np.delete(m.z_log_n.cpu().numpy(), 5) == n_sims2)
    for g, om in oracles.items():
        a = int(mv[g].item())
        om.update_with_move(a)
        s = state_of(own[g], opp[g])
        orc.place_stone(s, a, 1)
        it = iter(zlog[:n_sims2, g])
        om.rollout_fn = lambda st, c, it=it: int(next(it))
        om.get_move(s, 2, n_sims2)
        _cmp_tree(m.tree.dump(g, max_depth=64), mcts_py.dump_tree(om.root, max_depth=64), "g%d'" % g)
    policy.check_saturation()
    value.check_saturation()


@pytest.mark.parametrize("async_steps", ["persistent", False, 2])
def test_production_self_play_games_vs_oracle(shipped, async_steps):
    """Whole self-play games through SelfPlayEngine at the production defaults (what bench.py's
    PV-MCTS leg times), 20 playouts per move (n_thr = 15 needs > 15): every game's move list and
    result equal the oracle's selfplay_game (game.py:117-142 turn structure) fed the recorded z."""
    engine, ops, policy, value, rw = shipped
    G, n_sims = 8, 24
    # ("persistent": whole games in ONE launch, what bench.py times; True: asynchronous steps with the default 3
    # pieces; 2 / 4: the other layer splits and queue rotations of the value net's walk)
    persistent = async_steps == "persistent"
    async_steps = False if persistent else async_steps
    parts = async_steps if async_steps not in (False, True) else None
    m = engine.BatchedMCTS(G, policy, value, rw, n_thr=15, capacity=4096, seed=11, use_graph=True,
                           z_log_rows=128 * n_sims, async_steps=bool(async_steps), async_parts=parts,
                           persistent=persistent)
    assert m.persistent == persistent and m.value_cache
    assert persistent or (m.lookahead == 4 and m.async_steps == bool(async_steps))
    assert not async_steps or m.async_parts == (parts or 3)
    res = engine.SelfPlayEngine(m).play(n_sims)
    moves = res.move.cpu().numpy()           # (T, G), -1 = pass / finished
    valid = res.valid.cpu().numpy()
    z = res.z.cpu().numpy()
    zlog, zn = m.z_log.cpu().numpy(), m.z_log_n.cpu().numpy()
    probe = NetProbe(ops, policy, value)
    for g in range(G):
        it = iter(zlog[:zn[g], g])
        om = mcts_py.MCTS(probe.policy_fn, probe.value_fn, lambda s, c, it=it: int(next(it)), lmbda=0.5,
                          c_puct=1.0, n_thr=15)
        want_moves, want_z, _ = mcts_py.selfplay_game(om, n_sims)
        got = [int(moves[t, g]) if valid[t, g] else -1 for t in range(len(want_moves))]
        assert got == want_moves, g
        assert z[g] == want_z, g
        assert next(it, None) is None, g     # the oracle consumed exactly the playouts the GPU ran


@pytest.mark.parametrize("persistent", [True, False])
@pytest.mark.parametrize("n_thr,n_sims,n_sims2", [(15, 100, 44), (1, 60, 30)])
def test_production_search_with_uniform_rollouts_needs_no_replay(shipped, golden_rules, n_thr, n_sims, n_sims2,
                                                                persistent):
    """The whole playout reproduced by the oracle ITSELF, rollouts included: with the uniform rollout
    policy (arithmetic exact in float32: tests/test_rollout_gpu.py) the oracle plays Simulate from the
    same Philox stream the kernel draws from -- stream id = the search's playout counter -- instead of
    being fed recorded results.  n_thr = 15: the production path (hipGraph, look-ahead, value cache,
    fused descent / leaf evaluation, path backup).  n_thr = 1: the configuration of bench.py's
    `mcts_nthr1` leg -- no look-ahead possible, the policy net inside every playout
    (select, pending, policy_resident_kernel on the expanding leaves, expand, continued select,
    fresh_leaves, value_rollout_kernel, mix_backup), one hipGraph replay per playout."""
    engine, ops, policy, value, _ = shipped
    G, seed, base = 48, 9, 500
    own, opp = _positions(G, golden_rules)
    cap = engine.suggest_capacity(n_sims + n_sims2, n_thr, moves=2)
    m = engine.BatchedMCTS(G, policy, value, ops.uniform_weights(), lmbda=0.5, c_puct=1.0, n_thr=n_thr, capacity=cap,
                           seed=seed, game_id_base=base, use_graph=True, persistent=persistent)
    assert m.persistent == persistent and m.value_cache
    assert persistent or (m.use_graph and m.fused_leaf_eval and m.lookahead == (4 if n_thr == 15 else 0))
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    active = torch.ones(G, dtype=torch.uint8, device="cuda")
    m.search(o, p, active, n_sims)
    move = m.best_move(active)[0].cpu().numpy()
    probe = NetProbe(ops, policy, value)
    checked = list(range(0, 8)) + list(range(G // 2 - 2, G // 2 + 4))
    oracles = {}

    def rollout_of(g, counter):
        def fn(state, color):
            z = orc.random_playout(state, color, seed=seed, game_id=base + g, stream=counter[0])[0]
            counter[0] += 1
            return z
        return fn

    for g in checked:
        counter = [0]     # the engine's playout counter: one Philox stream per playout of the engine's life
        om = mcts_py.MCTS(probe.policy_fn, probe.value_fn, rollout_of(g, counter), lmbda=0.5, c_puct=1.0, n_thr=n_thr)
        want_move = om.get_move(state_of(own[g], opp[g]), 1, n_sims)
        assert counter[0] == n_sims
        _cmp_tree(m.tree.dump(g, max_depth=64), mcts_py.dump_tree(om.root, max_depth=64), "g%d" % g)
        assert move[g] == (-2 if want_move is None else want_move), g
        oracles[g] = (om, counter)
    mv = torch.from_numpy(np.where(move == -2, -1, move).astype(np.int8)).cuda()
    m.update_with_move(mv, active.clone())
    o2, p2 = o.clone(), p.clone()
    ops.apply_moves(o2, p2, mv)
    m.search(p2, o2, active, n_sims2)
    for g, (om, counter) in oracles.items():
        a = int(mv[g].item())
        om.update_with_move(a)
        s = state_of(own[g], opp[g])
        orc.place_stone(s, a, 1)
        om.get_move(s, 2, n_sims2)
        assert counter[0] == n_sims + n_sims2
        _cmp_tree(m.tree.dump(g, max_depth=64), mcts_py.dump_tree(om.root, max_depth=64), "g%d'" % g)


@pytest.mark.parametrize("persistent", [True, False])
def test_production_self_play_uniform_rollouts_no_replay(shipped, persistent):
    """Whole PV-MCTS self-play games (SelfPlayEngine at the production defaults) reproduced by the oracle
    with nothing replayed: uniform rollout policy, the oracle plays every leaf rollout itself from the
    Philox stream of that playout -- stream id = (turn of the game) x n_sims + playout, the engine's
    playout counter (every turn of the lockstep loop runs one search of n_sims playouts)."""
    engine, ops, policy, value, _ = shipped
    G, n_sims, seed, base = 8, 24, 13, 70
    m = engine.BatchedMCTS(G, policy, value, ops.uniform_weights(), n_thr=15, capacity=4096, seed=seed,
                           game_id_base=base, use_graph=True, persistent=persistent)
    assert m.persistent == persistent and m.value_cache and (persistent or (m.lookahead == 4 and m.use_graph))
    res = engine.SelfPlayEngine(m).play(n_sims)
    moves, valid, z = res.move.cpu().numpy(), res.valid.cpu().numpy(), res.z.cpu().numpy()
    f1, f2 = ops.tensor_to_bits(res.final_p1), ops.tensor_to_bits(res.final_p2)
    probe = NetProbe(ops, policy, value)
    for g in range(G):
        counter = [0]

        def roll(state, color, g=g, counter=counter):
            zz = orc.random_playout(state, color, seed=seed, game_id=base + g, stream=counter[0])[0]
            counter[0] += 1
            return zz

        om = mcts_py.MCTS(probe.policy_fn, probe.value_fn, roll, lmbda=0.5, c_puct=1.0, n_thr=15)
        # game.py:117-142,253-255 with both colours driven by MCTS.get_move (oracle.mcts_py.selfplay_game),
        # the turn counted for the stream ids
        state = orc.initial_state()
        stone_num, pass_flg, t, want = 4, False, 0, []
        while stone_num < 64:
            for color in (1, 2):
                acts = orc.legal_actions(state, color)
                if len(acts) > 0:
                    counter[0] = t * n_sims
                    a = om.get_move(state, color, n_sims)
                    om.update_with_move(a)
                    orc.place_stone(state, a, color)
                    stone_num += 1
                    pass_flg = False
                    want.append(a)
                else:
                    if pass_flg:
                        stone_num = 64
                    pass_flg = True
                    om.update_with_move(-1)
                    want.append(-1)
                t += 1
        got = [int(moves[k, g]) if valid[k, g] else -1 for k in range(len(want))]
        assert got == want, g
        assert z[g] == orc.judge(state, 1) and orc.state_to_bits(state) == (int(f1[g]), int(f2[g])), g
