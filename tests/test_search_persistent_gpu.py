"""The persistent search (iago_mcts_search_persistent, include/iago_hip.h; engine.BatchedMCTS(persistent=True)):
a whole search -- the loop of MCTS.get_move around MCTS.playout (MCTS.py:105-147) -- as ONE launch in
which every game runs on its own clock, the value and the policy net serving a queue of positions.  What
must hold: the trees (visit counts, float32 Q and P, child order, pool layout), the values stored in the
nodes, the rollout results every playout backed up and the number of net evaluations a search NEEDS are
those of the per-playout launches -- only the interleaving between games differs.  The comparison with
the oracle's restatement of MCTS.py runs in tests/test_mcts_production_gpu.py (persistent=True there).
"""
import numpy as np
import pytest
import torch

from tests.conftest import load_json
from tests.gpu_util import random_positions

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


def _positions(G):
    own, opp = random_positions(G, seed=78)
    own[: G // 2] = 0x0000000810000000
    opp[: G // 2] = 0x0000001008000000
    return own, opp


def _run(nets, G, n_sims, n_sims2, own, opp, idle=(), **kw):
    engine, ops, policy, value, rw = nets
    zrows = kw.pop("z_log_rows", n_sims + n_sims2)
    m = engine.BatchedMCTS(G, policy, value, rw, n_thr=kw.pop("n_thr", 15),
                           capacity=engine.suggest_capacity(n_sims + n_sims2, 15, moves=2), seed=21, game_id_base=300,
                           z_log_rows=zrows, **kw)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    active = torch.ones(G, dtype=torch.uint8, device="cuda")
    for g in idle:
        active[g] = 0
    m.search(o, p, active, n_sims)
    if n_sims2:
        mv = m.best_move(active)[0].clone()
        mv = torch.where(mv == -2, torch.full_like(mv, -1), mv)
        m.update_with_move(mv, active.clone())
        ops.apply_moves(o, p, mv)
        m.search(p, o, active, n_sims2)
    t = m.tree
    out = {k: getattr(t, k).cpu().numpy().copy() for k in ("n_visits", "q", "p", "first_child", "parent", "action",
                                                           "n_children", "n_nodes", "root", "v")}
    # (an unexpanded leaf the look-ahead has queued carries its prior-cache tag <= -2 in first_child: a leaf
    # either way)
    out["first_child"] = np.where(out["first_child"] < 0, -1, out["first_child"])
    if getattr(m, "z_log", None) is not None:
        out["z_log"], out["z_log_n"] = m.z_log.cpu().numpy().copy(), m.z_log_n.cpu().numpy().copy()
    out["leaf_value"] = m.leaf_value.cpu().numpy().copy()
    return m, out


@pytest.mark.parametrize("G,n_sims,n_sims2,net", [(1, 60, 30, 2), (16, 40, 0, 3), (20, 100, 45, 8), (96, 100, 45, None),
                                                  (330, 60, 20, 40)])
def test_trees_equal_the_per_playout_engine(nets, G, n_sims, n_sims2, net):
    own, opp = _positions(G)
    idle = (5,) if G > 16 else ()
    a, ta = _run(nets, G, n_sims, n_sims2, own, opp, idle, persistent=True, net_workgroups=net)
    b, tb = _run(nets, G, n_sims, n_sims2, own, opp, idle, persistent=False, use_graph=True)
    assert a.persistent and not b.persistent and b.lookahead == 4
    for k in ("n_visits", "q", "p", "first_child", "parent", "action", "n_children", "n_nodes", "root"):
        assert np.array_equal(ta[k], tb[k]), k
    # the stored values: the same numbers in the same nodes (NaN = never evaluated, in both)
    assert np.array_equal(np.isnan(ta["v"]), np.isnan(tb["v"]))
    assert np.array_equal(ta["v"][~np.isnan(ta["v"])], tb["v"][~np.isnan(tb["v"])])
    # every playout of every game backed up the same rollout result (same Philox stream per playout)
    assert np.array_equal(ta["z_log_n"], tb["z_log_n"]) and np.array_equal(ta["z_log"], tb["z_log"])
    assert a.n_leaf_evals == b.n_leaf_evals
    # the value net ran once per leaf that needed it -- the per-playout engine's count -- and the policy net
    # once per expansion with more than one legal move: never more than the look-ahead's batches evaluated
    # (fewer where the position table answered: positions another game had asked for)
    hits = int(a._ps["totals"][8].item())
    assert a.n_value_inline + hits == b.n_value_evals and a.n_value_evals > 0   # (every fresh leaf: asked for, or in the table)
    assert 0 < a.n_policy_evals <= b.n_policy_evals
    for g in idle:
        assert ta["n_nodes"][g] == 1 and ta["z_log_n"][g] == 0
    a.value_fn.check_saturation()
    a.policy_fn.check_saturation()


@pytest.mark.parametrize("lmbda,c_puct,n_thr,G,n_sims,n_sims2", [(0.0, 1.0, 15, 40, 80, 30), (0.3, 0.7, 3, 33, 50, 25),
                                                               (0.9, 2.5, 7, 65, 64, 0), (0.5, 1.0, 15, 9, 14, 16)])
def test_other_search_parameters(nets, lmbda, c_puct, n_thr, G, n_sims, n_sims2):
    """MCTS(lmbda, c_puct, n_thr) (MCTS.py:78-82) away from the defaults -- lmbda = 0 (no rollout at all: the leaf's
    value alone), a small n_thr (expansions all the time), a search shorter than n_thr (no expansion in the first
    search) -- : the same trees as the per-playout launches."""
    own, opp = _positions(G)
    kw = dict(lmbda=lmbda, c_puct=c_puct, n_thr=n_thr, z_log_rows=0)
    a, ta = _run(nets, G, n_sims, n_sims2, own, opp, persistent=True, **kw)
    b, tb = _run(nets, G, n_sims, n_sims2, own, opp, persistent=False, use_graph=True, **kw)
    assert a.persistent and not b.persistent
    for k in ("n_visits", "q", "p", "first_child", "parent", "action", "n_children", "n_nodes", "root", "leaf_value"):
        assert np.array_equal(ta[k], tb[k]), k
    assert np.array_equal(np.isnan(ta["v"]), np.isnan(tb["v"]))
    assert np.array_equal(ta["v"][~np.isnan(ta["v"])], tb["v"][~np.isnan(tb["v"])])
    assert a.n_leaf_evals == b.n_leaf_evals == G * (n_sims + n_sims2)


def test_values_ahead_on_idle_net_workgroups(nets, monkeypatch):
    """While net workgroups poll and nothing waits, the children of a node whose priors a game asks for are walked
    through the value net for the position table, ahead of their first visits (MCTS.value_func is a pure function of the
    position, MCTS.py:97-103).  Timing only: the same trees and stored values; and it does take evaluations off the
    games' critical path (here: 48 games against 256 CUs, the net workgroups poll most of the time)."""
    G, n_sims, n_sims2 = 48, 120, 40
    own, opp = _positions(G)
    monkeypatch.setenv("IAGO_PERSISTENT_AHEAD", "-1")
    a, ta = _run(nets, G, n_sims, n_sims2, own, opp, persistent=True)
    monkeypatch.setenv("IAGO_PERSISTENT_AHEAD", "1")
    b, tb = _run(nets, G, n_sims, n_sims2, own, opp, persistent=True)
    for k in ("n_visits", "q", "p", "first_child", "parent", "action", "n_children", "n_nodes", "root", "leaf_value", "z_log"):
        assert np.array_equal(ta[k], tb[k]), k
    assert np.array_equal(np.isnan(ta["v"]), np.isnan(tb["v"]))
    assert np.array_equal(ta["v"][~np.isnan(ta["v"])], tb["v"][~np.isnan(tb["v"])])
    assert a.n_value_ahead == 0 and b.n_value_ahead > 0
    assert b.n_value_inline < 0.5 * a.n_value_inline          # most first visits find their value in the table
    hits_a, hits_b = int(a._ps["totals"][8].item()), int(b._ps["totals"][8].item())
    assert a.n_value_inline + hits_a == b.n_value_inline + hits_b   # the same fresh leaves either way


def test_values_ahead_stay_within_the_ring(nets):
    """The requests nobody waits for share the value ring with the games' own (at most one per waiting game): they are
    limited to half the ring per iteration, and batches of more than half the ring's games send none."""
    engine, ops, policy, value, rw = nets
    from iago_amd import _lib
    res = {}
    for G in (_lib.SEARCH_QUEUE_ENTRIES // 2, _lib.SEARCH_QUEUE_ENTRIES // 2 + 32):
        own, opp = _positions(G)
        # (split=0: the single launch -- when net workgroups idle in a search this short depends on the launch form; the
        # ring's budget, which this test is about, does not)
        m = engine.BatchedMCTS(G, policy, value, rw, n_thr=15, capacity=engine.suggest_capacity(20, 15, moves=2), seed=4,
                               persistent=True, split=0)
        m.search(ops.bits_to_tensor(own), ops.bits_to_tensor(opp), torch.ones(G, dtype=torch.uint8, device="cuda"), 20)
        res[G] = m.n_value_ahead
        assert m.n_leaf_evals == 20 * G
        m.close()
    assert res[_lib.SEARCH_QUEUE_ENTRIES // 2] > 0 and res[_lib.SEARCH_QUEUE_ENTRIES // 2 + 32] == 0


def test_n_thr_one(nets):
    """n_thr = 1 (SURVEY 8d's second datapoint): every leaf expands at its second visit -- the policy net
    inside every playout."""
    G, n_sims = 48, 40
    own, opp = _positions(G)
    a, ta = _run(nets, G, n_sims, 0, own, opp, n_thr=1, persistent=True)
    b, tb = _run(nets, G, n_sims, 0, own, opp, n_thr=1, persistent=False, use_graph=True, z_log_rows=0)
    assert b.lookahead == 0   # (no visits to run ahead of: the per-playout engine, too, evaluates at the expansion)
    for k in ("n_visits", "q", "p", "first_child", "parent", "action", "n_children", "n_nodes", "leaf_value"):
        assert np.array_equal(ta[k], tb[k]), k
    assert np.all(ta["z_log_n"] == n_sims)
    # (the per-playout engine counts every expanding leaf, this one the leaves with more than one legal move)
    assert 0 < a.n_policy_evals <= b.n_policy_evals and a.n_value_inline + int(a._ps["totals"][8].item()) == b.n_value_evals


@pytest.mark.parametrize("games", ["1", "0"])
def test_whole_games(nets, games, monkeypatch):
    """SelfPlayEngine on the persistent search = on the per-playout launches: moves, visit counts, recorded
    positions, results.  games = "1": the whole game of every board in ONE launch (every game walks through its
    own turns); "0": one persistent launch per turn."""
    engine, ops, policy, value, rw = nets
    monkeypatch.setenv("IAGO_PERSISTENT_GAMES", games)
    res = {}
    hc = torch.zeros(40, dtype=torch.int64, device="cuda")
    hc[1::2] = 1 << (2 * 8 + 4)          # the handicap stone of src/train_rl.py:43-46 on every other board
    for persistent in (True, False):
        m = engine.BatchedMCTS(40, policy, value, rw, n_thr=15, capacity=4096, seed=11, use_graph=True,
                               persistent=persistent, z_log_rows=128 * 24)
        r = engine.SelfPlayEngine(m).play(24, handicap=hc)
        t = r.tuples()
        res[persistent] = dict(move=r.move.cpu().numpy(), pi=r.pi.cpu().numpy(), z=r.z.cpu().numpy(),
                               valid=r.valid.cpu().numpy(), own=r.own.cpu().numpy(), opp=r.opp.cpu().numpy(),
                               p1=r.final_p1.cpu().numpy(), p2=r.final_p2.cpu().numpy(), n_turns=r.n_turns,
                               tz=t["z"].cpu().numpy(), tgame=t["game"].cpu().numpy(),
                               zlog=m.z_log.cpu().numpy(), zn=m.z_log_n.cpu().numpy(), leaf=m.n_leaf_evals,
                               sim=m.sim_counter)
    a, b = res[True], res[False]
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert a["n_turns"] % 2 == 0 and a["valid"].sum() > 40 * 50


@pytest.mark.parametrize("pace,backlog,gpw,net,defer", [("-1", "128", "32", None, "0"), ("1", "0", "32", 6, "15"),
                                                        ("16", "4", "16", 24, "3"), ("16", "128", "24", None, "15")])
def test_scheduling_knobs_do_not_change_the_games(nets, pace, backlog, gpw, net, defer, monkeypatch):
    """Pacing of the leading games (held while requests queue), the games per game workgroup, the number of net
    workgroups and the rollouts put off to the next iteration's pass (0: never; 15: whatever a full pass of 16 leaves
    over) decide WHEN a game's playouts run, never what they are: whole games equal the default schedule's in every
    move, visit count and rollout result.  (pace 1 / backlog 0: a game one playout ahead of the mean holds whenever
    anything waits -- the pacing at its most intrusive.)"""
    engine, ops, policy, value, rw = nets
    res = []
    for variant in (False, True):
        if variant:
            monkeypatch.setenv("IAGO_PERSISTENT_PACE", pace)
            monkeypatch.setenv("IAGO_PERSISTENT_PACE_BACKLOG", backlog)
            monkeypatch.setenv("IAGO_PERSISTENT_GPW", gpw)
            monkeypatch.setenv("IAGO_PERSISTENT_ROLL_DEFER", defer)
        m = engine.BatchedMCTS(72, policy, value, rw, n_thr=15, capacity=2048, seed=5, persistent=True,
                               net_workgroups=net if variant else None, z_log_rows=64 * 30)
        assert m.games_per_workgroup == (int(gpw) if variant else 8)      # (72 games: 9 game workgroups of 8 by default)
        r = engine.SelfPlayEngine(m).play(30)
        res.append(dict(move=r.move.cpu().numpy(), pi=r.pi.cpu().numpy(), z=r.z.cpu().numpy(), valid=r.valid.cpu().numpy(),
                        p1=r.final_p1.cpu().numpy(), p2=r.final_p2.cpu().numpy(), n_turns=r.n_turns,
                        zlog=m.z_log.cpu().numpy(), zn=m.z_log_n.cpu().numpy(), leaf=m.n_leaf_evals,
                        n_nodes=m.tree.n_nodes.cpu().numpy()))
        m.close()
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
    assert res[0]["valid"].sum() > 72 * 50


def test_whole_games_below_n_thr_raise(nets):
    engine, ops, policy, value, rw = nets
    m = engine.BatchedMCTS(16, policy, value, rw, n_thr=15, capacity=1024, seed=1, persistent=True)
    with pytest.raises(ValueError):
        engine.SelfPlayEngine(m, max_turns=4).play(8)      # MCTS.get_move's max() of an empty dict (MCTS.py:147)


def test_clock_limit_ends_the_launch(nets):
    """Every wait in the launch is a bounded poll under a clock limit (a launch whose net workgroups never become
    resident must not hang the device): with a limit far below what the search needs the launch ends by itself, the
    engine reports it, and the device goes on working."""
    engine, ops, policy, value, rw = nets
    from iago_amd import _lib
    own, opp = _positions(512)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    active = torch.ones(512, dtype=torch.uint8, device="cuda")
    m = engine.BatchedMCTS(512, policy, value, rw, n_thr=15, capacity=engine.suggest_capacity(400, 15, moves=2), seed=2,
                           persistent=True)
    m.time_limit_ms = 1                       # 512 games x 400 playouts take ~15 ms
    with pytest.raises(_lib.IagoError, match="gave up"):
        m.search(o, p, active, 400)
    done = m._ps["done"].cpu().numpy()
    assert done.min() < 400                   # (it really was cut short)
    m.close()
    # the same search with the default limit, on the same device, right after
    m = engine.BatchedMCTS(512, policy, value, rw, n_thr=15, capacity=engine.suggest_capacity(400, 15, moves=2), seed=2,
                           persistent=True)
    m.search(o, p, active, 400)
    nv = m.tree.n_visits.view(512, -1).cpu().numpy()
    assert (nv[np.arange(512), m.tree.root.cpu().numpy()] == 400).all()
    m.close()


def test_too_many_games_take_the_per_playout_launches(nets):
    """More than 4096 games would leave the launch's game workgroups no net workgroup beside them: the default is the
    per-playout engine there, and asking for the persistent one raises."""
    engine, ops, policy, value, rw = nets
    m = engine.BatchedMCTS(4128, policy, value, rw, n_thr=15, capacity=64)
    assert not m.persistent
    with pytest.raises(ValueError):
        engine.BatchedMCTS(4128, policy, value, rw, n_thr=15, capacity=64, persistent=True)
    m2 = engine.BatchedMCTS(4096, policy, value, rw, n_thr=15, capacity=64)
    # (128 game workgroups: the role split by default -- two per CU on 64 CUs, a net workgroup on each of the other 192;
    # the single launch, one game workgroup per CU, has 128 left)
    assert m2.persistent and (m2.split_cus, m2.net_workgroups) in ((64, 192), (0, 128))
    m3 = engine.BatchedMCTS(4096, policy, value, rw, n_thr=15, capacity=64, split=0)
    assert m3.persistent and m3._split is None and m3.net_workgroups == 128


def _cu_masked_stream(n_cus):
    """A HIP stream whose kernels may only use `n_cus` of the device's CUs (hipExtStreamCreateWithCUMask), as a
    torch stream.  The mask enables every (256 / n_cus)-th CU so that the CUs come from all XCDs."""
    import ctypes as C
    import os
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    total = torch.cuda.get_device_properties(0).multi_processor_count
    step = total // n_cus
    words = (C.c_uint32 * ((total + 31) // 32))()
    for cu in range(0, total, step):
        words[cu // 32] |= 1 << (cu % 32)
    stream = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(stream), C.c_uint32(len(words)), words)
    assert rc == 0 and stream.value
    return hip, stream, torch.cuda.ExternalStream(stream.value)


def test_grid_follows_the_device(nets):
    """The launch sizes its grid from the device (iago_mcts_search_capacity: CUs x workgroups of the kernel per CU) and
    from max_cus, the CUs it may count on: on a stream masked to 64 of the CUs a search told so (max_cus=64) runs with
    64 - game workgroups net workgroups, builds the same trees as on the whole device and does not run into the
    clock limit; a launch whose game workgroups + one net workgroup do not fit is refused (IAGO_ERR_CAPACITY), and
    the engine's default there is the per-playout engine."""
    engine, ops, policy, value, rw = nets
    from iago_amd import _lib
    import ctypes as C
    cus, per = C.c_int32(0), C.c_int32(0)
    assert _lib.lib().iago_mcts_search_capacity(C.byref(cus), C.byref(per)) == 0
    props = torch.cuda.get_device_properties(0)
    assert cus.value == props.multi_processor_count and per.value == 1     # 512 registers, 70 KB of LDS: one per CU
    G, n_sims, n_sims2 = 96, 100, 45
    own, opp = _positions(G)
    m0, whole = _run(nets, G, n_sims, n_sims2, own, opp, persistent=True)
    assert m0.resident_workgroups == cus.value and m0.net_workgroups == cus.value - 12
    assert int(m0._ps["ctl"][7].item()) == m0.net_workgroups               # what the launch really ran with
    hip, raw, masked = _cu_masked_stream(64)
    try:
        with torch.cuda.stream(masked):
            m1, part = _run(nets, G, n_sims, n_sims2, own, opp, persistent=True, max_cus=64)
            masked.synchronize()
        assert m1.resident_workgroups == 64 and m1.net_workgroups == 64 - 12
        assert int(m1._ps["ctl"][7].item()) == 52 and int(m1._ps["ctl"][3].item()) == 0   # no clock-limit abort
        for k in whole:
            assert np.array_equal(whole[k], part[k], equal_nan=True), k
        # told nothing, the launch asks for the whole device's net workgroups; those the masked stream has no CU for
        # start when the others end -- at the end of the launch: same trees, no abort
        with torch.cuda.stream(masked):
            m2, blind = _run(nets, G, n_sims, n_sims2, own, opp, persistent=True)
            masked.synchronize()
        assert int(m2._ps["ctl"][3].item()) == 0
        for k in whole:
            assert np.array_equal(whole[k], blind[k], equal_nan=True), k
    finally:
        torch.cuda.synchronize()
        hip.hipStreamDestroy(raw)
    # 12 game workgroups + 1 net workgroup need 13 CUs: refused by the library, never launched
    m3 = engine.BatchedMCTS(G, policy, value, rw, n_thr=15, capacity=1024, seed=21, persistent=True, max_cus=24)
    m3.max_cus = 12
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    with pytest.raises(_lib.IagoError, match="do not fit"):
        m3.search(o, p, torch.ones(G, dtype=torch.uint8, device="cuda"), 20)
    # ... and the engine does not choose the persistent search where the games would take more than half the CUs
    m4 = engine.BatchedMCTS(G, policy, value, rw, n_thr=15, capacity=1024, max_cus=16)
    assert not m4.persistent
    with pytest.raises(ValueError):
        engine.BatchedMCTS(G, policy, value, rw, n_thr=15, capacity=1024, persistent=True, max_cus=16)


def test_whole_games_fall_back_to_the_turn_loop_when_a_pool_is_small(nets, monkeypatch):
    """The one-launch whole-game path cannot compact its pools: SelfPlayEngine.play takes it only for
    pools that can hold a whole game, replays the batch through the turn loop (whose searches compact) when a pool
    fills up all the same -- and the games are the same either way."""
    engine, ops, policy, value, rw = nets
    G, n_sims = 16, 60
    big = engine.BatchedMCTS(G, policy, value, rw, n_thr=15, capacity=engine.suggest_capacity(n_sims, 15), seed=9,
                             persistent=True)
    e0 = engine.SelfPlayEngine(big)
    want = e0.play(n_sims)
    assert getattr(e0, "n_replayed", 0) == 0 and want.game_turns is not None
    used = int(big.tree.n_nodes.max().item())
    assert used > 300

    def check(cap, replays):
        small = engine.BatchedMCTS(G, policy, value, rw, n_thr=15, capacity=cap, seed=9, persistent=True)
        e1 = engine.SelfPlayEngine(small)
        got = e1.play(n_sims)
        assert getattr(e1, "n_replayed", 0) == replays and got.game_turns is None
        assert got.n_turns == want.n_turns and small.n_leaf_evals == big.n_leaf_evals
        for k in ("own", "opp", "valid", "move", "pi", "z"):
            assert torch.equal(getattr(got, k), getattr(want, k)), (cap, k)
        assert small.n_compactions > 0
        assert small.n_value_evals >= big.n_value_inline      # (nothing of a failed attempt is counted twice)
        small.close()

    # a pool below half of suggest_capacity: the turn loop from the start
    check(256, 0)
    # a pool that passes play()'s test (made lenient here) but cannot hold what these games leave behind: the launch
    # reports the full pool and the batch is replayed turn by turn, from the same Philox streams
    monkeypatch.setattr(engine, "suggest_capacity", lambda *a, **k: 64)
    check(256, 1)
    big.close()


def test_role_split_builds_the_same_trees(nets):
    """The persistent search split by role (iago_mcts_search_split -- the game workgroups as a launch of their own, two
    per CU, on a stream masked to `split` CUs; the net workgroups on a stream masked to all the others; the engine's
    default beyond 32 game workgroups): the same trees, values, rollout results as the single launch, no clock-limit
    abort, one net workgroup per CU that is not the games'."""
    engine, ops, policy, value, rw = nets
    G, n_sims, n_sims2 = 96, 100, 45
    own, opp = _positions(G)
    a, ta = _run(nets, G, n_sims, n_sims2, own, opp, (5,), persistent=True)
    b, tb = _run(nets, G, n_sims, n_sims2, own, opp, (5,), persistent=True, split=8)
    if b._split is None:
        pytest.skip("this runtime gives no CU-masked streams")
    # (one net workgroup per CU that is not the games', at most 7/8 of the device's: the library's cap)
    assert a._split is None and b.split_cus == 8
    assert b.net_workgroups == b.resident_workgroups - max(8, b.resident_workgroups // 8)
    assert int(b._ps["ctl"][3].item()) == 0 and int(b._ps["ctl"][7].item()) == b.net_workgroups
    for k in ("n_visits", "q", "p", "first_child", "parent", "action", "n_children", "n_nodes", "root", "leaf_value", "z_log"):
        assert np.array_equal(ta[k], tb[k]), k
    # whole games: 256 games x 30 playouts, the split launch against the single one, record for record
    res = []
    for split in (0, 16):
        m = engine.BatchedMCTS(256, policy, value, rw, n_thr=15, capacity=engine.suggest_capacity(30, 15, moves=64), seed=9,
                               persistent=True, split=split)
        r = engine.SelfPlayEngine(m).play(30, record=True)
        res.append((r.move.cpu().numpy().copy(), r.pi.cpu().numpy().copy(), r.z.cpu().numpy().copy(), r.n_turns))
        assert (m._split is not None) == bool(split)
        m.close()
    assert res[0][3] == res[1][3]
    for x, y in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(x, y)
