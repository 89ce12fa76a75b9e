"""The batch bench.py times, checked at ITS size on the engine that times it (task 2).

bench.py's headline step is ONE launch of the persistent search kernel: 1024 PV-MCTS self-play games x 100
playouts per move, random-init SLPolicy / Value (seed 0), shipped RolloutPolicy weights, engine seed 7 -- with
the pacing engaged, 88 % pair walks, ~150 k position-table hits and ~0.9 M ring requests.  The oracle tests of
tests/test_mcts_production_gpu.py run the same kernel at 8 - 330 games; here the very batch of the bench is

  (i)   replayed record by record through the C oracle (oracle/othello_oracle.c): every recorded position is
        the oracle's position, the mover's legal set decides search / pass, the move is legal and is the first
        most-visited root child, the stone / pass / double-pass / stone_num books of game.py:117-142,253-255
        end the game at the recorded turn, the result is the oracle's judge (rl_env.py:141-149);
  (ii)  rebuilt search by search with oracle/mcts_py.MCTS (MCTS.py:105-154) for sampled games -- the first turns
        of 32 games and 3 whole games -- fed the rollout results the launch itself backed up (z_log) and the
        nets' outputs from the one-board kernels: every root's visit counts by action and every move equal;
  (iii) audited: after the batch every occupied entry of the position table is consistent (even sequence word,
        value word tagged with it) and the values of >= 4,096 of them are bit-equal to the one-board walk of the
        Value net on the stored position (a torn entry would silently change a tree);
  (iv)  and the same launch at 400 playouts per move (one GPU's share of BASELINE configs[3], the engine the
        bench's `mcts400` leg times) for its first 8 turns: records through the oracle, sampled searches
        rebuilt, table audited.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mcts_py
from oracle import oracle as orc
from tests.conftest import GOLDEN
from tests.bench_batch_util import Probe as _Probe, rebuild as _rebuild

pytestmark = pytest.mark.gpu

START = (0x0000000810000000, 0x0000001008000000)


def _engine(n_sims, max_turns, z_rows, n_thr=15):
    """bench.mcts_leg's engine, to the letter (nets, weights, seeds, capacity, defaults)."""
    from iago_amd import engine, network, ops
    g = json.load(open(os.path.join(GOLDEN, "simulate.json")))
    w, b = np.asarray(g["shipped_w"], np.float32), np.asarray(g["shipped_b"], np.float32)
    torch.manual_seed(0)
    policy = network.SLPolicy().cuda().eval()
    value = network.Value().cuda().eval()
    value.split_f16 = True
    m = engine.BatchedMCTS(1024, policy, value, ops.RolloutWeights(w, b), lmbda=0.5, c_puct=1.0, n_thr=n_thr, seed=7,
                           game_id_base=0, persistent=True, z_log_rows=z_rows,
                           capacity=engine.suggest_capacity(n_sims, n_thr, moves=64 if max_turns > 64 else max_turns + 4))
    assert m.persistent and m.value_cache and m.games_per_workgroup == 32
    eng = engine.SelfPlayEngine(m, max_turns=max_turns)
    m.warmup()
    return engine, ops, policy, value, m, eng


def _play(n_sims, max_turns, z_rows, n_thr=15):
    engine, ops, policy, value, m, eng = _engine(n_sims, max_turns, z_rows, n_thr)
    m._ps["totals"].zero_()
    res = eng.play(n_sims, record=True)
    assert getattr(eng, "n_replayed", 0) == 0        # the one-launch path, not the turn loop
    assert int(m.tree.overflow.sum().item()) == 0    # no pool filled up, no descent deeper than the path buffer
    out = dict(ops=ops, policy=policy, value=value, n_sims=n_sims, n_thr=n_thr, T=res.n_turns, path_stride=m.PATH_STRIDE,
               own=ops.tensor_to_bits(res.own), opp=ops.tensor_to_bits(res.opp),
               valid=res.valid.cpu().numpy(), move=res.move.cpu().numpy(), pi=res.pi.cpu().numpy(),
               z=res.z.cpu().numpy(), game_turns=res.game_turns.cpu().numpy(),
               f1=ops.tensor_to_bits(res.final_p1), f2=ops.tensor_to_bits(res.final_p2),
               zlog=m.z_log.cpu().numpy(), zn=m.z_log_n.cpu().numpy(),
               table=m._vtable.cpu().numpy().view(np.uint64).reshape(-1, 4),
               totals=[int(x) for x in m._ps["totals"].tolist()], leaf_evals=m.n_leaf_evals)
    m.close()
    return out


@pytest.fixture(scope="module")
def batch100():
    return _play(100, 128, 128 * 100)


@pytest.fixture(scope="module")
def batch400():
    return _play(400, 8, 8 * 400)


@pytest.fixture(scope="module")
def whole400():
    """What bench.py's `mcts400` leg times: ONE launch of 1024 WHOLE games x 400 playouts per move."""
    return _play(400, 128, 128 * 400)


@pytest.fixture(scope="module")
def whole_nthr1():
    """What bench.py's `mcts_nthr1` leg times: ONE launch of 1024 whole games x 100 playouts at n_thr = 1."""
    return _play(100, 128, 128 * 100, n_thr=1)


def _rebuild_in_workers(B, jobs, tmp_path, n_workers=4):
    """jobs: [(game, turns to rebuild, compare from turn)] spread over worker processes (tests/rebuild_worker.py: the
    Python restatement of MCTS.py with the nets' outputs from the production kernels on one board, each worker its
    own HIP context).  Returns (searches compared, {game: deepest path})."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for w in range(n_workers):
        mine = jobs[w::n_workers]
        if not mine:
            continue
        arrays = dict(n_sims=B["n_sims"], n_thr=B["n_thr"], games=np.array([j[0] for j in mine]),
                      n_turns=np.array([j[1] for j in mine]), compare_from=np.array([j[2] for j in mine]))
        for g, _, _ in mine:
            arrays["pi_%d" % g], arrays["move_%d" % g] = B["pi"][:, g], B["move"][:, g]
            arrays["zlog_%d" % g] = B["zlog"][:B["zn"][g], g]
            arrays["game_turns_%d" % g] = B["game_turns"][g]
        src, dst = os.path.join(str(tmp_path), "job%d.npz" % w), os.path.join(str(tmp_path), "job%d.json" % w)
        np.savez(src, **arrays)
        procs.append((subprocess.Popen([sys.executable, os.path.join(root, "tests", "rebuild_worker.py"), src, dst], cwd=root), dst))
    n, depth = 0, {}
    for p, dst in procs:
        rc = p.wait(timeout=600)
        out = json.load(open(dst))
        assert rc == 0 and "error" not in out, out.get("error")
        n += sum(out["compared"].values())
        depth.update({int(g): d for g, d in out["max_path"].items()})
    return n, depth


def _replay_records(B, whole):
    """(i): every record of every game through the C oracle.  Returns the number of records checked."""
    own, opp, valid, move, pi, T, n_sims = B["own"], B["opp"], B["valid"], B["move"], B["pi"], B["T"], B["n_sims"]
    n_rec = 0
    for g in range(1024):
        state = orc.initial_state()
        stone_num, pass_flg, t = 4, False, 0
        over = False
        while not over and t < T:
            for color in (1, 2):
                p1, p2 = orc.state_to_bits(state)
                mover = (p1, p2) if color == 1 else (p2, p1)
                assert (int(own[t, g]), int(opp[t, g])) == mover, (g, t)
                acts = orc.legal_actions(state, color)
                if len(acts) > 0:
                    assert valid[t, g] == 1, (g, t)
                    a = int(move[t, g])
                    assert a in acts, (g, t, a)
                    row = pi[t, g]
                    assert np.all(row[[x for x in range(64) if x not in acts]] == 0), (g, t)
                    assert a == int(np.argmax(row)) and row[a] > 0, (g, t)      # first most-visited child (MCTS.py:147)
                    # the root was a leaf for its first visits, then every playout went to a child (MCTS.py:109)
                    assert int(row.sum()) >= n_sims - B["n_thr"], (g, t)   # (+ the visits the reused subtree brought)
                    orc.place_stone(state, a, color)
                    stone_num += 1
                    pass_flg = False
                else:
                    assert valid[t, g] == 0 and move[t, g] == -1 and not pi[t, g].any(), (g, t)
                    if pass_flg:
                        stone_num = 64
                    pass_flg = True
                n_rec += 1
                t += 1
                if t >= T:
                    break
            if stone_num >= 64:
                over = True
        if whole:
            assert over and B["game_turns"][g] == t, (g, t, B["game_turns"][g])
            assert B["z"][g] == orc.judge(state, 1), g
            assert orc.state_to_bits(state) == (int(B["f1"][g]), int(B["f2"][g])), g
        else:
            assert B["game_turns"][g] == T, g
    return n_rec


def _audit_table(B, n_walk):
    """(iii): the position table after the batch."""
    tab = B["table"]
    seq, own, opp, val = tab[:, 0], tab[:, 1], tab[:, 2], tab[:, 3]
    used = np.nonzero(seq)[0]
    assert len(used) > 1000
    assert not np.any(seq[used] & np.uint64(1)), "an entry was left with an odd sequence word"
    assert np.array_equal(val[used] >> np.uint64(32), seq[used] & np.uint64(0xFFFFFFFF)), "value word / sequence word mismatch"
    writer = (seq[used] >> np.uint64(32)).astype(np.int64)
    assert writer.min() >= 0 and writer.max() < 1024           # the game that asked (or walked ahead)
    assert not np.any(own[used] & opp[used])                     # positions: disjoint stones, the centre occupied
    rs = np.random.RandomState(5)
    pick = used if len(used) <= n_walk else rs.choice(used, n_walk, replace=False)
    ops, value = B["ops"], B["value"]
    o, p = ops.bits_to_tensor(own[pick]), ops.bits_to_tensor(opp[pick])
    idx = torch.arange(len(pick), dtype=torch.int64, device="cuda")
    one = torch.ones(1, dtype=torch.int32, device="cuda")
    out = torch.full((len(pick),), float("nan"), dtype=torch.float32, device="cuda")
    with torch.no_grad():
        for i in range(len(pick)):      # ONE board per launch: the one-board walk (value to out[index[0]])
            value.forward_boards_counted(o, p, idx[i:i + 1], one, out)
    got = out.cpu().numpy().view(np.uint32)
    want = (val[pick] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, "%d of %d table values differ from the one-board walk (first: slot %d)" % (
        len(bad), len(pick), int(pick[bad[0]]))
    return len(used), len(pick)


def test_bench_batch_records_through_the_oracle(batch100):
    B = batch100
    n = _replay_records(B, whole=True)
    assert n > 1024 * 58 and B["leaf_evals"] == int(B["valid"].sum()) * 100
    t = B["totals"]
    # the regime of the bench: the nets behind the rings, pairs, the table, the values ahead -- all in use
    assert t[0] > 500_000 and t[1] > 100_000 and t[3] > 100_000 and t[8] > 50_000 and 0 < t[12] < t[8]


def test_bench_batch_searches_rebuilt_by_the_oracle(batch100):
    B = batch100
    probe = _Probe(B)
    n = 0
    for g in range(5, 1024, 32):       # 32 games, their first 6 turns
        n += _rebuild(B, probe, g, 6)
    for g in (0, 511, 1023):           # 3 whole games (incl. the endgame's pass chains)
        n += _rebuild(B, probe, g, 128)
    assert n >= 32 * 6 + 3 * 55


def test_bench_batch_position_table_audit(batch100):
    used, walked = _audit_table(batch100, 4096)
    assert used > 200_000 and walked == 4096


def test_config3_share_on_the_persistent_search(batch400):
    """(iv) 1024 games x 400 playouts per move, first 8 turns, on the engine the bench's mcts400 leg times."""
    B = batch400
    assert _replay_records(B, whole=False) == 1024 * 8
    probe = _Probe(B)
    n = sum(_rebuild(B, probe, g, 8) for g in range(3, 1024, 128))     # 8 games x 8 searches of 400 playouts
    assert n == 8 * 8
    used, walked = _audit_table(B, 4096)
    assert used > 10_000 and walked == 4096
    assert B["leaf_evals"] == 1024 * 8 * 400


def test_whole_games_at_400_playouts_through_the_oracle(whole400, tmp_path):
    """VERDICT r05, hole (i): the launch the bench's `mcts400` leg times, whole games.  All records of the 1024 games
    through the C oracle; 2 whole games and the LAST 6 searches of 8 more rebuilt by oracle/mcts_py.MCTS -- the endgame
    of a 400-playout game is where the kernel's pass-chain rule lives (MCTS.py:109-117: under a finished game a pass
    child under a pass child, one level every n_thr visits; the last turns descend through ~65 levels per playout) and
    where the recorded path is longest: the oracle's own deepest path must fit the path buffer; the table audited."""
    B = whole400
    n = _replay_records(B, whole=True)
    assert n > 1024 * 58 and B["leaf_evals"] == int(B["valid"].sum()) * 400
    jobs = [(g, 128, 0) for g in (1, 640)]                                           # whole games, every search
    jobs += [(g, 128, max(int(B["game_turns"][g]) - 8, 0)) for g in range(77, 1024, 128)]   # the last 6+ searches of 8 games
    n_cmp, depth = _rebuild_in_workers(B, jobs, tmp_path)
    assert n_cmp >= 2 * 55 + 8 * 6
    assert len(depth) == 10 and 30 < max(depth.values()) < B["path_stride"], depth
    print("deepest oracle path per rebuilt game:", depth)
    used, walked = _audit_table(B, 2048)
    assert used > 200_000 and walked == 2048


def test_whole_games_at_n_thr_1_through_the_oracle(whole_nthr1, tmp_path):
    """VERDICT r05, hole (ii): the launch the bench's `mcts_nthr1` leg times (n_thr = 1: every leaf expands at its
    second visit, the policy net inside every playout; MCTS.py:80,109): records through the oracle, the first 4
    searches of 8 games rebuilt."""
    B = whole_nthr1
    n = _replay_records(B, whole=True)
    assert n > 1024 * 58 and B["leaf_evals"] == int(B["valid"].sum()) * 100
    assert B["totals"][1] > 500_000                         # the policy net at (nearly) every playout
    n_cmp, depth = _rebuild_in_workers(B, [(g, 4, 0) for g in range(9, 1024, 128)], tmp_path, n_workers=2)
    assert n_cmp == 8 * 4
