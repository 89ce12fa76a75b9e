"""The host-side mirrors of the reference's Python interfaces (GameFunctions,
Simulate, MCTS, rl_self_play.Game, GameEnv) against the golden vectors recorded
from the reference and against the oracle.  These read like the calls the
reference itself makes (MCTS.py:94-137, src/train_rl.py:41-47, game.py:117-142).
"""
import numpy as np
import pytest
import torch

from oracle import mcts_py
from oracle import oracle as orc
from tests.conftest import load_json

pytestmark = pytest.mark.gpu


def st(p1, p2):
    return orc.bits_to_state(int(p1), int(p2))


def test_game_functions(golden_rules):
    from iago_amd.game import GameFunctions as gf
    tr = golden_rules["trace"]
    for rec in tr[:: 97]:
        p1, p2, color, legal, action, q1, q2 = (int(x) for x in rec)
        s = st(p1, p2)
        acts = gf.legal_actions(s, color)
        assert acts == orc.legal_actions(s, color) and orc.actions_to_mask(acts) == legal
        a = -1 if action == 0xFF else action
        s2 = gf.place_stone(s, a, color)
        assert s2 is s and orc.state_to_bits(s) == (q1, q2)
        x = gf.make_state_var(s, color)
        assert tuple(x.shape) == (1, 2, 8, 8)
        assert np.array_equal(x.cpu().numpy(), orc.make_state_var(s, color))
    assert gf.ac2pos([0, 19, 63]) == [[1, 1], [3, 4], [8, 8]]
    assert gf.is_outside([8, 0]) and not gf.is_outside([7, 7])


def test_simulate_mirror():
    """mcts_self_play.Simulate(state)(color) against the 40 recorded runs of the reference's own
    Simulate (tests/golden/simulate.json): driven by the uniforms numpy drew there, the mirror
    must end on the same board with the same result."""
    from iago_amd import mcts_self_play, ops
    g = load_json("simulate.json")
    ws = [ops.RolloutWeights(g["w"], g["b"]), ops.RolloutWeights(g["shipped_w"], g["shipped_b"])]
    assert len(g["cases"]) >= 40
    for c in g["cases"]:
        sim = mcts_self_play.Simulate(st(c["p1"], c["p2"]), weights=ws[c["weights"]], uniforms=c["uniforms"])
        z = sim(c["color"])
        assert z == c["z"]
        assert orc.state_to_bits(sim.state) == (c["q1"], c["q2"])
        assert sim.stone_num == 64
    # the Philox-driven form: a finished game, scored from the caller's side
    for c in g["cases"][:10]:
        sim = mcts_self_play.Simulate(st(c["p1"], c["p2"]), weights=ws[1], seed=3)
        z = sim(c["color"])
        assert z == orc.judge(sim.state, c["color"])
        assert not orc.legal_actions(sim.state, 1) and not orc.legal_actions(sim.state, 2)
    with pytest.raises(RuntimeError):
        mcts_self_play._DEFAULT_WEIGHTS = None
        mcts_self_play.Simulate(orc.initial_state())


class _Rollout(torch.nn.Module):
    """RolloutPolicy-shaped stand-in model on the GPU (float32 conv + softmax)."""

    def __init__(self, w, b):
        super().__init__()
        from iago_amd import network
        self.m = network.RolloutPolicy()
        with torch.no_grad():
            self.m.conv1.weight.copy_(torch.tensor(w, dtype=torch.float32).reshape(1, 2, 3, 3))
            self.m.bias2.b.copy_(torch.tensor(b, dtype=torch.float32))
        self.m.cuda().eval()

    def forward(self, x):
        return self.m(x)


def test_rl_game_mirror_replays_reference_games():
    """src/rl_self_play.Game with the uniforms numpy drew in the reference run:
    same recorded (swapped) states, same actions, same z."""
    from iago_amd import rl_self_play
    hcs = [None, (2, 4), (3, 5), (4, 2), (5, 3)]
    for c in load_json("rl_game.json"):
        game = rl_self_play.Game(_Rollout(c["w1"], c["b1"]), _Rollout(c["w2"], c["b2"]),
                                 uniforms=c["uniforms"])
        if c["handicap"]:
            pos = hcs[c["handicap"]]
            game.state[pos[0], pos[1]] = 2  # src/train_rl.py:43-46
        states, actions, z = game()
        assert actions == c["actions"]
        assert z == c["z"]
        assert [list(orc.state_to_bits(s)) for s in states] == c["states"]
        assert orc.state_to_bits(game.state) == (c["q1"], c["q2"])


def test_rl_play_batch_short_games_end_where_the_reference_loop_ends(monkeypatch):
    """play_batch logs the end-of-batch flag on the device for the first 56 turns instead of reading it back every
    second turn; a batch that is over earlier (here: boards nearly full of colour-2 stones from the start) must come
    back as if the loop had stopped there: same n_turns, records, results as with the read-back at every pair of turns
    (`while stone_num < 64`, src/rl_self_play.py:28-30)."""
    from iago_amd import ops, rl_self_play
    rs = np.random.RandomState(8)
    w1, b1 = rs.randn(18).astype(np.float32), (0.5 * rs.randn(64)).astype(np.float32)
    m1, m2 = _Rollout(w1, b1), _Rollout(w1[::-1].copy(), b1)
    B = 24
    start = 0x0000001818000000
    hc = np.zeros(B, np.uint64)
    for g in range(B):
        empty = rs.choice([c for c in range(64) if not (start >> c) & 1], size=2 + g % 7, replace=False)
        full = (~np.uint64(0)) & ~np.uint64(start)
        for c in empty:
            full &= ~(np.uint64(1) << np.uint64(c))
        hc[g] = full
    res = []
    for sync_from in (56, 0):
        monkeypatch.setattr(rl_self_play, "SYNC_FROM", sync_from)
        r = rl_self_play.play_batch(m1, m2, B, handicap=ops.bits_to_tensor(hc), seed=5, game_id_base=40)
        res.append(r)
    a, b = res
    assert a["n_turns"] == b["n_turns"] and 2 <= b["n_turns"] < 40
    for k in ("own", "opp", "action", "z", "final_p1", "final_p2"):
        assert torch.equal(a[k], b[k]), k
    assert (a["action"] >= 0).sum().item() > 0


@pytest.mark.parametrize("short", [False, True])
def test_rl_play_batch_one_launch_equals_the_turn_loop(short, monkeypatch):
    """iago_selfplay_policy (whole policy-vs-policy games in one launch, a workgroup per game) plays the games of the
    launch-per-turn loop (iago_policy_forward_split3, iago_sample_moves, iago_play_turn), record for record -- normal
    games with the handicap stones of src/train_rl.py:43-46, and boards nearly full from the start (games of a few
    turns, some over at once: the rows after a game's end, the batch's end)."""
    from iago_amd import network, ops, rl_self_play
    torch.manual_seed(5)
    m1, m2 = network.SLPolicy().cuda().eval(), network.SLPolicy().cuda().eval()
    rs = np.random.RandomState(3)
    B = 37
    if short:
        start = 0x0000001818000000
        hc = np.zeros(B, np.uint64)
        for g in range(B):
            empty = rs.choice([c for c in range(64) if not (start >> c) & 1], size=1 + g % 9, replace=False)
            full = (~np.uint64(0)) & ~np.uint64(start)
            for c in empty:
                full &= ~(np.uint64(1) << np.uint64(c))
            hc[g] = full
    else:
        cells = [None, (2, 4), (3, 5), (4, 2), (5, 3)]
        hc = np.array([0 if g % 5 == 0 else 1 << (cells[g % 5][0] * 8 + cells[g % 5][1]) for g in range(B)], np.uint64)
    res = []
    for one in (True, False):
        monkeypatch.setattr(rl_self_play, "ONE_LAUNCH", one)
        res.append(rl_self_play.play_batch(m1, m2, B, handicap=ops.bits_to_tensor(hc), seed=11, game_id_base=500))
    a, b = res
    assert a["n_turns"] == b["n_turns"] and (b["n_turns"] < 40 if short else b["n_turns"] >= 56)
    for k in ("own", "opp", "action", "z", "final_p1", "final_p2"):
        assert torch.equal(a[k], b[k]), k
    assert (a["action"] >= 0).sum().item() > (0 if short else B * 25)


def test_rl_play_batch_matches_oracle_games():
    """Batched lockstep games with Philox sampling == the oracle's rl_game driven
    by the same uniforms, game by game."""
    from iago_amd import ops, rl_self_play
    rs = np.random.RandomState(4)
    w1, b1 = rs.randn(18).astype(np.float32), (0.5 * rs.randn(64)).astype(np.float32)
    w2, b2 = rs.randn(18).astype(np.float32), (0.5 * rs.randn(64)).astype(np.float32)
    B, seed, base = 48, 77, 1000
    hc_cells = [None, (2, 4), (3, 5), (4, 2), (5, 3)]
    hc = np.array([0 if g % 5 == 0 else 1 << (hc_cells[g % 5][0] * 8 + hc_cells[g % 5][1])
                   for g in range(B)], np.uint64)
    r = rl_self_play.play_batch(_Rollout(w1, b1), _Rollout(w2, b2), B,
                                handicap=ops.bits_to_tensor(hc), seed=seed, game_id_base=base)
    act = r["action"].cpu().numpy()
    z = r["z"].cpu().numpy()
    f1, f2 = ops.tensor_to_bits(r["final_p1"]), ops.tensor_to_bits(r["final_p2"])
    p1 = lambda x: orc.rollout_policy(x, w1, b1)[0]
    p2 = lambda x: orc.rollout_policy(x, w2, b2)[0]
    mismatches = 0
    for g in range(B):
        # the oracle consumes one uniform per move; turn t of game g draws uniform(seed, base+g, t)
        class Us(object):
            def __init__(self):
                self.t = 0
        # replay turn by turn to know t at each draw
        state = orc.initial_state(hc_cells[g % 5])
        stone_num, pass_flg, t, acts_l = 4, False, 0, []
        ok = True
        while stone_num < 64 and ok:
            for color in (1, 2):
                la = orc.legal_actions(state, color)
                if la:
                    prob = (p1 if color == 1 else p2)(orc.make_state_var(state, color))
                    a = orc.choice_cdf(orc.masked_probs(prob, la), orc.uniform(seed, base + g, t))
                    if color == 1:
                        got = int(act[t // 2, g])
                        if got != a:  # float32 policy on the GPU vs the oracle's: rounding flip
                            ok = False
                            mismatches += 1
                            break
                        acts_l.append(a)
                    orc.place_stone(state, a, color)
                    pass_flg, stone_num = False, stone_num + 1
                else:
                    if pass_flg:
                        stone_num = 64
                    pass_flg = True
                t += 1
        if ok:
            assert orc.state_to_bits(state) == (int(f1[g]), int(f2[g])), g
            assert z[g] == orc.judge(state, 1)
    assert mismatches <= 1


def test_mcts_mirror_get_move_and_update():
    from iago_amd import MCTS as mcts_mod
    from tests.test_mcts_gpu import fake_nets
    policy_np, value_np, policy_t, value_t = fake_nets(7)
    m = mcts_mod.MCTS(lmbda=0.0, c_puct=1, n_thr=3, policy_net=policy_t, value_net=value_t,
                      n_sims=60, capacity=4096)
    om = mcts_py.MCTS(policy_np, value_np, None, lmbda=0.0, c_puct=1, n_thr=3)
    state, color = orc.initial_state(), 1
    for ply in range(6):
        a = m.get_move(state, color)
        assert a == om.get_move(state, color, 60)
        m.update_with_move(a)
        om.update_with_move(a)
        orc.place_stone(state, a, color)
        color = 3 - color
    with pytest.raises(ValueError):
        mcts_mod.MCTS(lmbda=0.5)
    m2 = mcts_mod.MCTS(lmbda=0.0, n_thr=15, policy_net=policy_t, value_net=value_t, n_sims=5)
    with pytest.raises(ValueError):
        m2.get_move(orc.initial_state(), 1)  # root never expanded: max() of empty dict


def test_game_env_mirror():
    from iago_amd import rl_env
    c = load_json("env.json")[0]
    base = _Rollout(c["w2"], c["b2"])

    class Opp(torch.nn.Module):
        def forward(self, x):
            out = base(x).reshape(64).clone()
            s = (x[0, 0] + 2 * x[0, 1]).cpu().numpy().astype(np.float32)
            for a in orc.legal_actions(s, 2):
                out[a] += 0.25
            return out.reshape(1, 64)

    np.random.seed(0)
    env = rl_env.GameEnv(None, Opp())
    obs = env.reset()
    assert np.array_equal(obs.cpu().numpy(), orc.env_obs(orc.initial_state()))
    assert env.valid_pos(1) == [[3, 4], [4, 3], [5, 6], [6, 5]]
    done, steps = False, 0
    while not done:
        pos = env.valid_pos(1)
        a = (pos[0][0] - 1) * 8 + pos[0][1] - 1 if pos else 0
        before = env.state.copy()
        obs, r, done, info = env.step(a)
        assert r == 0 and info is None
        assert np.array_equal(obs.cpu().numpy(), orc.env_obs(env.state))
        if pos:
            assert env.state[pos[0][0] - 1, pos[0][1] - 1] != 0 and before[pos[0][0] - 1, pos[0][1] - 1] == 0
        steps += 1
        assert steps < 70
    assert env() == orc.judge(env.state, 1)
    assert done and (env.stone_num >= 64 or env.pass_flg)


def test_game_env_says_when_the_retry_loop_cannot_end():
    """rl_env.py:152-172 samples from `out - min(out)` and draws again until the cell is legal: a net whose lowest
    outputs sit on every legal cell makes the reference draw for ever.  The mirror raises there instead."""
    from iago_amd import rl_env

    class Opp(torch.nn.Module):
        def forward(self, x):
            s = (x[0, 0] + 2 * x[0, 1]).cpu().numpy().astype(np.float32)
            out = torch.ones(64)
            for a in orc.legal_actions(s, 2):
                out[a] = 0.0
            return (out / out.sum()).reshape(1, 64)

    env = rl_env.GameEnv(None, Opp())
    env.reset()
    with pytest.raises(RuntimeError, match="never ends"):
        env.step(2 * 8 + 3)     # [3, 4]: legal for colour 1; colour 2's reply can not be drawn


def test_game_env_replays_golden_episodes():
    """a-14: the 6 recorded reference GameEnv episodes (rl_env.py:26-74,152-172; stand-in
    opponent, the uniforms numpy drew) replayed through the GPU mirror: board, done,
    stone_num, pass_flg and the observation after every step, and the final z.  The rules
    (valid_pos, place_stone, the obs planes) run through the HIP kernels; the opponent net is
    the fixture's stand-in, evaluated by the same float32 routine that produced the fixture,
    so that every recorded uniform lands on the recorded cell."""
    from iago_amd import rl_env
    for c in load_json("env.json"):
        w2, b2 = c["w2"], c["b2"]

        def opp(x, w2=w2, b2=b2):
            xs = x.cpu().numpy().astype(np.float32).reshape(2, 8, 8)
            state = (xs[0] + 2 * xs[1]).astype(np.float32)
            out = orc.rollout_policy(xs, w2, b2)[0].copy()
            for a in orc.legal_actions(state, 2):
                out[a] += np.float32(0.25)
            return torch.from_numpy(out)

        it = iter(c["uniforms"])

        def choice(n, p=None):
            return orc.choice_cdf(np.asarray(p, np.float64), next(it))

        def no_fallback(seq):
            raise AssertionError("the fixtures hold legal agent actions only")

        env = rl_env.GameEnv(None, opp, choice=choice, fallback_choice=no_fallback)
        obs = env.reset()
        assert np.array_equal(obs.cpu().numpy(), orc.env_obs(orc.initial_state()))
        for s in c["steps"]:
            obs, r, done, info = env.step(s["action"])
            assert boards_bits(env.state) == (s["p1"], s["p2"])
            assert (done, r, env.stone_num, env.pass_flg) == (s["done"], s["reward"], s["stone_num"],
                                                              s["pass_flg"])
            assert info is None and s["obs_ok"]
            assert np.array_equal(obs.cpu().numpy(), orc.env_obs(env.state))
        assert next(it, None) is None  # every recorded draw was consumed
        assert env() == c["z"]


def boards_bits(state):
    from iago_amd import boards
    return boards.state_to_bits(state)


class _Scaled(torch.nn.Module):
    def __init__(self, base, k):
        super().__init__()
        self.base, self.k = base, float(k)

    def forward(self, x):
        return self.base(x) * self.k


def test_value_self_play_replays_reference_runs():
    """f-4: the 14 recorded runs of the reference's value_self_play.SelfPlay (stand-in
    nets, the uniforms numpy / random drew; tests/golden/make_value_golden.py) through the
    GPU mirror: same recorded position, same result, same final board, every draw consumed.
    Covers stop_num = 4 (random move at ply 0), 64 (the SL policy plays the whole game, the
    recorded side is stuck: result -1), both stand-in output conventions."""
    from iago_amd import value_self_play
    for c in load_json("value_data.json"):
        m0 = _Scaled(_Rollout(c["w0"], c["b0"]), c["scale"])
        m1 = _Scaled(_Rollout(c["w1"], c["b1"]), c["scale"])
        sp = value_self_play.SelfPlay(c["stop_num"], m0, m1, draws=[u for _, u in c["draws"]])
        state, result = sp()
        assert boards_bits(state) == (c["opp"], c["own"]), c["stop_num"]   # the mover's stones are 2s
        assert result == c["result"]
        assert boards_bits(sp.state) == (c["final_p1"], c["final_p2"])


def test_value_self_play_batch_properties():
    """The lockstep batch (Philox draws): every recorded position has exactly stop_num
    stones unless phase one ended by a double pass, is reachable, results are in {-1,0,1},
    stuck games carry -1, the result equals the judge of the final board from the recorded
    side, and the batch is deterministic in (seed, game id)."""
    from iago_amd import network, ops, value_self_play
    torch.manual_seed(0)
    sl, rl = network.SLPolicy().cuda().eval(), network.RolloutPolicy().cuda().eval()
    r = value_self_play.generate(sl, rl, 512, seed=4)
    own, opp = ops.tensor_to_bits(r["own"]), ops.tensor_to_bits(r["opp"])
    stones = np.array([bin(int(a | b)).count("1") for a, b in zip(own, opp)])
    stop = r["stop_num"].cpu().numpy()
    assert np.all(own & opp == 0) and np.all(stones <= 64)
    assert np.mean(stones == stop) > 0.97 and np.all((stones == stop) | (stones < stop))
    z = r["z"].cpu().numpy()
    assert set(np.unique(z)) <= {-1, 0, 1} and len(set(np.unique(z))) == 3
    assert np.all(z[r["dropped"].cpu().numpy()] == -1)
    f1, f2 = ops.tensor_to_bits(r["final_p1"]), ops.tensor_to_bits(r["final_p2"])
    col = r["color"].cpu().numpy()
    for g in range(0, 512, 7):
        if r["dropped"][g]:
            continue
        n1, n2 = bin(int(f1[g])).count("1"), bin(int(f2[g])).count("1")
        d = (n1 - n2) if col[g] == 1 else (n2 - n1)
        assert z[g] == (d > 0) - (d < 0)
    r2 = value_self_play.generate(sl, rl, 512, seed=4)
    assert torch.equal(r["own"], r2["own"]) and torch.equal(r["z"], r2["z"])
    a, b, zz = value_self_play.generate_dataset(sl, rl, 96, batch=64, seed=1)
    assert a.shape == b.shape == zz.shape == (96,)
    # a net sees ONE batch size from the first turn to the last (a sub-batch gathered per phase changes size every
    # turn, and every new size sends MIOpen through its solver search: a 1024-game batch took 58 s that way)
    seen = set()
    hook = rl.register_forward_pre_hook(lambda mod, args: seen.add(tuple(args[0].shape)))
    value_self_play.generate(sl, rl, 96, seed=9)
    hook.remove()
    assert seen == {(96, 2, 8, 8)}
