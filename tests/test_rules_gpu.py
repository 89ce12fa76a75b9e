"""Bit-exact parity of the HIP rule kernels (through the C ABI) with the golden
vectors recorded from the reference and with the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_util import positions_from_trace, random_positions, state_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from iago_amd import ops as o
    assert torch.cuda.is_available(), "the -m gpu tests need a HIP device"
    return o


def T(ops, a):
    return ops.bits_to_tensor(a)


def test_golden_trace_legal_and_flips(ops, golden_rules):
    own, opp, legal, action, own2, opp2 = positions_from_trace(golden_rules["trace"])
    o, p = T(ops, own), T(ops, opp)
    got = ops.tensor_to_bits(ops.legal_moves(o, p))
    assert np.array_equal(got, legal)
    a = torch.from_numpy(action.copy()).cuda()
    ops.apply_moves(o, p, a)
    assert np.array_equal(ops.tensor_to_bits(o), own2)
    assert np.array_equal(ops.tensor_to_bits(p), opp2)


def test_golden_edge_boards(ops, golden_rules):
    boards, legal, place = (golden_rules["edge_boards"], golden_rules["edge_legal"],
                            golden_rules["edge_place"])
    for color in (1, 2):
        own = boards[:, color - 1]
        opp = boards[:, 2 - color]
        got = ops.tensor_to_bits(ops.legal_moves(T(ops, own), T(ops, opp)))
        assert np.array_equal(got, legal[:, color - 1])
    bi, color = place[:, 0].astype(int), place[:, 1].astype(int)
    own = np.where(color == 1, boards[bi, 0], boards[bi, 1])
    opp = np.where(color == 1, boards[bi, 1], boards[bi, 0])
    want_own = np.where(color == 1, place[:, 3], place[:, 4])
    want_opp = np.where(color == 1, place[:, 4], place[:, 3])
    action = place[:, 2].astype(np.uint8).view(np.int8)
    o, p = T(ops, own), T(ops, opp)
    ops.apply_moves(o, p, torch.from_numpy(action.copy()).cuda())
    assert np.array_equal(ops.tensor_to_bits(o), want_own)
    assert np.array_equal(ops.tensor_to_bits(p), want_opp)


def test_golden_planes(ops, golden_rules):
    tr, idx, pl = golden_rules["trace"], golden_rules["planes_idx"], golden_rules["planes"]
    p1, p2 = tr[idx, 0], tr[idx, 1]
    # make_state_var(state, 1): mover = colour 1; (state, 2): mover = colour 2
    got1 = ops.encode_planes(T(ops, p1), T(ops, p2)).cpu().numpy()
    got2 = ops.encode_planes(T(ops, p2), T(ops, p1)).cpu().numpy()
    assert np.array_equal(got1, pl[:, 0])
    assert np.array_equal(got2, pl[:, 1])
    # GameEnv observation [state==1, state==2]: own = player 2, opp = player 1
    assert np.array_equal(got2, pl[:, 2])


def test_golden_judge(ops, golden_rules):
    tr, games = golden_rules["trace"], golden_rules["games"]
    last = tr[games[:, 2] + games[:, 1] - 1]
    z = ops.judge(T(ops, last[:, 5]), T(ops, last[:, 6])).cpu().numpy()
    assert np.array_equal(z, games[:, 0].astype(np.int8))
    z2 = ops.judge(T(ops, last[:, 6]), T(ops, last[:, 5])).cpu().numpy()
    assert np.array_equal(z2, -games[:, 0].astype(np.int8))


@pytest.mark.parametrize("n", [1, 7, 33, 1000])
def test_random_positions_vs_oracle(ops, n):
    own, opp = random_positions(n, seed=100 + n)
    legal = ops.tensor_to_bits(ops.legal_moves(T(ops, own), T(ops, opp)))
    rs = np.random.RandomState(n)
    acts = np.empty(n, np.int8)
    want_own, want_opp = np.empty(n, np.uint64), np.empty(n, np.uint64)
    for i in range(n):
        s = state_of(own[i], opp[i])
        la = orc.legal_actions(s, 1)
        assert orc.actions_to_mask(la) == int(legal[i])
        # mostly legal moves, sometimes arbitrary cells / passes (no legality check)
        r = rs.rand()
        a = la[rs.randint(len(la))] if (la and r < 0.7) else (-1 if r > 0.95 else rs.randint(64))
        acts[i] = a
        orc.place_stone(s, int(a), 1)
        want_own[i], want_opp[i] = orc.state_to_bits(s)
    o, p = T(ops, own), T(ops, opp)
    ops.apply_moves(o, p, torch.from_numpy(acts).cuda())
    assert np.array_equal(ops.tensor_to_bits(o), want_own)
    assert np.array_equal(ops.tensor_to_bits(p), want_opp)
    planes = ops.encode_planes(T(ops, own), T(ops, opp)).cpu().numpy()
    z = ops.judge(T(ops, own), T(ops, opp)).cpu().numpy()
    for i in range(0, n, max(1, n // 50)):
        s = state_of(own[i], opp[i])
        assert np.array_equal(planes[i], orc.make_state_var(s, 1)[0])
        assert z[i] == orc.judge(s, 1)


def test_empty_batch(ops):
    e = torch.empty(0, dtype=torch.int64, device="cuda")
    assert ops.legal_moves(e, e).numel() == 0
    assert ops.encode_planes(e, e).shape == (0, 2, 8, 8)
    assert ops.judge(e, e).numel() == 0


def _transform(bits, kind):
    """Board symmetry on a uint64 array: 'flipud', 'fliplr' or 'transpose'."""
    b = ((bits[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(-1, 8, 8)
    b = {"flipud": b[:, ::-1, :], "fliplr": b[:, :, ::-1], "transpose": b.transpose(0, 2, 1)}[kind]
    return (b.reshape(-1, 64).astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(
        axis=1, dtype=np.uint64)


def test_full_size_symmetry_property(ops):
    """1M boards (>> BASELINE's 4096): legal-move generation and flips commute
    with the board symmetries -- a size-independent check of all 8 directions."""
    base_own, base_opp = random_positions(4096, seed=77)
    reps = 256
    own = np.tile(base_own, reps)
    opp = np.tile(base_opp, reps)
    o, p = T(ops, own), T(ops, opp)
    legal = ops.legal_moves(o, p)
    lb = ops.tensor_to_bits(legal)
    assert np.array_equal(lb[:4096], lb[-4096:])
    for kind in ("flipud", "fliplr", "transpose"):
        lt = ops.tensor_to_bits(ops.legal_moves(T(ops, _transform(base_own, kind)),
                                                T(ops, _transform(base_opp, kind))))
        assert np.array_equal(lt, _transform(lb[:4096], kind)), kind
    # play the lowest legal move everywhere: stone counts grow by 1 + flips,
    # opp loses exactly the flips, no cell is owned twice
    low = lb & (~lb + np.uint64(1))
    act_np = np.where(lb != 0, np.log2(np.maximum(low, 1).astype(np.float64)).round(), -1)
    act = torch.from_numpy(act_np.astype(np.int8)).cuda()
    o2, p2 = o.clone(), p.clone()
    ops.apply_moves(o2, p2, act.to(torch.int8))
    a, b, a2, b2 = (ops.tensor_to_bits(x) for x in (o, p, o2, p2))
    assert not np.any(a2 & b2)
    assert np.array_equal(a2 | b2, a | b | np.where(lb != 0, lb & (~lb + np.uint64(1)), 0))
    moved = lb != 0
    assert np.all((a2[moved] & a[moved]) == a[moved])
    flips = b[moved] & ~b2[moved]
    assert np.all(flips != 0)


def test_augment8_golden_and_oracle(ops, golden_rules):
    """iago_augment8 against the vectors recorded from load.py's own functions
    and against the numpy restatement on a large random batch."""
    from oracle import augment_np
    bits, act = golden_rules["aug_bits"], golden_rules["aug_act"]
    o, p, a = ops.augment8(T(ops, bits[0, :, 0]), T(ops, bits[0, :, 1]),
                           torch.from_numpy(act[0].astype(np.int8)).cuda())
    assert np.array_equal(ops.tensor_to_bits(o).reshape(8, -1), bits[:, :, 0])
    assert np.array_equal(ops.tensor_to_bits(p).reshape(8, -1), bits[:, :, 1])
    assert np.array_equal(a.cpu().numpy(), act.astype(np.int8))
    n = 5000
    own, opp = random_positions(n, seed=41)
    rs = np.random.RandomState(2)
    acts = rs.randint(-1, 64, size=n).astype(np.int8)
    o, p, a = ops.augment8(T(ops, own), T(ops, opp), torch.from_numpy(acts).cuda())
    states = np.stack([state_of(own[i], opp[i]) for i in range(n)])
    S, A = augment_np.augment8(states, np.maximum(acts, 0))
    ob, pb, ab = ops.tensor_to_bits(o).reshape(8, n), ops.tensor_to_bits(p).reshape(8, n), a.cpu().numpy()
    for k in range(8):
        for i in range(0, n, 50):
            assert orc.state_to_bits(S[k, i]) == (int(ob[k, i]), int(pb[k, i]))
        ok = acts >= 0
        assert np.array_equal(ab[k][ok], A[k][ok].astype(np.int8))
        assert np.all(ab[k][~ok] == -1)
    # the 8 variants of a position have the same number of stones and legal moves
    lm = ops.tensor_to_bits(ops.legal_moves(o.reshape(-1), p.reshape(-1))).reshape(8, n)
    pop = np.vectorize(lambda v: bin(int(v)).count("1"))
    mob = pop(lm[:, :400])
    assert np.all(mob == mob[0:1])


@pytest.mark.gpu
def test_encode_planes_indexed_matches_gather():
    import torch
    from iago_amd import ops
    g = torch.Generator().manual_seed(3)
    own = torch.randint(-2**62, 2**62, (97,), generator=g, dtype=torch.int64)
    opp = torch.randint(-2**62, 2**62, (97,), generator=g, dtype=torch.int64) & ~own
    idx = torch.tensor([5, 0, 96, 42, 42, 7], dtype=torch.int64)
    out = torch.zeros(8, 2, 8, 8, device="cuda")
    ops.encode_planes_indexed(own.cuda(), opp.cuda(), idx.cuda(), out)
    want = ops.encode_planes(own[idx].cuda(), opp[idx].cuda())
    assert torch.equal(out[:6], want) and float(out[6:].abs().sum()) == 0.0


def test_play_turn_equals_the_separate_calls(ops):
    """iago_play_turn (move + stone_num / pass_flg / done + swap of sides + the next mover's legal
    moves in one launch; Game.turn inside `while stone_num < 64: turn(c); turn(3 - c)`,
    src/rl_self_play.py:27-31,130-145, game.py:117-142,253-255) against iago_legal_moves,
    iago_apply_moves and the bookkeeping written out with tensor operations, over whole games of
    random legal moves from random positions: every array identical after every turn."""
    n = 3000
    own_np, opp_np = random_positions(n, seed=77)
    own_np[:100], opp_np[:100] = 0x0000000810000000, 0x0000001008000000
    own_np[100:110], opp_np[100:110] = 1, 1 << 63            # nobody can move: double pass at once
    g = torch.Generator(device="cuda").manual_seed(5)
    # reference state (tensor operations) and fused state
    ro, rp = ops.bits_to_tensor(own_np), ops.bits_to_tensor(opp_np)
    fo, fp = ro.clone(), rp.clone()
    r_stones = torch.tensor([bin(int(a) | int(b)).count("1") for a, b in zip(own_np, opp_np)], dtype=torch.int32,
                            device="cuda")
    f_stones = r_stones.clone()
    r_pass = torch.zeros(n, dtype=torch.bool, device="cuda")
    r_done = torch.zeros(n, dtype=torch.bool, device="cuda")
    f_pass = torch.zeros(n, dtype=torch.uint8, device="cuda")
    f_done = torch.zeros(n, dtype=torch.uint8, device="cuda")
    f_legal = ops.legal_moves(fo, fp)
    f_active = (f_legal != 0).to(torch.uint8)
    f_legal2, f_active2 = torch.empty_like(f_legal), torch.empty_like(f_active)
    equal = torch.full((n, 64), 1.0 / 64, dtype=torch.float32, device="cuda")
    for t in range(130):
        legal = ops.legal_moves(ro, rp)
        active = (legal != 0) & ~r_done
        masked = torch.where(active, legal, torch.zeros_like(legal))
        assert torch.equal(masked, f_legal) and torch.equal(active.to(torch.uint8), f_active), t
        u = torch.rand(n, generator=g, device="cuda", dtype=torch.float64)
        a = ops.sample_moves(equal, masked, uniforms=u, seed=1, id_base=0, step=t)   # a random legal move, -1 = pass
        # the separate calls (what rl_self_play.play_batch did before the fused turn)
        ops.apply_moves(ro, rp, a)
        r_stones = r_stones + active.to(torch.int32)
        passing = ~active & ~r_done
        r_stones = torch.where(passing & r_pass, torch.full_like(r_stones, 64), r_stones)
        r_pass = torch.where(r_done, r_pass, passing)
        ro, rp = rp, ro
        if t % 2 == 1:
            r_done = r_done | (r_stones >= 64)
        # the fused turn
        ops.play_turn(fo, fp, a, f_active, f_stones, f_pass, f_done, t % 2 == 1, f_legal2, f_active2)
        f_legal, f_legal2 = f_legal2, f_legal
        f_active, f_active2 = f_active2, f_active
        assert torch.equal(fo, ro) and torch.equal(fp, rp), t
        assert torch.equal(f_stones, r_stones) and torch.equal(f_pass.bool(), r_pass), t
        assert torch.equal(f_done.bool(), r_done), t
        if t % 2 == 1 and bool(r_done.all().item()):
            break
    assert bool(r_done.all().item()) and t < 129
    with pytest.raises(ValueError):
        ops.play_turn(fo, fp, a, f_active, f_stones, f_pass, f_done, True, f_legal, f_active)   # aliased flags
