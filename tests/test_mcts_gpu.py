"""Bit-exact parity of the batched HIP PV-MCTS (select / expand / leaf_values /
backup / best_move / advance_root through the C ABI) with the oracle's
restatement of MCTS.py, game by game: same visit counts, same float32 Q and P,
same child order, same chosen move, also after subtree reuse.

Both sides see identical network outputs (deterministic stand-in nets computed
from the planes, exactly representable in float32) and identical rollout
results (the GPU rollout's z is recorded per simulation and replayed into the
oracle; the rollout kernel itself is checked in test_rollout_gpu.py)."""
import numpy as np
import pytest
import torch

from oracle import mcts_py
from oracle import oracle as orc
from tests.gpu_util import random_positions, state_of
from tests.test_oracle_golden import _cmp_tree, _hash_probs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from iago_amd import engine, ops
    assert torch.cuda.is_available()
    return engine, ops


def fake_nets(salt):
    def policy_np(x):
        return _hash_probs(x, salt)[0] / np.float32(65536.0)

    def value_np(x):
        h = _hash_probs(x, salt)[1]
        return np.float32((((h >> 20) & 0x7FF) - 1024) / 1024.0)

    def policy_t(planes):
        p = planes.cpu().numpy()
        return torch.from_numpy(np.stack([policy_np(p[i]) for i in range(len(p))])).cuda()

    def value_t(planes):
        p = planes.cpu().numpy()
        return torch.from_numpy(np.array([value_np(p[i]) for i in range(len(p))], np.float32)).cuda()

    return policy_np, value_np, policy_t, value_t


def positions(n, seed, golden_rules):
    own, opp = random_positions(n, seed=seed)
    own[0], opp[0] = 0x0000000810000000, 0x0000001008000000
    eb = golden_rules["edge_boards"]
    # 'pass1' (index 6): colour 1 must pass; 'dead' (5): nobody can move; 'full' (8)
    own[1], opp[1] = eb[6][0], eb[6][1]
    own[2], opp[2] = eb[5][0], eb[5][1]
    own[3], opp[3] = eb[8][0], eb[8][1]
    return own, opp


# (sync_free = False drives the per-phase forms of include/iago_hip_experimental.h -- select / pending / expand / mix_backup,
# one host synchronisation per playout: kept for two of the settings, VERDICT r05 task 7)
@pytest.mark.parametrize("lmbda,c_puct,n_thr,n_sims,sync_free", [(0.5, 1.0, 15, 100, False), (0.5, 1.0, 15, 100, True),
                                                                  (0.5, 1.0, 1, 40, True), (0.0, 2.5, 4, 60, False),
                                                                  (0.0, 2.5, 4, 60, True), (0.25, 1.0, 2, 50, True),
                                                                  (0.5, 1.0, 15, 400, True)])  # BASELINE configs[3]
def test_search_trees_bit_exact(eng, golden_rules, lmbda, c_puct, n_thr, n_sims, sync_free):
    """sync_free=False: the host counts the expanding leaves (one sync per playout);
    True: the count stays on the device and every launch of the playout is enqueued
    unconditionally (what the hipGraph replays).  Same trees."""
    engine, ops = eng
    G = 12
    own, opp = positions(G, 21, golden_rules)
    policy_np, value_np, policy_t, value_t = fake_nets(int(n_thr))
    m = engine.BatchedMCTS(G, policy_t, value_t, None, lmbda=lmbda, c_puct=c_puct, n_thr=n_thr,
                           capacity=2048, seed=5, sync_free=sync_free)
    assert m.sync_free == sync_free
    zs = []
    m.rollout_hook = lambda mm: zs.append(mm.z.cpu().numpy().copy())
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    active = torch.ones(G, dtype=torch.uint8, device="cuda")
    active[5] = 0  # an idle game must stay untouched
    m.search(o, p, active, n_sims)
    move = m.best_move(active)[0].cpu().numpy()
    visits = m.visits.cpu().numpy()
    assert m.n_leaf_evals == n_sims * (G - 1)
    assert int(m.tree.n_nodes[5].item()) == 1 and int(m.tree.n_visits[5 * 2048].item()) == 0
    # every expansion of the first search was counted (on the host or on the device)
    used = torch.arange(2048, device="cuda").reshape(1, 2048) < m.tree.n_nodes.reshape(G, 1)
    expanded = int(((m.tree.first_child.reshape(G, 2048) >= 0) & used).sum().item())
    assert m.n_policy_evals == expanded

    oracles = {}
    for g in range(G):
        if g == 5:
            continue
        it = iter([z[g] for z in zs])
        om = mcts_py.MCTS(policy_np, value_np, lambda s, c: int(next(it)), lmbda=lmbda,
                          c_puct=c_puct, n_thr=n_thr)
        want_move = om.get_move(state_of(own[g], opp[g]), 1, n_sims)
        _cmp_tree(m.tree.dump(g), mcts_py.dump_tree(om.root), "g%d" % g)
        if want_move is None:
            assert move[g] == -2
        else:
            assert move[g] == want_move, g
            for a, ch in om.root.children.items():
                if a >= 0:
                    assert visits[g, a] == ch.n_visits
        oracles[g] = om

    # subtree reuse: advance by the chosen move, search again from the new root
    mv = torch.from_numpy(np.where(move == -2, -1, move).astype(np.int8)).cuda()
    mask = active.clone()
    m.update_with_move(mv, mask)
    o2, p2 = o.clone(), p.clone()
    ops.apply_moves(o2, p2, mv)
    zs.clear()
    m.search(p2, o2, active, 30)  # the other side is to move now
    for g, om in oracles.items():
        a = int(mv[g].item())
        om.update_with_move(a)
        s = state_of(own[g], opp[g])
        orc.place_stone(s, a, 1)
        it = iter([z[g] for z in zs])
        om.rollout_fn = lambda st, c, it=it: int(next(it))
        om.get_move(s, 2, 30)
        _cmp_tree(m.tree.dump(g), mcts_py.dump_tree(om.root), "g%d'" % g)


def test_leaf_values_and_unknown_move_resets_root(eng):
    engine, ops = eng
    from iago_amd import _lib
    import ctypes as C
    v = torch.tensor([0.3, -0.7, 1.25, 0.0], device="cuda")
    z = torch.tensor([1, -1, 0, 1], dtype=torch.int8, device="cuda")
    out = torch.empty(4, device="cuda")
    for lm in (0.5, 0.0, 1.0, 0.25):
        _lib.check(_lib.lib().iago_leaf_values(C.c_void_p(v.data_ptr()), C.c_void_p(z.data_ptr()),
                                               lm, C.c_void_p(out.data_ptr()), 4, None))
        torch.cuda.synchronize()
        want = [orc.leaf_value(lm, float(v[i]), int(z[i])) for i in range(4)]
        if lm == 1.0:
            want = [np.float32(int(z[i])) for i in range(4)]
        assert out.cpu().numpy().tolist() == [float(w) for w in want], lm
    m = engine.BatchedMCTS(2, None, None, None, lmbda=1.0, n_thr=1, capacity=64)
    o = torch.full((2,), 0x0000000810000000, dtype=torch.int64, device="cuda")
    p = torch.full((2,), 0x0000001008000000, dtype=torch.int64, device="cuda")
    act = torch.ones(2, dtype=torch.uint8, device="cuda")
    m.policy_fn = lambda x: torch.full((x.shape[0], 64), 1.0 / 64, device="cuda")
    m.search(o, p, act, 5)
    assert int(m.tree.n_nodes[0].item()) > 1
    mv = torch.tensor([0, 19], dtype=torch.int8, device="cuda")  # 0 is not a child; 19 is
    m.update_with_move(mv)
    torch.cuda.synchronize()
    assert int(m.tree.n_nodes[0].item()) == 1 and int(m.tree.root[0].item()) == 0
    assert int(m.tree.root[1].item()) != 0


def test_pool_compaction_preserves_the_search(eng, golden_rules):
    """iago_mcts_compact (the nodes abandoned by subtree reuse are freed, the live subtree
    re-laid from index 0): an engine with small pools that compacts along the way must play
    the same moves with the same visit counts and end with the same trees as one whose pools
    never fill; the compacted pools hold exactly the reachable nodes."""
    engine, ops = eng
    G, n_sims, turns = 12, 60, 7
    own, opp = positions(G, 33, golden_rules)
    _, _, policy_t, value_t = fake_nets(2)

    def make(cap):
        return engine.BatchedMCTS(G, policy_t, value_t, None, lmbda=0.5, c_puct=1.0, n_thr=2,
                                  capacity=cap, seed=11)

    small, big = make(1024), make(16384)
    oa, pa = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    ob, pb = oa.clone(), pa.clone()
    for t in range(turns):
        legal = ops.legal_moves(oa, pa)
        active = (legal != 0).to(torch.uint8)
        small.search(oa, pa, active, n_sims)
        big.search(ob, pb, active, n_sims)
        ma, va = small.best_move(active)
        mb, vb = big.best_move(active)
        act = active.bool()
        assert torch.equal(ma[act], mb[act]) and torch.equal(va[act], vb[act]), t
        mv = torch.where(act, ma, torch.full_like(ma, -1))
        for m_, o_, p_ in ((small, oa, pa), (big, ob, pb)):
            ops.apply_moves(o_, p_, mv)
            m_.update_with_move(mv)
        oa, pa, ob, pb = pa, oa, pb, ob
    assert small.n_compactions > 0 and big.n_compactions == 0
    assert int(small.tree.overflow.sum().item()) == 0
    for g in range(G):
        assert small.tree.dump(g, max_depth=64) == big.tree.dump(g, max_depth=64), g
    # a compaction on its own: pools hold exactly the nodes reachable from the root, in
    # breadth-first order, links consistent
    small.tree.compact()
    T, cap = small.tree, small.tree.capacity
    nn = T.n_nodes.cpu().numpy()
    par = T.parent.cpu().numpy().reshape(G, cap)
    fc = T.first_child.cpu().numpy().reshape(G, cap)
    nc = T.n_children.cpu().numpy().reshape(G, cap)
    assert np.all(T.root.cpu().numpy() == 0)
    for g in range(G):
        nxt = 1
        assert par[g, 0] == -1
        for i in range(nn[g]):
            if fc[g, i] >= 0:
                assert fc[g, i] == nxt and np.all(par[g, nxt:nxt + nc[g, i]] == i)
                nxt += nc[g, i]
        assert nxt == nn[g]
        assert small.tree.dump(g, max_depth=64) == big.tree.dump(g, max_depth=64), g


def test_pool_overflow_is_reported(eng):
    engine, ops = eng
    from iago_amd import _lib
    m = engine.BatchedMCTS(1, lambda x: torch.full((x.shape[0], 64), 1.0 / 64, device="cuda"),
                           None, None, lmbda=1.0, n_thr=1, capacity=8)
    o = torch.full((1,), 0x0000000810000000, dtype=torch.int64, device="cuda")
    p = torch.full((1,), 0x0000001008000000, dtype=torch.int64, device="cuda")
    with pytest.raises(_lib.IagoError):
        m.search(o, p, torch.ones(1, dtype=torch.uint8, device="cuda"), 20)


def test_full_size_tree_invariants(eng):
    """BASELINE configs[2] size: 1024 games x 100 playouts from the start position
    with the real (random-init) nets.  Size-independent properties of the
    reference's algorithm on the device trees: visit conservation at the root
    and at every expanded node, parent/child links, priors, action order,
    determinism for a fixed seed, and agreement of best_move with the stored
    visit counts."""
    engine, ops = eng
    from iago_amd import network
    g = __import__("tests.conftest", fromlist=["load_json"]).load_json("simulate.json")
    torch.manual_seed(0)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    G, n_sims, n_thr, cap = 1024, 100, 15, 1024
    w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
    own = torch.full((G,), engine.START_OWN, dtype=torch.int64, device="cuda")
    opp = torch.full((G,), engine.START_OPP, dtype=torch.int64, device="cuda")
    act = torch.ones(G, dtype=torch.uint8, device="cuda")

    def run():
        m = engine.BatchedMCTS(G, policy, value, w, lmbda=0.5, c_puct=1.0, n_thr=n_thr,
                               capacity=cap, seed=3)
        m.search(own, opp, act, n_sims)
        return m

    m = run()
    T = m.tree
    nn = T.n_nodes.cpu().numpy()
    par = T.parent.cpu().numpy().reshape(G, cap)
    fc = T.first_child.cpu().numpy().reshape(G, cap)
    nc = T.n_children.cpu().numpy().reshape(G, cap)
    nv = T.n_visits.cpu().numpy().reshape(G, cap)
    q = T.q.cpu().numpy().reshape(G, cap)
    pr = T.p.cpu().numpy().reshape(G, cap)
    ac = T.action.cpu().numpy().reshape(G, cap)
    assert int(T.overflow.sum().item()) == 0
    assert np.all(T.root.cpu().numpy() == 0) and np.all(par[:, 0] == -1)
    assert np.all(nv[:, 0] == n_sims)                    # every playout passes through the root
    move, visits = m.best_move(act)
    move, visits = move.cpu().numpy(), visits.cpu().numpy()
    for gi in range(0, G, 7):
        n = nn[gi]
        assert 5 <= n <= cap
        for i in range(n):
            if fc[gi, i] >= 0:
                k, c0 = nc[gi, i], fc[gi, i]
                kids = slice(c0, c0 + k)
                assert np.all(par[gi, kids] == i)
                assert np.all(np.diff(ac[gi, kids].astype(int)) > 0)          # ascending actions
                # a node is evaluated itself exactly n_thr times before it expands
                assert nv[gi, i] == n_thr + nv[gi, kids].sum(), (gi, i)
                assert np.all(pr[gi, kids] > 0.1 - 1e-7) and np.all(pr[gi, kids] <= 1.1 + 1e-6)
            else:
                assert nv[gi, i] < n_thr or i == 0 or nv[gi, i] == n_thr      # unexpanded leaf
        assert np.all(np.abs(q[gi, :n]) <= 0.5 * np.abs(q[gi, :n]).max() + 1.0)
        kids = slice(fc[gi, 0], fc[gi, 0] + nc[gi, 0])
        assert nc[gi, 0] == 4 and list(ac[gi, kids]) == [19, 26, 37, 44]      # the 4 opening moves
        assert visits[gi, ac[gi, kids]].tolist() == nv[gi, kids].tolist()
        assert move[gi] == ac[gi, kids][np.argmax(nv[gi, kids])]
    # all 1024 games start identically; their trees differ only through the rollouts' RNG
    assert len({int(x) for x in nn}) > 1
    m2 = run()
    for f in ("n_visits", "q", "p", "action", "first_child"):
        a, b = getattr(m.tree, f), getattr(m2.tree, f)
        used = torch.arange(cap, device="cuda").reshape(1, cap) < m.tree.n_nodes.reshape(G, 1)
        assert torch.equal(a.reshape(G, cap)[used], b.reshape(G, cap)[used]), f


def test_counted_kernels_equal_host_counted(eng):
    """n_dev (device-side count) variants of the expansion path's kernels: the first
    min(n, *n_dev) rows equal the host-counted call, rows past the count stay untouched."""
    engine, ops = eng
    from iago_amd import network
    torch.manual_seed(3)
    m = network.SLPolicy().cuda().eval()
    own, opp = random_positions(200, seed=12)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    idx = torch.randperm(200, device="cuda")[:160].to(torch.int64)
    for cap, k in [(160, 37), (160, 0), (160, 160), (20, 7), (1, 1), (1, 0), (160, 500)]:
        cnt = torch.tensor([k], dtype=torch.int32, device="cuda")
        kk = min(k, cap)
        planes = torch.full((cap, 2, 8, 8), 7.0, device="cuda")
        ops.encode_planes_indexed(o, p, idx[:cap].contiguous(), planes, n_dev=cnt)
        want = torch.full((cap, 2, 8, 8), 7.0, device="cuda")
        if kk:
            ops.encode_planes_indexed(o, p, idx[:kk].contiguous(), want[:kk])
        assert torch.equal(planes, want)
        with torch.no_grad():
            got = m.forward_counted(planes, cnt)
            m.split3 = False   # per-layer float32 kernels, planes fused into the stem
            got_b = m.forward_counted_boards(o, p, idx[:cap].contiguous(), cap, cnt)
            m.split3 = True    # the whole net in one launch (three-piece split)
            got_s = m.forward_counted_boards(o, p, idx[:cap].contiguous(), cap, cnt)
            if kk:
                ref = m(want[:kk].contiguous())   # host-counted float32 kernels (<= 192 boards)
                assert torch.equal(got[:kk], ref), (cap, k)
                assert torch.equal(got_b[:kk], ref), (cap, k)
                ref_s = m.forward_boards_split3(o, p, idx[:kk].contiguous(), kk)   # host-counted
                assert torch.equal(got_s[:kk], ref_s), (cap, k)
                assert (got_s[:kk] - ref).abs().max().item() < 2e-6   # random-init net: flat distributions


def test_sync_free_equals_host_counted_with_real_nets(eng):
    """The same search with the real nets through both playout paths: identical visit
    counts, moves and tree shapes; Q / P within the nets' tolerance (above 192 expanding
    leaves the host-counted path evaluates the policy through MIOpen)."""
    engine, ops = eng
    from iago_amd import network
    g = __import__("tests.conftest", fromlist=["load_json"]).load_json("simulate.json")
    torch.manual_seed(2)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
    G = 256
    own, opp = random_positions(G, seed=5)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    act = torch.ones(G, dtype=torch.uint8, device="cuda")
    out = []
    for sync_free in (False, True):
        m = engine.BatchedMCTS(G, policy, value, w, n_thr=4, capacity=512, seed=9, sync_free=sync_free)
        m.search(o, p, act, 30)
        out.append((m.tree.n_visits.cpu(), m.tree.action.cpu(), m.tree.n_nodes.cpu(), m.tree.q.cpu(),
                    m.tree.p.cpu(), m.best_move(act)[0].cpu().clone(), m.n_policy_evals))
    a, b = out
    assert torch.equal(a[2], b[2]) and a[6] == b[6] and a[6] > G
    used = torch.arange(512).reshape(1, 512) < a[2].reshape(G, 1)
    for k in (0, 1):
        assert torch.equal(a[k].reshape(G, 512)[used], b[k].reshape(G, 512)[used]), k
    for k in (3, 4):
        assert torch.allclose(a[k].reshape(G, 512)[used], b[k].reshape(G, 512)[used], rtol=0, atol=1e-5)
    assert torch.equal(a[5], b[5])


@pytest.mark.parametrize("n_thr,K,use_graph,cap,overlap", [(15, 4, False, 128, 2), (15, 4, True, 128, 2),
                                                           (15, 4, True, 128, 1), (15, 4, True, 128, 0),
                                                           (6, 4, False, 512, 2), (9, 8, True, 256, 2),
                                                           (15, 1, False, 128, 0)])
def test_policy_lookahead_builds_the_same_trees(eng, n_thr, K, use_graph, cap, overlap):
    """Policy look-ahead (leaves queued K visits before they expand, the policy net on the queue
    every K playouts, priors from a cache) against the reference's order of evaluation (the net
    inside the playout that expands): bit-identical trees -- visit counts, Q, P, child order --
    and moves, over several moves with subtree reuse, small pools that compact on the way and
    playout counts that are not multiples of K; with the batch of a group at the group's end
    (overlap 0) or on a second stream beside the first playouts of the next group."""
    engine, ops = eng
    from iago_amd import network
    g = __import__("tests.conftest", fromlist=["load_json"]).load_json("simulate.json")
    torch.manual_seed(4)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
    G = 200
    own, opp = random_positions(G, seed=8)
    own[:40] = 0x0000000810000000
    opp[:40] = 0x0000001008000000

    def make(lookahead):
        return engine.BatchedMCTS(G, policy, value, w, n_thr=n_thr, capacity=cap, seed=13, sync_free=True,
                                  lookahead=lookahead, lookahead_overlap=overlap,
                                  use_graph=use_graph and lookahead > 0)

    ref, la = make(0), make(K)
    assert ref.lookahead == 0 and la.lookahead == K
    oa, pa = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    ob, pb = oa.clone(), pa.clone()
    for t, n_sims in enumerate([70, 50, 61, 44, 37]):
        active = (ops.legal_moves(oa, pa) != 0).to(torch.uint8)
        ref.search(oa, pa, active, n_sims)
        la.search(ob, pb, active, n_sims)
        ma, va = ref.best_move(active)
        mb, vb = la.best_move(active)
        act = active.bool()
        assert torch.equal(ma[act], mb[act]) and torch.equal(va[act], vb[act]), t
        mv = torch.where(act, ma, torch.full_like(ma, -1))
        for m_, o_, p_ in ((ref, oa, pa), (la, ob, pb)):
            ops.apply_moves(o_, p_, mv)
            m_.update_with_move(mv)
        oa, pa, ob, pb = pa, oa, pb, ob
    assert ref.n_compactions == la.n_compactions and (n_thr != 15 or la.n_compactions > 0)
    for gi in range(0, G, 3):
        assert ref.tree.dump(gi, max_depth=64) == la.tree.dump(gi, max_depth=64), gi
    # every expansion had its priors; the look-ahead evaluates a few leaves that never expand
    assert la.n_policy_evals >= ref.n_policy_evals > G
    # (leaves that reach n_thr - K visits and never n_thr are evaluated for nothing: 1.3 x the
    # reference's evaluations at n_thr = 15, K = 4; more when the trigger is a small count)
    print("policy evaluations: look-ahead %d, reference order %d" % (la.n_policy_evals, ref.n_policy_evals))
    assert n_thr != 15 or la.n_policy_evals < 2 * ref.n_policy_evals


def test_pass_chains_deeper_than_the_plies_left(eng):
    """At a finished position (neither side can move) every expansion adds one more pass child
    (MCTS.py:112-114), so a search path grows far beyond the plies left -- here past 200 nodes.
    The one-launch descent records the path and the backup walks the record: same trees as the
    reference's order with the climb through `parent`."""
    engine, ops = eng
    from iago_amd import network
    g = __import__("tests.conftest", fromlist=["load_json"]).load_json("simulate.json")
    torch.manual_seed(9)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
    G = 16
    own = np.full(G, 1, np.uint64)                      # one stone each, far apart: no move for either side
    opp = np.full(G, 1 << 63, np.uint64)
    own[8:], opp[8:] = 0x0000000810000000, 0x0000001008000000   # and ordinary games beside them

    def make(lookahead):
        return engine.BatchedMCTS(G, policy, value, w, n_thr=3, capacity=4096, seed=2, sync_free=True,
                                  lookahead=lookahead, lookahead_overlap=0, use_graph=False)

    ref, la = make(0), make(1)
    assert la._la_path is not None and la.PATH_STRIDE >= 512
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    act = torch.ones(G, dtype=torch.uint8, device="cuda")
    ref.search(o, p, act, 700)
    la.search(o.clone(), p.clone(), act, 700)
    depth = int(la._la_path_len[:8].max().item())
    assert depth > 200, depth
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 5000))
    for gi in range(G):
        assert ref.tree.dump(gi, max_depth=1000) == la.tree.dump(gi, max_depth=1000), gi


@pytest.mark.parametrize("use_graph", [False, True])
def test_value_cache_builds_the_same_trees(eng, use_graph):
    """Value cache (the value net only on leaves it has not evaluated yet, every other visit takes
    the value stored in the node) against the reference's evaluation at every visit
    (MCTS.py:123-124): bit-identical trees -- visit counts, float32 Q, P, child order -- and
    moves over several moves with subtree reuse and compaction; the cache saves most evaluations;
    after an update of the value weights the stored values are dropped."""
    engine, ops = eng
    from iago_amd import network
    g = __import__("tests.conftest", fromlist=["load_json"]).load_json("simulate.json")
    torch.manual_seed(6)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
    G, cap = 256, 96    # >= 192 games: the uncached engine runs the same split-f16 kernels on all leaves
    own, opp = random_positions(G, seed=18)
    own[:50] = 0x0000000810000000
    opp[:50] = 0x0000001008000000

    def make(cache):
        return engine.BatchedMCTS(G, policy, value, w, n_thr=15, capacity=cap, seed=21, sync_free=True,
                                  value_cache=cache, use_graph=use_graph)

    ref, vc = make(False), make(True)
    assert not ref.value_cache and vc.value_cache and ref.tree.v is None
    oa, pa = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    ob, pb = oa.clone(), pa.clone()
    for t, n_sims in enumerate([70, 50, 61, 44, 37]):
        active = (ops.legal_moves(oa, pa) != 0).to(torch.uint8)
        ref.search(oa, pa, active, n_sims)
        vc.search(ob, pb, active, n_sims)
        ma, va = ref.best_move(active)
        mb, vb = vc.best_move(active)
        act = active.bool()
        assert torch.equal(ma[act], mb[act]) and torch.equal(va[act], vb[act]), t
        mv = torch.where(act, ma, torch.full_like(ma, -1))
        for m_, o_, p_ in ((ref, oa, pa), (vc, ob, pb)):
            ops.apply_moves(o_, p_, mv)
            m_.update_with_move(mv)
        oa, pa, ob, pb = pa, oa, pb, ob
    assert ref.n_compactions == vc.n_compactions > 0
    for gi in range(0, G, 3):
        assert ref.tree.dump(gi, max_depth=64) == vc.tree.dump(gi, max_depth=64), gi
    assert ref.n_value_evals == ref.n_leaf_evals == vc.n_leaf_evals
    assert vc.n_value_evals < 0.4 * vc.n_leaf_evals, (vc.n_value_evals, vc.n_leaf_evals)
    # new weights: every stored value is recomputed (one fresh evaluation per visited leaf)
    with torch.no_grad():
        value.fc11.weight.mul_(1.5)
    before = vc.n_value_evals
    active = (ops.legal_moves(ob, pb) != 0).to(torch.uint8)
    ref.search(oa, pa, active, 20)
    vc.search(ob, pb, active, 20)
    assert vc.n_value_evals - before >= int(active.sum().item())   # at least the roots were re-evaluated
    for gi in range(0, G, 5):
        assert ref.tree.dump(gi, max_depth=64) == vc.tree.dump(gi, max_depth=64), gi
    with torch.no_grad():
        value.fc11.weight.div_(1.5)


def test_policy_lookahead_reports_a_recycled_cache_slot(eng):
    engine, ops = eng
    from iago_amd import _lib, network
    g = __import__("tests.conftest", fromlist=["load_json"]).load_json("simulate.json")
    torch.manual_seed(4)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
    m = engine.BatchedMCTS(16, policy, value, w, n_thr=6, capacity=4096, seed=1, lookahead=4, lookahead_slots=1)
    o = torch.full((16,), engine.START_OWN, dtype=torch.int64, device="cuda")
    p = torch.full((16,), engine.START_OPP, dtype=torch.int64, device="cuda")
    with pytest.raises(_lib.IagoError):
        m.search(o, p, torch.ones(16, dtype=torch.uint8, device="cuda"), 400)


def test_graph_is_recaptured_when_weights_change(eng):
    """The captured graph holds device pointers of weight layouts cached per weight version;
    after an in-place update of the Value weights (training between searches, load_npz) the
    next search must re-capture instead of replaying stale pointers."""
    engine, ops = eng
    from iago_amd import network
    torch.manual_seed(4)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    G = 256
    own, opp = random_positions(G, seed=6)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    act = torch.ones(G, dtype=torch.uint8, device="cuda")
    m = engine.BatchedMCTS(G, policy, value, None, n_thr=3, capacity=512, seed=2, use_graph=True)
    m.search(o, p, act, 10)
    first = m._graph
    with torch.no_grad():
        for q in value.parameters():
            q.mul_(1.5)               # new versions -> new cached split-f16 layouts
    m.tree.reset()
    m.sim_counter = 0
    m.search(o, p, act, 10)
    assert m._graph is not first
    fresh = engine.BatchedMCTS(G, policy, value, None, n_thr=3, capacity=512, seed=2, use_graph=True)
    fresh.search(o, p, act, 10)
    used = torch.arange(512, device="cuda").reshape(1, 512) < fresh.tree.n_nodes.reshape(G, 1)
    assert torch.equal(m.tree.n_nodes, fresh.tree.n_nodes)
    for f in ("n_visits", "q", "p"):
        assert torch.equal(getattr(m.tree, f).reshape(G, 512)[used],
                           getattr(fresh.tree, f).reshape(G, 512)[used]), f


def test_graph_mode_equals_eager(eng):
    """hipGraph replay of the whole sync-free playout (select, pending, planes, policy net,
    expand, continue-select, value net, rollout, leaf mix, backup: one launch per playout)
    builds the same trees as eager launches."""
    engine, ops = eng
    from iago_amd import network
    g = __import__("tests.conftest", fromlist=["load_json"]).load_json("simulate.json")
    torch.manual_seed(1)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
    G = 8
    own, opp = random_positions(G, seed=3)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    act = torch.ones(G, dtype=torch.uint8, device="cuda")
    act[2] = 0
    trees = []
    for use_graph in (False, True):
        m = engine.BatchedMCTS(G, policy, value, w, n_thr=3, capacity=1024, seed=9,
                               use_graph=use_graph)
        m.warmup()
        m.search(o, p, act, 25)
        m.search(o, p, act, 15)          # second search replays the captured graph
        mv = m.best_move(act)[0].cpu().numpy().copy()
        trees.append((m.tree.n_visits.cpu(), m.tree.q.cpu(), m.tree.p.cpu(), m.tree.action.cpu(),
                      m.tree.n_nodes.cpu(), mv, m.n_leaf_evals, m.sim_counter))
    a, b = trees
    assert torch.equal(a[4], b[4]) and a[6] == b[6] and a[7] == b[7]
    used = torch.arange(1024).reshape(1, 1024) < a[4].reshape(G, 1)
    for k in (0, 3):   # visit counts and actions: identical trees
        assert torch.equal(a[k].reshape(G, 1024)[used], b[k].reshape(G, 1024)[used]), k
    # Q and P come through MIOpen, which may pick another algorithm (no workspace) under
    # stream capture: equal within the nets' 1e-5 parity tolerance, not bit for bit
    for k in (1, 2):
        assert torch.allclose(a[k].reshape(G, 1024)[used], b[k].reshape(G, 1024)[used],
                              rtol=0, atol=1e-5), k
    assert np.array_equal(a[5], b[5])


def test_pending_compaction():
    """iago_mcts_pending: mask, ascending ids (int64 and int32) and count in one launch."""
    import ctypes as C
    import torch
    from iago_amd import _lib
    g = torch.Generator().manual_seed(4)
    for n in (1, 63, 1024, 2500):
        need = (torch.rand(n, generator=g) < 0.07).to(torch.uint8).cuda()
        act = (torch.rand(n, generator=g) < 0.8).to(torch.uint8).cuda()
        pend = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        idx = torch.full((n,), -1, dtype=torch.int64, device="cuda")
        games = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
        tot = torch.full((1,), 5, dtype=torch.int64, device="cuda")
        _lib.check(_lib.lib().iago_mcts_pending(p(need), p(act), n, p(pend), p(idx), p(games), p(cnt),
                                                p(tot), None))
        want = torch.nonzero(need & act).reshape(-1)
        k = int(cnt.item())
        assert k == want.numel() and int(tot.item()) == 5 + k
        assert torch.equal(idx[:k], want) and torch.equal(games[:k].long(), want)
        assert torch.equal(pend, need & act)
