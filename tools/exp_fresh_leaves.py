import sys, json, torch
sys.path.insert(0, '/root/repo')
from iago_amd import engine, network, ops
g = json.load(open('/root/repo/tests/golden/simulate.json'))
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
G = 1024
m = engine.BatchedMCTS(G, policy, value, w, n_thr=15, capacity=8192, seed=3, lookahead=4)
own = torch.full((G,), engine.START_OWN, dtype=torch.int64, device="cuda")
opp = torch.full((G,), engine.START_OPP, dtype=torch.int64, device="cuda")
act = torch.ones(G, dtype=torch.uint8, device="cuda")
ar = torch.arange(G, device="cuda") * 8192
hist = []
for move in range(6):
    fresh = []
    for i in range(100):
        m._select(own, opp, act, True)
        import ctypes as C
        from iago_amd import _lib
        from iago_amd.engine import _p, _stream, check
        check(_lib.lib().iago_mcts_expand_cached(m.tree.ref(), _p(act), _p(m.needs_expand), _p(m.cur_node), _p(m.legal), C.byref(m._la), _p(m._pending), _stream()), "x")
        m._select(own, opp, m._pending, False)
        nv = m.tree.n_visits[ar + m.cur_node.long()]
        fresh.append(int((nv == 0).sum().item()))
        m._evaluate_and_backup(act, stream_id=m.sim_counter)
        m.sim_counter += 1
        if (i + 1) % 4 == 0:
            m._flush_lookahead()
    mv, _ = m.best_move(act)
    ops.apply_moves(own, opp, mv)
    m.update_with_move(mv)
    own, opp = opp, own
    print("move", move, "fresh leaves per playout: mean %.1f" % (sum(fresh) / len(fresh)), "first 20:", fresh[:20], "last 5:", fresh[-5:])
