"""BASELINE.json configs[3] and configs[4] at one GPU's share, and the RCCL code path.

configs[3]: PV-MCTS 400 sims/move, batch = 8192 games sharded over 8 GPUs (1024 per GPU),
RCCL all-gather of the training tuples.  configs[4]: self-play feeding the REINFORCE update
(src/train_rl.py:28-81) on the gathered tuples.  One MI355X is what `-m gpu` runs on, so:
the 1024-game share at 400 playouts (tree invariants; trees bit-exact against the oracle at
400 playouts are in test_mcts_gpu.py), the `nccl` backend with one rank through exactly the
calls the multi-rank run makes, and the training loop driven until the snapshot gating of
src/train_rl.py:71-79 fires, then resumed from its checkpoint.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rollout_weights(ops):
    import json
    g = json.load(open(os.path.join(GOLDEN, "simulate.json")))
    return ops.RolloutWeights(np.asarray(g["shipped_w"], np.float32), np.asarray(g["shipped_b"], np.float32))


@pytest.mark.parametrize("persistent", [True, False])
def test_config3_share_1024_games_400_sims(persistent):
    """One GPU's share of configs[3]: 1024 games x 400 playouts from the start position, pool sized by
    suggest_capacity(400), on the persistent search (the engine bench.py's mcts400 leg times: ONE launch) and on
    the per-playout launches (hipGraph replay, policy look-ahead).  Size-independent properties of
    MCTS.py:105-133 on the device trees.  (The same share against the ORACLE on the persistent search:
    tests/test_bench_batch_gpu.py::test_config3_share_on_the_persistent_search.)"""
    from iago_amd import engine, network, ops
    torch.manual_seed(0)
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    G, n_sims, n_thr = 1024, 400, 15
    cap = engine.suggest_capacity(n_sims, n_thr)
    m = engine.BatchedMCTS(G, policy, value, _rollout_weights(ops), lmbda=0.5, c_puct=1.0, n_thr=n_thr,
                           capacity=cap, seed=3, use_graph=not persistent, persistent=persistent)
    assert m.persistent == persistent
    own = torch.full((G,), engine.START_OWN, dtype=torch.int64, device="cuda")
    opp = torch.full((G,), engine.START_OPP, dtype=torch.int64, device="cuda")
    act = torch.ones(G, dtype=torch.uint8, device="cuda")
    m.search(own, opp, act, n_sims)
    assert m.n_leaf_evals == G * n_sims
    T = m.tree
    nn = T.n_nodes.cpu().numpy()
    used = nn.max()
    view = lambda t: t.reshape(G, cap)[:, :used].cpu().numpy()
    par, fc, nc, nv, pr, ac = (view(getattr(T, f)) for f in ("parent", "first_child", "n_children",
                                                              "n_visits", "p", "action"))
    assert int(T.overflow.sum().item()) == 0 and np.all(nv[:, 0] == n_sims)
    live = np.arange(used)[None, :] < nn[:, None]
    expanded = live & (fc >= 0)
    # every expansion had the policy evaluated exactly once (single-move / pass expansions too:
    # the batch evaluates every queued leaf); with the policy look-ahead the leaves that are
    # queued but not yet expanded (first_child <= -2) have been evaluated as well
    if persistent:
        # the policy net exactly where the reference runs it: at the expansion of a node with a choice (MCTS.py:118-120)
        assert m.lookahead == 0
        assert m.n_policy_evals == int((expanded & (nc > 1)).sum())
    else:
        assert m.lookahead == 4
        assert m.n_policy_evals == int(expanded.sum()) + int((live & (fc <= -2)).sum())
    for g in range(0, G, 37):
        for i in np.nonzero(expanded[g])[0]:
            kids = slice(fc[g, i], fc[g, i] + nc[g, i])
            assert np.all(par[g, kids] == i)
            assert np.all(np.diff(ac[g, kids].astype(int)) > 0)
            assert nv[g, i] == n_thr + nv[g, kids].sum()      # visit conservation, MCTS.py:109
            assert np.all(pr[g, kids] > 0.1 - 1e-7) and np.all(pr[g, kids] <= 1.1 + 1e-6)
    mv, visits = m.best_move(act)
    mv, visits = mv.cpu().numpy(), visits.cpu().numpy()
    for g in range(0, G, 37):
        kids = slice(fc[g, 0], fc[g, 0] + nc[g, 0])
        assert list(ac[g, kids]) == [19, 26, 37, 44]
        assert visits[g, ac[g, kids]].tolist() == nv[g, kids].tolist()
        assert mv[g] == ac[g, kids][np.argmax(nv[g, kids])]
    assert len({int(x) for x in nn}) > 8   # the games differ through their rollouts


def test_nccl_code_path_with_one_rank(tmp_path):
    """The `nccl` (= RCCL) backend through the calls the N > 1 run makes, with world_size 1
    (all a one-GPU box offers): dist.gather_tuples on the tuples of a real self-play round,
    the bench's side-stream all_gather_into_tensor of a round buffer ordered behind its
    launches by an event, and the replica broadcast of train_rl."""
    import ctypes
    import torch.distributed as dist
    from iago_amd import engine, network, ops
    from iago_amd.dist import broadcast_tensors, gather_tuples, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 300)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert shard_range(8192, 0, 8) == (0, 1024) and shard_range(1024) == (0, 1024)
        torch.manual_seed(0)
        policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
        m = engine.BatchedMCTS(64, policy, value, _rollout_weights(ops), n_thr=2, capacity=2048, seed=5)
        res = engine.SelfPlayEngine(m, max_turns=6).play(8, record=True)
        tup = res.tuples()
        got = gather_tuples(tup)
        assert set(got) == set(tup) and got["z"].numel() > 64 * 4
        for k in tup:
            assert got[k].dtype == tup[k].dtype and torch.equal(got[k], tup[k].contiguous()), k
        # the bench's pattern: K launches into one round buffer, the collective on a side
        # stream behind an event, no host sync in between
        import bench
        rr = bench.RolloutRounds(4096, 4, 1, 0, _rollout_weights(ops), ops)
        main, comm = torch.cuda.current_stream(), torch.cuda.Stream()
        mptr = ctypes.c_void_p(main.cuda_stream)
        gathered = torch.zeros(rr.bufs[0].numel(), dtype=torch.uint8, device="cuda")
        done = torch.cuda.Event()
        for k in range(4):
            assert rr.launch(0, k, k, mptr) == 0
        done.record(main)
        with torch.cuda.stream(comm):
            comm.wait_event(done)
            dist.all_gather_into_tensor(gathered, rr.bufs[0])
        main.wait_stream(comm)
        torch.cuda.synchronize()
        assert torch.equal(gathered, rr.bufs[0]) and rr.board_steps(0) > 4 * 4096 * 55
        z = rr.views(0, 3)[2]
        assert set(torch.unique(z).tolist()) <= {-1, 0, 1}
        a = torch.arange(5, dtype=torch.float32, device="cuda")
        broadcast_tensors([a])
        assert torch.equal(a.cpu(), torch.arange(5, dtype=torch.float32))
    finally:
        dist.destroy_process_group()


def test_reinforce_loop_until_the_gating_fires_and_resume(tmp_path):
    """configs[4] in miniature: 30+ sets of (self-play -> tuples -> REINFORCE update) with an
    opponent pool.  The learner starts from the reference's shipped SL policy, the pool from
    a random-init model0 (the reference's pool starts from model0 = the SL policy,
    models/RL/model0.npz), so the learner wins most sets and the snapshot gating
    (src/train_rl.py:71-79: cnt > 4 sqrt(models) and rate > 0.6) fires; the log must replay
    the reference's counters exactly, the snapshot + optimizer npz must resume
    (src/train_rl.py:22-27) into the same next update as the uninterrupted run."""
    from iago_amd import network
    from iago_amd.train_rl import ReinforceTrainer
    pool = str(tmp_path)
    torch.manual_seed(1)
    network.SLPolicy().save_npz(os.path.join(pool, "model0.npz"))
    learner = network.SLPolicy().load_npz(os.path.join(GOLDEN, "sl_model.npz"))
    tr = ReinforceTrainer(learner, pool_dir=pool, N=16, seed=3, alpha=1e-5)
    saves, snap = [], None
    for s in range(36):
        out = tr.step()
        assert np.isfinite(out["loss"]) and not out["stop"]
        if out["saved"]:
            saves.append(s)
            if snap is None:  # state right after the first snapshot
                snap = dict(rs=tr.rs.get_state(), set_index=tr.set_index, t=tr.opt.t, models=tr.models)
                nxt = tr.step()
                after = {k: v.copy() for k, v in tr.model1.npz_dict().items()}
                saves_extra = nxt["saved"]
                break
    assert snap is not None, "the gating branch never fired: %s" % [o["rate"] for o in tr.log]
    # the reference's counters replayed from the logged win rates
    models, cnt, want = 1, 0, []
    for i, o in enumerate(tr.log[:saves[0] + 1]):
        if o["rate"] > 0.5:
            cnt += 1
        if cnt > 4 * math.sqrt(models) and o["rate"] > 0.6:
            want.append(i)
            models, cnt = models + 1, 0
    assert want == saves and saves[0] >= 4          # cnt must exceed 4 first
    assert os.path.exists(os.path.join(pool, "model1.npz"))
    assert os.path.exists(os.path.join(pool, "optimizers", "1.npz"))
    assert not [f for f in os.listdir(pool) if ".part" in f]
    st = np.load(os.path.join(pool, "optimizers", "1.npz"))
    assert int(st["t"]) == snap["t"] == saves[0] + 1 and "block1/conv/W/m" in st.files
    # resume: model1.npz + optimizers/1.npz, same draws -> the same next update
    tr2 = ReinforceTrainer.from_checkpoint(pool, 1, N=16, seed=3, alpha=1e-5)
    assert tr2.opt.t == snap["t"] and tr2.models == snap["models"] == 2 and tr2.cnt == 0
    tr2.rs.set_state(snap["rs"])
    tr2.set_index = snap["set_index"]
    out2 = tr2.step()
    assert out2["saved"] == saves_extra and tr2.opt.t == snap["t"] + 1
    assert out2["n_tuples"] == tr.log[-1]["n_tuples"] and out2["rate"] == tr.log[-1]["rate"]
    for k, v in tr2.model1.npz_dict().items():
        assert np.allclose(v, after[k], rtol=0, atol=1e-6), k
