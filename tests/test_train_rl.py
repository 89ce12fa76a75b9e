"""REINFORCE update (src/train_rl.py:55-66): the loss formula against an
independent float64 numpy restatement, ChainerAdam against a numpy restatement
of Chainer's documented rule, npz optimizer-state layout, and one full
self-play set + update on the GPU."""
import math

import numpy as np
import pytest
import torch

from iago_amd import network
from oracle import nets_np
from oracle import oracle as orc


def test_chainer_adam_matches_numpy_rule():
    from iago_amd.train_rl import ChainerAdam
    torch.manual_seed(0)
    m = network.RolloutPolicy().double()
    opt = ChainerAdam(m, alpha=1e-3, weight_decay=5e-4)
    w = {n: p.detach().numpy().copy() for n, p in m.named_parameters()}
    mm = {n: np.zeros_like(v) for n, v in w.items()}
    vv = {n: np.zeros_like(v) for n, v in w.items()}
    rs = np.random.RandomState(0)
    for t in range(1, 6):
        grads = {n: rs.randn(*v.shape) for n, v in w.items()}
        for n, p in m.named_parameters():
            p.grad = torch.from_numpy(grads[n].copy())
        opt.update()
        a_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        for n in w:
            g = grads[n] + 5e-4 * w[n]
            mm[n] += (1 - 0.9) * (g - mm[n])
            vv[n] += (1 - 0.999) * (g * g - vv[n])
            w[n] -= a_t * mm[n] / (np.sqrt(vv[n]) + 1e-8)
        for n, p in m.named_parameters():
            assert np.allclose(p.detach().numpy(), w[n], rtol=1e-12, atol=1e-15)
    st = opt.state_dict_npz()
    assert set(st) == {"t", "epoch", "conv1/W/t", "conv1/W/m", "conv1/W/v", "bias2/b/t",
                       "bias2/b/m", "bias2/b/v"}  # layout of models/rollout_optimizer.npz
    assert int(st["t"]) == 5 and st["conv1/W/m"].shape == (1, 2, 3, 3)


def numpy_loss(params, x, y, r):
    """mean_i r_i * CE(softmax(logits)) with the reference's double softmax."""
    pred = nets_np.sl_policy(x, params)                      # (B,64) probabilities
    ls = pred - np.log(np.exp(pred).sum(axis=1, keepdims=True))  # log_softmax(pred)
    c = -ls[np.arange(len(y)), y]
    return float(np.mean(c * r))


def _planes_and_bits(n, seed):
    own, opp, xs = [], [], []
    rs = np.random.RandomState(seed)
    for g in range(n):
        z, final, tr = orc.random_playout(orc.initial_state(), 1, seed=seed, game_id=g)
        s, color = orc.initial_state(), 1
        for a in tr[: 2 * rs.randint(0, len(tr) // 2)]:
            orc.place_stone(s, a, color)
            color = 3 - color
        p1, p2 = orc.state_to_bits(s)
        own.append(p1)
        opp.append(p2)
        xs.append(orc.make_state_var(s, 1)[0])
    return np.array(own, np.uint64), np.array(opp, np.uint64), np.stack(xs)


@pytest.mark.gpu
def test_reinforce_loss_matches_numpy_gpu():
    from iago_amd import ops
    from iago_amd.train_rl import reinforce_loss
    params = nets_np.random_params("sl", 9)
    model = network.SLPolicy().load_npz(params).cuda()
    own, opp, x = _planes_and_bits(24, 3)
    rs = np.random.RandomState(1)
    y = rs.randint(0, 64, size=24)
    r = rs.choice([-1, 0, 1], size=24)
    loss = reinforce_loss(model, ops.bits_to_tensor(own), ops.bits_to_tensor(opp),
                          torch.from_numpy(y).cuda(), torch.from_numpy(r).cuda())
    assert abs(float(loss.item()) - numpy_loss(params, x, y, r)) < 1e-5
    # the same through iago_policy_reinforce_grad (the update's HIP kernels): the loss and the model's output
    probs = torch.empty(24, 64, device="cuda")
    native = model.reinforce_grads(ops.bits_to_tensor(own), ops.bits_to_tensor(opp), torch.from_numpy(y).cuda(),
                                   torch.from_numpy(r).cuda(), probs=probs)
    assert abs(float(native.item()) - numpy_loss(params, x, y, r)) < 1e-5
    assert float(np.abs(probs.cpu().numpy() - nets_np.sl_policy(x, params)).max()) < 1e-5


@pytest.mark.gpu
def test_one_training_set_on_gpu(tmp_path):
    from iago_amd.train_rl import ReinforceTrainer
    torch.manual_seed(0)
    model = network.SLPolicy()
    before = {k: v.copy() for k, v in model.npz_dict().items()}
    tr = ReinforceTrainer(model, pool_dir=str(tmp_path), N=8, seed=5)
    out = tr.step()
    assert 0.0 <= out["rate"] <= 1.0 and np.isfinite(out["loss"])
    assert 16 * 20 < out["n_tuples"] <= 16 * 32   # ~30 learner plies per game
    after = model.npz_dict()
    assert any(not np.array_equal(before[k], after[k]) for k in before)
    assert tr.opt.t == 1
    out2 = tr.step()
    assert np.isfinite(out2["loss"]) and tr.opt.t == 2


@pytest.mark.gpu
def test_step_from_pv_mcts_tuples_gpu():
    """BASELINE configs[4] as it is worded: PV-MCTS self-play tuples feed the REINFORCE update.
    32 games x 20 playouts per move (6 turns) -> SelfPlayResult.tuples() (own, opp, move, z from the mover's view)
    -> ReinforceTrainer.step_from_tuples: the loss equals the float64 numpy restatement of
    src/train_rl.py:55-66 on exactly those rows (x = the mover's planes, y = the move, r = z), one
    Adam step is taken, and colour = 1 restricts the batch to the learner's plies as the reference
    records them (src/rl_self_play.py:134-138)."""
    import json
    import os
    from iago_amd import engine, ops
    from iago_amd.train_rl import ReinforceTrainer
    from tests.conftest import GOLDEN
    params = nets_np.random_params("sl", 21)
    model = network.SLPolicy().load_npz(params)
    tr = ReinforceTrainer(model, pool_dir=None, N=32, seed=4)
    torch.manual_seed(3)
    value = network.Value().cuda().eval()
    with open(os.path.join(GOLDEN, "simulate.json")) as f:
        g = json.load(f)
    tr.model1.eval()
    m = engine.BatchedMCTS(32, tr.model1, value, ops.RolloutWeights(g["shipped_w"], g["shipped_b"]), n_thr=15,
                           capacity=4096, seed=2)      # (the default engine: the persistent search)
    res = engine.SelfPlayEngine(m, max_turns=6).play(20)
    tup = res.tuples()
    n = int(tup["z"].numel())
    assert n == 32 * 6 and set(tup) >= {"own", "opp", "move", "z", "colour", "game", "turn"}
    # float64 restatement on the same rows, in the canonical (turn, game) order the trainer sorts into
    order = np.lexsort((tup["game"].cpu().numpy(), tup["turn"].cpu().numpy()))
    own, opp = ops.tensor_to_bits(tup["own"])[order], ops.tensor_to_bits(tup["opp"])[order]
    x = np.stack([orc.make_state_var(orc.bits_to_state(int(o), int(p)), 1)[0] for o, p in zip(own, opp)])
    y = tup["move"].cpu().numpy().astype(np.int64)[order]
    r = tup["z"].cpu().numpy().astype(np.float64)[order]
    assert np.all(y >= 0) and set(np.unique(r)) <= {-1.0, 0.0, 1.0}
    want = numpy_loss(params, x, y, r)
    before = {k: v.copy() for k, v in tr.model1.npz_dict().items()}
    out = tr.step_from_tuples(tup)
    assert out["n_tuples"] == n and abs(out["loss"] - want) < 1e-5, (out["loss"], want)
    assert tr.opt.t == 1 and any(not np.array_equal(before[k], v) for k, v in tr.model1.npz_dict().items())
    # the learner's plies only
    one = tr.step_from_tuples(tup, colour=1)
    assert one["n_tuples"] == int((tup["colour"] == 1).sum().item()) == 32 * 3 and tr.opt.t == 2


@pytest.mark.gpu
def test_supervised_trainers_gpu():
    """train_policy / train_value loops: losses against numpy restatements, and a
    few epochs on a learnable synthetic set make the loss fall."""
    from iago_amd import ops
    from iago_amd.train_supervised import SupervisedTrainer, policy_loss, value_loss
    own, opp, x = _planes_and_bits(64, 7)
    rs = np.random.RandomState(2)
    y = rs.randint(0, 64, size=64)
    r = rs.choice([-1.0, 0.0, 1.0], size=64).astype(np.float32)
    params = nets_np.random_params("sl", 11)
    pm = network.SLPolicy().load_npz(params).cuda()
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    with torch.no_grad():
        lp, _ = policy_loss(pm, o, p, torch.from_numpy(y).cuda())
    assert abs(float(lp) - numpy_loss(params, x, y, np.ones(64))) < 1e-5
    vparams = nets_np.random_params("value", 12)
    vm = network.Value().load_npz(vparams).cuda().eval()
    with torch.no_grad():
        lv, _ = value_loss(vm, o, p, torch.from_numpy(r).cuda())
    want = float(np.mean((nets_np.value(x, vparams) - r) ** 2))
    assert abs(float(lv) - want) < 1e-5
    # learnable task: label = lowest legal move of the position
    n = 4096 + 512
    bo, bp, _ = _planes_and_bits(200, 9)
    bo, bp = np.tile(bo, 24)[:n], np.tile(bp, 24)[:n]
    O, P = ops.bits_to_tensor(bo), ops.bits_to_tensor(bp)
    legal = ops.tensor_to_bits(ops.legal_moves(O, P))
    lab = np.array([(int(m) & -int(m)).bit_length() - 1 if m else 0 for m in legal], np.int64)
    tr = SupervisedTrainer(network.RolloutPolicy(), "policy", seed=1)
    L = torch.from_numpy(lab).cuda()
    first = tr.evaluate(O, P, L)["loss"]
    for _ in range(6):
        tr.epoch(O, P, L)
    last = tr.evaluate(O, P, L)
    assert last["loss"] < first and 0.0 <= last["accuracy"] <= 1.0
    assert tr.opt.t == 12  # 2 minibatches per epoch
    # evaluate() walks a test set beyond one minibatch in pieces: the same means as the whole set at once
    with torch.no_grad():
        whole, pred = policy_loss(tr.model, O, P, L)
    assert abs(float(whole) - last["loss"]) < 1e-6
    assert abs(float((pred.argmax(dim=1) == L).float().mean()) - last["accuracy"]) < 1e-6
