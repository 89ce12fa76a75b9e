"""The split-f16 gradient kernels of the REINFORCE update (csrc/policy_grad_kernels.hip) against float64 autograd
of the same layers (src/train_rl.py:55-66 runs loss.backward() through Chainer's float32 convolutions)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("n,cin", [(1, 128), (37, 64), (300, 128), (2048, 128)])
def test_weight_gradient_of_a_block(n, cin):
    from iago_amd import ops
    g = torch.Generator(device="cuda").manual_seed(n + cin)
    x = torch.relu(torch.randn(n, cin, 8, 8, device="cuda", generator=g)) * 3.0
    dy = torch.randn(n, 128, 8, 8, device="cuda", generator=g) * (torch.rand(n, 1, 1, 1, device="cuda", generator=g) ** 8)
    dy = dy * (torch.rand(n, 128, 8, 8, device="cuda", generator=g) > 0.4)       # a ReLU's zeros
    e = 14 - int(np.ceil(np.log2(float(dy.abs().max()))))
    xs, dys = ops.split_nchw(x), ops.split_nchw(dy * 2.0 ** e)
    dw = ops.conv3x3_wgrad_split(dys, xs, scale_exp=torch.tensor([e], dtype=torch.int32, device="cuda"))
    # float64 reference: the weight gradient of conv2d
    w = torch.zeros(128, cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    F.conv2d(x.double(), w, padding=1).backward(dy.double())
    assert rel_err(dw, w.grad) < 2e-6, rel_err(dw, w.grad)
    # what float32 arithmetic itself gives on the same data
    w32 = torch.zeros(128, cin, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x, w32, padding=1).backward(dy)
    print("weight gradient, %d boards: split-f16 %.2e, float32 autograd %.2e of float64" % (n, rel_err(dw, w.grad), rel_err(w32.grad, w.grad)))


@pytest.mark.parametrize("n,cin", [(3, 128), (130, 64), (1024, 128)])
def test_input_gradient_of_a_block(n, cin):
    """Backward-data through the ReLU of the block below, and the scaled split of its float32 result."""
    from iago_amd import ops
    g = torch.Generator(device="cuda").manual_seed(7 * n + cin)
    w = torch.randn(128, cin, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)
    pre = torch.randn(n, cin, 8, 8, device="cuda", generator=g)            # the block below, before its ReLU
    dy = torch.randn(n, 128, 8, 8, device="cuda", generator=g) * 1e-6 * (torch.rand(n, 1, 1, 1, device="cuda", generator=g) ** 6)
    e = 13 - int(np.floor(np.log2(float(dy.abs().max()))))
    dys = ops.split_nchw(dy * 2.0 ** e)
    saved = ops.split_nchw(torch.relu(pre))
    wt_hi, wt_lo = ops.split_weights_transposed(w)
    dx, max_bits = ops.conv3x3_bwd_data_split(dys, torch.tensor([e], dtype=torch.int32, device="cuda"), wt_hi, wt_lo, saved)
    x64 = pre.double().requires_grad_(True)
    F.conv2d(torch.relu(x64), w.double(), padding=1).backward(dy.double())
    got = ops.blocks_to_nchw(dx)
    assert rel_err(got, x64.grad) < 2e-6, rel_err(got, x64.grad)
    assert bool((got[x64.grad == 0] == 0).all())      # the mask of the ReLU below (tiny rows may flush to zero besides)
    assert int(max_bits.item()) == int(dx.abs().max().view(torch.int32).item())
    s, e2, db = ops.split_scaled(dx, max_bits, bias_grad=True)
    ref_db = got.double().sum(dim=(0, 2, 3))
    assert float((db.double() - ref_db).abs().max()) <= 1e-5 * float(got.double().abs().sum(dim=(0, 2, 3)).max())
    top = float(dx.abs().max()) * 2.0 ** int(e2.item())
    assert 2 ** 13 <= top < 2 ** 14
    back = (s.hi.float() + s.lo.float() / 2048.0) * 2.0 ** -int(e2.item())
    assert float((back - dx).abs().max()) <= 2.0 ** -21 * float(dx.abs().max())


def _rows(n, seed):
    """n recorded learner positions from real policy-vs-policy games (own = the mover), their moves and results."""
    from iago_amd import network, rl_self_play
    torch.manual_seed(seed)
    m = network.SLPolicy().cuda().eval()
    r = rl_self_play.play_batch(m, m, 64, seed=seed)
    valid = r["action"] >= 0
    z = r["z"].reshape(1, -1).expand_as(r["action"])
    own, opp, act, zz = r["own"][valid], r["opp"][valid], r["action"][valid], z[valid]
    reps = (n + own.numel() - 1) // own.numel()
    return [t.repeat(reps)[:n].contiguous() for t in (own, opp, act, zz)]


def _relu_masks(model, own, opp):
    """[x_k > 0] of blocks 1..8 as the split-f16 forward computes them (the same kernels as the update's forward)."""
    from iago_amd import ops
    a = ops.value_stem_boards(own, opp, model.block1.conv.weight.detach(), model.block1.conv.bias.detach())
    masks = [ops.merge_nchw(a) > 0]
    for (hi, mid, lo, bias) in model._split3_layers():
        a = ops.conv3x3_split(a, hi, mid, bias)
        masks.append(ops.merge_nchw(a) > 0)
    return masks


def _autograd64(model, own, opp, act, z, masks=None):
    """The reference's loss and gradients in float64 autograd.  masks: take every ReLU's on / off decisions from
    there instead of from the float64 pre-activations (a pre-activation within rounding of zero flips a ReLU between
    two arithmetics, and ONE flipped cell moves a weight gradient -- a sum of cancelling terms -- by 1e-3 of its
    largest entry: float32 autograd against float64 shows exactly that)."""
    import copy
    from iago_amd import ops
    m64 = copy.deepcopy(model).double().train()
    for p in m64.parameters():
        p.grad = None
    h = ops.encode_planes(own, opp).double()
    for k in range(1, 9):
        pre = getattr(m64, "block%d" % k).conv(h)
        h = torch.relu(pre) if masks is None else pre * masks[k - 1]
    pred = torch.softmax(m64.bias10(m64.conv9(h).reshape(-1, 64)), dim=1)
    c = F.cross_entropy(pred, act.to(torch.int64), reduction="none")     # log-softmax AGAIN (src/train_rl.py:62)
    loss = torch.sum(c * z.double()) / own.numel()
    loss.backward()
    return loss.detach(), {k: p.grad for k, p in m64.named_parameters()}


@pytest.mark.parametrize("n,shipped", [(1, False), (5, False), (70, False), (1900, False), (1900, True)])
def test_reinforce_gradients_against_float64_autograd(n, shipped):
    """src/train_rl.py:61-65 through iago_policy_reinforce_grad against float64 autograd of the same loss, next to
    what float32 autograd (the tensor library's convolutions) gives on the same rows."""
    import os
    from iago_amd import network, train_rl
    own, opp, act, z = _rows(n, seed=n)
    torch.manual_seed(5)
    model = network.SLPolicy().cuda()
    if shipped:
        model.load_npz(os.path.join(os.path.dirname(__file__), "golden", "sl_model.npz"))
    loss64, ref = _autograd64(model, own, opp, act, z, masks=_relu_masks(model, own, opp))
    _, ref_plain = _autograd64(model, own, opp, act, z)
    loss = model.reinforce_grads(own, opp, act, z)
    model.check_saturation()
    got = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.train()
    for p in model.parameters():
        p.grad = None
    train_rl.reinforce_loss(model, own, opp, act, z).backward()
    got32 = {k: p.grad for k, p in model.named_parameters()}
    assert abs(float(loss) - float(loss64)) <= 1e-6 * max(1.0, abs(float(loss64)))

    def worst(a, b):
        return max(float((a[k].double() - b[k]).abs().max()) / float(b[k].abs().max()) for k in b)

    print("max relative error per tensor: split-f16 %.2e with the same ReLU decisions, %.2e against plain float64 "
          "(float32 autograd: %.2e)" % (worst(got, ref), worst(got, ref_plain), worst(got32, ref_plain)))
    if not shipped:
        assert worst(got, ref) < 1e-5
    # the shipped net's distributions are near one-hot: its gradients are differences of nearly equal numbers (block8's
    # bias gradient: 1e-6 at its largest, of terms of 1e-2), and float32 arithmetic itself is 1e-3 off there -- tensor
    # by tensor the split-f16 kernels must be as close to float64 as float32 autograd is
    top = max(float(ref[k].abs().max()) for k in ref)
    for k in ref:
        scale = float(ref[k].abs().max())
        if scale < 1e-4 * top:
            continue        # (a tensor that cancels to nothing: block8's bias on the shipped net, 1e-6 of terms of 1e-2)
        mine = float((got[k].double() - ref[k]).abs().max()) / scale
        theirs = float((got32[k].double() - ref[k]).abs().max()) / scale
        # (float32 autograd's own error depends on the convolution algorithms the tensor library picks on this box)
        assert mine < max(10 * theirs, 1e-5) and mine < 1e-3, (k, mine, theirs)
    assert worst(got, ref_plain) < 10 * worst(got32, ref_plain) + 1e-5


def test_reinforce_gradients_are_deterministic():
    """Fixed summation orders everywhere: the same rows give the same bits, run after run (the replicas of a
    multi-GPU job compute their updates from the same gathered batch)."""
    from iago_amd import network
    own, opp, act, z = _rows(1500, seed=3)
    torch.manual_seed(9)
    model = network.SLPolicy().cuda()
    runs = []
    for _ in range(3):
        loss = model.reinforce_grads(own, opp, act, z)
        runs.append([float(loss)] + [p.grad.clone() for p in model.parameters()])
        torch.empty(1 << 26, device="cuda").normal_()        # (other bytes where freed scratch may have been)
    for other in runs[1:]:
        assert other[0] == runs[0][0]
        for a, b in zip(runs[0][1:], other[1:]):
            assert torch.equal(a, b)


def test_fused_adam_equals_the_elementwise_rule_bit_for_bit():
    """iago_adam_chainer against ChainerAdam's multi-tensor restatement of Chainer's rule (tests/test_train_rl.py holds
    that one to the numpy rule): the same bits after five steps, moments included."""
    import copy
    from iago_amd import network, train_rl
    torch.manual_seed(2)
    a = network.SLPolicy().cuda()
    b = copy.deepcopy(a)
    oa, ob = train_rl.ChainerAdam(a), train_rl.ChainerAdam(b)
    g = torch.Generator(device="cuda").manual_seed(4)
    for t in range(5):
        for pa, pb in zip(a.parameters(), b.parameters()):
            pa.grad = torch.randn(pa.shape, device="cuda", generator=g) * 10.0 ** float(torch.randint(-8, 1, (1,)).item())
            pb.grad = pa.grad.clone()
        v0 = [p._version for p in a.parameters()]
        oa.update()
        assert all(p._version > v for p, v in zip(a.parameters(), v0))
        was, train_rl.NATIVE_GRAD = train_rl.NATIVE_GRAD, False
        try:
            ob.update()
        finally:
            train_rl.NATIVE_GRAD = was
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert torch.equal(pa, pb), (t, n)
            assert torch.equal(oa.state[n][0], ob.state[n][0]) and torch.equal(oa.state[n][1], ob.state[n][1]), (t, n)


def test_bad_arguments_are_refused():
    """The entry points of the update check what they are given (status codes, nothing launched)."""
    import ctypes as C
    from iago_amd import _lib
    L = _lib.lib()
    assert L.iago_policy_grad_workspace_bytes(-1) == -1
    assert L.iago_policy_grad_workspace_bytes(2048) > 2048 * 304 * 1024    # 308 KB per row + 247 MB
    a = _lib.PolicyGradArgs()
    assert L.iago_policy_reinforce_grad(C.byref(a), None) == -1          # n_mean 0, null pointers
    assert b"iago_policy_reinforce_grad" in L.iago_last_error()
    assert L.iago_conv3x3_wgrad_split(None, None, None, None, 4, 96, None, 32, None, None, None) == -1   # cin
    assert L.iago_conv3x3_wgrad_split(None, None, None, None, 4, 128, None, 12, None, None, None) == -1  # groups % 8
    assert L.iago_conv3x3_bwd_data_split(None, None, None, None, None, None, None, 32, None, None, 4, None) == -1
    assert L.iago_split_scaled(None, None, None, None, None, 4, 24, None, None, None) == -1
    ad = _lib.AdamArgs()
    assert L.iago_adam_chainer(C.byref(ad), None) == -1                  # no tensors


def test_rows_in_chunks_give_the_one_call_gradients(monkeypatch):
    """Batches beyond GRAD_CHUNK_ROWS rows run in chunks (the kernels' scratch is 308 KB per row), every chunk dividing
    by the whole batch's row count, the chunks' gradients added in chunk order: the one-call gradients within float32
    rounding of the sums, the same bits from run to run."""
    from iago_amd import network
    own, opp, act, z = _rows(700, seed=11)
    torch.manual_seed(3)
    model = network.SLPolicy().cuda()
    loss1 = model.reinforce_grads(own, opp, act, z)
    one = {k: p.grad.clone() for k, p in model.named_parameters()}
    monkeypatch.setattr(network.SLPolicy, "GRAD_CHUNK_ROWS", 256)
    probs = torch.empty(700, 64, device="cuda")
    loss3 = model.reinforce_grads(own, opp, act, z, probs=probs)
    three = {k: p.grad.clone() for k, p in model.named_parameters()}
    loss3b = model.reinforce_grads(own, opp, act, z)
    assert abs(float(loss1) - float(loss3)) <= 1e-6 * max(1.0, abs(float(loss1)))
    for k in one:
        assert rel_err(three[k], one[k]) < 2e-6, (k, rel_err(three[k], one[k]))
        assert torch.equal(three[k], dict(model.named_parameters())[k].grad), k      # deterministic
    assert float(loss3) == float(loss3b)
    assert float((probs.sum(dim=1) - 1).abs().max()) < 1e-5                           # every chunk wrote its rows


def test_an_action_outside_the_board_raises_the_flag_and_the_trainer_applies_nothing(tmp_path):
    """F.softmax_cross_entropy raises on a label outside 0 .. 63 (src/train_rl.py:62); the head kernel raises bit 1
    of the module's overflow word, and ReinforceTrainer reads the word BEFORE Adam: no update, no broadcast."""
    from iago_amd import network, _lib
    from iago_amd.train_rl import ReinforceTrainer
    own, opp, act, z = _rows(40, seed=2)
    torch.manual_seed(4)
    model = network.SLPolicy().cuda()
    model.reinforce_grads(own, opp, act, z)
    assert int(model._overflow_flag(own.device).item()) == 0
    bad = act.clone()
    bad[7] = -1                                   # a pass, as rl_self_play records it
    model.reinforce_grads(own, opp, bad, z)
    assert int(model._overflow_flag(own.device).item()) & 2
    model._overflow_flag(own.device).zero_()
    tr = ReinforceTrainer(model, pool_dir=str(tmp_path), N=2, seed=1)
    before = {k: v.copy() for k, v in model.npz_dict().items()}
    with pytest.raises(_lib.IagoError, match="outside 0 .. 63"):
        tr._update(own, opp, bad, z)
    after = model.npz_dict()
    assert all(np.array_equal(before[k], after[k]) for k in before) and tr.opt.t == 0
    assert int(model._overflow_flag(own.device).item()) == 0          # cleared with the error
    tr._update(own, opp, act, z)                                       # the next good batch goes through
    assert tr.opt.t == 1


def test_a_float32_model_takes_the_autograd_update(tmp_path, monkeypatch):
    """`split3 = False` is the documented remedy when activations leave the f16 range: the update then runs through
    float32 autograd as well, not through the split-f16 kernels."""
    from iago_amd import network
    from iago_amd.train_rl import ReinforceTrainer
    own, opp, act, z = _rows(40, seed=6)
    torch.manual_seed(8)
    model = network.SLPolicy().cuda()
    model.split3 = False
    called = []
    monkeypatch.setattr(network.SLPolicy, "reinforce_grads", lambda self, *a, **k: called.append(1))
    tr = ReinforceTrainer(model, pool_dir=str(tmp_path), N=2, seed=1)
    loss = tr._update(own, opp, act, z)
    assert not called and np.isfinite(float(loss.item())) and tr.opt.t == 1
