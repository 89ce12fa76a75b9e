"""Split-f16 MFMA convolution (csrc/conv_kernels.hip) against float32 PyTorch:
the Block of network.py:5-13 and the Value forward of network.py:66-96."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _boards(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = torch.rand(n, 64, generator=g)
    own = (r < 0.3).float()
    opp = ((r >= 0.3) & (r < 0.6)).float()
    return torch.stack([opp, own], 1).reshape(n, 2, 8, 8)


def test_split_merge_round_trip():
    from iago_amd import ops
    torch.manual_seed(1)
    x = (torch.randn(7, 64, 8, 8) * torch.logspace(-3, 2, 64).view(1, 64, 1, 1)).cuda()
    y = ops.merge_nchw(ops.split_nchw(x))
    # 22 significant bits; values below the f16 normal range (6e-5) keep an absolute
    # error of 2^-24 * 2^-11
    excess = ((y - x).abs() - 2.0 ** -21 * x.abs()).max().item()
    assert excess < 1e-10, excess


@pytest.mark.parametrize("cin", [64, 128])
@pytest.mark.parametrize("n", [1, 4, 5, 37])
def test_integer_data_is_exact(cin, n):
    """Small-integer operands are exact in f16 and their sums in f32: any slip in the
    MFMA operand / accumulator lane maps, the padding or the tap order shows up as an
    exact mismatch.  Asymmetric weights and inputs."""
    from iago_amd import ops
    g = torch.Generator().manual_seed(cin * 100 + n)
    x = torch.randint(-3, 4, (n, cin, 8, 8), generator=g).float()
    w = torch.randint(-2, 3, (128, cin, 3, 3), generator=g).float()
    b = torch.randint(-5, 6, (128,), generator=g).float()
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    w_hi, w_lo = ops.split_weights(w.cuda())
    assert float(w_lo.abs().max()) == 0.0
    y = ops.merge_nchw(ops.conv3x3_split(ops.split_nchw(x.cuda()), w_hi, w_lo, b.cuda()))
    assert torch.equal(y.cpu(), ref)


@pytest.mark.parametrize("cin", [64, 128])
def test_random_data_matches_float32(cin):
    from iago_amd import ops
    torch.manual_seed(cin)
    n = 33
    x = torch.rand(n, cin, 8, 8) * 2.0
    w = torch.randn(128, cin, 3, 3) / np.sqrt(9 * cin)
    b = torch.randn(128) * 0.1
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    w_hi, w_lo = ops.split_weights(w.cuda())
    y = ops.merge_nchw(ops.conv3x3_split(ops.split_nchw(x.cuda()), w_hi, w_lo, b.cuda()))
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 2e-6 * float(ref.abs().max()), err


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("n", [192, 333, 1024])
def test_value_forward_split_vs_float32(n, fused):
    """network.py:66-96 (train=False): the split-f16 stack (one fused launch, and the stem /
    trunk / head launches) against MIOpen float32 on the GPU and against float64 on the CPU;
    tolerance = the 1e-5 parity bar."""
    from iago_amd import network
    torch.manual_seed(5)
    m = network.Value().eval()
    m.fused = fused
    # random init gives outputs ~1e-2: scale the weights up to O(1) activations
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.6)
    x = _boards(n, seed=n)
    with torch.no_grad():
        ref64 = m.double()(x.double()).float()
        m.float()
        mg = m.cuda()
        mg.split_f16 = False
        y32 = mg(x.cuda()).cpu()
        mg.split_f16 = True
        assert mg._use_split(x.cuda())
        ys = mg(x.cuda()).cpu()
    scale = float(ref64.abs().max())
    assert (ys - ref64).abs().max().item() < 1e-5 * max(1.0, scale)
    assert (ys - y32).abs().max().item() < 1e-5 * max(1.0, scale)
    # the split path is as close to float64 as MIOpen's float32 within a small factor
    assert (ys - ref64).abs().max().item() < 8 * max((y32 - ref64).abs().max().item(), 1e-7 * scale)


def test_value_split_cache_follows_weight_updates():
    from iago_amd import network
    m = network.Value().eval().cuda()
    x = _boards(256).cuda()
    with torch.no_grad():
        a = m(x).clone()
        m.block5.conv.weight.mul_(0.5)
        b = m(x)
        m.split_f16 = False
        c = m(x)
    assert not torch.allclose(a, b)
    assert torch.allclose(b, c, atol=1e-6)


def test_value_stem_and_head_match_float32():
    from iago_amd import network, ops
    torch.manual_seed(11)
    m = network.Value().eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.6)
        m.block1.conv.bias.normal_(0, 0.1)
        m.block9.conv.bias.fill_(0.05)
    x = _boards(37, seed=3)
    with torch.no_grad():
        ref1 = F.relu(m.block1.conv(x))
        mg = m.cuda()
        a = ops.value_stem(x.cuda(), mg.block1.conv.weight, mg.block1.conv.bias)
        got1 = ops.merge_nchw(a).cpu()
        assert (got1 - ref1).abs().max().item() < 2e-6 * float(ref1.abs().max())
        h = torch.rand(37, 128, 8, 8) * 1.5
        m.cpu()
        r9 = F.relu(m.block9.conv(h)).reshape(-1, 64)
        ref = m.fc11(m.fc10(r9)).reshape(-1)
        mg = m.cuda()
        got = ops.value_head(ops.split_nchw(h.cuda()), mg.block9.conv.weight, mg.block9.conv.bias,
                             mg.fc10.weight, mg.fc11.weight).cpu()
        assert (got - ref).abs().max().item() < 2e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cin", [64, 128])
@pytest.mark.parametrize("n", [1, 3, 64])
def test_f32_conv_integer_data_is_exact(cin, n):
    from iago_amd import ops
    g = torch.Generator().manual_seed(cin + n)
    x = torch.randint(-3, 4, (n, cin, 8, 8), generator=g).float()
    w = torch.randint(-2, 3, (128, cin, 3, 3), generator=g).float()
    b = torch.randint(-5, 6, (128,), generator=g).float()
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    y = ops.conv3x3_f32(x.cuda(), ops.f32_weights(w.cuda()), b.cuda())
    assert torch.equal(y.cpu(), ref)


def test_f32_conv_random_data():
    from iago_amd import ops
    torch.manual_seed(2)
    x = torch.rand(9, 128, 8, 8) * 2.0
    w = torch.randn(128, 128, 3, 3) / np.sqrt(9 * 128)
    b = torch.randn(128) * 0.1
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    y = ops.conv3x3_f32(x.cuda(), ops.f32_weights(w.cuda()), b.cuda())
    assert (y.cpu().double() - ref).abs().max().item() < 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("n", [1, 5, 64, 128])
def test_policy_small_batch_kernels_vs_miopen_and_float64(n):
    """SLPolicy.__call__ (network.py:15-47) through stem_f32 / conv3x3_f32 / policy_head."""
    from iago_amd import network
    torch.manual_seed(9)
    m = network.SLPolicy().eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.5)
        m.bias10.b.normal_(0, 0.3)
        m.block3.conv.bias.normal_(0, 0.1)
    x = _boards(n, seed=40 + n)
    with torch.no_grad():
        ref64 = m.double()(x.double()).float()
        m.float()
        mg = m.cuda()
        assert mg._use_f32_kernels(x.cuda())
        got = mg(x.cuda()).cpu()
        mg.F32_MAX_BATCH = 0
        miopen = mg(x.cuda()).cpu()
    assert (got - ref64).abs().max().item() < 1e-5
    assert (got - miopen).abs().max().item() < 1e-5
    assert torch.allclose(got.sum(1), torch.ones(n), atol=1e-5)


def test_value_small_batch_uses_f32_kernels():
    from iago_amd import network
    torch.manual_seed(10)
    m = network.Value().eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.6)
    x = _boards(3, seed=77)
    with torch.no_grad():
        ref64 = m.double()(x.double()).float()
        m.float()
        got = m.cuda()(x.cuda()).cpu()
    assert (got - ref64).abs().max().item() < 1e-5 * max(1.0, float(ref64.abs().max()))


def test_value_forward_boards_equals_forward_on_planes():
    from iago_amd import network, ops
    from tests.gpu_util import random_positions
    torch.manual_seed(12)
    m = network.Value().eval().cuda()
    own, opp = random_positions(300, seed=5)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    with torch.no_grad():
        a = m.forward_boards(o, p)
        b = m(ops.encode_planes(o, p))
        assert m.forward_boards(o[:5], p[:5]) is None      # small batches: the caller encodes planes
    assert a is not None and torch.equal(a, b)


@pytest.mark.parametrize("n", [192, 193, 255, 1024])
def test_fused_value_forward_equals_three_launches(n):
    """iago_value_forward_split (block1, blocks 2..8 and the head in ONE launch, activations
    resident in LDS) against iago_value_stem(_boards) + iago_conv3x3_split_trunk +
    iago_value_head: block1 and the trunk are the same arithmetic; the head sums block9's
    products in another order (9-row MFMA product + 9 shifted adds instead of FMA chains), so
    the outputs agree to float32 rounding, for ragged batches, from planes and from boards."""
    from iago_amd import network, ops
    from tests.gpu_util import random_positions
    torch.manual_seed(31)
    m = network.Value().eval().cuda()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.6)
    own, opp = random_positions(n, seed=n)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    x = ops.encode_planes(o, p)
    with torch.no_grad():
        m.fused = False
        ref = m(x)
        ref_b = m.forward_boards(o, p)
        m.fused = True
        got = m(x)
        got_b = m.forward_boards(o, p)
    assert torch.equal(ref, ref_b) and torch.equal(got, got_b)
    scale = max(1.0, float(ref.abs().max()))
    assert (got - ref).abs().max().item() < 2e-6 * scale, (got - ref).abs().max().item()
    m.check_saturation()


@pytest.mark.parametrize("count", [0, 1, 7, 200, 512, 513, 1000])
def test_counted_value_forward_is_bit_identical(count):
    """iago_value_forward_split with a device-side gather list and count (what the value cache
    runs: two boards per workgroup up to 512 rows, four above, chosen on the device) against the
    plain forward of all boards: the listed boards get bit-identical values, the others are not
    written."""
    from iago_amd import network, ops
    from tests.gpu_util import random_positions
    torch.manual_seed(32)
    m = network.Value().eval().cuda()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.6)
    n = 1024
    own, opp = random_positions(n, seed=77)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    gen = torch.Generator().manual_seed(count)
    pick = torch.randperm(n, generator=gen)[:count].sort().values.cuda()
    index = torch.zeros(n, dtype=torch.int64, device="cuda")
    index[:count] = pick
    n_dev = torch.tensor([count], dtype=torch.int32, device="cuda")
    with torch.no_grad():
        full = m.forward_boards(o, p)                      # 1024 rows: four boards per workgroup
        out = torch.full((n,), -77.0, dtype=torch.float32, device="cuda")
        m.forward_boards_counted(o, p, index, n_dev, out)
    assert torch.equal(out[pick], full[pick])
    rest = torch.ones(n, dtype=torch.bool, device="cuda")
    rest[pick] = False
    assert bool((out[rest] == -77.0).all())
    m.check_saturation()


@pytest.mark.parametrize("count", [0, 5, 180, 256, 257, 511, 700, 1024])
def test_value_and_rollout_in_one_launch(count):
    """iago_value_rollout (the value net on a device-side list of boards and the rollout of ALL
    boards as two kinds of workgroups of one launch) against the two separate calls: identical
    values for the listed boards, identical games (z, final boards, turns) for all."""
    from iago_amd import network, ops
    from tests.conftest import load_json
    from tests.gpu_util import random_positions
    torch.manual_seed(33)
    m = network.Value().eval().cuda()
    g = load_json("simulate.json")
    w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
    n = 1024
    own, opp = random_positions(n, seed=91)
    o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
    gen = torch.Generator().manual_seed(count + 1)
    pick = torch.randperm(n, generator=gen)[:count].sort().values.cuda()
    index = torch.zeros(n, dtype=torch.int64, device="cuda")
    index[:count] = pick
    n_dev = torch.tensor([count], dtype=torch.int32, device="cuda")
    sid = torch.tensor([5], dtype=torch.int32, device="cuda")
    with torch.no_grad():
        want_v = torch.full((n,), -77.0, dtype=torch.float32, device="cuda")
        m.forward_boards_counted(o, p, index, n_dev, want_v)
        want = ops.rollout(o, p, w, seed=11, id_base=3000, stream_id=2, stream_id_dev=sid, want_final=True,
                           want_turns=True)
        got_v = torch.full((n,), -77.0, dtype=torch.float32, device="cuda")
        res = ops.RolloutResult()
        prep = ops.rollout_prepare(o, p, w, seed=11, id_base=3000, stream_id=2, stream_id_dev=sid, want_final=True,
                                   want_turns=True, out=res)
        m.forward_boards_counted(o, p, index, n_dev, got_v, rollout=prep)
    torch.cuda.synchronize()
    assert torch.equal(got_v, want_v)
    assert torch.equal(res.z, want.z) and torch.equal(res.n_turns, want.n_turns)
    assert torch.equal(res.final_own, want.final_own) and torch.equal(res.final_opp, want.final_opp)


def test_trunk_kernel_equals_layer_by_layer():
    """iago_conv3x3_split_trunk (several layers, one launch) against the per-layer launches, for a ragged batch and a
    64-channel first layer.  Both are the same split-f16 convolution with float32 accumulation; since round 5 the trunk
    sums a product's 32 input channels per v_mfma_f32_16x16x32_f16 where the per-layer kernel sums 16 per
    v_mfma_f32_32x32x16_f16, so the two differ by float32 rounding of the accumulation order (they were bit-equal while
    both used the long shape): equal to a few float32 ulps of the layer's largest activation after four layers, and
    both within the same distance of a float64 convolution of the same split operands."""
    from iago_amd import ops
    torch.manual_seed(21)
    n = 37
    x0 = (torch.rand(n, 64, 8, 8) * 2).cuda()
    a0 = ops.split_nchw(x0)
    layers, f64 = [], []
    cin = 64
    for k in range(4):
        w = (torch.randn(128, cin, 3, 3) / np.sqrt(9 * cin)).cuda()
        b = (torch.randn(128) * 0.1).cuda()
        layers.append(ops.split_weights(w) + (b,))
        f64.append((w.double(), b.double()))
        cin = 128
    ref = a0
    for w_hi, w_lo, b in layers:
        ref = ops.conv3x3_split(ref, w_hi, w_lo, b)
    got = ops.conv3x3_split_trunk(a0, layers)
    g, r = ops.merge_nchw(got), ops.merge_nchw(ref)
    scale = float(r.abs().max())
    assert float((g - r).abs().max()) < 4e-6 * scale, (float((g - r).abs().max()), scale)
    # ... and neither is further from the float64 result than the other by more than that
    want = x0.double()
    for w, b in f64:
        want = torch.relu(torch.nn.functional.conv2d(want, w, b, padding=1))
    eg, er = float((g.double() - want).abs().max()), float((r.double() - want).abs().max())
    assert eg < 2e-5 * scale and er < 2e-5 * scale, (eg, er, scale)
