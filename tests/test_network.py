"""PyTorch modules of iago_amd.network against the float64 numpy restatement of
network.py (oracle/nets_np.py); tolerance 1e-5 on probabilities as
BASELINE.json's north_star states.  CPU versions here, GPU versions marked."""
import os

import numpy as np
import pytest
import torch

from iago_amd import network
from oracle import nets_np
from oracle import oracle as orc

TOL = 1e-5
REF_MODELS = "/root/reference/models"  # rollout_model.npz only; sl / value live in tests/golden


def sample_planes(n, seed):
    rs = np.random.RandomState(seed)
    xs = []
    for g in range(n):
        z, final, tr = orc.random_playout(orc.initial_state(), 1, seed=seed, game_id=g)
        s, color = orc.initial_state(), 1
        for a in tr[: rs.randint(0, len(tr))]:
            orc.place_stone(s, a, color)
            color = 3 - color
        xs.append(orc.make_state_var(s, color)[0])
    return np.stack(xs).astype(np.float32)


def build(kind, seed, device="cpu"):
    params = nets_np.random_params(kind, seed)
    cls = {"sl": network.SLPolicy, "value": network.Value, "rollout": network.RolloutPolicy}[kind]
    m = cls().load_npz(params).to(device).eval()
    return m, params


ORACLE = {"sl": nets_np.sl_policy, "value": nets_np.value, "rollout": nets_np.rollout_policy}


@pytest.mark.parametrize("kind", ["sl", "value", "rollout"])
def test_modules_match_float64_oracle_cpu(kind):
    m, params = build(kind, seed=3)
    x = sample_planes(16, seed=5)
    with torch.no_grad():
        got = m(torch.from_numpy(x)).numpy()
    want = ORACLE[kind](x, params)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) < TOL
    if kind != "value":
        assert np.allclose(got.sum(axis=1), 1.0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sl", "value", "rollout"])
def test_modules_match_float64_oracle_gpu(kind):
    m, params = build(kind, seed=4, device="cuda")
    x = sample_planes(64, seed=6)
    with torch.no_grad():
        got = m(torch.from_numpy(x).cuda()).cpu().numpy()
    want = ORACLE[kind](x, params)
    assert np.max(np.abs(got - want)) < TOL


def test_npz_roundtrip_uses_reference_key_names(tmp_path):
    m = network.SLPolicy()
    keys = set(m.npz_dict())
    assert {"block1/conv/W", "block1/conv/b", "block8/conv/W", "conv9/W", "bias10/b"} <= keys
    assert len(keys) == 18
    v = network.Value()
    assert {"block9/conv/W", "block9/conv/b", "fc10/W", "fc11/W"} <= set(v.npz_dict())
    assert v.npz_dict()["fc10/W"].shape == (128, 64) and v.npz_dict()["fc11/W"].shape == (1, 128)
    r = network.RolloutPolicy()
    assert set(r.npz_dict()) == {"conv1/W", "bias2/b"}
    assert sum(p.numel() for p in m.parameters()) == 960768
    assert sum(p.numel() for p in v.parameters()) == 970049
    assert sum(p.numel() for p in r.parameters()) == 82
    path = str(tmp_path / "m.npz")
    m.save_npz(path)
    m2 = network.SLPolicy().load_npz(path)
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    with pytest.raises(KeyError):
        network.Value().load_npz(path)


def test_value_dropout_only_in_training_mode():
    v = network.Value()
    x = torch.from_numpy(sample_planes(4, seed=1))
    v.eval()
    with torch.no_grad():
        assert torch.equal(v(x), v(x))
    v.train()
    torch.manual_seed(0)
    a = v(x)
    b = v(x)
    assert not torch.equal(a, b)


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="shipped checkpoints live in /root/reference")
def test_shipped_checkpoints_known_answers():
    """SURVEY.md section 8a KATs (measured there with an independent torch
    restatement) + the float64 oracle on the same weights."""
    s = orc.initial_state()
    x = np.concatenate([orc.make_state_var(s, 1), orc.make_state_var(s, 2)]).astype(np.float32)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sl = network.SLPolicy().load_npz(os.path.join(gold, "sl_model.npz")).eval()
    va = network.Value().load_npz(os.path.join(gold, "value_model.npz")).eval()
    ro = network.RolloutPolicy().load_npz(os.path.join(REF_MODELS, "rollout_model.npz")).eval()
    with torch.no_grad():
        p = sl(torch.from_numpy(x)).numpy()
        v = va(torch.from_numpy(x)).numpy()
        r = ro(torch.from_numpy(x)).numpy()
    assert abs(p[0, 44] - 0.99992) < 1e-5 and abs(p[0, 37] - 7.4479e-05) < 1e-8
    assert abs(p[1, 43] - 1.0) < 1e-5 and abs(p[1, 29] - 7.8165e-07) < 1e-10
    assert abs(v[0] - (-0.026380)) < 1e-6 and abs(v[1] - (-0.036053)) < 1e-6
    assert abs(r[0, [19, 26, 37, 44]].sum() - 0.50168) < 1e-5
    assert abs(r[1, [20, 29, 34, 43]].sum() - 0.51121) < 1e-5
    for m, fn in ((sl, nets_np.sl_policy), (va, nets_np.value), (ro, nets_np.rollout_policy)):
        with torch.no_grad():
            got = m(torch.from_numpy(x)).numpy()
        assert np.max(np.abs(got - fn(x, m.npz_dict()))) < TOL
    # the 82 shipped rollout floats are also the golden data the benchmark uses
    import json
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "simulate.json")))
    w, b = ro.kernel_weights()
    assert np.array_equal(w, np.asarray(g["shipped_w"], np.float32))
    assert np.array_equal(b, np.asarray(g["shipped_b"], np.float32))
