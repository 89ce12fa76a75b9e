"""Host-side weight layouts of the convolution kernels (no GPU): what
ops.split_weights / ops.f32_weights hand to iago_conv3x3_split / iago_conv3x3_f32
(include/iago_hip.h) for Block.conv.W of network.py:5-13."""
import numpy as np
import torch

from iago_amd import ops


def test_split_weights_layout_and_value():
    torch.manual_seed(0)
    w = torch.randn(128, 64, 3, 3) * torch.logspace(-4, 1, 64).view(1, 64, 1, 1)
    hi, lo = ops.split_weights(w)
    assert hi.shape == lo.shape == (4, 3, 3, 128, 16) and hi.dtype == lo.dtype == torch.float16
    rec = hi.float() + lo.float() / 2048.0                 # [cb][ky][kx][co][16]
    rec = rec.permute(3, 0, 4, 1, 2).reshape(128, 64, 3, 3)  # co, (cb, c16), ky, kx
    excess = ((rec - w).abs() - 2.0 ** -21 * w.abs()).max().item()
    assert excess < 1e-10
    # element [cb][ky][kx][co][c] is W[co][16 cb + c][ky][kx]
    assert float(hi[2, 1, 0, 77, 5]) == float(w[77, 37, 1, 0].half())


def test_f32_weights_layout():
    w = torch.arange(128 * 128 * 9, dtype=torch.float32).reshape(128, 128, 3, 3)
    w4 = ops.f32_weights(w)
    assert w4.shape == (4, 9, 128, 32)
    # element [g][tap][ci][c] is W[32 g + c][ci][tap // 3][tap % 3]
    for g, tap, ci, c in ((0, 0, 0, 0), (3, 8, 127, 31), (1, 5, 64, 7)):
        assert float(w4[g, tap, ci, c]) == float(w[32 * g + c, ci, tap // 3, tap % 3])


def test_layout_helpers_reject_other_shapes():
    import pytest
    with pytest.raises(ValueError):
        ops.split_weights(torch.zeros(64, 64, 3, 3))
    with pytest.raises(ValueError):
        ops.split_weights(torch.zeros(128, 8, 3, 3))
    with pytest.raises(ValueError):
        ops.f32_weights(torch.zeros(128, 32, 3, 3))
