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
    assert rel_err(dw, w.grad) < 4 * max(rel_err(w32.grad, w.grad), 2e-7)
