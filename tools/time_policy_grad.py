"""Times the kernels of the split-f16 REINFORCE gradients (csrc/policy_grad_kernels.hip) on synthetic data."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import ops


def timed(f, reps=20):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    ops.WGRAD_GROUPS = int(os.environ.get("WGRAD_GROUPS", ops.WGRAD_GROUPS))
    for cin in (128, 64):
        x = ops.split_nchw(torch.relu(torch.randn(n, cin, 8, 8, device="cuda")))
        dy = ops.split_nchw(torch.randn(n, 128, 8, 8, device="cuda"))
        part = torch.empty((ops.WGRAD_GROUPS, 9, 128, cin), dtype=torch.float32, device="cuda")
        us = timed(lambda: ops.conv3x3_wgrad_split(dy, x, part=part, groups=ops.WGRAD_GROUPS))
        flop = 2.0 * n * 64 * 128 * cin * 9
        print("wgrad n=%d cin=%d: %.1f us, %.0f TFLOP/s algorithmic, %.0f executed" % (n, cin, us, flop / us / 1e6, 3 * flop / us / 1e6))


if __name__ == "__main__":
    main()


def whole(n=1900):
    """The whole update (gradients + ChainerAdam) on real rows, native against autograd."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_policy_grad_gpu import _rows
    from iago_amd import network, train_rl
    own, opp, act, z = _rows(n, seed=1)
    model = network.SLPolicy().cuda()
    opt = train_rl.ChainerAdam(model)

    def native():
        model.reinforce_grads(own, opp, act, z)

    def native_step():
        model.reinforce_grads(own, opp, act, z)
        opt.update()

    def autograd_step():
        model.train()
        for p in model.parameters():
            p.grad = None
        train_rl.reinforce_loss(model, own, opp, act, z, pad_to=512).backward()
        opt.update()

    print("n=%d rows: gradients %.0f us, gradients + Adam %.0f us" % (n, timed(native), timed(native_step)))
    if len(sys.argv) > 2:
        print("autograd + Adam %.0f us" % timed(autograd_step, reps=5))


if __name__ == "__main__":
    whole()
