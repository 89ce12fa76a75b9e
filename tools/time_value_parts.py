#!/usr/bin/env python3
"""Value forward at n boards, per part (HIP events, back-to-back launches of one part):
stem, trunk (blocks 2..8), head, the three in sequence, and the fused launch."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iago_amd import network, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
torch.manual_seed(0)
m = network.Value().cuda().eval()
g = torch.Generator().manual_seed(1)
own = torch.randint(0, 2 ** 62, (n,), generator=g).cuda()
opp = torch.randint(0, 2 ** 62, (n,), generator=g).cuda() & ~own
layers = [m._split_weights(k) + (getattr(m, "block%d" % k).conv.bias,) for k in range(2, 9)]
ovf = m._overflow_flag(own.device)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


with torch.no_grad():
    a0 = ops.value_stem_boards(own, opp, m.block1.conv.weight, m.block1.conv.bias, overflow=ovf)
    a8 = ops.conv3x3_split_trunk(a0, layers, overflow=ovf)
    t_stem = timed(lambda: ops.value_stem_boards(own, opp, m.block1.conv.weight, m.block1.conv.bias, overflow=ovf))
    t_trunk = timed(lambda: ops.conv3x3_split_trunk(a0, layers, overflow=ovf))
    t_head = timed(lambda: ops.value_head(a8, m.block9.conv.weight, m.block9.conv.bias, m.fc10.weight, m.fc11.weight))
    m.fused = False
    t_three = timed(lambda: m.forward_boards(own, opp))
    m.fused = True
    t_fused = timed(lambda: m.forward_boards(own, opp))
print("n = %d: stem %.1f us, trunk %.1f us, head %.1f us (sum %.1f); three launches %.1f us; fused %.1f us"
      % (n, t_stem, t_trunk, t_head, t_stem + t_trunk + t_head, t_three, t_fused))
