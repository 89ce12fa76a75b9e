#!/usr/bin/env python3
# NOTE (round 3): the -D variant code this script builds (ROW_EXP_* / TRUNK_EXP_* / ROW_OLD_*) was removed from the
# product sources; it lives in the history: run this from a checkout of commit b61d6ed.
"""Per-phase cycle stamps of the LDS-resident trunk (library built with -DTRUNK_EXP_STAMPS):
IAGO_HIP_LIB=tools/_build/trunk_stamps.so python3 tools/exp_trunk_stamps.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import network, ops
torch.manual_seed(0)
m = network.Value().cuda().eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator().manual_seed(1)
own = torch.randint(0, 2 ** 62, (n,), generator=g).cuda()
opp = torch.randint(0, 2 ** 62, (n,), generator=g).cuda() & ~own
with torch.no_grad():
    a = ops.value_stem_boards(own, opp, m.block1.conv.weight, m.block1.conv.bias)
    layers = [m._split_weights(k) + (getattr(m, "block%d" % k).conv.bias,) for k in range(2, 9)]
    for _ in range(3):
        out = ops.conv3x3_split_trunk(a, layers)
    torch.cuda.synchronize()
    for base in ((0, 64) if n >= 404 else (0,)):
        st = out.hi.view(torch.int64).reshape(-1)[base:base + 22].cpu().tolist()
        print("block", 0 if base == 0 else 100, "s_memtime ticks (100 MHz?) per layer: K loop / epilogue / barrier")
        for L in range(7):
            print("  layer %d: %6d %6d %6d" % (L, st[3 * L + 1] - st[3 * L], st[3 * L + 2] - st[3 * L + 1], st[3 * L + 3] - st[3 * L + 2]))
        print("  total", st[21])
