#!/usr/bin/env python3
"""Lab tool: the Value net's pair / single walks by FOUR waves (one per SIMD, 32 output channels each: the product) against
EIGHT (two per SIMD, 16 channels each: IAGO_WALK_WAVES=8 in iago_value_forward_batch), every CU walking pair after pair:
224 workgroups stride over 224 x 2 x 24 rows.  Values bit-equal; time per pair of boards."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iago_amd import network  # noqa: E402

torch.manual_seed(0)
m = network.Value().cuda().eval()
for tb in (2, 1):
    wgs, passes = 224, 24
    n = wgs * tb * passes
    g = torch.Generator().manual_seed(1)
    own = torch.randint(0, 2 ** 62, (n,), generator=g).cuda()
    opp = torch.randint(0, 2 ** 62, (n,), generator=g).cuda() & ~own
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    res = {}
    for waves in ("4", "8", "4", "8"):
        os.environ["IAGO_WALK_WAVES"] = waves
        out = torch.zeros(n, dtype=torch.float32, device="cuda")
        with torch.no_grad():
            m.forward_boards_batch(own, opp, n_dev, out, tb, wgs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                m.forward_boards_batch(own, opp, n_dev, out, tb, wgs)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        res.setdefault(waves, out.clone())
        print("%d board(s) per walk, %s waves: %.2f ms per launch = %.1f us per walk (%.1f per board)"
              % (tb, waves, dt * 1e3, dt * 1e6 / passes, dt * 1e6 / passes / tb), flush=True)
    print("  values bit-equal:", bool(torch.equal(res["4"], res["8"])))
