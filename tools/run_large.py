#!/usr/bin/env python3
"""One launch size of the rollout, repeated: what tools/profile_large.sh profiles.
    python3 tools/run_large.py [boards=1048576] [launches=10]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iago_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = json.load(open(os.path.join(ROOT, "tests", "golden", "simulate.json")))
weights = ops.RolloutWeights(np.asarray(g["shipped_w"], np.float32), np.asarray(g["shipped_b"], np.float32))
own = torch.full((n,), 0x0000000810000000, dtype=torch.int64, device="cuda")
opp = torch.full((n,), 0x0000001008000000, dtype=torch.int64, device="cuda")
out = ops.RolloutResult()
out.z = torch.empty(n, dtype=torch.int8, device="cuda")
out.n_turns = torch.empty(n, dtype=torch.uint8, device="cuda")
for k in range(3):
    ops.rollout(own, opp, weights, seed=1, id_base=0, stream_id=k, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(reps):
    ops.rollout(own, opp, weights, seed=2, id_base=0, stream_id=k, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(json.dumps({"boards": n, "kernel_ms": ms, "games_per_sec": n / ms * 1e3,
                  "board_steps_per_game": float(out.n_turns.to(torch.float64).mean().item())}))
