#!/usr/bin/env python3
"""Lab tool: max |p - p_float64| of the one-launch three-piece SLPolicy kernel on the shipped net (the 1e-5 bar of
BASELINE.json's north star), by number of rows; and the Value net's error."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iago_amd import network, ops  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
gold = np.load(os.path.join(G, "nets_shipped.npz"))
policy = network.SLPolicy().load_npz(os.path.join(G, "sl_model.npz")).cuda().eval()
value = network.Value().load_npz(os.path.join(G, "value_model.npz")).cuda().eval()
own, opp = gold["own"], gold["opp"]
for n in (1, 64, 255, 256):
    idx = np.arange(n) % len(own)
    o, p = ops.bits_to_tensor(own[idx]), ops.bits_to_tensor(opp[idx])
    pr = policy.forward_boards_split3(o, p).cpu().numpy().astype(np.float64)
    print("SLPolicy rows %4d: max |dp| %.4g" % (n, np.max(np.abs(pr - gold["sl_probs"][idx]))))
idx = np.arange(len(own))
o, p = ops.bits_to_tensor(own), ops.bits_to_tensor(opp)
v = value.forward_boards(o, p)
if v is not None:
    print("Value rows %d: max |dv| %.4g" % (len(own), float(np.max(np.abs(v.cpu().numpy().astype(np.float64) - gold["value"][idx])))))
