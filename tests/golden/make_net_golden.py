#!/usr/bin/env python3
"""Golden vectors for the reference's SHIPPED networks (models/sl_model.npz,
models/value_model.npz; loaded by MCTS.py:82-85) -- run in the build container:

    python tests/golden/make_net_golden.py

Writes (data only):
  tests/golden/sl_model.npz, value_model.npz   the reference's checkpoints, byte for byte
                                               (trained parameters: data, not code)
  tests/golden/nets_shipped.npz                256 traced positions (own = side to move) and
                                               the float64 outputs of oracle/nets_np.py on them:
                                               SLPolicy probabilities (256, 64), Value (256,)

The positions are the two known-answer inputs of SURVEY.md section 8a (start position,
colour 1 / colour 2 to move) followed by 254 positions of the golden rule traces
(tests/golden/rules.npz, recorded from the reference's rl_env.py), spread over all plies,
both colours and the handicap starts.  Chainer is not installable here and the reference
holds no net outputs, so the expected values come from the float64 restatement of
network.py:34-47,83-96 (oracle/nets_np.py: PARITY UNPINNED BY THE REFERENCE); what this
fixture pins is that every GPU path evaluates the REAL weights -- near-one-hot policy
outputs, trained dynamic range -- within 1e-5 of that restatement, and the section 8a
known answers, which were measured by an independent torch restatement.
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_MODELS = "/root/reference/models"

from oracle import nets_np  # noqa: E402

START_P1 = 0x0000000810000000  # colour 1: (3,4), (4,3)
START_P2 = 0x0000001008000000


def planes(own, opp):
    """game.py:168-174: channel 0 = opponent of the side to move, channel 1 = side to move."""
    x = np.zeros((len(own), 2, 64), np.float32)
    for i, (o, p) in enumerate(zip(own, opp)):
        for a in range(64):
            x[i, 0, a] = (int(p) >> a) & 1
            x[i, 1, a] = (int(o) >> a) & 1
    return x.reshape(-1, 2, 8, 8)


def main():
    for name in ("sl_model.npz", "value_model.npz"):
        shutil.copyfile(os.path.join(REF_MODELS, name), os.path.join(HERE, name))
        os.chmod(os.path.join(HERE, name), 0o644)
    tr = np.load(os.path.join(HERE, "rules.npz"))["trace"]
    own, opp = [START_P1, START_P2], [START_P2, START_P1]
    seen = set(zip(own, opp))
    idx = np.linspace(0, len(tr) - 1, 400).astype(int)
    for i in idx:
        p1, p2, color = int(tr[i, 0]), int(tr[i, 1]), int(tr[i, 2])
        o, p = (p1, p2) if color == 1 else (p2, p1)
        if (o, p) in seen:
            continue
        seen.add((o, p))
        own.append(o)
        opp.append(p)
        if len(own) == 256:
            break
    assert len(own) == 256
    own, opp = np.array(own, np.uint64), np.array(opp, np.uint64)
    x = planes(own, opp)
    sl = dict(np.load(os.path.join(HERE, "sl_model.npz")))
    va = dict(np.load(os.path.join(HERE, "value_model.npz")))
    probs = np.concatenate([nets_np.sl_policy(x[i:i + 32], sl) for i in range(0, 256, 32)])
    value = np.concatenate([nets_np.value(x[i:i + 32], va) for i in range(0, 256, 32)])
    # SURVEY.md section 8a known answers (independent torch-CPU fp32 restatement)
    assert abs(probs[0, 44] - 0.99992) < 1e-5 and abs(probs[0, 37] - 7.4479e-05) < 1e-8
    assert abs(probs[1, 43] - 1.0) < 1e-5 and abs(probs[1, 29] - 7.8165e-07) < 1e-10
    assert abs(value[0] + 0.026380) < 1e-6 and abs(value[1] + 0.036053) < 1e-6
    np.savez_compressed(os.path.join(HERE, "nets_shipped.npz"), own=own, opp=opp,
                        sl_probs=probs, value=value)
    print("positions", len(own), "stones", sorted(set(bin(int(o | p)).count("1") for o, p in zip(own, opp)))[:3],
          "... max policy entry: median %.4f" % np.median(probs.max(axis=1)),
          "value range [%.3f, %.3f]" % (value.min(), value.max()))


if __name__ == "__main__":
    main()
