#!/usr/bin/env python3
"""Golden runs of the reference's value-data generator (value_self_play.py:11-59,
gen_value_data.py:6-19) -- run in the build container only (needs /root/reference):

    python tests/golden/make_value_golden.py      -> tests/golden/value_data.json

value_self_play.py is imported UNMODIFIED under the stub chainer / numba modules of
make_golden.py plus two more stubs: the module `SLPolicy` it imports does not exist in the
reference tree (value_self_play.py:9), and `L.Classifier` only has to carry `.predictor`.
The two nets are replaced by stand-in callables with the reference's call signature
(`model.predictor(x).data` -> 64 numbers, to which value_self_play.py:143 applies its own
softmax): half of the cases use a RolloutPolicy-shaped net that returns probabilities (as
network.SLPolicy does: the double softmax then is almost uniform and most draws fall back
to random.choice, value_self_play.py:145-148), the other half the same net scaled by 24
(peaked after the softmax: the drawn cell is usually legal).  numpy.random.choice and
python's random.choice are replaced by recorders with the same selection rule so that the
uniforms that drove each game are part of the fixture.  Only data is written.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the chainer / numba stubs, imports the reference)

# the extra names value_self_play.py needs at import time
loss_pkg = types.ModuleType("chainer.functions.loss")
sce = types.ModuleType("chainer.functions.loss.softmax_cross_entropy")
sce.softmax_cross_entropy = lambda *a, **k: None
sys.modules["chainer.functions.loss"] = loss_pkg
sys.modules["chainer.functions.loss.softmax_cross_entropy"] = sce
sl_mod = types.ModuleType("SLPolicy")
sl_mod.SLPolicyNet = lambda *a, **k: None
sys.modules["SLPolicy"] = sl_mod


class _Classifier(object):
    def __init__(self, predictor, lossfun=None):
        self.predictor = predictor


sys.modules["chainer.links"].Classifier = _Classifier
import value_self_play as ref_vsp  # noqa: E402


class Draws(object):
    """One recorded stream for both selection calls of a game, in consumption order:
    ('c', u) numpy.random.choice(64, p) = searchsorted(cumsum(p)/sum, u, 'right');
    ('r', u) random.choice(seq) = seq[floor(u * len(seq))]."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.log = []

    def np_choice(self, n, p=None):
        u = self.rs.random_sample()
        self.log.append(["c", float(u)])
        cdf = np.asarray(p, dtype=np.float64).cumsum()
        cdf /= cdf[-1]
        return int(cdf.searchsorted(u, side="right"))

    def py_choice(self, seq):
        u = self.rs.random_sample()
        self.log.append(["r", float(u)])
        return seq[min(int(u * len(seq)), len(seq) - 1)]


class Scaled(object):
    def __init__(self, base, k):
        self.base, self.k = base, np.float32(k)

    def __call__(self, x):
        return mg._Out((self.base(x).data * self.k).astype(np.float32))


def main():
    rs = np.random.RandomState(29)
    cases = []
    real_np, real_py = np.random.choice, ref_vsp.random.choice
    stops = [4, 5, 9, 17, 30, 44, 57, 63, 12, 26, 38, 61, 64, 6]
    try:
        for g, stop in enumerate(stops):
            w0, b0 = rs.randn(18).astype(np.float32), (0.5 * rs.randn(64)).astype(np.float32)
            w1, b1 = rs.randn(18).astype(np.float32), (0.5 * rs.randn(64)).astype(np.float32)
            scale = 1.0 if g % 2 == 0 else 24.0
            d = Draws(9000 + g)
            np.random.choice = d.np_choice
            ref_vsp.random.choice = d.py_choice
            sp = ref_vsp.SelfPlay(stop)
            sp.model0 = mg._Pred(Scaled(mg.FakeRollout(w0, b0), scale))
            sp.model1 = mg._Pred(Scaled(mg.FakeRollout(w1, b1), scale))
            state, result = sp()
            np.random.choice, ref_vsp.random.choice = real_np, real_py
            s1, s2 = mg.to_bits(state)       # the mover's stones are 2s in the returned board
            f1, f2 = mg.to_bits(sp.state)
            cases.append(dict(stop_num=int(stop), scale=scale, w0=w0.tolist(), b0=b0.tolist(),
                              w1=w1.tolist(), b1=b1.tolist(), draws=d.log, own=s2, opp=s1,
                              result=int(result), final_p1=f1, final_p2=f2,
                              stone_num=int(sp.stone_num)))
    finally:
        np.random.choice, ref_vsp.random.choice = real_np, real_py
    with open(os.path.join(HERE, "value_data.json"), "w") as f:
        json.dump(cases, f)
    print("cases", len(cases), "results", [c["result"] for c in cases],
          "draws", [len(c["draws"]) for c in cases],
          "fallbacks", [sum(1 for k, _ in c["draws"] if k == "r") for c in cases])


if __name__ == "__main__":
    main()
