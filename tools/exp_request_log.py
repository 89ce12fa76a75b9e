"""Lab tool: positions the games of a whole batch ask the nets for more than once (what a position table can answer),
by kind of request and by quarter of the batch.  Needs the logging variant of the search kernel:

    python tools/build_search_variants.py
    IAGO_PERSISTENT_TABLE=0 IAGO_HIP_LIB=$PWD/tools/_build/search_log.so python tools/exp_request_log.py
"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iago_amd import engine, network, ops
w, b = bench.shipped_rollout_weights()
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
m = engine.BatchedMCTS(1024, policy, value, ops.RolloutWeights(w, b), n_thr=15, seed=7, persistent=True,
                       capacity=engine.suggest_capacity(100, 15, moves=64))
m.trace = torch.zeros((1_300_000, 4), dtype=torch.int64, device="cuda")
res = engine.SelfPlayEngine(m).play(100, record=False)
torch.cuda.synchronize()
t = m.trace.cpu().numpy()
allt = t
for kind, name in ((2, "policy"), (1, "value")):
    t = allt[allt[:, 3] == kind]
    print(name, "requests", len(t))
    if not len(t):
        continue
    keys = t[:, 0].astype(np.uint64).astype(object) * (1 << 64) + t[:, 1].astype(np.uint64).astype(object)
    u = len(set(keys.tolist()))
    print(" unique positions", u, "duplicates %.3f" % (1 - u / len(t)))
    gk = set(zip(t[:, 2].tolist(), t[:, 0].tolist(), t[:, 1].tolist()))
    print(" unique (game, position)", len(gk), "within-game duplicates %.4f" % (1 - len(gk) / len(t)))
    # by quarter of the log (requests are logged in time order)
    q = len(t) // 4
    seen = set()
    for i in range(4):
        part = keys[i * q:(i + 1) * q].tolist()
        dup = sum(1 for k in part if k in seen)
        # (duplicates inside the part itself count too)
        s2 = set()
        d2 = 0
        for k in part:
            if k in s2 and k not in seen:
                d2 += 1
            s2.add(k)
        seen |= s2
        print("  quarter %d: %.3f seen before" % (i, (dup + d2) / max(1, len(part))))
t = allt[allt[:, 3] == 1]
print("value requests", len(t))
keys = t[:, 0].astype(np.uint64).astype(object) * (1 << 64) + t[:, 1].astype(np.uint64).astype(object)
u = len(set(keys.tolist()))
print("unique positions", u, "duplicates %.3f" % (1 - u / len(t)))
# within-game duplicates only
gk = set(zip(t[:, 2].tolist(), t[:, 0].tolist(), t[:, 1].tolist()))
print("unique (game, position)", len(gk), "within-game duplicates %.4f" % (1 - len(gk) / len(t)))
