"""Diagnostic: phases of a wave's life in the 16-lanes-per-board rollout kernel
# NOTE (round 3): the -D variant code this script builds (ROW_EXP_* / TRUNK_EXP_* / ROW_OLD_*) was removed from the
# product sources; it lives in the history: run this from a checkout of commit b61d6ed.
(-DROW_EXP_STAMPS build): table staging, setup (loads, Philox), the turn loop."""
import os, sys, subprocess, json, shutil
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
so = os.path.join(R, 'iago_amd', 'libiago_hip.so')
shutil.copy(so, '/tmp/orig.so')
subprocess.check_call('/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared %s -o %s %s/iago_amd/csrc/*.hip' % (' '.join(['-DROW_EXP_STAMPS'] + sys.argv[1:]), so, R), shell=True)
try:
    from iago_amd import ops
    g = json.load(open(os.path.join(R, 'tests', 'golden', 'simulate.json')))
    w = ops.RolloutWeights(g['shipped_w'], g['shipped_b'])
    B = 4096
    own = torch.full((B,), 0x0000000810000000, dtype=torch.int64, device='cuda')
    opp = torch.full((B,), 0x0000001008000000, dtype=torch.int64, device='cuda')
    for it in range(20):
        r = ops.rollout(own, opp, w, seed=1, stream_id=it, want_final=True, want_turns=True)
    torch.cuda.synchronize()
    a = ops.tensor_to_bits(r.final_own); t0 = ops.tensor_to_bits(r.final_opp).astype(np.float64)
    segs = np.stack([a & 0x1FFFFF, (a >> 21) & 0x1FFFFF, (a >> 42) & 0x1FFFFF], 1).astype(np.float64)
    nt = r.n_turns.cpu().numpy().reshape(-1, 4).max(1).astype(np.float64)
    nt4 = np.ceil(nt / 2) * 2  # the loop leaves after a pair of turns
    w_ = segs[::4]
    print('waves %d; cycles: staging %.0f, setup %.0f, loop %.0f (min %.0f max %.0f); turns run %.1f; loop cycles/turn %.1f'
          % (len(w_), w_[:, 0].mean(), w_[:, 1].mean(), w_[:, 2].mean(), w_[:, 2].min(), w_[:, 2].max(), nt4.mean(),
             (w_[:, 2] / nt4).mean()))
    print('wave start spread (cycles of s_memtime): %.0f; last end - first start %.0f'
          % (t0.max() - t0.min(), (t0[::4] + w_.sum(1)).max() - t0.min()))
finally:
    shutil.copy('/tmp/orig.so', so)
