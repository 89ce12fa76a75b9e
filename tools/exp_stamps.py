"""Diagnostic: per-segment cycle shares of a rollout turn (EXP_STAMPS build)."""
import os, sys, subprocess, json, shutil
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
so = os.path.join(R, 'iago_amd', 'libiago_hip.so')
shutil.copy(so, '/tmp/orig.so')
subprocess.check_call('/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DEXP_STAMPS -o %s %s/iago_amd/csrc/*.hip' % (so, R), shell=True)
try:
    from iago_amd import ops
    g = json.load(open(os.path.join(R, 'tests', 'golden', 'simulate.json')))
    w = ops.RolloutWeights(g['shipped_w'], g['shipped_b'])
    B = 4096
    own = torch.full((B,), 0x0000000810000000, dtype=torch.int64, device='cuda')
    opp = torch.full((B,), 0x0000001008000000, dtype=torch.int64, device='cuda')
    for it in range(3):
        r = ops.rollout(own, opp, w, seed=1, stream_id=it, want_final=True, want_turns=True)
    torch.cuda.synchronize()
    a = ops.tensor_to_bits(r.final_own); b = ops.tensor_to_bits(r.final_opp)
    segs = np.stack([a & 0x1FFFFF, (a >> 21) & 0x1FFFFF, (a >> 42) & 0x1FFFFF, b & 0x1FFFFF, (b >> 21) & 0x1FFFFF, (b >> 42) & 0x1FFFFF], 1).astype(np.float64)
    # all boards of a wave share stamps; turns executed by the wave = max over its 8 boards
    nt = r.n_turns.cpu().numpy().reshape(-1, 8).max(1).repeat(8)
    names = ['window+E issue', 'movegen+LM issue', 'products', 'cumsum/scan/count', 'flips', 'update']
    tot = segs.sum(1)
    print('mean cycles per wave (stamped build): %.0f, turns %.1f' % (tot.mean(), nt.mean()))
    for i, n in enumerate(names):
        print('%-20s %7.1f cycles/turn  %5.1f%%' % (n, (segs[:, i] / nt).mean(), 100 * segs[:, i].sum() / tot.sum()))
finally:
    shutil.copy('/tmp/orig.so', so)
