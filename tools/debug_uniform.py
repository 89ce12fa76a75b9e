import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import ops
from oracle import oracle as orc
from tests.gpu_util import random_positions, state_of
n, seed, id_base = 2048, 4, 0
own, opp = random_positions(n, seed=seed)
own[: n // 3] = 0x0000000810000000
opp[: n // 3] = 0x0000001008000000
res = ops.rollout(ops.bits_to_tensor(own), ops.bits_to_tensor(opp), None, seed=seed, id_base=id_base, want_final=True, want_turns=True, want_trace=True)
torch.cuda.synchronize()
tr = res.trace.cpu().numpy(); nt = res.n_turns.cpu().numpy()
bad = 0
for b in range(n):
    oz, final, otr = orc.random_playout(state_of(own[b], opp[b]), 1, seed=seed, game_id=id_base + b)
    got = [(-1 if a == 0xFF else int(a)) for a in tr[:nt[b], b]]
    if got != otr:
        bad += 1
        t = next(i for i in range(min(len(got), len(otr))) if got[i] != otr[i])
        s = state_of(own[b], opp[b]); c = 1
        for a in otr[:t]:
            orc.place_stone(s, a, c); c = 3 - c
        la = orc.legal_actions(s, c)
        u = orc.uniform(seed, id_base + b, t)
        print('game', b, 'turn', t, 'gpu', got[t], 'oracle', otr[t], 'legal', la, 'n', len(la), 'u', repr(u), 'u*n', repr(np.float32(u) * np.float32(len(la))))
        if bad > 5: break
print('bad games', bad)
