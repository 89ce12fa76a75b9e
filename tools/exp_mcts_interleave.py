"""Experiment: two half-batch MCTS engines interleaved on two streams vs one full batch."""
import time, sys, os, json
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from iago_amd import engine, network, ops
g = json.load(open(os.path.join(R, 'tests', 'golden', 'simulate.json')))
torch.manual_seed(0)
policy = network.SLPolicy().cuda().eval(); value = network.Value().cuda().eval()
w = ops.RolloutWeights(g['shipped_w'], g['shipped_b'])
def mk(B, base):
    m = engine.BatchedMCTS(B, policy, value, w, capacity=2048, seed=1, game_id_base=base)
    m.warmup(); return m
def boards(B):
    return (torch.full((B,), engine.START_OWN, dtype=torch.int64, device='cuda'),
            torch.full((B,), engine.START_OPP, dtype=torch.int64, device='cuda'),
            torch.ones(B, dtype=torch.uint8, device='cuda'))
NS = 150
m = mk(1024, 0); o, p, a = boards(1024)
for _ in range(20): m.simulate(o, p, a)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(NS): m.simulate(o, p, a)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print('1 x 1024: %.3f ms/sim  %.0f leaf-evals/s' % (dt / NS * 1e3, 1024 * NS / dt))
for parts in (2, 4):
    B = 1024 // parts
    ms = [mk(B, i * B) for i in range(parts)]
    bs = [boards(B) for _ in range(parts)]
    ss = [torch.cuda.Stream() for _ in range(parts)]
    torch.cuda.synchronize()
    def run(n):
        for _ in range(n):
            for mm, (o, p, a), s in zip(ms, bs, ss):
                with torch.cuda.stream(s):
                    mm.simulate(o, p, a)
    run(20); torch.cuda.synchronize(); t = time.perf_counter()
    run(NS); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print('%d x %d interleaved: %.3f ms/sim-round  %.0f leaf-evals/s' % (parts, B, dt / NS * 1e3, 1024 * NS / dt))
