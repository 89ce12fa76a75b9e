import os, sys, time, json, ctypes
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from iago_amd import ops
g = json.load(open(os.path.join(R, 'tests', 'golden', 'simulate.json')))
w = ops.RolloutWeights(g['shipped_w'], g['shipped_b'])
B = 4096
own = torch.full((B,), 0x0000000810000000, dtype=torch.int64, device='cuda')
opp = torch.full((B,), 0x0000001008000000, dtype=torch.int64, device='cuda')
K = 400
preps = [ops.rollout_prepare(own, opp, w, seed=1, id_base=k * B) for k in range(K)]
for S in (1, 2, 4, 8, 16):
    streams = [torch.cuda.Stream() for _ in range(S)]
    sptr = [ctypes.c_void_p(s.cuda_stream) for s in streams]
    for k, p in enumerate(preps[:32]): p.launch(sptr[k % S])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k, p in enumerate(preps): p.launch(sptr[k % S])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('S=%d issue %.2f us/launch, total %.2f us/launch' % (S, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
# graph capture
S = 4
streams = [torch.cuda.Stream() for _ in range(S)]
gph = torch.cuda.CUDAGraph()
with torch.cuda.graph(gph):
    main = torch.cuda.current_stream()
    for s in streams: s.wait_stream(main)
    for k, p in enumerate(preps):
        with torch.cuda.stream(streams[k % S]):
            assert p.launch() == 0
    for s in streams: main.wait_stream(s)
torch.cuda.synchronize()
gph.replay(); torch.cuda.synchronize()
t0 = time.perf_counter(); gph.replay(); torch.cuda.synchronize(); t1 = time.perf_counter()
print('graph S=4: %.2f us/launch' % ((t1 - t0) / K * 1e6))
for S in (8, 16):
    streams = [torch.cuda.Stream() for _ in range(S)]
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        main = torch.cuda.current_stream()
        for s in streams: s.wait_stream(main)
        for k, p in enumerate(preps):
            with torch.cuda.stream(streams[k % S]):
                assert p.launch() == 0
        for s in streams: main.wait_stream(s)
    torch.cuda.synchronize()
    gph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); gph.replay(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print('graph S=%d: %.2f us/launch' % (S, (t1 - t0) / K * 1e6))
