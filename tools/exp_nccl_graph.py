import os, sys, json, time, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/iago_amd') else os.getcwd())
mode = sys.argv[1]
# (GPU_MAX_HW_QUEUES: the runtime default; see bench.py)
if mode.startswith("nccl"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    kw = dict(device_id=torch.device("cuda", 0)) if mode == "nccl_eager" else {}
    dist.init_process_group("nccl", rank=0, world_size=1, **kw)
    if mode == "nccl_used":
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    if mode in ("nccl_gather", "nccl_barrier"):
        comm = torch.cuda.Stream()
        buf = torch.zeros(24_000_000, dtype=torch.uint8, device="cuda"); out = torch.empty_like(buf)
        main = torch.cuda.current_stream()
        if mode == "nccl_gather":
            for _ in range(3):
                ev = torch.cuda.Event(); ev.record(main)
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    dist.all_gather_into_tensor(out, buf)
            main.wait_stream(comm)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
from iago_amd import engine, network, ops
g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "..", "tests", "golden", "simulate.json")))
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
w = ops.RolloutWeights(g["shipped_w"], g["shipped_b"])
G = 1024
m = engine.BatchedMCTS(G, policy, value, w, n_thr=15, capacity=8192, seed=3, use_graph=True)
own = torch.full((G,), engine.START_OWN, dtype=torch.int64, device="cuda")
opp = torch.full((G,), engine.START_OPP, dtype=torch.int64, device="cuda")
act = torch.ones(G, dtype=torch.uint8, device="cuda")
m.search(own, opp, act, 104)
torch.cuda.synchronize()
if mode == "nccl_barrier2":
    dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m.search(own, opp, act, 104)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(mode, "overlap", m.lookahead_overlap, "%.3f M leaf-evals/s" % (5 * 104 * G / dt / 1e6))
