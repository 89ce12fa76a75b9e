"""Which part of bench.py's flow slows the PV-MCTS leg when a process group exists?
python tools/exp_nccl_bench_leg.py <dist:0|1> <rollout_leg_first:0|1>"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (GPU_MAX_HW_QUEUES: the runtime default; see bench.py)
import torch
import bench
use_dist, first = int(sys.argv[1]), int(sys.argv[2])
torch.cuda.set_device(0)
dist = None
if use_dist:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
args = types.SimpleNamespace(boards=4096, steps=20, warmup=5, repeats=0, streams=32, rollout_only=False, mcts_only=True,
                             large_boards=0)
if first:
    bench.rollout_leg(args, 1, 0, dist)
    if len(sys.argv) > 3 and sys.argv[3] == "empty":
        import gc
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    if len(sys.argv) > 3 and sys.argv[3] == "avoid":
        pass
out = bench.mcts_leg(1024, 100, 8, False, 1, 0, dist)
print("dist", use_dist, "rollout leg first", first, "%.3f M leaf-evals/s" % (out["leaf_evals_per_sec"] / 1e6))
