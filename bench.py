#!/usr/bin/env python3
"""Benchmark of the MI355X self-play hot path (contract: see DESIGN.md, "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): per GPU, 4096 parallel Othello boards from
the standard start position are played to the end by the fused HIP rollout
kernel with the reference's shipped RolloutPolicy weights (82 floats, kept as
golden data in tests/golden/simulate.json) -- rollout-policy-only self-play.
One step = 4096 finished games per GPU.  Steps are independent batches: by default
up to 2048 consecutive steps share one kernel launch (the default 2000 steps are one
launch of 8,192,000 boards), which the library plays with its lane-per-board rollout
kernel: the ramp and the tail of a launch (waves of different game lengths leave CUs
idle at its end) are paid once (256 steps per launch: -10 %).  `--steps-per-launch 1` issues every
step as its own launch of the 8-lanes-per-board kernel, overlapped on 32 HIP
streams / 16 hardware queues (a single 4096-board launch is only 512 waves).
With N > 1 every rank plays its own 4096-board shard (weak scaling, Philox streams
keyed by a rank-major global game id) and the finished (final boards, z, turns)
tuples of the whole round are all-gathered over RCCL inside the timed region: the
run is then two launches, and the tuples of the first travel as one collective on a
side stream beside the second.

Rank 0 prints ONE JSON line.  `roofline` prices the rollout kernel against the
HBM roof with SURVEY.md section 8(d)'s algorithmic bytes (33 B per board-step);
`cpu_baseline` times the CPU oracle (oracle/, a C port of the reference's
Python loops) on the host cores over a bounded sample of the same workload.
Extra objects: `large_batch` (the same kernel at 1M boards in one launch),
`mcts` (BASELINE configs[2]: PV-MCTS 100 sims/move, 1024 games, played to the
end: leaf-evals/s and games/s, with its own 1-core CPU baseline).
"""
import argparse
import ctypes
import json
import os
import sys
import time

# The independent 4096-board steps overlap on the chip through HIP streams; the
# runtime maps streams onto 4 hardware queues unless told otherwise, which caps
# the overlap at ~3.4 launches.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BOARDS_PER_GPU = 4096
START_OWN = 0x0000000810000000  # colour 1 (moves first): (3,4), (4,3)
START_OPP = 0x0000001008000000  # colour 2: (3,3), (4,4)
BYTES_PER_BOARD_STEP = 33       # load+store 2 x u64, + 1 B action
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s


def shipped_rollout_weights():
    with open(os.path.join(ROOT, "tests", "golden", "simulate.json")) as f:
        g = json.load(f)
    return np.asarray(g["shipped_w"], np.float32), np.asarray(g["shipped_b"], np.float32)


def host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(w, b, budget_s=10.0):
    """Oracle rollouts (same start position, same weights, same Philox keying)
    on the host cores; ctypes releases the GIL so plain threads scale.  The
    sample is sized from a short parallel probe so the leg takes ~budget_s."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    orc.build()
    cores = min(host_cores(), 64)
    s0 = orc.initial_state()

    def run(n_each, base):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            futs = [ex.submit(orc.simulate_batch, s0, 1, w, b, 0, base + i * n_each, n_each)
                    for i in range(cores)]
            steps = sum(f.result()[1] for f in futs)
        return time.perf_counter() - t0, steps

    t0 = time.perf_counter()
    orc.simulate_batch(s0, 1, w, b, 0, 1_000_000, 2000)
    one_core = 2000 / (time.perf_counter() - t0)
    probe_t, _ = run(100, 5_000_000)
    n_each = max(100, min(200_000, int(100 * budget_s / probe_t)))
    dt, steps = run(n_each, 10_000_000)
    games = n_each * cores
    return {"value": games / dt, "unit": "games/s", "cores": cores, "kind": "port",
            "sample": "%d rollout-policy games from the start position (%d per thread, "
                      "oracle/othello_oracle.c, %.1f s)" % (games, n_each, dt),
            "board_steps_per_game": steps / games, "one_core_games_per_sec": one_core}


def mcts_leg(n_games, n_sims, n_turns, full_games, world, rank, dist, value_f32=False, use_graph=True):
    """BASELINE configs[2]: PV-MCTS self-play, `n_games` lockstep games per GPU,
    `n_sims` playouts per move, SLPolicy + Value with random-init weights
    (Chainer-default LeCunNormal, seed 0), reference constants lmbda=0.5,
    c_puct=1, n_thr=15, both colours search.  One leaf-eval = one playout
    (value net + rollout at the leaf; the policy net runs on expansions).  The fixed
    tail of a playout and the next descent replay as one hipGraph launch.
    By default the games are played to the end (games/s); --mcts-turns N > 0
    times a bounded sample of the first N turns instead."""
    from iago_amd import engine, network, ops
    w, b = shipped_rollout_weights()
    torch.manual_seed(0)
    policy = network.SLPolicy().cuda().eval()
    value = network.Value().cuda().eval()
    value.split_f16 = not value_f32
    m = engine.BatchedMCTS(n_games, policy, value, ops.RolloutWeights(w, b), lmbda=0.5, c_puct=1.0,
                           n_thr=15, capacity=engine.suggest_capacity(n_sims, 15), seed=7,
                           game_id_base=rank * n_games, use_graph=use_graph)
    eng = engine.SelfPlayEngine(m, max_turns=(128 if full_games else n_turns))
    m.enable_stats()
    m.warmup()                 # MIOpen kernel selection for every batch bucket
    engine.SelfPlayEngine(m, max_turns=4).play(16, record=False)  # allocator, code objects
    m.n_leaf_evals = m.n_policy_evals = 0
    m.stats.zero_()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    res = eng.play(n_sims, record=True)
    gathered = None
    if dist is not None and full_games:
        from iago_amd.dist import gather_tuples
        gathered = gather_tuples(res.tuples())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    stats = torch.tensor([dt, m.n_leaf_evals, m.n_policy_evals], dtype=torch.float64, device="cuda")
    if dist is not None:
        tm = stats[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats)
        stats[0] = tm[0]
    dt, leaf, pol = (float(x) for x in stats.tolist())
    flops = leaf * 122_994_944 + pol * 122_847_232  # Value / SLPolicy per eval
    out = {"leaf_evals_per_sec": leaf / dt, "leaf_evals": int(leaf), "policy_evals": int(pol),
           "seconds": dt, "turns_played": res.n_turns, "sims_per_move": n_sims,
           "games_per_gpu": n_games, "full_games": bool(full_games),
           "net_tflops_fp32": flops / dt / 1e12,   # float32-equivalent FLOP/s of both nets
           "value_conv": ("f32 (MIOpen)" if value_f32 else
                          "split-f16 MFMA: f16 hi/lo operands, 3 MFMAs per product sum, f32 accumulation; "
                          "Value forward within 1e-6 of the f32 one (tests/test_conv_gpu.py)"),
           "policy_conv": "f32 (MIOpen)",
           "roofline": _mcts_roofline(leaf, pol, dt, world, value_f32),
           "config": "BASELINE configs[2]: PV-MCTS %d sims/move, %d games per GPU, SLPolicy+Value "
                     "random init fp32, lmbda=0.5 c_puct=1 n_thr=15" % (n_sims, n_games),
           "tree_pool_bytes_per_gpu": m.tree.bytes(), "tree_traffic_rank0": m.tree_bytes(),
           "tree_nodes_used_max": int(m.tree.n_nodes.max().item()),
           "tree_capacity": m.tree.capacity}
    if full_games:
        out["games_per_sec"] = world * n_games / dt
        if gathered is not None:
            out["gathered_tuples"] = int(gathered["z"].numel())
    return out


def _mcts_roofline(leaf, pol, dt, world, value_f32):
    """The convolutions bound this leg.  f32 path: float32 matrix/vector peak 157.3
    TFLOP/s.  Split-f16 path: the Value convolutions of blocks 2..8 (122.68 MFLOP per
    evaluation) execute 3 f16 MFMAs per product sum against the dense f16 peak of
    2,500 TFLOP/s (MI355X_MICROARCH.md)."""
    if value_f32:
        a = (leaf * 122_994_944 + pol * 122_847_232) / dt / 1e12 / world
        return {"bound": "mfma", "achieved": a, "peak": 157.3, "unit": "TFLOP/s", "frac": a / 157.3,
                "flops_per_leaf_eval": 122_994_944, "flops_per_policy_eval": 122_847_232}
    a = leaf * 3 * 122_683_392 / dt / 1e12 / world
    return {"bound": "mfma", "achieved": a, "peak": 2500.0, "unit": "TFLOP/s", "frac": a / 2500.0,
            "dtype": "f16 MFMA operands (split f32), f32 accumulate",
            "mfma_flops_per_leaf_eval": 3 * 122_683_392, "flops_per_leaf_eval": 122_994_944,
            "flops_per_policy_eval": 122_847_232}


def mcts_cpu_baseline(n_sims=600):
    """The reference's own algorithm for the PV-MCTS leg on ONE host core: the
    oracle's restatement of MCTS.playout (oracle/mcts_py.py) with float32
    torch-CPU SLPolicy / Value (B = 1 calls, one thread, like the reference's
    Chainer calls) and the C oracle's rollout; same constants as the GPU leg."""
    from iago_amd import network
    from oracle import mcts_py
    from oracle import oracle as orc
    w, b = shipped_rollout_weights()
    torch.manual_seed(0)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        policy, value = network.SLPolicy().eval(), network.Value().eval()
        counter = [0]

        def pol(x):
            with torch.no_grad():
                return policy(torch.from_numpy(x)).numpy().reshape(64)

        def val(x):
            with torch.no_grad():
                return value(torch.from_numpy(x)).numpy().reshape(1)[0]

        def roll(state, color):
            counter[0] += 1
            return orc.simulate(state, color, w, b, seed=3, game_id=counter[0])[0]

        m = mcts_py.MCTS(pol, val, roll, lmbda=0.5, c_puct=1, n_thr=15)
        m.get_move(orc.initial_state(), 1, 5)  # warm-up
        t0 = time.perf_counter()
        m.get_move(orc.initial_state(), 1, n_sims)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthreads)
    return {"value": n_sims / dt, "unit": "leaf-evals/s", "cores": 1, "kind": "port",
            "sample": "%d playouts of one game from the start position, oracle/mcts_py.py + "
                      "torch-CPU fp32 nets (1 thread) + C rollout, %.1f s" % (n_sims, dt)}


def mcts_b1_leg(n_sims=200):
    """Serving mode of game.py:112-113: ONE game, MCTS.get_move with a simulation
    budget (the reference spends 10 s per move at ~100 playouts/s on one core)."""
    from iago_amd import MCTS as mcts_mod
    from iago_amd import boards, network, ops
    w, b = shipped_rollout_weights()
    torch.manual_seed(0)
    m = mcts_mod.MCTS(policy_net=network.SLPolicy().cuda().eval(),
                      value_net=network.Value().cuda().eval(),
                      rollout_weights=ops.RolloutWeights(w, b), n_sims=n_sims, capacity=65536,
                      use_graph=True)
    state = boards.initial_state()
    m._m.warmup()
    m.get_move(state, 1)  # warm-up move (also fills the root)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = m.get_move(state, 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"playouts_per_sec": n_sims / dt, "ms_per_playout": dt / n_sims * 1e3,
            "sims": n_sims, "move": int(a), "hipgraph": True}


def reinforce_leg(n_iters, world, rank, dist):
    """BASELINE configs[4] in miniature: `n_iters` iterations of the REINFORCE loop
    (src/train_rl.py:28-81): one set of 2N = 64 SLPolicy-vs-SLPolicy games sharded
    over the ranks, all-gather of the (state, action, z) tuples, one update on
    every rank.  Random-init SLPolicy (seed 0), opponent = the current weights."""
    from iago_amd import network
    from iago_amd.train_rl import ReinforceTrainer
    torch.manual_seed(0)
    tr = ReinforceTrainer(network.SLPolicy(), pool_dir=None, N=32, seed=rank)
    for _ in range(2):
        tr.step()  # warm-up: MIOpen forward/backward kernel selection, allocator, weight-layout caches
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    tuples = 0
    for _ in range(n_iters):
        tuples += tr.step()["n_tuples"]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    return {"iters_per_sec": n_iters / dt, "games_per_sec": 64 * n_iters / dt,
            "tuples_per_iter": tuples / n_iters, "iters": n_iters,
            "config": "64 policy-vs-policy games per set (SLPolicy, random init, fp32) + "
                      "double-softmax REINFORCE update, ChainerAdam + WD 5e-4"}


VALU_PEAK_GINST = 256 * 4 * 2.4 / 2  # wave64 VALU instructions/ns: 1024 SIMD-32s, 2 cycles each


def measured_pmc():
    """Per-launch figures from the committed rocprofv3 --pmc passes (profiles/), if any."""
    path = os.path.join(ROOT, "profiles", "rollout_traffic.json")
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return {}


def measured_traffic(boards_per_launch):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes, if any, scaled from
    the profiled launch size to this run's (the traffic is per board: inputs, outputs and
    the table staged once per block)."""
    pmc = measured_pmc()
    if "hbm_bytes_per_launch" not in pmc:
        return None
    return pmc["hbm_bytes_per_launch"] * boards_per_launch / pmc.get("boards_per_launch", boards_per_launch)


def valu_utilisation(boards, seconds):
    """What really bounds the kernel: wave-level VALU instructions (PMC count per
    board from profiles/, the instruction stream does not depend on the batch) over
    the chip's issue peak.  None without a committed PMC profile."""
    pmc = measured_pmc()
    if "valu_insts_per_launch" not in pmc:
        return None
    per_board = pmc["valu_insts_per_launch"] / pmc["boards_per_launch"]
    achieved = per_board * boards / seconds / 1e9
    return {"valu_insts_per_board": per_board, "achieved_ginst_per_s": achieved,
            "peak_ginst_per_s": VALU_PEAK_GINST, "frac": achieved / VALU_PEAK_GINST}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--boards", type=int, default=BOARDS_PER_GPU)
    ap.add_argument("--steps-per-launch", type=int, default=2048,
                    help="consecutive steps played by one kernel launch (1 = one launch per "
                         "step, overlapped on --streams HIP streams)")
    ap.add_argument("--streams", type=int, default=32,
                    help="HIP streams the independent steps are issued on")
    ap.add_argument("--launch-streams", type=int, default=1,
                    help="HIP streams the multi-step launches alternate on (1: back to back, the "
                         "kernel duration is then a launch that has the chip to itself)")
    ap.add_argument("--one-launch", action="store_true",
                    help="N > 1: keep all steps in one launch (the tuple all-gather then follows it "
                         "instead of overlapping the later launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--large-boards", type=int, default=1 << 20,
                    help="extra occupancy datapoint: boards in one launch (0 = skip)")
    ap.add_argument("--train-iters", type=int, default=3,
                    help="REINFORCE iterations of the training leg (0 = skip)")
    ap.add_argument("--mcts-games", type=int, default=1024)
    ap.add_argument("--mcts-sims", type=int, default=100)
    ap.add_argument("--mcts-turns", type=int, default=-1,
                    help="PV-MCTS leg: -1 = play the games to the end (default), N > 0 = a bounded "
                         "sample of the first N turns, 0 = skip the leg")
    ap.add_argument("--mcts-eager", action="store_true",
                    help="PV-MCTS leg: plain launches instead of the hipGraph tail (rocprofv3 does not "
                         "attribute kernels launched from a graph)")
    ap.add_argument("--mcts-value-f32", action="store_true",
                    help="PV-MCTS leg: MIOpen float32 convolutions for the Value net instead of the "
                         "split-f16 MFMA kernels")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (any N)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from iago_amd import _lib, ops

    B, K, W = args.boards, args.steps, args.warmup
    w, b = shipped_rollout_weights()
    weights = ops.RolloutWeights(w, b)
    # G consecutive steps share one launch (G x B boards): with >= 32768 boards the
    # library plays them with the lane-per-board kernel, which needs ~40 % fewer
    # wave-instructions per board but 8x more boards in flight to fill the chip.
    # G = 1 issues every step as its own launch, overlapped on S HIP streams.
    G = max(1, min(args.steps_per_launch, K))
    if dist is not None and G >= K and K >= 512 and not args.one_launch:
        # N > 1: two launches, so that the all-gather of the first one's tuples (a side
        # stream) runs beside the second instead of after the only one (a launch of half the
        # boards costs ~3 % more per board; four launches cost more than they hide)
        G = (K + 1) // 2
    own = torch.full((G * B,), START_OWN, dtype=torch.int64, device="cuda")
    opp = torch.full((G * B,), START_OPP, dtype=torch.int64, device="cuda")
    # the round's finished tuples, resident in HBM: K steps x B games; the tuples of one
    # launch are ONE contiguous block [final own | final opp | z | turns] of the round
    # buffer, so that its all-gather needs no packing pass and no host sync
    n = K * B
    roundbuf = torch.empty(n * 18, dtype=torch.uint8, device="cuda")

    def block(k0, g):
        """Views (final_own, final_opp, z, n_turns, bytes) of the launch playing steps k0..k0+g-1."""
        m = g * B
        blk = roundbuf[k0 * B * 18:(k0 + g) * B * 18]
        return (blk[0:8 * m].view(torch.int64), blk[8 * m:16 * m].view(torch.int64),
                blk[16 * m:17 * m].view(torch.int8), blk[17 * m:18 * m], blk)

    use_gather = world > 1 or "RANK" in os.environ
    gathered = [torch.empty(world * min(G, K - k) * B * 18, dtype=torch.uint8, device="cuda")
                for k in range(0, K, G)] if use_gather else None

    S = max(1, args.streams) if G == 1 else max(1, args.launch_streams)
    streams = [torch.cuda.Stream() for _ in range(S)]
    sptr = [ctypes.c_void_p(st.cuda_stream) for st in streams]

    def prepare(k0, g, id_step):
        """One launch playing steps k0 .. k0+g-1 into their slots of the round buffer.
        Global game id = (rank * 2^20 + step) * B + board (rank-major, so that the ids
        of consecutive steps of a rank are contiguous)."""
        r = ops.RolloutResult()
        r.final_own, r.final_opp, r.z, r.n_turns, _ = block(k0, g)
        return ops.rollout_prepare(own[:g * B], opp[:g * B], weights, seed=2024,
                                   id_base=((rank * (1 << 20) + id_step) * B) & 0xFFFFFFFF, out=r)

    warm = [prepare(0, min(G, K), 500_000 + k) for k in range(0, W, G)]
    timed = [prepare(k, min(G, K - k), k) for k in range(0, K, G)]
    n_launches = len(timed)
    SAMPLE = 32 if G == 1 else 1  # launches bracketed by an event pair on their stream
    evs = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for i in range(0, n_launches, SAMPLE)}

    def barrier():
        if dist is not None:
            dist.barrier()

    rc = 0
    for i, p in enumerate(warm):
        rc |= p.launch(sptr[i % S])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    span0, span1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    span0.record(main)
    for st in streams:
        st.wait_stream(main)
    comm = torch.cuda.Stream() if dist is not None else None
    done = [torch.cuda.Event() for _ in range(n_launches)] if dist is not None else None
    for i, p in enumerate(timed):
        j = i % S
        e = evs.get(i)
        if e is not None:
            e[0].record(streams[j])
        rc |= p.launch(sptr[j])
        if e is not None:
            e[1].record(streams[j])
        if dist is not None:
            done[i].record(streams[j])
    for st in streams:
        main.wait_stream(st)
    span1.record(main)
    if dist is not None:
        # the tuples of every launch: one collective each on the side stream, ordered
        # behind its launch by an event, beside the launches that follow; no host sync
        # (enqueued after all launches so that the host never stands between two launches)
        with torch.cuda.stream(comm):
            for i in range(n_launches):
                comm.wait_event(done[i])
                dist.all_gather_into_tensor(gathered[i], block(i * G, min(G, K - i * G))[4])
    if comm is not None:
        main.wait_stream(comm)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rc != 0:
        raise SystemExit("iago_rollout failed: %s" % _lib.lib().iago_last_error())
    # average duration of ONE launch (start -> end on its own stream)
    full = [i for i in evs if timed[i].args.n == G * B] or list(evs)
    kernel_ms = sum(evs[i][0].elapsed_time(evs[i][1]) for i in full) / len(full)
    span_ms = span0.elapsed_time(span1)  # GPU time of the K launches together

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    steps_total = sum(block(k, min(G, K - k))[3].to(torch.int64).sum() for k in range(0, K, G)).reshape(1)
    launch_steps = G  # steps per (full) launch
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(steps_total)
        for i, k in enumerate(range(0, K, G)):
            mine = block(k, min(G, K - k))[4]
            assert torch.equal(gathered[i][rank * mine.numel():(rank + 1) * mine.numel()], mine)
    dt = float(tmax.item())
    board_steps = int(steps_total.item())

    # the same kernel with the chip full (not the headline config): occupancy evidence
    large = None
    if args.large_boards > 0 and rank == 0:
        LB = args.large_boards
        lown = torch.full((LB,), START_OWN, dtype=torch.int64, device="cuda")
        lopp = torch.full((LB,), START_OPP, dtype=torch.int64, device="cuda")
        lout = ops.RolloutResult()
        lout.z = torch.empty(LB, dtype=torch.int8, device="cuda")
        lout.n_turns = torch.empty(LB, dtype=torch.uint8, device="cuda")
        for k in range(3):
            ops.rollout(lown, lopp, weights, seed=1, id_base=0, stream_id=k, out=lout)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        LK = 10
        for k in range(LK):
            ops.rollout(lown, lopp, weights, seed=2, id_base=0, stream_id=k, out=lout)
        e1.record()
        torch.cuda.synchronize()
        lms = e0.elapsed_time(e1) / LK
        lsteps = int(lout.n_turns.to(torch.int64).sum().item())
        large = {"boards": LB, "kernel_ms": lms, "games_per_sec": LB / (lms * 1e-3),
                 "board_steps_per_sec": lsteps / (lms * 1e-3),
                 "hbm_frac": BYTES_PER_BOARD_STEP * lsteps / (lms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "valu": valu_utilisation(LB, lms * 1e-3)}
        del lown, lopp, lout

    mcts = None
    if args.mcts_turns != 0:
        mcts = mcts_leg(args.mcts_games, args.mcts_sims, max(args.mcts_turns, 0),
                        args.mcts_turns < 0, world, rank, dist, value_f32=args.mcts_value_f32,
                        use_graph=not args.mcts_eager)

    if mcts is not None and not args.mcts_value_f32 and args.mcts_turns < 0:
        # the same leg with MIOpen float32 convolutions for the Value net, on a bounded
        # sample (first 4 turns), for comparison with the split-f16 kernels
        ref = mcts_leg(args.mcts_games, args.mcts_sims, 4, False, world, rank, dist, value_f32=True)
        mcts["value_f32_sample"] = {k: ref[k] for k in ("leaf_evals_per_sec", "leaf_evals", "seconds",
                                                        "turns_played", "value_conv")}

    train = reinforce_leg(args.train_iters, world, rank, dist) if args.train_iters > 0 else None
    b1 = mcts_b1_leg() if (mcts is not None and rank == 0) else None

    if rank == 0:
        games = world * K * B
        alg_bytes_per_launch = BYTES_PER_BOARD_STEP * board_steps / (world * K) * launch_steps
        # Roofline of the dominant kernel.  One launch moves `alg_bytes_per_launch`
        # (algorithmic) in `kernel_ms`: an event pair brackets the launch's slot on its
        # stream = execution + queueing behind the other streams' launches (rocprofv3's
        # average, execution only, is ~25-30 % shorter under this overlap and equal with
        # --streams 1; the committed figure is `rocprof_kernel_avg_ms`).
        # `launches_in_flight` slots overlap, so the chip moves
        # achieved = bytes per launch / launch duration x launches in flight
        #          = bytes of the K launches / GPU span of the timed region,
        # which is also value x bytes per game.
        per_launch = alg_bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        achieved = alg_bytes_per_launch * (K / launch_steps) / (span_ms * 1e-3) / 1e9
        line = {
            "metric": "self-play games/sec", "value": games / dt, "unit": "games/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 bitboards + f32 policy", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d parallel Othello boards per GPU, "
                                   "rollout-policy-only playouts from the start position, "
                                   "shipped RolloutPolicy weights" % B,
                       "boards_per_gpu": B, "games_per_step": world * B,
                       "tuple_allgather": "rccl" if dist is not None else "none",
                       "steps_per_launch": G, "hip_streams": S},
            "board_steps_per_sec": board_steps / dt,
            "board_steps_per_game": board_steps / games,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(launch_steps * B),
                         "kernel": "rollout_lpb_kernel" if G * B >= 32768 else "rollout_kernel",
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "launches_in_flight": kernel_ms * (K / launch_steps) / span_ms,
                         "boards_per_launch": launch_steps * B,
                         "per_launch_achieved": per_launch,
                         "rocprof_kernel_avg_ms": measured_pmc().get("rocprof_kernel_avg_ms"),
                         "valu": valu_utilisation(K * B, span_ms * 1e-3)},
        }
        if large is not None:
            line["large_batch"] = large
        if mcts is not None:
            line["mcts"] = mcts
        if train is not None:
            line["reinforce"] = train
        if b1 is not None:
            line["mcts_single_game"] = b1
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is an N = 1 figure
            line["cpu_baseline"] = cpu_baseline(w, b)
            if mcts is not None:
                mcts["cpu_baseline"] = mcts_cpu_baseline()
        print(json.dumps(line), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
