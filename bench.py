#!/usr/bin/env python3
"""Benchmark of the MI355X self-play hot path (contract: see DESIGN.md, "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): per GPU, 4096 parallel Othello boards from
the standard start position are played to the end by the fused HIP rollout
kernel (16 lanes per board) with the reference's shipped RolloutPolicy weights (82 floats, kept as
golden data in tests/golden/simulate.json) -- rollout-policy-only self-play.
One step = ONE launch of 4096 boards = 4096 finished games per GPU; the launches
of a run are serialized on one stream, so 4096 boards are in flight at any time,
as the config says.  A round is K = --steps steps; the timed region repeats the
round R times so that it lasts >= 150 ms whatever K is (R is printed as
`repeats`; ms_per_step = region / (K R)), which makes `value` independent of
--steps.  With N > 1 every rank plays its own boards (weak scaling, Philox
streams keyed by a global game id (launch * N + rank) * 4096 + board) and the
finished (final boards, z, turns) tuples of every round are all-gathered over
RCCL inside the timed region, on a side stream beside the next round.

Rank 0 prints ONE JSON line.  `roofline` prices the rollout kernel against the
HBM roof with SURVEY.md section 8(d)'s algorithmic bytes (33 B per board-step):
bytes of one launch / the launch's duration, measured with HIP event pairs on the
launch stream around a sample of the timed launches (single tenant: equals the
rocprofv3 average of tools/profile_rollout.sh, committed under profiles/).
`cpu_baseline` times the CPU oracle (oracle/, a C port of the reference's Python
loops) on the host cores over a bounded sample of the same workload, with the
Python-loop restatement on one core beside it.
Extra objects, none of them `value`: `overlapped` (the same 4096-board launches
on 32 HIP streams), `large_batch` (one launch of 1M boards, lane-per-board
kernel), `mcts` (BASELINE configs[2]: PV-MCTS 100 sims/move, 1024 games, played
to the end: leaf-evals/s = playouts/s and games/s, with its own 1-core CPU
baseline; `value_evals` / `policy_evals` = rows the nets really processed -- the
value of a leaf is computed at its first visit only and the policy runs a few
visits ahead of the expansion), `reinforce`,
`mcts_single_game`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# (GPU_MAX_HW_QUEUES is left at the runtime's default of 4 hardware queues.  Round 1 raised it to
# 16 for the `overlapped` extra; with the 16-lane kernel 4 queues give MORE overlap (238 vs 219 M
# games/s), and with 8 or more queues the PV-MCTS graph -- two streams -- replays 1.5x slower once
# RCCL has created its own streams (tools/exp_nccl_bench_leg.py: 4.4 vs 6.6 M leaf-evals/s).)

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


def mcts_leg(n_games, n_sims, n_turns, full_games, world, rank, dist, value_f32=False, use_graph=True, n_thr=15,
             persistent=None):
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
    # the engine: the persistent search (one launch per whole game, every game on its own clock) wherever it
    # applies -- the split-f16 value net -- else per-playout launches replayed as hipGraphs (persistent=False: the
    # comparison figure `mcts_per_playout_launches`)
    if persistent is None:
        persistent = not value_f32 and use_graph
    m = engine.BatchedMCTS(n_games, policy, value, ops.RolloutWeights(w, b), lmbda=0.5, c_puct=1.0,
                           n_thr=n_thr, seed=7, game_id_base=rank * n_games, use_graph=use_graph and not persistent,
                           persistent=persistent,
                           capacity=engine.suggest_capacity(n_sims, n_thr, moves=64 if full_games else n_turns + 4))
    eng = engine.SelfPlayEngine(m, max_turns=(128 if full_games else n_turns))
    m.enable_stats()
    m.warmup()                 # MIOpen kernel selection for every batch bucket
    warm = engine.SelfPlayEngine(m, max_turns=4).play(16, record=True)  # allocator, code objects
    if dist is not None and full_games:
        from iago_amd.dist import gather_tuples
        gather_tuples(warm.tuples())   # (RCCL sets up a collective of a new size class on its first use)
    m.n_leaf_evals = m.n_policy_evals = 0
    m._value_total.zero_()
    if getattr(m, "persistent", False):
        m._ps["totals"].zero_()
    m.stats.zero_()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    res = eng.play(n_sims, record=True)
    gathered = None
    t_play = None
    if dist is not None and full_games:
        from iago_amd.dist import gather_tuples
        torch.cuda.synchronize()
        t_play = time.perf_counter() - t0
        gathered = gather_tuples(res.tuples())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    stats = torch.tensor([dt, m.n_leaf_evals, m.n_policy_evals, m.n_value_evals], dtype=torch.float64,
                         device="cuda")
    if dist is not None:
        tm = stats[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats)
        stats[0] = tm[0]
    dt, leaf, pol, val = (float(x) for x in stats.tolist())
    # Value / SLPolicy FLOPs per evaluation, for the evaluations EXECUTED (with the
    # value cache the net runs on a leaf's first visit only; the policy look-ahead also evaluates
    # leaves that never expand)
    flops = val * 122_994_944 + pol * 122_847_232
    out = {"leaf_evals_per_sec": leaf / dt, "leaf_evals": int(leaf), "policy_evals": int(pol),
           "value_evals": int(val), "value_inline": int(m.n_value_inline), "value_ahead": int(m.n_value_ahead),
           "async_steps": int(m.n_steps) if m.async_steps else None,
           "persistent": ({"net_workgroups": m.net_workgroups, "totals": [int(x) for x in m._ps["totals"].tolist()]}
                          if getattr(m, "persistent", False) else None),
           "leaf_eval_definition": "one leaf-eval = one playout (MCTS.py:105-133) ending in the leaf "
                                   "evaluation of MCTS.py:123-127: value_func(leaf) + rollout + backup; "
                                   "value_func is a pure function of the position, computed at a leaf's "
                                   "first visit and taken from the node afterwards (value_evals = net "
                                   "launches' rows), the rollout runs at every visit; trees bit-identical "
                                   "to evaluating the net at every visit (tests/test_mcts_gpu.py)",
           "engine": ("persistent search: ONE launch per batch of whole self-play games, game workgroups (%d games each: descent, "
                      "rollout, backup, moves; leading games paced) + %d net workgroups serving two rings of positions with "
                      "one-board / two-board walks of the value net and one-board walks of the policy net (at the expansion, "
                      "as the reference; while net workgroups idle, the values of an expanding node's children ahead of "
                      "their first visit: value_ahead)" % (m.games_per_workgroup, m.net_workgroups) if m.persistent else
                      "per-playout launches (descent, leaf evaluation, backup) replayed as hipGraphs, policy look-ahead "
                      "batches on a second stream"),
           "value_cache": bool(m.value_cache), "policy_lookahead": int(m.lookahead),
           "seconds": dt, "turns_played": res.n_turns, "sims_per_move": n_sims,
           "games_per_gpu": n_games, "full_games": bool(full_games),
           "net_tflops_fp32": flops / dt / 1e12,   # float32-equivalent FLOP/s of both nets, executed evaluations
           "value_conv": ("f32 (MIOpen)" if value_f32 else
                          "split-f16 MFMA: f16 hi/lo operands, 3 MFMAs per product sum, f32 accumulation; "
                          "Value forward within 1e-6 of the f32 one (tests/test_conv_gpu.py)"),
           "policy_conv": ("three-piece split on the f16 matrix units, the whole net in one launch "
                           "(iago_policy_forward_split3: 6 MFMAs per product sum, float32-exact products; within "
                           "1e-5 of the float64 outputs on the shipped net, tests/test_nets_shipped.py)"
                           if policy.split3 else
                           "f32: hand-written conv3x3_f32 / policy_head kernels (batches <= 192); MIOpen above"),
           "roofline": _mcts_roofline(val, pol, dt, world, value_f32, policy.split3, bool(m.persistent)),
           "config": "BASELINE configs[2]: PV-MCTS %d sims/move, %d games per GPU, SLPolicy+Value "
                     "random init fp32, lmbda=0.5 c_puct=1 n_thr=%d" % (n_sims, n_games, n_thr),
           "n_thr": n_thr,
           "tree_pool_bytes_per_gpu": m.tree.bytes(), "tree_traffic_rank0": m.tree_bytes(),
           "tree_nodes_used_max": int(m.tree.n_nodes.max().item()),
           "tree_capacity": m.tree.capacity}
    if full_games:
        out["games_per_sec"] = world * n_games / dt
        if gathered is not None:
            out["gathered_tuples"] = int(gathered["z"].numel())
            out["play_seconds_rank0"] = t_play   # the rest of `seconds`: packing + all-gather of the tuples + barrier
    out["device_memory_bytes"] = m.memory_bytes()
    m.close()   # the captured graphs go now, not whenever the garbage collector finds the engine
    return out


MFMA_FLOP_32x32x16 = 2 * 32 * 32 * 16    # one v_mfma_f32_32x32x16_f16 wave-instruction
MIN_PROFILE_LAUNCHES = 100
F16_PEAK_TF, F32_MATRIX_PEAK_TF = 2500.0, 157.3    # MI355X_MICROARCH.md: dense f16 MFMA / f32 matrix
VALUE_FLOP, POLICY_FLOP = 122_994_944, 122_847_232   # algorithmic FLOPs per evaluation
# wave-level MFMA instructions one evaluated board executes in the search's net kernels: blocks 2..8 =
# (36 + 6 x 72) k-steps x (2 tiles x 3 or 6 MFMAs) x 4 waves (+ the Value head's 8 x 2 x 3 on one wave)
VALUE_MFMA_PER_BOARD = 468 * 6 * 4 + 48
POLICY_MFMA_PER_BOARD = 468 * 12 * 4


def csrc_sha16():
    """Identity of the kernel sources (iago_amd/csrc/*, include/iago_hip.h): what a committed rocprofv3
    summary must carry (tools/summarize_mcts_profile.py writes it) to be quoted beside measured numbers."""
    import glob
    import hashlib
    h = hashlib.sha256()
    here = os.path.dirname(os.path.abspath(__file__))   # (the sources of THIS file's tree, wherever ROOT points)
    for path in sorted(glob.glob(os.path.join(here, "iago_amd", "csrc", "*"))) + [os.path.join(here, "include", "iago_hip.h")]:
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def net_kernel_profiles(persistent=False):
    """The two net kernels that ARE on the timed path of the PV-MCTS leg -- value_rollout_kernel
    (the leaf evaluation: one-board Value walks + the rollouts) and policy_resident_kernel (the
    look-ahead batches) -- from the newest committed FULL-GAME eager profile of the leg
    (tools/profile_mcts.sh -> profiles/*_mcts_fullgame_pmc_summary.json): rocprofv3's average
    duration and the executed f16 MFMA FLOP/s = SQ_INSTS_MFMA x 32,768 / duration against the
    2.5 PFLOP/s dense peak.  A kernel with fewer than 100 launches in the profile is refused
    (a one-launch sample of a variant off the path had been reported here).
    These are numbers of a COMMITTED profile, not of this run: the entry names the profile and says
    whether it was taken on the kernel sources of this tree (`current`: its csrc_sha16 equals
    csrc_sha16() now); a stale profile is still listed, marked, and its fractions are not lifted to
    the top-level line.  Per kernel, beside the executed f16 MFMA rate: the evaluated boards per
    launch (MFMA instructions / instructions per board), the USEFUL rate = boards x SURVEY 8(d)'s
    FLOPs per evaluation / duration, as fraction of the f16 peak and as multiple of the float32
    matrix peak, and the bytes a CU pulls from L2 per evaluation (TCP_TCC_READ_REQ x 128 B)."""
    import glob
    # (the persistent engine: ONE search_kernel launch per whole self-play game -- tools/profile_mcts.sh <tag>
    # persistent -> profiles/*_mcts_persistent_pmc_summary.json; a launch lasts ~0.5 s, so the rule is a total
    # profiled duration of >= 0.1 s instead of 100 launches)
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mcts_%s_pmc_summary.json"
                                          % ("persistent" if persistent else "fullgame"))))
    now = csrc_sha16()
    for path in reversed(paths):
        with open(path) as f:
            prof = json.load(f)
        ks = prof.get("kernels", {})
        out = {}
        for name in (("search_kernel",) if persistent else
                     ("value_rollout_kernel", "policy_resident_kernel", "descend_kernel", "mix_backup_path_kernel")):
            k = ks.get(name)
            if not k:
                continue
            if name == "search_kernel":
                if k.get("calls", 0) * k.get("avg_us", 0.0) < 1e5 or not k.get("pmc_launches"):
                    continue
            elif k.get("calls", 0) < MIN_PROFILE_LAUNCHES or k.get("pmc_launches", 0) < MIN_PROFILE_LAUNCHES:
                continue
            e = {"launches": k["calls"], "rocprof_avg_us": k["avg_us"],
                 "hbm_bytes_per_launch": k.get("hbm_bytes_per_launch")}
            if k.get("SQ_INSTS_MFMA"):
                tf = k["SQ_INSTS_MFMA"] * MFMA_FLOP_32x32x16 / (k["avg_us"] * 1e-6) / 1e12
                e.update({"mfma_insts_per_launch": k["SQ_INSTS_MFMA"], "executed_tflops": tf,
                          "bound": "mfma", "peak": F16_PEAK_TF, "frac": tf / F16_PEAK_TF})
                per_board, flop = {"value_rollout_kernel": (VALUE_MFMA_PER_BOARD, VALUE_FLOP),
                                   # (a launch of the two-launch forward walks half a net)
                                   "policy_resident_kernel": (POLICY_MFMA_PER_BOARD / 2, POLICY_FLOP / 2)}.get(name, (0, 0))
                if per_board:
                    boards = k["SQ_INSTS_MFMA"] / per_board
                    useful = boards * flop / (k["avg_us"] * 1e-6) / 1e12
                    e.update({"boards_per_launch": boards, "useful_tflops": useful,
                              "useful_frac_f16_peak": useful / F16_PEAK_TF,
                              "useful_x_f32_matrix_peak": useful / F32_MATRIX_PEAK_TF})
                    if k.get("TCP_TCC_READ_REQ_sum"):
                        per_eval = k["TCP_TCC_READ_REQ_sum"] * 128.0 / boards
                        e["l2_to_cu_bytes_per_eval"] = per_eval * (2 if name == "policy_resident_kernel" else 1)
            elif k.get("hbm_bytes_per_launch"):
                gb = k["hbm_bytes_per_launch"] / (k["avg_us"] * 1e-6) / 1e9
                e.update({"bound": "hbm", "achieved_gb_per_s": gb, "peak": HBM_PEAK_GBS, "frac": gb / HBM_PEAK_GBS})
            out[name] = e
        if ("search_kernel" in out) if persistent else ("value_rollout_kernel" in out and "policy_resident_kernel" in out):
            out["profile"] = os.path.basename(path)
            out["command"] = prof.get("command")
            out["profile_csrc_sha16"] = prof.get("csrc_sha16")
            out["current"] = prof.get("csrc_sha16") == now
            out["provenance"] = ("committed rocprofv3 profile of an earlier run of this command, %s"
                                 % ("taken on these kernel sources" if out["current"] else
                                    "STALE: taken on other kernel sources (csrc_sha16 %s, now %s)"
                                    % (prof.get("csrc_sha16"), now)))
            return out
    return None


def _mcts_roofline(leaf, pol, dt, world, value_f32, policy_split3=False, persistent=False):  # leaf = value-net evaluations executed
    """The convolutions bound this leg.  f32 path: float32 matrix/vector peak 157.3
    TFLOP/s.  Split-f16 path: the Value convolutions of blocks 2..8 (122.68 MFLOP per
    evaluation) execute 3 f16 MFMAs per product sum -- and the SLPolicy ones (same shape) 6 with
    the three-piece split -- against the dense f16 peak of 2,500 TFLOP/s (MI355X_MICROARCH.md)."""
    if value_f32:
        a = (leaf * 122_994_944 + pol * 122_847_232) / dt / 1e12 / world
        return {"bound": "mfma", "achieved": a, "peak": 157.3, "unit": "TFLOP/s", "frac": a / 157.3,
                "flops_per_leaf_eval": 122_994_944, "flops_per_policy_eval": 122_847_232}
    a = (leaf * 3 + (pol * 6 if policy_split3 else 0)) * 122_683_392 / dt / 1e12 / world
    useful = (leaf * VALUE_FLOP + pol * POLICY_FLOP) / dt / 1e12 / world
    return {"bound": "mfma", "achieved": a, "peak": 2500.0, "unit": "TFLOP/s", "frac": a / 2500.0,
            # the same loop in SURVEY 8(d)'s algorithmic FLOPs (one multiply-add per product, whatever the
            # number of f16 pieces it is executed in)
            "useful_tflops": useful, "useful_frac_f16_peak": useful / F16_PEAK_TF,
            "useful_x_f32_matrix_peak": useful / F32_MATRIX_PEAK_TF,
            "dtype": "f16 MFMA operands (split f32), f32 accumulate",
            "mfma_flops_per_leaf_eval": 3 * 122_683_392,
            "mfma_flops_per_policy_eval": (6 * 122_683_392 if policy_split3 else 0),
            "flops_per_leaf_eval": 122_994_944,
            "flops_per_policy_eval": 122_847_232,
            "note": "loop level: executed f16 MFMA FLOPs of both nets' trunks (evaluations actually run: "
                    "the value cache skips re-evaluations of a leaf, the policy look-ahead also evaluates "
                    "leaves that never expand) over the WHOLE leg's wall time (tree kernels and rollouts "
                    "included); the net kernels on the timed path, per launch: `kernels` (committed full-game "
                    "rocprofv3 profile, >= 100 launches each).  The one-board-per-workgroup "
                    "launches of the search (Value 3.9 MB, SLPolicy 5.8 MB of weights per board into ONE CU) "
                    "are bound by that CU's L2 bandwidth (~70 GB/s: 56 of 69 us, 83 of 113 us), not by the "
                    "matrix pipe (DESIGN.md section 5)",
            "kernels": net_kernel_profiles(persistent)}


def cpu_workers(kind, budget_s):
    """the reference's own execution model "on P = os.cpu_count() independent
    worker processes, P stated" -- the reference is one single-threaded Python process per game
    (MCTS.py:139-147, mcts_self_play.py:25-29), so P of them side by side is what the host's
    cores give it.  P = host_cores() copies of this script in --cpu-worker mode (CHILD processes
    that never touch the GPU), all timing the same `budget_s` window that starts at a common wall
    clock instant after their imports; the aggregate is the sum of their counts over the window."""
    import subprocess
    P = min(host_cores(), 64)
    start_at = time.time() + 20.0        # imports + warm-up of the slowest worker
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", kind,
                               "--cpu-worker-budget", str(budget_s), "--cpu-worker-start", repr(start_at),
                               "--cpu-worker-seed", str(i)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                              env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1"))
             for i in range(P)]
    rows = []
    for pr in procs:
        out, _ = pr.communicate(timeout=600)
        if pr.returncode == 0 and out.strip():
            rows.append(json.loads(out.strip().splitlines()[-1]))
    if len(rows) != P:
        return {"error": "%d of %d CPU workers finished" % (len(rows), P)}
    window = max(r["seconds"] for r in rows)
    late = max(r["late_s"] for r in rows)
    return {"count": sum(r["count"] for r in rows), "seconds": window, "processes": P, "max_start_lag_s": late,
            "steps": sum(r.get("steps", 0) for r in rows)}


def cpu_worker_main(kind, budget_s, start_at, seed):
    """One worker of cpu_workers(): warm up, wait for the common start, run for budget_s."""
    w, b = shipped_rollout_weights()
    torch.set_num_threads(1)
    count = steps = 0
    if kind == "python_loops":
        from iago_amd import network
        from oracle import py_loops
        ro = network.RolloutPolicy().eval()
        with torch.no_grad():
            ro.conv1.weight.copy_(torch.from_numpy(w.reshape(1, 2, 3, 3)))
            ro.bias2.b.copy_(torch.from_numpy(b))

        def policy(x):
            with torch.no_grad():
                return ro(torch.from_numpy(x)).numpy()

        s0 = np.zeros((8, 8), np.float32)
        s0[4, 3] = s0[3, 4] = 1
        s0[3, 3] = s0[4, 4] = 2
        rs = np.random.RandomState(seed)

        def unit():
            return 1, py_loops.simulate(s0, 1, policy, rs)[1]
    elif kind == "mcts":
        from iago_amd import network
        from oracle import mcts_py
        from oracle import oracle as orc
        torch.manual_seed(0)
        policy_net, value_net = network.SLPolicy().eval(), network.Value().eval()
        counter = [seed << 24]

        def pol(x):
            with torch.no_grad():
                return policy_net(torch.from_numpy(x)).numpy().reshape(64)

        def val(x):
            with torch.no_grad():
                return value_net(torch.from_numpy(x)).numpy().reshape(1)[0]

        def roll(state, color):
            counter[0] += 1
            return orc.simulate(state, color, w, b, seed=3, game_id=counter[0])[0]

        m = mcts_py.MCTS(pol, val, roll, lmbda=0.5, c_puct=1, n_thr=15)

        def unit():   # 20 playouts of the worker's own game tree from the start position
            m.get_move(orc.initial_state(), 1, 20)
            return 20, 0
    else:
        raise SystemExit("unknown --cpu-worker kind %r" % kind)
    unit()                                  # warm-up
    late = max(0.0, time.time() - start_at)
    while time.time() < start_at:
        time.sleep(0.01)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        c, st = unit()
        count += c
        steps += st
    print(json.dumps({"count": count, "steps": steps, "seconds": time.perf_counter() - t0, "late_s": late}), flush=True)


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
    out = {"value": n_sims / dt, "unit": "leaf-evals/s", "cores": 1, "kind": "port",
           "sample": "%d playouts of one game from the start position, oracle/mcts_py.py + "
                     "torch-CPU fp32 nets (1 thread) + C rollout, %.1f s" % (n_sims, dt)}
    pw = cpu_workers("mcts", 6.0)
    if "error" in pw:
        out["all_cores"] = pw
    else:
        out["all_cores"] = {"value": pw["count"] / pw["seconds"], "unit": "leaf-evals/s", "cores": pw["processes"],
                            "kind": "port",
                            "sample": "%d independent single-threaded worker processes (one game tree each, the "
                                      "same restatement), %d playouts in a common %.1f s window"
                                      % (pw["processes"], pw["count"], pw["seconds"])}
    return out


def mcts_b1_leg(n_sims=200):
    """Serving mode of game.py:112-113: ONE game, MCTS.get_move with a simulation
    budget (the reference spends 10 s per move at ~100 playouts/s on one core)."""
    from iago_amd import MCTS as mcts_mod
    from iago_amd import boards, network, ops
    w, b = shipped_rollout_weights()
    torch.manual_seed(0)
    m = mcts_mod.MCTS(policy_net=network.SLPolicy().cuda().eval(),
                      value_net=network.Value().cuda().eval(),
                      rollout_weights=ops.RolloutWeights(w, b), n_sims=n_sims, capacity=65536)
    # (the engine's default: the persistent search -- one launch per get_move; 25.3 k against 20.0 k playouts/s on
    # the per-playout launches replayed as hipGraphs, tools/_build/b1.py, round 4)
    state = boards.initial_state()
    m._m.warmup()
    m.get_move(state, 1)  # warm-up move (also fills the root)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = m.get_move(state, 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m._m.close()
    return {"playouts_per_sec": n_sims / dt, "ms_per_playout": dt / n_sims * 1e3,
            "sims": n_sims, "move": int(a), "engine": "persistent search" if m._m.persistent else "per-playout launches"}


def miopen_find_db_state():
    """MIOpen's user find-db decides which backward-convolution solvers the REINFORCE update runs on:
    on a fresh box (no tuned entries) the immediate-mode fallback solvers take ~40 ms per update, once
    any process has run a find (torch.backends.cudnn.benchmark = True: ~110 s of tuning, not done here)
    ~10 ms (LABNOTES.md, round 3).  'warm' = the user db holds find records for this GPU."""
    import glob
    home = os.environ.get("MIOPEN_USER_DB_PATH") or os.path.join(os.path.expanduser("~"), ".config", "miopen")
    files = [f for f in glob.glob(os.path.join(home, "**", "*.ufdb.txt"), recursive=True) if os.path.getsize(f) > 0]
    return "warm" if files else "cold"


def reinforce_leg(n_iters, world, rank, dist, mcts_rounds=1):
    """BASELINE configs[4] in miniature: `n_iters` iterations of the REINFORCE loop
    (src/train_rl.py:28-81): one set of 2N = 64 SLPolicy-vs-SLPolicy games sharded
    over the ranks, all-gather of the (state, action, z) tuples, one update on
    every rank.  Random-init SLPolicy (seed 0), opponent = the current weights.
    Then, as configs[4] words it ("self-play feeding train_rl.py REINFORCE update on gathered
    (s, pi, z)"), `mcts_rounds` rounds of PV-MCTS self-play (64 games sharded over the ranks, 20
    playouts per move, the learner as the search's policy net) -> SelfPlayResult.tuples() ->
    ReinforceTrainer.step_from_tuples (src/train_rl.py:55-66 on the gathered rows)."""
    from iago_amd import engine, network, ops
    from iago_amd.dist import shard_range
    from iago_amd.train_rl import ReinforceTrainer
    torch.manual_seed(0)
    db_before = miopen_find_db_state()
    tr = ReinforceTrainer(network.SLPolicy(), pool_dir=None, N=32, seed=rank)
    for _ in range(4):
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
    # what decides the figure is which backward-convolution solvers MIOpen picked for the update (tuned ones from a
    # find-db that holds THIS problem: ~10 ms; its immediate-mode fallback: ~40 ms): measured, not guessed -- one more
    # set's update timed on its own after the timed region (the user db's mere presence says little: a db warmed by
    # other problems still falls back here)
    tup, _ = tr.play_set(tr.pick_opponent())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tr._update(tup["own"], tup["opp"], tup["action"], tup["z"])
    torch.cuda.synchronize()
    update_ms = (time.perf_counter() - t1) * 1e3
    out = {"iters_per_sec": n_iters / dt, "games_per_sec": 64 * n_iters / dt,
           "tuples_per_iter": tuples / n_iters, "iters": n_iters, "ms_per_iter": dt / n_iters * 1e3,
           # the figure depends on this state (19 sets/s with the fallback solvers, 45 with tuned ones on one MI355X):
           # stated, not hidden
           "update_ms": update_ms, "miopen_solvers": "tuned (find-db)" if update_ms < 20.0 else "immediate-mode fallback",
           "miopen_find_db": db_before,
           "config": "64 policy-vs-policy games per set (SLPolicy, random init, fp32) + "
                     "double-softmax REINFORCE update, ChainerAdam + WD 5e-4"}
    if mcts_rounds > 0:
        w, b = shipped_rollout_weights()
        games, sims = 64, 20
        lo, hi = shard_range(games, rank, world)
        value = network.Value().cuda().eval()
        m = engine.BatchedMCTS(hi - lo, tr.model1, value, ops.RolloutWeights(w, b), n_thr=15,
                               capacity=engine.suggest_capacity(sims, 15), seed=1, game_id_base=lo, persistent=True)
        sp = engine.SelfPlayEngine(m)

        def one():
            tr.model1.eval()
            res = sp.play(sims)     # (one launch per round: nothing to re-capture when the weights change)
            return tr.step_from_tuples(res.tuples())

        one()                       # warm-up round
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        leaf0, n_tup, loss = m.n_leaf_evals, 0, None
        for _ in range(mcts_rounds):
            r = one()
            n_tup += r["n_tuples"]
            loss = r["loss"]
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt2 = time.perf_counter() - t0
        out["mcts_fed"] = {"rounds": mcts_rounds, "rounds_per_sec": mcts_rounds / dt2, "seconds": dt2,
                           "tuples_per_round": n_tup / mcts_rounds, "loss": loss,
                           "leaf_evals_rank0": m.n_leaf_evals - leaf0,
                           "config": "%d PV-MCTS self-play games per round (sharded over the ranks), %d playouts per "
                                     "move, learner = the search's policy net -> tuples (own, opp, move, z) of both "
                                     "colours -> step_from_tuples (gather, canonical order, REINFORCE update)"
                                     % (games, sims)}
        m.close()
    return out


VALU_PEAK_GINST = 256 * 4 * 2.4 / 2  # wave64 VALU instructions/ns: 1024 SIMD-32s, 2 cycles each
TARGET_REGION_S = 0.15               # the timed region is at least this long whatever --steps is


def measured_pmc(boards_per_launch):
    """Per-launch figures of the rollout kernel from the committed rocprofv3 passes
    (profiles/rollout_traffic.json, keyed by the launch size they were taken at)."""
    path = os.path.join(ROOT, "profiles", "rollout_traffic.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f).get("by_boards_per_launch", {}).get(str(boards_per_launch), {})


def valu_utilisation(boards_per_launch, launches, seconds):
    """What really bounds the rollout kernels: wave-level VALU instructions (PMC count of a
    launch of this size, from profiles/) over the chip's issue peak."""
    pmc = measured_pmc(boards_per_launch)
    if "valu_insts_per_launch" not in pmc:
        return None
    achieved = pmc["valu_insts_per_launch"] * launches / seconds / 1e9
    return {"valu_insts_per_board": pmc["valu_insts_per_launch"] / boards_per_launch,
            "achieved_ginst_per_s": achieved, "peak_ginst_per_s": VALU_PEAK_GINST,
            "frac": achieved / VALU_PEAK_GINST, "profile": pmc.get("profile")}


def python_loop_baseline(w, b, budget_s=4.0):
    """The reference's execution model on ONE core: interpreted Python loops over an (8,8)
    numpy board (oracle/py_loops.py restates rl_env.py:88-138 / mcts_self_play.py:25-134)
    with a B = 1 torch-CPU RolloutPolicy call per move."""
    from iago_amd import network
    from oracle import py_loops
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        ro = network.RolloutPolicy().eval()
        with torch.no_grad():
            ro.conv1.weight.copy_(torch.from_numpy(w.reshape(1, 2, 3, 3)))
            ro.bias2.b.copy_(torch.from_numpy(b))

        def policy(x):
            with torch.no_grad():
                return ro(torch.from_numpy(x)).numpy()

        s0 = np.zeros((8, 8), np.float32)
        s0[4, 3] = s0[3, 4] = 1
        s0[3, 3] = s0[4, 4] = 2
        rs = np.random.RandomState(0)
        games = steps = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            steps += py_loops.simulate(s0, 1, policy, rs)[1]
            games += 1
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthreads)
    out = {"value": games / dt, "unit": "games/s", "cores": 1, "kind": "port",
           "sample": "%d rollout-policy games, oracle/py_loops.py (Python loops over a numpy board, "
                     "torch-CPU B=1 policy calls), %.1f s" % (games, dt),
           "board_steps_per_game": steps / max(games, 1)}
    pw = cpu_workers("python_loops", 4.0)
    if "error" in pw:
        out["all_cores"] = pw
    else:
        out["all_cores"] = {"value": pw["count"] / pw["seconds"], "unit": "games/s", "cores": pw["processes"],
                            "kind": "port",
                            "sample": "%d independent single-threaded worker processes, %d games in a common "
                                      "%.1f s window" % (pw["processes"], pw["count"], pw["seconds"])}
    return out


class RolloutRounds(object):
    """BASELINE configs[1] on one rank: one step = ONE launch of B = 4096 boards from the
    start position played to the end.  Results land in an exchange buffer of K steps resident
    in HBM: per step one contiguous block [final own | final opp | z | turns] (18 B per game);
    two buffers alternate so that the all-gather of one (N > 1, side stream) runs beside the
    launches that fill the other."""

    def __init__(self, B, K, world, rank, weights, ops):
        self.B, self.K, self.world, self.rank, self.ops = B, K, world, rank, ops
        self.own = torch.full((B,), START_OWN, dtype=torch.int64, device="cuda")
        self.opp = torch.full((B,), START_OPP, dtype=torch.int64, device="cuda")
        self.bufs = [torch.zeros(K * B * 18, dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.launches = [[self._prepare(buf, k, weights) for k in range(K)] for buf in range(2)]

    def views(self, buf, k):
        B = self.B
        blk = self.bufs[buf][k * B * 18:(k + 1) * B * 18]
        return (blk[0:8 * B].view(torch.int64), blk[8 * B:16 * B].view(torch.int64),
                blk[16 * B:17 * B].view(torch.int8), blk[17 * B:18 * B])

    def _prepare(self, buf, k, weights):
        r = self.ops.RolloutResult()
        r.final_own, r.final_opp, r.z, r.n_turns = self.views(buf, k)
        return self.ops.rollout_prepare(self.own, self.opp, weights, seed=2024, id_base=0, out=r)

    def id_base(self, launch_index):
        """Global game id of board 0 of a launch: (launch * world + rank) * B -- unique over
        ranks and launches (the Philox counter word is 32 bits wide: checked by the caller)."""
        return (launch_index * self.world + self.rank) * self.B

    def launch(self, buf, k, launch_index, stream_ptr, stream_id=0):
        p = self.launches[buf][k]
        p.args.id_base = self.id_base(launch_index) & 0xFFFFFFFF
        p.args.stream_id = stream_id
        return p.launch(stream_ptr)

    def board_steps(self, buf):
        return sum(int(self.views(buf, k)[3].to(torch.int64).sum().item()) for k in range(self.K))


def rollout_leg(args, world, rank, dist):
    """The headline leg; returns the dict of measurements rank 0 prints."""
    from iago_amd import _lib, ops
    from iago_amd.dist import all_gather_into   # (= dist.all_gather_into_tensor under nccl)
    B, K, W = args.boards, args.steps, args.warmup
    w, b = shipped_rollout_weights()
    weights = ops.RolloutWeights(w, b)
    main = torch.cuda.current_stream()
    mptr = ctypes.c_void_p(main.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    rc = 0
    cal = RolloutRounds(B, min(K, 32), world, rank, weights, ops)
    for i in range(W):                      # untimed warm-up steps (their own Philox stream)
        rc |= cal.launch(i & 1, i % cal.K, i, mptr, stream_id=1)
    torch.cuda.synchronize()
    # calibration: the duration of a step decides how many rounds make a >= 150 ms region
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ncal = 32
    c0.record(main)
    for i in range(ncal):
        rc |= cal.launch(0, i % cal.K, i, mptr, stream_id=1)
    c1.record(main)
    torch.cuda.synchronize()
    t_step = torch.tensor([c0.elapsed_time(c1) * 1e-3 / ncal], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t_step, op=dist.ReduceOp.MAX)
    R = args.repeats if args.repeats > 0 else max(1, int(np.ceil(TARGET_REGION_S / (K * float(t_step.item())))))
    n_launches = R * K
    if (n_launches * world + world) * B >= 1 << 32:
        raise SystemExit("game ids exceed the 32-bit Philox counter word: lower --steps / --repeats")

    # N > 1: the finished tuples are all-gathered in exchanges of S steps = whole rounds worth
    # ~24 MB per rank (one RCCL collective per exchange on a side stream, beside the launches
    # that fill the other buffer); the last exchange may be partial
    use_gather = dist is not None
    G = max(1, min(R, int(round(24e6 / (K * B * 18))))) if use_gather else 1
    S = G * K
    rr = RolloutRounds(B, S, world, rank, weights, ops)
    del cal
    n_exch = (n_launches + S - 1) // S
    gathered = [torch.empty(world * S * B * 18, dtype=torch.uint8, device="cuda") for _ in range(2)] \
        if use_gather else None
    comm = torch.cuda.Stream() if use_gather else None
    played = [torch.cuda.Event() for _ in range(n_exch)] if use_gather else None
    shipped = [torch.cuda.Event() for _ in range(n_exch)] if use_gather else None
    # kernel duration: event pairs around a sample of the launches, on the launch stream
    every = max(1, n_launches // 64)
    evs = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for i in range(0, n_launches, every)}
    span0, span1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    span0.record(main)
    for i in range(n_launches):
        x, slot = divmod(i, S)
        buf = x & 1
        if use_gather and slot == 0 and x >= 2:
            main.wait_event(shipped[x - 2])   # this buffer's previous exchange has left
        e = evs.get(i)
        if e is not None:
            e[0].record(main)
        rc |= rr.launch(buf, slot, i, mptr)
        if e is not None:
            e[1].record(main)
        if use_gather and (slot == S - 1 or i == n_launches - 1):
            # ONE collective on the side stream, ordered behind the exchange's last launch
            # by an event, beside the launches that follow; no host sync
            nfill = (slot + 1) * B * 18
            played[x].record(main)
            with torch.cuda.stream(comm):
                comm.wait_event(played[x])
                all_gather_into(gathered[buf][:world * nfill], rr.bufs[buf][:nfill])
                shipped[x].record(comm)
    span1.record(main)
    if use_gather:
        main.wait_stream(comm)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rc != 0:
        raise SystemExit("iago_rollout failed: %s" % _lib.lib().iago_last_error())

    # Launch duration.  The launches are back to back on one stream, so the period
    # span / launches bounds the kernel's duration from above (kernel + dispatch gap); an
    # event pair around ONE launch adds the cost of the two event packets to it (~3 us).
    # The roofline uses the smaller of the two -- rocprofv3's average (profiles/) is the
    # kernel alone and must not exceed it.
    pair_ms = sum(e[0].elapsed_time(e[1]) for e in evs.values()) / len(evs)
    span_ms = span0.elapsed_time(span1)
    kernel_ms = min(pair_ms, span_ms / n_launches)
    x_last, slot_last = divmod(n_launches - 1, S)
    last, nlast = x_last & 1, slot_last + 1          # the buffer / steps of the last exchange
    steps_last = sum(int(rr.views(last, k)[3].to(torch.int64).sum().item()) for k in range(nlast))
    steps_round = torch.tensor([steps_last], dtype=torch.float64, device="cuda")
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_gather:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(steps_round)
        nb = nlast * B * 18
        g = gathered[last]
        assert torch.equal(g[rank * nb:(rank + 1) * nb], rr.bufs[last][:nb])
        for other in range(world):            # every rank played games of its own
            if other != rank:
                assert not torch.equal(g[other * nb:(other + 1) * nb], rr.bufs[last][:nb]), \
                    "ranks %d and %d played identical games" % (rank, other)
    dt = float(tmax.item())
    steps_per_game = float(steps_round.item()) / (world * nlast * B)
    games = world * R * K * B
    alg = BYTES_PER_BOARD_STEP * steps_per_game * B          # algorithmic bytes of ONE launch
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    pmc = measured_pmc(B)
    out = {
        "exchange": "%s all-gather of the finished tuples on a side stream, one per %d steps "
                    "(%.1f MB per rank)" % ("rccl" if dist is None or dist.get_backend() == "nccl" else dist.get_backend(),
                                            S, S * B * 18 / 1e6),
        "value": games / dt, "ms_per_step": dt / (R * K) * 1e3, "repeats": R,
        "timed_region_s": dt, "board_steps_per_game": steps_per_game,
        "board_steps_per_sec": steps_per_game * games / dt,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get("hbm_bytes_per_launch"),
                     "kernel": "rollout_row_kernel<false> (16 lanes per board)", "kernel_ms": kernel_ms,
                     "kernel_ms_event_pairs": pair_ms, "kernel_ms_period": span_ms / n_launches,
                     "kernel_ms_samples": len(evs), "boards_per_launch": B,
                     "algorithmic_bytes_per_launch": alg,
                     "rocprof_kernel_avg_ms": pmc.get("rocprof_kernel_avg_ms"),
                     "profile": pmc.get("profile"),
                     "valu": valu_utilisation(B, 1, kernel_ms * 1e-3),
                     "note": "the HBM roof is nominal for this path: a board (16 B) lives in VGPRs "
                             "for the whole game, measured traffic is far below the algorithmic "
                             "bytes; the real bound of one 4096-board launch (1024 waves = one per "
                             "SIMD) is the length of a wave's instruction stream: a lone wave issues "
                             "one instruction per 4 cycles (DESIGN.md section 5)"},
    }

    # ---- extra datapoints (rank 0, not the headline): the same steps overlapped on HIP
    # streams, and one launch large enough to fill the chip
    if rank == 0 and not args.rollout_only and not args.mcts_only:
        NS = max(1, args.streams)
        streams = [torch.cuda.Stream() for _ in range(NS)]
        sptr = [ctypes.c_void_p(st.cuda_stream) for st in streams]
        no = max(256, min(2048, K))
        for rep in range(2):                 # first pass warms the streams up
            torch.cuda.synchronize()
            o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            o0.record(main)
            for st in streams:
                st.wait_stream(main)
            for i in range(no):
                rc |= rr.launch(0, i % rr.K, i, sptr[i % NS], stream_id=2)
            for st in streams:
                main.wait_stream(st)
            o1.record(main)
            torch.cuda.synchronize()
        oms = o0.elapsed_time(o1)
        out["overlapped"] = {"games_per_sec": no * B / (oms * 1e-3), "launches": no, "hip_streams": NS,
                             "boards_per_launch": B, "ms_per_step": oms / no,
                             "hbm_frac": alg * no / (oms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "the same 4096-board launches issued on %d HIP streams "
                                     "(independent steps overlap on the chip); not `value`" % NS}
        if args.large_boards > 0:
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
            lp = measured_pmc(LB)
            out["large_batch"] = {"boards_in_flight": LB, "kernel": "rollout_lpb_kernel (lane per board)",
                                  "kernel_ms": lms, "games_per_sec": LB / (lms * 1e-3),
                                  "board_steps_per_sec": lsteps / (lms * 1e-3),
                                  "hbm_frac": BYTES_PER_BOARD_STEP * lsteps / (lms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": lp.get("hbm_bytes_per_launch"),
                                  "valu": valu_utilisation(LB, 1, lms * 1e-3),
                                  "note": "one launch of %d boards: a different configuration from "
                                          "BASELINE configs[1] (4096 boards in flight); not `value`" % LB}
            del lown, lopp, lout
    return out, (w, b)


def spawn_ranks(n):
    """`python bench.py --gpus N` started without torch.distributed.run: N copies of this
    script as CHILD processes, one per GPU (RANK = LOCAL_RANK = 0..N-1, rendezvous on
    127.0.0.1 at a free port), started before this process makes any HIP call -- a process
    that has initialised the GPU must never exec or fork another GPU program on this pool.
    Rank 0 inherits stdout and prints the ONE JSON line; the other ranks' stdout goes to
    stderr.  Returns the exit status: 0 when every rank returned 0, else the first failure
    (the remaining ranks are then terminated by PID)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this host driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    status = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 128 - rc
                for q in live:          # a rank died: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--boards", type=int, default=BOARDS_PER_GPU)
    ap.add_argument("--repeats", type=int, default=0,
                    help="rounds of --steps steps in the timed region (0 = as many as make it >= 150 ms)")
    ap.add_argument("--streams", type=int, default=32,
                    help="HIP streams of the `overlapped` datapoint")
    ap.add_argument("--rollout-only", action="store_true",
                    help="only the headline leg (what tools/profile_rollout.sh profiles)")
    ap.add_argument("--mcts-only", action="store_true",
                    help="skip the extra datapoints of the rollout leg and the single-game leg (what "
                         "tools/profile_mcts.sh profiles)")
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
                    help="PV-MCTS leg: plain launches instead of hipGraph replay (rocprofv3 does not "
                         "attribute kernels launched from a graph)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks as child processes even for --gpus 1 (what --gpus N > 1 does "
                         "when no launcher has set WORLD_SIZE)")
    ap.add_argument("--mcts-per-playout", action="store_true",
                    help="PV-MCTS leg: the per-playout launches instead of the persistent search")
    ap.add_argument("--mcts-value-f32", action="store_true",
                    help="PV-MCTS leg: MIOpen float32 convolutions for the Value net instead of the "
                         "split-f16 MFMA kernels")
    ap.add_argument("--mcts400-turns", type=int, default=4,
                    help="PV-MCTS at 400 playouts per move (one GPU's share of BASELINE configs[3]): turns of the "
                         "bounded sample, 0 = skip")
    ap.add_argument("--nthr1-turns", type=int, default=8,
                    help="PV-MCTS with n_thr = 1 (SURVEY.md 8d: the policy net inside every playout): turns of "
                         "the bounded sample, 0 = skip")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-budget", type=float, default=4.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-start", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-seed", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:   # a CPU-baseline worker process (cpu_workers): never touches the GPU
        cpu_worker_main(args.cpu_worker, args.cpu_worker_budget, args.cpu_worker_start, args.cpu_worker_seed)
        return
    if args.rollout_only:
        args.mcts_turns, args.train_iters, args.no_cpu_baseline, args.nthr1_turns, args.mcts400_turns = 0, 0, True, 0, 0

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher
        # (nothing in it has touched the GPU yet) and exits with its ranks' status
        sys.exit(spawn_ranks(args.gpus))
    # stdout carries ONE line, rank 0's JSON: whatever a library prints there (RCCL's version banner
    # at communicator creation, MIOpen notices) goes to stderr instead -- file descriptor 1 points
    # at stderr for the whole run and the line is written to the original descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (torch.distributed.run "
                         "--nproc-per-node %d, or no launcher at all)" % (args.gpus, world, args.gpus))
    # (rehearsal of the N > 1 path on a one-GPU box, tests/test_dist_gpu.py: IAGO_BENCH_BACKEND=gloo
    # with every rank on IAGO_BENCH_DEVICE=0 -- RCCL refuses two ranks on one device)
    backend = os.environ.get("IAGO_BENCH_BACKEND", "nccl")
    device = int(os.environ.get("IAGO_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (any N)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    B, K, W = args.boards, args.steps, args.warmup
    head, (w, b) = rollout_leg(args, world, rank, dist)

    mcts = None
    if args.mcts_turns != 0:
        mcts = mcts_leg(args.mcts_games, args.mcts_sims, max(args.mcts_turns, 0),
                        args.mcts_turns < 0, world, rank, dist, value_f32=args.mcts_value_f32,
                        use_graph=not args.mcts_eager,
                        persistent=False if (args.mcts_per_playout or args.mcts_eager) else None)
    if mcts is not None and not args.mcts_value_f32 and args.mcts_turns < 0 and not args.mcts_only:
        # the same leg with MIOpen float32 convolutions for the Value net, on a bounded
        # sample (first 4 turns), for comparison with the split-f16 kernels
        ref = mcts_leg(args.mcts_games, args.mcts_sims, 4, False, world, rank, dist, value_f32=True)
        mcts["value_f32_sample"] = {k: ref[k] for k in ("leaf_evals_per_sec", "leaf_evals", "seconds",
                                                        "turns_played", "value_conv")}
    if mcts is not None and not args.mcts_value_f32 and args.mcts_turns < 0 and not args.mcts_only and not args.mcts_eager:
        # the same full games on the per-playout launches (rounds 1-3's engine), for comparison
        ref = mcts_leg(args.mcts_games, args.mcts_sims, 0, True, world, rank, None, persistent=False)
        mcts["per_playout_launches"] = {k: ref[k] for k in ("leaf_evals_per_sec", "games_per_sec", "leaf_evals", "policy_evals",
                                                             "value_evals", "seconds", "turns_played", "engine")}
    nthr1 = None
    if mcts is not None and args.nthr1_turns > 0 and not args.mcts_only and not args.mcts_value_f32:
        # "also report n_thr = 1" (MCTS.py:80,109): every leaf expands at its
        # second visit, so the look-ahead cannot apply (no visits to run ahead of) and the policy net
        # sits inside every playout: select, pending, policy_resident_kernel x 2 on the expanding
        # leaves, expand, continued select, fresh_leaves, value_rollout_kernel, mix_backup -- one
        # hipGraph replay per playout.  Bounded sample: the first turns of the same 1024 games
        r1 = mcts_leg(args.mcts_games, args.mcts_sims, args.nthr1_turns, False, world, rank, dist, n_thr=1,
                      use_graph=not args.mcts_eager)
        nthr1 = {k: r1[k] for k in ("leaf_evals_per_sec", "leaf_evals", "policy_evals", "value_evals", "seconds",
                                    "turns_played", "sims_per_move", "games_per_gpu", "n_thr", "policy_lookahead",
                                    "value_cache", "tree_nodes_used_max", "tree_capacity", "config")}
        nthr1["sample"] = ("the first %d turns of the games (bounded sample: the value net's position table answers more "
                           "requests there than over a whole game), policy net inside every playout" % args.nthr1_turns)
        nthr1["default_n_thr15_leaf_evals_per_sec"] = mcts["leaf_evals_per_sec"]
    m400 = None
    if mcts is not None and args.mcts400_turns > 0 and not args.mcts_only and not args.mcts_value_f32:
        # BASELINE configs[3] at one GPU's share: the reference's 10 s budget per move (MCTS.py:80,139-147) as
        # 400 playouts, `--mcts-games` games per GPU; bounded sample of the first turns (a full game at 400
        # playouts is ~4 s: tools/time_value_ahead.py with SIMS=400), with the tuple gather when N > 1
        r4 = mcts_leg(args.mcts_games, 400, args.mcts400_turns, False, world, rank, dist, use_graph=not args.mcts_eager)
        m400 = {k: r4[k] for k in ("leaf_evals_per_sec", "leaf_evals", "policy_evals", "value_evals", "seconds",
                                   "turns_played", "sims_per_move", "games_per_gpu", "tree_nodes_used_max",
                                   "tree_capacity", "config")}
        m400["config"] = m400["config"].replace("configs[2]", "configs[3] (one GPU's share)")
        m400["sample"] = ("the first %d turns of the games (bounded sample; the games start from ONE position, so these turns "
                          "repeat each other's positions and the value net's position table answers more requests than over a "
                          "whole game: value_evals are the evaluations executed)" % args.mcts400_turns)
    train = reinforce_leg(args.train_iters, world, rank, dist) if args.train_iters > 0 else None
    b1 = mcts_b1_leg() if (mcts is not None and rank == 0 and not args.mcts_only) else None

    if rank == 0:
        line = {
            "metric": "self-play games/sec", "value": head["value"], "unit": "games/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 bitboards + f32 policy", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d parallel Othello boards per GPU in ONE launch "
                                   "per step, launches serialized on one stream, rollout-policy-only "
                                   "playouts from the start position to the end, shipped "
                                   "RolloutPolicy weights" % B,
                       "boards_per_gpu": B, "boards_per_launch": B, "games_per_step": world * B,
                       "launches_in_flight": 1,
                       "tuple_allgather": head["exchange"] if dist is not None else "none"},
            "repeats": head["repeats"], "timed_region_s": head["timed_region_s"],
            "board_steps_per_sec": head["board_steps_per_sec"],
            "board_steps_per_game": head["board_steps_per_game"],
            "roofline": head["roofline"],
        }
        for key in ("overlapped", "large_batch"):
            if key in head:
                line[key] = head[key]
        if mcts is not None:
            # the north star's second metric as top-level scalars (the driver's `parsed` keeps scalars
            # only): PV-MCTS configs[2], split-f16 Value + three-piece SLPolicy, whole job over all ranks
            line["leaf_evals_per_sec"] = mcts["leaf_evals_per_sec"]
            if "games_per_sec" in mcts:
                line["mcts_games_per_sec"] = mcts["games_per_sec"]
            if "per_playout_launches" in mcts:
                line["leaf_evals_per_sec_per_playout_launches"] = mcts["per_playout_launches"]["leaf_evals_per_sec"]
            rl = mcts.get("roofline") or {}
            if "useful_tflops" in rl:
                line["mcts_useful_tflops"] = rl["useful_tflops"]                # measured in this run
                line["mcts_useful_frac_f16_peak"] = rl["useful_frac_f16_peak"]
                line["mcts_executed_frac_f16_peak"] = rl["frac"]
            ks = rl.get("kernels") or {}
            if ks:
                # per-kernel fractions come from a COMMITTED profile (named here), and only from one taken
                # on the kernel sources of this tree
                line["mcts_kernel_profile"] = ks.get("profile")
                line["mcts_kernel_profile_current"] = bool(ks.get("current"))
            for name, key in (("search_kernel", "mcts_search_kernel"), ("value_rollout_kernel", "mcts_value_kernel"),
                              ("policy_resident_kernel", "mcts_policy_kernel")):
                if ks.get("current") and name in ks and "frac" in ks[name]:
                    line[key + "_mfma_frac_committed_profile"] = ks[name]["frac"]
                    if "useful_frac_f16_peak" in ks[name]:
                        line[key + "_useful_frac_committed_profile"] = ks[name]["useful_frac_f16_peak"]
            line["mcts"] = mcts
        if m400 is not None:
            line["mcts400"] = m400
            line["leaf_evals_per_sec_400"] = m400["leaf_evals_per_sec"]
        if nthr1 is not None:
            line["mcts_nthr1"] = nthr1
            line["leaf_evals_per_sec_nthr1"] = nthr1["leaf_evals_per_sec"]
        if train is not None:
            line["reinforce"] = train
            line["reinforce_iters_per_sec"] = train["iters_per_sec"]
            line["reinforce_miopen_find_db"] = train["miopen_find_db"]
            line["reinforce_update_ms"] = train["update_ms"]
            line["reinforce_miopen_solvers"] = train["miopen_solvers"]
            if "mcts_fed" in train:
                line["reinforce_mcts_fed_rounds_per_sec"] = train["mcts_fed"]["rounds_per_sec"]
        if b1 is not None:
            line["mcts_single_game"] = b1
        if not args.no_cpu_baseline and world == 1:  # the CPU baselines are N = 1 figures
            line["cpu_baseline"] = cpu_baseline(w, b)
            line["cpu_baseline"]["python_loops_one_core"] = python_loop_baseline(w, b)
            if mcts is not None:
                mcts["cpu_baseline"] = mcts_cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
