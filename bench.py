#!/usr/bin/env python3
"""Benchmark of the MI355X self-play hot path (contract: see DESIGN.md, "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline workload (BASELINE.json configs[2], the configuration the north star's metric is quoted on): per
GPU, 1024 PV-MCTS self-play games of 8x8 Othello, 100 playouts per move, both colours searching, SLPolicy +
Value nets with random-init weights (Chainer-default LeCunNormal, seed 0), the reference's constants
lmbda = 0.5, c_puct = 1, n_thr = 15, played from the start position to the end.  ONE STEP = one batch of
1024 whole games per GPU = ONE launch of the persistent search kernel (csrc/search_kernel.hip) + with N > 1 the
all-gather of the batch's (s, pi, z) tuples over RCCL.  W untimed warm-up steps of the same kind, then EXACTLY K
timed steps between barrier + torch.cuda.synchronize() on both sides; the maximum over the ranks; every step
plays new games and starts from an EMPTY position table (nothing computed outside a step answers a request
inside it).  `value` = whole-job self-play games/s = N x 1024 x K / timed region; `leaf_evals_per_sec` = playouts/s
beside it (the metric's second half); `ms_per_step` = timed region / K; `step_ms_min / median / max` = the K
steps one by one.

`roofline` prices the dominant kernel, search_kernel, against the dense f16 MFMA peak (2.5 PFLOP/s,
MI355X_MICROARCH.md): `achieved` = SURVEY.md 8(d)'s algorithmic FLOPs (122.99 MFLOP per Value evaluation, 122.85
per SLPolicy evaluation) x the evaluations a launch executed / the launch's duration, measured with HIP event pairs
on the launch stream around every timed launch; `executed_*` = the f16 MFMA FLOPs issued for them (3 MFMAs per
product of the Value net, 6 of SLPolicy: float32-grade results on f16 units); `traffic`, `rocprof_kernel_avg_ms`
and `executed_frac_pmc` come from the committed rocprofv3 profile of this command (tools/profile_mcts.sh ->
profiles/), quoted only when it was taken on these kernel sources.  `cpu_baseline` times the oracle's restatement
of MCTS.py (oracle/mcts_py.py, torch-CPU B = 1 nets, C rollout) on the host: P single-threaded processes, and one
core beside it.

Nested, none of them `value`: `rollout` (BASELINE configs[1], the headline of rounds 1-4: 4096 boards per launch
played to the end by the fused rollout kernel, with its own HBM / VALU roofline and CPU baselines;
--rollout-only prints it as the line's headline), `mcts.per_playout_launches` (the same games on the per-playout
engine), `mcts400` / `mcts_nthr1` (one batch of whole games at 400 playouts per move -- one GPU's share of
configs[3] -- / with n_thr = 1) and their `*_opening` samples, `reinforce` (configs[4] in miniature: a set's 64 games as one
launch, the update -- forward, loss, backward, Adam -- as this repository's split-f16 kernels; with the CPU side beside it), `configs0` (one SL-vs-SL game on the CPU restatement), `mcts_single_game`.
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
             persistent=None, steps=1, warmup_steps=0, fresh_table=True):
    """BASELINE configs[2]: PV-MCTS self-play, `n_games` games per GPU, `n_sims` playouts per move, SLPolicy +
    Value with random-init weights (Chainer-default LeCunNormal, seed 0), reference constants lmbda=0.5,
    c_puct=1, n_thr=15, both colours search.  One leaf-eval = one playout (value net + rollout at the leaf; the
    policy net runs on expansions).  ONE STEP = one batch of `n_games` whole self-play games per GPU (with the
    persistent search: ONE launch) + with N > 1 the all-gather of its (s, pi, z) tuples; `warmup_steps` untimed
    steps, then exactly `steps` timed ones between barrier + synchronize on both sides.  Every step plays NEW
    games (the rollouts' Philox streams go on from the previous step's) and -- fresh_table -- starts from an
    EMPTY position table: nothing computed outside a step (warm-up, earlier steps) answers a request inside
    it.  --mcts-turns N > 0 times a bounded sample of the first N turns instead of whole games."""
    from iago_amd import engine, network, ops
    from iago_amd.dist import all_gather_into, gather_tuples
    w, b = shipped_rollout_weights()
    torch.manual_seed(0)
    policy = network.SLPolicy().cuda().eval()
    value = network.Value().cuda().eval()
    value.split_f16 = not value_f32
    # the engine: the persistent search (one launch per whole game, every game on its own clock) wherever it
    # applies -- the split-f16 value net -- else per-playout launches replayed as hipGraphs (persistent=False: the
    # comparison figure `per_playout_launches`)
    if persistent is None:
        persistent = not value_f32 and use_graph
    m = engine.BatchedMCTS(n_games, policy, value, ops.RolloutWeights(w, b), lmbda=0.5, c_puct=1.0,
                           n_thr=n_thr, seed=7, game_id_base=rank * n_games, use_graph=use_graph and not persistent,
                           persistent=persistent,
                           capacity=engine.suggest_capacity(n_sims, n_thr, moves=64 if full_games else n_turns + 4))
    eng = engine.SelfPlayEngine(m, max_turns=(128 if full_games else n_turns))
    m.enable_stats()
    m.warmup()                 # MIOpen kernel selection for every batch bucket
    gather = dist is not None and full_games
    if warmup_steps <= 0:
        warm = engine.SelfPlayEngine(m, max_turns=4).play(16, record=True)  # allocator, code objects
        if gather:
            gather_tuples(warm.tuples())   # (RCCL sets up a collective of a new size class on its first use)
    for _ in range(max(warmup_steps, 0)):      # untimed warm-up STEPS: whole batches like the timed ones
        warm = eng.play(n_sims, record=True)
        if gather:
            gather_tuples(warm.tuples())
    m.n_leaf_evals = m.n_policy_evals = 0
    m._value_total.zero_()
    is_p = bool(getattr(m, "persistent", False))
    if is_p:
        m._ps["totals"].zero_()
        m.launch_events = []
    m.stats.zero_()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_s, play_s, gather_s = [], [], []
    res = gathered = None
    n_tuples = 0
    for _ in range(steps):
        ts = time.perf_counter()
        if is_p and fresh_table and m._vtable is not None:
            m._vtable.zero_()   # (32 MB memset on the launch's stream, inside the timed region)
        res = eng.play(n_sims, record=True)    # (ends with the read-back of the launch's flags: the games are done)
        tp = time.perf_counter()
        if gather:
            gathered = gather_tuples(res.tuples())
            n_tuples = int(gathered["z"].numel())
            torch.cuda.synchronize()
        te = time.perf_counter()
        step_s.append(te - ts)
        play_s.append(tp - ts)
        gather_s.append(te - tp)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms = [e0.elapsed_time(e1) for e0, e1 in m.launch_events] if is_p else []
    if is_p:
        m.launch_events = None
    stats = torch.tensor([dt, m.n_leaf_evals, m.n_policy_evals, m.n_value_evals], dtype=torch.float64,
                         device="cuda")
    diag = None
    if dist is not None:
        tm = stats[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats)
        stats[0] = tm[0]
        # the N > 1 run says what it did (the driver's 8-rank run cannot be debugged): the ranks the process group
        # really has, every rank's time inside its own games, the time of the gathers, and whether the ranks
        # played different games (a checksum of every rank's recorded moves and final boards)
        mv = res.move.to(torch.int64).reshape(-1)
        chk = ((mv + 2) * (torch.arange(mv.numel(), device=mv.device, dtype=torch.int64) % 1000003 + 1)).sum() \
            + (res.final_p1 ^ (res.final_p2 * 31)).sum()
        mine = torch.stack([chk.to(torch.int64).reshape(()),
                            torch.tensor(int(sum(play_s) * 1e6), dtype=torch.int64, device="cuda"),
                            torch.tensor(int(sum(gather_s) * 1e6), dtype=torch.int64, device="cuda"),
                            torch.tensor(int(rank), dtype=torch.int64, device="cuda")])
        every = torch.empty(world * 4, dtype=torch.int64, device="cuda")
        all_gather_into(every, mine)
        every = every.reshape(world, 4).cpu()
        sums, plays, gathers, ranks = (every[:, i].tolist() for i in range(4))
        plays, gathers = [x * 1e-6 for x in plays], [x * 1e-6 for x in gathers]
        diag = {"ranks_seen": int(dist.get_world_size()), "ranks_reporting": sorted(int(r) for r in ranks),
                "rank_play_seconds_min": min(plays), "rank_play_seconds_max": max(plays),
                "per_rank_games_per_sec": [n_games * steps / max(x, 1e-9) for x in plays] if full_games else None,
                "gather_ms_per_step_max": max(gathers) / max(steps, 1) * 1e3,
                "ranks_played_different_games": len(set(sums)) == world,
                "backend": dist.get_backend()}
    dt, leaf, pol, val = (float(x) for x in stats.tolist())
    totals = [int(x) for x in m._ps["totals"].tolist()] if is_p else None
    # Value / SLPolicy FLOPs per evaluation, for the evaluations EXECUTED (with the
    # value cache the net runs on a leaf's first visit only; the policy look-ahead also evaluates
    # leaves that never expand)
    flops = val * VALUE_FLOP + pol * POLICY_FLOP
    ss = sorted(step_s)
    out = {"leaf_evals_per_sec": leaf / dt, "leaf_evals": int(leaf), "policy_evals": int(pol),
           "value_evals": int(val), "value_inline": int(m.n_value_inline), "value_ahead": int(m.n_value_ahead),
           "steps": steps, "warmup_steps": max(warmup_steps, 0),
           "step_seconds": step_s, "step_ms_min": ss[0] * 1e3, "step_ms_median": ss[len(ss) // 2] * 1e3,
           "step_ms_max": ss[-1] * 1e3,
           "async_steps": int(m.n_steps) if m.async_steps else None,
           "persistent": ({"net_workgroups": getattr(m, "net_workgroups_launched", m.net_workgroups),
                           "game_workgroups": -(-n_games // m.games_per_workgroup),
                           "resident_workgroups_of_the_device": m.resident_workgroups,
                           # (0: ONE launch, every workgroup a CU of its own; else the role split: the game workgroups as a
                           # launch of their own, two per CU on this many CUs, the net workgroups on all the others)
                           "role_split_game_cus": int(getattr(m, "split_cus", 0)),
                           "totals": totals,
                           "totals_legend": "[0] value evaluations on a game's critical path, [1] policy evaluations, [2] "
                                            "game-workgroup iterations, [3] pair walks, [4] / [5] net workgroups' waiting / "
                                            "walking time (100 MHz ticks, summed), [6] idle game-workgroup iterations, [7] game "
                                            "workgroups' run time, [8] position-table hits, [11] values walked ahead of "
                                            "their first visit, [12] hits on a value the asking game itself put there",
                           "position_table": {"fresh_per_step": bool(fresh_table), "hits": totals[8],
                                              "hits_same_game": totals[12], "hits_cross_game": totals[8] - totals[12],
                                              "note": "cross-game hits exist because all games of a batch start from "
                                                      "ONE position with ONE net: value_func (MCTS.py:97-103) is a pure "
                                                      "function of the position"},
                           "kernel_ms": kernel_ms}
                          if is_p else None),
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
                      "their first visit: value_ahead)" % (m.games_per_workgroup, getattr(m, "net_workgroups_launched", m.net_workgroups))
                      if m.persistent else
                      "per-playout launches (descent, leaf evaluation, backup) replayed as hipGraphs, policy look-ahead "
                      "batches on a second stream"),
           "value_cache": bool(m.value_cache), "policy_lookahead": int(m.lookahead),
           "seconds": dt, "turns_played": res.n_turns, "sims_per_move": n_sims,
           "games_per_gpu": n_games, "full_games": bool(full_games),
           "batches_replayed_turn_by_turn": int(getattr(eng, "n_replayed", 0)),
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
    if diag is not None:
        out["ranks"] = diag
    if full_games:
        out["games_per_sec"] = world * n_games * steps / dt
        if gathered is not None:
            out["gathered_tuples"] = n_tuples
            out["play_seconds_rank0"] = sum(play_s)   # the rest of `seconds`: packing + all-gather of the tuples + barrier
    if is_p and kernel_ms:
        # the dominant kernel of the leg, per launch, on THIS rank's GPU: HIP event pairs around the launches on
        # their stream (the pair also spans the four small memsets that zero the rings before the kernel)
        n_l = len(kernel_ms)
        k_s = sum(kernel_ms) * 1e-3
        val_r, pol_r = totals[0] + totals[11], totals[1]
        useful = (val_r * VALUE_FLOP + pol_r * POLICY_FLOP) / k_s / 1e12
        executed = (val_r * VALUE_MFMA_PER_BOARD + pol_r * POLICY_MFMA_PER_BOARD) * MFMA_FLOP_32x32x16 / k_s / 1e12
        out["kernel_roofline"] = {
            "bound": "mfma", "achieved": useful, "peak": F16_PEAK_TF, "unit": "TFLOP/s", "frac": useful / F16_PEAK_TF,
            "kernel": "search_kernel (persistent search: one launch = one batch of whole games)",
            "launches": n_l, "kernel_ms": k_s / n_l * 1e3, "kernel_ms_min": min(kernel_ms), "kernel_ms_max": max(kernel_ms),
            "algorithmic_flops_per_launch": (val_r * VALUE_FLOP + pol_r * POLICY_FLOP) / n_l,
            "units_per_launch": {"value_evaluations": val_r / n_l, "policy_evaluations": pol_r / n_l,
                                 "leaf_evals": m.n_leaf_evals / n_l},
            "flops_per_unit": {"value_evaluation": VALUE_FLOP, "policy_evaluation": POLICY_FLOP},
            "executed_tflops": executed, "executed_frac": executed / F16_PEAK_TF,
            "executed_definition": "f16 MFMA FLOPs issued: 3 MFMAs per product of the Value net, 6 per product of "
                                   "SLPolicy (float32-grade results on f16 matrix units), counted from the evaluations "
                                   "this run executed in units of 32,768 FLOP = two v_mfma_f32_16x16x32_f16 (the K loops' "
                                   "shape since round 5: 16,384 FLOP each) or one v_mfma_f32_32x32x16_f16 (the heads)",
            "achieved_definition": "ALGORITHMIC FLOPs (SURVEY.md 8(d): 122.99 MFLOP per Value evaluation, 122.85 "
                                   "per SLPolicy evaluation) of the evaluations executed / the launches' duration",
            "useful_x_f32_matrix_peak": useful / F32_MATRIX_PEAK_TF,
        }
    out["device_memory_bytes"] = m.memory_bytes()
    m.close()   # the captured graphs go now, not whenever the garbage collector finds the engine
    return out


MFMA_FLOP_32x32x16 = 2 * 32 * 32 * 16    # one v_mfma_f32_32x32x16_f16 wave-instruction
MFMA_FLOP_16x16x32 = 2 * 16 * 16 * 32    # one v_mfma_f32_16x16x32_f16: the walks' K loops since round 5 (half the FLOPs, half the cycles)
MIN_PROFILE_LAUNCHES = 100
F16_PEAK_TF, F32_MATRIX_PEAK_TF = 2500.0, 157.3    # MI355X_MICROARCH.md: dense f16 MFMA / f32 matrix
VALUE_FLOP, POLICY_FLOP = 122_994_944, 122_847_232   # algorithmic FLOPs per evaluation
# MFMA work one evaluated board executes in the search's net kernels, in units of 32,768 FLOP (one 32x32x16 instruction,
# the shape of rounds 2-4; the K loops now issue two v_mfma_f32_16x16x32_f16 for each): blocks 2..8 = (36 + 6 x 72)
# k16-steps x (2 tiles x 3 or 6 MFMAs) x 4 waves (+ the Value head's 8 x 2 x 3 on one wave)
VALUE_MFMA_PER_BOARD = 468 * 6 * 4 + 48
POLICY_MFMA_PER_BOARD = 468 * 12 * 4


def csrc_sha16():
    """Identity of the kernel sources (iago_amd/csrc/*, include/iago_hip.h): what a committed rocprofv3
    summary must carry (tools/summarize_mcts_profile.py writes it) to be quoted beside measured numbers."""
    import glob
    import hashlib
    h = hashlib.sha256()
    here = os.path.dirname(os.path.abspath(__file__))   # (the sources of THIS file's tree, wherever ROOT points)
    for path in sorted(glob.glob(os.path.join(here, "iago_amd", "csrc", "*"))) + sorted(glob.glob(os.path.join(here, "include", "*.h"))):
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
                # (SQ_INSTS_MFMA counts wave-instructions of either shape: a profile of the round-5 walks is priced at
                # the short shape's 16,384 FLOP -- the heads' few 32x32x16 instructions, 0.2 % of a walk, are undercounted)
                short = prof.get("mfma_shape", "32x32x16") == "16x16x32"
                unit = 0.5 if short else 1.0
                tf = k["SQ_INSTS_MFMA"] * (MFMA_FLOP_16x16x32 if short else MFMA_FLOP_32x32x16) / (k["avg_us"] * 1e-6) / 1e12
                e.update({"mfma_insts_per_launch": k["SQ_INSTS_MFMA"], "executed_tflops": tf,
                          "bound": "mfma", "peak": F16_PEAK_TF, "frac": tf / F16_PEAK_TF})
                per_board, flop = {"value_rollout_kernel": (VALUE_MFMA_PER_BOARD, VALUE_FLOP),
                                   # (a launch of the two-launch forward walks half a net)
                                   "policy_resident_kernel": (POLICY_MFMA_PER_BOARD / 2, POLICY_FLOP / 2)}.get(name, (0, 0))
                if per_board:
                    boards = k["SQ_INSTS_MFMA"] * unit / per_board
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


def variant_profile(tag):
    """The search kernel's figures from the committed rocprofv3 profile of a variant of the headline leg
    (profiles/r*_<tag>_persistent_pmc_summary.json: tools/profile_mcts.sh with PROFILE_ARGS / PROFILE_SUF), quoted
    only when it was taken on these kernel sources."""
    import glob
    now = csrc_sha16()
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_persistent_pmc_summary.json" % tag)))):
        with open(path) as f:
            prof = json.load(f)
        k = prof.get("kernels", {}).get("search_kernel")
        if not k or k.get("calls", 0) * k.get("avg_us", 0.0) < 1e5:
            continue
        cur = prof.get("csrc_sha16") == now
        out = {"profile": os.path.basename(path), "profile_current": cur, "profile_command": prof.get("command")}
        if cur:
            tf = k["SQ_INSTS_MFMA"] * MFMA_FLOP_16x16x32 / (k["avg_us"] * 1e-6) / 1e12
            out.update({"traffic": k.get("hbm_bytes_per_launch"), "rocprof_kernel_avg_ms": k["avg_us"] / 1e3,
                        "executed_frac_pmc": tf / F16_PEAK_TF, "l2_hit_rate": k.get("l2_hit_rate"),
                        "lds_bank_conflict_share": (k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"]
                                                    if k.get("SQ_LDS_IDX_ACTIVE") else None)})
        return out
    return {"profile": None, "profile_current": False}


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


CPU_WINDOWS = (("mcts", 8.0), ("python_loops", 4.0), ("sl_game", 4.0))   # kinds of CPU work and their timed windows (s)
CPU_WINDOW_GAP = 1.5
_cpu_pool = {}


def cpu_workers(kind, budget_s=None):
    """the reference's own execution model "on P = os.cpu_count() independent
    worker processes, P stated" -- the reference is one single-threaded Python process per game
    (MCTS.py:139-147, mcts_self_play.py:25-29), so P of them side by side is what the host's
    cores give it.  P = host_cores() copies of this script in --cpu-worker mode (CHILD processes
    that never touch the GPU) run ALL kinds of CPU work one after the other (CPU_WINDOWS: one import of
    torch per process), every kind in a window of its own that starts at a common wall clock instant;
    the aggregate of a kind is the sum of the workers' counts over its window."""
    import subprocess
    if not _cpu_pool:
        P = min(host_cores(), 64)
        start_at = time.time() + 20.0        # imports + warm-up of the slowest worker
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "all",
                                   "--cpu-worker-start", repr(start_at), "--cpu-worker-seed", str(i)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                  env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1"))
                 for i in range(P)]
        rows = []
        for pr in procs:
            out, _ = pr.communicate(timeout=600)
            if pr.returncode == 0 and out.strip():
                rows.append(json.loads(out.strip().splitlines()[-1]))
        for k, _ in CPU_WINDOWS:
            mine = [r[k] for r in rows if k in r]
            if len(mine) != P:
                _cpu_pool[k] = {"error": "%d of %d CPU workers finished" % (len(mine), P)}
            else:
                _cpu_pool[k] = {"count": sum(r["count"] for r in mine), "seconds": max(r["seconds"] for r in mine),
                                "processes": P, "max_start_lag_s": max(r["late_s"] for r in mine),
                                "steps": sum(r.get("steps", 0) for r in mine)}
    return _cpu_pool[kind]


def cpu_worker_main(kind, budget_s, start_at, seed):
    """One worker of cpu_workers(): warm every kind up, then per kind wait for the common start of its window and
    run for its budget; ONE line on stdout: {kind: {count, steps, seconds, late_s}}."""
    w, b = shipped_rollout_weights()
    torch.set_num_threads(1)

    def make(kind):
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
            return lambda: (1, py_loops.simulate(s0, 1, policy, rs)[1])
        if kind == "mcts":
            from oracle import oracle as orc
            m = _cpu_mcts(w, b, seed)

            def unit():   # 20 playouts of the worker's own game tree from the start position
                m.get_move(orc.initial_state(), 1, 20)
                return 20, 0
            return unit
        if kind == "sl_game":
            play = _cpu_sl_game(seed)
            return lambda: (1, play())   # one SLPolicy-vs-SLPolicy game (src/rl_self_play.py:27-31)
        raise SystemExit("unknown --cpu-worker kind %r" % kind)

    windows = CPU_WINDOWS if kind == "all" else ((kind, budget_s),)
    units = {k: make(k) for k, _ in windows}
    for k in units:
        units[k]()                           # warm-up
    out, t_start = {}, start_at
    for k, budget in windows:
        late = max(0.0, time.time() - t_start)
        while time.time() < t_start:
            time.sleep(0.01)
        count = steps = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget:
            c, st = units[k]()
            count += c
            steps += st
        out[k] = {"count": count, "steps": steps, "seconds": time.perf_counter() - t0, "late_s": late}
        t_start += budget + CPU_WINDOW_GAP
    print(json.dumps(out), flush=True)


def _cpu_mcts(w, b, seed):
    """The reference's own algorithm for the PV-MCTS leg on ONE host core: the oracle's restatement of MCTS.playout
    (oracle/mcts_py.py) with float32 torch-CPU SLPolicy / Value (B = 1 calls, one thread, like the reference's
    Chainer calls) and the C oracle's rollout; same constants as the GPU leg."""
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

    return mcts_py.MCTS(pol, val, roll, lmbda=0.5, c_puct=1, n_thr=15)


def _cpu_sl_game(seed):
    """BASELINE configs[0]: one SLPolicy-vs-SLPolicy game (src/rl_self_play.py:27-31,111-145) as the reference runs
    it -- the oracle's restatement (oracle/mcts_py.rl_game) with a float32 torch-CPU SLPolicy called on ONE board
    per move (random init, seed 0), one thread, numpy draws.  Returns a callable that plays one game and returns
    the moves colour 1 made."""
    from iago_amd import network
    from oracle import mcts_py
    torch.manual_seed(0)
    net = network.SLPolicy().eval()
    rs = np.random.RandomState(seed)

    def pol(x):
        with torch.no_grad():
            return net(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))).numpy().reshape(64)

    def uniforms():
        while True:
            yield float(rs.random_sample())

    def play():
        states, actions, z, final = mcts_py.rl_game(pol, pol, uniforms())
        return len(actions)

    return play


def _all_cores(kind, budget_s, unit, what):
    pw = cpu_workers(kind, budget_s)
    if "error" in pw:
        return pw
    return {"value": pw["count"] / pw["seconds"], "unit": unit, "cores": pw["processes"], "kind": "port",
            "sample": "%d independent single-threaded worker processes (%s), %d in a common %.1f s window"
                      % (pw["processes"], what, pw["count"], pw["seconds"])}


def mcts_cpu_baseline(n_sims=600, leaf_evals_per_game=None):
    """cpu_baseline of the headline: the MCTS restatement on ONE core (n_sims playouts of one game from the start
    position) and on P = host_cores() independent processes; `leaf_evals_per_game` (from the GPU run: playouts one
    finished self-play game took) converts leaf-evals/s into the headline's unit, games/s."""
    from oracle import oracle as orc
    w, b = shipped_rollout_weights()
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        m = _cpu_mcts(w, b, 0)
        m.get_move(orc.initial_state(), 1, 5)  # warm-up
        t0 = time.perf_counter()
        m.get_move(orc.initial_state(), 1, n_sims)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthreads)
    one = {"value": n_sims / dt, "unit": "leaf-evals/s", "cores": 1, "kind": "port",
           "sample": "%d playouts of one game from the start position, oracle/mcts_py.py + "
                     "torch-CPU fp32 nets (1 thread) + C rollout, %.1f s" % (n_sims, dt)}
    out = _all_cores("mcts", None, "leaf-evals/s", "one game tree each, the same restatement")
    if "error" in out:
        out = dict(one, all_cores=out)
    else:
        out["one_core"] = one
    if leaf_evals_per_game:
        out["leaf_evals_per_game"] = leaf_evals_per_game
        out["games_per_sec"] = out["value"] / leaf_evals_per_game
        if "one_core" in out:
            out["one_core"]["games_per_sec"] = one["value"] / leaf_evals_per_game
    # SURVEY.md section 6 / BASELINE.md section 2: the reference's OWN board code (rl_env.py imported under stub
    # chainer modules) measured in the survey container, for scale -- the restatements here are ~3x faster
    out["reference_loops_games_per_sec_per_core"] = 73.9
    out["reference_loops_note"] = ("uniform-random playouts, board logic only, rl_env.py itself on 1 core of the survey "
                                   "container (BASELINE.md section 2); oracle/py_loops.py, timed in `rollout.cpu_baseline`, "
                                   "runs the same loops ~3x faster (no numpy scalar boxing in the inner loop)")
    return out


def sl_game_cpu_baseline(budget_s=4.0):
    """BASELINE configs[0] (plumbing, no GPU) and the CPU side of configs[4]'s self-play: SLPolicy-vs-SLPolicy games
    on the CPU restatement, 1 core and P processes."""
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        play = _cpu_sl_game(0)
        play()
        games = moves = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            moves += play()
            games += 1
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthreads)
    out = {"value": games / dt, "unit": "games/s", "cores": 1, "kind": "port",
           "seconds_per_game": dt / games, "colour1_moves_per_game": moves / games,
           "sample": "%d SLPolicy-vs-SLPolicy games, oracle/mcts_py.rl_game + torch-CPU fp32 SLPolicy on one board per "
                     "move (1 thread), %.1f s" % (games, dt)}
    out["all_cores"] = _all_cores("sl_game", budget_s, "games/s", "the same restatement")
    return out


def reinforce_cpu_baseline(sl):
    """CPU side of BASELINE configs[4]: one REINFORCE set = 64 SL-vs-SL games (src/train_rl.py:41-51; rate from
    sl_game_cpu_baseline) + one update (src/train_rl.py:55-66: planes of the recorded states, double-softmax
    cross-entropy x z, mean, Adam) on torch-CPU float32 with all host threads, on a synthetic set of the recorded
    size (64 games x ~30 colour-1 moves)."""
    from iago_amd import network
    import torch.nn.functional as F
    n = int(round(64 * sl.get("colour1_moves_per_game", 30.0)))
    torch.manual_seed(0)
    threads0 = torch.get_num_threads()
    torch.set_num_threads(min(host_cores(), 64))      # (the cores this process may really use, not the machine's)
    net = network.SLPolicy().train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
    x = (torch.rand(n, 2, 8, 8) < 0.3).float()
    y = torch.randint(0, 64, (n,))
    z = torch.randint(0, 2, (n,)).float() * 2 - 1
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        opt.zero_grad()
        pred = net(x)
        loss = (F.cross_entropy(pred, y, reduction="none") * z).mean()   # (pred is a softmax output already: the quirk)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    upd = min(times)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(threads0)
    ac = sl.get("all_cores", {})
    gps_all = ac.get("value")
    out = {"unit": "sets/s", "kind": "port", "update_seconds": upd, "update_threads": nthreads,
           "update_rows": n,
           "one_core_games": {"value": 1.0 / (64.0 / sl["value"] + upd), "cores": 1,
                              "note": "64 games on one core + the update"},
           "sample": "64 x the measured seconds per SL-vs-SL game + one torch-CPU update of %d rows (best of 3)" % n}
    if gps_all:
        out["value"] = 1.0 / (64.0 / gps_all + upd)
        out["cores"] = ac.get("cores")
    else:
        out["value"], out["cores"] = out["one_core_games"]["value"], 1
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
    from iago_amd import train_rl
    from iago_amd.train_rl import ReinforceTrainer
    torch.manual_seed(0)
    db_before = miopen_find_db_state()
    tr = ReinforceTrainer(network.SLPolicy(), pool_dir=None, N=32, seed=rank)
    for _ in range(4):
        tr.step()  # warm-up: allocator, weight-layout caches (autograd path: MIOpen's kernel selection)
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
    # one more set's update timed on its own after the timed region.  (IAGO_NATIVE_GRAD=0, autograd over the tensor
    # library's float32 convolutions: which backward-convolution solvers MIOpen picked then decides the figure -- tuned
    # ones from a find-db that holds THIS problem: ~10 ms; its immediate-mode fallback: ~40 ms)
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
           "update_ms": update_ms,
           # the update's matrix work: forward + input gradients + weight gradients = 3 x 122.85 MFLOP per row
           # (SURVEY 8d's per-evaluation figure), each product sum as 3 f16 MFMAs; Adam, the weight pieces and the
           # gathers are inside update_ms
           "update_rows": int(tup["z"].numel()),
           "update_tflops_algorithmic": 3 * 122.85e6 * int(tup["z"].numel()) / (update_ms * 1e-3) / 1e12,
           "update_frac_f16_peak_executed": 9 * 122.85e6 * int(tup["z"].numel()) / (update_ms * 1e-3) / 1e12 / 2500.0,
           "update_engine": ("iago_policy_reinforce_grad: forward, loss and backward as split-f16 HIP kernels (3 MFMAs per "
                             "product sum, float32 accumulation; gradients within 1e-5 of float64 autograd, "
                             "tests/test_policy_grad_gpu.py) + ChainerAdam") if train_rl.NATIVE_GRAD else
                            "torch autograd over MIOpen float32 convolutions + ChainerAdam",
           "miopen_solvers": "not used" if train_rl.NATIVE_GRAD else
                             ("tuned (find-db)" if update_ms < 20.0 else "immediate-mode fallback"),
           "miopen_find_db": db_before,
           "config": "64 policy-vs-policy games per set (SLPolicy, random init, fp32) + "
                     "double-softmax REINFORCE update, ChainerAdam + WD 5e-4",
           "scaling_note": "the reference-sized set is a latency chain (60 sequential one-board policy walks per game): "
                           "sharding its 64 games over N GPUs does not shorten it; with N > 1 see weak_scaling_set"}
    if world > 1:
        # The reference-sized set cannot speed up with N: its 64 games are 60 sequential one-board policy walks each
        # (a latency chain of ~6 ms on as many CUs as there are games), sharded they take the same 6 ms on every GPU.
        # The form that scales (SURVEY 8d-5 allows "64+ games"): 64 games PER RANK, one update on the gathered
        # 64 x N games' rows (src/train_rl.py:41-66 with N = 32 x world).
        torch.manual_seed(0)
        tw = ReinforceTrainer(network.SLPolicy(), pool_dir=None, N=32 * world, seed=rank)
        for _ in range(min(2, n_iters)):
            tw.step()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        tw.gather_seconds, n_t = 0.0, 0
        for _ in range(n_iters):
            n_t += tw.step()["n_tuples"]
        torch.cuda.synchronize()
        dist.barrier()
        dtw = time.perf_counter() - t0
        out["weak_scaling_set"] = {"games_per_set": 64 * world, "games_per_rank": 64, "sets_per_sec": n_iters / dtw,
                                   "games_per_sec": 64 * world * n_iters / dtw, "tuples_per_iter": n_t / n_iters,
                                   "ms_per_iter": dtw / n_iters * 1e3, "gather_ms": tw.gather_seconds / n_iters * 1e3,
                                   "config": "64 policy-vs-policy games per RANK per set, the set's rows all-gathered, one "
                                             "double-softmax REINFORCE update on all of them on every rank"}
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
    pw = cpu_workers("python_loops")
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


def value_spread(n_games, n_sims, this_value, pattern="r06*_bench.json"):
    """min / max of `value` over the committed bench lines of this round's boxes (profiles/r06*_bench.json: one-GPU
    lines of the same workload) and this run: the pool's boxes differ by several per cent."""
    import glob
    vals, files = [], []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            with open(path) as f:
                d = json.loads(f.read().strip().splitlines()[-1])
            c = d.get("config", {})
            if (d.get("metric") == "self-play games/sec" and d.get("n_gpus") == 1 and c.get("games_per_gpu") == n_games
                    and c.get("sims_per_move") == n_sims and c.get("full_games")):
                vals.append(float(d["value"]))
                files.append(os.path.basename(path))
        except (OSError, ValueError, KeyError, IndexError):
            continue
    every = vals + [float(this_value)]
    return {"min": min(every), "max": max(every), "boxes": len(every), "committed_lines": files,
            "note": "games/s of this workload on one GPU: the committed lines of this round's boxes and this run"}


def preflight(dist, world, rank, device, json_fd, timeout_s=60.0, uuid=None, on="cuda"):
    """Before anything is timed on N > 1 ranks: ONE 1-element all-reduce and one all-gather of the ranks' device
    uuids, polled for at most `timeout_s` (no blocking wait: a collective that never completes must not hang the
    run).  Checks sum(rank) == N (N - 1) / 2 and that every rank sits on a DIFFERENT device.  On failure rank 0 (or
    whichever rank notices, if rank 0 is the one that hangs) prints one JSON line with "error" and the process exits
    non-zero at once (os._exit: no destructor waits on a dead communicator; nothing is re-executed).  Returns the
    dict that goes into the line."""
    t0 = time.perf_counter()
    if uuid is None:   # (uuid / on: the CPU test of this function, tests/test_dist_cpu.py)
        props = torch.cuda.get_device_properties(device)
        uuid = str(getattr(props, "uuid", "")) or "%s#%d" % (props.name, device)
    code = torch.tensor(list(uuid.encode()[:48].ljust(48, b" ")), dtype=torch.uint8, device=on)
    one = torch.tensor([rank], dtype=torch.int64, device=on)
    every = torch.empty(world * 48, dtype=torch.uint8, device=on)

    def fail(msg):
        out = {"error": "preflight: " + msg, "n_gpus": world, "rank": rank, "backend": dist.get_backend(),
               "device": device, "device_uuid": uuid, "seconds": time.perf_counter() - t0}
        sys.stdout.flush()
        os.write(json_fd if rank == 0 else 2, (json.dumps(out) + "\n").encode())
        os._exit(3)

    def finish(work, what):
        while not work.is_completed():
            if time.perf_counter() - t0 > timeout_s:
                fail("%s did not complete within %.0f s (rank %d of %d, backend %s)" % (what, timeout_s, rank, world,
                                                                                       dist.get_backend()))
            time.sleep(0.005)
        work.wait()

    finish(dist.all_reduce(one, async_op=True), "the 1-element all-reduce")
    finish(dist.all_gather_into_tensor(every, code, async_op=True) if dist.get_backend() == "nccl" else
           dist.all_gather(list(every.view(world, 48).unbind(0)), code, async_op=True), "the all-gather of the device uuids")
    if on == "cuda":
        torch.cuda.synchronize()
    got = int(one.item())
    if got != world * (world - 1) // 2:
        fail("all-reduce of the ranks gave %d, expected %d" % (got, world * (world - 1) // 2))
    uuids = [bytes(r.tolist()).decode().strip() for r in every.view(world, 48).cpu()]
    shared = os.environ.get("IAGO_BENCH_DEVICE") is not None   # the one-GPU rehearsal: every rank on one device
    if not shared and len(set(uuids)) != world:
        fail("ranks share a device: %s" % uuids)
    return {"seconds": time.perf_counter() - t0, "allreduce_sum_of_ranks": got, "distinct_devices": len(set(uuids)),
            "device_check": "skipped (rehearsal: IAGO_BENCH_DEVICE puts every rank on one device)" if shared else "ok",
            "backend": dist.get_backend()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps.  Default run: whole-game PV-MCTS batches (default 5); with --rollout-only: "
                         "4096-board rollout launches (default 2000)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps of the same kind (default 1 / 100)")
    ap.add_argument("--boards", type=int, default=BOARDS_PER_GPU)
    ap.add_argument("--rollout-steps", type=int, default=2000, help="steps of the nested configs[1] leg (`rollout`)")
    ap.add_argument("--rollout-warmup", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=0,
                    help="rollout leg: rounds of its steps in the timed region (0 = as many as make it >= 150 ms)")
    ap.add_argument("--streams", type=int, default=32,
                    help="HIP streams of the `overlapped` datapoint")
    ap.add_argument("--rollout-only", action="store_true",
                    help="only the configs[1] leg, printed as the line's headline (what tools/profile_rollout.sh profiles)")
    ap.add_argument("--mcts-only", action="store_true",
                    help="only the headline leg (what tools/profile_mcts.sh profiles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout-leg", action="store_true", help="skip the nested configs[1] leg")
    ap.add_argument("--large-boards", type=int, default=1 << 20,
                    help="extra occupancy datapoint of the rollout leg: boards in one launch (0 = skip)")
    ap.add_argument("--train-iters", type=int, default=3,
                    help="REINFORCE iterations of the training leg (0 = skip)")
    ap.add_argument("--mcts-games", type=int, default=1024)
    ap.add_argument("--mcts-sims", type=int, default=100)
    ap.add_argument("--mcts-nthr", type=int, default=15,
                    help="n_thr of the headline leg (MCTS.py:80,109; 15 = the reference's; 1 for profiles of the mcts_nthr1 variant)")
    ap.add_argument("--no-saturated", action="store_true", help="skip the 2048- / 4096-game batches (`mcts_saturated`)")
    ap.add_argument("--mcts-turns", type=int, default=-1,
                    help="headline leg: -1 = play the games to the end (default), N > 0 = a bounded "
                         "sample of the first N turns (rehearsals)")
    ap.add_argument("--keep-table", action="store_true",
                    help="headline leg: keep the position table across the steps (default: every step starts from an "
                         "empty table)")
    ap.add_argument("--mcts-eager", action="store_true",
                    help="PV-MCTS legs on the per-playout launches, eagerly (rocprofv3 does not attribute kernels "
                         "launched from a graph)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks as child processes even for --gpus 1 (what --gpus N > 1 does "
                         "when no launcher has set WORLD_SIZE)")
    ap.add_argument("--mcts-per-playout", action="store_true",
                    help="headline leg on the per-playout launches instead of the persistent search")
    ap.add_argument("--mcts-value-f32", action="store_true",
                    help="headline leg: MIOpen float32 convolutions for the Value net instead of the "
                         "split-f16 MFMA kernels")
    ap.add_argument("--mcts400-turns", type=int, default=-1,
                    help="PV-MCTS at 400 playouts per move (one GPU's share of BASELINE configs[3]): -1 = one batch of "
                         "whole games (default), N > 0 = only the opening sample of N turns, 0 = skip")
    ap.add_argument("--nthr1-turns", type=int, default=-1,
                    help="PV-MCTS with n_thr = 1 (SURVEY.md 8d: the policy net inside every playout): as --mcts400-turns")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-budget", type=float, default=4.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-start", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-seed", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:   # a CPU-baseline worker process (cpu_workers): never touches the GPU
        cpu_worker_main(args.cpu_worker, args.cpu_worker_budget, args.cpu_worker_start, args.cpu_worker_seed)
        return
    if args.steps is None:
        args.steps = 2000 if args.rollout_only else 5
    if args.warmup is None:
        args.warmup = 100 if args.rollout_only else 1
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("--steps >= 1 and --warmup >= 0 expected")
    if args.rollout_only:
        args.rollout_steps, args.rollout_warmup = args.steps, args.warmup

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
    pre = preflight(dist, world, rank, device, json_fd) if (dist is not None and world > 1) else None

    def log(msg):   # progress on stderr: a long run shows where it is
        if rank == 0:
            print("[bench %6.1f s] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)

    K, W = args.steps, args.warmup
    w, b = shipped_rollout_weights()
    line = None

    # ---- the nested configs[1] leg (the headline of rounds 1-4, and of --rollout-only)
    roll = None
    if args.rollout_only or not (args.no_rollout_leg or args.mcts_only):
        import copy
        ra = copy.copy(args)
        ra.steps, ra.warmup = args.rollout_steps, args.rollout_warmup
        log("configs[1]: %d-board rollout launches" % args.boards)
        roll, _ = rollout_leg(ra, world, rank, dist)
        roll["steps"], roll["warmup"] = ra.steps, ra.warmup
        roll["config"] = {"workload": "BASELINE configs[1]: %d parallel Othello boards per GPU in ONE launch "
                                      "per step, launches serialized on one stream, rollout-policy-only "
                                      "playouts from the start position to the end, shipped "
                                      "RolloutPolicy weights" % args.boards,
                          "boards_per_gpu": args.boards, "boards_per_launch": args.boards,
                          "games_per_step": world * args.boards, "launches_in_flight": 1,
                          "tuple_allgather": roll["exchange"] if dist is not None else "none"}
    if args.rollout_only:
        if rank == 0:
            line = {"metric": "self-play games/sec", "value": roll["value"], "unit": "games/s",
                    "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": roll["ms_per_step"],
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "u64 bitboards + f32 policy", "data": "synthetic", "config": roll["config"],
                    "repeats": roll["repeats"], "timed_region_s": roll["timed_region_s"],
                    "board_steps_per_sec": roll["board_steps_per_sec"],
                    "board_steps_per_game": roll["board_steps_per_game"], "roofline": roll["roofline"],
                    "note": "--rollout-only: the nested `rollout` leg of the default run (BASELINE configs[1]) as the headline"}
            for key in ("overlapped", "large_batch"):
                if key in roll:
                    line[key] = roll[key]
            if not args.no_cpu_baseline and world == 1:
                line["cpu_baseline"] = cpu_baseline(w, b)
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- the headline: BASELINE configs[2], K whole-game batches under the clock
    full = args.mcts_turns < 0
    log("configs[2]: %d warm-up + %d timed batches of %d games x %d playouts per move%s"
        % (W, K, args.mcts_games, args.mcts_sims, "" if full else " (first %d turns)" % args.mcts_turns))
    mcts = mcts_leg(args.mcts_games, args.mcts_sims, max(args.mcts_turns, 0), full, world, rank, dist,
                    value_f32=args.mcts_value_f32, use_graph=not args.mcts_eager, n_thr=args.mcts_nthr,
                    persistent=False if (args.mcts_per_playout or args.mcts_eager) else None,
                    steps=K, warmup_steps=W, fresh_table=not args.keep_table)
    log("configs[2]: %.3f s per batch, %.2f M leaf-evals/s" % (mcts["seconds"] / K, mcts["leaf_evals_per_sec"] / 1e6))
    extras = not args.mcts_only and not args.mcts_value_f32
    if extras and full and not args.mcts_eager:
        # the same games on the per-playout launches (rounds 1-3's engine), one batch, for comparison
        log("configs[2] on the per-playout launches (one batch)")
        ref = mcts_leg(args.mcts_games, args.mcts_sims, 0, True, world, rank, None, persistent=False)
        mcts["per_playout_launches"] = {k: ref[k] for k in ("leaf_evals_per_sec", "games_per_sec", "leaf_evals", "policy_evals",
                                                             "value_evals", "seconds", "turns_played", "engine")}

    def variant(name, sims, n_thr, turns_arg, what):
        """A variant of the headline leg: one batch of WHOLE games (turns_arg < 0) and the sample of its opening
        turns that rounds 3-4 reported (`*_opening`: from ONE start position the first turns repeat each other's
        positions, so the position table answers more requests there than over a whole game)."""
        outs = {}
        keys = ("leaf_evals_per_sec", "games_per_sec", "leaf_evals", "policy_evals", "value_evals", "seconds", "turns_played",
                "sims_per_move", "games_per_gpu", "n_thr", "policy_lookahead", "value_cache", "tree_nodes_used_max",
                "tree_capacity", "config", "batches_replayed_turn_by_turn", "kernel_roofline")
        if turns_arg < 0:
            log("%s: one batch of whole games" % name)
            r = mcts_leg(args.mcts_games, sims, 0, True, world, rank, dist, n_thr=n_thr, use_graph=not args.mcts_eager,
                         persistent=False if args.mcts_eager else None, steps=1, warmup_steps=0)
            outs[name] = {k: r[k] for k in keys if k in r}
            outs[name]["config"] = outs[name]["config"].replace("configs[2]", what)
            outs[name]["sample"] = "one batch of whole games (games_per_sec), a fresh position table"
            if "kernel_roofline" in outs[name]:   # (same keys as the headline's `roofline`; measured in this run)
                outs[name]["roofline"] = dict(outs[name].pop("kernel_roofline"), traffic=None)
                outs[name]["roofline"].update(variant_profile("mcts400" if sims >= 400 else "mctsnthr1"))
                if r.get("persistent"):
                    outs[name]["totals"] = r["persistent"]["totals"]
        n_open = 4 if sims >= 400 else 8
        if turns_arg != 0:
            n_open = n_open if turns_arg < 0 else turns_arg
            r = mcts_leg(args.mcts_games, sims, n_open, False, world, rank, dist, n_thr=n_thr, use_graph=not args.mcts_eager,
                         persistent=False if args.mcts_eager else None, steps=1, warmup_steps=0)
            o = {k: r[k] for k in keys if k in r and k != "kernel_roofline"}
            o["config"] = o["config"].replace("configs[2]", what)
            o["sample"] = ("the first %d turns of the games only (NOT a game rate: the openings of games from one start position "
                           "repeat each other's positions)" % n_open)
            outs[name + "_opening"] = o
        return outs

    more = {}
    if extras and args.mcts400_turns != 0:
        # BASELINE configs[3] at one GPU's share: the reference's 10 s budget per move (MCTS.py:80,139-147) as 400 playouts
        more.update(variant("mcts400", 400, 15, args.mcts400_turns, "configs[3] (one GPU's share)"))
    if extras and args.nthr1_turns != 0:
        # "also report n_thr = 1" (MCTS.py:80,109): every leaf expands at its second visit --
        # the policy net inside every playout
        more.update(variant("mcts_nthr1", args.mcts_sims, 1, args.nthr1_turns, "configs[2] with n_thr = 1"))
    saturated = None
    if extras and full and world == 1 and not args.no_saturated and not args.mcts_eager and not args.mcts_per_playout:
        # what the same kernel sustains OUTSIDE the latency-bound regime BASELINE's batch of 1024 is: one batch each of
        # 2048 and 4096 whole games per launch (the ABI's limit: the game workgroups may take half the device)
        saturated = {}
        for n_g in (2048, 4096):
            log("mcts_saturated: one batch of %d whole games" % n_g)
            failed = None
            try:
                r = mcts_leg(n_g, args.mcts_sims, 0, True, world, rank, None, steps=1, warmup_steps=0)
            except Exception as e:      # (a side leg never takes the line down: the same batch on the single launch)
                failed = "%s: %s" % (type(e).__name__, str(e)[:300])
                log("mcts_saturated: %s -- again as one launch" % failed)
                torch.cuda.synchronize()
                keep = os.environ.get("IAGO_SEARCH_SPLIT")
                os.environ["IAGO_SEARCH_SPLIT"] = "0"
                try:
                    r = mcts_leg(n_g, args.mcts_sims, 0, True, world, rank, None, steps=1, warmup_steps=0)
                finally:
                    if keep is None:
                        del os.environ["IAGO_SEARCH_SPLIT"]
                    else:
                        os.environ["IAGO_SEARCH_SPLIT"] = keep
            kr2 = r.get("kernel_roofline") or {}
            saturated[str(n_g)] = {"games_per_launch": n_g, "games_per_sec": r.get("games_per_sec"),
                                   "leaf_evals_per_sec": r["leaf_evals_per_sec"], "seconds": r["seconds"],
                                   "kernel_ms": kr2.get("kernel_ms"), "frac": kr2.get("frac"),
                                   "executed_frac": kr2.get("executed_frac"),
                                   "game_workgroups": (r.get("persistent") or {}).get("game_workgroups"),
                                   "role_split_game_cus": (r.get("persistent") or {}).get("role_split_game_cus"),
                                   "net_workgroups": (r.get("persistent") or {}).get("net_workgroups"),
                                   "batches_replayed_turn_by_turn": r.get("batches_replayed_turn_by_turn")}
            if failed:
                saturated[str(n_g)]["role_split_failed"] = failed
        saturated["note"] = ("one batch each, whole games, same nets and constants as the headline; the headline's 1024 games per "
                             "launch is BASELINE configs[2]'s batch.  Beyond 32 game workgroups the engine splits the search by "
                             "role (role_split_game_cus: two game workgroups per CU on that many CUs, one net workgroup on each "
                             "of the others)")
    train = None
    if args.train_iters > 0 and not args.mcts_only:
        log("configs[4] in miniature: REINFORCE sets")
        train = reinforce_leg(args.train_iters, world, rank, dist)
    b1 = None
    if rank == 0 and not args.mcts_only:
        log("one game (serving mode)")
        b1 = mcts_b1_leg()

    if rank == 0:
        kr = mcts.get("kernel_roofline")
        prof = (mcts.get("roofline") or {}).get("kernels") or {}
        sk = prof.get("search_kernel") or {}
        if kr is not None:
            roof = dict(kr)
            # HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE KiB, MI355X_MICROARCH.md) and rocprofv3's average
            # duration come from the committed profile of this command -- quoted only when it was taken on these
            # kernel sources
            roof["traffic"] = sk.get("hbm_bytes_per_launch") if prof.get("current") else None
            roof["profile"] = prof.get("profile")
            roof["profile_current"] = bool(prof.get("current"))
            roof["rocprof_kernel_avg_ms"] = ((sk.get("rocprof_avg_us") or 0.0) / 1e3 or None) if prof.get("current") else None
            if prof.get("current") and sk.get("frac") is not None:
                roof["executed_frac_pmc"] = sk["frac"]     # SQ_INSTS_MFMA x 32,768 / duration / peak, committed profile
        else:
            r0 = mcts["roofline"]
            roof = {"bound": r0["bound"], "achieved": r0.get("useful_tflops", r0["achieved"]), "peak": r0["peak"],
                    "unit": r0["unit"], "frac": r0.get("useful_frac_f16_peak", r0["frac"]), "traffic": None,
                    "kernel": "per-playout launches: loop level (no single dominant launch is timed here)",
                    "executed_tflops": r0["achieved"], "executed_frac": r0["frac"]}
        games_step = world * args.mcts_games
        line = {
            "metric": "self-play games/sec", "value": games_step * K / mcts["seconds"], "unit": "games/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": mcts["seconds"] / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 MFMA operands (float32 split in 2 / 3 pieces), f32 accumulate; u64 bitboards; f32 / f64 tree statistics",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: PV-MCTS self-play, %d games per GPU x %d playouts per move, %s, SLPolicy + "
                                   "Value random init (seed 0), lmbda=0.5 c_puct=1 n_thr=15, both colours search; one step = one "
                                   "batch of games%s" % (args.mcts_games, args.mcts_sims,
                                                         "whole games" if full else "FIRST %d TURNS ONLY (rehearsal)" % args.mcts_turns,
                                                         " = ONE launch of the persistent search" if mcts["persistent"] else ""),
                       "games_per_gpu": args.mcts_games, "sims_per_move": args.mcts_sims, "games_per_step": games_step,
                       "full_games": full, "n_thr": 15, "lmbda": 0.5, "c_puct": 1.0,
                       "position_table": "empty at the start of every step" if not args.keep_table else "kept across the steps",
                       "tuple_allgather": ("%s all-gather of every batch's (s, pi, z) tuples inside its step"
                                           % mcts["ranks"]["backend"].replace("nccl", "rccl")) if "ranks" in mcts else "none"},
            "step_ms_min": mcts["step_ms_min"], "step_ms_median": mcts["step_ms_median"], "step_ms_max": mcts["step_ms_max"],
            "timed_region_s": mcts["seconds"],
            "leaf_evals_per_sec": mcts["leaf_evals_per_sec"], "mcts_games_per_sec": mcts.get("games_per_sec"),
            "leaf_evals_per_step": mcts["leaf_evals"] / K,
            "roofline": roof,
        }
        if mcts["persistent"]:
            line["table_hits"] = mcts["persistent"]["position_table"]
        if "ranks" in mcts:
            line.update({"ranks_seen": mcts["ranks"]["ranks_seen"],
                         "rank_play_seconds_min": mcts["ranks"]["rank_play_seconds_min"],
                         "rank_play_seconds_max": mcts["ranks"]["rank_play_seconds_max"],
                         "gather_ms": mcts["ranks"]["gather_ms_per_step_max"],
                         "per_rank_games_per_sec_min": min(mcts["ranks"]["per_rank_games_per_sec"] or [0.0]),
                         "per_rank_games_per_sec_max": max(mcts["ranks"]["per_rank_games_per_sec"] or [0.0]),
                         "preflight": pre,
                         "ranks_played_different_games": mcts["ranks"]["ranks_played_different_games"]})
        if "per_playout_launches" in mcts:
            line["leaf_evals_per_sec_per_playout_launches"] = mcts["per_playout_launches"]["leaf_evals_per_sec"]
        rl = mcts.get("roofline") or {}
        if "useful_tflops" in rl:     # loop level (whole timed region, all ranks): measured in this run
            line["mcts_useful_tflops"] = rl["useful_tflops"]
            line["mcts_useful_frac_f16_peak"] = rl["useful_frac_f16_peak"]
            line["mcts_executed_frac_f16_peak"] = rl["frac"]
        if saturated is not None:
            line["mcts_saturated"] = saturated
        line["value_spread"] = value_spread(args.mcts_games, args.mcts_sims, line["value"])
        line["mcts"] = mcts
        for k, v in more.items():
            line[k] = v
        if "mcts400" in more:
            line["leaf_evals_per_sec_400"] = more["mcts400"]["leaf_evals_per_sec"]
            line["games_per_sec_400"] = more["mcts400"].get("games_per_sec")
        elif "mcts400_opening" in more:
            line["leaf_evals_per_sec_400_opening"] = more["mcts400_opening"]["leaf_evals_per_sec"]
        if "mcts_nthr1" in more:
            line["leaf_evals_per_sec_nthr1"] = more["mcts_nthr1"]["leaf_evals_per_sec"]
        elif "mcts_nthr1_opening" in more:
            line["leaf_evals_per_sec_nthr1_opening"] = more["mcts_nthr1_opening"]["leaf_evals_per_sec"]
        if roll is not None:
            line["rollout"] = roll
            line["rollout_games_per_sec"] = roll["value"]
        if train is not None:
            line["reinforce"] = train
            line["reinforce_iters_per_sec"] = train["iters_per_sec"]
            if "weak_scaling_set" in train:
                line["reinforce_weak_games_per_sec"] = train["weak_scaling_set"]["games_per_sec"]
            line["reinforce_miopen_find_db"] = train["miopen_find_db"]
            line["reinforce_update_ms"] = train["update_ms"]
            line["reinforce_miopen_solvers"] = train["miopen_solvers"]
            if "mcts_fed" in train:
                line["reinforce_mcts_fed_rounds_per_sec"] = train["mcts_fed"]["rounds_per_sec"]
        if b1 is not None:
            line["mcts_single_game"] = b1
        if not args.no_cpu_baseline and world == 1:  # the CPU baselines are N = 1 figures
            log("CPU baselines")
            per_game = mcts["leaf_evals"] / (games_step * K) if full else None
            line["cpu_baseline"] = mcts_cpu_baseline(leaf_evals_per_game=per_game)
            if roll is not None:
                roll["cpu_baseline"] = cpu_baseline(w, b)
                roll["cpu_baseline"]["python_loops_one_core"] = python_loop_baseline(w, b)
            if not args.mcts_only:
                sl = sl_game_cpu_baseline()
                line["configs0"] = dict(sl, config="BASELINE configs[0]: one SLPolicy-vs-SLPolicy game, CPU only (plumbing; "
                                                   "src/rl_self_play.py:27-31 semantics, self_play.py itself is broken)")
                if train is not None:
                    train["cpu_baseline"] = reinforce_cpu_baseline(sl)
        log("done")
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


T_START = time.perf_counter()

if __name__ == "__main__":
    main()
