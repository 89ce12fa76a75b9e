#!/usr/bin/env python3
"""Lab tool: where a net workgroup's walk spends its time UNDER LOAD (VERDICT r04 task 3: "per-layer clock stamps
inside trunk_item<true, 2> / policy_item in the persistent kernel under load").

    python tools/build_search_variants.py search_walkstamps
    IAGO_HIP_LIB=tools/_build/search_walkstamps.so python tools/exp_walk_stamps.py [games] [playouts]

Plays bench.py's batch (1024 games x 100 playouts per move, whole games) on the stamped build and prints, per kind of
walk (value pair, value single, policy), the mean time per phase: block1, then per layer the K loop / the wait at the
barrier behind it (the workgroup's slowest wave) / the epilogue, then the head; the in-kernel shader clock
(s_memtime cycles / s_memrealtime ticks) and what the K loops' MFMAs need at that clock."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from iago_amd import _lib  # noqa: E402


def main():
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    sims = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    L = _lib.lib()
    L.iago_debug_walk_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 192)()
    # warm-up batch inside mcts_leg (warmup_steps=1), then clear before the timed ones: cleared here by reading after
    # the run and subtracting nothing -- the warm-up walks are in the sums too (same workload)
    try:
        out = bench.mcts_leg(games, sims, 0, True, 1, 0, None, steps=2, warmup_steps=1)
        res = {"leaf_evals_per_sec": out["leaf_evals_per_sec"], "seconds_per_batch": out["seconds"] / 2}
    except Exception as e:      # (the timing-only builds compute wrong numbers: a saturation flag may end the leg)
        res = {"error": str(e)[:200]}
    assert L.iago_debug_walk_stamps(buf, 1) == 0
    st = list(buf)
    for name, base, mfma_cycles in (("value_pair", 0, 234 * 48 * 16), ("value_single", 32, 234 * 24 * 16), ("policy", 64, 234 * 48 * 16)):
        n = st[base + 31]
        if not n:
            continue
        ph = [st[base + i] / n / 100.0 for i in range(23)]          # us
        total = st[base + 29] / n / 100.0
        clock = st[base + 30] / max(st[base + 29], 1) * 100.0        # MHz: cycles per 100 MHz tick
        k = [ph[1 + 3 * L] for L in range(7)]
        bar = [ph[2 + 3 * L] for L in range(7)]
        epi = [ph[3 + 3 * L] for L in range(7)]
        res[name] = {"walks": n, "us_per_walk": total, "clock_mhz": clock, "block1_us": ph[0],
                     "k_loops_us": sum(k), "barrier_wait_us": sum(bar), "epilogues_us": sum(epi), "head_us": ph[22],
                     "k_loop_by_layer_us": [round(x, 2) for x in k], "barrier_by_layer_us": [round(x, 2) for x in bar],
                     "epilogue_by_layer_us": [round(x, 2) for x in epi],
                     "mfma_issue_us_at_that_clock": mfma_cycles / clock}
        if base < 64:   # (the value walks carry finer stamps)
            res[name]["head_parts_us"] = dict(zip(("mfma", "barrier1", "tap_sums_barrier2", "fc10_barrier3", "sum_store"),
                                                  [round(st[base + 23 + i] / n / 100.0, 2) for i in range(5)]))
            res[name]["block1_until_weights_staged_us"] = round(st[base + 28] / n / 100.0, 2)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
