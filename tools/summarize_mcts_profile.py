#!/usr/bin/env python3
"""Per-kernel means of the PMC passes of tools/profile_mcts.sh (one JSON object on stdout):
for every kernel of the PV-MCTS playout the launches seen, the mean of every counter, the
rocprofv3 average duration, and HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB
(MI355X_MICROARCH.md: gfx950 reports half the bytes of 16 B/lane reads)."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
KERNELS = ["search_kernel", "value_rollout_kernel", "policy_resident_kernel", "mix_backup_path_kernel", "trunk_resident_kernel", "conv3x3_split_trunk_kernel", "descend_kernel", "fresh_leaves_kernel", "select_kernel",
           "mix_backup_lookahead_kernel",
           "expand_cached_kernel", "store_priors_kernel", "mix_backup_kernel", "expand_kernel",
           "pending_kernel", "value_stem_kernel", "value_head_kernel", "conv3x3_f32_counted_kernel", "conv3x3_f32_kernel",
           "stem_f32_kernel", "policy_head_kernel", "rollout_row_kernel", "rollout_kernel", "encode_planes_kernel",
           "best_move_kernel", "advance_root_kernel"]
out = collections.OrderedDict((k, collections.OrderedDict()) for k in KERNELS)
for path in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    for r in list(csv.reader(open(path)))[1:]:
        for k in KERNELS:
            if "::" + k + "(" in r[0].replace("<", "(") or r[0].startswith(k + "("):
                out[k]["calls"] = out[k].get("calls", 0) + int(r[1])
                out[k]["total_ns"] = out[k].get("total_ns", 0) + int(float(r[2]))
                out[k]["min_us"] = min(out[k].get("min_us", 1e30), float(r[5]) / 1e3)
                out[k]["max_us"] = max(out[k].get("max_us", 0.0), float(r[6]) / 1e3)
for k in KERNELS:
    if "calls" in out[k]:
        out[k]["avg_us"] = out[k]["total_ns"] / out[k]["calls"] / 1e3
for path in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        for k in KERNELS:
            if "::" + k + "(" in r["Kernel_Name"].replace("<", "(") or r["Kernel_Name"].startswith(k + "("):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            out[k][c] = sum(v) / len(v)
            out[k].setdefault("pmc_launches", len(v))
for k in KERNELS:
    if "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        out[k]["hbm_bytes_per_launch"] = (2.0 * out[k]["FETCH_SIZE"] + out[k]["WRITE_SIZE"]) * 1024.0
        if "avg_us" in out[k]:
            out[k]["hbm_gb_per_s"] = out[k]["hbm_bytes_per_launch"] / out[k]["avg_us"] / 1e3
# L2 side of the one-board net kernels: requests a launch's CUs send to L2
# (TCP_TCC_READ_REQ: 64-byte requests -- MI355X_MICROARCH.md tallies a 128-byte line as one request of 64 B
# on gfx950, so bytes = requests x 128 for 16 B/lane streams; both figures are given), L2 hit rate, and
# the rate per busy CU: bytes / (workgroups of the launch x its duration)
for k in KERNELS:
    d = out[k]
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "TCP_TCC_READ_REQ_sum" in d and "avg_us" in d:
        d["l2_read_bytes_per_launch_64B_req"] = d["TCP_TCC_READ_REQ_sum"] * 64.0
        d["l2_read_bytes_per_launch_128B_req"] = d["TCP_TCC_READ_REQ_sum"] * 128.0
        if "SQ_WAVES" in d and d["SQ_WAVES"] > 0:
            wgs = d["SQ_WAVES"] / 4.0          # 256-thread workgroups, one per CU
            d["l2_read_gb_per_s_per_workgroup_64B_req"] = d["l2_read_bytes_per_launch_64B_req"] / wgs / d["avg_us"] / 1e3
            d["l2_read_gb_per_s_per_workgroup_128B_req"] = d["l2_read_bytes_per_launch_128B_req"] / wgs / d["avg_us"] / 1e3
if "calls" in out["search_kernel"]:
    out["search_kernel"]["note"] = ("one launch = a whole batch of games (1024 games x 100 playouts per move, played to the end); "
                                    "every launch of the profiled command -- its warm-up batch and its timed ones -- is such a "
                                    "batch: the counters and avg_us are means over all of them")
cmd = os.path.join(src, "command.txt")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (csrc_sha16: the kernel sources this profile was taken on)
print(json.dumps({"command": open(cmd).read().strip() if os.path.exists(cmd) else None,
                  "csrc_sha16": bench.csrc_sha16(),
                  "mfma_shape": "16x16x32",   # the walks' K loops (round 5): SQ_INSTS_MFMA x 16,384 FLOP
                  "kernels": {k: v for k, v in out.items() if v}}, indent=1))
