#!/usr/bin/env python3
"""Condense a tools/profile_rollout.sh output directory into the files kept
under profiles/: the rocprofv3 kernel-stats CSV, a PMC summary (mean per launch
of the named kernel) and the HBM traffic figure bench.py reports.

    python tools/summarize_profile.py gpurun_out/prof_r02a r02a [kernel-substring]

The directory's `command.txt` (written by tools/profile_rollout.sh) is recorded as the source
of the numbers; profiles/rollout_traffic.json is keyed by the launch size (boards per launch)
the counters were taken at, so that bench.py never pairs a figure with another launch size.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
kern = sys.argv[3] if len(sys.argv) > 3 else "rollout_lpb_kernel"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
rocprof_avg_ns = None
if stats:
    rows = list(csv.reader(open(stats[0])))
    for r in rows[1:]:
        if kern in r[0]:
            rocprof_avg_ns = float(r[3])
    with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:120]
            w.writerow(r)

pmc = collections.OrderedDict()
for path in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                      "SGPR_Count")}
    for k, v in agg.items():
        pmc[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    if meta:
        pmc["_dispatch"] = meta
cmd_path = os.path.join(src, "command.txt")
command = open(cmd_path).read().strip() if os.path.exists(cmd_path) else "unknown"
summary = {"kernel": kern, "source": "rocprofv3 --kernel-trace --stats, then one --pmc pass per counter "
           "group, of: " + command, "counters": pmc}
if rocprof_avg_ns is not None:
    summary["rocprof_kernel_avg_ms"] = rocprof_avg_ns / 1e6
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fetch_kb, write_kb = pmc["FETCH_SIZE"]["mean"], pmc["WRITE_SIZE"]["mean"]
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
    # reports half the bytes of a 16 B/lane coalesced read stream -> doubled;
    # WRITE_SIZE is exact.
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    summary["hbm_bytes_per_launch"] = hbm
    summary["traffic_note"] = ("(2*FETCH_SIZE + WRITE_SIZE) KiB per launch; FETCH_SIZE doubled per "
                               "the gfx950 correction for 16 B/lane reads (the table staging is "
                               "float4 loads; board loads are 8 B/lane and uncalibrated)")
    extra = {}
    if rocprof_avg_ns is not None:
        extra["rocprof_kernel_avg_ms"] = rocprof_avg_ns / 1e6
    if "SQ_INSTS_VALU" in pmc and "Grid_Size" in pmc.get("_dispatch", {}):
        # wave-level VALU instructions of one launch and the boards it played (8 lanes each)
        lanes_per_board = 1 if "lpb" in kern else (16 if "row" in kern else 8)
        extra.update({"valu_insts_per_launch": pmc["SQ_INSTS_VALU"]["mean"],
                      "boards_per_launch": int(pmc["_dispatch"]["Grid_Size"]) // lanes_per_board,
                      "kernel": kern})
    if "boards_per_launch" in extra:
        tpath = os.path.join(out, "rollout_traffic.json")
        table = {}
        if os.path.exists(tpath):
            table = json.load(open(tpath))
            if "by_boards_per_launch" not in table:
                table = {}
        table.setdefault("by_boards_per_launch", {})[str(extra["boards_per_launch"])] = dict(
            {"hbm_bytes_per_launch": hbm, "fetch_size_kib": fetch_kb, "write_size_kib": write_kb,
             "profile": tag, "command": command, "note": summary["traffic_note"]}, **extra)
        with open(tpath, "w") as f:
            json.dump(table, f, indent=1)
with open(os.path.join(out, "%s_pmc_summary.json" % tag), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1)[:3000])
