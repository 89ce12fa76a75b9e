#!/bin/bash
# counters of selfplay_policy_kernel (the 64 games of a REINFORCE set as one launch) under tools/dev_step_breakdown.py:
#   bash tools/profile_selfplay.sh <name>  -> gpurun_out/prof_<name>_selfplay/summary.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${1:-x}_selfplay
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/tools/dev_step_breakdown.py > "$OUT/trace.log" 2>&1
for C in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" FETCH_SIZE WRITE_SIZE "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 $REPO/tools/dev_step_breakdown.py > "$OUT/pmc_$N.log" 2>&1 || echo "pass $N failed"
done
python3 - "$OUT" <<'EOF'
import csv, glob, sys, collections
vals = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "selfplay_policy_kernel" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg_us = None
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "selfplay_policy_kernel" in r["Name"]:
            avg_us = float(r["AverageNs"]) / 1e3
            calls = int(r["Calls"])
per = {k: sum(v) / len(v) for k, v in vals.items()}
wgs = per.get("SQ_WAVES", 256.0) / 4.0
lines = ["selfplay_policy_kernel: %d launches, %.1f us on average, %d workgroups (one game each, one per CU)" % (calls, avg_us, wgs),
         "per launch: MFMA %.4g instructions (x 16,384 FLOP = %.0f TFLOP/s on %d CUs = %.3f of those CUs' share of 2.5 PFLOP/s), VALU %.4g, LDS %.4g; "
         "LDS bank-conflict / LDS-active cycles %.3f" % (per.get("SQ_INSTS_MFMA", 0), per.get("SQ_INSTS_MFMA", 0) * 16384 / avg_us / 1e6, wgs,
                                                        per.get("SQ_INSTS_MFMA", 0) * 16384 / avg_us / 1e6 / (2500.0 * wgs / 256.0),
                                                        per.get("SQ_INSTS_VALU", 0), per.get("SQ_INSTS_LDS", 0),
                                                        per.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, per.get("SQ_LDS_IDX_ACTIVE", 1))),
         "L2: %.4g read requests from the CUs = %.1f MB at 128 B (%.1f GB/s per workgroup), hit rate %.3f; from beyond L2 %.1f MB read (2 x FETCH_SIZE KiB), %.1f MB written" % (
             per.get("TCP_TCC_READ_REQ_sum", 0), per.get("TCP_TCC_READ_REQ_sum", 0) * 128 / 1e6,
             per.get("TCP_TCC_READ_REQ_sum", 0) * 128 / wgs / avg_us / 1e3,
             per.get("TCC_HIT_sum", 0) / max(1.0, per.get("TCC_HIT_sum", 0) + per.get("TCC_MISS_sum", 0)),
             2 * per.get("FETCH_SIZE", 0) * 1024 / 1e6, per.get("WRITE_SIZE", 0) * 1024 / 1e6)]
open(sys.argv[1] + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
EOF
