#!/bin/bash
# rocprofv3 kernel stats of the split-f16 REINFORCE update (tools/time_policy_grad.py); run on the GPU box:
#   bash tools/profile_grad.sh <name>   -> gpurun_out/prof_<name>_grad/kernel_stats.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${1:-x}_grad
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/tools/time_policy_grad.py 2048 > "$OUT/trace.log" 2>&1
python3 - "$OUT" <<'EOF'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(sys.argv[1] + "/kernel_stats.txt", "w") as o:
    for r in rows[:24]:
        line = "%-90s %6s calls %9.1f us avg %6s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"])
        print(line)
        o.write(line + "\n")
EOF
# counters of the update's kernels, one pass per group (no trace domains beside --pmc)
for C in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" FETCH_SIZE WRITE_SIZE; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 $REPO/tools/time_policy_grad.py 2048 > "$OUT/pmc_$N.log" 2>&1 || echo "pass $N failed"
done
python3 - "$OUT" <<'EOF'
import csv, glob, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if any(s in k for s in ("wgrad_split", "conv3x3_bwd_data", "conv3x3_split_kernel", "head_grad", "split_scaled", "stem_wgrad")):
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/pmc_summary.txt", "w") as o:
    for k, c in sorted(vals.items()):
        per = {name: sum(v) / len(v) for name, v in c.items()}          # mean per launch
        line = ("%-26s per launch: MFMA %.3g, VALU %.3g, LDS %.3g instructions; LDS bank-conflict / LDS-active cycles %.3f; "
                "from beyond L2 %.1f MB read (2 x FETCH_SIZE KiB: MI355X_MICROARCH.md), %.1f MB written") % (
            k[:26], per.get("SQ_INSTS_MFMA", 0), per.get("SQ_INSTS_VALU", 0), per.get("SQ_INSTS_LDS", 0),
            per.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, per.get("SQ_LDS_IDX_ACTIVE", 1)),
            2 * per.get("FETCH_SIZE", 0) * 1024 / 1e6, per.get("WRITE_SIZE", 0) * 1024 / 1e6)
        print(line)
        o.write(line + "\n")
EOF
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
