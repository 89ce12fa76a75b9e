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
