#!/bin/bash
# Extra SQ counters of the rollout kernel (issue / fetch side):  tools/profile_pmc_extra.sh <tag>
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcx_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 512 --warmup 256 --no-cpu-baseline --mcts-turns 0 --large-boards 0 --train-iters 0"
i=0
for C in "SQ_INST_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VSKIPPED SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/p$i" -- $BENCH > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for path in sorted(glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "rollout" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, round(sum(v) / len(v), 1), len(v))
PY
find "$OUT" -name "*.db" -delete
