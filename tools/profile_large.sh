#!/bin/bash
# PMC counters of the 1M-board launch (chip full): gpurun_out/prof_<tag>_large/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_large
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --mcts-turns 0 --train-iters 0"
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- $BENCH > "$OUT/pmc_$N.log" 2>&1
done
find "$OUT" -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
for path in sorted(glob.glob("$OUT/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "rollout" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1000000:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, len(v), sum(v) / len(v))
PY
