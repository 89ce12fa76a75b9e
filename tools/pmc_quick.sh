#!/bin/bash
# One rocprofv3 --pmc pass of the bench's headline leg, kernel means printed:
#   tools/pmc_quick.sh <tag> "<counters>" [kernel substring] [extra bench args]
set -u
TAG=${1:-q}; C=${2:-"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU"}
K=${3:-rollout}; shift; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $C --output-format csv -d "$OUT/p" -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --rollout-only --repeats 4 "$@" > "$OUT/p.log" 2>&1
python3 - "$OUT" "$K" <<'PY'
import csv, glob, sys, collections
for path in sorted(glob.glob(sys.argv[1] + "/p/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(k, round(sum(v) / len(v), 1), len(v))
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*.csv" -size +2M -delete
