#!/bin/bash
# PMC counters of the split-f16 convolution kernel:  tools/profile_conv.sh <tag>
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/conv_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/exp_conv_layer.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/p$i" -- $CMD > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for path in sorted(glob.glob(sys.argv[1] + "/trace/*/*_kernel_stats.csv")):
    for r in list(csv.reader(open(path)))[:4]:
        print(r[0][:60], r[1:4])
for path in sorted(glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "conv3x3" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, round(sum(v) / len(v), 1), len(v))
PY
find "$OUT" -name "*.db" -delete
