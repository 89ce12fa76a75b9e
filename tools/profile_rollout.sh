#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench's
# headline leg, with the driver's own arguments.
# Usage: tools/profile_rollout.sh <tag> [extra bench args]    -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r02}
shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# --rollout-only = the timed leg exactly as in the full run, the other legs (which launch the
# same kernel at other sizes / under overlap and would pollute its average) skipped
ARGS="--gpus 1 --steps 20 --warmup 5 --rollout-only $*"
echo "python3 bench.py $ARGS" > "$OUT/command.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py $ARGS > "$OUT/trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 $REPO/bench.py $ARGS --repeats 4 > "$OUT/pmc_$N.log" 2>&1
done
# keep the merge-back small: drop per-dispatch traces, keep stats + counters
find "$OUT" -name "*_kernel_trace.csv" -size +4M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
