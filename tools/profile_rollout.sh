#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# Usage: tools/profile_rollout.sh <tag>     -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 1024 --warmup 256 --no-cpu-baseline --mcts-turns 0 --large-boards 0 --train-iters 0"
# kernel trace of the same command the bench line comes from (default --steps / --warmup)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py --no-cpu-baseline --mcts-turns 0 --large-boards 0 --train-iters 0 > "$OUT/trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- $BENCH > "$OUT/pmc_$N.log" 2>&1
done
# keep the merge-back small: drop per-dispatch traces, keep stats + counters
find "$OUT" -name "*_kernel_trace.csv" -size +4M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
find "$OUT" -name "*.csv" | head -50
