#!/bin/bash
# rocprofv3 stats + PMC of ONE launch size of the rollout (default 1M boards, lane-per-board
# kernel): gpurun_out/prof_<tag>/ in the layout tools/summarize_profile.py reads.
# Usage: tools/profile_large.sh <tag> [boards] [launches]
set -u
TAG=${1:-r02large}
BOARDS=${2:-1048576}
REPS=${3:-10}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "python3 tools/run_large.py $BOARDS $REPS" > "$OUT/command.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/tools/run_large.py $BOARDS $REPS > "$OUT/trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 $REPO/tools/run_large.py $BOARDS $REPS > "$OUT/pmc_$N.log" 2>&1
done
find "$OUT" -name "*_kernel_trace.csv" -size +4M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
