#!/bin/bash
# rocprofv3 kernel stats of the PV-MCTS leg (bounded sample): gpurun_out/prof_<tag>_mcts/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_mcts
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --large-boards 0 --train-iters 0 --mcts-turns 4 --mcts-eager > "$OUT/trace.log" 2>&1
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
