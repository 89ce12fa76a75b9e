#!/bin/bash
# rocprofv3 kernel stats + PMC of the PV-MCTS leg (bounded sample: the first 4 turns of
# 1024 games x 100 playouts, eager launches -- rocprofv3 does not attribute kernels
# launched from a hipGraph): gpurun_out/prof_<tag>_mcts/
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_mcts
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --large-boards 0 --train-iters 0 --mcts-turns 4 --mcts-eager --mcts-only"
echo "python3 bench.py $ARGS" > "$OUT/command.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py $ARGS > "$OUT/trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 $REPO/bench.py $ARGS > "$OUT/pmc_$N.log" 2>&1
done
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*.db" -delete
python3 $REPO/tools/summarize_mcts_profile.py "$OUT" > "$OUT/summary.json"
du -sh "$OUT"
