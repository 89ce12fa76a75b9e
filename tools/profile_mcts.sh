#!/bin/bash
# rocprofv3 kernel stats + PMC of the PV-MCTS leg with eager launches (rocprofv3 does not
# attribute kernels launched from a hipGraph).
#   tools/profile_mcts.sh <tag>        bounded sample: the first 4 turns of 1024 games x 100 playouts
#   tools/profile_mcts.sh <tag> full   the games played to the end (>= 6,000 launches per playout kernel):
#                                      what bench.py's mcts.roofline.kernels reads (*_mcts_fullgame_*)
# -> gpurun_out/prof_<tag>_mcts[_fullgame]/ ; summary.json = tools/summarize_mcts_profile.py
set -u
TAG=${1:-r03}
MODE=${2:-first4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
#   tools/profile_mcts.sh <tag> persistent   the persistent engine (what bench.py times since round 4): ONE search_kernel
#                                      launch per whole self-play game -> profiles/*_mcts_persistent_*
EAGER="--mcts-eager"
if [ "$MODE" = "full" ]; then TURNS=-1; SUF=_mcts_fullgame; elif [ "$MODE" = "persistent" ]; then TURNS=-1; SUF=_mcts_persistent; EAGER=""; else TURNS=4; SUF=_mcts; fi
SUF=${PROFILE_SUF:-$SUF}
OUT=$REPO/gpurun_out/prof_${TAG}${SUF}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (round 5: the headline of bench.py IS this leg -- K whole-game batches, one search_kernel launch each; 1 warm-up + 5 timed
# batches per pass: every launch of a pass is a whole batch, so rocprofv3's average is the average of whole batches)
# PROFILE_ARGS: extra bench arguments of the headline leg (round 6: "--mcts-sims 400" = one GPU's share of configs[3],
# "--mcts-nthr 1" = the n_thr = 1 variant), PROFILE_SUF: what the output directory and the committed summaries are called
# (e.g. _mcts400_persistent); PROFILE_STEPS: timed batches per pass
ARGS="--gpus 1 --steps ${PROFILE_STEPS:-5} --warmup 1 --no-cpu-baseline --mcts-turns $TURNS $EAGER --mcts-only ${PROFILE_ARGS:-}"
echo "python3 bench.py $ARGS" > "$OUT/command.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py $ARGS > "$OUT/trace.log" 2>&1
# one pass per counter group (MI355X_MICROARCH.md: PMC in runs of their own, no trace domains);
# the last three groups are the L2 / fabric side of the one-board walks
for C in FETCH_SIZE WRITE_SIZE \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 $REPO/bench.py $ARGS > "$OUT/pmc_$N.log" 2>&1 \
    || echo "pass $N failed (see pmc_$N.log)"
done
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*.db" -delete
python3 $REPO/tools/summarize_mcts_profile.py "$OUT" > "$OUT/summary.json"
# the counter CSVs of a full game on the per-playout launches are tens of MB: keep the summary and the stats
# (the persistent engine's are two rows per counter: they stay, so that the summary can be rebuilt from the merge-back)
if [ "$MODE" = "full" ]; then find "$OUT" -name "*_counter_collection.csv" -delete; fi
du -sh "$OUT"
