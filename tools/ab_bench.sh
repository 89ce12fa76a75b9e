#!/bin/bash
# A/B of library builds on ONE box (boxes differ by several per cent): tools/ab_bench.sh <name> libA.so libB.so ... [-- extra bench args]
# -> gpurun_out/<name>_<i>.json per library, each run twice, alternating
NAME=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for rep in 1 2; do
  i=0
  for L in "${LIBS[@]}"; do
    IAGO_HIP_LIB=$L python bench.py --steps 3 --warmup 1 --mcts-only --no-cpu-baseline "$@" > gpurun_out/${NAME}_${i}_r${rep}.json 2> gpurun_out/${NAME}_${i}_r${rep}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/${NAME}_${i}_r${rep}.json"))
t=d["mcts"]["persistent"]["totals"]
print("$L", "rep", $rep, "%.3f s/batch  %.2f M leaf-evals/s  busy %.3f" % (d["ms_per_step"]/1e3, d["leaf_evals_per_sec"]/1e6, t[5]/(t[4]+t[5])))
PY
    i=$((i+1))
  done
done
