#!/bin/bash
# A/B of environment settings (and libraries: IAGO_HIP_LIB=...) on ONE box: tools/ab_env.sh "VAR=v ..." "VAR=v ..." ...
# (each twice, alternating; AB_ARGS = extra bench arguments, e.g. "--mcts-sims 400").  Per run: leaf-evals/s, the game
# workgroups' iteration (totals[7] / totals[2], 100 MHz ticks), the net workgroups' busy share (totals[5] / ([4] + [5])),
# walking time per value-equivalent evaluation (a policy walk counted as 1.9 value boards), values walked ahead, table hits
for rep in 1 2; do
  for SET in "$@"; do
    env $SET python bench.py --steps 3 --warmup 1 --mcts-only --no-cpu-baseline $AB_ARGS > gpurun_out/abenv.json 2> gpurun_out/abenv.err || { tail -5 gpurun_out/abenv.err; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/abenv.json"))
m=d["mcts"]; t=m["persistent"]["totals"]; k=d["steps"]
ve=(t[0]+t[11]+1.9*t[1])
print("$SET rep $rep: %.2f M leaf-evals/s  %.1f ms/batch  game iteration %.1f us  net busy %.3f  walk %.1f us per value-equivalent  ahead %d hits %d  net wgs %d"
      % (d["leaf_evals_per_sec"]/1e6, d["ms_per_step"], t[7]/max(t[2],1)*0.01, t[5]/max(t[4]+t[5],1), t[5]*0.01/max(ve,1), t[11]/k, t[8]/k, m["persistent"]["net_workgroups"]))
PY
  done
done
