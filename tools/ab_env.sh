#!/bin/bash
# A/B of environment settings (and libraries) on ONE box: tools/ab_env.sh "VAR=v ..." "VAR=v ..." ...  (each twice, alternating)
for rep in 1 2; do
  for SET in "$@"; do
    env $SET python bench.py --steps 3 --warmup 1 --mcts-only --no-cpu-baseline > gpurun_out/abenv.json 2> gpurun_out/abenv.err
    python - <<PY
import json
d=json.load(open("gpurun_out/abenv.json"))
m=d["mcts"]
print("$SET rep $rep: %.2f M leaf-evals/s  value_evals %d ahead %d hits %d" % (d["leaf_evals_per_sec"]/1e6, m["value_evals"]/3, m["value_ahead"]/3, d["table_hits"]["hits"]/3))
PY
  done
done
