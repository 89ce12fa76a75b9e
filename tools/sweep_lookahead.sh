#!/bin/bash
# PV-MCTS leg of bench.py under the look-ahead knobs (K playouts per policy batch, j playouts the batch
# runs beside, launches per batch): leaf-evals/s per setting, two runs each, on one box.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for CFG in ${CONFIGS:-"4,2,2 4,3,2 3,2,2 5,2,2 5,3,2 6,3,2 4,2,3 4,2,1 4,2,2"}; do
  K=${CFG%%,*}; R=${CFG#*,}; J=${R%%,*}; P=${R#*,}
  for REP in 1 2; do
    IAGO_LOOKAHEAD=$K IAGO_LOOKAHEAD_OVERLAP=$J IAGO_POLICY_PARTS=$P timeout -k 10 200 python $REPO/bench.py --mcts-only --no-cpu-baseline --nthr1-turns 0 --train-iters 0 --large-boards 0 2>gpurun_out/sweep_err.log \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$K j=$J parts=$P: %.3f M leaf-evals/s, policy evals %d' % (d['leaf_evals_per_sec']/1e6, d['mcts']['policy_evals']))"
  done
done
