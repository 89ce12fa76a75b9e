#!/bin/bash
# Fresh-process stress of the role split (round 6: ONE launch of the setting "24 game CUs, 43 workgroups of 24 games" once
# gave up at its clock limit; cause not found): tools/stress_split.sh <runs> "VAR=v ..." ["VAR=v ..." ...]
# Every run = a new process playing 2 batches of 1024 whole games; prints ok / the error's tail per run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for SET in "$@"; do
  ok=0; bad=0
  for i in $(seq 1 $N); do
    if env $SET timeout -k 10 200 python bench.py --steps 1 --warmup 1 --mcts-only --no-cpu-baseline --no-saturated > gpurun_out/stress.json 2> gpurun_out/stress.err; then ok=$((ok+1)); else bad=$((bad+1)); echo "FAILED ($SET, run $i):"; tail -4 gpurun_out/stress.err; fi
  done
  echo "$SET: $ok ok, $bad failed of $N"
done
