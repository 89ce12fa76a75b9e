#!/bin/bash
# rocprofv3 kernel stats only (no PMC) of the PV-MCTS leg, eager launches, first N turns:
#   tools/prof_quick.sh <tag> [turns] -> gpurun_out/pq_<tag>/stats.csv (top kernels printed)
set -u
TAG=${1:-q}
TURNS=${2:-4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --large-boards 0 --train-iters 0 --mcts-turns $TURNS --mcts-eager --mcts-only --nthr1-turns 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py $ARGS > "$OUT/trace.log" 2>&1
cp "$OUT"/trace/*/*_kernel_stats.csv "$OUT/stats.csv"
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete
python3 - "$OUT/stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:14]:
    print("%-70s calls %7s total_ms %9.2f avg_us %8.2f" % (r[0][:70], r[1], float(r[2]) / 1e6, float(r[3]) / 1e3))
PY
tail -c 400 "$OUT/trace.log" | head -c 400
