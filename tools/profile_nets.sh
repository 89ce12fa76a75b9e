#!/bin/bash
# per-kernel time of one net forward:  tools/profile_nets.sh value 1024
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/nets_$1_$2
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/tools/exp_net_kernels.py $1 $2 > "$OUT/trace.log" 2>&1
tail -1 "$OUT/trace.log" | head -1; grep "ms per forward" "$OUT/trace.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for path in glob.glob(sys.argv[1] + "/trace/*/*_kernel_stats.csv"):
    rows = list(csv.reader(open(path)))
    for r in rows[1:]:
        if int(r[1]) >= 200:
            print("%-64s calls %5s avg %8.1f us  per fwd %7.1f us" % (r[0][:64], r[1], float(r[3]) / 1e3, float(r[2]) / 203e3))
PY
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete
