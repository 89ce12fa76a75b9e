#!/bin/bash
# For several stream counts: bench.py's event-pair kernel duration vs rocprofv3's average.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for S in 6 8 12 16 24; do
  rm -rf /tmp/pa; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python3 $REPO/bench.py --streams $S --no-cpu-baseline --mcts-turns 0 --large-boards 0 --train-iters 0 > /tmp/pa.log 2>&1
  python3 - <<PY
import json,csv,glob
d=json.loads([l for l in open('/tmp/pa.log') if l.startswith('{"metric"')][-1]); r=d['roofline']
rows=list(csv.DictReader(open(glob.glob('/tmp/pa/*/*_kernel_stats.csv')[0])))
avg=[float(x['AverageNs']) for x in rows if 'rollout' in x['Name']][0]
print('S=$S value %.0fM  events %.1f us  rocprof %.1f us  in-flight %.1f' % (d['value']/1e6, r['kernel_ms']*1e3, avg/1e3, r['launches_in_flight']))
PY
done
