cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== soak, split 32"; IAGO_SEARCH_SPLIT=32 timeout -k 10 300 python tools/soak_persistent.py 30 > gpurun_out/soak_split.txt 2>&1; tail -4 gpurun_out/soak_split.txt
echo "== (24, 24) x 6"
for i in 1 2 3 4 5 6; do IAGO_SEARCH_SPLIT=24 IAGO_PERSISTENT_GPW=24 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --mcts-only --no-cpu-baseline --no-saturated > gpurun_out/s24.json 2> gpurun_out/s24_$i.err && python -c "
import json; d=json.load(open('gpurun_out/s24.json')); print('ok %.2f M' % (d['leaf_evals_per_sec']/1e6), d['step_ms_min'], d['step_ms_max'])" || tail -3 gpurun_out/s24_$i.err; done
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=45 > gpurun_out/gputests.txt 2>&1; tail -60 gpurun_out/gputests.txt
