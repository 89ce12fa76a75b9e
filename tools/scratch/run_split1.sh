set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
IAGO_SEARCH_SPLIT=16 timeout -k 10 400 python -m pytest tests/test_search_persistent_gpu.py -x -q > gpurun_out/split_tests.txt 2>&1
tail -5 gpurun_out/split_tests.txt
timeout -k 10 600 bash tools/ab_env.sh "IAGO_SEARCH_SPLIT=0" "IAGO_SEARCH_SPLIT=16" "IAGO_SEARCH_SPLIT=24" "IAGO_SEARCH_SPLIT=32" "IAGO_SEARCH_SPLIT=32 IAGO_PERSISTENT_GPW=16" "IAGO_SEARCH_SPLIT=24 IAGO_PERSISTENT_GPW=24" > gpurun_out/split_ab.txt 2>&1
cat gpurun_out/split_ab.txt
