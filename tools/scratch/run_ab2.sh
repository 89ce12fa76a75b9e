cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 bash tools/ab_env.sh "IAGO_HIP_LIB=tools/_build/libiago_oldwalk.so" "IAGO_SEARCH_SPLIT=0" "IAGO_SEARCH_SPLIT=32 IAGO_PERSISTENT_GPW=16" "IAGO_SEARCH_SPLIT=32" "IAGO_SEARCH_SPLIT=24 IAGO_PERSISTENT_GPW=24" > gpurun_out/ab2.txt 2>&1
cat gpurun_out/ab2.txt
AB_ARGS="--mcts-sims 400" timeout -k 10 600 bash tools/ab_env.sh "IAGO_HIP_LIB=tools/_build/libiago_oldwalk.so" "IAGO_SEARCH_SPLIT=0" "IAGO_SEARCH_SPLIT=32 IAGO_PERSISTENT_GPW=16" > gpurun_out/ab2_400.txt 2>&1
cat gpurun_out/ab2_400.txt
