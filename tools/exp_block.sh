cd ${GRAFT_REPO_ROOT:-/root/repo}
cp iago_amd/libiago_hip.so /tmp/orig.so
for BLK in 256 512 1024; do
  sed "s/const int block = (threads >= 256) ? 256 : 64;/const int block = (threads >= $BLK) ? $BLK : 64;/; s/__launch_bounds__(256) void rollout_kernel/__launch_bounds__($BLK) void rollout_kernel/" iago_amd/csrc/rollout_kernel.hip > /tmp/rk_var.hip
  cp /tmp/rk_var.hip /tmp/rollout_kernel.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I iago_amd/csrc -I include -o iago_amd/libiago_hip.so iago_amd/csrc/abi_common.hip iago_amd/csrc/rules_kernels.hip iago_amd/csrc/mcts_kernels.hip /tmp/rollout_kernel.hip 2>&1 | grep -E "error" | head -3
  python bench.py --no-cpu-baseline --mcts-turns 0 --train-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('block $BLK', round(d['value']/1e6,1), 'M games/s; large', round(d['large_batch']['games_per_sec']/1e6,1), 'M', round(d['large_batch']['kernel_ms'],3),'ms')"
done
cp /tmp/orig.so iago_amd/libiago_hip.so
