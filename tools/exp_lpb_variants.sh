# Builds variants of the lane-per-board rollout kernel (block size, waves per SIMD) HERE
# NOTE (round 3): the -D variants this script builds (ROW_PAD4 / ROW_OLD_* / IAGO_LPB_* / TRUNK_EXP_*) were removed from
# the product sources; they live in the history: run this from a checkout of commit b61d6ed.
# and benches each on the GPU box:  bash tools/exp_lpb_variants.sh build ; gpurun -- 'bash tools/exp_lpb_variants.sh run'
cd ${GRAFT_REPO_ROOT:-/root/repo}
VARIANTS="b256: b128:-DIAGO_LPB_BLOCK=128 b64:-DIAGO_LPB_BLOCK=64 b256w5:-DIAGO_LPB_ATTR=__attribute__((amdgpu_waves_per_eu(5,5))) b64w5:-DIAGO_LPB_BLOCK=64|-DIAGO_LPB_ATTR=__attribute__((amdgpu_waves_per_eu(5,5))) b64w6:-DIAGO_LPB_BLOCK=64|-DIAGO_LPB_ATTR=__attribute__((amdgpu_waves_per_eu(6,6)))"
if [ "$1" = build ]; then
  mkdir -p tools/_build
  for v in $VARIANTS; do
    name=${v%%:*}; defs=$(echo "${v#*:}" | tr '|' ' ')
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I iago_amd/csrc -I include $defs \
      -o tools/_build/lpb_$name.so iago_amd/csrc/*.hip 2>&1 | grep -E "error|spill" | head -3
    echo built $name
  done
else
  for v in $VARIANTS; do
    name=${v%%:*}
    IAGO_HIP_LIB=$PWD/tools/_build/lpb_$name.so python bench.py --no-cpu-baseline --mcts-turns 0 --train-iters 0 --large-boards 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name', round(d['value']/1e6,1), 'M games/s', round(r['kernel_ms'],4), 'ms')"
  done
fi
