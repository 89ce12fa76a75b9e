# Variants of the 16-lane rollout kernel by -D flags: build here, time on the GPU box.
# NOTE (round 3): the -D variants this script builds (ROW_PAD4 / ROW_OLD_* / IAGO_LPB_* / TRUNK_EXP_*) were removed from
# the product sources; they live in the history: run this from a checkout of commit b61d6ed.
#   bash tools/exp_row_variants.sh build "name:-DFLAG ..." ; gpurun -- 'bash tools/exp_row_variants.sh run "name ..."'
cd ${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
  mkdir -p tools/_build
  others=$(ls iago_amd/_obj/*.o | grep -v rollout_row)
  for v in $2; do
    name=${v%%:*}; defs=$(echo "${v#*:}" | tr '|' ' '); [ "$defs" = "$name" ] && defs=""
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -I iago_amd/csrc -I include $defs \
        -c iago_amd/csrc/rollout_row_kernel.hip -o tools/_build/row_$name.o 2>&1 | grep -E "error"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/row_$name.so tools/_build/row_$name.o $others
    echo built $name
  done
else
  for name in $2; do
    IAGO_HIP_LIB=$PWD/tools/_build/row_$name.so python bench.py --gpus 1 --steps 20 --warmup 5 --rollout-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M games/s', round(d['ms_per_step']*1e3,3), 'us')"
  done
fi
