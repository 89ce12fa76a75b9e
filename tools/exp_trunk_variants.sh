# Timing-only variants of the LDS-resident trunk kernel (an operand stream or the epilogue
# NOTE (round 3): the -D variants this script builds (ROW_PAD4 / ROW_OLD_* / IAGO_LPB_* / TRUNK_EXP_*) were removed from
# the product sources; they live in the history: run this from a checkout of commit b61d6ed.
# removed: wrong results, right cost):  bash tools/exp_trunk_variants.sh build ; gpurun -- 'bash tools/exp_trunk_variants.sh run'
cd ${GRAFT_REPO_ROOT:-/root/repo}
VARIANTS="base: noA:-DTRUNK_EXP_NO_A noB:-DTRUNK_EXP_NO_B noEpi:-DTRUNK_EXP_NO_EPI noAB:-DTRUNK_EXP_NO_A|-DTRUNK_EXP_NO_B stamps:-DTRUNK_EXP_STAMPS"
if [ "$1" = build ]; then
  mkdir -p tools/_build
  for v in $VARIANTS; do
    name=${v%%:*}; defs=$(echo "${v#*:}" | tr '|' ' ')
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I iago_amd/csrc -I include $defs \
      -o tools/_build/trunk_$name.so iago_amd/csrc/*.hip 2>&1 | grep -E "error" | head -3
    echo built $name
  done
else
  for v in $VARIANTS; do
    name=${v%%:*}
    echo -n "$name: "; IAGO_HIP_LIB=$PWD/tools/_build/trunk_$name.so python tools/time_value.py 1024 2>/dev/null | tail -1
  done
fi
