# Variants of the lane-per-board rollout kernel with per-turn re-binning of a block's boards by
# NOTE (round 3): the -D variants this script builds (ROW_PAD4 / ROW_OLD_* / IAGO_LPB_* / TRUNK_EXP_*) were removed from
# the product sources; they live in the history: run this from a checkout of commit b61d6ed.
# mobility (IAGO_LPB_REBIN).  bash tools/exp_lpb_rebin.sh build   (here; needs iago_amd/_obj from
# `python -m iago_amd.build`), then on the GPU box: bash tools/exp_lpb_rebin.sh run
cd ${GRAFT_REPO_ROOT:-/root/repo}
VARIANTS=${VARIANTS:-"r0:-DIAGO_LPB_REBIN=0 r1:-DIAGO_LPB_REBIN=1 r1b512:-DIAGO_LPB_REBIN=1|-DIAGO_LPB_BLOCK=512 r1b128:-DIAGO_LPB_REBIN=1|-DIAGO_LPB_BLOCK=128"}
if [ "$1" = build ]; then
  mkdir -p tools/_build
  others=$(ls iago_amd/_obj/*.o | grep -v rollout_lpb)
  for v in $VARIANTS; do
    name=${v%%:*}; defs=$(echo "${v#*:}" | tr '|' ' ')
    mkdir -p tools/_build/tmp_$name
    ( cd tools/_build/tmp_$name && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden \
        -I ../../../iago_amd/csrc -I ../../../include $defs -save-temps -c ../../../iago_amd/csrc/rollout_lpb_kernel.hip -o lpb.o 2>&1 | grep -E "error" ;
      grep -E "\.vgpr_count|spill_count|group_segment_fixed" rollout_lpb_kernel-hip-amdgcn-amd-amdhsa-gfx950.s | tr '\n' ' ' ; echo )
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/lpb_$name.so tools/_build/tmp_$name/lpb.o $others
    echo built $name
  done
else
  for v in $VARIANTS; do
    name=${v%%:*}
    IAGO_HIP_LIB=$PWD/tools/_build/lpb_$name.so python tools/run_large.py ${BOARDS:-1048576} 20 | sed "s/^/$name /"
  done
fi
