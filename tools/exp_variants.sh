#!/bin/bash
# Builds timing-only variants of the rollout kernel (wrong results!) and times them.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for V in "" "-DEXP_CONST_E" "-DEXP_NO_RAY" "-DEXP_CONST_E -DEXP_NO_RAY"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $V -o /tmp/libvar.so iago_amd/csrc/*.hip 2>/dev/null
  cp iago_amd/libiago_hip.so /tmp/orig.so; cp /tmp/libvar.so iago_amd/libiago_hip.so
  echo "variant [$V]: $(python bench.py --steps 200 --no-cpu-baseline --mcts-turns 0 --large-boards 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
  cp /tmp/orig.so iago_amd/libiago_hip.so
done
