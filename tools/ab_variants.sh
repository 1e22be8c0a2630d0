#!/bin/bash
# Build kernel variants with extra -D flags into ab_build/ (git-ignored, travels to the GPU box) and, on the GPU box,
# bench them interleaved in ONE call:  tools/ab_variants.sh build  (here)   /   tools/ab_variants.sh run  (GPU box)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SRC=$R/drl_uav_cellularnet_amd/csrc/uavenv_capi.hip
OUT=$R/ab_build
VARIANTS=(${UAVENV_AB_VARIANTS:-"full:-DUAVENV_WAVES_PER_BLOCK=4" "skeleton:-DUAVENV_SKELETON"})
if [ "$1" = build ]; then
  mkdir -p $OUT
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}; flags=${v#*:}
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 $flags -o $OUT/libuavenv_$name.so $SRC &
  done
  wait; ls -la $OUT
else
  for round in ${UAVENV_AB_ROUNDS:-1 2 3}; do
    for v in "${VARIANTS[@]}"; do
      name=${v%%:*}
      UAVENV_LIB=$OUT/libuavenv_$name.so python3 $R/bench.py --steps 1500 --warmup 100 --no-cpu-baseline "${@:2}" 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('round $round  $name  %.3f us  %.1f M env-steps/s' % (d['roofline']['avg_launch_us'], d['value']/1e6))"
    done
  done
fi
