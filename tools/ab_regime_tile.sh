#!/bin/bash
# SPH-pass time of k_sph_tile builds (variants/) against k_sph_walk along the collapse of config 3: usage ab_regime_tile.sh lib...
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  export SPH_HIP_LIB=$R/variants/$lib
  echo "== $lib"
  python3 $R/tools/regime_sweep.py 3 330 60 4,3 2>&1 | grep -v amdgpu.ids
done
