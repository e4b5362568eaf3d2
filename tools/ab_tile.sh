#!/bin/bash
# A/B of engine builds under variants/ through tools/tile_probe.py (k_sph_tile on config 3): usage ab_tile.sh [steps] lib...
R=${GRAFT_REPO_ROOT:-/root/repo}
ST=$1; shift
for lib in "$@"; do
  if [ "$lib" = default ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$R/variants/$lib; fi
  python3 $R/tools/tile_probe.py 3 $ST 5 2>&1 | grep -v amdgpu.ids
done
