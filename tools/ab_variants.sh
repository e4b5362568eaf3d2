#!/bin/bash
# A/B of engine builds under variants/ (SPH_HIP_LIB): SPH-pass time on config 3 (substeps 5-55) and fallback counters
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  echo "== $lib"
  SPH_HIP_LIB=$R/variants/$lib python3 $R/tools/time_pair.py 3 5 50 0 2>&1 | grep -v amdgpu.ids
  SPH_HIP_LIB=$R/variants/$lib python3 $R/tools/list_stats.py 3 56 0 2>&1 | grep -v amdgpu.ids | awk 'NR%10==6'
done
