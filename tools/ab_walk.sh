#!/bin/bash
# A/B of engine builds under variants/: SPH-pass time of kernel $1 on config 3, launches 5..54 (the bench window) and 0..5
R=${GRAFT_REPO_ROOT:-/root/repo}
NB=$1; shift
for lib in "$@"; do
  if [ "$lib" = default ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$R/variants/$lib; fi
  python3 $R/tools/time_kernels.py 3 $NB 50 5 2>&1 | grep -v amdgpu.ids
done
