#!/bin/bash
# A/B of engine builds under variants/ (SPH_HIP_LIB): SPH pass near the lattice state (substeps 5-55 of config 3) and along
# the collapse (substeps 100 / 200 / 300).  usage: ab_regime.sh a.so b.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  echo "== $lib"
  SPH_HIP_LIB=$R/variants/$lib python3 $R/tools/time_pair.py 3 5 50 0 2>&1 | grep -v amdgpu.ids
  SPH_HIP_LIB=$R/variants/$lib python3 $R/tools/regime_sweep.py 3 300 100 2 2>&1 | grep -v amdgpu.ids
done
