#!/bin/bash
# Build an engine variant into variants/<name>.so with extra -D flags (A/B timing through SPH_HIP_LIB).
# usage: tools/build_variant.sh name [-DSPH_WALK_CUT=1 ...]
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/variants
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -Wno-unused-function "$@" \
  -o $R/variants/$name.so $R/componentframeworks-smoothed-particle-hydrodynamics_amd/csrc/sph_engine.hip && echo built variants/$name.so
