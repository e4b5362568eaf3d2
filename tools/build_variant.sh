#!/bin/bash
# Build an engine variant into variants/<name>.so (A/B timing through SPH_HIP_LIB).  The shipped sources stay free of experiment switches:
# a variant is the shipped csrc/ copied to variants/<name>_src/ with a patch script applied (python file taking the directory), plus -D flags.
# usage: tools/build_variant.sh name [patch.py ...] [-DSPH_WALK_MAXN=36 ...]
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/variants
src=$R/variants/${name}_src
rm -rf $src && mkdir -p $src/pkg/csrc $src/include && cp $R/componentframeworks-smoothed-particle-hydrodynamics_amd/csrc/* $src/pkg/csrc/ && cp $R/include/*.h $src/include/
flags=()
for a in "$@"; do
  case "$a" in
    *.py) python3 "$a" $src/pkg/csrc || exit 1 ;;
    *) flags+=("$a") ;;
  esac
done
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -Wno-unused-function "${flags[@]}" \
  -Rpass-analysis=kernel-resource-usage -o $R/variants/$name.so $src/pkg/csrc/sph_engine.hip 2> $R/variants/$name.res || { cat $R/variants/$name.res; exit 1; }
grep -A12 "Function Name: _ZN3sph10k_sph_walkILi41ELi3ELi144ELb1" $R/variants/$name.res | grep -o "VGPRs: [0-9]*\|ScratchSize.*: [0-9]*\|Occupancy.*: [0-9]*\|LDS Size.*: [0-9]*\|SGPRs: [0-9]*" | tr '\n' ' '
echo; echo built variants/$name.so
