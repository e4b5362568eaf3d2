R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do echo "== $lib"; SPH_HIP_LIB=$R/variants/$lib python3 $R/tools/time_pair.py 3 5 50 0 2>&1 | grep -v amdgpu.ids; done
