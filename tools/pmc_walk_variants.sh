#!/bin/bash
# VALU wave-instructions and L1 line accesses per launch of k_sph_walk for engine builds under variants/ (config 3, launches 5..24): usage pmc_walk_variants.sh lib...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export SPH_HIP_LIB=$R/variants/$lib
  for c in SQ_INSTS_VALU TCP_TOTAL_CACHE_ACCESSES_sum; do
    rm -rf $R/gpurun_out/pv_$c
    timeout -k 10 300 rocprofv3 --pmc $c -d $R/gpurun_out/pv_$c -o p -- python3 $R/tools/time_kernels.py 3 3 20 5 > /dev/null 2>&1
  done
  python3 - "$R" "$lib" <<'PY'
import sqlite3, sys, glob
R, lib = sys.argv[1:3]
out = {}
for c in ("SQ_INSTS_VALU", "TCP_TOTAL_CACHE_ACCESSES_sum"):
    for db in glob.glob(f"{R}/gpurun_out/pv_{c}/*.db"):
        con = sqlite3.connect(db)
        rows = [v for name, did, v in con.execute("select kernel_name, dispatch_id, value from counters_collection where counter_name like ? order by dispatch_id", (c.split("_sum")[0] + "%",)) if "k_sph_walk" in name]
        w = rows[5:25]
        out[c] = sum(w) / max(len(w), 1)
print(lib, {k: round(v / 1e6, 2) for k, v in out.items()}, "(millions per launch, launches 5..24)")
PY
done
