#!/bin/bash
# Phase breakdown of k_sph_list: builds under variants/ compiled with -DSPH_LIST_CUT=1 (stop after sweep 1) / =2 (stop after
# sweep 2) against the full kernel, substeps 0-5 of config 3 (the cut builds do not move the particles, so only the first
# launches see comparable states).  Time per launch and instruction counters per wave.  usage: phase_cut.sh base.so cut1.so cut2.so
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  echo "== $lib"
  SPH_HIP_LIB=$R/variants/$lib python3 $R/tools/time_pair.py 3 0 6 0 2>&1 | grep -v amdgpu.ids
  export SPH_HIP_LIB=$R/variants/$lib
  rm -rf $R/gpurun_out/cut_$lib
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $R/gpurun_out/cut_$lib -o p -- python3 $R/tools/time_pair.py 3 0 6 0 > $R/gpurun_out/cut_$lib.log 2>&1
  python3 - "$R/gpurun_out/cut_$lib" <<'PY'
import sqlite3, glob, sys, json
for db in glob.glob(sys.argv[1] + "/*.db"):
    con = sqlite3.connect(db)
    rows = {c: m for n, c, m in con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name") if "k_sph_list" in n}
    w = rows.get("SQ_WAVES", 1) or 1
    print(json.dumps({k: round(v / w, 1) for k, v in rows.items()}))
PY
done
