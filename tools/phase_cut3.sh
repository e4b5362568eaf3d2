#!/bin/bash
# Phase breakdown of an SPH-pass kernel: builds under variants/ (…_CUT=1 stops after sweep 1, =2 after sweep 2) against the
# full kernel, launches 0-5 of config 3 (the cut builds do not move the particles, so only the first launches see comparable
# states).  Time per launch and SQ counters per wave.
# usage: phase_cut3.sh <kernel-name-substring> <neighbor id> lib1.so [lib2.so ...]   ("default" = the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
KN=$1; NB=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  echo "== $lib"
  if [ "$lib" = default ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$R/variants/$lib; fi
  python3 $R/tools/time_kernels.py 3 $NB 6 0 2>&1 | grep -v amdgpu.ids
  rm -rf $R/gpurun_out/cut_$lib
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $R/gpurun_out/cut_$lib -o p -- python3 $R/tools/time_kernels.py 3 $NB 6 0 > $R/gpurun_out/cut_$lib.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $R/gpurun_out/cutb_$lib -o p -- python3 $R/tools/time_kernels.py 3 $NB 6 0 > $R/gpurun_out/cutb_$lib.log 2>&1
  python3 - "$R/gpurun_out/cut_$lib" "$R/gpurun_out/cutb_$lib" "$KN" <<'PY'
import sqlite3, glob, sys, json
out = {}
for d in sys.argv[1:3]:
    for db in glob.glob(d + "/*.db"):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='view' or type='table'")]
        try:
            rows = {c: m for n, c, m in con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name") if sys.argv[3] in n}
        except Exception as ex:
            print("no counters_collection:", ex, tabs[:8]); continue
        w = rows.get("SQ_WAVES", 1) or 1
        out.update({k: round(v / w, 1) for k, v in rows.items()})
print(json.dumps(out))
PY
done
