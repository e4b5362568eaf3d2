#!/bin/bash
# L1 (TCP) / L2 counters of the SPH pass: the cache lines its loads touch in L1 and what goes on to L2.
# usage: mem_counters.sh <kernel substring> <neighbor id> lib...   ("default" = in-tree library)
# (each rocprofv3 pass runs under its own timeout; TA_* / TD_* counter sets made rocprofv3 abort on this image and are left out)
R=${GRAFT_REPO_ROOT:-/root/repo}
KN=$1; NB=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  echo "== $lib"
  if [ "$lib" = default ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$R/variants/$lib; fi
  python3 $R/tools/time_kernels.py 3 $NB 6 0 2>&1 | grep -v amdgpu.ids
  i=0
  for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf $R/gpurun_out/mc_$i; echo "pass $i" >> $R/gpurun_out/mc_progress.log
    timeout -k 10 150 rocprofv3 --pmc $set -d $R/gpurun_out/mc_$i -o p -- python3 $R/tools/time_kernels.py 3 $NB 6 0 > $R/gpurun_out/mc_$i.log 2>&1 || echo "set $i failed"
  done
  python3 - $R/gpurun_out "$KN" <<'PY'
import sqlite3, glob, sys, json
out = {}
for db in glob.glob(sys.argv[1] + "/mc_*/*.db"):
    con = sqlite3.connect(db)
    try:
        for n, c, v in con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
            if sys.argv[2] in n: out[c] = round(v, 1)
    except Exception as ex:
        print("err", ex)
w = out.get("SQ_WAVES", 65536.0)
out["tcp_line_accesses_per_wave"] = round(out.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / w, 1)
out["tcp_l2_read_requests_per_wave"] = round(out.get("TCP_TCC_READ_REQ_sum", 0) / w, 1)
out["tcp_accesses_per_cycle_per_cu"] = round(out.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / 256 / max(out.get("GRBM_GUI_ACTIVE", 1) / 8, 1), 3)
print(json.dumps(out))
PY
done
