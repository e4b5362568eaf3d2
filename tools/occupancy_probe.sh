#!/bin/bash
# Average resident waves and effective clock of the SPH pass (and of the VALU microbenchmark): SQ_LEVEL_WAVES / SQ_BUSY_CYCLES,
# GRBM_GUI_ACTIVE / 8 / wall.  usage: occupancy_probe.sh <neighbor id> [lib]
R=${GRAFT_REPO_ROOT:-/root/repo}
NB=${1:-3}
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export SPH_HIP_LIB=$R/variants/$2
rm -rf $R/gpurun_out/occ_a $R/gpurun_out/occ_b $R/gpurun_out/occ_c
rocprofv3 --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT -d $R/gpurun_out/occ_a -o p -- python3 $R/tools/time_kernels.py 3 $NB 6 0 > $R/gpurun_out/occ_a.log 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/occ_b -o p -- python3 $R/tools/time_kernels.py 3 $NB 6 0 > $R/gpurun_out/occ_b.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d $R/gpurun_out/occ_c -o p -- $R/tools/micro/valu_rate > $R/gpurun_out/occ_c.log 2>&1
python3 - $R/gpurun_out/occ_a $R/gpurun_out/occ_b $R/gpurun_out/occ_c <<'PY'
import sqlite3, glob, sys, json
def q(d, sql):
    for db in glob.glob(d + "/*.db"):
        con = sqlite3.connect(db)
        try:
            return list(con.execute(sql))
        except Exception as ex:
            return [("ERR", str(ex))]
    return []
rows = q(sys.argv[1], "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name")
by = {}
for n, c, v, k in rows:
    if "k_sph" in n: by.setdefault(n[:40], {})[c] = round(v, 1)
print("SPH pass counters (avg per launch):", json.dumps(by))
for n, d in by.items():
    if d.get("SQ_BUSY_CYCLES"):
        print(n, "avg resident waves per SE-cycle?", d["SQ_LEVEL_WAVES"] / d["SQ_BUSY_CYCLES"], " LEVEL/GUI", d["SQ_LEVEL_WAVES"] / max(d.get("GRBM_GUI_ACTIVE", 1), 1))
tabs = q(sys.argv[2], "select name from sqlite_master where type in ('table','view')")
print([t[0] for t in tabs][:40])
kd = q(sys.argv[2], "select name from sqlite_master where name like '%kernel%'")
print(kd)
rows = q(sys.argv[3], "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name")
by = {}
for n, c, v, k in rows: by.setdefault(n[:30], {})[c] = round(v, 1)
for n, d in list(by.items())[:6]:
    print("micro", n, json.dumps(d))
PY
