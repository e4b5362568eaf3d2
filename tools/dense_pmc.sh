#!/bin/bash
# Counters of k_sph_walk at substeps 300..302 of config 3: shipped exact fallback (debug 0) against accepted-pair walks out of LDS (debug 64), in variants/dense1.so.
R=${GRAFT_REPO_ROOT:-/root/repo}
export SPH_HIP_LIB=$R/variants/dense1.so
cd /tmp && export TMPDIR=/tmp
for dbg in 0 64; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1)); rm -rf $R/gpurun_out/dpmc_${dbg}_$i
    timeout -k 10 200 rocprofv3 --pmc $set -d $R/gpurun_out/dpmc_${dbg}_$i -o p -- python3 $R/tools/dense_pmc.py $dbg > $R/gpurun_out/dpmc_${dbg}_$i.log 2>&1 || echo "set $i failed"
  done
  python3 - $R/gpurun_out $dbg <<'PY'
import sqlite3, glob, sys, json
out = {"debug": int(sys.argv[2]), "launches": "the last 3 launches of k_sph_walk (substeps 300..302 of config 3)"}
for db in sorted(glob.glob(f"{sys.argv[1]}/dpmc_{sys.argv[2]}_*/*.db")):
    con = sqlite3.connect(db)
    t = [r[0] for r in con.execute("select name from sqlite_master") if r[0].startswith("counters_collection")][0]
    rows = {}
    for n, did, c, v in con.execute(f"select kernel_name, dispatch_id, counter_name, value from {t} order by dispatch_id"):
        if "k_sph_walk" in n: rows.setdefault(c, []).append(v)
    for c, vs in rows.items():
        w = vs[-3:]
        out[c] = round(sum(w) / max(len(w), 1), 1)
print(json.dumps(out))
PY
done
