#!/bin/bash
# Counter sets (one rocprofv3 --pmc pass each) over k_sph_walk on config 3, launches 5..24 of tools/time_kernels.py; averages per launch.
# usage: pmc_sets.sh <tag> <lib under variants/ | default> "<set 1>" "<set 2>" ...   -> gpurun_out/<tag>_pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; LIB=$2; shift 2
if [ "$LIB" = default ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$R/variants/$LIB; fi
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf $R/gpurun_out/${TAG}_$i; echo "$TAG pass $i: $set" >> $R/gpurun_out/pmc_progress.log
  timeout -k 10 200 rocprofv3 --pmc $set -d $R/gpurun_out/${TAG}_$i -o p -- python3 $R/tools/time_kernels.py 3 3 20 5 > $R/gpurun_out/${TAG}_$i.log 2>&1 || echo "set $i ($set) failed"
done
python3 - $R/gpurun_out $TAG $LIB <<'PY'
import sqlite3, glob, sys, json
out = {"lib": sys.argv[3], "kernel": "k_sph_walk", "launches": "5..24 of tools/time_kernels.py 3 3 20 5 (config 3)"}
for db in sorted(glob.glob(f"{sys.argv[1]}/{sys.argv[2]}_*/*.db")):
    con = sqlite3.connect(db)
    try:
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' or type='view'")]
        t = [x for x in tabs if x.startswith("counters_collection")][0]
        rows = {}
        for n, did, c, v in con.execute(f"select kernel_name, dispatch_id, counter_name, value from {t} order by dispatch_id"):
            if "k_sph_walk" in n: rows.setdefault(c, []).append(v)
        for c, vs in rows.items():
            w = vs[5:25]
            out[c] = round(sum(w) / max(len(w), 1), 1)
    except Exception as ex:
        out["err_" + db.split("/")[-2]] = str(ex)
print(json.dumps(out))
open(f"{sys.argv[1]}/{sys.argv[2]}_pmc.json", "w").write(json.dumps(out, indent=1))
PY
