#!/bin/bash
# PMC passes over k_sph_tile on config 3 (tools/tile_probe.py), summarised into gpurun_out/<tag>_pmc.json
set -e
TAG=${1:-tilepmc}; KS=${2:-k_sph_tile}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 -d $R/gpurun_out/${TAG}_$1 -o p -- python3 $R/tools/tile_probe.py 3 6 3 > $R/gpurun_out/${TAG}_$1.log 2>&1; }
run a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU"
run c "GRBM_GUI_ACTIVE SQ_INSTS_VALU"
python3 - "$R" "$TAG" "$KS" <<'PY'
import sqlite3, json, sys, glob
R, TAG, KS = sys.argv[1:4]
out = {}
for p in "abc":
    for db in glob.glob(f"{R}/gpurun_out/{TAG}_{p}/*.db"):
        con = sqlite3.connect(db)
        q = "select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"
        for name, c, m, n, dur in con.execute(q):
            if KS in name:
                out[c] = {"mean": m, "launches": n, "mean_duration_us": dur / 1e3}
w = out.get("SQ_WAVES", {}).get("mean", 0) or 1
summ = {"kernel": KS, "per_wave": {k: round(v["mean"] / w, 1) for k, v in out.items() if k.startswith("SQ_")}, "counters": out}
json.dump(summ, open(f"{R}/gpurun_out/{TAG}_pmc.json", "w"), indent=1)
print(json.dumps(summ["per_wave"]), {k: round(v["mean"]) for k, v in out.items() if not k.startswith("SQ_")}, out.get("SQ_WAVES"))
PY
