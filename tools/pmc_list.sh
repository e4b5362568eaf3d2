#!/bin/bash
# PMC passes over the SPH pass of config 3 (substeps 3-8), summarised into gpurun_out/<tag>_pmc.json.
# usage (on the GPU box): bash tools/pmc_list.sh <tag> [kernel substring]
set -e
TAG=${1:-pmc}; KS=${2:-k_sph_list}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 -d $R/gpurun_out/${TAG}_$1 -o p -- python3 $R/tools/time_pair.py 3 3 6 0 > $R/gpurun_out/${TAG}_$1.log 2>&1; }
run a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU"
run c "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE"
run d "FETCH_SIZE"
run e "WRITE_SIZE"
python3 - "$R" "$TAG" "$KS" <<'PY'
import sqlite3, json, sys, glob
R, TAG, KS = sys.argv[1:4]
out = {}
for p in "abcde":
    for db in glob.glob(f"{R}/gpurun_out/{TAG}_{p}/*.db"):
        con = sqlite3.connect(db)
        q = "select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"
        for name, c, m, n, dur in con.execute(q):
            if KS in name:
                out[c] = {"mean": m, "launches": n, "mean_duration_us": dur / 1e3}
w = out.get("SQ_WAVES", {}).get("mean", 0) or 1
summ = {"kernel": KS, "workload": "config3 (4194304 particles, 128^3 cells), substeps 3-8", "command": "rocprofv3 --pmc <counters of one pass> -- python3 tools/time_pair.py 3 3 6 0 (five separate passes)",
        "per_wave": {k: round(v["mean"] / w, 1) for k, v in out.items() if k.startswith("SQ_")}, "counters": out}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    summ["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"]["mean"] + out["WRITE_SIZE"]["mean"]) * 1024.0
    summ["hbm_note"] = "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as reported; KiB units"
json.dump(summ, open(f"{R}/gpurun_out/{TAG}_pmc.json", "w"), indent=1)
print(json.dumps({k: summ[k] for k in ("per_wave",)}), summ.get("hbm_bytes_per_launch"), {k: round(v["mean"]) for k, v in out.items() if not k.startswith("SQ_")}, out.get("SQ_WAVES"))
PY
