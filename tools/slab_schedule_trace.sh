#!/bin/bash
# Timeline of one boundary-first substep of the slab path (bench.py --slab-path, weak5) for engine builds under variants/:
# when the face launches end, when the exchange chain ends, when the interior ends, and the substep time without the profiler.
# usage (GPU box): bash tools/slab_schedule_trace.sh default ia0.so ia2.so
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = default ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$R/variants/$lib; fi
  rm -rf $R/gpurun_out/sst_$lib
  timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/sst_$lib -o p -- python3 $R/bench.py --slab-path --workload weak5 --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/sst_$lib.err
  ms=$(timeout -k 10 300 python3 $R/bench.py --slab-path --workload weak5 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  python3 - $R/gpurun_out/sst_$lib/p_results.db "$lib" "$ms" <<'PY'
import sqlite3, sys, statistics
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_bin" in r[0]]
out = []
for a, b in zip(idx[10:-2], idx[11:-1]):
    seg = rows[a:b]
    walks = [r for r in seg if "k_sph_walk" in r[0]]
    if len(walks) != 3: continue
    t0 = min(w[1] for w in walks)
    by_len = sorted(walks, key=lambda w: w[2] - w[1])
    faces_end = max(by_len[0][2], by_len[1][2]); interior_end = by_len[2][2]
    chain = [r for r in seg if "k_slab_commit" in r[0]]
    out.append(((faces_end - t0) / 1e3, (chain[-1][2] - t0) / 1e3 if chain else float("nan"), (interior_end - t0) / 1e3, (rows[b][1] - rows[a][1]) / 1e3))
med = lambda k: round(statistics.median(o[k] for o in out), 1)
print(f"{sys.argv[2]}: us after the start of the SPH pass: faces done {med(0)}, exchange kernels done {med(1)}, interior done {med(2)}; substep {med(3)} us under the profiler, {float(sys.argv[3]) * 1e3:.1f} us without")
PY
done
