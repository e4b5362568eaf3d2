#!/bin/bash
# rocprofv3 passes over EXACTLY the command bench.py's headline comes from, summarised for exactly its timed launches
# (launches W .. W+K-1 of the SPH kernel): kernel-trace average + HBM bytes + VALU instructions + L1 cache-line accesses.
# usage (on the GPU box): bash tools/profile_bench.sh [steps=50] [warmup=5] [kernel=k_sph_walk] [tag=r04]
#   -> gpurun_out/<tag>_bench_*.json (copy to profiles/)
set -e
K=${1:-50}; W=${2:-5}; KN=${3:-k_sph_walk}; TAG=${4:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python3 $R/bench.py --steps $K --warmup $W --no-cpu-baseline --no-breakdown"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pb_stats -o p -- $CMD > $R/gpurun_out/pb_stats.json 2> $R/gpurun_out/pb_stats.err
python3 $R/tools/rocpd_summary.py stats $R/gpurun_out/pb_stats/p_results.db $R/gpurun_out/${TAG}_bench_kernel_stats.csv
python3 $R/tools/rocpd_summary.py window $R/gpurun_out/pb_stats/p_results.db $KN $W $K $R/gpurun_out/${TAG}_bench_kernel_window.json
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU TCP_TOTAL_CACHE_ACCESSES_sum; do
  echo "pmc pass $c" >> $R/gpurun_out/pb_progress.log
  timeout -k 10 300 rocprofv3 --pmc $c -d $R/gpurun_out/pb_$c -o p -- $CMD > $R/gpurun_out/pb_$c.json 2> $R/gpurun_out/pb_$c.err
done
HASH=$(cd $R && python3 -c "import importlib; print(importlib.import_module('componentframeworks-smoothed-particle-hydrodynamics_amd').build.csrc_hash())")
python3 $R/tools/rocpd_summary.py bench-counters $R/gpurun_out/pb_FETCH_SIZE/p_results.db $R/gpurun_out/pb_WRITE_SIZE/p_results.db $R/gpurun_out/pb_SQ_INSTS_VALU/p_results.db \
  $KN $W $K config3 "profiles/${TAG}_bench_counters.json: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU | TCP_TOTAL_CACHE_ACCESSES_sum (one counter per pass) -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline --no-breakdown; mean over launches $W..$((W+K-1)) of $KN (the timed window); builder-measured, not measured in the driver's run" \
  $R/gpurun_out/${TAG}_bench_counters.json $R/gpurun_out/pb_TCP_TOTAL_CACHE_ACCESSES_sum/p_results.db $HASH
tail -1 $R/gpurun_out/pb_stats.json
