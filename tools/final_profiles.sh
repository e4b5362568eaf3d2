#!/bin/bash
# Everything profiles/<tag>_* is refreshed from, in one GPU call (run on the GPU box from the repo root; steps joined with &&):
#   bash tools/final_profiles.sh [tag=r05]  -> gpurun_out/<tag>_*.{json,log,csv}  (copy to profiles/)
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
bash tools/profile_bench.sh 50 5 k_sph_walk $TAG > $O/${TAG}_profile_bench.log 2>&1 && echo "profile_bench done" &&
python3 bench.py 2> $O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench.json && echo "bench done" &&
python3 tools/time_kernels.py 3 3 20 300 2>&1 | grep -v amdgpu.ids > $O/${TAG}_kernels_at_substep_300.json && echo "kernels@300 done" &&
python3 bench.py --workload weak5 --steps 1000 --warmup 5 --no-cpu-baseline --settled-after 0 2> $O/${TAG}_weak5.err | tail -1 > $O/${TAG}_bench_weak5_1000.json && echo "weak5 done" &&
python3 bench.py --slab-path --steps 60 --warmup 5 --no-cpu-baseline --no-breakdown 2> $O/${TAG}_slabpath.err | grep '^{' | tail -1 > $O/${TAG}_bench_slab_path_weak5_one_rank.json && echo "slab path done" &&
python3 bench.py --workload weak5 --steps 60 --warmup 5 --no-cpu-baseline --no-breakdown --settled-after 0 2> /dev/null | tail -1 > $O/${TAG}_bench_weak5_plain_60.json && echo "weak5 plain 60 done" &&
python3 tools/ab_grid_build.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_ab_grid_build_config2.json && echo "A/B grid build done"
