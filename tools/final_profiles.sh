#!/bin/bash
# Everything profiles/r02_* is refreshed from, in one GPU call (run on the GPU box from the repo root):
#   bash tools/final_profiles.sh  -> gpurun_out/r02_*.{json,log,csv}  (copy to profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
bash tools/profile_bench.sh 50 5 > $O/r02_profile_bench.log 2>&1 && echo "profile_bench done"
python3 bench.py 2> $O/r02_bench.err | tail -1 > $O/r02_bench.json && echo "bench done"
python3 tools/regime_sweep.py 3 330 30 2 2>&1 | grep -v amdgpu.ids > $O/r02_regime_sweep.log && echo "regime done"
python3 tools/time_kernels.py 3 2 3 300 2>&1 | grep -v amdgpu.ids > $O/r02_kernels_at_substep_300.json && echo "kernels@300 done"
python3 bench.py --workload weak5 --steps 1000 --warmup 5 --no-cpu-baseline --settled-after 0 2> $O/r02_weak5.err | tail -1 > $O/r02_bench_weak5_1000.json && echo "weak5 done"
python3 tools/small_scene_pass.py 2>&1 | grep -v amdgpu.ids > $O/r02_small_scene.json && echo "small scene done"
python3 tools/slab_overhead.py 3 40 2>&1 | grep -v amdgpu.ids > $O/r02_slab_overhead.json && echo "slab overhead done"
