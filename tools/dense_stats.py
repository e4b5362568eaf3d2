#!/usr/bin/env python3
"""What the compressed regime consists of: candidates and true neighbours per particle along a run of BASELINE config 3.
usage: dense_stats.py [config index=3] [checkpoints, e.g. 0,50,150,300] [sample=20000]
Per checkpoint: particles per occupied cell, candidates per particle (27 cells), neighbours within h (sampled with a k-d
tree), fallback counters of k_sph_walk, and the time of the SPH pass."""
import importlib, json, os, sys
import numpy as np
from scipy.spatial import cKDTree
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
syn = pkg.synthetic
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 3
marks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,50,150,300").split(",")]
sample = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
cfg = syn.CONFIGS[ci]
rec, _ = syn.make_particles(cfg)
sp = pkg.default_params(**syn.params_fields(cfg))
g = pkg.compute_grid_extents(sp)
gx, gy, gz = [int(v) for v in g.dims]
gmin = np.array([float(v) for v in g.gridMin], dtype=np.float32)
cs = np.float32(g.cellSize)
h = float(cfg.h) if hasattr(cfg, "h") else float(cs)
f = pkg.SPHFluidGPU.from_particles(rec, sp)
f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, 3)
done = 0
rng = np.random.default_rng(1)
for m in marks:
    if m > done:
        f.DispatchN(m - done)
        done = m
    P = f.download()
    pos = P["pos"][:, :3].astype(np.float32)
    c = np.clip(np.floor((pos - gmin) / cs).astype(np.int64), 0, [gx - 1, gy - 1, gz - 1])
    cnt = np.zeros((gz, gy, gx), dtype=np.int64)
    np.add.at(cnt, (c[:, 2], c[:, 1], c[:, 0]), 1)
    pad = np.pad(cnt, 1)
    box = np.zeros_like(cnt)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                box += pad[dz:dz + gz, dy:dy + gy, dx:dx + gx]
    cand = box[c[:, 2], c[:, 1], c[:, 0]]
    tree = cKDTree(pos)
    idx = rng.choice(len(pos), size=min(sample, len(pos)), replace=False)
    nb = tree.query_ball_point(pos[idx], r=float(cs), return_length=True) - 1
    f.set_option(pkg.SPH_OPT_DEBUG, 8)
    f.set_option(pkg.SPH_OPT_TIMING, 1)
    f.debug_counters(reset=True); f.kernel_times(reset=True)
    f.DispatchCompute(); done += 1
    dc = f.debug_counters(reset=True)
    kt = f.kernel_times(reset=True)
    f.set_option(pkg.SPH_OPT_DEBUG, 0)
    f.set_option(pkg.SPH_OPT_TIMING, 0)
    occ = cnt[cnt > 0]
    print(json.dumps({"substep": m, "per_occupied_cell_mean": round(float(occ.mean()), 2), "per_cell_max": int(cnt.max()),
                      "candidates_mean": round(float(cand.mean()), 1), "candidates_p99": int(np.percentile(cand, 99)),
                      "neighbours_within_h_mean": round(float(nb.mean()), 1), "neighbours_p99": int(np.percentile(nb, 99)),
                      "hit_rate": round(float(nb.mean() / cand.mean()), 3), "max_density_over_rho0": round(float(P["density"].max() / sp.param_restDensity), 2) if hasattr(sp, "param_restDensity") else None,
                      "fallback_targets": dc["slow_targets"], "overflow": dc["overflow_targets"], "lanes": dc["lanes"],
                      "sph_ms": round(kt["sph"][0], 3) if "sph" in kt else None}), flush=True)
