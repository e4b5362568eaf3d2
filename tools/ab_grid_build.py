#!/usr/bin/env python3
"""BASELINE.json configs[1]: 262 144 particles, 64^3 grid, linked-list vs counting-sort grid build.
Reports per-substep build time and the downstream SPH-pass time for each (hipEvents)."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
syn = pkg.synthetic
cfg = syn.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rec, _ = syn.make_particles(cfg)
sp = pkg.default_params(**syn.params_fields(cfg))
out = {"config": cfg.name, "steps": steps}
for label, build, neighbor in (("counting_sort+k_sph_walk", 0, 3), ("counting_sort+k_sph_list", 0, 2), ("counting_sort+k_sph_slow", 0, 1),
                               ("linked_list+k_sph_ll", 1, 3)):
    sim = pkg.SPHFluidGPU.from_particles(rec, sp)
    sim.set_option(pkg.SPH_OPT_GRID_BUILD, build)
    sim.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, neighbor)
    sim.DispatchN(5)
    sim.set_option(pkg.SPH_OPT_TIMING, 1)
    sim.kernel_times(reset=True)
    sim.DispatchN(steps)
    kt = sim.kernel_times(reset=True)
    us = {k: round(ms / steps * 1e3, 2) for k, (ms, c) in kt.items() if c}
    build_us = us.get("bin", 0) + us.get("scan", 0) + us.get("scatter", 0)
    out[label] = {"build_us": round(build_us, 2), "sph_us": us.get("sph", 0) + us.get("other", 0), "kernels_us": us}
    sim.close()
print(json.dumps(out, indent=1))
