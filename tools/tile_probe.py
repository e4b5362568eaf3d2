#!/usr/bin/env python3
"""k_sph_tile on a bench workload: blocks of cells it took / left to k_sph_walk, list statistics, SPH-pass time.
usage: tile_probe.py [config index=3] [substeps=20] [untimed substeps first=5]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
syn = pkg.synthetic
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = syn.CONFIGS[ci]
sp = pkg.default_params(**syn.params_fields(cfg))
rec, _ = syn.make_particles(cfg)
f = pkg.SPHFluidGPU.from_particles(rec, sp)
f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, 4)
f.DispatchN(warm)
f.set_option(pkg.SPH_OPT_DEBUG, 8)
f.debug_counters(reset=True)
f.DispatchN(1)
c = f.debug_counters(reset=True)
f.set_option(pkg.SPH_OPT_DEBUG, 0)
f.set_option(pkg.SPH_OPT_TIMING, 1)
f.kernel_times(reset=True)
f.DispatchN(steps)
kt = f.kernel_times(reset=True)
print(json.dumps({"lib": os.environ.get("SPH_HIP_LIB", "default"), "config": cfg.name, "counters_one_substep": c,
                  "us": {k: round(ms / steps * 1e3, 1) for k, (ms, n) in kt.items() if n}}), flush=True)
f.close()
