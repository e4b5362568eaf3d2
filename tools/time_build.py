#!/usr/bin/env python3
"""Per-kernel times of the substep on config 3, launches 5..54, hipEvents around every kernel class (SPH_OPT_TIMING 1)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
syn = pkg.synthetic
cfg = syn.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 3]
rec, _ = syn.make_particles(cfg)
sp = pkg.default_params(**syn.params_fields(cfg))
f = pkg.SPHFluidGPU.from_particles(rec, sp)
f.DispatchN(5)
f.set_option(pkg.SPH_OPT_TIMING, 1)
f.kernel_times(reset=True)
f.DispatchN(50)
kt = f.kernel_times(reset=True)
print(json.dumps({"lib": os.environ.get("SPH_HIP_LIB", "default"), "us": {k: round(ms / 50 * 1e3, 1) for k, (ms, c) in kt.items() if c}}))
