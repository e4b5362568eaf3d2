#!/usr/bin/env python3
"""Config 3 to substep 300, then 3 substeps with SPH_OPT_DEBUG = argv[1] (0 = shipped exact fallback, 64 = accepted-pair walks out of LDS in a build patched by
tools/patches/dense_lds_walk.py): the workload of tools/dense_pmc.sh's counter passes (the last 3 launches of k_sph_walk are the ones summarised)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
syn = pkg.synthetic
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = syn.CONFIGS[3]
sp = pkg.default_params(**syn.params_fields(cfg))
rec, _ = syn.make_particles(cfg)
f = pkg.SPHFluidGPU.from_particles(rec, sp)
f.DispatchN(300)
f.set_option(pkg.SPH_OPT_DEBUG, dbg)
f.DispatchN(3)
f.download()
f.close()
