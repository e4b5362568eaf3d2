#!/usr/bin/env python3
"""The compressed regime with accepted-pair walks out of LDS (variant built by tools/patches/dense_lds_walk.py, enabled by SPH_OPT_DEBUG bit 6)
against the shipped exact fallback, along the collapse of config 3: SPH-pass and whole-substep time of 3 substeps from the SAME state, and whether
the two paths leave the same bits.  usage: SPH_HIP_LIB=variants/dense1.so dense_bound.py [steps, e.g. 30,150,300]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
syn = pkg.synthetic
marks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "30,150,300").split(",")]
cfg = syn.CONFIGS[3]
sp = pkg.default_params(**syn.params_fields(cfg))
rec, _ = syn.make_particles(cfg)
f = pkg.SPHFluidGPU.from_particles(rec, sp)
step = 0
REPS = 3
for mark in marks:
    f.DispatchN(mark - step); step = mark
    state = f.download()
    row = {"lib": os.path.basename(os.environ.get("SPH_HIP_LIB", "default")), "substep": mark}
    outs = []
    for name, dbg in (("shipped", 0), ("dense", 64)):
        g = pkg.SPHFluidGPU.from_particles(state, sp)
        g.set_option(pkg.SPH_OPT_DEBUG, dbg)
        g.DispatchN(1)                                   # (import + first sort)
        g.set_option(pkg.SPH_OPT_TIMING, 1)
        g.kernel_times(reset=True)
        g.DispatchN(REPS)
        kt = g.kernel_times(reset=True)
        g.set_option(pkg.SPH_OPT_TIMING, 0)
        row[name + "_sph_us"] = round(kt["sph"][0] / REPS * 1e3, 1)
        row[name + "_substep_us"] = round(sum(ms for k, (ms, c) in kt.items()) / REPS * 1e3, 1)
        outs.append(g.download())
        g.close()
    row["same_bits"] = bool(outs[0].tobytes() == outs[1].tobytes())
    row["speedup_sph"] = round(row["shipped_sph_us"] / row["dense_sph_us"], 3)
    row["speedup_substep"] = round(row["shipped_substep_us"] / row["dense_substep_us"], 3)
    print(json.dumps(row), flush=True)
f.close()
