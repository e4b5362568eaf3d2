#!/usr/bin/env python3
"""SPH-pass time of k_sph_walk with test hooks forced (SPH_OPT_DEBUG): 0 = shipped, 4 = no LDS windows (every candidate row walked with per-lane global loads),
1 = every list treated as overflowed (plain sweeps 2 / 3), 2 = sweep-3 fallback for every target.  Config 3, launches 5..24.  usage: time_debug_flags.py [flags, e.g. 0,4,1,2]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
syn = pkg.synthetic
flags = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,4,1,2,0").split(",")]
cfg = syn.CONFIGS[3]
sp = pkg.default_params(**syn.params_fields(cfg))
rec, _ = syn.make_particles(cfg)
for dbg in flags:
    f = pkg.SPHFluidGPU.from_particles(rec, sp)
    f.set_option(pkg.SPH_OPT_DEBUG, dbg)
    f.DispatchN(5)
    f.set_option(pkg.SPH_OPT_TIMING, 1)
    f.kernel_times(reset=True)
    f.DispatchN(20)
    kt = f.kernel_times(reset=True)
    print(json.dumps({"debug": dbg, "sph_us": round(kt["sph"][0] / 20 * 1e3, 1)}), flush=True)
    f.close()
