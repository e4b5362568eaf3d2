#!/usr/bin/env python3
"""SPH-pass time of the reference's default scene (SPHFluidGPU(50000), 52^3 grid) once it has settled into a pool
(after `settle` substeps), per kernel class.  usage: small_scene_pass.py [settle=2000] [timed=200]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
settle = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
timed = int(sys.argv[2]) if len(sys.argv) > 2 else 200
f = pkg.SPHFluidGPU(50000, seed=1)
f.DispatchN(settle)
f.set_option(pkg.SPH_OPT_DEBUG, 8)
f.DispatchN(1)
c = f.debug_counters(reset=True)
f.set_option(pkg.SPH_OPT_DEBUG, 0)
f.set_option(pkg.SPH_OPT_TIMING, 1)
f.kernel_times(reset=True)
f.DispatchN(timed)
kt = f.kernel_times(reset=True)
d = f.download()
print(json.dumps({"scene": "SPHFluidGPU(50000) default members", "settled_substeps": settle,
                  "us": {k: round(ms / max(cnt, 1) * 1e3, 1) for k, (ms, cnt) in kt.items() if cnt},
                  "density_max_over_rest": round(float(d["density"].max()) / 1000.0, 2), "entries_per_lane": round(c["list_entries"] / max(c["lanes"], 1), 1),
                  "fallback_targets": c["slow_targets"], "overflow": c["overflow_targets"], "far": c["far_targets"]}))
