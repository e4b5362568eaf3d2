#!/usr/bin/env python3
"""k_sph_list diagnostics (SPH_OPT_DEBUG bit 3): fallbacks, list entries and staged candidates per substep.
usage: list_stats.py [config index=3] [substeps=5] [compare-with-slow 0|1] [kernel 2|3]"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
syn = pkg.synthetic
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cmp_slow = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kernel = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = syn.CONFIGS[ci]
rec, _ = syn.make_particles(cfg)
sp = pkg.default_params(**syn.params_fields(cfg))
f = pkg.SPHFluidGPU.from_particles(rec, sp)
f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, kernel)
f.set_option(pkg.SPH_OPT_DEBUG, 8)
g = None
if cmp_slow:
    g = pkg.SPHFluidGPU.from_particles(rec, sp)
    g.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, 1)
for s in range(steps):
    f.DispatchCompute()
    c = f.debug_counters(reset=True)
    lanes = max(c["lanes"], 1)
    row = {"substep": s, "fallback_targets": c["slow_targets"], "overflow": c["overflow_targets"], "far": c["far_targets"],
           "waves_with_fallback": c["waves_with_fallback"], "waves": lanes // 64, "rows_unstaged": c["rows_unstaged"], "rows": c["rows"], "entries_per_lane": round(c["list_entries"] / lanes, 2)}
    if g is not None:
        g.DispatchCompute()
        a, b = f.download(), g.download()
        bad = np.nonzero(a.view(np.uint8).reshape(len(a), -1) != b.view(np.uint8).reshape(len(b), -1))[0]
        row["records_differ"] = int(len(np.unique(bad)))
        if len(bad):
            i = int(bad[0]); row["first"] = i
            row["a"] = [float(x) for x in a[i]["pos"][:3]] + [float(a[i]["density"])]; row["b"] = [float(x) for x in b[i]["pos"][:3]] + [float(b[i]["density"])]
    print(json.dumps(row), flush=True)
