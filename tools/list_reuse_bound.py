#!/usr/bin/env python3
"""what would reusing the neighbour lists and the sort order over several substeps buy?
Needs an engine built with -DSPH_WALK_EXP=3 (tools/build_variant.sh w_exp3 -DSPH_WALK_EXP=3; SPH_HIP_LIB=variants/w_exp3.so).
Timing only: the reuse substeps read the lists one substep kept in global memory (sweep 1 = a walk over 10-15 listed neighbours instead
of the scan of 58 candidates) and skip the grid build; they all compute from the same frozen sorted copy, so the values are not a
simulation.  usage: list_reuse_bound.py [config index=3] [first substep=5] [substeps timed=20]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
import torch
syn = pkg.synthetic
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 3
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = syn.CONFIGS[ci]
sp = pkg.default_params(**syn.params_fields(cfg))
rec, _ = syn.make_particles(cfg)
f = pkg.SPHFluidGPU.from_particles(rec, sp)
f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, 3)
f.DispatchN(first)


def timed(n, dbg):
    f.set_option(pkg.SPH_OPT_DEBUG, dbg)
    f.set_option(pkg.SPH_OPT_TIMING, 1)
    f.kernel_times(reset=True)
    torch.cuda.synchronize(); f.sync()
    t0 = time.perf_counter()
    f.DispatchN(n)
    f.sync()
    wall = (time.perf_counter() - t0) / n * 1e6
    kt = f.kernel_times(reset=True)
    f.set_option(pkg.SPH_OPT_TIMING, 0)
    return {"wall_us_per_substep_with_events": round(wall, 1), "us": {k: round(ms / n * 1e3, 1) for k, (ms, c) in kt.items() if c}}


out = {"lib": os.environ.get("SPH_HIP_LIB", "default"), "config": cfg.name, "first_substep": first}
out["ordinary_substeps"] = timed(steps, 0)
f.set_option(pkg.SPH_OPT_DEBUG, 64)            # one substep keeps its lists
f.DispatchN(1)
out["reuse_substeps"] = timed(steps, 32)
print(json.dumps(out), flush=True)
f.close()
