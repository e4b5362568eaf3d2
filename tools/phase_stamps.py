#!/usr/bin/env python3
"""Where a wave of k_sph_walk spends its life (variants/stamps.so, tools/patches/phase_stamps.py): s_memtime differences per phase, written by lanes 0..4 of every
wave into the unused fourth word of acc, averaged over the waves of substep 25 of config 3.  usage: SPH_HIP_LIB=variants/stamps.so phase_stamps.py"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
syn = pkg.synthetic
cfg = syn.CONFIGS[3]
sp = pkg.default_params(**syn.params_fields(cfg))
rec, _ = syn.make_particles(cfg)
f = pkg.SPHFluidGPU.from_particles(rec, sp)
f.DispatchN(5)
f.set_option(pkg.SPH_OPT_DEBUG, 8)
f.set_option(pkg.SPH_OPT_TIMING, 1)
f.kernel_times(reset=True)
f.DispatchN(20)
kt = f.kernel_times(reset=True)
out = f.download()
w = np.ascontiguousarray(out["acc"][:, 3]).view(np.uint32)
w = w[(w & 0x80000000) != 0]
phase, cyc = (w & 7).astype(int), ((w & 0x7fffffff) >> 3).astype(np.float64)
names = ["prologue", "sweep1_nine_rows", "sweep2_integrate", "sweep3_finish_stores", "first_three_rows_of_sweep1"]
res = {"sph_us_with_stamps": round(kt["sph"][0] / 20 * 1e3, 1), "stamped_waves": int((phase == 0).sum())}
for k, nme in enumerate(names):
    v = cyc[phase == k]
    res[nme] = {"mean": round(float(v.mean())), "p10": round(float(np.percentile(v, 10))), "p90": round(float(np.percentile(v, 90)))}
res["total_mean"] = sum(res[n]["mean"] for n in names[:4])
print(json.dumps(res))
f.close()
