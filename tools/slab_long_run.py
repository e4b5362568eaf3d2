#!/usr/bin/env python3
"""Does a long run of a BASELINE config keep inside the slab exchange's one-layer assumption?  Config through its collapse as W slabs
in one process (boundary-first steps), status flags every 50 substeps, and the final records against a single engine.
usage: slab_long_run.py [config=2] [slabs=4] [substeps=400]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
PKG = "componentframeworks-smoothed-particle-hydrodynamics_amd"
pkg = importlib.import_module(PKG)
halo = importlib.import_module(PKG + ".halo")
syn = pkg.synthetic
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 2
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
cfg = syn.CONFIGS[ci]
rec, _ = syn.make_particles(cfg)
sp = pkg.default_params(**syn.params_fields(cfg))
g = pkg.compute_grid_extents(sp)
dims = tuple(int(v) for v in g.dims)
cz = np.clip(np.floor(((rec["pos"][:, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)).astype(np.float32)), 0, dims[2] - 1).astype(np.int64)
ids = np.arange(len(rec), dtype=np.uint32)
face = len(rec) // world + 65536
def make_engine(p, i, prm, z0, z1, lo, hi):
    return halo.HipSlabEngine(p, i, prm, z0, z1, lo, hi, capacity=int(len(rec) * 0.6) + 65536)
grp = halo.SlabGroup.from_particles(rec, ids, sp, dims, world, make_engine,
                                    lambda n: torch.zeros((n, halo.REC_WORDS), dtype=torch.float32, device="cuda"), face, cz)
grp.enable_overlap(face)
single = pkg.SPHFluidGPU.from_particles(rec, sp)
first_report = None
for s in range(steps):
    grp.DispatchCompute()
    single.DispatchCompute()
    if (s + 1) % 50 == 0 or s == steps - 1:
        msgs = []
        for x in grp.sims:
            try:
                st = x.engine.status()
                msgs.append(st + [x.engine.message_bytes()[:2]])      # flags + the bytes the last exchange sent to lo / hi (a whole face: message_bytes()[2])
            except pkg.SphError as ex:
                msgs.append(str(ex)[:90])
                first_report = first_report or s + 1
        got = halo.merge_into_records(rec, grp.download()) if first_report is None else None
        same = None
        if got is not None:
            ref = single.download()
            same = bool(got.tobytes() == ref.tobytes())
        print(s + 1, "rho_max/rho0 %.1f" % (float(single.download()["density"].max()) / 1000.0), "equal to the single engine:", same, msgs, flush=True)
        if first_report:
            break
print("first report at substep", first_report)
