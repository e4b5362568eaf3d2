#!/usr/bin/env python3
"""Two z-slab engines in ONE process on ONE GPU (device-to-device copies as the transport): per-substep wall time of
(a) the sequential schedule (pack -> hand-off -> unpack -> DispatchCompute, host synchronisation between the engines) and
(b) the boundary-first schedule (sph_slab_step_begin / sph_slab_step_finish_local: the exchange of the next substep on the
engines' second streams beside the interior of the SPH pass), against (c) one engine over the whole domain.
Both engines share the one GPU, so (a) and (b) measure the schedule, not a scaling: what (b) saves is the exchange and the
host synchronisation that (a) leaves on the critical path.
usage: slab_overlap.py [config index=3] [substeps=40] [slabs=2]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
halo = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd.halo")
syn = pkg.synthetic
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = syn.CONFIGS[ci]
sp = pkg.default_params(**syn.params_fields(cfg))
rec, _ = syn.make_particles(cfg)
g = pkg.compute_grid_extents(sp)
q = ((rec["pos"][:, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)).astype(np.float32)
cz = np.clip(np.floor(q), 0, g.dims[2] - 1).astype(np.int64)
ids = np.arange(len(rec), dtype=np.uint32)
face = int(len(rec) / g.dims[2] * 2.0) + 8192


def group():
    def make_engine(p, i, prm, z0, z1, lo, hi):
        return halo.HipSlabEngine(p, i, prm, z0, z1, lo, hi, capacity=int(len(p) * 1.5) + 4 * face)
    return halo.SlabGroup.from_particles(rec, ids, sp, tuple(g.dims), world, make_engine, lambda n: None, face, cz)


def run(sim):
    for _ in range(5):
        sim.DispatchCompute(-1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.DispatchCompute(-1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one = run(pkg.SPHFluidGPU.from_particles(rec, sp))
a = group(); a.enable_async(face)
seq = run(a)
st_a = [s.engine.status() for s in a.sims]
del a
b = group(); b.enable_overlap(face)
ovl = run(b)
st_b = [s.engine.status() for s in b.sims]
print(json.dumps({"config": cfg.name, "slabs": world, "substeps": steps, "one_engine_ms": round(one, 4), "slabs_sequential_schedule_ms": round(seq, 4),
                  "slabs_boundary_first_schedule_ms": round(ovl, 4), "face_capacity_records": face,
                  "records_packed_last_substep": [x[:2] for x in st_b], "error_flags": [x[4] for x in st_a + st_b]}))
