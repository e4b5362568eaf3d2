#!/usr/bin/env python3
"""Diagnose a failing seed of tests/test_gpu_fuzz.py::test_random_scene_as_z_slabs...: which exchange schedule, how many
slabs, and after how many substeps the slab group first differs from the oracle.  usage: slab_fuzz_diag.py seed [max steps]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import PKG_NAME, to_oracle_params
import test_gpu_fuzz as tf
from oracle import oracle
oracle.lib()
pkg = importlib.import_module(PKG_NAME)
halo = importlib.import_module(PKG_NAME + ".halo")
seed = int(sys.argv[1]); maxsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rec, sp, steps, what = tf._scene(pkg, seed)
print(what)
g = pkg.compute_grid_extents(sp)
dims = tuple(int(v) for v in g.dims)
print("grid", dims, "euler", list(sp.param_boxEulerDeg), "ghosts", int((rec["isGhost"] != 0).sum()))
op = to_oracle_params(oracle, sp)
q = ((rec["pos"][:, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)).astype(np.float32)
cz = np.clip(np.floor(q), 0, dims[2] - 1).astype(np.int64)
ids = np.arange(len(rec), dtype=np.uint32)
face = len(rec) + 1024
wants = [rec]
for s in range(maxsteps):
    wants.append(oracle.substep(wants[-1], op))
for world in (2, 3):
    if dims[2] < 2 * world: continue
    for mode in ("host", "async", "overlap"):
        def make_engine(p, i, prm, z0, z1, lo, hi):
            return halo.HipSlabEngine(p, i, prm, z0, z1, lo, hi, capacity=int(len(rec) * 1.2) + 8192)
        grp = halo.SlabGroup.from_particles(rec, ids, sp, dims, world, make_engine,
                                            lambda n: torch.zeros((n, halo.REC_WORDS), dtype=torch.float32, device="cuda"), face, cz)
        if mode == "async": grp.enable_async(face)
        if mode == "overlap": grp.enable_overlap(face)
        row = []
        for s in range(maxsteps):
            torch.cuda.synchronize()
            grp.DispatchCompute()
            got = halo.merge_into_records(rec, grp.download())
            w = wants[s + 1]
            bad = np.nonzero((got.view(np.uint8).reshape(len(got), -1) != w.view(np.uint8).reshape(len(w), -1)).any(axis=1))[0]
            row.append(len(bad))
            if len(bad) and sum(1 for r in row if r) == 1:
                i = int(bad[0])
                czi = int(np.clip(np.floor((w["pos"][i, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)), 0, dims[2] - 1))
                print(f"   first bad record {i}: cz {czi} pos {got['pos'][i]} vs {w['pos'][i]} density {got['density'][i]} vs {w['density'][i]} ghost {rec['isGhost'][i]}")
        st = [x.engine.status() for x in grp.sims] if mode != "host" else None
        print(f"world {world} mode {mode}: differing records after 1..{maxsteps} substeps: {row} status {st}")
        for x in grp.sims: x.engine.close()
single = pkg.SPHFluidGPU.from_particles(rec, sp)
single.DispatchN(maxsteps)
got = single.download()
print("single engine vs oracle:", int((got.view(np.uint8).reshape(len(got), -1) != wants[maxsteps].view(np.uint8).reshape(len(got), -1)).any(axis=1).sum()))
