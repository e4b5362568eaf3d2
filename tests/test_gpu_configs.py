"""BASELINE.json configs[3] and configs[4] on the HIP path (SURVEY.md section 8(d) "config 4" / "config 5").

Both are far too big for the CPU oracle inside a test, so they are checked through size-independent
properties: the SPH passes (k_sph_walk, k_sph_slow) agree bit for bit, a z-slab group of engines agrees
bit for bit with the single engine, particles stay inside the container, the velocity cap holds.  The
arithmetic itself is pinned against the oracle at small sizes in test_gpu_parity.py / test_gpu_slab.py.
"""
import importlib

import numpy as np
import pytest

from conftest import PKG_NAME, assert_records_equal

pytestmark = pytest.mark.gpu


def _engine(pkg, rec, sp, neighbor):
    f = pkg.SPHFluidGPU.from_particles(rec, sp)
    f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, neighbor)
    return f


def _cell_z(pkg, sp, P):
    g = pkg.compute_grid_extents(sp)
    q = ((P["pos"][:, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)).astype(np.float32)
    return np.clip(np.floor(q), 0, g.dims[2] - 1).astype(np.int64), tuple(g.dims)


def _group(pkg, halo, P, sp, world):
    import torch
    cz, dims = _cell_z(pkg, sp, P)
    ids = np.arange(len(P), dtype=np.uint32)
    face = int(len(P) / dims[2] * 4) + 8192

    def make_engine(p, i, prm, z0, z1, lo, hi):
        return halo.HipSlabEngine(p, i, prm, z0, z1, lo, hi, capacity=int(len(p) * 1.2) + 4 * face)

    return halo.SlabGroup.from_particles(P, ids, sp, dims, world, make_engine,
                                         lambda n: torch.zeros((n, halo.REC_WORDS), dtype=torch.float32, device="cuda"), face, cz)


def _properties(syn, cfg, out, dt=1e-3):
    half = syn.box_half_for_grid(cfg.grid)
    assert np.isfinite(out["pos"]).all() and np.isfinite(out["vel"]).all()
    assert np.all(np.abs(out["pos"][:, :3]) <= half[None, :] + 1e-4)                       # OBBConstraints: box container
    assert np.linalg.norm(out["vel"][:, :3].astype(np.float64), axis=1).max() <= 0.4 * 0.28 / dt * (1 + 1e-6)   # SPHFluid3D.cpp:488
    assert out["density"].min() >= 500.0


def test_config4_16m_256cubed(pkg):
    """BASELINE.json configs[3]: 16 777 216 particles, 256^3 cells (spacing 0.85 h), z-slabs of 64 cell layers."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    syn = pkg.synthetic
    cfg = syn.CONFIGS[4]
    rec, _ = syn.make_particles(cfg)
    assert len(rec) == 16777216
    sp = pkg.default_params(**syn.params_fields(cfg))
    g = pkg.compute_grid_extents(sp)
    assert tuple(g.dims) == (256, 256, 256)
    outs = []
    for neighbor in (3, 1):
        f = _engine(pkg, rec, sp, neighbor)
        f.DispatchN(3)
        outs.append(f.download())
        f.close()
    assert_records_equal(outs[0], outs[1], "k_sph_walk vs k_sph_slow at 16M / 256^3")
    _properties(syn, cfg, outs[0])
    grp = _group(pkg, halo, rec, sp, 4)                   # the 4 x 64-layer decomposition of the config, in one process
    assert [(s.z0, s.z1) for s in grp.sims] == [(0, 64), (64, 128), (128, 192), (192, 256)]
    for _ in range(3):
        grp.DispatchCompute()
    got = halo.merge_into_records(rec, grp.download())
    for s in grp.sims:
        s.engine.close()
    assert_records_equal(got, outs[0], "4 z-slabs vs one engine at 16M / 256^3")


def test_config5_slab_8m_obb_wave(pkg):
    """BASELINE.json configs[4], one GPU's share: 8 388 608 particles in a 256 x 256 x 64-cell slab (spacing
    0.775 h), OBBConstraints on, ApplyWaveImpulse every 16th substep with the scene's values (A = 1.5,
    lambda = 3, +Y, phase += 4 * 16 * dt; Scene0p.h:144-147), 33 substeps (three impulses)."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    syn = pkg.synthetic
    cfg = syn.weak_config(1)
    rec, _ = syn.make_particles(cfg)
    assert len(rec) == 8388608 and cfg.grid == (256, 256, 64)
    sp = pkg.default_params(**syn.params_fields(cfg))
    steps = 33

    def run(obj):
        phase = 0.0
        for s in range(steps):
            if s % 16 == 0:
                obj.ApplyWaveImpulse(1.5, 3.0, phase, (0.0, 1.0, 0.0))
                phase += 4.0 * 16 * 1e-3
            obj.DispatchCompute()

    outs = []
    for neighbor in (3, 1):
        f = _engine(pkg, rec, sp, neighbor)
        run(f)
        outs.append(f.download())
        f.close()
    assert_records_equal(outs[0], outs[1], "k_sph_walk vs k_sph_slow on the configs[4] slab")
    _properties(syn, cfg, outs[0])
    assert np.abs(outs[0]["vel"][:, 1]).max() > 1.0       # the impulses and gravity did act
    grp = _group(pkg, halo, rec, sp, 2)
    run(grp)
    got = halo.merge_into_records(rec, grp.download())
    for s in grp.sims:
        s.engine.close()
    assert_records_equal(got, outs[0], "2 z-slabs vs one engine on the configs[4] slab")
