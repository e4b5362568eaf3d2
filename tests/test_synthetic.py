"""Bench input generator: sizes of SURVEY.md section 8d and slab-independence."""
import numpy as np
import pytest


def test_config_geometry(pkg):
    syn = pkg.synthetic
    for idx, cap in ((1, 35 ** 3), (2, 72 ** 3), (3, 162 ** 3), (4, 298 ** 3)):
        cfg = syn.CONFIGS[idx]
        (nx, ny, nz), s, half = syn.lattice_dims(cfg)
        assert nx * ny * nz == cap and cfg.n <= cap
        sp = pkg.default_params(**syn.params_fields(cfg))
        g = pkg.compute_grid_extents(sp)
        assert tuple(g.dims) == cfg.grid
    w = syn.weak_config(8)
    assert w.n == 67108864 and w.grid == (256, 256, 512)
    sp = pkg.default_params(**syn.params_fields(w))
    assert tuple(pkg.compute_grid_extents(sp).dims) == w.grid


def test_config1_particles(pkg):
    syn = pkg.synthetic
    cfg = syn.CONFIGS[1]
    rec, gid = syn.make_particles(cfg)
    assert len(rec) == 32768 and np.array_equal(gid, np.arange(32768))
    half = syn.box_half_for_grid(cfg.grid)
    assert np.all(np.abs(rec["pos"][:, 0]) <= half[0]) and np.all(np.abs(rec["pos"][:, 2]) <= half[2])
    assert rec["pos"][:, 1].min() >= -half[1]
    assert np.all(rec["vel"] == 0) and np.all(rec["density"] == 0)
    rec2, _ = syn.make_particles(cfg)
    assert rec.tobytes() == rec2.tobytes()


def test_slabs_partition_the_domain(pkg):
    """Each z-slab generated alone equals the matching subset of the whole domain."""
    syn = pkg.synthetic
    cfg = syn.BenchConfig(7, "t", 20000, (24, 24, 32), 0.8, 4)
    full, gid = syn.make_particles(cfg)
    sp = pkg.default_params(**syn.params_fields(cfg))
    g = pkg.compute_grid_extents(sp)
    cz = np.clip(np.floor((full["pos"][:, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)), 0, g.dims[2] - 1).astype(int)
    seen = 0
    for r in range(4):
        z0, z1 = 8 * r, 8 * (r + 1)
        part, pid = syn.make_particles(cfg, z_cells=(z0, z1))
        m = (cz >= z0) & (cz < z1)
        assert np.array_equal(pid, gid[m])
        assert part.tobytes() == full[m].tobytes()
        seen += len(part)
    assert seen == cfg.n
