"""Synthetic inputs for the BASELINE.json configs.

All configs: h = 0.28 (= cell size), rho0 = 1000, k = 2000, mu = 3.5, sigma = 0.0728,
g = (0, -980, 0), dt = 1e-3, box container, centre 0, Euler 0; cubic grid G^3 obtained with
boxHalf = (G/2 - 1) h - 0.01 h; particles on a jittered cubic lattice of spacing s filling the
full box footprint bottom-up (y slowest, then z, x fastest), stop at N, mass = rho0 s^3
(rule of SPHFluid3D.cpp:92), vel = 0, density = pressure = 0, flags 0.

Jitter comes from a counter-based hash of the global lattice index, so any z-slab of the
global configuration can be generated independently (multi-GPU ranks) and is identical to
the corresponding part of the single-domain configuration.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .engine import PARTICLE_DTYPE

H = 0.28


@dataclass(frozen=True)
class BenchConfig:
    index: int
    name: str
    n: int                 # particles (total over all GPUs)
    grid: tuple            # (gx, gy, gz) cells
    spacing_factor: float  # s / h
    gpus: int

    @property
    def seed(self) -> int:
        return 12345 + self.index


# BASELINE.json configs[0..4]; config 5 is weak-scaled: per-GPU slab 256 x 256 x 64 cells.
CONFIGS = {
    1: BenchConfig(1, "32k/32^3", 32768, (32, 32, 32), 0.85, 1),
    2: BenchConfig(2, "256k/64^3", 262144, (64, 64, 64), 0.85, 1),
    3: BenchConfig(3, "4M/128^3", 4194304, (128, 128, 128), 0.775, 1),
    4: BenchConfig(4, "16M/256^3", 16777216, (256, 256, 256), 0.85, 4),
}


def weak_config(n_gpus: int, per_gpu: int = 8388608, slab_cells=(256, 256, 64)) -> BenchConfig:
    """BASELINE.json configs[4] at n_gpus ranks: the global grid grows along z."""
    gx, gy, gz = slab_cells
    return BenchConfig(5, f"weak {per_gpu} x {n_gpus}", per_gpu * n_gpus, (gx, gy, gz * n_gpus), 0.775, n_gpus)


def box_half_for_grid(grid) -> np.ndarray:
    return np.array([(g / 2.0 - 1.0) * H - 0.01 * H for g in grid], np.float32)


def params_fields(cfg: BenchConfig) -> dict:
    """param_* values for a config (field names of SphParams / the reference members)."""
    s = np.float32(np.float32(cfg.spacing_factor) * np.float32(H))
    mass = np.float32(np.float32(1000.0) * s * s * s)
    return dict(
        param_h=H, param_mass=float(mass), param_restDensity=1000.0, param_gasConstant=2000.0,
        param_viscosity=3.5, param_gravityX=0.0, param_gravityY=-980.0, param_gravityZ=0.0,
        param_surfaceTension=0.0728, param_timeStep=1e-3, param_foamGen=1.0, param_foamVelRef=8.0,
        param_boxCenter=(0.0, 0.0, 0.0), param_boxHalf=tuple(float(x) for x in box_half_for_grid(cfg.grid)),
        param_boxEulerDeg=(0.0, 0.0, 0.0), param_shapeType=0, param_wallRestitution=0.15,
        param_wallFriction=0.02, grid_cap=max(160, max(cfg.grid)),
    )


def _hash_u01(idx: np.ndarray, axis: int, seed: int) -> np.ndarray:
    """splitmix64 finaliser on (index, axis, seed) -> uniform [0,1) with 24 bits."""
    with np.errstate(over="ignore"):
        x = idx.astype(np.uint64) * np.uint64(3) + np.uint64(axis)
        x = x + np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return ((x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def lattice_dims(cfg: BenchConfig):
    half = box_half_for_grid(cfg.grid)
    s = np.float32(np.float32(cfg.spacing_factor) * np.float32(H))
    nx, ny, nz = (int(np.floor(2.0 * float(hf) / float(s))) for hf in half)
    return (nx, ny, nz), s, half


def make_particles(cfg: BenchConfig, z_cells: tuple | None = None, jitter: float = 0.2):
    """Particle records of `cfg` (whole domain, or only the particles whose grid cell has
    z index in [z_cells[0], z_cells[1])).  Returns (records, global_ids)."""
    (nx, ny, nz), s, half = lattice_dims(cfg)
    cap = nx * ny * nz
    if cfg.n > cap:
        raise ValueError(f"config {cfg.name}: lattice capacity {cap} < N {cfg.n}")
    layer = nx * nz
    full_layers, rem = divmod(cfg.n, layer)
    # z-range of lattice nodes that can land in the requested cell slab (jitter < 1 node)
    zlo, zhi = 0, nz
    gmin_z = float(np.float32(0.0) - (np.float32(half[2]) + np.float32(H)))   # ComputeGridExtents in fp32
    if z_cells is not None:
        z0w = gmin_z + z_cells[0] * H
        z1w = gmin_z + z_cells[1] * H
        zlo = max(0, int(np.floor((z0w + float(half[2])) / float(s) - 0.5)) - 1)
        zhi = min(nz, int(np.ceil((z1w + float(half[2])) / float(s) - 0.5)) + 2)
    ny_used = full_layers + (1 if rem else 0)
    ys = np.arange(ny_used, dtype=np.int64)
    zs = np.arange(zlo, zhi, dtype=np.int64)
    xs = np.arange(nx, dtype=np.int64)
    Y, Z, X = np.meshgrid(ys, zs, xs, indexing="ij")
    gid = ((Y * nz + Z) * nx + X).ravel()
    keep = gid < cfg.n
    gid = gid[keep]
    X = X.ravel()[keep]; Y = Y.ravel()[keep]; Z = Z.ravel()[keep]
    amp = np.float32(jitter) * s

    def coord(k, hf, axis):
        base = (np.float32(-hf) + np.float32(0.5) * s + k.astype(np.float32) * s).astype(np.float32)
        u = _hash_u01(gid, axis, cfg.seed)
        return (base + (np.float32(2.0) * u - np.float32(1.0)) * amp).astype(np.float32)

    px, py, pz = coord(X, half[0], 0), coord(Y, half[1], 1), coord(Z, half[2], 2)
    if z_cells is not None:
        q = ((pz - np.float32(gmin_z)) / np.float32(H)).astype(np.float32)
        cz = np.clip(np.floor(q), 0, cfg.grid[2] - 1).astype(np.int64)
        m = (cz >= z_cells[0]) & (cz < z_cells[1])
        px, py, pz, gid = px[m], py[m], pz[m], gid[m]
    rec = np.zeros(len(gid), PARTICLE_DTYPE)
    rec["pos"][:, 0] = px; rec["pos"][:, 1] = py; rec["pos"][:, 2] = pz
    return rec, gid.astype(np.int64)
