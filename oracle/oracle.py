"""ctypes front end for the CPU oracle (oracle/sph_oracle.c) plus an independent
brute-force numpy restatement of the same shader math.

TEST INFRASTRUCTURE ONLY (see the header of sph_oracle.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package.
PARITY UNPINNED: the reference has no tests or fixtures for this path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

# 80-byte record, /root/reference/ComponentFramework/SPHFluid3D.h:12-24
PARTICLE_DTYPE = np.dtype(
    [
        ("pos", "<f4", (4,)),
        ("vel", "<f4", (4,)),
        ("acc", "<f4", (4,)),
        ("density", "<f4"),
        ("pressure", "<f4"),
        ("padA", "<f4"),
        ("padB", "<f4"),
        ("isGhost", "<i4"),
        ("isActive", "<i4"),
        ("padC", "<i4"),
        ("pad0", "<i4"),
    ]
)
assert PARTICLE_DTYPE.itemsize == 80


class OParams(C.Structure):
    """Mirror of OParams in sph_oracle.c (param_* members, SPHFluid3D.h:94-124)."""

    _fields_ = [
        ("h", C.c_float), ("mass", C.c_float), ("restDensity", C.c_float),
        ("gasConstant", C.c_float), ("viscosity", C.c_float),
        ("gravity", C.c_float * 3),
        ("surfaceTension", C.c_float), ("timeStep", C.c_float),
        ("pause", C.c_int32),
        ("useJitter", C.c_int32), ("jitterAmp", C.c_float),
        ("foamGen", C.c_float), ("foamVelRef", C.c_float),
        ("boxCenter", C.c_float * 3), ("boxHalf", C.c_float * 3), ("boxEulerDeg", C.c_float * 3),
        ("shapeType", C.c_int32), ("shapeAux", C.c_float * 3),
        ("mixPattern", C.c_int32), ("dyePattern", C.c_int32),
        ("wallRestitution", C.c_float), ("wallFriction", C.c_float),
        ("gridCap", C.c_int32),
    ]


class OGrid(C.Structure):
    _fields_ = [
        ("dims", C.c_int32 * 3),
        ("numCells", C.c_int32),
        ("gridMin", C.c_float * 3),
        ("cellSize", C.c_float),
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc only, no reference sources)."""
    src = os.path.join(_HERE, "sph_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        # SPH_ORACLE_LIB: another build of the same source (the Makefile's `ubsan` target) for a sanitizer run
        alt = os.environ.get("SPH_ORACLE_LIB")
        if not alt:
            build()
        L = C.CDLL(alt or _LIB_PATH)
        pp = C.POINTER(OParams)
        vp = C.c_void_p
        L.sph_oracle_sizeof_particle.restype = C.c_int
        L.sph_oracle_sizeof_params.restype = C.c_int
        L.sph_oracle_default_params.argtypes = [pp]
        L.sph_oracle_sinf.argtypes = [C.c_float]
        L.sph_oracle_sinf.restype = C.c_float
        L.sph_oracle_cosf.argtypes = [C.c_float]
        L.sph_oracle_cosf.restype = C.c_float
        L.sph_oracle_atan2f.argtypes = [C.c_float, C.c_float]
        L.sph_oracle_atan2f.restype = C.c_float
        L.sph_oracle_powf.argtypes = [C.c_float, C.c_float]
        L.sph_oracle_powf.restype = C.c_float
        L.sph_oracle_shape_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.sph_oracle_shape_table.restype = C.c_int
        L.sph_oracle_rotation.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.sph_oracle_effective_half.argtypes = [pp, C.POINTER(C.c_float)]
        L.sph_oracle_grid_extents.argtypes = [pp, C.POINTER(OGrid)]
        L.sph_oracle_build_grid.argtypes = [vp, C.c_int, C.POINTER(OGrid), vp, vp, vp, vp, vp]
        L.sph_oracle_obb.argtypes = [vp, C.c_int, pp]
        L.sph_oracle_sph_pass.argtypes = [vp, vp, C.c_int, pp, C.c_float]
        L.sph_oracle_substep.argtypes = [vp, vp, C.c_int, pp, C.c_float]
        L.sph_oracle_wave_impulse.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float,
                                              C.POINTER(C.c_float), C.c_float, C.c_float]
        L.sph_oracle_vortex_impulse.argtypes = [vp, C.c_int, pp, C.c_float, C.c_float]
        L.sph_oracle_attractor_impulse.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_float]
        L.sph_oracle_stencil_attract.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, C.c_float]
        L.sph_oracle_curl_flow.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float]
        L.sph_oracle_spawn.argtypes = [pp, C.c_int, C.c_uint32, vp, C.POINTER(C.c_float)]
        L.sph_oracle_spawn.restype = C.c_int
        L.sph_oracle_shape_supported.argtypes = [C.c_int]
        L.sph_oracle_shape_supported.restype = C.c_int
        L.sph_oracle_max_threads.restype = C.c_int
        L.sph_oracle_set_contract.argtypes = [C.c_int]
        L.sph_oracle_get_contract.restype = C.c_int
        L.sph_oracle_rsqrt.argtypes = [C.c_float]
        L.sph_oracle_rsqrt.restype = C.c_float
        L.sph_oracle_set_threads.argtypes = [C.c_int]
        assert L.sph_oracle_sizeof_particle() == 80
        assert L.sph_oracle_sizeof_params() == C.sizeof(OParams)
        _lib = L
    return _lib


def default_params(**overrides) -> OParams:
    p = OParams()
    lib().sph_oracle_default_params(C.byref(p))
    set_params(p, **overrides)
    return p


def set_params(p, **kw):
    for k, v in kw.items():
        cur = getattr(p, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(p, k, v)
    return p


def _ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def f3(x):
    return (C.c_float * 3)(*[float(v) for v in x])


def grid_extents(p: OParams) -> OGrid:
    g = OGrid()
    lib().sph_oracle_grid_extents(C.byref(p), C.byref(g))
    return g


def rotation(euler_deg) -> np.ndarray:
    out = (C.c_float * 9)()
    lib().sph_oracle_rotation(f3(euler_deg), out)
    return np.array(out, dtype=np.float32)


def effective_half(p: OParams) -> np.ndarray:
    out = (C.c_float * 3)()
    lib().sph_oracle_effective_half(C.byref(p), out)
    return np.array(out, dtype=np.float32)


def build_grid(P: np.ndarray, p: OParams, linked_list: bool = False):
    g = grid_extents(p)
    n = len(P)
    cell_start = np.zeros(g.numCells + 1, np.int32)
    order = np.zeros(max(n, 1), np.int32)
    pcell = np.zeros(max(n, 1), np.int32)
    head = np.zeros(g.numCells, np.int32) if linked_list else None
    nxt = np.zeros(max(n, 1), np.int32) if linked_list else None
    lib().sph_oracle_build_grid(_ptr(P), n, C.byref(g), _ptr(cell_start), _ptr(order), _ptr(pcell),
                                _ptr(head) if linked_list else None, _ptr(nxt) if linked_list else None)
    out = dict(grid=g, cell_start=cell_start, order=order[:n], particle_cell=pcell[:n])
    if linked_list:
        out.update(cell_head=head, particle_next=nxt[:n])
    return out


def sph_pass(P: np.ndarray, p: OParams, dt: float = -1.0) -> np.ndarray:
    out = np.zeros_like(P)
    lib().sph_oracle_sph_pass(_ptr(P), _ptr(out), len(P), C.byref(p), dt)
    return out


def obb(P: np.ndarray, p: OParams) -> np.ndarray:
    out = P.copy()
    lib().sph_oracle_obb(_ptr(out), len(out), C.byref(p))
    return out


class OFountain(C.Structure):
    """fountain* members of SPHFluidGPU (SPHFluid3D.h:161-168)."""
    _fields_ = [("mode", C.c_int32), ("offset", C.c_float * 3), ("radius", C.c_float), ("spread", C.c_float),
                ("jetSpeedLive", C.c_float), ("drainLevel", C.c_float), ("drainPerSec", C.c_float), ("seed", C.c_uint32)]


def default_fountain(**kw) -> OFountain:
    f = OFountain(0, (C.c_float * 3)(0.0, -5.0, 0.0), 1.0, 0.25, 25.0, 1.0, 2.0, 0)
    for k, v in kw.items():
        if k == "offset":
            for i in range(3):
                f.offset[i] = v[i]
        else:
            setattr(f, k, v)
    return f


def substep(P: np.ndarray, p: OParams, dt: float = -1.0, steps: int = 1, fountain: "OFountain | None" = None) -> np.ndarray:
    """DispatchCompute x steps; returns a new array.  `fountain` (its seed advances in place)
    adds step 6, the fountain recycle."""
    cur = P.copy()
    scratch = np.zeros_like(cur)
    L = lib()
    L.sph_oracle_substep_fountain.restype = None
    for _ in range(steps):
        if fountain is None:
            L.sph_oracle_substep(_ptr(cur), _ptr(scratch), len(cur), C.byref(p), C.c_float(dt))
        else:
            L.sph_oracle_substep_fountain(_ptr(cur), _ptr(scratch), len(cur), C.byref(p), C.c_float(dt), C.byref(fountain))
    return cur


def fountain_recycle(P: np.ndarray, p: OParams, f: OFountain, dt: float, seed: int) -> np.ndarray:
    out = P.copy()
    L = lib()
    L.sph_oracle_fountain.restype = None
    L.sph_oracle_fountain(_ptr(out), len(out), C.byref(p), C.byref(f), C.c_float(dt), C.c_uint32(seed))
    return out


class ORiver(C.Structure):
    """river / terrain members of SPHFluidGPU (SPHFluid3D.h:171-196), same names."""
    _fields_ = [("riverMode", C.c_int32), ("terrainW", C.c_int32), ("terrainH", C.c_int32),
                ("terrainWorldMinX", C.c_float), ("terrainWorldMinZ", C.c_float), ("terrainWorldSizeX", C.c_float), ("terrainWorldSizeZ", C.c_float),
                ("riverEmitterPos", C.c_float * 3), ("riverEmitterVel", C.c_float * 3), ("riverEmitterRadius", C.c_float),
                ("riverSinkY", C.c_float), ("riverSinkZMax", C.c_float), ("riverAmp", C.c_float), ("riverFreq", C.c_float),
                ("riverPhase", C.c_float), ("riverChannelWidth", C.c_float), ("riverChannelDepth", C.c_float), ("riverSlopeDrop", C.c_float)]


def default_river(**kw) -> ORiver:
    r = ORiver()
    L = lib()
    L.sph_oracle_river_default.restype = None
    L.sph_oracle_river_default(C.byref(r))
    for k, v in kw.items():
        setattr(r, k, v)
    return r


def river_terrain(p: OParams, seed: int, river: "ORiver | None" = None):
    """GenerateRiverTerrain (SPHFluid3D.cpp:772-878): fills the river members, returns (river, heights); writes p.gravity."""
    r = river if river is not None else default_river()
    heights = np.zeros(r.terrainW * r.terrainH, np.float32)
    L = lib()
    L.sph_oracle_river_terrain.restype = None
    L.sph_oracle_river_terrain(C.byref(p), C.c_int(seed), C.byref(r), heights.ctypes.data_as(C.c_void_p))
    return r, heights


def river_spawn(p: OParams, r: ORiver, heights: np.ndarray, n_requested: int, seed: int):
    """InitializeParticles, river branch (SPHFluid3D.cpp:104-160)."""
    out = np.zeros(max(n_requested, 1), PARTICLE_DTYPE)
    mass = C.c_float()
    h = np.ascontiguousarray(heights, np.float32)
    L = lib()
    L.sph_oracle_river_spawn.restype = C.c_int
    n = L.sph_oracle_river_spawn(C.byref(p), C.byref(r), h.ctypes.data_as(C.c_void_p), C.c_int(n_requested), C.c_uint32(seed), _ptr(out), C.byref(mass))
    return out[:n].copy(), float(mass.value)


def river_step(P: np.ndarray, p: OParams, r: ORiver, heights: np.ndarray) -> np.ndarray:
    """Step 5 of DispatchCompute alone: terrain, channel, emit."""
    out = P.copy()
    h = np.ascontiguousarray(heights, np.float32)
    L = lib()
    L.sph_oracle_river.restype = None
    L.sph_oracle_river(_ptr(out), len(out), C.byref(p), C.byref(r), h.ctypes.data_as(C.c_void_p))
    return out


def substep_river(P: np.ndarray, p: OParams, r: ORiver, heights: np.ndarray, dt: float = -1.0, steps: int = 1) -> np.ndarray:
    """DispatchCompute x steps in river mode (SPHFluid3D.cpp:431-522 with step 5, without the fountain step)."""
    cur = P.copy()
    scratch = np.zeros_like(cur)
    h = np.ascontiguousarray(heights, np.float32)
    L = lib()
    L.sph_oracle_substep_river.restype = None
    for _ in range(steps):
        L.sph_oracle_substep_river(_ptr(cur), _ptr(scratch), len(cur), C.byref(p), C.c_float(dt), C.byref(r), h.ctypes.data_as(C.c_void_p))
    return cur


def wave_impulse(P, amplitude, wavelength, phase, direction, y_min=-3.4028235e38, y_max=3.4028235e38):
    out = P.copy()
    lib().sph_oracle_wave_impulse(_ptr(out), len(out), amplitude, wavelength, phase, f3(direction), y_min, y_max)
    return out


def vortex_impulse(P, p: OParams, tangent_kick, inward_kick):
    out = P.copy()
    lib().sph_oracle_vortex_impulse(_ptr(out), len(out), C.byref(p), tangent_kick, inward_kick)
    return out


def attractor_impulse(P, point, pull_kick, radius):
    out = P.copy()
    lib().sph_oracle_attractor_impulse(_ptr(out), len(out), f3(point), pull_kick, radius)
    return out


def stencil_attract(P, targets, pull_kick, damp_kick):
    out = P.copy()
    t = np.ascontiguousarray(targets, dtype=np.float32).reshape(-1, 4)
    lib().sph_oracle_stencil_attract(_ptr(out), len(out), _ptr(t) if len(t) else None, len(t), pull_kick, damp_kick)
    return out


def curl_flow(P, kick, scale, time):
    out = P.copy()
    lib().sph_oracle_curl_flow(_ptr(out), len(out), kick, scale, time)
    return out


def spawn(p: OParams, n_requested: int, seed: int):
    """InitializeParticles standard fill; returns (particles, mass)."""
    buf = np.zeros(n_requested, PARTICLE_DTYPE)
    mass = C.c_float()
    n = lib().sph_oracle_spawn(C.byref(p), n_requested, seed, _ptr(buf), C.byref(mass))
    return buf[:n].copy(), float(mass.value)


def sinf(x: float) -> float:
    return float(lib().sph_oracle_sinf(float(x)))


def cosf(x: float) -> float:
    return float(lib().sph_oracle_cosf(float(x)))


def atan2f(y: float, x: float) -> float:
    return float(lib().sph_oracle_atan2f(float(y), float(x)))


def powf(x: float, p: float) -> float:
    return float(lib().sph_oracle_powf(float(x), float(p)))


def shape_table(params):
    """Sampled curve of container shapes 9 / 11 / 12 / 14: (points[k,3], best0[3])."""
    tab = np.zeros((128, 3), np.float32)
    b0 = np.zeros(3, np.float32)
    n = lib().sph_oracle_shape_table(C.byref(params), tab.ctypes.data_as(C.c_void_p), b0.ctypes.data_as(C.c_void_p))
    return tab[:n].copy(), b0


def set_contract(c: int):
    """1 (default): the engine's arithmetic contract; 0: the literal restatement (IEEE sqrt / division
    exactly where SPHFluid.comp has them).  See the comment above o_sph_one in sph_oracle.c."""
    lib().sph_oracle_set_contract(int(c))


def get_contract() -> int:
    return int(lib().sph_oracle_get_contract())


def rsqrt(x: float) -> float:
    return float(lib().sph_oracle_rsqrt(float(x)))


def set_threads(n: int):
    lib().sph_oracle_set_threads(n)


def max_threads() -> int:
    return lib().sph_oracle_max_threads()


# --------------------------------------------------------------------------------------
# Independent brute-force restatement (numpy).  No cell lists, no counting sort: every
# pair (i, j) is considered, filtered only by "j's cell is one of the 27 cells around
# i's ENTRY cell" (which the shader's traversal implies), and accumulated in the
# canonical order (ascending cell index of j, then ascending j).  fp32 throughout;
# fmaf is emulated through float64 (exact product, one extra rounding that can differ
# from a true fma in ~2^-29 of the cases, hence the 1-ulp-scale tolerance in the tests).
# Follows shaders/SPHFluid.comp:66-221 and OBBConstraints.comp:297-330 in the engine's arithmetic contract
# (contract 1 of sph_oracle.c: same sums, accept tests and order; rsqrt / factored constants as documented there).
# --------------------------------------------------------------------------------------

_F = np.float32


def _fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def _dot3(ax, ay, az, bx, by, bz):
    return _fma(az, bz, _fma(ay, by, (ax * bx).astype(np.float32)))


def _rsqrt(x):
    """sph_oracle_rsqrt on an fp32 array (integer seed + three Newton steps)."""
    x = np.asarray(x, _F)
    y = (np.uint32(0x5F3759DF) - (x.view(np.uint32) >> np.uint32(1))).view(_F)
    xh = (_F(0.5) * x).astype(_F)
    for _ in range(3):
        t = (y * y).astype(_F)
        e = _fma(-xh, t, _F(0.5))
        y = _fma(y, e, y)
    return y


def brute_force_sph_pass(P: np.ndarray, p: OParams, dt: float = -1.0) -> np.ndarray:
    n = len(P)
    g = grid_extents(p)
    dims = np.array(g.dims[:], np.int64)
    gmin = np.array(g.gridMin[:], _F)
    cs = _F(g.cellSize)
    dt = _F(dt if dt > 0 else p.timeStep)
    h = _F(p.h)
    h2 = _F(h * h)
    h3 = _F(h2 * h)
    h6 = _F(h3 * h3)
    h9 = _F(h6 * h3)
    pi_f = _F(3.141592653589)
    poly6C = _F(_F(315.0) / _F(_F(_F(64.0) * pi_f) * h9))
    spikyC = _F(_F(-45.0) / _F(pi_f * h6))
    viscC = _F(_F(45.0) / _F(pi_f * h6))
    mass = _F(p.mass)
    rho0 = _F(p.restDensity)
    kgas = _F(p.gasConstant)
    visc = _F(p.viscosity)
    sigma = _F(p.surfaceTension)
    grav = np.array(p.gravity[:], _F)
    max_speed = _F(_F(_F(0.4) * h) / max(dt, _F(1e-6)))

    pos = P["pos"][:, :3].astype(_F)
    vel = P["vel"][:, :3].astype(_F)
    rho_in = P["density"].astype(_F)
    prs_in = P["pressure"].astype(_F)

    q = ((pos - gmin) / cs).astype(_F)
    cc = np.clip(np.floor(q), 0, (dims - 1).astype(_F)).astype(np.int64)
    cell = (cc[:, 2] * dims[1] + cc[:, 1]) * dims[0] + cc[:, 0]

    # neighbour matrix in canonical order, -1 padded
    lists = []
    for i in range(n):
        adj = np.all(np.abs(cc - cc[i]) <= 1, axis=1)
        idx = np.nonzero(adj)[0]
        idx = idx[np.lexsort((idx, cell[idx]))]
        lists.append(idx)
    K = max(len(x) for x in lists)
    NB = np.full((n, K), -1, np.int64)
    for i, x in enumerate(lists):
        NB[i, : len(x)] = x
    ar = np.arange(n)

    mp6 = _F(mass * poly6C)
    nhm = _F(_F(-mass) * _F(0.5))
    max_speed2 = _F(max_speed * max_speed)
    inv_rho0 = _F(_F(1.0) / rho0)
    inv_foam = _F(_F(1.0) / max(_F(p.foamVelRef), _F(1e-3)))
    tiny = _F(1e-30)

    out = P.copy()
    # ---- sweep 1: density = (mass*poly6C) * sum max(h2 - r2, 0)^3
    dsum = np.zeros(n, _F)
    for k in range(K):
        j = NB[:, k]
        ok = j >= 0
        jj = np.where(ok, j, 0)
        d = (pos - pos[jj]).astype(_F)
        r2 = _dot3(d[:, 0], d[:, 1], d[:, 2], d[:, 0], d[:, 1], d[:, 2])
        t = np.where(ok, np.maximum((h2 - r2).astype(_F), _F(0)), _F(0)).astype(_F)
        dsum = _fma((t * t).astype(_F), t, dsum)
    dens = np.maximum((mp6 * dsum).astype(_F), _F(rho0 * _F(0.5)))
    prs = np.maximum((kgas * (dens - rho0).astype(_F)).astype(_F), _F(0))

    # ---- sweep 2
    fP = np.zeros((n, 3), _F)
    fV = np.zeros((n, 3), _F)
    gC = np.zeros((n, 3), _F)
    lapC = np.zeros(n, _F)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for k in range(K):
            j = NB[:, k]
            ok = (j >= 0) & (j != ar)
            jj = np.where(j >= 0, j, 0)
            d = (pos - pos[jj]).astype(_F)
            r2 = _dot3(d[:, 0], d[:, 1], d[:, 2], d[:, 0], d[:, 1], d[:, 2])
            rj = rho_in[jj]
            acc = ok & (r2 < h2) & (rj > 0)
            rinv = _rsqrt(np.maximum(r2, tiny))
            r = (r2 * rinv).astype(_F)
            hr = (h - r).astype(_F)
            sr = ((spikyC * (hr * hr).astype(_F)).astype(_F) * rinv).astype(_F)
            gW = (sr[:, None] * d).astype(_F)
            inv_rho = (_F(1.0) / rj).astype(_F)
            mor = (mass * inv_rho).astype(_F)
            pterm = (((prs + prs_in[jj]).astype(_F) * nhm).astype(_F) * inv_rho).astype(_F)
            ml = (mor * (viscC * hr).astype(_F)).astype(_F)
            for a in range(3):
                fP[:, a] = np.where(acc, _fma(gW[:, a], pterm, fP[:, a]), fP[:, a])
                fV[:, a] = np.where(acc, _fma((vel[jj, a] - vel[:, a]).astype(_F), ml, fV[:, a]), fV[:, a])
                gC[:, a] = np.where(acc, _fma(mor, gW[:, a], gC[:, a]), gC[:, a])
            lapC = np.where(acc, (lapC + ml).astype(_F), lapC)
        gl2 = _dot3(gC[:, 0], gC[:, 1], gC[:, 2], gC[:, 0], gC[:, 1], gC[:, 2])
        sc = np.where(gl2 > _F(1e-12), (((-sigma) * lapC).astype(_F) * _rsqrt(np.maximum(gl2, tiny))).astype(_F), _F(0)).astype(_F)
        fS = (sc[:, None] * gC).astype(_F)
    rr = _rsqrt(np.maximum(dens, tiny))
    inv_rho_i = (rr * rr).astype(_F)
    accv = np.zeros((n, 3), _F)
    nvel = np.zeros((n, 3), _F)
    npos = np.zeros((n, 3), _F)
    for a in range(3):
        fG = (grav[a] * dens).astype(_F)
        t = _fma(visc, fV[:, a], fP[:, a])
        t = (t + fG).astype(_F)
        t = (t + fS[:, a]).astype(_F)
        accv[:, a] = (t * inv_rho_i).astype(_F)
        v = _fma(accv[:, a], dt, vel[:, a])
        v = (v * _F(0.995)).astype(_F)
        nvel[:, a] = v
        npos[:, a] = _fma(v, dt, pos[:, a])

    # ---- sweep 3 (own state updated, neighbours at entry, entry cell)
    xs = np.zeros((n, 3), _F)
    norm = np.zeros(n, _F)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for k in range(K):
            j = NB[:, k]
            ok = (j >= 0) & (j != ar)
            jj = np.where(j >= 0, j, 0)
            d = (npos - pos[jj]).astype(_F)
            r2 = _dot3(d[:, 0], d[:, 1], d[:, 2], d[:, 0], d[:, 1], d[:, 2])
            rj = rho_in[jj]
            acc = ok & (r2 < h2) & (rj > 0)
            t = (h2 - r2).astype(_F)
            w3 = ((t * t).astype(_F) * t).astype(_F)
            wm = (w3 * (mass * (_F(1.0) / rj).astype(_F)).astype(_F)).astype(_F)
            for a in range(3):
                xs[:, a] = np.where(acc, _fma((vel[jj, a] - nvel[:, a]).astype(_F), wm, xs[:, a]), xs[:, a])
            norm = np.where(acc, (norm + w3).astype(_F), norm)
        nr = _rsqrt(np.maximum(norm, tiny))
        xs = np.where((norm > 0)[:, None], (xs * (nr * nr).astype(_F)[:, None]).astype(_F), xs).astype(_F)
    for a in range(3):
        nvel[:, a] = _fma(_F(0.12), xs[:, a], nvel[:, a])
    sp2 = _dot3(nvel[:, 0], nvel[:, 1], nvel[:, 2], nvel[:, 0], nvel[:, 1], nvel[:, 2])
    f = (max_speed * _rsqrt(np.maximum(sp2, tiny))).astype(_F)
    nvel = np.where((sp2 > max_speed2)[:, None], (nvel * f[:, None]).astype(_F), nvel).astype(_F)
    s2 = _dot3(nvel[:, 0], nvel[:, 1], nvel[:, 2], nvel[:, 0], nvel[:, 1], nvel[:, 2])
    speed = (s2 * _rsqrt(np.maximum(s2, tiny))).astype(_F)
    aer = (np.clip(((rho0 - dens).astype(_F) * inv_rho0).astype(_F), 0, 1).astype(_F)
           * np.clip((speed * inv_foam).astype(_F), 0, 1).astype(_F)).astype(_F)
    foam = np.maximum((aer * _F(p.foamGen)).astype(_F), (P["padA"] * _F(0.995)).astype(_F))

    fluid = P["isGhost"] != 1
    out["pos"][fluid, :3] = npos[fluid]
    out["vel"][fluid, :3] = nvel[fluid]
    out["acc"][fluid, :3] = accv[fluid]
    out["acc"][fluid, 3] = 0
    out["density"][fluid] = dens[fluid]
    out["pressure"][fluid] = prs[fluid]
    out["padA"][fluid] = foam[fluid]
    gh = (P["isGhost"] == 1) & (P["isActive"] != 0)
    out["vel"][gh] = 0
    out["acc"][gh] = 0
    out["density"][gh] = rho0
    out["pressure"][gh] = 0
    return out


def brute_force_obb_box(P: np.ndarray, p: OParams) -> np.ndarray:
    """OBBConstraints.comp box branch (:297-309) + response (:311-330), float64 math
    rounded to fp32 at the end (tolerance check only, not bit-exact)."""
    R = rotation(p.boxEulerDeg[:]).astype(np.float64).reshape(3, 3).T  # R[i, j]: row i, column j
    c = np.array(p.boxCenter[:], np.float64)
    half = np.array(p.boxHalf[:], np.float64)
    out = P.copy()
    pos = P["pos"][:, :3].astype(np.float64)
    vel = P["vel"][:, :3].astype(np.float64)
    pL = (pos - c) @ R
    qL = np.clip(pL, -half, half)
    delta = pL - qL
    d = np.abs(delta)
    hit = np.any(d > 0, axis=1) & (P["isGhost"] == 0)
    ax = np.where((d[:, 0] >= d[:, 1]) & (d[:, 0] >= d[:, 2]), 0, np.where((d[:, 1] >= d[:, 0]) & (d[:, 1] >= d[:, 2]), 1, 2))
    nL = np.zeros_like(pL)
    nL[np.arange(len(P)), ax] = np.sign(delta[np.arange(len(P)), ax])
    nW = nL @ R.T
    ln = np.linalg.norm(nW, axis=1, keepdims=True)
    nW = np.divide(nW, ln, out=np.zeros_like(nW), where=ln > 0)
    pW = c + qL @ R.T
    vn = np.sum(vel * nW, axis=1, keepdims=True)
    vN = vn * nW
    vT = vel - vN
    vnew = -p.wallRestitution * vN + (1.0 - p.wallFriction) * vT
    out["pos"][hit, :3] = pW[hit].astype(np.float32)
    out["vel"][hit, :3] = vnew[hit].astype(np.float32)
    return out
