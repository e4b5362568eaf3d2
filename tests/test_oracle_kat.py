"""Known-answer tests for the CPU oracle, derived from the shader formulas
(SURVEY.md section 8c list (1)-(11)); no reference run is possible (parity unpinned)."""
import math

import numpy as np
import pytest

PI_F = np.float32(3.141592653589)


def one(o, pos, vel=(0, 0, 0), **kw):
    P = np.zeros(1, o.PARTICLE_DTYPE)
    P["pos"][0, :3] = pos
    P["vel"][0, :3] = vel
    for k, v in kw.items():
        P[k][0] = v
    return P


def test_isolated_particle_density(oracle):
    """(1) rho = max(m*315/(64 pi h^3), rho0/2), P = max(k(rho-rho0),0); spawn mass => 0.9621 rho0."""
    p = oracle.default_params()
    s = 0.85 * p.h
    p.mass = p.restDensity * s ** 3
    out = oracle.sph_pass(one(oracle, (0, 0, 0)), p)
    expect = p.mass * 315.0 / (64.0 * math.pi * p.h ** 3)
    assert out["density"][0] == pytest.approx(expect, rel=2e-6)
    assert out["density"][0] / p.restDensity == pytest.approx(0.9621, abs=2e-4)
    assert out["pressure"][0] == 0.0


def test_lattice_interior_density(oracle):
    """(2) cubic lattice at 0.85h: six neighbours at 0.85h => rho = 0.9621 (1 + 6 (1-0.7225)^3) rho0."""
    p = oracle.default_params()
    s = np.float32(0.85) * np.float32(p.h)
    p.mass = float(np.float32(p.restDensity) * s * s * s)
    g = np.arange(-3, 4)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    P = np.zeros(X.size, oracle.PARTICLE_DTYPE)
    P["pos"][:, 0] = X.ravel() * s
    P["pos"][:, 1] = Y.ravel() * s
    P["pos"][:, 2] = Z.ravel() * s
    out = oracle.sph_pass(P, p)
    centre = int(np.nonzero((X.ravel() == 0) & (Y.ravel() == 0) & (Z.ravel() == 0))[0][0])
    w0 = 0.9621 * (1 + 6 * (1 - 0.7225) ** 3)
    assert out["density"][centre] / p.restDensity == pytest.approx(w0, rel=3e-4)


def test_pair_forces_equal_and_opposite(oracle):
    """(3) two particles, equal rho/P: pressure accelerations are opposite; spiky magnitude."""
    p = oracle.default_params(gravity=(0, 0, 0), viscosity=0.0, surfaceTension=0.0)
    d = 0.6 * p.h
    P = np.zeros(2, oracle.PARTICLE_DTYPE)
    P["pos"][1, 0] = d
    P["density"][:] = 1200.0
    P["pressure"][:] = 50.0
    out = oracle.sph_pass(P, p)
    assert out["acc"][0, 0] == pytest.approx(-out["acc"][1, 0], rel=1e-6)
    assert out["acc"][0, 0] < 0 < out["acc"][1, 0]          # repulsive
    # own P is this substep's k(rho_i - rho0); the neighbour's rho/P are the STORED 1200 / 50
    rho_i, p_i = float(out["density"][0]), float(out["pressure"][0])
    assert p_i == pytest.approx(p.gasConstant * (rho_i - p.restDensity), rel=1e-6)
    grad = 45.0 / (math.pi * p.h ** 6) * (p.h - d) ** 2
    expect = grad * p.mass * (p_i + 50.0) / (2.0 * 1200.0) / rho_i
    assert abs(out["acc"][0, 0]) == pytest.approx(expect, rel=1e-5)
    assert np.all(out["acc"][:, 1:3] == 0)


def test_first_substep_gravity_only(oracle):
    """(4) after reset every rho_j = 0 => pair terms skipped: acc = g, vel = g dt 0.995, pos += vel dt."""
    p = oracle.default_params()
    rng = np.random.default_rng(1)
    P = np.zeros(200, oracle.PARTICLE_DTYPE)
    P["pos"][:, :3] = rng.uniform(-1, 1, (200, 3)).astype(np.float32)
    out = oracle.sph_pass(P, p)
    g = np.float32(p.gravity[1])
    dt = np.float32(p.timeStep)
    np.testing.assert_allclose(out["acc"][:, 1], g, rtol=1e-6)
    assert np.all(out["acc"][:, 0] == 0) and np.all(out["acc"][:, 2] == 0) and np.all(out["acc"][:, 3] == 0)
    v = np.float32(np.float32(g * dt) * np.float32(0.995))
    np.testing.assert_allclose(out["vel"][:, 1], v, rtol=1e-6)
    np.testing.assert_allclose(out["pos"][:, 1], P["pos"][:, 1] + v * dt, rtol=0, atol=1e-6)
    assert np.array_equal(out["pos"][:, 0], P["pos"][:, 0])


def test_velocity_cap(oracle):
    """(5) |v| <= 0.4 h / dt."""
    p = oracle.default_params(gravity=(0, 0, 0))
    out = oracle.sph_pass(one(oracle, (0, 0, 0), vel=(500.0, 0, 0)), p)
    cap = 0.4 * p.h / p.timeStep
    assert np.linalg.norm(out["vel"][0, :3]) == pytest.approx(cap, rel=1e-6)
    # dt override changes the cap (SPHFluid3D.cpp:488 uses the override)
    out = oracle.sph_pass(one(oracle, (0, 0, 0), vel=(500.0, 0, 0)), p, dt=2e-3)
    assert np.linalg.norm(out["vel"][0, :3]) == pytest.approx(0.4 * p.h / 2e-3, rel=1e-6)


def test_foam(oracle):
    """(6) padA <- max(aer*foamGen, 0.995 padA)."""
    p = oracle.default_params(gravity=(0, 0, 0))
    out = oracle.sph_pass(one(oracle, (0, 0, 0), vel=(4.0, 0, 0), padA=0.9), p)
    rho = out["density"][0]
    speed = np.linalg.norm(out["vel"][0, :3])
    aer = min(max((p.restDensity - rho) / p.restDensity, 0), 1) * min(max(speed / p.foamVelRef, 0), 1)
    assert out["padA"][0] == pytest.approx(max(aer * p.foamGen, 0.9 * 0.995), rel=1e-6)
    out = oracle.sph_pass(one(oracle, (0, 0, 0), vel=(0, 0, 0), padA=0.5), p)
    assert out["padA"][0] == pytest.approx(0.5 * 0.995, rel=1e-6)


def test_build_grid(oracle):
    """(7) cell = (cz*gy+cy)*gx+cx with clamp; every particle exactly once in its cell's list."""
    p = oracle.default_params()
    rng = np.random.default_rng(3)
    n = 5000
    P = np.zeros(n, oracle.PARTICLE_DTYPE)
    P["pos"][:, :3] = rng.uniform(-9, 9, (n, 3)).astype(np.float32)   # some outside the grid: clamped
    b = oracle.build_grid(P, p, linked_list=True)
    g = b["grid"]
    dims = np.array(g.dims[:])
    assert list(dims) == [52, 52, 52]
    q = np.floor((P["pos"][:, :3] - np.array(g.gridMin[:], np.float32)) / np.float32(g.cellSize))
    cc = np.clip(q, 0, dims - 1).astype(np.int64)
    cell = (cc[:, 2] * dims[1] + cc[:, 1]) * dims[0] + cc[:, 0]
    assert np.array_equal(cell, b["particle_cell"])
    cs = b["cell_start"]
    assert cs[0] == 0 and cs[-1] == n and np.all(np.diff(cs) >= 0)
    assert np.array_equal(np.sort(b["order"]), np.arange(n))
    assert np.array_equal(cell[b["order"]], np.repeat(np.arange(g.numCells), np.diff(cs)))
    for c in np.unique(cell)[:200]:                                     # ascending index inside a cell
        seg = b["order"][cs[c]:cs[c + 1]]
        assert np.all(np.diff(seg) > 0)
    # the shader's own linked list: walk every list, each particle seen exactly once
    seen = np.zeros(n, np.int32)
    for c in range(g.numCells):
        j = b["cell_head"][c]
        while j != -1:
            assert cell[j] == c
            seen[j] += 1
            j = b["particle_next"][j]
    assert np.all(seen == 1)


def test_grid_extents(oracle):
    """(8) defaults => gridMin = -7.28, 52^3; 45 deg Y rotation => ext_x = ext_z = 7 sqrt2 + 0.28."""
    p = oracle.default_params()
    g = oracle.grid_extents(p)
    assert list(g.dims) == [52, 52, 52] and g.numCells == 140608
    np.testing.assert_allclose(g.gridMin[:], -7.28, rtol=1e-6)
    assert g.cellSize == pytest.approx(0.28)
    p = oracle.default_params(boxEulerDeg=(0, 45, 0))
    g = oracle.grid_extents(p)
    e = 7 * math.sqrt(2) + 0.28
    assert g.gridMin[0] == pytest.approx(-e, rel=1e-6) and g.gridMin[2] == pytest.approx(-e, rel=1e-6)
    assert g.gridMin[1] == pytest.approx(-7.28, rel=1e-6)
    assert g.dims[0] == math.ceil(2 * e / 0.28) and g.dims[1] == 52
    p = oracle.default_params(boxHalf=(40, 7, 7))                     # cap 160 (SPHFluid3D.cpp:370)
    assert oracle.grid_extents(p).dims[0] == 160
    p = oracle.default_params(boxHalf=(40, 7, 7), gridCap=512)
    assert oracle.grid_extents(p).dims[0] == math.ceil(2 * 40.28 / 0.28)


def test_effective_half(oracle):
    p = oracle.default_params(boxHalf=(3, 2, 1), shapeAux=(5, 0.35, 2.5))
    expect = {0: (3, 2, 1), 1: (3, 3, 3), 2: (3, 2, 3), 3: (5, 2, 5), 4: (3, 5, 3), 5: (3, 2, 3), 6: (3, 2, 3),
              7: (3, 2, 3), 8: (3, 2, 3), 9: (11, 3.05, 11), 10: (5, 7, 5), 11: (5, 2.35, 5), 12: (5, 5.45, 2.3),
              13: (3, 3, 3), 14: (5, 2.35, 5)}
    for shape, e in expect.items():
        p.shapeType = shape
        np.testing.assert_allclose(oracle.effective_half(p), e, rtol=1e-6)


def test_obb_box(oracle):
    """(9) point at local (half.x+d, 0, 0), v=(vx,vy,0) => p.x = half.x, v = (-e vx, (1-f) vy, 0)."""
    p = oracle.default_params()
    P = one(oracle, (7.3, 1.0, -2.0), vel=(5.0, 3.0, 0.0))
    out = oracle.obb(P, p)
    np.testing.assert_allclose(out["pos"][0, :3], (7.0, 1.0, -2.0), rtol=1e-6)
    np.testing.assert_allclose(out["vel"][0, :3], (-0.15 * 5.0, 0.98 * 3.0, 0.0), rtol=1e-6, atol=1e-7)
    # applied whenever outside, whatever the sign of v.n
    out = oracle.obb(one(oracle, (7.3, 0, 0), vel=(-5.0, 0, 0)), p)
    assert out["vel"][0, 0] == pytest.approx(0.75, rel=1e-6)
    # inside: untouched; ghost: untouched
    P = one(oracle, (1, 2, 3), vel=(1, 1, 1))
    assert oracle.obb(P, p).tobytes() == P.tobytes()
    P = one(oracle, (9, 0, 0), vel=(1, 1, 1), isGhost=2)
    assert oracle.obb(P, p).tobytes() == P.tobytes()
    # rotated box against the float64 restatement
    p = oracle.default_params(boxEulerDeg=(20, 35, -50), boxCenter=(0.5, -0.25, 1.0), boxHalf=(3, 2, 1))
    rng = np.random.default_rng(5)
    P = np.zeros(2000, oracle.PARTICLE_DTYPE)
    P["pos"][:, :3] = rng.uniform(-5, 5, (2000, 3)).astype(np.float32)
    P["vel"][:, :3] = rng.uniform(-9, 9, (2000, 3)).astype(np.float32)
    a, b = oracle.obb(P, p), oracle.brute_force_obb_box(P, p)
    np.testing.assert_allclose(a["pos"][:, :3], b["pos"][:, :3], rtol=0, atol=2e-5)
    np.testing.assert_allclose(a["vel"][:, :3], b["vel"][:, :3], rtol=0, atol=2e-4)


def test_obb_shapes_project_inside(oracle):
    """Shapes 1..6: projected points lie on/inside the container, inside points untouched."""
    rng = np.random.default_rng(11)
    P = np.zeros(4000, oracle.PARTICLE_DTYPE)
    P["pos"][:, :3] = rng.uniform(-6, 6, (4000, 3)).astype(np.float32)
    P["vel"][:, :3] = rng.uniform(-3, 3, (4000, 3)).astype(np.float32)
    x, y, z = (P["pos"][:, i].astype(np.float64) for i in range(3))

    def inside(shape, X, Y, Z, tol):
        rxz = np.hypot(X, Z)
        if shape == 1: return np.sqrt(X * X + Y * Y + Z * Z) <= 3 + tol
        if shape == 2: return (rxz <= 3 + tol) & (np.abs(Y) <= 2 + tol)
        if shape == 3: return np.hypot(rxz - 3, Y) <= 2 + tol
        if shape == 4: return np.sqrt(X * X + Z * Z + (Y - np.clip(Y, -2, 2)) ** 2) <= 3 + tol
        if shape == 5: return (np.abs(Y) <= 2 + tol) & (rxz <= 1 + (3 - 1) * np.abs(np.clip(Y, -2, 2)) / 2 + tol)
        if shape == 6: return (X / 3) ** 2 + (Y / 2) ** 2 + (Z / 3) ** 2 <= 1 + tol
    for shape in range(1, 7):
        p = oracle.default_params(shapeType=shape, boxHalf=(3, 2, 1))
        assert oracle.lib().sph_oracle_shape_supported(shape)
        out = oracle.obb(P, p)
        X, Y, Z = (out["pos"][:, i].astype(np.float64) for i in range(3))
        was_in = inside(shape, x, y, z, 0.0)
        assert np.all(inside(shape, X, Y, Z, 1e-4)), shape
        same = out["pos"][:, :3] == P["pos"][:, :3]
        assert np.all(same[was_in & ~np.isclose((x / 3) ** 2 + (y / 2) ** 2 + (z / 3) ** 2, 1)].all(axis=1) | (shape == 6)), shape
        assert np.any(~was_in)


def test_wave_impulse(oracle):
    """(10) dv = n A sin(2 pi/lambda x.n + phi); outside the y band untouched; host early-outs."""
    rng = np.random.default_rng(2)
    P = np.zeros(500, oracle.PARTICLE_DTYPE)
    P["pos"][:, :3] = rng.uniform(-5, 5, (500, 3)).astype(np.float32)
    P["isGhost"][::50] = 1
    A, lam, phi, d = 1.5, 3.0, 0.4, (1.0, 2.0, -0.5)
    out = oracle.wave_impulse(P, A, lam, phi, d, -1.0, 2.0)
    n = np.array(d) / np.linalg.norm(d)
    theta = 2 * math.pi / lam * (P["pos"][:, :3].astype(np.float64) @ n) + phi
    band = (P["pos"][:, 1] >= -1.0) & (P["pos"][:, 1] <= 2.0) & (P["isGhost"] == 0)
    dv = np.where(band[:, None], n[None, :] * (A * np.sin(theta))[:, None], 0.0)
    np.testing.assert_allclose(out["vel"][:, :3], dv, rtol=0, atol=3e-6)
    assert oracle.wave_impulse(P, 0.0, lam, phi, d).tobytes() == P.tobytes()
    assert oracle.wave_impulse(P, A, 1e-7, phi, d).tobytes() == P.tobytes()
    out = oracle.wave_impulse(P, A, lam, phi, (0, 0, 0))              # degenerate dir -> +Y
    assert np.all(out["vel"][:, 0] == 0) and np.any(out["vel"][:, 1] != 0)


def test_sinf_accuracy(oracle):
    xs = np.concatenate([np.linspace(-100, 100, 20001), np.linspace(-1e4, 1e4, 2001), [0.0, 1e-8, -1e-8]]).astype(np.float32)
    got = np.array([oracle.sinf(float(x)) for x in xs], np.float64)
    ref = np.sin(xs.astype(np.float64))
    err = np.abs(got - ref)
    assert err[np.abs(xs) <= 100].max() < 2.5e-7
    assert err.max() < 2e-6
    assert oracle.sinf(0.0) == 0.0


def test_poly6_normalisation(oracle):
    """(11) sanity: integral of W over the support is 1 (numerical quadrature of the formula
    the density sweep uses: density of a uniform continuum of number density n equals m n)."""
    h = 0.28
    r = np.linspace(0, h, 200001)
    w = 315.0 / (64 * math.pi * h ** 9) * (h * h - r * r) ** 3
    assert np.trapezoid(w * 4 * math.pi * r * r, r) == pytest.approx(1.0, rel=1e-6)


def test_ghost_branch(oracle):
    """SPHFluid.comp:72-83."""
    p = oracle.default_params()
    P = one(oracle, (0, 0, 0), vel=(1, 2, 3), isGhost=1, isActive=0, density=5.0, pressure=6.0, padA=0.3)
    P["vel"][0, 3] = 9.0
    assert oracle.sph_pass(P, p).tobytes() == P.tobytes()
    P["isActive"][0] = 1
    out = oracle.sph_pass(P, p)
    assert np.all(out["vel"][0] == 0) and np.all(out["acc"][0] == 0)
    assert out["density"][0] == p.restDensity and out["pressure"][0] == 0 and out["padA"][0] == np.float32(0.3)
    assert np.array_equal(out["pos"], P["pos"])


def test_pause_and_empty(oracle):
    p = oracle.default_params(pause=1)
    P = one(oracle, (0, 0, 0))
    assert oracle.substep(P, p).tobytes() == P.tobytes()
    p.pause = 0
    assert len(oracle.substep(np.zeros(0, oracle.PARTICLE_DTYPE), p)) == 0


def test_spawn(oracle):
    """InitializeParticles standard fill: count = min(requested, lattice capacity), mass rule, block extents."""
    p = oracle.default_params()
    P, mass = oracle.spawn(p, 50000, seed=3)
    assert len(P) == 50000
    s = np.float32(0.28) * np.float32(0.85)
    assert mass == pytest.approx(float(np.float32(1000.0) * s * s * s), rel=1e-6)
    P2, _ = oracle.spawn(p, 100000, seed=3)
    assert len(P2) == 50 * 23 * 50
    assert P2["pos"][:, 1].min() >= -7 + s * (1 - 0.2) - 1e-5
    assert np.all(P2["vel"] == 0) and np.all(P2["density"] == 0) and np.all(P2["isGhost"] == 0)
    assert set(np.unique(P2["padC"])) == {0, 1}
    assert P2["padB"].min() >= 0 and P2["padB"].max() <= 1
    Q, _ = oracle.spawn(p, 50000, seed=3)
    assert Q.tobytes() == P.tobytes()
    Q, _ = oracle.spawn(p, 50000, seed=4)
    assert Q.tobytes() != P.tobytes()
    p.useJitter = 0
    Q, _ = oracle.spawn(p, 1000, seed=4)
    assert np.allclose(np.diff(Q["pos"][:50, 2]), s, rtol=1e-5)


# ---- container shapes 7..14 (OBBConstraints.comp:144-296) ---------------------------------

def test_pinned_cos_atan2_pow_accuracy(oracle):
    """The fixed fp32 cos / atan2 / pow (oracle semantic 10) against float64 libm."""
    rng = np.random.default_rng(3)
    xs = rng.uniform(-40, 40, 4000).astype(np.float32)
    assert max(abs(oracle.cosf(x) - np.cos(np.float64(x))) for x in xs) < 2e-7
    ys, xx = rng.normal(size=4000).astype(np.float32), rng.normal(size=4000).astype(np.float32)
    assert max(abs(oracle.atan2f(y, x) - np.arctan2(np.float64(y), np.float64(x))) for y, x in zip(ys, xx)) < 5e-7
    b, pw = rng.uniform(1e-6, 30, 4000).astype(np.float32), rng.uniform(-3, 8, 4000).astype(np.float32)
    for x, p in zip(b, pw):
        r = np.float64(x) ** np.float64(p)
        if 1e-30 < r < 1e30:
            assert abs(oracle.powf(x, p) - r) / r < 5e-6
    assert oracle.powf(0.0, 2.0) == 0.0 and oracle.powf(1.0, 7.0) == 1.0 and oracle.powf(3.0, 0.0) == 1.0
    assert oracle.atan2f(0.0, 0.0) == 0.0 and oracle.atan2f(1.0, 0.0) == np.float32(np.pi / 2)
    assert oracle.atan2f(0.0, -1.0) == np.float32(np.pi) and oracle.cosf(0.0) == 1.0


def _ext_shape_float64(shape, half, aux, p):
    """Independent float64 numpy statement of the shader's projection for shapes 7..14:
    returns (hit, q) per point."""
    hx, hy = half[0], half[1]
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    q = p.copy()
    hit = np.zeros(len(p), bool)

    def tube(c, best0, r):
        d2 = ((p[:, None, :] - c[None, :, :]) ** 2).sum(-1)
        k = d2.argmin(1)
        best = c[k]
        d = p - best
        dl = np.sqrt((d * d).sum(1))
        h = dl > r
        nl = d / np.maximum(dl, 1e-6)[:, None]
        return h, np.where(h[:, None], best + nl * r, p), np.sort(d2, 1)

    if shape == 7:
        pts, depth = max(3.0, aux[0]), np.clip(aux[1], 0.0, 0.9)
        ang = np.arctan2(z, x)
        rmax = hx * (1 - depth * (0.5 + 0.5 * np.cos(pts * ang)))
        lxz = np.hypot(x, z)
        s = np.where(lxz > rmax, rmax / np.maximum(lxz, 1e-6), 1.0)
        q = np.stack([x * s, np.clip(y, -hy, hy), z * s], 1)
        hit = np.sqrt(((p - q) ** 2).sum(1)) > 1e-6
    elif shape == 8:
        a, b, n = max(hx, 1e-6), max(hy, 1e-6), np.clip(aux[2], 0.6, 8.0)
        e = np.array([a, b, a])
        F = ((np.abs(p) / e) ** n).sum(1)
        hit = F > 1
        q = np.where(hit[:, None], p * (F ** (-1.0 / n))[:, None], p)
    elif shape == 9:
        t = 6.2831853 * np.arange(48) / 48.0
        c = hx * np.stack([np.sin(t) + 2 * np.sin(2 * t), 0.35 * -np.sin(3 * t), np.cos(t) - 2 * np.cos(2 * t)], 1)
        return tube(c, None, hy)
    elif shape in (11, 14):
        turns, H = max(1.0, aux[0]), max(aux[1], hy)
        f = np.arange(64) / 63.0
        t = f * turns * 6.2831853
        c = np.stack([hx * np.cos(t), (f - 0.5) * 2 * H, hx * np.sin(t)], 1)
        if shape == 11:
            c = np.concatenate([c, np.stack([hx * np.cos(t + 3.14159265), (f - 0.5) * 2 * H, hx * np.sin(t + 3.14159265)], 1)])
        return tube(c, None, hy)
    elif shape == 12:
        t = 6.2831853 * np.arange(64) / 64.0
        S = hx * 0.0625
        c = np.stack([S * 16 * np.sin(t) ** 3, S * (13 * np.cos(t) - 5 * np.cos(2 * t) - 2 * np.cos(3 * t) - np.cos(4 * t)), 0 * t], 1)
        return tube(c, None, hy)
    elif shape == 10:
        R, wH, tH = hx, hy, max(aux[0], 0.05)
        phi = np.arctan2(z, x)
        er = np.stack([np.cos(phi), 0 * phi, np.sin(phi)], 1)
        ey = np.array([0.0, 1.0, 0.0])[None, :]
        psi = 0.5 * phi
        w = np.cos(psi)[:, None] * er + np.sin(psi)[:, None] * ey
        tt = -np.sin(psi)[:, None] * er + np.cos(psi)[:, None] * ey
        c = R * er
        o = p - c
        cu = np.clip((o * w).sum(1), -wH, wH)
        cv = np.clip((o * tt).sum(1), -tH, tH)
        q = c + cu[:, None] * w + cv[:, None] * tt
        hit = np.sqrt(((p - q) ** 2).sum(1)) > 1e-5
    elif shape == 13:
        R, sc, th = hx, max(aux[0], 0.1), np.clip(aux[1], 0.2, 2.5)
        lp = np.sqrt((p * p).sum(1))
        out = lp > R
        qq = p * sc
        sx, sy, sz, cx, cy, cz = np.sin(qq[:, 0]), np.sin(qq[:, 1]), np.sin(qq[:, 2]), np.cos(qq[:, 0]), np.cos(qq[:, 1]), np.cos(qq[:, 2])
        g = sx * cy + sy * cz + sz * cx
        grad = sc * np.stack([cx * cy - sz * sx, -sx * sy + cy * cz, -sy * sz + cz * cx], 1)
        gl = np.maximum(np.sqrt((grad * grad).sum(1)), 1e-5)
        nl = np.sign(g)[:, None] * grad / gl[:, None]
        inner = (~out) & (np.abs(g) > th)
        q = np.where(out[:, None], p / np.maximum(lp, 1e-6)[:, None] * R, np.where(inner[:, None], p - nl * ((np.abs(g) - th) / gl)[:, None], p))
        hit = out | inner
        return hit, q, np.abs(np.abs(g) - th)
    return hit, q, None


@pytest.mark.parametrize("shape", list(range(7, 15)))
def test_obb_ext_shapes_match_float64_statement(oracle, shape):
    """Shapes 7..14: the oracle's projected positions against an independent float64 numpy
    statement of the shader (identity rotation, centred container)."""
    rng = np.random.default_rng(100 + shape)
    n = 3000
    half, aux = (3.0, 1.2, 1.0), (4.0, 0.5, 3.0)
    P = np.zeros(n, oracle.PARTICLE_DTYPE)
    P["pos"][:, :3] = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    P["vel"][:, :3] = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    P["isActive"] = 1
    p = oracle.default_params(shapeType=shape, boxHalf=half, shapeAux=aux, boxCenter=(0, 0, 0), boxEulerDeg=(0, 0, 0))
    assert oracle.lib().sph_oracle_shape_supported(shape)
    out = oracle.obb(P, p)
    hit, q, aux_metric = _ext_shape_float64(shape, half, aux, P["pos"][:, :3].astype(np.float64))
    moved = (out["pos"][:, :3] != P["pos"][:, :3]).any(1)
    # decisions can only differ for points within rounding of a threshold (or nearly equidistant samples)
    if shape in (9, 11, 12, 14):
        robust = (aux_metric[:, 1] - aux_metric[:, 0]) > 1e-3          # clear nearest sample
        robust &= np.abs(np.sqrt(aux_metric[:, 0]) - half[1]) > 1e-4
    elif shape == 13:
        robust = (aux_metric > 1e-4) & (np.abs(np.sqrt((P["pos"][:, :3].astype(np.float64) ** 2).sum(1)) - half[0]) > 1e-4)
    elif shape == 8:
        robust = np.abs(((np.abs(P["pos"][:, :3].astype(np.float64)) / np.array([3.0, 1.2, 3.0])) ** 3.0).sum(1) - 1) > 1e-4
    else:
        robust = np.sqrt(((P["pos"][:, :3] - q) ** 2).sum(1)) > 1e-4
        robust |= ~hit
    assert robust.mean() > 0.95 and hit.sum() > n // 4
    assert np.array_equal(moved[robust], hit[robust])
    err = np.abs(out["pos"][:, :3].astype(np.float64) - q)[robust]
    assert err.max() < 5e-5, (shape, err.max())
    # response: the normal velocity component is reflected with restitution, tangential damped -> speed never grows
    v0 = np.sqrt((P["vel"][:, :3].astype(np.float64) ** 2).sum(1))
    v1 = np.sqrt((out["vel"][:, :3].astype(np.float64) ** 2).sum(1))
    assert np.all(v1 <= v0 * (1 + 1e-5))
    assert np.array_equal(out["vel"][~moved], P["vel"][~moved])
