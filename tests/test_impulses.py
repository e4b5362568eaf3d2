"""Per-frame impulse kernels: VortexImpulse, AttractorImpulse,
StencilAttract, CurlFlow.  CPU: known-answer tests of the oracle from the shader formulas.
GPU: the HIP kernels reproduce the oracle bit for bit, also between substeps."""
import math

import numpy as np
import pytest

from conftest import assert_records_equal, small_scene, to_oracle_params


def _cloud(o, n=600, seed=0, span=5.0):
    rng = np.random.default_rng(seed)
    P = np.zeros(n, o.PARTICLE_DTYPE)
    P["pos"][:, :3] = rng.uniform(-span, span, (n, 3)).astype(np.float32)
    P["vel"][:, :3] = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    P["isGhost"][::37] = 1
    return P


def _smooth(e0, e1, x):
    t = np.clip((x - e0) / (e1 - e0), 0, 1)
    return t * t * (3 - 2 * t)


def test_vortex_kat(oracle):
    p = oracle.default_params(boxCenter=(0.5, -1.0, 0.25), boxEulerDeg=(0, 0, 0))
    P = _cloud(oracle)
    out = oracle.vortex_impulse(P, p, 0.8, 0.3)
    rel = P["pos"][:, :3].astype(np.float64) - np.array([0.5, -1.0, 0.25])
    axis = np.array([0.0, 1.0, 0.0])
    radial = rel - np.outer(rel @ axis, axis)
    r = np.linalg.norm(radial, axis=1)
    rhat = radial / r[:, None]
    that = np.cross(axis, rhat)
    fall = _smooth(0.0, 0.35 * 7.0, r)
    dv = that * (0.8 * fall)[:, None] - rhat * (0.3 * fall)[:, None]
    dv[P["isGhost"] != 0] = 0
    np.testing.assert_allclose(out["vel"][:, :3] - P["vel"][:, :3], dv, atol=3e-6)
    assert oracle.vortex_impulse(P, p, 0.0, 0.0).tobytes() == P.tobytes()
    # a rotated container swirls around ITS local Y axis: a pure tangential kick has no component along it
    p = oracle.default_params(boxEulerDeg=(30, 0, 40))
    R = oracle.rotation(p.boxEulerDeg[:]).astype(np.float64)
    out = oracle.vortex_impulse(P, p, 1.0, 0.0)
    dv = (out["vel"][:, :3] - P["vel"][:, :3]).astype(np.float64)
    assert np.abs(dv @ R[3:6]).max() < 1e-5 and np.abs(dv).max() > 0.1


def test_attractor_kat(oracle):
    P = _cloud(oracle, seed=1)
    pt, pull, radius = (1.0, 0.5, -2.0), 0.6, 4.0
    out = oracle.attractor_impulse(P, pt, pull, radius)
    rel = np.array(pt) - P["pos"][:, :3].astype(np.float64)
    d = np.linalg.norm(rel, axis=1)
    soften = max(0.15 * radius, 0.2)
    k = pull * soften / (d + soften) * (1 - _smooth(0.6 * radius, radius, d))
    dv = rel / d[:, None] * k[:, None]
    dv[P["isGhost"] != 0] = 0
    np.testing.assert_allclose(out["vel"][:, :3] - P["vel"][:, :3], dv, atol=3e-6)
    assert np.all((out["vel"][:, :3] == P["vel"][:, :3])[d > radius])          # nothing beyond the radius
    assert oracle.attractor_impulse(P, pt, 0.0, radius).tobytes() == P.tobytes()


def test_stencil_kat(oracle):
    P = _cloud(oracle, seed=2)
    rng = np.random.default_rng(3)
    targets = rng.uniform(-3, 3, (7, 4)).astype(np.float32)
    out = oracle.stencil_attract(P, targets, 0.25, 0.9)                          # damp is capped at 0.5
    t = targets[np.arange(len(P)) % 7, :3].astype(np.float64)
    v = (P["vel"][:, :3] + (t - P["pos"][:, :3]) * 0.25) * 0.5
    fluid = P["isGhost"] == 0
    np.testing.assert_allclose(out["vel"][fluid, :3], v[fluid], atol=3e-6)
    assert np.array_equal(out["vel"][~fluid], P["vel"][~fluid])
    assert oracle.stencil_attract(P, np.zeros((0, 4), np.float32), 0.25, 0.1).tobytes() == P.tobytes()
    assert oracle.stencil_attract(P, targets, 0.0, 0.0).tobytes() == P.tobytes()


def test_curl_flow_kat(oracle):
    """Independent numpy port of CurlFlow.comp in fp32 (same GLSL definitions of fract / mix /
    smooth fade; the hash is chaotic, so an fp64 port would not be comparable): |dv| <= kick,
    time dependence, ghosts untouched, agreement to a few ulp."""
    F = np.float32
    P = _cloud(oracle, n=300, seed=4)

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)

    def fract(x):
        return (x - np.floor(x)).astype(F)

    def hash13(x, y, z):
        x, y, z = fract((x * F(0.1031)).astype(F)), fract((y * F(0.1031)).astype(F)), fract((z * F(0.1031)).astype(F))
        dd = fma(z, (x + F(31.32)).astype(F), fma(y, (y + F(31.32)).astype(F), (x * (z + F(31.32)).astype(F)).astype(F)))
        x, y, z = (x + dd).astype(F), (y + dd).astype(F), (z + dd).astype(F)
        return fract(((x + y).astype(F) * z).astype(F))

    def mix(a, b, t):
        return ((a * (F(1.0) - t).astype(F)).astype(F) + (b * t).astype(F)).astype(F)

    def vnoise(x, y, z):
        ix, iy, iz = np.floor(x).astype(F), np.floor(y).astype(F), np.floor(z).astype(F)
        fade = lambda f: ((f * f).astype(F) * (F(3.0) - (F(2.0) * f).astype(F)).astype(F)).astype(F)
        fx, fy, fz = fade((x - ix).astype(F)), fade((y - iy).astype(F)), fade((z - iz).astype(F))
        o = F(1.0)
        c = lambda dx, dy, dz: hash13((ix + dx).astype(F) if dx else ix, (iy + dy).astype(F) if dy else iy, (iz + dz).astype(F) if dz else iz)
        return mix(mix(mix(c(0, 0, 0), c(o, 0, 0), fx), mix(c(0, o, 0), c(o, o, 0), fx), fy),
                   mix(mix(c(0, 0, o), c(o, 0, o), fx), mix(c(0, o, o), c(o, o, o), fx), fy), fz)

    p1 = vnoise
    p2 = lambda x, y, z: vnoise((x + F(31.416)).astype(F), (y + F(47.853)).astype(F), (z + F(12.793)).astype(F))
    p3 = lambda x, y, z: vnoise((x + F(-233.145)).astype(F), (y + F(93.912)).astype(F), (z + F(55.121)).astype(F))
    kick, scale, time = F(0.4), F(0.7), F(1.3)
    pos = P["pos"][:, :3]
    qx, qy, qz = (pos[:, 0] * scale).astype(F), (pos[:, 1] * scale).astype(F), ((pos[:, 2] * scale).astype(F) + time).astype(F)
    h = F(0.35)
    pl, mi = lambda v: (v + h).astype(F), lambda v: (v - h).astype(F)
    d = lambda a, b: (a - b).astype(F)
    dP3dy = d(p3(qx, pl(qy), qz), p3(qx, mi(qy), qz)); dP2dz = d(p2(qx, qy, pl(qz)), p2(qx, qy, mi(qz)))
    dP1dz = d(p1(qx, qy, pl(qz)), p1(qx, qy, mi(qz))); dP3dx = d(p3(pl(qx), qy, qz), p3(mi(qx), qy, qz))
    dP2dx = d(p2(pl(qx), qy, qz), p2(mi(qx), qy, qz)); dP1dy = d(p1(qx, pl(qy), qz), p1(qx, mi(qy), qz))
    inv = F(2.0) * h
    cx, cy, cz = (d(dP3dy, dP2dz) / inv).astype(F), (d(dP1dz, dP3dx) / inv).astype(F), (d(dP2dx, dP1dy) / inv).astype(F)
    m = np.sqrt(fma(cz, cz, fma(cy, cy, (cx * cx).astype(F)))).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        dirs = [np.where(m > F(1e-5), (c / m).astype(F), F(0)).astype(F) for c in (cx, cy, cz)]
    mm = np.minimum(m, F(1.0))
    want = P["vel"][:, :3].copy()
    fluid = P["isGhost"] == 0
    for a in range(3):
        want[fluid, a] = (P["vel"][:, a] + ((dirs[a] * mm).astype(F) * kick).astype(F)).astype(F)[fluid]
    out = oracle.curl_flow(P, float(kick), float(scale), float(time))
    np.testing.assert_allclose(out["vel"][:, :3], want, rtol=0, atol=2e-6)
    got = (out["vel"][:, :3] - P["vel"][:, :3]).astype(np.float64)
    assert np.linalg.norm(got, axis=1).max() <= float(kick) * (1 + 1e-5) and np.linalg.norm(got, axis=1).max() > 0.05
    assert np.all(got[~fluid] == 0)
    assert oracle.curl_flow(P, 0.0, float(scale), float(time)).tobytes() == P.tobytes()
    assert oracle.curl_flow(P, float(kick), float(scale), float(time) + 0.5).tobytes() != out.tobytes()


@pytest.mark.gpu
def test_hip_impulses_bit_exact(pkg, oracle):
    rec, sp = small_scene(pkg, n=4096, grid=16, seed=61)
    sp.param_boxEulerDeg[0], sp.param_boxEulerDeg[2] = 20.0, -35.0
    op = to_oracle_params(oracle, sp)
    rng = np.random.default_rng(6)
    rec["vel"][:, :3] = rng.normal(0, 3, (len(rec), 3)).astype(np.float32)
    rec["isGhost"][::41] = 2
    targets = rng.uniform(-2, 2, (333, 4)).astype(np.float32)
    for lazy in (False, True):
        f = pkg.SPHFluidGPU.from_particles(rec, sp)
        if lazy:
            f.set_option(pkg.SPH_OPT_AOS_MODE, 1)
        want = rec
        f.SetStencilTargets(targets)
        for step in range(6):
            f.ApplyVortexImpulse(0.5, 0.1)
            want = oracle.vortex_impulse(want, op, 0.5, 0.1)
            f.ApplyAttractorImpulse((0.3, -0.2, 0.5), 0.4, 2.5)
            want = oracle.attractor_impulse(want, (0.3, -0.2, 0.5), 0.4, 2.5)
            f.ApplyStencilAttract(0.05, 0.02)
            want = oracle.stencil_attract(want, targets, 0.05, 0.02)
            f.ApplyCurlFlow(0.3, 0.8, 0.1 * step)
            want = oracle.curl_flow(want, 0.3, 0.8, 0.1 * step)
            if step == 0:
                assert_records_equal(f.download(), want, f"impulses before any substep (lazy={lazy})")
            f.DispatchCompute()
            want = oracle.substep(want, op)
        assert_records_equal(f.download(), want, f"impulses interleaved with substeps (lazy={lazy})")
        f.close()


# ---- fountain recycle (DispatchCompute step 6) -------------------------------------------

def _fountain_scene(pkg, n=6000):
    sp = pkg.default_params(param_boxHalf=(3.0, 3.0, 3.0))
    rec, mass = pkg.spawn_particles(sp, n, seed=21)
    sp.param_mass = mass
    rec["isGhost"][::37] = 1          # ghosts stay put (FountainRecycle.comp:34)
    rec["isGhost"][5::41] = 2         # isGhost == 2 is NOT skipped by the fountain (== 1 test), but by OBB
    rec["isActive"][3::29] = 0
    return rec, sp


def test_oracle_fountain_kat(oracle):
    """FountainRecycle.comp:24-54 against a direct numpy statement (LCG stream, nozzle disc, jet cone)."""
    n = 5000
    rng = np.random.default_rng(8)
    P = np.zeros(n, oracle.PARTICLE_DTYPE)
    P["pos"][:, :3] = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    P["vel"][:, :3] = rng.normal(size=(n, 3)).astype(np.float32)
    P["acc"][:, :3] = 1.0
    P["density"], P["pressure"], P["isActive"] = 900.0, 5.0, 1
    P["isGhost"][::10] = 1
    P["padC"] = np.arange(n) % 3
    p = oracle.default_params(boxHalf=(3, 3, 3), boxCenter=(0.5, 0.25, -0.5))
    f = oracle.default_fountain(mode=1, offset=(0.0, -2.0, 0.0), drainPerSec=300.0, drainLevel=2.0)
    dt, seed = 1e-3, 12345
    out = oracle.fountain_recycle(P, p, f, dt, seed)
    drain_y = np.float32(np.float32(0.25 - 3.0) + np.float32(2.0))
    chance = np.float32(min(1.0, 300.0 * dt))
    u32 = lambda v: np.uint32(v & 0xFFFFFFFF)
    moved = 0
    for i in range(n):
        s = (int(i) ^ ((seed * 747796405) & 0xFFFFFFFF)) + 2891336453 & 0xFFFFFFFF
        def lcg():
            nonlocal s
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            return np.float32(np.float32(s & 0xFFFFFF) / np.float32(16777215.0))
        rec = P[i]
        expect_move = rec["isGhost"] != 1 and rec["pos"][1] < drain_y and lcg() <= chance
        if not expect_move:
            assert out[i].tobytes() == rec.tobytes(), i
            continue
        moved += 1
        r1, r2, r3, r4 = lcg(), lcg(), lcg(), lcg()
        ang = np.float64(np.float32(6.2831853) * r1)
        rad = np.float64(np.sqrt(r2))
        want = np.array([0.5 + np.cos(ang) * rad, 0.25 - 2.0 + 0.2 * r3, -0.5 + np.sin(ang) * rad])
        np.testing.assert_allclose(out[i]["pos"][:3], want, atol=2e-6)
        side = np.array([np.cos(ang), 1.0 / max(0.25 * r4, 1e-30), np.sin(ang)]) * (0.25 * r4)
        side[1] = 1.0
        v = 25.0 * side / np.linalg.norm(side)
        np.testing.assert_allclose(out[i]["vel"][:3], v, rtol=2e-6, atol=2e-6)
        assert np.all(out[i]["acc"] == 0) and out[i]["density"] == np.float32(1000.0) and out[i]["pressure"] == 0
        assert out[i]["padC"] == rec["padC"] and out[i]["isActive"] == rec["isActive"]
    assert 100 < moved < n


@pytest.mark.gpu
@pytest.mark.parametrize("neighbor,aos", [(3, 0), (2, 0), (1, 0), (3, 1)])
def test_fountain_matches_oracle(pkg, oracle, neighbor, aos):
    rec, sp = _fountain_scene(pkg)
    f = pkg.SPHFluidGPU.from_particles(rec, sp)
    f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, neighbor)
    f.set_option(pkg.SPH_OPT_AOS_MODE, aos)
    f.fountainMode = 1
    f.fountainOffset = (0.2, -2.5, -0.1)
    f.fountainDrainPerSec = 150.0
    f.fountainDrainLevel = 1.5
    f.fountainSeed = 7
    of = oracle.default_fountain(mode=1, offset=(0.2, -2.5, -0.1), drainPerSec=150.0, drainLevel=1.5, seed=7)
    op = to_oracle_params(oracle, sp)
    f.DispatchN(3)
    want = oracle.substep(rec, op, steps=3, fountain=of)
    assert f.fountainSeed == 10 == of.seed
    f.fountainJetSpeedLive = 40.0                 # audio-kicked per frame in the reference (Scene0p.cpp:3560-3580)
    of.jetSpeedLive = 40.0
    for _ in range(3):
        f.DispatchCompute(5e-4)
    want = oracle.substep(want, op, dt=5e-4, steps=3, fountain=of)
    got = f.download()
    assert_records_equal(got, want, "fountain")
    assert (got["density"] == np.float32(1000.0)).sum() > 20       # some particles really were recycled
    f.fountainMode = 0
    f.DispatchCompute()
    assert f.fountainSeed == 13                   # no advance while the mode is off
    assert_records_equal(f.download(), oracle.substep(want, op, steps=1), "fountain off")
    f.close()
