"""River / stream mode (last item): GenerateRiverTerrain, the river branch of the spawn
and step 5 of DispatchCompute.  CPU part: the oracle against known answers, and the host-side functions of the C-ABI
library (pure host code, no GPU) against the oracle.  The GPU part is tests/test_gpu_river.py.

The reference holds no fixture for this path (it is dead code there: Scene0p.cpp:1660 is the only writer of riverMode),
so apart from the published Microsoft rand() sequence below the pins are analytic.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import to_oracle_params

f32 = np.float32
# std::rand() of the Microsoft C runtime after srand(1): the documented sequence of `holdrand * 214013 + 2531011`
MSVC_RAND_SEED1 = [41, 18467, 6334, 26500, 19169, 15724, 11478, 29358, 26962, 24464]


def _frand(k):
    return f32(MSVC_RAND_SEED1[k]) / f32(32767.0)


def test_generate_river_terrain_uses_the_msvc_rand_sequence(oracle):
    p = oracle.default_params()
    r, heights = oracle.river_terrain(p, 1)
    assert f32(r.riverAmp) == f32(0.5) + _frand(0) * f32(1.5)              # SPHFluid3D.cpp:777-782
    assert f32(r.riverFreq) == f32(0.18) + _frand(1) * f32(0.18)
    assert f32(r.riverPhase) == _frand(2) * f32(6.2831)
    assert f32(r.riverChannelWidth) == f32(1.8) + _frand(3) * f32(1.2)
    assert f32(r.riverChannelDepth) == f32(3.5) + _frand(4) * f32(1.0)
    assert f32(r.riverSlopeDrop) == f32(0.3) + _frand(5) * f32(0.5)
    assert (p.gravity[1], p.gravity[2]) == (-120.0, 0.0)                   # :864-865
    assert (r.terrainWorldMinX, r.terrainWorldMinZ, r.terrainWorldSizeX, r.terrainWorldSizeZ) == (-7.0, -7.0, 14.0, 14.0)   # :792-795
    assert f32(r.riverEmitterRadius) == f32(r.riverChannelWidth) * f32(0.35)
    assert f32(r.riverSinkY) == f32(-7.0) + f32(0.3) and f32(r.riverSinkZMax) == f32(6.5)
    assert tuple(r.riverEmitterVel) == (0.0, -0.5, 0.5) and f32(r.riverEmitterPos[2]) == f32(-6.5)
    assert heights.shape == (64 * 64,) and np.isfinite(heights).all()


def test_terrain_shape(oracle):
    """Flat floor on the centreline sloping downstream, parabolic walls up to floor + depth, banks above the rim, nothing
    below the box floor - 0.3 (SPHFluid3D.cpp:818-847)."""
    p = oracle.default_params(boxHalf=[6.0, 5.0, 9.0], boxCenter=[1.0, 0.5, -2.0])
    r, heights = oracle.river_terrain(p, 77)
    H = heights.reshape(r.terrainH, r.terrainW)
    yBase = f32(0.5) - f32(5.0)
    assert H.min() >= yBase - f32(0.3)
    xs = f32(r.terrainWorldMinX) + np.arange(64, dtype=f32) / f32(63) * f32(r.terrainWorldSizeX)
    for iz in (0, 17, 40, 63):
        wz = f32(r.terrainWorldMinZ) + f32(iz) / f32(63) * f32(r.terrainWorldSizeZ)
        t = (wz - f32(r.terrainWorldMinZ)) / f32(r.terrainWorldSizeZ)
        floor = yBase + f32(1.0) - t * f32(r.riverSlopeDrop)
        cx = f32(1.0) + f32(r.riverAmp) * f32(np.sin(f32(f32(r.riverFreq) * wz + f32(r.riverPhase))))
        d = np.abs(xs - cx)
        inner = d < 0.45 * r.riverChannelWidth
        assert inner.any() and np.all(H[iz][inner] == floor)
        outside = d > 1.02 * r.riverChannelWidth
        assert np.all(H[iz][outside] >= floor + f32(r.riverChannelDepth) + f32(0.3) - f32(1e-5))
        wall = (d > 0.55 * r.riverChannelWidth) & (d < 0.98 * r.riverChannelWidth)
        assert np.all((H[iz][wall] > floor) & (H[iz][wall] < floor + f32(r.riverChannelDepth)))


def _flat_river(oracle, level=-3.0, **kw):
    r = oracle.default_river(riverMode=1, terrainW=8, terrainH=8, terrainWorldMinX=-7.0, terrainWorldMinZ=-7.0, terrainWorldSizeX=14.0,
                             terrainWorldSizeZ=14.0, riverAmp=0.0, riverFreq=0.25, riverPhase=0.0, riverChannelWidth=3.0,
                             riverSinkY=-6.0, riverSinkZMax=6.0, **kw)
    return r, np.full(64, level, np.float32)


def _one(oracle, pos, vel, ghost=0):
    P = np.zeros(1, oracle.PARTICLE_DTYPE)
    P["pos"][0, :3] = pos
    P["vel"][0] = (*vel, 7.0)
    P["isGhost"] = ghost
    P["density"], P["pressure"], P["acc"][0, 0] = 1234.0, 5.0, 9.0
    return P


def test_terrain_constraint_known_answer(oracle):
    """Flat heightfield: N = (0, 1, 0) exactly, so the response is -e v_y on the normal and (1 - f) on the tangent
    (TerrainConstraints.comp:63-77 with e = 0.02, f = 0.05); straight channel (amp 0): tangent (0, 1), +80 dt on v_z."""
    p = oracle.default_params(timeStep=0.002)
    r, T = _flat_river(oracle)
    out = oracle.river_step(_one(oracle, (1.0, -3.5, 2.0), (1.0, -2.0, 3.0)), p, r, T)
    assert out["pos"][0, 1] == f32(-3.0) + f32(0.001)
    e, fr = f32(0.02), f32(1.0) - f32(0.05)
    assert out["vel"][0, 0] == f32(-e * f32(-0.0) + fr * f32(1.0))
    assert out["vel"][0, 1] == f32(-e * f32(-2.0)) + fr * f32(0.0)
    assert out["vel"][0, 2] == f32(f32(fr * f32(3.0)) + f32(f32(1.0) * f32(80.0)) * f32(0.002))
    assert out["vel"][0, 3] == 7.0 and out["density"][0] == 1234.0          # not recycled: vel.w, density untouched
    # moving away from the terrain: only the push
    out = oracle.river_step(_one(oracle, (1.0, -3.5, 2.0), (1.0, 2.0, 0.0)), p, r, T)
    assert out["pos"][0, 1] == f32(-3.0) + f32(0.001) and out["vel"][0, 0] == 1.0 and out["vel"][0, 1] == 2.0
    # above the terrain / outside its footprint: untouched by the terrain pass
    out = oracle.river_step(_one(oracle, (1.0, 0.0, 2.0), (0.0, -1.0, 0.0)), p, r, T)
    assert out["pos"][0, 1] == 0.0 and out["vel"][0, 1] == -1.0
    out = oracle.river_step(_one(oracle, (8.0, -5.0, 2.0), (0.0, -1.0, 0.0)), p, r, T)
    assert out["pos"][0, 1] == -5.0


def test_channel_wall_and_ghosts(oracle):
    p = oracle.default_params()
    r, T = _flat_river(oracle)
    out = oracle.river_step(_one(oracle, (4.5, 0.0, 0.0), (2.0, 0.0, 0.0)), p, r, T)       # outside on +x, moving out
    assert out["pos"][0, 0] == 3.0 and out["vel"][0, 0] == 0.0
    out = oracle.river_step(_one(oracle, (-4.5, 0.0, 0.0), (2.0, 0.0, 0.0)), p, r, T)      # outside on -x, moving back in
    assert out["pos"][0, 0] == -3.0 and out["vel"][0, 0] == 2.0
    g = _one(oracle, (4.5, -9.0, 9.0), (2.0, 0.0, 0.0), ghost=1)                           # isGhost == 1: all three passes skip
    assert oracle.river_step(g, p, r, T).tobytes() == g.tobytes()
    g2 = _one(oracle, (4.5, 0.0, 0.0), (2.0, 0.0, 0.0), ghost=2)                           # flags.x == 2 is not skipped (== 1 test)
    assert oracle.river_step(g2, p, r, T)["pos"][0, 0] == 3.0


def test_stream_emit_known_answer(oracle):
    """StreamEmit.comp:31-60 for particle index 0: seed = 1013904223, the LCG gives r1, (r2), r3, r4."""
    p = oracle.default_params(restDensity=998.0)
    r, T = _flat_river(oracle, riverEmitterRadius=1.25)
    r.riverEmitterPos[:] = (0.5, 2.0, -6.0)
    r.riverEmitterVel[:] = (0.25, -0.5, 0.75)
    s, draws = 1013904223, []
    for _ in range(4):
        draws.append(f32(s & 0xFFFF) / f32(65535.0))
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
    r1, _, r3, r4 = draws
    # below the sink (outside the terrain's footprint: inside it the terrain pass lifts the particle first) / past the end
    for pos in ((8.0, -6.5, 0.0), (0.0, 0.0, 6.5)):
        out = oracle.river_step(_one(oracle, pos, (1.0, 1.0, 1.0)), p, r, T)
        z = f32(-6.0) + r1 * (f32(6.0) - f32(-6.0))
        assert out["pos"][0, 2] == z
        assert out["pos"][0, 0] == f32(0.0) + f32(f32(f32(r4 - f32(0.5)) * f32(2.0)) * f32(1.25))   # amp 0: centreline = boxCenterX
        assert out["pos"][0, 1] == f32(2.0) + r3 * f32(0.6)
        assert tuple(out["vel"][0]) == (0.25, -0.5, 0.75, 0.0)
        assert tuple(out["acc"][0]) == (0.0, 0.0, 0.0, 0.0) and out["density"][0] == 998.0 and out["pressure"][0] == 0.0


def test_river_spawn(oracle):
    p = oracle.default_params(boxHalf=[5.0, 5.0, 5.0])
    r, T = oracle.river_terrain(p, 3)
    P, mass = oracle.river_spawn(p, r, T, 4000, 9)
    spacing = f32(0.28) * f32(0.85)
    assert len(P) == 4000 and f32(mass) == f32(1000.0) * spacing * spacing * spacing
    assert np.all(P["isGhost"] == 0) and np.all(P["isActive"] == 0) and np.all(P["padC"] == np.arange(4000) % 2)
    chan = P["vel"][:, 2] == 0.5
    assert chan.any() and np.all(P["vel"][~chan, 2] == 2.0)                  # channel fill first, emitter fill after
    assert np.all(np.diff(chan.astype(int)) <= 0)
    cx = r.riverAmp * np.sin(r.riverFreq * P["pos"][chan, 2] + r.riverPhase)
    assert np.all(np.abs(P["pos"][chan, 0] - cx) <= r.riverChannelWidth + 3 * spacing)
    few, _ = oracle.river_spawn(p, r, T, 100000, 9)                          # more than the channel holds: the rest at the emitter
    assert len(few) == 100000 and (few["vel"][:, 2] == 2.0).sum() > 0
    d = few[few["vel"][:, 2] == 2.0]["pos"]
    assert np.all(np.abs(d[:, 0] - r.riverEmitterPos[0]) <= 0.5 * r.riverChannelWidth + 1e-5)


@pytest.mark.parametrize("seed,half,center", [(1, (7.0, 7.0, 7.0), (0.0, 0.0, 0.0)), (12345, (6.0, 5.0, 9.0), (1.0, 0.5, -2.0)), (-7, (4.0, 8.0, 4.5), (-3.0, 2.0, 0.25))])
def test_host_river_functions_equal_the_oracle(pkg, oracle, seed, half, center):
    """sph_generate_river_terrain / sph_spawn_river_particles (host code of the C-ABI library) == oracle, bit for bit."""
    sp = pkg.default_params(param_boxHalf=half, param_boxCenter=center)
    op = to_oracle_params(oracle, sp)
    r, heights = pkg.generate_river_terrain(sp, seed)
    orv, oh = oracle.river_terrain(op, seed)
    assert bytes(r) == bytes(orv) and heights.tobytes() == oh.tobytes()
    assert (sp.param_gravityY, sp.param_gravityZ) == (-120.0, 0.0) == (op.gravity[1], op.gravity[2])
    for n in (500, 30000):
        P, m = pkg.spawn_river_particles(sp, r, heights, n, 42)
        Q, mo = oracle.river_spawn(op, orv, oh, n, 42)
        assert m == mo and P.tobytes() == Q.tobytes()


def test_river_struct_layout(pkg, oracle):
    L = oracle.lib()
    assert C.sizeof(pkg.SphRiver) == C.sizeof(oracle.ORiver) == L.sph_oracle_sizeof_river() == 88
    assert bytes(pkg.default_river()) == bytes(oracle.default_river())
    d = pkg.default_river()                                                   # SPHFluid3D.h:172-196 initialisers
    assert (d.riverMode, d.terrainW, d.terrainH, d.riverSinkY, d.riverSinkZMax, d.riverAmp, d.riverFreq) == (0, 64, 64, -8.5, 9.0, 2.0, 0.25)
    assert tuple(d.riverEmitterPos) == (0.0, 3.0, -9.0) and tuple(d.riverEmitterVel) == (0.0, -0.5, 4.0)
