"""The C-ABI library loads, exports every symbol include/sph_abi.h declares, fails loudly
without a device, and its host-only entry points agree with the oracle (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, to_oracle_params


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "sph_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_[a-z0-9_]+)\s*\(", src)))


def test_exports_match_header(pkg):
    L = pkg.load_library()
    declared = _declared_functions()
    assert sorted(pkg.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name
    assert L.sph_abi_version() == 3


def test_struct_layouts(pkg):
    assert pkg.PARTICLE_DTYPE.itemsize == 80
    assert [pkg.PARTICLE_DTYPE.fields[f][1] for f in ("pos", "vel", "acc", "density", "pressure", "padA", "padB", "isGhost", "isActive", "padC", "pad0")] == \
        [0, 16, 32, 48, 52, 56, 60, 64, 68, 72, 76]
    p = pkg.default_params()
    assert (p.param_h, p.param_restDensity, p.param_gasConstant, p.param_viscosity) == pytest.approx((0.28, 1000, 2000, 3.5))
    assert (p.param_gravityX, p.param_gravityY, p.param_gravityZ) == (0, -980, 0)
    assert list(p.param_boxHalf) == [7, 7, 7] and p.param_shapeType == 0 and p.grid_cap == 160
    assert p.param_wallRestitution == pytest.approx(0.15) and p.param_wallFriction == pytest.approx(0.02)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present: the failure path is not reachable")
def test_no_device_fails_loudly(pkg):
    with pytest.raises(pkg.SphError, match="HIP|device"):
        pkg.SPHFluidGPU(1000)


def test_host_helpers_match_oracle(pkg, oracle):
    rng = np.random.default_rng(0)
    for shape in range(15):
        for _ in range(4):
            sp = pkg.default_params(
                param_boxHalf=tuple(rng.uniform(0.5, 9, 3)), param_boxCenter=tuple(rng.uniform(-2, 2, 3)),
                param_boxEulerDeg=tuple(rng.uniform(-180, 180, 3)), param_shapeAux=tuple(rng.uniform(0.2, 6, 3)),
                param_h=float(rng.uniform(0.1, 0.5)), grid_cap=int(rng.choice([160, 64, 400])))
            sp.param_shapeType = shape      # validate_params only gates engine creation, not host helpers
            op = to_oracle_params(oracle, sp)
            assert np.array_equal(pkg.rotation_mat3(sp.param_boxEulerDeg), oracle.rotation(op.boxEulerDeg[:]))
            assert np.array_equal(pkg.effective_half(sp), oracle.effective_half(op))
            a, b = pkg.compute_grid_extents(sp), oracle.grid_extents(op)
            assert list(a.dims) == list(b.dims) and a.numCells == b.numCells
            assert list(a.gridMin) == list(b.gridMin) and a.cellSize == b.cellSize


@pytest.mark.parametrize("shape", list(range(15)))
@pytest.mark.parametrize("mix,dye,jit", [(0, 0, 1), (1, 1, 1), (2, 2, 1), (0, 2, 0)])
def test_spawn_matches_oracle(pkg, oracle, shape, mix, dye, jit):
    sp = pkg.default_params(param_shapeType=shape, param_boxHalf=(4.0, 3.0, 2.5), param_boxCenter=(0.5, 1.0, -1.0),
                            param_shapeAux=(3.0, 0.6, 3.0),
                            param_mixPattern=mix, param_dyePattern=dye, param_useJitter=jit)
    a, ma = pkg.spawn_particles(sp, 20000, seed=77)
    b, mb = oracle.spawn(to_oracle_params(oracle, sp), 20000, seed=77)
    assert ma == mb and len(a) == len(b) and len(a) > 0
    assert a.tobytes() == b.tobytes()


def test_python_mirror_surface(pkg):
    """The host mirror keeps the reference's member names (SPHFluid3D.h:26-124)."""
    cls = pkg.SPHFluidGPU
    for name in ("DispatchCompute", "ResetSimulation", "ApplyWaveImpulse", "EffectiveHalf", "GetNumFluids",
                 "ComputeGridExtents", "SimulateSubstep", "particles", "gridSizeX", "gridSizeY", "gridSizeZ",
                 "numCells", "gridMinV", "cellSize"):
        assert hasattr(cls, name), name
    ref_members = ["param_h", "param_mass", "param_restDensity", "param_gasConstant", "param_viscosity", "param_gravityY",
                   "param_gravityX", "param_gravityZ", "param_surfaceTension", "param_timeStep", "param_pause",
                   "param_useJitter", "param_jitterAmp", "param_foamGen", "param_foamVelRef", "param_boxCenter",
                   "param_boxHalf", "param_boxEulerDeg", "param_shapeType", "param_shapeAux", "param_mixPattern",
                   "param_dyePattern", "param_wallRestitution", "param_wallFriction"]
    assert [f[0] for f in pkg.SphParams._fields_][:-1] == ref_members


def test_slab_message_sizing_rule(pkg):
    """The z-slab exchange sizes a message from counts that both ends of a link know (host-only rule, no device): whole faces
    unless the face's record count was calm over the last two known exchanges, then count + 25 % + 1024, never above the capacity."""
    L = pkg.load_library()
    f = L.sph_slab_message_records
    cap = 200000
    assert f(100000, 100500, cap) == 100000 + 25000 + 1024            # calm (0.5 %): the records in use + a quarter + 1024
    assert f(100000, 104000, cap) == cap and f(104000, 100000, cap) == cap   # 4 % apart: not calm, the whole face
    assert f(190000, 190000, cap) == cap                               # the margin never exceeds the capacity
    assert f(0, 0, cap) == 1024 and f(10, 70, cap) == 10 + 2 + 1024    # small faces: the absolute slack of 64 records counts as calm
    assert f(10, 200, cap) == cap
    for seen, before in ((5000, 5100), (5100, 5000), (123456, 120000)):
        assert 0 < f(seen, before, cap) <= cap
