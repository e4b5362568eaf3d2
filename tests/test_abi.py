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
    assert L.sph_abi_version() == 4


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


def _plan(pkg, **kw):
    I = pkg.SphSlabIntent()
    I.magic = 0x504c414e
    I.exchangeNo, I.stepNo, I.faceCap = 7, 7, 5000
    I.holdEvents, I.paramsHash, I.flags = 2, 0xabcdef01, 0
    for k, v in kw.items():
        cur = getattr(I, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(I, k, v)
    return I


def _agree(pkg, mine, nb, side):
    why = C.create_string_buffer(320)
    r = pkg.load_library().sph_slab_plans_agree(C.byref(mine), C.byref(nb), side, why, 320)
    return r, why.value.decode()


def test_plans_of_two_neighbours_agree_or_say_what_differs(pkg):
    """The host logic that decides whether an exchange may move a record (sph_slab_plans_agree, what sph_slab_step_finish_local and
    the RCCL handshake call): the lower rank owns layers [0, 8), the upper [8, 16); what one sends the other must expect, record for record."""
    lo = _plan(pkg, zRange=0 | (8 << 16), sendHalo=(0, 1300), sendMig=(0, 1100), recvHalo=(0, 1250), recvMig=(0, 1024))
    hi = _plan(pkg, zRange=8 | (16 << 16), sendHalo=(1250, 0), sendMig=(1024, 0), recvHalo=(1300, 0), recvMig=(1100, 0))
    assert _agree(pkg, lo, hi, 1) == (1, "") and _agree(pkg, hi, lo, 0) == (1, "")
    cases = [("exchangeNo", 8, "exchange"), ("holdEvents", 3, "one rank only"), ("paramsHash", 5, "members"), ("faceCap", 4096, "face capacity"),
             ("flags", 4, "hold"), ("magic", 0, "no plan"), ("zRange", 9 | (16 << 16), "not adjacent")]
    for field, value, word in cases:
        bad = _plan(pkg, zRange=8 | (16 << 16), sendHalo=(1250, 0), sendMig=(1024, 0), recvHalo=(1300, 0), recvMig=(1100, 0), **{field: value}) if field != "zRange" else \
            _plan(pkg, zRange=value, sendHalo=(1250, 0), sendMig=(1024, 0), recvHalo=(1300, 0), recvMig=(1100, 0))
        r, why = _agree(pkg, lo, bad, 1)
        assert r == 0 and word in why, (field, why)
        if field != "magic":                                    # (an engine's own plan always carries the magic)
            r2, why2 = _agree(pkg, bad, lo, 0)                  # ... and the other end of the link refuses too
            assert r2 == 0 and why2, (field, why2)
    # sizes: a sender that holds whole faces against a receiver that does not (the one-sided impulse of VERDICT r04)
    whole = _plan(pkg, zRange=8 | (16 << 16), sendHalo=(5000, 0), sendMig=(5000, 0), recvHalo=(5000, 0), recvMig=(5000, 0))
    r, why = _agree(pkg, lo, whole, 1)
    assert r == 0 and "will send 5000 halo copies + 5000 migrants, this rank expects 1250 + 1024" in why, why
    r, why = _agree(pkg, whole, lo, 0)
    assert r == 0 and "expects" in why
    # one direction only
    short = _plan(pkg, zRange=8 | (16 << 16), sendHalo=(1250, 0), sendMig=(1024, 0), recvHalo=(1299, 0), recvMig=(1100, 0))
    r, why = _agree(pkg, lo, short, 1)
    assert r == 0 and "this rank will send 1300 halo copies + 1100 migrants" in why and "expects 1299 + 1100" in why, why
