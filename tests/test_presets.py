"""Preset loader: PresetIO parsing semantics and the sim / box / fountain
subset of Scene0p::ApplyPresetKV; the reference's 13 shipped presets (tests/golden/presets.json,
extracted by tests/golden/make_presets.py) as regression scenes on the GPU."""
import importlib
import json
import os
import types

import numpy as np
import pytest

from conftest import assert_records_equal, to_oracle_params

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "presets.json")
PRESETS = json.load(open(GOLDEN))


@pytest.fixture(scope="module")
def presets(pkg):
    return importlib.import_module(pkg.__name__ + ".presets")


def _members(pkg):
    p = pkg.default_params()
    ns = types.SimpleNamespace(**{name: (list(getattr(p, name)) if hasattr(getattr(p, name), "__len__") else getattr(p, name))
                                  for name, _ in p._fields_})
    ns.numParticles = 50000
    ns.fountainMode, ns.fountainOffset, ns.fountainRadius, ns.fountainSpread = 0, [0.0, -5.0, 0.0], 1.0, 0.25
    ns.fountainJetSpeedLive, ns.fountainDrainLevel, ns.fountainDrainPerSec = 25.0, 1.0, 2.0
    return ns


def test_parse_semantics(presets):
    """PresetIO.cpp:26-40."""
    text = "# SPH Fluid Preset v1\r\nversion=1\r\n\r\nsim.h=0.25\nsim.h=0.5\n=novalue\ngarbage line\nbox.half=1,2,3\nlook.name=a=b\n#x=1\n"
    kv = presets.parse(text)
    assert kv == {"version": "1", "sim.h": "0.25", "box.half": "1,2,3", "look.name": "a=b"}


def test_typed_accessors(presets):
    """PresetIO.cpp:137-164: strtof / strtol prefix parsing, defaults on failure."""
    kv = {"a": "1.5abc", "b": "abc", "c": " -7", "d": "3.9", "e": "1,2", "f": "1, 2 ,3.5", "g": "0.100000001", "h": "1e2", "i": "x,1,2"}
    assert presets.get_f(kv, "a", 9.0) == 1.5 and presets.get_f(kv, "b", 9.0) == 9.0 and presets.get_f(kv, "zz", 2.0) == 2.0
    assert presets.get_f(kv, "g", 0.0) == float(np.float32(0.1)) and presets.get_f(kv, "h", 0.0) == 100.0
    assert presets.get_i(kv, "c", 1) == -7 and presets.get_i(kv, "d", 1) == 3 and presets.get_i(kv, "b", 4) == 4
    assert presets.get_b(kv, "c", False) is True and presets.get_b(kv, "zz", True) is True
    assert presets.get_f3(kv, "e", [7, 8, 9]) == [7, 8, 9] and presets.get_f3(kv, "f", [7, 8, 9]) == [1.0, 2.0, 3.5]
    assert presets.get_f3(kv, "i", [7, 8, 9]) == [7, 8, 9] and presets.get_f3(kv, "zz", [7, 8, 9]) == [7, 8, 9]


def test_apply_maps_reference_keys(pkg, presets):
    """Scene0p.cpp:2341-2377, :2389-2392, :2482-2490."""
    kv = PRESETS["THE CUBE"]
    m = presets.apply(_members(pkg), kv)
    assert m.param_gasConstant == float(np.float32(4387.23047)) and m.param_gravityY == float(np.float32(-784.989563))
    assert m.param_boxHalf == [float(np.float32(7.14582825))] * 3 and m.param_shapeType == 0 and m.numParticles == 50000
    assert m.fountainMode == 0 and m.fountainRadius == float(np.float32(1.56728125)) and m.fountainJetSpeedLive == float(np.float32(33.8744278))
    # non-structural application (drop sequencer, Scene0p.cpp:2541,2560) leaves spawn-time members alone
    m2 = _members(pkg)
    m2.param_mixPattern, m2.numParticles = 2, 1234
    presets.apply(m2, {"look.mixPattern": "1", "sim.particleCount": "99999", "sim.viscosity": "7"}, structural=False)
    assert m2.param_mixPattern == 2 and m2.numParticles == 1234 and m2.param_viscosity == 7.0
    presets.apply(m2, {"sim.particleCount": "10"})
    assert m2.numParticles == 1000                      # std::max(1000, pc), :2362


@pytest.mark.parametrize("name", sorted(PRESETS))
def test_shipped_presets_are_valid_scenes(pkg, oracle, presets, name):
    """Every shipped preset yields members the engine's host helpers and the oracle accept."""
    m = presets.apply(_members(pkg), PRESETS[name])
    sp = pkg.default_params(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in vars(m).items() if k.startswith("param_")})
    g = pkg.compute_grid_extents(sp)
    assert 1 <= g.numCells <= 160 ** 3 and oracle.lib().sph_oracle_shape_supported(sp.param_shapeType)
    rec, mass = pkg.spawn_particles(sp, 3000, seed=2)
    ref, mass2 = oracle.spawn(to_oracle_params(oracle, sp), 3000, seed=2)
    assert len(rec) > 0 and rec.tobytes() == ref.tobytes() and mass == mass2


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PRESETS))
def test_shipped_presets_match_oracle(pkg, oracle, presets, name):
    """Load preset -> ResetSimulation (as Scene0p does on a structural load, :2162 + :1456-1462) ->
    12 substeps, against the oracle on the same initial records and members."""
    f = pkg.SPHFluidGPU(2000, seed=4)
    presets.apply(f, PRESETS[name])
    f.numParticles = min(f.numParticles, 30000)        # keeps the CPU oracle quick; the scene is otherwise the preset's
    f.ResetSimulation(seed=4)
    rec = f.download()
    assert len(rec) == f.GetNumFluids() > 1000
    f.fountainMode = 1 if name in ("ASID", "TRIP2") else f.fountainMode    # exercise step 6 on two of them as well
    of = None
    if f.fountainMode:
        of = oracle.default_fountain(mode=1, offset=tuple(f.fountainOffset), radius=f.fountainRadius, spread=f.fountainSpread,
                                     jetSpeedLive=f.fountainJetSpeedLive, drainLevel=f.fountainDrainLevel,
                                     drainPerSec=f.fountainDrainPerSec, seed=f.fountainSeed)
    op = to_oracle_params(oracle, f.params)
    f.DispatchN(12)
    assert_records_equal(f.download(), oracle.substep(rec, op, steps=12, fountain=of), name)
    f.close()


# ---- pinned by the reference itself: oracle/_ref = the reference's PresetIO.cpp (tests/golden/make_presets_parsed.py) ----
PARSED = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "presets_parsed.json")))
_F_SENT, _I_SENT, _V_SENT = -12345.5, -777, [-1.25, -2.5, -3.75]


def _fbits(x):
    return format(int(np.float32(x).view(np.uint32)), "08x")


def _same_float(hexbits, got):
    want = np.uint32(int(hexbits, 16)).view(np.float32)
    return (np.isnan(want) and np.isnan(np.float32(got))) or _fbits(got) == hexbits


@pytest.mark.parametrize("name", sorted(PARSED["presets"]) + ["__" + k for k in sorted(PARSED["edge"])])
def test_loader_equals_the_references_presetio(presets, name):
    """presets.parse / get_f / get_i / get_b / get_f3 against what PresetIO::Parse / GetF / GetI / GetB / GetF3 of the
    reference (compiled as oracle/_ref) returned for the same text: the 13 shipped presets and the edge-case files."""
    if name.startswith("__"):
        ref = PARSED["edge"][name[2:]]
        kv = presets.parse(PARSED["edge_text"][name[2:]])
    else:
        ref = PARSED["presets"][name]
        kv_all = presets.parse("".join(f"{k}={v}\n" for k, v in ref["kv"].items()))     # the file's own pairs, re-parsed
        assert kv_all == ref["kv"]
        kv = kv_all
        # the committed extraction of the shipped preset (presets.json) is a subset of what the reference parsed
        assert all(ref["kv"].get(k) == v for k, v in PRESETS[name].items())
    assert ref["loaded"] and kv == ref["kv"]
    for key, hexbits in ref["f"].items():
        assert _same_float(hexbits, presets.get_f(kv, key, _F_SENT)), (key, kv.get(key))
    for key, want in ref["i"].items():
        assert presets.get_i(kv, key, _I_SENT) == want, (key, kv.get(key))
    for key, (w0, w1) in ref["b"].items():
        assert presets.get_b(kv, key, False) == bool(w0) and presets.get_b(kv, key, True) == bool(w1), key
    for key, want in ref["v"].items():
        got = presets.get_f3(kv, key, _V_SENT)
        assert [_fbits(x) for x in got] == want, (key, kv.get(key), got)
