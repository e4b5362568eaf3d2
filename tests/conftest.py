import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "componentframeworks-smoothed-particle-hydrodynamics_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (hyphenated directory name => importlib)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): oracle/oracle.py over oracle/liboracle.so."""
    from oracle import oracle as o
    o.lib()
    return o


def to_oracle_params(o, sp):
    """SphParams (product) -> OParams (oracle), field by field."""
    return o.default_params(
        h=sp.param_h, mass=sp.param_mass, restDensity=sp.param_restDensity, gasConstant=sp.param_gasConstant,
        viscosity=sp.param_viscosity, gravity=(sp.param_gravityX, sp.param_gravityY, sp.param_gravityZ),
        surfaceTension=sp.param_surfaceTension, timeStep=sp.param_timeStep, pause=sp.param_pause,
        useJitter=sp.param_useJitter, jitterAmp=sp.param_jitterAmp, foamGen=sp.param_foamGen,
        foamVelRef=sp.param_foamVelRef, boxCenter=list(sp.param_boxCenter), boxHalf=list(sp.param_boxHalf),
        boxEulerDeg=list(sp.param_boxEulerDeg), shapeType=sp.param_shapeType, shapeAux=list(sp.param_shapeAux),
        mixPattern=sp.param_mixPattern, dyePattern=sp.param_dyePattern, wallRestitution=sp.param_wallRestitution,
        wallFriction=sp.param_wallFriction, gridCap=sp.grid_cap,
    )


def assert_records_equal(a, b, what=""):
    """Bit-exact comparison of two 80-byte record arrays with a readable report."""
    assert a.dtype == b.dtype and a.shape == b.shape
    if a.tobytes() == b.tobytes():
        return
    for name in a.dtype.names:
        x, y = a[name], b[name]
        if x.tobytes() != y.tobytes():
            bad = np.nonzero((x.view(np.uint32) != y.view(np.uint32)).reshape(len(a), -1).any(axis=1))[0]
            i = int(bad[0])
            raise AssertionError(f"{what}: field {name} differs in {len(bad)}/{len(a)} records; first i={i}: {x[i]!r} vs {y[i]!r}")
    raise AssertionError(f"{what}: records differ")


def small_scene(pkg, n=4096, grid=16, spacing=0.85, seed=7, jitter=0.2):
    """Seeded jittered-lattice block in a grid^3 box (same recipe as synthetic.py)."""
    syn = pkg.synthetic
    cfg = syn.BenchConfig(99, f"test {n}/{grid}^3", n, (grid, grid, grid), spacing, 1)
    cfg = syn.BenchConfig(seed - 12345, cfg.name, n, cfg.grid, spacing, 1)
    rec, _ = syn.make_particles(cfg, jitter=jitter)
    sp = pkg.default_params(**syn.params_fields(cfg))
    return rec, sp
