"""Regression fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from this
repo's oracle; the reference has none for this path).  CPU: the oracle still reproduces them.
GPU: the HIP path reproduces them through the C-ABI."""
import hashlib
import os

import numpy as np
import pytest

from conftest import ROOT, assert_records_equal, small_scene, to_oracle_params

G = os.path.join(ROOT, "tests", "golden")


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_oracle_reproduces_scene4096(pkg, oracle):
    z = _load("scene4096.npz")
    rec, sp = small_scene(pkg, n=4096, grid=16, seed=7)
    assert rec.tobytes() == z["initial"].tobytes()
    op = to_oracle_params(oracle, sp)
    assert_records_equal(oracle.substep(rec, op, steps=1), z["after_1"], "after 1")
    assert_records_equal(oracle.substep(z["after_1"], op, steps=9), z["after_10"], "after 10")
    b = oracle.build_grid(rec, op)
    assert np.array_equal(b["cell_start"], z["cell_start"]) and np.array_equal(b["particle_cell"], z["particle_cell"])


def test_oracle_reproduces_cylinder(pkg, oracle):
    z = _load("cylinder2000.npz")
    sp = pkg.default_params(param_shapeType=2, param_boxHalf=(2.2, 1.6, 0.9), param_boxEulerDeg=(10.0, -25.0, 40.0),
                            param_boxCenter=(0.2, -0.1, 0.3))
    rec, mass = pkg.spawn_particles(sp, 2000, seed=5)
    assert rec.tobytes() == z["initial"].tobytes() and np.float32(mass) == z["mass"]
    sp.param_mass = mass
    op = to_oracle_params(oracle, sp)
    cur = oracle.wave_impulse(rec, 1.5, 3.0, 0.7, (0.3, 1.0, 0.1), -1.0, 1.0)
    assert_records_equal(oracle.substep(cur, op, steps=5), z["after"], "cylinder")


@pytest.mark.gpu
def test_hip_reproduces_golden(pkg):
    z = _load("scene4096.npz")
    rec, sp = small_scene(pkg, n=4096, grid=16, seed=7)
    f = pkg.SPHFluidGPU.from_particles(rec, sp)
    cnt, pcell = f.download_grid()
    assert np.array_equal(cnt, np.diff(z["cell_start"])) and np.array_equal(pcell, z["particle_cell"])
    f.DispatchCompute()
    assert_records_equal(f.download(), z["after_1"], "HIP after 1")
    f.DispatchN(9)
    assert_records_equal(f.download(), z["after_10"], "HIP after 10")
    f.DispatchN(90)
    assert_records_equal(f.download(), z["after_100"], "HIP after 100")
    f.close()
    # config 1, 100 substeps: digest of the whole state + sampled records
    z = _load("config1_100.npz")
    syn = pkg.synthetic
    cfg = syn.CONFIGS[1]
    rec1, _ = syn.make_particles(cfg)
    assert hashlib.sha256(rec1.tobytes()).digest() == z["sha256_initial"].tobytes()
    f = pkg.SPHFluidGPU.from_particles(rec1, pkg.default_params(**syn.params_fields(cfg)))
    f.DispatchN(100)
    got = f.download()
    assert_records_equal(got[::64], z["sample"], "config 1 sample")
    assert hashlib.sha256(got.tobytes()).digest() == z["sha256"].tobytes()
    f.close()
    # cylinder + rotation + wave impulse
    z = _load("cylinder2000.npz")
    sp = pkg.default_params(param_shapeType=2, param_boxHalf=(2.2, 1.6, 0.9), param_boxEulerDeg=(10.0, -25.0, 40.0),
                            param_boxCenter=(0.2, -0.1, 0.3), param_mass=float(z["mass"]))
    f = pkg.SPHFluidGPU.from_particles(z["initial"], sp)
    f.ApplyWaveImpulse(1.5, 3.0, 0.7, (0.3, 1.0, 0.1), -1.0, 1.0)
    f.DispatchN(5)
    assert_records_equal(f.download(), z["after"], "HIP cylinder")
    f.close()
