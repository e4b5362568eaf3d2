#!/usr/bin/env python3
"""Writes the regression fixtures under tests/golden/ from THIS repo's CPU oracle.

The reference ships no golden vectors for this path and cannot run here, so these
fixtures pin the oracle against regressions; they are not reference outputs ("parity unpinned").

    python tests/golden/make_golden.py
"""
import hashlib
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import PKG_NAME, small_scene, to_oracle_params  # noqa: E402
from oracle import oracle as o  # noqa: E402

pkg = importlib.import_module(PKG_NAME)


def main():
    # (1) 4096 particles, 16^3 grid: full state after 1, 10, 100 substeps + grid after build
    rec, sp = small_scene(pkg, n=4096, grid=16, seed=7)
    op = to_oracle_params(o, sp)
    out = {"initial": rec}
    cur = rec
    done = 0
    for k in (1, 10, 100):
        cur = o.substep(cur, op, steps=k - done)
        done = k
        out[f"after_{k}"] = cur
    b = o.build_grid(rec, op)
    out["cell_start"] = b["cell_start"]
    out["particle_cell"] = b["particle_cell"]
    np.savez_compressed(os.path.join(HERE, "scene4096.npz"), **out)
    # (2) BASELINE config 1 (32768 / 32^3), 100 substeps: digest + every 64th record
    syn = pkg.synthetic
    cfg = syn.CONFIGS[1]
    rec1, _ = syn.make_particles(cfg)
    sp1 = pkg.default_params(**syn.params_fields(cfg))
    end = o.substep(rec1, to_oracle_params(o, sp1), steps=100)
    np.savez_compressed(os.path.join(HERE, "config1_100.npz"), sample=end[::64], sample_initial=rec1[::64],
                        sha256=np.frombuffer(hashlib.sha256(end.tobytes()).digest(), np.uint8),
                        sha256_initial=np.frombuffer(hashlib.sha256(rec1.tobytes()).digest(), np.uint8))
    # (3) wave impulse + rotated box + cylinder container, 2000 particles, 5 substeps
    sp2 = pkg.default_params(param_shapeType=2, param_boxHalf=(2.2, 1.6, 0.9), param_boxEulerDeg=(10.0, -25.0, 40.0),
                             param_boxCenter=(0.2, -0.1, 0.3))
    rec2, mass = pkg.spawn_particles(sp2, 2000, seed=5)
    sp2.param_mass = mass
    op2 = to_oracle_params(o, sp2)
    cur = o.wave_impulse(rec2, 1.5, 3.0, 0.7, (0.3, 1.0, 0.1), -1.0, 1.0)
    cur = o.substep(cur, op2, steps=5)
    np.savez_compressed(os.path.join(HERE, "cylinder2000.npz"), initial=rec2, after=cur, mass=np.float32(mass))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
