#!/usr/bin/env python3
"""Launch-bound case: BASELINE.json configs[0] (the reference's default scene: 50 000 requested
particles, 52^3 grid) stepped as Scene0p does, 16 substeps per frame (Scene0p.cpp:1482-1494),
eager launches vs the hipGraph replay of sph_dispatch_n (SPH_OPT_GRAPH).  Prints one JSON line."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")


def run(graph, frames=150, sub=16, n=50000):
    f = pkg.SPHFluidGPU(n, seed=1)
    f.set_option(pkg.SPH_OPT_GRAPH, graph)
    for _ in range(10):
        f.DispatchN(sub)
    f.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        f.DispatchN(sub)
    f.sync()
    dt = time.perf_counter() - t0
    np_ = f.GetNumFluids()
    out = {"particles": np_, "substeps_per_s": frames * sub / dt, "us_per_substep": dt / (frames * sub) * 1e6,
           "particle_substeps_per_s": np_ * frames * sub / dt, "graph_replays": f.get_option(pkg.SPH_OPT_GRAPH_LAUNCHES)}
    f.close()
    return out


if __name__ == "__main__":
    print(json.dumps({"workload": "configs[0]: SPHFluidGPU(50000) default members, 16 substeps per frame",
                      "eager": run(0), "graph": run(1)}))
