#!/usr/bin/env python3
"""Fixed per-substep cost of the z-slab driver without any neighbour, against the plain engine on the same particles:
SlabSimulation with world = 1 through (a) the host-count path (pack kernel, count read-back, Python) and (b) the C-ABI
exchange without host round trips (sph_slab_exchange on a one-rank RCCL communicator: pack, header, unpack, commit).
usage: slab_overhead.py [config index=3 | weak5] [substeps=40]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
halo = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd.halo")
syn = pkg.synthetic
ci = sys.argv[1] if len(sys.argv) > 1 else "3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = syn.weak_config(1) if ci == "weak5" else syn.CONFIGS[int(ci)]   # weak5 = one rank's share of BASELINE configs[4]
sp = pkg.default_params(**syn.params_fields(cfg))
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream


def run(sim):
    for _ in range(5):
        sim.DispatchCompute(-1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.DispatchCompute(-1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


rec, _ = syn.make_particles(cfg)
plain = run(pkg.SPHFluidGPU.from_particles(rec, sp, stream=stream))
slab = run(halo.SlabSimulation.from_config(cfg, sp, 0, 1, stream=stream))
slab_async = run(halo.SlabSimulation.from_config(cfg, sp, 0, 1, stream=stream, transport="rccl"))
print(json.dumps({"config": cfg.name, "substeps": steps, "plain_engine_ms": round(plain, 4), "slab_world1_hostcounts_ms": round(slab, 4),
                  "slab_world1_c_abi_exchange_ms": round(slab_async, 4), "fixed_overhead_hostcounts_us": round((slab - plain) * 1e3, 1),
                  "fixed_overhead_c_abi_exchange_us": round((slab_async - plain) * 1e3, 1)}))
