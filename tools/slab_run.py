#!/usr/bin/env python3
"""One rank of the z-slab driver without neighbours on config 3 (the C-ABI exchange on a one-rank RCCL communicator), for
rocprofv3 --kernel-trace --stats: which kernels the fixed per-substep cost of the slab path consists of.
usage: slab_run.py [substeps=45]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd")
halo = importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd.halo")
syn = pkg.synthetic
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 45
cfg = syn.CONFIGS[3]
sp = pkg.default_params(**syn.params_fields(cfg))
torch.cuda.set_device(0)
sim = halo.SlabSimulation.from_config(cfg, sp, 0, 1, stream=torch.cuda.current_stream().cuda_stream, transport="rccl")
for _ in range(steps):
    sim.DispatchCompute(-1.0)
torch.cuda.synchronize()
print("done")
