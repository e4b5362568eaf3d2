"""MI355X-native SPH substep engine behind the reference's SPHFluidGPU surface.

Only the hot path SURVEY.md section 8 names lives here: csrc/ (HIP kernels + C-ABI,
include/sph_abi.h), engine.py (host mirror of the reference class), synthetic.py (bench
inputs), halo.py (z-slab decomposition).  Import with
importlib.import_module("componentframeworks-smoothed-particle-hydrodynamics_amd").
"""
from .engine import (  # noqa: F401
    ABI_SYMBOLS, KERNEL_CLASSES, PARTICLE_DTYPE, SPH_OPT_AOS_MODE, SPH_OPT_DEBUG, SPH_OPT_GRAPH, SPH_OPT_GRAPH_LAUNCHES, SPH_OPT_GRID_BUILD,
    SPH_ERR_TIMEOUT, SPH_OPT_NEIGHBOR_KERNEL, SPH_OPT_TIMING, SPHFluidGPU, SphError, SphFountain, SphGridInfo, SphParams, SphRiver, SphSlabIntent,
    compute_grid_extents, default_params, default_river, effective_half, generate_river_terrain, load_library, rotation_mat3, spawn_particles,
    spawn_river_particles,
)
from . import build, synthetic  # noqa: F401
