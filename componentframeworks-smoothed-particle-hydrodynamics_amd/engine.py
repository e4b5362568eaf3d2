"""Host-side mirror of the reference's `SPHFluidGPU` class over the C-ABI (include/sph_abi.h).

The reference boundary is the C++ class SPHFluidGPU
(/root/reference/ComponentFramework/SPHFluid3D.h:26-210): public methods plus public
`param_*` data members that the caller pokes directly (Scene0p.cpp:936-1056) and that are
re-read at every DispatchCompute (SPHFluid3D.cpp:458-506).  This module keeps those names:
`DispatchCompute`, `ResetSimulation`, `ApplyWaveImpulse`, `EffectiveHalf`, `GetNumFluids`,
`ComputeGridExtents`, `param_h` ... `param_wallFriction`, `numParticles`, `particles`,
`gridSizeX/Y/Z`, `numCells`, `gridMinV`, `cellSize`.  The C++ twin of this file is
include/SPHFluidGPU_hip.hpp.

There is no CPU fallback: if libsph_hip.so is missing or HIP has no device, construction
raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

FLT_MAX = 3.4028234663852886e38

# 80-byte record, SPHFluid3D.h:12-24
PARTICLE_DTYPE = np.dtype(
    [
        ("pos", "<f4", (4,)), ("vel", "<f4", (4,)), ("acc", "<f4", (4,)),
        ("density", "<f4"), ("pressure", "<f4"), ("padA", "<f4"), ("padB", "<f4"),
        ("isGhost", "<i4"), ("isActive", "<i4"), ("padC", "<i4"), ("pad0", "<i4"),
    ]
)
assert PARTICLE_DTYPE.itemsize == 80


class SphParams(C.Structure):
    """struct SphParams of include/sph_abi.h (param_* members, SPHFluid3D.h:94-124)."""

    _fields_ = [
        ("param_h", C.c_float), ("param_mass", C.c_float), ("param_restDensity", C.c_float),
        ("param_gasConstant", C.c_float), ("param_viscosity", C.c_float),
        ("param_gravityY", C.c_float), ("param_gravityX", C.c_float), ("param_gravityZ", C.c_float),
        ("param_surfaceTension", C.c_float), ("param_timeStep", C.c_float),
        ("param_pause", C.c_int32),
        ("param_useJitter", C.c_int32), ("param_jitterAmp", C.c_float),
        ("param_foamGen", C.c_float), ("param_foamVelRef", C.c_float),
        ("param_boxCenter", C.c_float * 3), ("param_boxHalf", C.c_float * 3), ("param_boxEulerDeg", C.c_float * 3),
        ("param_shapeType", C.c_int32), ("param_shapeAux", C.c_float * 3),
        ("param_mixPattern", C.c_int32), ("param_dyePattern", C.c_int32),
        ("param_wallRestitution", C.c_float), ("param_wallFriction", C.c_float),
        ("grid_cap", C.c_int32),
    ]


class SphFountain(C.Structure):
    """fountain* members of the reference class (SPHFluid3D.h:161-168), same names."""
    _fields_ = [("fountainMode", C.c_int32), ("fountainOffset", C.c_float * 3), ("fountainRadius", C.c_float),
                ("fountainSpread", C.c_float), ("fountainJetSpeedLive", C.c_float), ("fountainDrainLevel", C.c_float),
                ("fountainDrainPerSec", C.c_float), ("fountainSeed", C.c_uint32)]


class SphRiver(C.Structure):
    """river / terrain members of the reference class (SPHFluid3D.h:171-196), same names."""
    _fields_ = [("riverMode", C.c_int32), ("terrainW", C.c_int32), ("terrainH", C.c_int32),
                ("terrainWorldMinX", C.c_float), ("terrainWorldMinZ", C.c_float), ("terrainWorldSizeX", C.c_float), ("terrainWorldSizeZ", C.c_float),
                ("riverEmitterPos", C.c_float * 3), ("riverEmitterVel", C.c_float * 3), ("riverEmitterRadius", C.c_float),
                ("riverSinkY", C.c_float), ("riverSinkZMax", C.c_float), ("riverAmp", C.c_float), ("riverFreq", C.c_float),
                ("riverPhase", C.c_float), ("riverChannelWidth", C.c_float), ("riverChannelDepth", C.c_float), ("riverSlopeDrop", C.c_float)]


class SphGridInfo(C.Structure):
    _fields_ = [("dims", C.c_int32 * 3), ("numCells", C.c_int32), ("gridMin", C.c_float * 3), ("cellSize", C.c_float)]


# option / kernel-class constants of sph_abi.h
SPH_OPT_NEIGHBOR_KERNEL, SPH_OPT_GRID_BUILD, SPH_OPT_AOS_MODE, SPH_OPT_TIMING, SPH_OPT_DEBUG = 1, 2, 3, 4, 100
SPH_OPT_GRAPH, SPH_OPT_GRAPH_LAUNCHES = 5, 6
KERNEL_CLASSES = ("bin", "scan", "scatter", "sph", "writeback", "impulse", "other")

# every symbol include/sph_abi.h declares (checked by tests/test_abi.py)
class SphSlabIntent(C.Structure):
    """The plan of one sized halo exchange (include/sph_abi.h SphSlabIntent): what both ends of a link must agree on before a record moves."""
    _fields_ = [("magic", C.c_uint32), ("exchangeNo", C.c_uint32), ("stepNo", C.c_uint32), ("faceCap", C.c_uint32),
                ("sendHalo", C.c_uint32 * 2), ("sendMig", C.c_uint32 * 2), ("recvHalo", C.c_uint32 * 2), ("recvMig", C.c_uint32 * 2),
                ("holdEvents", C.c_uint32), ("paramsHash", C.c_uint32), ("flags", C.c_uint32), ("zRange", C.c_uint32)]


assert C.sizeof(SphSlabIntent) == 64
SPH_ERR_TIMEOUT = -5


ABI_SYMBOLS = (
    "sph_abi_version", "sph_params_default", "sph_rotation_mat3", "sph_effective_half",
    "sph_compute_grid_extents", "sph_spawn_particles", "sph_last_error", "sph_create",
    "sph_create_from_particles", "sph_destroy", "sph_reset", "sph_set_params", "sph_get_params",
    "sph_set_option", "sph_get_option", "sph_dispatch", "sph_dispatch_n", "sph_apply_wave_impulse",
    "sph_num_particles", "sph_grid_info", "sph_upload_particles", "sph_download_particles",
    "sph_device_particles", "sph_pack_render_buffer", "sph_initial_particles", "sph_download_grid", "sph_sync", "sph_kernel_times",
    "sph_debug_counters", "sph_apply_vortex_impulse", "sph_apply_attractor_impulse", "sph_set_stencil_targets",
    "sph_apply_stencil_attract", "sph_apply_curl_flow", "sph_fountain_default", "sph_set_fountain", "sph_get_fountain", "sph_create_slab", "sph_slab_pack", "sph_slab_unpack", "sph_slab_download",
    "sph_slab_alloc_faces", "sph_slab_face_buffer", "sph_slab_pack_async", "sph_slab_unpack_async", "sph_slab_status",
    "sph_comm_unique_id", "sph_comm_create", "sph_comm_destroy", "sph_comm_selftest", "sph_comm_selftest_timed", "sph_slab_exchange",
    "sph_slab_step_begin", "sph_slab_step_finish", "sph_slab_step_finish_local",
    "sph_slab_face_bytes", "sph_slab_clear_flags", "sph_slab_message_bytes", "sph_slab_message_records", "sph_slab_step_times",
    "sph_river_default", "sph_generate_river_terrain", "sph_spawn_river_particles", "sph_set_river", "sph_get_river",
    "sph_slab_set_verify", "sph_slab_set_deadline", "sph_slab_plan", "sph_slab_plans_agree", "sph_sync_deadline",
    "sph_slab_debug_tight_messages", "sph_comm_selftest_faces",
)
# sph_debug_counters (SPH_OPT_DEBUG bit 3): diagnostics of k_sph_walk / k_sph_list, summed over launches:
# [0] candidate rows walked from global memory (window too large; k_sph_walk), [1] targets on an exact fallback sweep,
# [2] neighbour-list entries, [3] candidate rows (k_sph_walk), [4] lanes, [5] targets whose list overflowed, [6] targets that
# left the list's slack, [7] waves with at least one fallback target
STAMP_NAMES = ("rows_unstaged", "slow_targets", "list_entries", "rows", "lanes", "overflow_targets", "far_targets", "waves_with_fallback")


class SphError(RuntimeError):
    pass


_lib = None


def lib_path() -> str:
    return _build.LIB_PATH


def load_library(build_if_missing: bool = True) -> C.CDLL:
    """Load libsph_hip.so (building it in-tree first if asked and needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing:
        _build.build()
    if not os.path.exists(_build.LIB_PATH):
        raise SphError(f"{_build.LIB_PATH} is missing: run __graft_entry__.build(); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same soname as the
    # system one).  Whichever loads first wins, and torch cannot see the GPU behind the system
    # runtime, so let torch load first when it is installed (it is only plumbing here: halo
    # buffers, streams, torch.distributed).
    if os.environ.get("SPH_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    # developer A/B of kernel variants: SPH_HIP_LIB names another build of the same sources
    L = C.CDLL(os.environ.get("SPH_HIP_LIB") or _build.LIB_PATH)
    vp, pp, gp = C.c_void_p, C.POINTER(SphParams), C.POINTER(SphGridInfo)
    f3 = C.POINTER(C.c_float)
    L.sph_abi_version.restype = C.c_int
    L.sph_last_error.restype = C.c_char_p
    L.sph_params_default.argtypes = [pp]
    L.sph_rotation_mat3.argtypes = [f3, f3]
    L.sph_effective_half.argtypes = [pp, f3]
    L.sph_compute_grid_extents.argtypes = [pp, gp]
    L.sph_spawn_particles.argtypes = [pp, C.c_size_t, C.c_uint32, vp, C.POINTER(C.c_size_t), f3]
    L.sph_create.argtypes = [C.POINTER(vp), C.c_size_t, pp, C.c_uint32, vp]
    L.sph_create_from_particles.argtypes = [C.POINTER(vp), vp, C.c_size_t, pp, vp]
    L.sph_destroy.argtypes = [vp]
    L.sph_reset.argtypes = [vp, C.c_size_t, C.c_uint32]
    L.sph_set_params.argtypes = [vp, pp]
    L.sph_get_params.argtypes = [vp, pp]
    L.sph_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.sph_get_option.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.sph_dispatch.argtypes = [vp, C.c_float]
    L.sph_dispatch_n.argtypes = [vp, C.c_float, C.c_int]
    L.sph_apply_wave_impulse.argtypes = [vp, C.c_float, C.c_float, C.c_float, f3, C.c_float, C.c_float]
    L.sph_num_particles.argtypes = [vp]
    L.sph_num_particles.restype = C.c_size_t
    L.sph_grid_info.argtypes = [vp, gp]
    L.sph_upload_particles.argtypes = [vp, vp, C.c_size_t]
    L.sph_download_particles.argtypes = [vp, vp, C.c_size_t]
    L.sph_device_particles.argtypes = [vp, C.POINTER(vp)]
    L.sph_pack_render_buffer.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.sph_initial_particles.argtypes = [vp, vp, C.c_size_t]
    L.sph_download_grid.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
    L.sph_sync.argtypes = [vp]
    L.sph_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    L.sph_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int, C.c_int]
    L.sph_apply_vortex_impulse.argtypes = [vp, C.c_float, C.c_float]
    L.sph_apply_attractor_impulse.argtypes = [vp, f3, C.c_float, C.c_float]
    L.sph_set_stencil_targets.argtypes = [vp, vp, C.c_size_t]
    L.sph_apply_stencil_attract.argtypes = [vp, C.c_float, C.c_float]
    L.sph_apply_curl_flow.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    L.sph_fountain_default.argtypes = [C.POINTER(SphFountain)]
    L.sph_set_fountain.argtypes = [vp, C.POINTER(SphFountain)]
    L.sph_get_fountain.argtypes = [vp, C.POINTER(SphFountain)]
    L.sph_create_slab.argtypes = [C.POINTER(vp), vp, vp, C.c_size_t, pp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp]
    L.sph_slab_pack.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.sph_slab_unpack.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
    L.sph_slab_download.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sph_slab_alloc_faces.argtypes = [vp, C.c_uint32]
    L.sph_slab_face_buffer.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.sph_slab_pack_async.argtypes = [vp]
    L.sph_slab_unpack_async.argtypes = [vp, vp, vp, C.c_uint32]
    L.sph_slab_status.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.sph_comm_unique_id.argtypes = [vp]
    L.sph_comm_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int]
    L.sph_comm_destroy.argtypes = [vp]
    L.sph_comm_selftest.argtypes = [vp, C.c_uint64]
    L.sph_comm_selftest_timed.argtypes = [vp, C.c_uint64, C.POINTER(C.c_float)]
    L.sph_slab_exchange.argtypes = [vp, vp]
    L.sph_slab_step_begin.argtypes = [vp, C.c_float]
    L.sph_slab_step_finish.argtypes = [vp, vp]
    L.sph_slab_step_finish_local.argtypes = [vp, vp, vp]
    L.sph_slab_face_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.sph_slab_clear_flags.argtypes = [vp, C.c_uint32]
    L.sph_slab_message_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.sph_slab_step_times.argtypes = [vp, C.POINTER(C.c_float)]
    L.sph_slab_message_records.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.sph_river_default.argtypes = [C.POINTER(SphRiver)]
    L.sph_generate_river_terrain.argtypes = [pp, C.c_int, C.POINTER(SphRiver), vp]
    L.sph_spawn_river_particles.argtypes = [pp, C.POINTER(SphRiver), vp, C.c_size_t, C.c_uint32, vp, C.POINTER(C.c_size_t), C.POINTER(C.c_float)]
    L.sph_set_river.argtypes = [vp, C.POINTER(SphRiver), vp]
    L.sph_get_river.argtypes = [vp, C.POINTER(SphRiver)]
    L.sph_slab_set_verify.argtypes = [vp, C.c_int]
    L.sph_slab_set_deadline.argtypes = [vp, C.c_double]
    L.sph_slab_plan.argtypes = [vp, C.POINTER(SphSlabIntent), C.POINTER(C.c_float)]
    L.sph_slab_plans_agree.argtypes = [C.POINTER(SphSlabIntent), C.POINTER(SphSlabIntent), C.c_int, C.c_char_p, C.c_size_t]
    L.sph_sync_deadline.argtypes = [vp, C.c_double]
    L.sph_slab_debug_tight_messages.argtypes = [vp, C.c_int]
    L.sph_comm_selftest_faces.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if name not in ("sph_last_error", "sph_num_particles", "sph_abi_version", "sph_fountain_default", "sph_river_default"):
            fn.restype = C.c_int
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        msg = load_library().sph_last_error()
        raise SphError(f"sph C-ABI error {rc}: {msg.decode() if msg else '?'}")


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def default_params(**overrides) -> SphParams:
    p = SphParams()
    _check(load_library().sph_params_default(C.byref(p)))
    for k, v in overrides.items():
        cur = getattr(p, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(p, k, v)
    return p


def compute_grid_extents(p: SphParams) -> SphGridInfo:
    g = SphGridInfo()
    _check(load_library().sph_compute_grid_extents(C.byref(p), C.byref(g)))
    return g


def rotation_mat3(euler_deg) -> np.ndarray:
    out = (C.c_float * 9)()
    _check(load_library().sph_rotation_mat3(_f3(euler_deg), out))
    return np.array(out, np.float32)


def effective_half(p: SphParams) -> np.ndarray:
    out = (C.c_float * 3)()
    _check(load_library().sph_effective_half(C.byref(p), out))
    return np.array(out, np.float32)


def spawn_particles(p: SphParams, n_requested: int, seed: int):
    buf = np.zeros(max(n_requested, 1), PARTICLE_DTYPE)
    n = C.c_size_t()
    mass = C.c_float()
    _check(load_library().sph_spawn_particles(C.byref(p), n_requested, seed, buf.ctypes.data_as(C.c_void_p), C.byref(n), C.byref(mass)))
    return buf[: n.value].copy(), float(mass.value)


def default_river(**kw) -> SphRiver:
    r = SphRiver()
    load_library().sph_river_default(C.byref(r))
    for k, v in kw.items():
        setattr(r, k, v)
    return r


def generate_river_terrain(p: SphParams, seed: int, river: SphRiver | None = None):
    """GenerateRiverTerrain (SPHFluid3D.cpp:772-878) as a pure host function: fills `river`, returns (river, heights);
    writes param_gravityY / Z of `p` like the reference."""
    r = river if river is not None else default_river()
    heights = np.zeros(r.terrainW * r.terrainH, np.float32)
    _check(load_library().sph_generate_river_terrain(C.byref(p), int(seed), C.byref(r), heights.ctypes.data_as(C.c_void_p)))
    return r, heights


def spawn_river_particles(p: SphParams, river: SphRiver, heights: np.ndarray, n_requested: int, seed: int):
    """The river branch of InitializeParticles (SPHFluid3D.cpp:104-160)."""
    h = np.ascontiguousarray(heights, np.float32)
    buf = np.zeros(max(n_requested, 1), PARTICLE_DTYPE)
    n = C.c_size_t()
    mass = C.c_float()
    _check(load_library().sph_spawn_river_particles(C.byref(p), C.byref(river), h.ctypes.data_as(C.c_void_p), n_requested, seed,
                                                    buf.ctypes.data_as(C.c_void_p), C.byref(n), C.byref(mass)))
    return buf[: n.value].copy(), float(mass.value)


_PARAM_NAMES = {f[0] for f in SphParams._fields_}
_RIVER_NAMES = {f[0] for f in SphRiver._fields_}
_FOUNTAIN_NAMES = {f[0] for f in SphFountain._fields_}


class SPHFluidGPU:
    """Drop-in for the reference class of the same name (SPHFluid3D.h:26).

    SPHFluidGPU(numParticles)                    -> spawn as InitializeParticles does (seeded)
    SPHFluidGPU.from_particles(records, params)  -> caller-provided 80-byte records
    """

    def __init__(self, numParticles_: int = 50000, params: SphParams | None = None, seed: int = 1, stream: int | None = None,
                 _particles: np.ndarray | None = None):
        L = load_library()
        object.__setattr__(self, "_L", L)
        object.__setattr__(self, "_p", params if params is not None else default_params())
        object.__setattr__(self, "_h", C.c_void_p())
        object.__setattr__(self, "_f", SphFountain())
        L.sph_fountain_default(C.byref(self._f))
        object.__setattr__(self, "_r", SphRiver())
        L.sph_river_default(C.byref(self._r))
        object.__setattr__(self, "terrainHeights", np.zeros(0, np.float32))   # SPHFluid3D.h:175
        object.__setattr__(self, "_terrain_sent", None)
        object.__setattr__(self, "numParticles", int(numParticles_))
        object.__setattr__(self, "seed", int(seed))
        if _particles is not None:
            arr = np.ascontiguousarray(_particles, dtype=PARTICLE_DTYPE)
            _check(L.sph_create_from_particles(C.byref(self._h), arr.ctypes.data_as(C.c_void_p), len(arr), C.byref(self._p), stream))
        else:
            _check(L.sph_create(C.byref(self._h), self.numParticles, C.byref(self._p), self.seed, stream))
        _check(L.sph_get_params(self._h, C.byref(self._p)))   # spawn overwrote param_mass (SPHFluid3D.cpp:92)

    @classmethod
    def from_particles(cls, particles: np.ndarray, params: SphParams, stream: int | None = None) -> "SPHFluidGPU":
        return cls(len(particles), params=params, stream=stream, _particles=particles)

    # -- public param_* members ----------------------------------------------------------
    def __getattr__(self, name):
        if name in _PARAM_NAMES or name in _FOUNTAIN_NAMES or name in _RIVER_NAMES:
            v = getattr(object.__getattribute__(self, "_p" if name in _PARAM_NAMES else ("_f" if name in _FOUNTAIN_NAMES else "_r")), name)
            return list(v) if hasattr(v, "__len__") else v
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _PARAM_NAMES or name in _FOUNTAIN_NAMES or name in _RIVER_NAMES:
            st = self._p if name in _PARAM_NAMES else (self._f if name in _FOUNTAIN_NAMES else self._r)
            cur = getattr(st, name)
            if hasattr(cur, "__len__"):
                for i, x in enumerate(value):
                    cur[i] = x
            else:
                setattr(st, name, value)
        else:
            object.__setattr__(self, name, value)

    @property
    def params(self) -> SphParams:
        return self._p

    # -- reference methods ---------------------------------------------------------------
    def _push_members(self):
        _check(self._L.sph_set_params(self._h, C.byref(self._p)))  # members are re-read every dispatch (:458-506)
        _check(self._L.sph_set_fountain(self._h, C.byref(self._f)))  # fountain* members (:519-541)
        self._push_river()

    def _push_river(self):                                           # river members (:511-516); the heightfield only when it changed
        th = self.terrainHeights
        fresh = th is not self._terrain_sent and len(th) > 0
        ptr = None
        if fresh:
            th = np.ascontiguousarray(th, np.float32)
            if len(th) != self._r.terrainW * self._r.terrainH:
                raise SphError(f"terrainHeights has {len(th)} samples, terrainW x terrainH = {self._r.terrainW * self._r.terrainH}")
            ptr = th.ctypes.data_as(C.c_void_p)
        _check(self._L.sph_set_river(self._h, C.byref(self._r), ptr))
        if fresh:
            object.__setattr__(self, "_terrain_sent", self.terrainHeights)

    def set_river(self, river: SphRiver, heights):                   # all river members + terrainHeights in one go
        C.memmove(C.byref(self._r), C.byref(river), C.sizeof(SphRiver))
        object.__setattr__(self, "terrainHeights", np.ascontiguousarray(heights, np.float32).copy())
        self._push_river()

    def GenerateRiverTerrain(self, seed: int):                       # SPHFluid3D.cpp:772-878
        heights = np.zeros(self._r.terrainW * self._r.terrainH, np.float32)
        _check(self._L.sph_generate_river_terrain(C.byref(self._p), int(seed), C.byref(self._r), heights.ctypes.data_as(C.c_void_p)))
        object.__setattr__(self, "terrainHeights", heights)
        self._push_river()

    def DispatchCompute(self, overrideDt: float = -1.0):            # SPHFluid3D.cpp:431
        self._push_members()
        _check(self._L.sph_dispatch(self._h, overrideDt))
        _check(self._L.sph_get_fountain(self._h, C.byref(self._f)))  # fountainSeed++ (:541)

    SimulateSubstep = DispatchCompute   # BASELINE.json's name for the same entry point

    def DispatchN(self, n: int, overrideDt: float = -1.0):           # Scene0p.cpp:3720-3739 loop
        self._push_members()
        _check(self._L.sph_dispatch_n(self._h, overrideDt, int(n)))
        _check(self._L.sph_get_fountain(self._h, C.byref(self._f)))

    def ResetSimulation(self, seed: int | None = None):             # SPHFluid3D.cpp:713
        if seed is not None:
            self.seed = int(seed)
        _check(self._L.sph_set_params(self._h, C.byref(self._p)))
        self._push_river()                                           # riverMode && !terrainHeights.empty() picks the spawn branch (:104)
        _check(self._L.sph_reset(self._h, self.numParticles, self.seed))
        _check(self._L.sph_get_params(self._h, C.byref(self._p)))

    def ApplyWaveImpulse(self, amplitude, wavelength, phase, dir, yMin=-FLT_MAX, yMax=FLT_MAX):   # SPHFluid3D.cpp:604
        _check(self._L.sph_apply_wave_impulse(self._h, amplitude, wavelength, phase, _f3(dir), yMin, yMax))

    def ApplyVortexImpulse(self, tangentKick, inwardKick):          # SPHFluid3D.cpp:627
        _check(self._L.sph_set_params(self._h, C.byref(self._p)))   # reads param_boxCenter / EulerDeg / half
        _check(self._L.sph_apply_vortex_impulse(self._h, tangentKick, inwardKick))

    def ApplyAttractorImpulse(self, point, pullKick, radius):        # SPHFluid3D.cpp:650
        _check(self._L.sph_apply_attractor_impulse(self._h, _f3(point), pullKick, radius))

    def SetStencilTargets(self, points):                             # SPHFluid3D.cpp:684
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
        _check(self._L.sph_set_stencil_targets(self._h, pts.ctypes.data_as(C.c_void_p), len(pts)))
        object.__setattr__(self, "stencilCount", len(pts))

    def ApplyStencilAttract(self, pullKick, dampKick):               # SPHFluid3D.cpp:695
        _check(self._L.sph_apply_stencil_attract(self._h, pullKick, dampKick))

    def ApplyCurlFlow(self, kick, scale, time):                      # SPHFluid3D.cpp:668
        _check(self._L.sph_apply_curl_flow(self._h, kick, scale, time))

    def EffectiveHalf(self):                                         # SPHFluid3D.h:127
        return effective_half(self._p)

    def ComputeGridExtents(self):                                    # SPHFluid3D.cpp:354
        return compute_grid_extents(self._p)

    def GetNumFluids(self) -> int:                                   # SPHFluid3D.cpp:601
        return int(self._L.sph_num_particles(self._h))

    # -- reference data members ----------------------------------------------------------
    @property
    def particles(self) -> np.ndarray:
        """Host copy of the INITIAL records (never refreshed, as in the reference)."""
        n = self.GetNumFluids()
        out = np.zeros(n, PARTICLE_DTYPE)
        _check(self._L.sph_initial_particles(self._h, out.ctypes.data_as(C.c_void_p), n))
        return out

    def _grid(self) -> SphGridInfo:
        g = SphGridInfo()
        _check(self._L.sph_grid_info(self._h, C.byref(g)))
        return g

    gridSizeX = property(lambda self: self._grid().dims[0])
    gridSizeY = property(lambda self: self._grid().dims[1])
    gridSizeZ = property(lambda self: self._grid().dims[2])
    numCells = property(lambda self: self._grid().numCells)
    gridMinV = property(lambda self: list(self._grid().gridMin))
    cellSize = property(lambda self: self._grid().cellSize)

    # -- engine extras -------------------------------------------------------------------
    def set_option(self, option: int, value: int):
        _check(self._L.sph_set_option(self._h, option, value))

    def get_option(self, option: int) -> int:
        v = C.c_int(0)
        _check(self._L.sph_get_option(self._h, option, C.byref(v)))
        return v.value

    def upload(self, particles: np.ndarray):
        arr = np.ascontiguousarray(particles, dtype=PARTICLE_DTYPE)
        _check(self._L.sph_upload_particles(self._h, arr.ctypes.data_as(C.c_void_p), len(arr)))

    def download(self) -> np.ndarray:
        n = self.GetNumFluids()
        out = np.zeros(n, PARTICLE_DTYPE)
        _check(self._L.sph_download_particles(self._h, out.ctypes.data_as(C.c_void_p), n))
        return out

    def device_particles(self) -> int:
        """Device address of the 80-byte AoS (the `ssbo` renderers bind, Scene0p.cpp:1625)."""
        p = C.c_void_p()
        _check(self._L.sph_device_particles(self._h, C.byref(p)))
        return int(p.value)

    def pack_render_buffer(self, dev_ptr: int, w_mode: int = 0):
        """(x, y, z, w) per particle in original order into a caller-owned device buffer (a mapped vertex
        buffer in a renderer; a torch tensor's data_ptr() in the tests)."""
        _check(self._L.sph_pack_render_buffer(self._h, C.c_void_p(dev_ptr), self.GetNumFluids(), int(w_mode)))

    def download_grid(self):
        g = compute_grid_extents(self._p)
        n = self.GetNumFluids()
        cnt = np.zeros(g.numCells, np.int32)
        pc = np.zeros(max(n, 1), np.int32)
        _check(self._L.sph_set_params(self._h, C.byref(self._p)))
        _check(self._L.sph_download_grid(self._h, cnt.ctypes.data_as(C.c_void_p), g.numCells, pc.ctypes.data_as(C.c_void_p), n))
        return cnt, pc[:n]

    def sync(self):
        _check(self._L.sph_sync(self._h))

    def kernel_times(self, reset: bool = False):
        ms = (C.c_double * len(KERNEL_CLASSES))()
        cnt = (C.c_int64 * len(KERNEL_CLASSES))()
        _check(self._L.sph_kernel_times(self._h, ms, cnt, 1 if reset else 0))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(KERNEL_CLASSES)}

    def debug_counters(self, reset: bool = False) -> dict:
        buf = (C.c_uint64 * len(STAMP_NAMES))()
        _check(self._L.sph_debug_counters(self._h, buf, len(STAMP_NAMES), 1 if reset else 0))
        return {k: int(buf[i]) for i, k in enumerate(STAMP_NAMES)}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.sph_destroy(self._h)
            object.__setattr__(self, "_h", C.c_void_p())

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
