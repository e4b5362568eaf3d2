"""The header-only C++ twin of the reference class (include/SPHFluidGPU_hip.hpp)."""
import os
import shutil
import subprocess

import pytest

from conftest import PKG_NAME, ROOT

PKG_DIR = os.path.join(ROOT, PKG_NAME)
EXE = os.path.join(ROOT, "examples", "headless_scene")


def _build_example():
    cmd = ["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "headless_scene.cpp"),
           "-L", PKG_DIR, "-lsph_hip", "-Wl,-rpath," + PKG_DIR, "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", EXE]
    subprocess.run(cmd, check=True, capture_output=True)


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_shim_compiles_and_links_against_the_c_abi(pkg):
    pkg.load_library()
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"), "-x", "c++",
                    os.path.join(ROOT, "include", "SPHFluidGPU_hip.hpp")], check=True, capture_output=True)
    _build_example()
    assert os.path.exists(EXE)
    src = open(os.path.join(ROOT, "include", "SPHFluidGPU_hip.hpp")).read()
    for name in ("DispatchCompute", "ResetSimulation", "ApplyWaveImpulse", "EffectiveHalf", "ComputeGridExtents", "GetNumFluids",
                 "param_h", "param_mass", "param_restDensity", "param_gasConstant", "param_viscosity", "param_gravityY",
                 "param_surfaceTension", "param_timeStep", "param_pause", "param_boxCenter", "param_boxHalf", "param_boxEulerDeg",
                 "param_shapeType", "param_shapeAux", "param_wallRestitution", "param_wallFriction", "numParticles", "particles",
                 "gridSizeX", "numCells", "gridMinV", "cellSize", "GetFluidVBO", "ssbo", "riverMode"):
        assert name in src, name


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_headless_scene_runs_on_the_gpu(pkg):
    """Scene0p's call pattern (ctor, per-frame impulse, 16-substep frames, param edits, reset)
    through the C++ shim."""
    pkg.load_library()
    _build_example()
    env = dict(os.environ, LD_LIBRARY_PATH=PKG_DIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    res = subprocess.run([EXE, "50000"], capture_output=True, text=True, env=env, timeout=300)
    print(res.stdout, res.stderr)
    assert res.returncode == 0 and "headless_scene OK" in res.stdout


SLAB_EXE = os.path.join(ROOT, "examples", "slab_pair")


def _build_slab_example():
    cmd = ["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "slab_pair.cpp"),
           "-L", PKG_DIR, "-lsph_hip", "-Wl,-rpath," + PKG_DIR, "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", SLAB_EXE]
    subprocess.run(cmd, check=True, capture_output=True)


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_slab_example_links_against_the_c_abi(pkg):
    pkg.load_library()
    _build_slab_example()
    assert os.path.exists(SLAB_EXE)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_slab_pair_through_the_c_abi_only(pkg):
    """A C++ host drives two z-slab engines through include/sph_abi.h alone (device-side counts, no host round trip in
    the exchange) and gets the single-engine result bit for bit."""
    pkg.load_library()
    _build_slab_example()
    env = dict(os.environ, LD_LIBRARY_PATH=PKG_DIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    res = subprocess.run([SLAB_EXE, "60000", "24"], capture_output=True, text=True, env=env, timeout=300)
    print(res.stdout, res.stderr)
    assert res.returncode == 0 and "slab_pair OK" in res.stdout
