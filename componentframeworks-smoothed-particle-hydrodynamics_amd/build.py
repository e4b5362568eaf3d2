"""In-tree build of the HIP engine (libsph_hip.so) for gfx950.

`hipcc` cross-compiles without a GPU, so this runs in the CPU-only build container and
the resulting .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libsph_hip.so")
SOURCES = ["sph_engine.hip"]
HEADERS = ["sph_device.h", "sph_host.h", "sph_kernels.h", "sph_pass.h", "sph_walk.h", "sph_shapes_ext.h", os.path.join("..", "..", "include", "sph_abi.h")]

# -ffp-contract=off: the arithmetic contract (DESIGN.md "Numerics") fixes where fmaf() is
# used; the compiler must not fuse anything else.  Division / sqrt stay at hipcc's
# correctly rounded defaults (no -ffast-math, no -fno-hip-fp32-correctly-rounded-divide-sqrt).
HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP engine cannot be built")


def csrc_hash() -> str:
    """sha256 over the engine's sources (and the flags they are built with): ties a profile under profiles/ to the code it measured."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), *HIPCC_FLAGS, "-o", LIB_PATH, *[os.path.join(CSRC, s) for s in SOURCES]]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
