"""Preset -> SPHFluidGPU members.

The reference stores "look" presets as `key=value` text (PresetIO.h:14-18) and applies them by
writing public members of SPHFluidGPU (Scene0p::ApplyPresetKV, Scene0p.cpp:2341-2377, :2389-2392,
:2482-2490).  A C++ host keeps using the reference's own PresetIO.cpp / ApplyPresetKV unchanged,
because the HIP shim has the same member names; this module is the same mapping for the Python
mirror, so that the shipped presets can serve as regression scenes.  Only keys that reach the
substep path are applied (sim.*, box.*, look.mixPattern / look.dyePattern, motion.fountain*);
render / audio / camera keys are ignored, like unknown keys are in the reference.
"""
from __future__ import annotations

import re

import numpy as np

_FLOAT_PREFIX = re.compile(r"^[ \t\n\v\f\r]*([+-]?(?:(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?|inf(?:inity)?|nan))", re.IGNORECASE)
_INT_PREFIX = re.compile(r"^[ \t\n\v\f\r]*([+-]?\d+)")


def parse(text: str) -> dict:
    """PresetIO::Parse (PresetIO.cpp:26-40): '#' comments, blank and garbage lines ignored,
    CRLF-safe, first value wins on duplicate keys."""
    kv: dict = {}
    for line in text.split("\n"):
        if line.endswith("\r"):
            line = line[:-1]
        if not line or line[0] == "#":
            continue
        eq = line.find("=")
        if eq <= 0:
            continue
        kv.setdefault(line[:eq], line[eq + 1:])
    return kv


def load(path: str) -> dict:
    with open(path, "r", newline="") as fh:
        return parse(fh.read())


def get_f(kv: dict, key: str, default: float) -> float:
    """PresetIO::GetF (:137-143): strtof prefix parse, default when nothing parses."""
    m = _FLOAT_PREFIX.match(kv.get(key, "")) if key in kv else None
    return float(np.float32(float(m.group(1)))) if m else default


def get_i(kv: dict, key: str, default: int) -> int:
    """PresetIO::GetI (:144-150): strtol base 10 prefix parse."""
    m = _INT_PREFIX.match(kv.get(key, "")) if key in kv else None
    return int(m.group(1)) if m else default


def get_b(kv: dict, key: str, default: bool) -> bool:
    return get_i(kv, key, 1 if default else 0) != 0


def get_f3(kv: dict, key: str, current):
    """PresetIO::GetF3 (:154-164): "x,y,z"; unchanged if missing or not three numbers."""
    if key not in kv:
        return list(current)
    out = []
    for tok in kv[key].replace(",", " ").split()[:3]:
        m = _FLOAT_PREFIX.match(tok)
        if not m or m.end() != len(tok):
            # operator>> stops at the first token it cannot fully convert only if NO prefix parses;
            # a valid prefix followed by garbage still yields the prefix, then the next read fails
            if not m:
                return list(current)
            out.append(float(np.float32(float(m.group(1)))))
            break
        out.append(float(np.float32(float(m.group(1)))))
    return out if len(out) == 3 else list(current)


def apply(fluid, kv: dict, structural: bool = True):
    """Scene0p::ApplyPresetKV restricted to the members of SPHFluidGPU.  `fluid` is the Python
    mirror (engine.SPHFluidGPU) or any object with the reference's member names.  With
    structural=True the keys that only matter at the next ResetSimulation are applied as well
    (jitter, particle count, mix / dye pattern), as the reference does on an explicit load."""
    f = fluid
    for key, member in (("sim.h", "param_h"), ("sim.mass", "param_mass"), ("sim.restDensity", "param_restDensity"),
                        ("sim.gasConstant", "param_gasConstant"), ("sim.viscosity", "param_viscosity"),
                        ("sim.gravityY", "param_gravityY"), ("sim.surfaceTension", "param_surfaceTension"),
                        ("sim.timeStep", "param_timeStep"), ("sim.foamGen", "param_foamGen"),
                        ("sim.foamVelRef", "param_foamVelRef"), ("sim.wallRestitution", "param_wallRestitution"),
                        ("sim.wallFriction", "param_wallFriction")):
        setattr(f, member, get_f(kv, key, getattr(f, member)))
    if structural:                                                        # :2352-2355, :2360-2364
        f.param_useJitter = 1 if get_b(kv, "sim.useJitter", bool(f.param_useJitter)) else 0
        f.param_jitterAmp = get_f(kv, "sim.jitterAmp", f.param_jitterAmp)
        f.numParticles = max(1000, get_i(kv, "sim.particleCount", int(f.numParticles)))
        f.param_mixPattern = get_i(kv, "look.mixPattern", f.param_mixPattern)
        f.param_dyePattern = get_i(kv, "look.dyePattern", f.param_dyePattern)
    f.param_boxCenter = get_f3(kv, "box.center", f.param_boxCenter)       # :2366-2375
    f.param_boxHalf = get_f3(kv, "box.half", f.param_boxHalf)
    f.param_boxEulerDeg = get_f3(kv, "box.euler", f.param_boxEulerDeg)
    f.param_shapeType = get_i(kv, "box.shapeType", f.param_shapeType)
    f.param_shapeAux = get_f3(kv, "box.aux", f.param_shapeAux)
    f.fountainMode = 1 if get_b(kv, "motion.fountainOn", bool(f.fountainMode)) else 0   # :2482-2490
    f.fountainOffset = get_f3(kv, "motion.fountainPos", f.fountainOffset)
    f.fountainRadius = get_f(kv, "motion.fountainRadius", f.fountainRadius)
    f.fountainSpread = get_f(kv, "motion.fountainSpread", f.fountainSpread)
    f.fountainDrainLevel = get_f(kv, "motion.fountainDrainLevel", f.fountainDrainLevel)
    f.fountainDrainPerSec = get_f(kv, "motion.fountainDrainRate", f.fountainDrainPerSec)
    # Scene0p keeps the slider value (fountainJetSpeed) and writes fountainJetSpeedLive per frame
    # (audio kick, Scene0p.cpp:3560-3580); without audio the live value is the slider value.
    f.fountainJetSpeedLive = get_f(kv, "motion.fountainJet", f.fountainJetSpeedLive)
    return f
