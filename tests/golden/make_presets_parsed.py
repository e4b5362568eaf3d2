#!/usr/bin/env python3
"""Pins the preset loader (componentframeworks-..._amd/presets.py) to the REFERENCE's own parser: builds
oracle/_ref/presetio_dump (the reference's PresetIO.cpp, compiled where it lies, + this repo's driver), runs it on the
13 presets the reference ships and on a handful of edge-case files, and writes what PresetIO::Parse / GetF / GetI / GetB /
GetF3 return to tests/golden/presets_parsed.json.  Build container only (the reference tree does not travel):

    python tests/golden/make_presets_parsed.py [/root/reference/ComponentFramework]
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

EDGE = {
    "edge_crlf_dups": "# SPH Fluid Preset v1\r\nversion=1\r\n\r\nsim.h=0.25\nsim.h=0.5\n=novalue\ngarbage line\nbox.half=1,2,3\nlook.name=a=b\n#x=1\n",
    "edge_numbers": "edge.a=1.5abc\nedge.b=abc\nedge.c= -7\nedge.d=3.9\nedge.e=1,2\nedge.f=1, 2 ,3.5\nedge.g=0.100000001\nedge.h=1e2\nedge.i=x,1,2\n"
                    "edge.j=1.5,2.5e0,-3\nedge.nan=nan\nedge.sp=  4.25  \nedge.zero=0\nsim.useJitter=2\nmotion.fountainOn=0\n",
    "edge_empty": "",
}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/ComponentFramework"
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref", f"REF={ref}"], check=True, capture_output=True)
    exe = os.path.join(ROOT, "oracle", "_ref", "presetio_dump")
    pdir = os.path.join(ref, "presets")
    files = {name[:-4]: os.path.join(pdir, name) for name in sorted(os.listdir(pdir)) if name.endswith(".txt")}
    out = {"_source": "PresetIO::LoadFile/GetF/GetI/GetB/GetF3 of the reference (PresetIO.cpp:26-57,137-164), run by tests/golden/make_presets_parsed.py; "
                      "floats as hex of their fp32 bits; sentinel defaults f=-12345.5, i=-777, v=(-1.25,-2.5,-3.75)",
           "presets": {}, "edge": {}, "edge_text": EDGE}
    with tempfile.TemporaryDirectory() as td:
        for k, text in EDGE.items():
            p = os.path.join(td, k + ".txt")
            with open(p, "w", newline="") as fh:
                fh.write(text)
            files["__" + k] = p
        res = json.loads(subprocess.run([exe, *files.values()], check=True, capture_output=True, text=True).stdout)
    for name, path in files.items():
        (out["edge"] if name.startswith("__") else out["presets"])[name.lstrip("_")] = res[path]
    dst = os.path.join(HERE, "presets_parsed.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(dst, len(out["presets"]), "presets,", len(out["edge"]), "edge cases")


if __name__ == "__main__":
    main()
