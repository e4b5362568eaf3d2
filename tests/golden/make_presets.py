#!/usr/bin/env python3
"""Extracts, from the 13 presets the reference ships (ComponentFramework/presets/*.txt, data
files), the keys that reach the substep path and writes them to presets.json next to this
script.  Run in the build container (the reference tree is not present on the GPU box):

    python tests/golden/make_presets.py [/root/reference/ComponentFramework/presets]

Values are kept as the strings the files hold, so the loader's parsing is what is tested."""
import json
import os
import sys

KEEP = ("sim.", "box.center", "box.half", "box.euler", "box.shapeType", "box.aux", "look.mixPattern", "look.dyePattern",
        "motion.fountain")


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/ComponentFramework/presets"
    out = {}
    for name in sorted(os.listdir(src)):
        if not name.endswith(".txt"):
            continue
        kv = {}
        with open(os.path.join(src, name), newline="") as fh:
            for line in fh.read().split("\n"):
                line = line.rstrip("\r")
                if not line or line[0] == "#" or "=" not in line[1:]:
                    continue
                k, v = line.split("=", 1)
                if k.startswith(KEEP):
                    kv.setdefault(k, v)
        out[name[:-4]] = kv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "presets.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(path, len(out), "presets")


if __name__ == "__main__":
    main()
