#!/usr/bin/env python3
"""Run the random scenes of tests/test_gpu_fuzz.py for a range of seeds beyond the committed ones (bug hunting on the GPU box).
usage: fuzz_sweep.py first last [what: bit 0 = z-slabs too, bit 1 = call sequences too, bit 2 = without the plain scenes, bit 3 = call sequences on z-slabs; default 1]"""
import importlib, os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import PKG_NAME
import test_gpu_fuzz as tf
from oracle import oracle
oracle.lib()
pkg = importlib.import_module(PKG_NAME)
first, last = int(sys.argv[1]), int(sys.argv[2])
slabs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
bad = []
for seed in range(first, last):
    for name, fn in (("single", tf.test_random_scene_against_the_oracle), ("slabs", tf.test_random_scene_as_z_slabs_with_boundary_first_steps),
                     ("calls", tf.test_random_call_sequences_against_the_oracle), ("slab calls", tf.test_random_call_sequences_on_z_slabs)):
        if (name == "slabs" and not (slabs & 1)) or (name == "calls" and not (slabs & 2)) or (name == "single" and (slabs & 4)) or (name == "slab calls" and not (slabs & 8)):
            continue
        try:
            fn(pkg, oracle, seed)
        except BaseException as ex:                     # pytest.skip raises a BaseException subclass
            if type(ex).__name__ == "Skipped":
                continue
            msg = str(ex).splitlines()[0][:300] if str(ex) else type(ex).__name__
            bad.append((seed, name, msg))
            print("FAIL", seed, name, msg, flush=True)
    if seed % 10 == 0:
        print("seed", seed, "done", flush=True)
print("failures:", len(bad))
for b in bad:
    print(b)
