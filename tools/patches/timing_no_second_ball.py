"""TIMING-ONLY variant (results are NOT the contract's): sweep 1 without the second ball of the list test (no 2 m.d term: 4 of the 17 vector instructions
per candidate gone, lists a little shorter).  Measures what a cheaper list test could buy at most.  usage: timing_no_second_ball.py <csrc dir>"""
import os, sys
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
old = "        return fmaf(ez, dz, fmaf(ey, dy, fmaf(ex, dx, r2 + c0)));"
assert s.count(old) == 1
s = s.replace(old, "        return 1.0f;")
open(p, "w").write(s)
