"""Variant patch: the list's second ball with ONE exact component of 2 m.d (the y axis: gravity's, in every BASELINE workload) and the other two bounded:
|d + m|^2 - hp^2 >= r2 + 2 m_y d_y + |m|^2 - hp^2 - 2 |m_perp| (hp + |m|)  (a candidate inside the ball has |d| < hp + |m|), so the test stays a SUPERSET
of the shipped one (same bits; lists a little longer where the fluid moves across y) and costs 2 vector instructions per candidate instead of 4.
usage: list_test_one_axis.py <csrc dir>"""
import os, sys
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
old1 = "    const float c0 = mm - (hp * hp) * 1.0001f;"
assert s.count(old1) == 1
s = s.replace(old1, '''    const float mperp = sqrtf(fmaf(mvz, mvz, mvx * mvx));
    const float c0 = mm - (hp * hp) * 1.0001f - 2.0002f * mperp * (hp + sqrtf(mm));''')
old2 = "        return fmaf(ez, dz, fmaf(ey, dy, fmaf(ex, dx, r2 + c0)));"
assert s.count(old2) == 1
s = s.replace(old2, "        return fmaf(ey, dy, r2 + c0);")
open(p, "w").write(s)
