"""Variant patch: the 18 run bounds of a target's nine candidate rows are loaded three rows ahead of their use instead of all up front
(frees about twelve VGPRs through sweep 1).  usage: rolling_bounds.py <csrc dir>"""
import sys, os
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
old = '''#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
        const bool in = live && nz >= 0 && nz < k.gz && ny >= 0 && ny < k.gy;
        const int rowBase = in ? (nz * k.gy + ny) * k.gx : 0;
        const uint32_t a = cellStart[rowBase + xlo], b = cellStart[rowBase + xhi + 1];
        qs[r] = in ? a : 0u; qe[r] = in ? b : 0u;
    }
'''
assert s.count(old) == 1
new = '''    auto bounds = [&](int r) {
        const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
        const bool in = live && nz >= 0 && nz < k.gz && ny >= 0 && ny < k.gy;
        const int rowBase = in ? (nz * k.gy + ny) * k.gx : 0;
        const uint32_t a = cellStart[rowBase + xlo], b = cellStart[rowBase + xhi + 1];
        qs[r] = in ? a : 0u; qe[r] = in ? b : 0u;
    };
    constexpr int kBoundsAhead = 3;
#pragma unroll
    for (int r = 0; r < kBoundsAhead; ++r) bounds(r);
'''
s = s.replace(old, new)
old2 = "        if (r < 8) plan(qs[r + 1], qe[r + 1]);"
assert s.count(old2) == 1
s = s.replace(old2, "        if (r + kBoundsAhead < 9) bounds(r + kBoundsAhead);\n" + old2)
open(p, "w").write(s)
