"""Variant patch: sweep 1's staged group loop reads the FIRST candidate of the next group one group ahead (4 VGPRs), so that a group's arithmetic starts
at once and the window reads of its other candidates pass behind the first candidate's 19 instructions.  The tail's first candidate is in registers already.
usage: sweep1_first_ahead.py <csrc dir>"""
import sys, os
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
old_a = s.index("            const float4* __restrict__ wp = &stage[wv][off];")
old_b = s.index("            __builtin_amdgcn_wave_barrier();\n        } else {                                           // a window that does not fit")
new = '''            const float4* __restrict__ wp = &stage[wv][off];
            uint32_t m = 0;
            float4 Jn = wp[0];                             // the first candidate of the next group: read a group ahead (the window is padded)
            for (; m + UNROLL <= len; m += UNROLL) {       // full groups: no validity tests, immediate LDS offsets
                float4 J[UNROLL];
                J[0] = Jn;
#pragma unroll
                for (int u = 1; u < UNROLL; ++u) J[u] = wp[m + (uint32_t)u];
                Jn = wp[m + (uint32_t)UNROLL];
                cur = min(cur, curEnd);                    // rows MAXN .. MAXN + UNROLL - 1 absorb the writes of a full list
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) visit(J[u], fbits(J[u].w), selfRow);
            }
            if (m < len) {
                cur = min(cur, curEnd);
                visit(Jn, fbits(Jn.w), selfRow);
                for (m += 1u; m < len; ++m) {
                    const float4 Jt = wp[m];
                    cur = min(cur, curEnd);
                    visit(Jt, fbits(Jt.w), selfRow);
                }
            }
'''
s = s[:old_a] + new + s[old_b:]
s = s.replace("__shared__ float4 stage[kB / 64][CAP];", "__shared__ float4 stageFlat[(kB / 64) * CAP + 2 * UNROLL];   // (+ pad: the look-ahead read of the last wave's window stays inside the array)\n    float4 (*const stage)[CAP] = reinterpret_cast<float4 (*)[CAP]>(&stageFlat[0]);")
open(p, "w").write(s)
