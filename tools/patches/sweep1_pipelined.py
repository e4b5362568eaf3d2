"""Variant patch: sweep 1's staged group loop with the NEXT group's window reads in flight during this group's arithmetic
(shipped: ds_read x3, s_waitcnt, arithmetic).  The reads are unconditional (the window is padded), the tail of a run (1 or 2
candidates) is already in registers: no singles loop.  usage: sweep1_pipelined.py <csrc dir>"""
import sys, os
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
old_a = s.index("            const float4* __restrict__ wp = &stage[wv][off];")
old_b = s.index("            __builtin_amdgcn_wave_barrier();\n        } else {                                           // a window that does not fit")
new = '''            const float4* __restrict__ wp = &stage[wv][off];
            uint32_t m = 0;
            float4 Jn[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) Jn[u] = wp[(uint32_t)u];
            for (; m + UNROLL <= len; m += UNROLL) {       // full groups: no validity tests; the next group's reads are in flight
                float4 J[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) J[u] = Jn[u];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) Jn[u] = wp[m + (uint32_t)(UNROLL + u)];
                cur = min(cur, curEnd);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) visit(J[u], fbits(J[u].w), selfRow);
            }
            cur = min(cur, curEnd);
#pragma unroll
            for (int u = 0; u < UNROLL - 1; ++u)           // the tail is in registers already
                if (m + (uint32_t)u < len) visit(Jn[u], fbits(Jn[u].w), selfRow);
'''
s = s[:old_a] + new + s[old_b:]
s = s.replace("__shared__ float4 stage[kB / 64][CAP];", "__shared__ float4 stageFlat[(kB / 64) * CAP + 2 * UNROLL];   // (+ pad: the look-ahead reads of the last wave's window stay inside the array)\n    float4 (*const stage)[CAP] = reinterpret_cast<float4 (*)[CAP]>(&stageFlat[0]);")
open(p, "w").write(s)
