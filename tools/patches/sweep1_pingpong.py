"""Variant patch: sweep 1's staged group loop software-pipelined WITHOUT register copies: two register sets, the loop unrolled by two, the next group's
window reads in flight during this group's arithmetic; the tail of a run (1 or 2 candidates) is already in registers.  usage: sweep1_pingpong.py <csrc dir>"""
import sys, os
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
old_a = s.index("            const float4* __restrict__ wp = &stage[wv][off];")
old_b = s.index("            __builtin_amdgcn_wave_barrier();\n        } else {                                           // a window that does not fit")
new = '''            const float4* __restrict__ wp = &stage[wv][off];
            uint32_t m = 0;
            float4 Ja[UNROLL], Jb[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { Ja[u] = wp[(uint32_t)u]; Jb[u] = Ja[u]; }
            bool tailInA = true;                           // the set that holds the candidates not visited yet
            while (m + UNROLL <= len) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) Jb[u] = wp[m + (uint32_t)(UNROLL + u)];
                cur = min(cur, curEnd);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) visit(Ja[u], fbits(Ja[u].w), selfRow);
                m += UNROLL;
                if (!(m + UNROLL <= len)) { tailInA = false; break; }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) Ja[u] = wp[m + (uint32_t)(UNROLL + u)];
                cur = min(cur, curEnd);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) visit(Jb[u], fbits(Jb[u].w), selfRow);
                m += UNROLL;
            }
            cur = min(cur, curEnd);
#pragma unroll
            for (int u = 0; u < UNROLL - 1; ++u)           // the tail is in registers already
                if (m + (uint32_t)u < len) {
                    const float4 T = tailInA ? Ja[u] : Jb[u];
                    visit(T, fbits(T.w), selfRow);
                }
'''
s = s[:old_a] + new + s[old_b:]
s = s.replace("__shared__ float4 stage[kB / 64][CAP];", "__shared__ float4 stageFlat[(kB / 64) * CAP + 2 * UNROLL];   // (+ pad: the look-ahead reads of the last wave's window stay inside the array)\n    float4 (*const stage)[CAP] = reinterpret_cast<float4 (*)[CAP]>(&stageFlat[0]);")
open(p, "w").write(s)
