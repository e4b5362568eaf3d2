"""Variant patch: as walk_paired.py, but the two lanes that fetch the halves of one record are NEIGHBOURS (2k, 2k + 1), so that their two 16-byte
accesses fall into one group of the texture addresser and cost ONE cache-line lookup (walk_paired.py showed that lanes 32 apart are not merged).
Instruction 1 loads the even lane's record (even lane: first half, odd lane: second half), instruction 2 the odd lane's record (odd lane: first half,
even lane: second half); the lane's own first half is R1 on even and R2 on odd lanes (four selects), the partner's second half sits in the other
register (four selects) and comes over by a DPP quad permutation.  Data movement only: same bits.  usage: walk_paired_adjacent.py <csrc dir>"""
import os, sys
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
a = s.index("    auto fetch = [&](uint32_t at, float4& J, float4& JV) {")
b = s.index("    static_assert(kSpare >= 2,")
new = r'''    auto swapAdj = [&](uint32_t x) -> uint32_t {             // the value of the neighbouring lane (2k <-> 2k + 1): DPP quad_perm [1, 0, 3, 2]
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);
    };
    const bool evenLane = (lane & 1) == 0;
    // cnt: this lane's entries to walk (0: none -- it still helps its partner)
    auto listed = [&](uint32_t cnt, auto&& f) {
        const uint32_t nPair = max(cnt, swapAdj(cnt));
        constexpr int D = 2, SETS = D + 1;
        float4 A[SETS], B[SETS];
#pragma unroll
        for (int i = 0; i < SETS; ++i) A[i] = B[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        auto pfetch = [&](uint32_t i, float4& R1, float4& R2) {
            uint32_t qb = 0xFFFFFFC0u;                     // no entry: out of the buffer's range (returns 0, fetches nothing)
            if (i < cnt) {
                const uint32_t ent = *reinterpret_cast<const uint16_t*>(nlBytes + (uint32_t)tid * 2u + i * kRowBytes);
                const uint32_t base = *reinterpret_cast<const uint32_t*>(rowBytes + ((ent >> 10) & 0x3cu));
                qb = base + ((ent & 0xff0u) << 1);
            }
            const uint32_t t = swapAdj(qb) + 16u;          // the second half of the partner's record
            R1 = buf_load4(bufPV, evenLane ? qb : t);      // the EVEN lane's record: (first half | second half)
            R2 = buf_load4(bufPV, evenLane ? t : qb);      // the ODD lane's record:  (second half | first half)
        };
#pragma unroll
        for (int i = 0; i < D; ++i) if ((uint32_t)i < nPair) pfetch((uint32_t)i, A[i], B[i]);
        uint32_t i = 0;
        while (i < nPair) {
#pragma unroll
            for (int k = 0; k < SETS; ++k) {
                if (i + (uint32_t)D < nPair) pfetch(i + (uint32_t)D, A[(k + D) % SETS], B[(k + D) % SETS]);
                const float4 R1 = A[k], R2 = B[k];
                const float4 J = evenLane ? R1 : R2;       // this lane's own first half
                const float4 X = evenLane ? R2 : R1;       // the PARTNER's second half
                const float4 JV = make_float4(bitsf(swapAdj(fbits(X.x))), bitsf(swapAdj(fbits(X.y))), bitsf(swapAdj(fbits(X.z))), bitsf(swapAdj(fbits(X.w))));
                if (i < cnt) f(J, JV);
                i += 1u;
                if (!(i < nPair)) break;
            }
        }
    };
'''
s = s[:a] + new + s[b:]
s = s.replace("    if (listOk) listed(force_at); else plain(force_plain);",
              "    listed(listOk ? (cur - (uint32_t)tid * 2u) / kRowBytes : 0u, force_at);\n    if (!listOk) plain(force_plain);")
s = s.replace("    if (listOk && near) listed(xsph_at); else plain(xsph_plain);",
              "    listed((listOk && near) ? (cur - (uint32_t)tid * 2u) / kRowBytes : 0u, xsph_at);\n    if (!(listOk && near)) plain(xsph_plain);")
open(p, "w").write(s)
