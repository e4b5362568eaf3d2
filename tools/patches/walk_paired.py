"""Variant patch: the walks of sweeps 2 / 3 fetch BOTH halves of a list entry's 32-byte record with ONE load instruction.
Shipped: two buffer_load_dwordx4 per entry (x, y, z, 1/rho | vx, vy, vz, P), i.e. every cache line is looked up twice, and the pass is bound by L1 tag
lookups.  Here lanes L and L + 32 of a wave work as a pair: instruction 1 loads the record of lane L's entry (lane L the first 16 bytes, lane L + 32 the
second), instruction 2 the record of lane L + 32's entry, and four v_permlane32_swap (gfx950) turn the two registers into (J, JV) on both lanes.  The pair
walks max(own entries, partner's entries) steps; a lane without an entry loads out of range (no fetch).  Data movement only: same bits.
usage: walk_paired.py <csrc dir>"""
import os, sys
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
a = s.index("    auto fetch = [&](uint32_t at, float4& J, float4& JV) {")
b = s.index("    static_assert(kSpare >= 2,")
new = r'''    auto swap32 = [&](float& x, float& y) {                 // lanes 32..63 of x <-> lanes 0..31 of y
        const auto r = __builtin_amdgcn_permlane32_swap(fbits(x), fbits(y), false, false);
        x = bitsf(r[0]); y = bitsf(r[1]);
    };
    auto swap32u = [&](uint32_t& x, uint32_t& y) {
        const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
        x = r[0]; y = r[1];
    };
    const uint32_t hi16 = ((uint32_t)lane & 32u) >> 1;     // the upper lane of a pair loads the second half of a record
    // cnt: this lane's entries to walk (0: none -- it still helps its partner)
    auto listed = [&](uint32_t cnt, auto&& f) {
        uint32_t c1 = cnt, c2 = cnt;
        swap32u(c1, c2);                                   // c1: entries of the pair's lower lane, c2: of its upper lane (on both lanes)
        const uint32_t nPair = max(c1, c2);
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
            uint32_t q1 = qb, q2 = qb;
            swap32u(q1, q2);                               // q1: the record of the lower lane's entry, q2: of the upper lane's (on both lanes)
            R1 = buf_load4(bufPV, q1 + hi16);
            R2 = buf_load4(bufPV, q2 + hi16);
        };
#pragma unroll
        for (int i = 0; i < D; ++i) if ((uint32_t)i < nPair) pfetch((uint32_t)i, A[i], B[i]);
        uint32_t i = 0;
        while (i < nPair) {
#pragma unroll
            for (int k = 0; k < SETS; ++k) {
                if (i + (uint32_t)D < nPair) pfetch(i + (uint32_t)D, A[(k + D) % SETS], B[(k + D) % SETS]);
                float4 J = A[k], JV = B[k];                // lower lane: (first half of ITS record, first half of the partner's); upper lane: (second half of the partner's, second half of ITS)
                swap32(J.x, JV.x); swap32(J.y, JV.y); swap32(J.z, JV.z); swap32(J.w, JV.w);
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
