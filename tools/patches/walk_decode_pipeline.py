"""Variant patch: the walks of sweeps 2 / 3 decode their list entries AHEAD of the loads that need them.
Shipped: every fetch is ds_read_u16 (entry) -> s_waitcnt -> ds_read_b32 (row base) -> s_waitcnt -> two buffer loads: two exposed LDS round trips per entry in
the wave's in-order instruction stream, in front of the pair arithmetic.  Here the entry of i + 4 and the row base of i + 3 are read one iteration before they
are needed (their latency passes during entry i's pair arithmetic), the loads of i + 2 are issued from values that arrived an iteration ago.  Same loads, same
order of the arithmetic: same bits.  usage: walk_decode_pipeline.py <csrc dir>"""
import os, sys
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
a = s.index("    auto fetch = [&](uint32_t at, float4& J, float4& JV) {")
b = s.index("    static_assert(kSpare >= 2,")
new = r'''    auto listed = [&](auto&& f) {
        const uint32_t at0 = (uint32_t)tid * 2u;
        const uint32_t n = (cur - at0) / kRowBytes;            // this lane's entries
        constexpr uint32_t kLastRow = (uint32_t)(MAXN + kSpare - 1);
        auto readEnt = [&](uint32_t i) -> uint32_t {           // (a row past the list holds a stale entry: decoded, never loaded from)
            return *reinterpret_cast<const uint16_t*>(nlBytes + at0 + min(i, kLastRow) * kRowBytes);
        };
        auto readBase = [&](uint32_t ent) -> uint32_t { return *reinterpret_cast<const uint32_t*>(rowBytes + ((ent >> 10) & 0x3cu)); };
        auto offOf = [&](uint32_t ent) -> uint32_t { return (ent & 0xff0u) << 1; };
        constexpr int D = 2, SETS = D + 1;
        float4 J[SETS], V[SETS];
#pragma unroll
        for (int i = 0; i < SETS; ++i) J[i] = V[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const uint32_t e0 = readEnt(0u), e1 = readEnt(1u), e2 = readEnt(2u), e3 = readEnt(3u);
        const uint32_t b0 = readBase(e0), b1 = readBase(e1), b2 = readBase(e2);
        if (0u < n) { const uint32_t q = b0 + offOf(e0); J[0] = buf_load4(bufPV, q); V[0] = buf_load4(bufPV, q + 16u); }
        if (1u < n) { const uint32_t q = b1 + offOf(e1); J[1] = buf_load4(bufPV, q); V[1] = buf_load4(bufPV, q + 16u); }
        uint32_t qA = b2 + offOf(e2);                          // byte offset of entry i + 2
        uint32_t entB = e3;                                    // entry i + 3 (its row base is read in this iteration)
        uint32_t i = 0;
        while (i < n) {
#pragma unroll
            for (int k = 0; k < SETS; ++k) {
                if (i + 2u < n) { J[(k + 2) % SETS] = buf_load4(bufPV, qA); V[(k + 2) % SETS] = buf_load4(bufPV, qA + 16u); }
                const uint32_t baseNext = readBase(entB);      // arrives during the pair arithmetic below
                const uint32_t entNext = readEnt(i + 4u);
                f(J[k], V[k]);
                qA = baseNext + offOf(entB);
                entB = entNext;
                i += 1u;
                if (!(i < n)) break;
            }
        }
    };
'''
s = s[:a] + new + s[b:]
open(p, "w").write(s)
