"""Variant patch: the exact fallback of sweeps 2 / 3 for waves of compressed fluid as ACCEPTED-PAIR WALKS OUT OF LDS.
Per candidate row and chunk of CH candidates: the wave stages the chunk's 32-byte records into its window, every lane runs over ITS part of the
chunk (one exact r2 per candidate, branch-free append of the accepted ones to a per-lane list in LDS), then walks its list with both halves of
the record read from LDS (no gather, no L1 tag) through the same pair functions.  Same candidates in the same order as the plain sweep, a
candidate that is not accepted adds exactly +-0 there: same bits.  Enabled per dispatch by SPH_OPT_DEBUG bit 6 (64); DENSE_MIN lanes of a wave
must need the fallback (env DENSE_MIN, default 1).  usage: dense_lds_walk.py <csrc dir>"""
import os, sys
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
CH = int(os.environ.get("DENSE_CH", "40"))
DMIN = int(os.environ.get("DENSE_MIN", "1"))
anchor = "    auto force_at = [&](const float4& J, const float4& JV) { pair_force_other(k, o, J, JV); };"
assert s.count(anchor) == 1
dense = f'''    // ---- dense path (experiment): accepted-pair walks out of LDS ----
    auto dense = [&](auto&& f) {{
        constexpr uint32_t CH = {CH}u;
        static_assert(2 * CH <= CAP && CH <= MAXN, "a chunk's records fit the wave's window, its accepted candidates a lane's list");
        float4* const win = &stage[wv][0];
        const uint32_t adv512 = live ? kRowBytes : 0u;
        for (int r = 0; r < 9; ++r) {{
            const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
            const bool in = live && nz >= 0 && nz < k.gz && ny >= 0 && ny < k.gy;
            const int rowBase = in ? (nz * k.gy + ny) * k.gx : 0;
            uint32_t a = cellStart[rowBase + xlo], b = cellStart[rowBase + xhi + 1];
            if (!in) {{ a = 0u; b = 0u; }}
            const unsigned long long mne = __ballot(b > a);
            if (mne == 0ull) continue;
            const int lf = __ffsll((long long)mne) - 1, ll = 63 - __clzll((long long)mne);
            const uint32_t A = (uint32_t)__builtin_amdgcn_readlane((int)a, lf), B = (uint32_t)__builtin_amdgcn_readlane((int)b, ll);
            for (uint32_t c = A; c < B; c += CH) {{
                const uint32_t nf = min(CH, B - c) * 2u;                  // float4s of this chunk
                if ((uint32_t)lane < nf) win[lane] = S.pv[2u * (size_t)c + (uint32_t)lane];
                if ((uint32_t)lane + 64u < nf) win[lane + 64] = S.pv[2u * (size_t)c + (uint32_t)lane + 64u];
                __builtin_amdgcn_wave_barrier();
                const uint32_t lo = (a > c ? a : c) - c;
                const uint32_t hiAbs = min(b, c + CH);
                const uint32_t hi = hiAbs > c ? hiAbs - c : 0u;
                uint32_t lc = (uint32_t)tid * 2u;
                for (uint32_t m = lo; m < hi; ++m) {{
                    const float4 J = win[2u * m];
                    const float dx = o.px - J.x, dy = o.py - J.y, dz = o.pz - J.z;
                    float sgn = dot3(dx, dy, dz, dx, dy, dz) - k.h2;      // < 0: within h
                    if (r == 4) sgn = (c + m == (uint32_t)s) ? 1.0f : sgn;
                    *reinterpret_cast<uint16_t*>(nlBytes + lc) = (uint16_t)m;
                    lc += (fbits(sgn) >> 22) & adv512;
                }}
                for (uint32_t at = (uint32_t)tid * 2u; at < lc; at += kRowBytes) {{
                    const uint32_t m = *reinterpret_cast<const uint16_t*>(nlBytes + at);
                    f(win[2u * m], win[2u * m + 1u]);
                }}
                __builtin_amdgcn_wave_barrier();
            }}
        }}
    }};
'''
s = s.replace(anchor, dense + anchor)
s = s.replace("    if (listOk) listed(force_at); else plain(force_plain);",
f'''    const bool useDense = (dbg & 64) != 0;
    if (useDense && __popcll(__ballot(live && !listOk)) >= {DMIN}) dense(force_at);
    else if (listOk) listed(force_at); else plain(force_plain);''')
s = s.replace("    if (listOk && near) listed(xsph_at); else plain(xsph_plain);",
f'''    if (useDense && __popcll(__ballot(live && !(listOk && near))) >= {DMIN}) dense(xsph_at);
    else if (listOk && near) listed(xsph_at); else plain(xsph_plain);''')
open(p, "w").write(s)
