"""Variant patch: the walks of sweeps 2 / 3 keep AHEAD entries' gathers in flight (shipped: 2 ahead out of 4 register sets).
AHEAD + 1 register sets, the loop unrolled over them.  usage: walk_ahead.py <csrc dir>   (AHEAD from env WALK_AHEAD, default 3)"""
import os, sys
d = sys.argv[1]
D = int(os.environ.get("WALK_AHEAD", "3"))
p = os.path.join(d, "sph_walk.h")
s = open(p).read()
a = s.index("        const uint32_t end = cur;                          // <= curEnd - kRowBytes here")
b = s.index("    static_assert(kSpare >= 2,")
new = f'''        const uint32_t end = cur;                          // <= curEnd - kRowBytes here
        uint32_t at = (uint32_t)tid * 2u;
        constexpr int D = {D}, SETS = D + 1;
        float4 J[SETS], V[SETS];
#pragma unroll
        for (int i = 0; i < SETS; ++i) J[i] = V[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int i = 0; i < D; ++i) if (at + (uint32_t)i * kRowBytes < end) fetch(at + (uint32_t)i * kRowBytes, J[i], V[i]);
        while (at < end) {{
#pragma unroll
            for (int k = 0; k < SETS; ++k) {{
                if (at + (uint32_t)D * kRowBytes < end) fetch(at + (uint32_t)D * kRowBytes, J[(k + D) % SETS], V[(k + D) % SETS]);
                f(J[k], V[k]);
                at += kRowBytes;
                if (!(at < end)) break;
            }}
        }}
    }};
'''
s = s[:a] + new + s[b:]
s = s.replace("constexpr int kSpare = UNROLL > 2 ? UNROLL : 2;", f"constexpr int kSpare = UNROLL > {D} ? UNROLL : {D};")
open(p, "w").write(s)
