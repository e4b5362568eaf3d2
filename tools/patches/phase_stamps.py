"""Instrumentation variant (not a product path): every wave of k_sph_walk stamps s_memtime at its phase boundaries and lanes 0..4 write the differences into the unused
fourth word of their particle's acc (SPH_OPT_DEBUG bit 3; read back through the 80-byte records): [0] prologue (own loads, 18 run bounds: until the first row is planned), [1] sweep 1
(nine rows), [2] sweep 2 + integrate, [3] sweep 3 + finish + stores, [4] the first THREE rows of sweep 1 alone, [7] waves.  Cycles of the shader clock per wave,
summed.  usage: phase_stamps.py <csrc dir>"""
import os, sys
p = os.path.join(sys.argv[1], "sph_walk.h")
s = open(p).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a[:60]
    s = s.replace(a, b)
rep("    const int sRaw = first + vb * kB + tid;", "    const unsigned long long tStart = __builtin_amdgcn_s_memtime();\n    const int sRaw = first + vb * kB + tid;")
rep("    plan(qs[0], qe[0]);\n", "    plan(qs[0], qe[0]);\n    unsigned long long tProl = 0ull, tRow3 = 0ull;\n    if (qs[0] + qe[8] + 1u != 0u) tProl = __builtin_amdgcn_s_memtime();\n")
rep("        if (mne == 0ull) continue;", "        if (r == 3) tRow3 = __builtin_amdgcn_s_memtime();\n        if (mne == 0ull) continue;")
rep("    finish_density(k, o);\n", "    finish_density(k, o);\n    unsigned long long tS1 = 0ull; if (o.rho >= 0.0f) tS1 = __builtin_amdgcn_s_memtime();\n")
rep("    integrate(k, o);\n", "    integrate(k, o);\n    unsigned long long tS2 = 0ull; if (o.px == o.px) tS2 = __builtin_amdgcn_s_memtime();\n")
a = s.index("    if (dbg & 8) {   // diagnostics:")
b2 = s.index("}\n\n}  // namespace sph")
s = s[:a] + '''    if (dbg & 8) {   // (no atomics: 65 536 waves adding to the same words take milliseconds on this chip; the differences go out in the unused fourth word of acc)
        const unsigned long long tEnd = __builtin_amdgcn_s_memtime();
        const unsigned long long d0 = tProl - tStart, d1 = tS1 - tProl, d2 = tS2 - tS1, d3 = tEnd - tS2, d4 = tRow3 - tProl;
        const unsigned long long dv = lane == 0 ? d0 : lane == 1 ? d1 : lane == 2 ? d2 : lane == 3 ? d3 : d4;
        if (lane < 5 && live) reinterpret_cast<float*>(&out.acc[s])[3] = bitsf((uint32_t)((dv > 0x0fffffffull ? 0x0fffffffull : dv) << 3) | (uint32_t)lane | 0x80000000u);
    }
''' + s[b2:]
open(p, "w").write(s)
