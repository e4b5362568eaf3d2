"""Variant patch: a tiny pre-pass (k_wave_plan, one thread per wave of 64 sorted slots) hands every wave of k_sph_walk the bounds [A, B) of its nine candidate-row
windows, so that (i) the first window's loads go out one memory round trip earlier -- the shipped prologue is own record -> 18 run bounds -> first window, three
DEPENDENT round trips, 12 % of a wave's life -- and (ii) the per-row ballot / ffs / readlane / compare work of plan() disappears.  The bounds come from the wave's
first and last live slot (run starts / ends ascend with the slot) and are a SUPERSET of what plan() derives (equal unless the first / last lanes' own runs are empty);
a row for which either end lane's candidate row lies outside the grid is marked and planned in the kernel as before.  Launches that do not start at slot 0 (z-slab
face / interior ranges) keep the in-kernel plan.  Same candidates, same order: same bits.  usage: wave_plan.py <csrc dir>"""
import os, sys
d = sys.argv[1]
p = os.path.join(d, "sph_walk.h")
s = open(p).read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (s.count(a), a[:70])
    s = s.replace(a, b)
# the pre-pass
rep("template <int MAXN, int UNROLL, int CAP, bool SMALLH>\n__global__ __launch_bounds__(256, SPH_WALK_WAVES) void k_sph_walk(",
'''constexpr uint32_t kPlanInKernel = 0xFFFFFFFFu;
__global__ __launch_bounds__(256) void k_wave_plan(SimK k, const float4* __restrict__ own, const uint32_t* __restrict__ cellStart, const uint32_t* __restrict__ liveCount,
                                                   int n, uint32_t* __restrict__ tab) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    const int bound = liveCount ? min(n, (int)*liveCount) : n;
    if (w * 64 >= ((n + 63) / 64) * 64) return;
    uint32_t* t = tab + (size_t)w * 32;
    if (w * 64 >= bound) { for (int i = 0; i < 18; ++i) t[i] = 0u; return; }
    const uint32_t c0 = fbits(own[w * 64].x), c1 = fbits(own[min(w * 64 + 63, bound - 1)].x);
    const int cx0 = (int)(c0 & 1023u), cy0 = (int)((c0 >> 10) & 1023u), cz0 = (int)(c0 >> 20);
    const int cx1 = (int)(c1 & 1023u), cy1 = (int)((c1 >> 10) & 1023u), cz1 = (int)(c1 >> 20);
    const int xlo0 = max(cx0 - 1, 0), xhi1 = min(cx1 + 1, k.gx - 1);
    for (int r = 0; r < 9; ++r) {
        const int nz0 = cz0 + r / 3 - 1, ny0 = cy0 + r % 3 - 1, nz1 = cz1 + r / 3 - 1, ny1 = cy1 + r % 3 - 1;
        const bool in0 = nz0 >= 0 && nz0 < k.gz && ny0 >= 0 && ny0 < k.gy, in1 = nz1 >= 0 && nz1 < k.gz && ny1 >= 0 && ny1 < k.gy;
        uint32_t A = kPlanInKernel, B = 0u;
        if (in0 && in1) { A = cellStart[(nz0 * k.gy + ny0) * k.gx + xlo0]; B = cellStart[(nz1 * k.gy + ny1) * k.gx + xhi1 + 1]; }
        else if (!in0 && !in1 && cy0 == cy1 && cz0 == cz1) { A = 0u; B = 0u; }      // the whole wave sits in one row of cells whose neighbour row is outside the grid
        t[2 * r] = A; t[2 * r + 1] = B;
    }
}

template <int MAXN, int UNROLL, int CAP, bool SMALLH>
__global__ __launch_bounds__(256, SPH_WALK_WAVES) void k_sph_walk(''')
rep("const uint32_t* __restrict__ rangeHi) {", "const uint32_t* __restrict__ rangeHi, const uint32_t* __restrict__ planTab) {")
# the wave's table (wave-uniform pointer -> scalar loads)
rep("    const float4 P = S.P(s), V = S.V(s), O = S.own[s];",
'''    const bool tabled = planTab != nullptr && first == 0 && !ends;
    const uint32_t* __restrict__ tab = planTab + (size_t)__builtin_amdgcn_readfirstlane(((first + vb * kB) >> 6) + wv) * 32;
    const float4 P = S.P(s), V = S.V(s), O = S.own[s];''')
rep("    auto plan = [&](uint32_t q0, uint32_t q1) {\n        const bool ne = q1 > q0;\n        mneN = __ballot(ne);\n        aN = bN = 0u; stagedN = false;\n        if (mneN == 0ull) return;",
'''    auto plan = [&](int r, uint32_t q0, uint32_t q1) {
        aN = bN = 0u; stagedN = false;
        uint32_t tA = kPlanInKernel, tB = 0u;
        if (tabled) { tA = tab[2 * r]; tB = tab[2 * r + 1]; }
        if (tA != kPlanInKernel) {                          // (wave-uniform) the bounds were handed over
            mneN = tB > tA ? 1ull : 0ull;
            if (mneN == 0ull) return;
            aN = tA; bN = tB;
            stagedN = (tB - tA) <= (uint32_t)CAP && !(dbg & 4);
            if (stagedN) {
                const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(S.pv + 2u * (size_t)tA), 0, (int)((tB - tA) * 32u), 0x00020000);
                pre0 = buf_load4(rw, laneOff32);
                if (CAP > 64) pre1 = buf_load4(rw, laneOff32 + 2048u);
                if (CAP > 128) pre2 = buf_load4(rw, laneOff32 + 4096u);
            }
            return;
        }
        const bool ne = q1 > q0;
        mneN = __ballot(ne);
        if (mneN == 0ull) return;''')
rep("    plan(qs[0], qe[0]);\n", "    plan(0, qs[0], qe[0]);\n")
rep("        if (r < 8) plan(qs[r + 1], qe[r + 1]);", "        if (r < 8) plan(r + 1, qs[r + 1], qe[r + 1]);")
open(p, "w").write(s)
# the engine: buffer + launch
e = os.path.join(d, "sph_engine.hip")
t = open(e).read()
def rept(a, b, cnt=1):
    global t
    assert t.count(a) == cnt, (t.count(a), a[:70])
    t = t.replace(a, b)
rept("    float4 *d_sPV = nullptr, *d_sOwn = nullptr;", "    uint32_t* d_planTab = nullptr; size_t planCap = 0;\n    float4 *d_sPV = nullptr, *d_sOwn = nullptr;")
rept("        Timed t(e, SPH_K_SPH);\n        if (e->optNeighbor >= 3 && (size_t)n < ((size_t)1 << 27)) {",
'''        if (e->optNeighbor >= 3 && (size_t)n < ((size_t)1 << 27)) {
            const size_t waves = ((size_t)n + 63) / 64;
            if (e->planCap < waves) { dev_free(e->d_planTab); if ((rc = dev_alloc(&e->d_planTab, waves * 32))) return rc; e->planCap = waves; }
            Timed tp(e, SPH_K_SCATTER);
            hipLaunchKernelGGL(k_wave_plan, dim3((unsigned)((waves + 255) / 256)), dim3(256), 0, e->stream, k, (const float4*)e->d_sOwn, e->d_cellStart, live, n, e->d_planTab);
        }
        Timed t(e, SPH_K_SPH);
        if (e->optNeighbor >= 3 && (size_t)n < ((size_t)1 << 27)) {''')
rept("e->d_order, e->d_cellStart, live, n, dbg, e->d_stats, lo, hi);", "e->d_order, e->d_cellStart, live, n, dbg, e->d_stats, lo, hi, e->d_planTab);", 2)
open(e, "w").write(t)
