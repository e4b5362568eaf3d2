// sph_walk.h -- k_sph_walk: the SPH pass (SPHFluid.comp:66-221 + fused OBBConstraints.comp), round-3 form.
//
// Same plan as k_sph_list (one target per lane over the sorted copy, wave-private LDS windows for the candidate rows,
// sweeps 2 / 3 over a neighbour list) and the same arithmetic contract, therefore the same bits.  What changed is the
// instruction selection, guided by the issue costs measured on gfx950 (profiles/r03_valu_rate2.txt: v_add/sub/mul/fma,
// integer add/and/or/shift-right cost about 2.5 cycles of a SIMD, every v_pk_*, v_cmp, v_cndmask, v_max/min, shift-left
// and three-operand integer form about 4.3, v_cndmask with an implicit vcc far more):
//   * sweep 1 evaluates ONE exact r2 per candidate; the second ball of the list test (within h + eps of the predicted
//     position) comes from it by |d + m|^2 = r2 + 2 m.d + |m|^2 (m = predicted move): 4 fast ops instead of a second distance;
//   * max(h2 - r2, 0) is the subtraction's clamp modifier (h2 <= 1; the template falls back to v_max otherwise);
//   * the accept decision never touches vcc: s = fma(-2^40, t, w) is negative exactly when the candidate is inside either
//     ball, and the list cursor advances by (bits(s) >> 22) & 512;
//   * the particle itself is left out of the list when it is built (sweeps 2 / 3 skip it by contract), so the walks need no
//     per-entry validity mask; they are per-lane loops (no rounding of the list length to a group size), entries decode to
//     byte offsets with two fast ops, and the gathers are bounds-checked buffer loads, issued two entries ahead and only for
//     entries that exist (the pass is as much bound by the cache lines its gathers touch in L1 as by vector issue).
#pragma once
#include "sph_pass.h"

namespace sph {

#ifndef SPH_WALK_MAXN
#define SPH_WALK_MAXN 41     // list entries per target
#endif
#ifndef SPH_WALK_CAP
#define SPH_WALK_CAP 144     // wave-private LDS window of one candidate row (<= 192): on the jittered lattice the rows come with 1, 2 or 4
                             // lattice lines per cell row, and a wave of a thin row next to a thick one needs a window well beyond its 64 targets
#endif
#ifndef SPH_WALK_UNROLL
#define SPH_WALK_UNROLL 3    // candidates per iteration of sweep 1
#endif
#ifndef SPH_WALK_WAVES
#define SPH_WALK_WAVES 5     // __launch_bounds__ minimum waves per SIMD
#endif
#ifndef SPH_WALK_EPS
#define SPH_WALK_EPS 0.04f   // slack of the list around the predicted position, in units of h (0.03 / 0.04 / 0.06 / 0.08: 442 / 441 / 451 / 463 us)
#endif

typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t byteOff) {
    const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byteOff, 0, 0);   // out of range reads return 0, never fault
    return make_float4(bitsf(v.x), bitsf(v.y), bitsf(v.z), bitsf(v.w));
}

// ---- pair functions of sweeps 2 / 3 for a candidate that is NOT the target itself: the arithmetic of pair_force /
// pair_xsph (sph_device.h) with the accept masks formed by integer ops (identical bits: the masks select invRho or +0).
__device__ __forceinline__ void pair_force_other(const SimK& k, Own& o, const float4& J, const float4& JV) {
    const float dx = o.px - J.x, dy = o.py - J.y, dz = o.pz - J.z;
    const float r2 = dot3(dx, dy, dz, dx, dy, dz);
    // ie = (r2 < h2 && invRho > 0) ? invRho : 0.   invRho (J.w) is 1 / rho_j or +0, never negative: the r2 test decides.
    const int32_t in = (int32_t)fbits(r2 - k.h2) >> 31;                 // all ones iff r2 < h2
    const float ie = bitsf(fbits(J.w) & (uint32_t)in);
    const float rinv = t_rsqrt(fmaxf(r2, SPH_TINY));
    const float r = r2 * rinv;
    const float hr = k.h - r;
    const float sr = (k.spikyC * (hr * hr)) * rinv;
    const float gx = sr * dx, gy = sr * dy, gz = sr * dz;
    const float mor = k.mass * ie;
    const float pterm = ((o.prs + JV.w) * k.negHalfMass) * ie;
    const float ml = mor * (k.viscC * hr);
    o.fPx = fmaf(gx, pterm, o.fPx); o.fPy = fmaf(gy, pterm, o.fPy); o.fPz = fmaf(gz, pterm, o.fPz);
    o.fVx = fmaf(JV.x - o.vx, ml, o.fVx); o.fVy = fmaf(JV.y - o.vy, ml, o.fVy); o.fVz = fmaf(JV.z - o.vz, ml, o.fVz);
    o.gCx = fmaf(mor, gx, o.gCx); o.gCy = fmaf(mor, gy, o.gCy); o.gCz = fmaf(mor, gz, o.gCz);
    o.lapC = o.lapC + ml;
}
template <bool SMALLH>
__device__ __forceinline__ void pair_xsph_other(const SimK& k, Own& o, const float4& J, const float4& JV) {
    const float dx = o.px - J.x, dy = o.py - J.y, dz = o.pz - J.z;
    const float r2 = dot3(dx, dy, dz, dx, dy, dz);
    const float t0 = SMALLH ? __builtin_amdgcn_fmed3f(k.h2 - r2, 0.0f, 1.0f) : fmaxf(k.h2 - r2, 0.0f);
    const int32_t has = (int32_t)(0u - fbits(J.w)) >> 31;               // all ones iff invRho > 0 (bits of a positive float)
    const float t = bitsf(fbits(t0) & (uint32_t)has);
    const float w3 = (t * t) * t;
    const float wm = w3 * (k.mass * J.w);
    o.xsx = fmaf(JV.x - o.vx, wm, o.xsx); o.xsy = fmaf(JV.y - o.vy, wm, o.xsy); o.xsz = fmaf(JV.z - o.vz, wm, o.xsz);
    o.norm = o.norm + w3;
}

template <int MAXN, int UNROLL, int CAP, bool SMALLH>
__global__ __launch_bounds__(256, SPH_WALK_WAVES) void k_sph_walk(SimK k, SortedIn S, StateIn in, StateOut out, const uint32_t* __restrict__ order,
                                                                  const uint32_t* __restrict__ cellStart, const uint32_t* __restrict__ liveCount, int n,
                                                                  int dbg, unsigned long long* __restrict__ stats, const uint32_t* __restrict__ rangeLo,
                                                                  const uint32_t* __restrict__ rangeHi) {
    constexpr int kB = 256;
    constexpr uint32_t kRowBytes = kB * 2;                 // one list row = one entry of every thread
    static_assert(kRowBytes == 512, "the cursor advance reads bit 9 of (sign >> 22)");
    constexpr int kSpare = UNROLL > 2 ? UNROLL : 2;        // rows past MAXN: absorb the writes of a full list / the walks' look-ahead
    __shared__ uint16_t nl[MAXN + kSpare][kB];             // entry e of thread t: (row << 12) | ((slot - first slot of the wave's window of that row) << 4)
    __shared__ uint32_t rowA[kB / 64][16];                 // per wave: BYTE offset (slot * 32) of the first slot of the wave's window of each candidate row
    __shared__ float4 stage[kB / 64][CAP];                 // the window: (x, y, z, bits(entry value of this candidate))
    static_assert(CAP <= 192, "the window is staged with three loads per lane, and a window offset has 8 bits in an entry");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    // XCD-aware block mapping as in k_sph_list: blocks b and b + 8 share an XCD, each XCD walks one contiguous eighth of the LIVE slots.
    // A launch may cover only the slot range [*rangeLo, *rangeHi) (device-side bounds, nullptr = open end): a z-slab engine
    // runs the slots next to its faces first, so that the halo exchange can start while the interior is still being computed.
    // dbg bit 8 (256): the launch covers BOTH ENDS instead, [0, *rangeLo) and [*rangeHi, live count), in one grid (one tail instead of two).
    const int boundAll = liveCount ? min(n, (int)*liveCount) : n;
    const bool ends = (dbg & 256) != 0;
    int first = (rangeLo && !ends) ? min((int)*rangeLo, boundAll) : 0;
    int bound = rangeHi ? min((int)*rangeHi, boundAll) : boundAll;
    int vb;
    if (!ends) {
        const int nBlocks = (max(bound - first, 0) + kB - 1) / kB, perXcd = (nBlocks + 7) >> 3;
        vb = ((int)blockIdx.x & 7) * perXcd + ((int)blockIdx.x >> 3);
        if (((int)blockIdx.x >> 3) >= perXcd || vb >= nBlocks) return;   // whole block, uniformly
    } else {
        const int endLo = min((int)*rangeLo, boundAll), startHi = max(bound, endLo);
        const int nLo = (endLo + kB - 1) / kB, nHi = (max(boundAll - startHi, 0) + kB - 1) / kB, perXcd = (nLo + nHi + 7) >> 3;
        vb = ((int)blockIdx.x & 7) * perXcd + ((int)blockIdx.x >> 3);
        if (((int)blockIdx.x >> 3) >= perXcd || vb >= nLo + nHi) return;
        if (vb < nLo) { first = 0; bound = endLo; } else { first = startHi; bound = boundAll; vb -= nLo; }
    }
    const int sRaw = first + vb * kB + tid;
    bool live = sRaw < bound;                              // every lane stays to the end (the staging is a wave-wide cooperation)
    const int s = live ? sRaw : max(bound - 1, 0);
    const float4 P = S.P(s), V = S.V(s), O = S.own[s];
    if (live && special_slot(k, S, in, out, order, s, P, V, O)) live = false;
    Own o;
    own_reset(o);
    o.px = P.x; o.py = P.y; o.pz = P.z; o.vx = V.x; o.vy = V.y; o.vz = V.z; o.rho = 0.0f; o.prs = 0.0f;
    const uint32_t cb = fbits(O.x);
    const int cx = (int)(cb & 1023u), cy = (int)((cb >> 10) & 1023u), cz = (int)(cb >> 20);
    const int xlo = max(cx - 1, 0), xhi = min(cx + 1, k.gx - 1);
    const __amdgpu_buffer_rsrc_t bufPV = __builtin_amdgcn_make_buffer_rsrc((void*)S.pv, 0, (int)((uint32_t)n * 32u), 0x00020000);
    // The list must hold every candidate within h of the ENTRY position (sweep 2) and of the position after this substep's
    // integration (sweep 3).  The latter is predicted as entry + m, m = 0.995 (v + g dt) dt; what the forces of this substep add
    // is covered by eps and checked after integrate().  (A search heuristic, not part of the arithmetic contract.)
    const float eps = SPH_WALK_EPS * k.h;
    const float hp = k.h + eps;
    const float qx = fmaf(0.995f * fmaf(k.gravx, k.dt, o.vx), k.dt, o.px), qy = fmaf(0.995f * fmaf(k.gravy, k.dt, o.vy), k.dt, o.py),
                qz = fmaf(0.995f * fmaf(k.gravz, k.dt, o.vz), k.dt, o.pz);      // free-flight prediction
    const float mvx = qx - o.px, mvy = qy - o.py, mvz = qz - o.pz;
    const float ex = mvx + mvx, ey = mvy + mvy, ez = mvz + mvz;
    // w = |d + m|^2 - hp^2 up to rounding (the slack left by the `near` test below is 1e3 times the rounding); hp^2 a little large
    const float mm = dot3(mvx, mvy, mvz, mvx, mvy, mvz);
    const float c0 = mm - (hp * hp) * 1.0001f;
    constexpr float kBig = 0x1p40f;                        // (h2 - r2 > 0) * 2^40 outweighs any w of a 27-cell candidate
    uint32_t qs[9], qe[9];                                 // all 18 run bounds first (independent loads in flight)
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
        const bool in = live && nz >= 0 && nz < k.gz && ny >= 0 && ny < k.gy;
        const int rowBase = in ? (nz * k.gy + ny) * k.gx : 0;
        const uint32_t a = cellStart[rowBase + xlo], b = cellStart[rowBase + xhi + 1];
        qs[r] = in ? a : 0u; qe[r] = in ? b : 0u;
    }
    // (the rounding of w grows with |m|^2: beyond 16 h of predicted move -- an uploaded or impulse-driven velocity far above the cap --
    //  the sign of s is no longer safe, and such a target takes the exact sweeps)
    bool listOk = !(dbg & 1) && !(mm > 256.0f * k.h2);
    // ---- sweep 1: density over every candidate (branch-free: a rejected candidate adds +0), and the list ----
    uint32_t cur = (uint32_t)tid * 2u;                     // byte offset of this thread's next list entry inside nl
    const uint32_t curEnd = (uint32_t)tid * 2u + (uint32_t)MAXN * kRowBytes;
    const uint32_t adv = live ? kRowBytes : 0u;            // lanes without a target never advance
    char* const nlBytes = reinterpret_cast<char*>(&nl[0][0]);
    uint32_t eSelf = 0xffffffffu;                          // entry value of the target itself (row 4)
    auto density = [&](const float4& J, float& r2, float& t) {
        const float dx = o.px - J.x, dy = o.py - J.y, dz = o.pz - J.z;
        r2 = dot3(dx, dy, dz, dx, dy, dz);
        t = SMALLH ? __builtin_amdgcn_fmed3f(k.h2 - r2, 0.0f, 1.0f) : fmaxf(k.h2 - r2, 0.0f);
        o.dsum = fmaf(t * t, t, o.dsum);
        return fmaf(ez, dz, fmaf(ey, dy, fmaf(ex, dx, r2 + c0)));
    };
    // one candidate with entry value e (a staged candidate carries it in J.w)
    auto visit = [&](const float4& J, uint32_t e, bool selfRow) {
        float r2, t;
        const float w = density(J, r2, t);
        float sg = fmaf(-kBig, t, w);                      // < 0 iff r2 < h2 or |d + m| < hp
        if (selfRow) sg = (e == eSelf) ? 1.0f : sg;        // the target itself: density only
        *reinterpret_cast<uint16_t*>(nlBytes + cur) = (uint16_t)e;
        cur += (fbits(sg) >> 22) & adv;
    };
    // The wave-uniform facts of a row and the window's loads are formed ONE ROW AHEAD (as in k_sph_list).
    unsigned long long mneN = 0ull;
    uint32_t aN = 0u, bN = 0u;
    bool stagedN = false;
    float4 pre0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pre1 = pre0, pre2 = pre0;
    const uint32_t laneOff32 = (uint32_t)lane * 32u;
    auto plan = [&](uint32_t q0, uint32_t q1) {
        const bool ne = q1 > q0;
        mneN = __ballot(ne);
        aN = bN = 0u; stagedN = false;
        if (mneN == 0ull) return;                          // nobody has a candidate in this row
        const int lf = __ffsll((long long)mneN) - 1, ll = 63 - __clzll((long long)mneN);
        const uint32_t A = (uint32_t)__builtin_amdgcn_readlane((int)q0, lf);   // lanes are consecutive sorted slots: run starts / ends ascend with the lane
        const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)q1, ll);
        aN = A; bN = B;
        stagedN = (B - A) <= (uint32_t)CAP && !(dbg & 4);  // wave-uniform
        if (stagedN) {                                     // the window's loads through a buffer resource over exactly [A, B): no per-lane clamp or 64-bit address, a load past B returns 0 (never read)
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(S.pv + 2u * (size_t)A), 0, (int)((B - A) * 32u), 0x00020000);
            pre0 = buf_load4(rw, laneOff32);
            if (CAP > 64) pre1 = buf_load4(rw, laneOff32 + 2048u);
            if (CAP > 128) pre2 = buf_load4(rw, laneOff32 + 4096u);
        }
    };
    int nRows = 0, nUnstaged = 0;
    plan(qs[0], qe[0]);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const uint32_t q0 = qs[r], q1 = qe[r];
        const bool ne = q1 > q0;
        const unsigned long long mne = mneN;
        const uint32_t A = aN, B = bN;
        const bool staged = stagedN;
        const bool selfRow = (r == 4);
        if (mne != 0ull && staged) {                       // this row's window into LDS (lanes past the union store duplicates)
            const uint32_t e0 = ((uint32_t)r << 12) | ((uint32_t)lane << 4);
            stage[wv][lane] = make_float4(pre0.x, pre0.y, pre0.z, bitsf(e0));
            if (CAP > 64 && (CAP >= 128 || lane < CAP - 64)) stage[wv][lane + 64] = make_float4(pre1.x, pre1.y, pre1.z, bitsf(e0 + (64u << 4)));
            if (CAP > 128 && lane < CAP - 128) stage[wv][lane + 128] = make_float4(pre2.x, pre2.y, pre2.z, bitsf(e0 + (128u << 4)));
        }
        if (r < 8) plan(qs[r + 1], qe[r + 1]);
        if (mne == 0ull) continue;
        nRows += 1; nUnstaged += staged ? 0 : 1;           // (wave-uniform; diagnostics only)
        const uint32_t len = q1 - q0;
        if (lane == 0) rowA[wv][r] = A * 32u;
        if (B - A > 255u) listOk = false;                  // offsets beyond the entry format (wave-uniform)
        const uint32_t off = ne ? q0 - A : 0u;
        if (selfRow) eSelf = (4u << 12) | ((((uint32_t)s - A) << 4) & 0xff0u);
        if (staged) {
            __builtin_amdgcn_wave_barrier();
            const float4* __restrict__ wp = &stage[wv][off];
            uint32_t m = 0;
            for (; m + UNROLL <= len; m += UNROLL) {       // full groups: no validity tests, immediate LDS offsets
                float4 J[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) J[u] = wp[m + (uint32_t)u];
                cur = min(cur, curEnd);                    // rows MAXN .. MAXN + UNROLL - 1 absorb the writes of a full list
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) visit(J[u], fbits(J[u].w), selfRow);
            }
            for (; m < len; ++m) {
                const float4 J = wp[m];
                cur = min(cur, curEnd);
                visit(J, fbits(J.w), selfRow);
            }
            __builtin_amdgcn_wave_barrier();
        } else {                                           // a window that does not fit: the same walk with per-lane loads from global memory
            uint32_t e = ((uint32_t)r << 12) | ((off << 4) & 0xff0u);
            uint32_t m = 0;
            for (; m + UNROLL <= len; m += UNROLL) {
                float4 J[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) J[u] = S.P(q0 + m + (uint32_t)u);
                cur = min(cur, curEnd);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) visit(J[u], e + 16u * (uint32_t)u, selfRow);
                e += 16u * UNROLL;
            }
            for (; m < len; ++m, e += 16u) {
                const float4 J = S.P(q0 + m);
                cur = min(cur, curEnd);
                visit(J, e, selfRow);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                       // rowA written by lane 0, read by every lane below
    listOk = (listOk && cur < curEnd) || !live;            // a cursor that reached the end may have dropped entries; lanes without a target never fall back
    finish_density(k, o);


    // ---- walks of sweeps 2 / 3: per-lane loops over the list; the 32-byte records of the next two entries are in flight ----
    const char* const rowBytes = reinterpret_cast<const char*>(&rowA[tid >> 6][0]);
    auto fetch = [&](uint32_t at, float4& J, float4& JV) {
        const uint32_t ent = *reinterpret_cast<const uint16_t*>(nlBytes + at);
        const uint32_t base = *reinterpret_cast<const uint32_t*>(rowBytes + ((ent >> 10) & 0x3cu));
        const uint32_t qb = base + ((ent & 0xff0u) << 1);
        J = buf_load4(bufPV, qb); JV = buf_load4(bufPV, qb + 16u);   // (a stale entry past the list gives any offset: bounds-checked, unused)
    };
    auto listed = [&](auto&& f) {
        // The pass is bound by the cache lines its gathers touch in L1 (profiles/r03_walk_mem_counters.log: ~0.9 tag lookups per
        // cycle and CU), so a look-ahead load is issued only for an entry that exists: no lane ever fetches past its list.
        const uint32_t end = cur;                          // <= curEnd - kRowBytes here
        uint32_t at = (uint32_t)tid * 2u;
        float4 J0, V0, J1, V1, J2, V2, J3, V3;
        J0 = V0 = J1 = V1 = J2 = V2 = J3 = V3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (at < end) fetch(at, J0, V0);
        if (at + kRowBytes < end) fetch(at + kRowBytes, J1, V1);
        while (at < end) {
            if (at + 2u * kRowBytes < end) fetch(at + 2u * kRowBytes, J2, V2);
            f(J0, V0);
            if (at + 3u * kRowBytes < end) fetch(at + 3u * kRowBytes, J3, V3);
            if (at + kRowBytes < end) f(J1, V1);
            at += 2u * kRowBytes;
            if (!(at < end)) break;
            if (at + 2u * kRowBytes < end) fetch(at + 2u * kRowBytes, J0, V0);
            f(J2, V2);
            if (at + 3u * kRowBytes < end) fetch(at + 3u * kRowBytes, J1, V1);
            if (at + kRowBytes < end) f(J3, V3);
            at += 2u * kRowBytes;
        }
    };
    static_assert(kSpare >= 2, "the walks read up to two rows past the last entry");
    // Exact fallback of a sweep for lanes whose list cannot be used: every candidate again, in canonical order, from global memory.
    auto plain = [&](auto&& f) {
        for (int r = 0; r < 9; ++r) {
            const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
            if (nz < 0 || nz >= k.gz || ny < 0 || ny >= k.gy) continue;
            const int rowBase = (nz * k.gy + ny) * k.gx;
            const uint32_t a = cellStart[rowBase + xlo], b = cellStart[rowBase + xhi + 1];
            // (a candidate outside h of every lane that is here adds +-0 everywhere: the wave skips its pair arithmetic)
            auto within = [&](const float4& J) { const float dx = o.px - J.x, dy = o.py - J.y, dz = o.pz - J.z; return dot3(dx, dy, dz, dx, dy, dz) < k.h2; };
            uint32_t q = a;
            for (; q + 2u <= b; q += 2u) {                    // two candidates' loads in flight (in compressed fluid they are broadcasts: the wave's targets share their candidates)
                const float4 J0 = S.P(q), V0 = S.V(q), J1 = S.P(q + 1u), V1 = S.V(q + 1u);
                if (__any(within(J0))) f(J0, V0, (int32_t)((int)q != s ? -1 : 0));
                if (__any(within(J1))) f(J1, V1, (int32_t)((int)(q + 1u) != s ? -1 : 0));
            }
            if (q < b) { const float4 J = S.P(q); if (__any(within(J))) f(J, S.V(q), (int32_t)((int)q != s ? -1 : 0)); }
        }
    };
    auto force_at = [&](const float4& J, const float4& JV) { pair_force_other(k, o, J, JV); };
    auto xsph_at = [&](const float4& J, const float4& JV) { pair_xsph_other<SMALLH>(k, o, J, JV); };
    auto force_plain = [&](const float4& J, const float4& JV, int32_t ok) { pair_force(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, JV.w, J.w, ok); };
    auto xsph_plain = [&](const float4& J, const float4& JV, int32_t ok) { pair_xsph(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, J.w, ok); };
    // ---- sweep 2 ----
    if (listOk) listed(force_at); else plain(force_plain);
    integrate(k, o);
    // ---- sweep 3: the list stays a superset only while the displacement is inside its slack ----
    const float mx = o.px - qx, my = o.py - qy, mz = o.pz - qz;
    const float lim = 0.98f * eps;
    const bool near = (dot3(mx, my, mz, mx, my, mz) <= lim * lim && !(dbg & 2)) || !live;
    if (listOk && near) listed(xsph_at); else plain(xsph_plain);
    const float foamOut = finish_particle(k, o, O.y);
    if (live) store_fields(k, out, s, fbits(O.z), fbits(O.w), o.px, o.py, o.pz, o.vx, o.vy, o.vz, o.ax, o.ay, o.az, o.rho, o.prs, foamOut, cz);
    if (dbg & 8) {   // diagnostics: [0] candidate rows walked from global memory (window too large), [3] candidate rows, [1] targets on an exact fallback sweep, [2] list entries, [4] lanes, [5] overflowed lists, [6] far targets, [7] waves with a fallback
        const unsigned long long slowT = (unsigned long long)__popcll(__ballot(live && !(listOk && near)));
        unsigned long long ents = (unsigned long long)((live && listOk) ? (cur - (uint32_t)tid * 2u) / kRowBytes : 0u);
        for (int d = 32; d >= 1; d >>= 1) ents += (unsigned long long)__shfl_xor((int)ents, d, 64);
        const unsigned long long ovf = (unsigned long long)__popcll(__ballot(live && !listOk)), far = (unsigned long long)__popcll(__ballot(live && listOk && !near));
        if (lane == 0) { atomicAdd(&stats[0], (unsigned long long)nUnstaged); atomicAdd(&stats[3], (unsigned long long)nRows); atomicAdd(&stats[1], slowT); atomicAdd(&stats[2], ents & 0xffffffffull); atomicAdd(&stats[4], 64ull); atomicAdd(&stats[5], ovf); atomicAdd(&stats[6], far); atomicAdd(&stats[7], slowT ? 1ull : 0ull); }
    }
}

}  // namespace sph
