// sph_pass.h -- the SPH pass (SPHFluid.comp:66-221 + fused OBBConstraints.comp) of the engine.
//
//   k_sph_list  (default)  one target per lane over the physically sorted copy: wave-cooperative LDS windows for the
//               candidate scan (density + neighbour list), sweeps 2 / 3 over the listed neighbours only.
//   k_sph_slow  one target per thread, three full candidate sweeps straight from global memory: the plain statement
//               of the same arithmetic (A/B variant SPH_OPT_NEIGHBOR_KERNEL = 1).
// Same functions (sph_device.h), same candidate order, therefore the same bits.
#pragma once
#include <type_traits>

#include "sph_kernels.h"

namespace sph {

struct SortedIn {
    // One 32-byte record per particle in (cell, id) order of THIS substep: (x, y, z, 1/rho or 0), (vx, vy, vz, P).
    // Interleaved since round 3: a neighbour's two halves sit in one cache line, so the walks of sweeps 2 / 3 pull one
    // line per neighbour instead of two (the walks are bound by the lines their gathers move into L1).
    const float4* __restrict__ pv;
    const float4* __restrict__ own;    // (bits(cx | cy << 10 | cz << 20), foam, bits(flags), bits(id))
    __device__ __forceinline__ float4 P(uint32_t q) const { return pv[2u * q]; }          // (x, y, z, 1/rho)
    __device__ __forceinline__ float4 V(uint32_t q) const { return pv[2u * q + 1u]; }     // (vx, vy, vz, P)
};   // written by k_rank<true>

__device__ __forceinline__ void store_fields(const SimK& k, const StateOut& out, int s, uint32_t flags, uint32_t id, float px, float py,
                                             float pz, float vx, float vy, float vz, float ax, float ay, float az, float rho, float prs,
                                             float foamOut, int czEntry) {
    if (!(flags & F_GHOSTNZ)) obb_apply(k, px, py, pz, vx, vy, vz);   // OBBConstraints.comp:46
    slab_check_layer_move(k, czEntry, pz);
    out.pos[s] = make_float4(px, py, pz, bitsf(flags));
    out.vel[s] = make_float4(vx, vy, vz, bitsf(id));
    out.rp[s] = make_float2(rho, prs);
    out.foam[s] = foamOut;
    if (out.aos) aos_write_fluid(out.aos, id - out.idBase, px, py, pz, vx, vy, vz, ax, ay, az, rho, prs, foamOut);
    else out.acc[s] = make_float4(ax, ay, az, 0.0f);
}

// Halo copy / ghost branch of SPHFluid.comp:72-83 for sorted slot s.  Returns true when the slot is done.
__device__ __forceinline__ bool special_slot(const SimK& k, const SortedIn& S, const StateIn& in, const StateOut& out,
                                             const uint32_t* __restrict__ order, int s, const float4& P, const float4& V, const float4& O) {
    const uint32_t flags = fbits(O.z), id = fbits(O.w);
    if (flags & F_HALO) { out.pos[s] = make_float4(P.x, P.y, P.z, O.z); return true; }   // neighbour rank's particle: candidate only
    if (flags & F_GHOST1) {
        float gvx = V.x, gvy = V.y, gvz = V.z, grho = in.rp[order[s]].x, gprs = V.w;
        if (!(flags & F_INACTIVE)) { gvx = gvy = gvz = 0.0f; grho = k.rho0; gprs = 0.0f; }
        out.pos[s] = make_float4(P.x, P.y, P.z, O.z);
        out.vel[s] = make_float4(gvx, gvy, gvz, O.w);
        out.rp[s] = make_float2(grho, gprs);
        out.foam[s] = O.y;
        if (out.aos) { if (!(flags & F_INACTIVE)) aos_write_active_ghost(out.aos, id - out.idBase, k.rho0); }
        else out.acc[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return true;
    }
    return false;
}

// One target, everything from global memory, candidates in canonical order.
__device__ __forceinline__ void sph_slow_one(const SimK& k, const SortedIn& S, const StateIn& in, const StateOut& out,
                                          const uint32_t* __restrict__ order, const uint32_t* __restrict__ cellStart, int s) {
    const float4 P = S.P(s), V = S.V(s), O = S.own[s];
    if (special_slot(k, S, in, out, order, s, P, V, O)) return;
    const uint32_t cb = fbits(O.x);
    const int cx = (int)(cb & 1023u), cy = (int)((cb >> 10) & 1023u), cz = (int)(cb >> 20);
    Own o;
    own_reset(o);
    o.px = P.x; o.py = P.y; o.pz = P.z; o.vx = V.x; o.vy = V.y; o.vz = V.z; o.rho = 0.0f; o.prs = 0.0f;
    const int xlo = max(cx - 1, 0), xhi = min(cx + 1, k.gx - 1);
    auto rows = [&](auto&& f) {
        for (int r = 0; r < 9; ++r) {
            const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
            if (nz < 0 || nz >= k.gz || ny < 0 || ny >= k.gy) continue;
            const int rowBase = (nz * k.gy + ny) * k.gx;
            const uint32_t qs = cellStart[rowBase + xlo], qe = cellStart[rowBase + xhi + 1];
            for (uint32_t q = qs; q < qe; ++q) f(q);
        }
    };
    rows([&](uint32_t q) { const float4 J = S.P(q); pair_density(k, o, J.x, J.y, J.z, (int32_t)-1); });
    finish_density(k, o);
    rows([&](uint32_t q) {
        const float4 J = S.P(q), JV = S.V(q);
        pair_force(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, JV.w, J.w, (int32_t)((int)q != s ? -1 : 0));
    });
    integrate(k, o);
    rows([&](uint32_t q) {
        const float4 J = S.P(q), JV = S.V(q);
        pair_xsph(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, J.w, (int32_t)((int)q != s ? -1 : 0));
    });
    const float foamOut = finish_particle(k, o, O.y);
    store_fields(k, out, s, fbits(O.z), fbits(O.w), o.px, o.py, o.pz, o.vx, o.vy, o.vz, o.ax, o.ay, o.az, o.rho, o.prs, foamOut, cz);
}

__global__ __launch_bounds__(kBlock) void k_sph_slow(SimK k, SortedIn S, StateIn in, StateOut out, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ cellStart, int n) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n || (uint32_t)s >= cellStart[k.numCells]) return;
    sph_slow_one(k, S, in, out, order, cellStart, s);
}

// tuning knobs of k_sph_list (values measured best on MI355X)
#ifndef SPH_LIST_MAXN
#define SPH_LIST_MAXN 47     // list entries per target
#endif
#ifndef SPH_LIST_CAP
#define SPH_LIST_CAP 96      // wave-private LDS window of one candidate row
#endif
#ifndef SPH_LIST_UNROLL
#define SPH_LIST_UNROLL 3    // candidates per iteration of sweep 1
#endif
#ifndef SPH_LIST_WAVES
#define SPH_LIST_WAVES 5     // __launch_bounds__ minimum waves per SIMD (5: at most 96 VGPRs, 20 waves per CU with 32 KB of LDS per block)
#endif
#ifndef SPH_LIST_BLOCK
#define SPH_LIST_BLOCK 256   // threads per block of k_sph_list
#endif
#ifndef SPH_LIST_EPS
#define SPH_LIST_EPS 0.06f  // slack of the list around the predicted position, in units of h
#endif
#ifndef SPH_LIST_LISTU
#define SPH_LIST_LISTU 4     // list entries fetched together in sweeps 2 / 3
#endif
#ifndef SPH_LIST_CHUNKG
#define SPH_LIST_CHUNKG 4    // candidates filtered per step of the chunked fallback
#endif

// k_sph_list: one target per lane over the sorted copy.  Sweep 1 walks the 9 contiguous (dy,dz) candidate rows once
// (density + a list of everything within h of the entry position or within h + eps of the predicted new position);
// the 64 targets of a wave are consecutive sorted slots, so their ranges of one row overlap heavily and the wave
// stages their union in a wave-private LDS window with coalesced loads.  Sweeps 2 / 3 touch only the listed
// neighbours.  Lists and windows are private to the thread / the wave: no __syncthreads.
template <int MAXN, int UNROLL, int CAP>
__global__ __launch_bounds__(SPH_LIST_BLOCK, SPH_LIST_WAVES) void k_sph_list(SimK k, SortedIn S, StateIn in, StateOut out, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ cellStart, const uint32_t* __restrict__ liveCount, int n, int dbg,
                                                     unsigned long long* __restrict__ stats) {
    constexpr int kB = SPH_LIST_BLOCK;       // wave-private LDS only: the block size is just a dispatch / LDS granule
    __shared__ uint16_t nl[MAXN + UNROLL][kB];   // entry e of thread t: (row << 12) | (slot - rowA[row]); rows >= MAXN absorb the writes of a full list
    __shared__ uint32_t rowA[kB / 64][12];       // per wave: first slot of the wave's union of each of the 9 candidate rows
    __shared__ float4 stage[kB / 64][CAP];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    // XCD-aware block mapping: blocks b and b+8 run on the same XCD (round-robin dispatch); each XCD gets one contiguous
    // eighth of the sorted order, so that its L2 holds only that part of the sorted copy.  Any mapping is correct.
    const int bound = liveCount ? min(n, (int)*liveCount) : n;
    // (over the LIVE slots: a z-slab launch covers the slab's slot capacity, and an eighth of the capacity per XCD would
    // leave the last XCDs without work)
    const int nBlocks = (bound + kB - 1) / kB, perXcd = (nBlocks + 7) >> 3;
    const int vb = ((int)blockIdx.x & 7) * perXcd + ((int)blockIdx.x >> 3);
    // (a z-slab launch covers the slot CAPACITY: blocks past the 8 x perXcd that map one-to-one onto the live slots leave,
    // otherwise they would compute some virtual block a second time)
    if (((int)blockIdx.x >> 3) >= perXcd || vb >= nBlocks) return;   // whole block, uniformly
    const int sRaw = vb * kB + tid;
    bool live = sRaw < bound;                                // every lane stays to the end (the staging is a wave-wide cooperation)
    const int s = live ? sRaw : max(bound - 1, 0);
    const float4 P = S.P(s), V = S.V(s), O = S.own[s];
    if (live && special_slot(k, S, in, out, order, s, P, V, O)) live = false;
    Own o;
    own_reset(o);
    o.px = P.x; o.py = P.y; o.pz = P.z; o.vx = V.x; o.vy = V.y; o.vz = V.z; o.rho = 0.0f; o.prs = 0.0f;
    const uint32_t cb = fbits(O.x);
    const int cx = (int)(cb & 1023u), cy = (int)((cb >> 10) & 1023u), cz = (int)(cb >> 20);
    const int xlo = max(cx - 1, 0), xhi = min(cx + 1, k.gx - 1);
    // The list must hold every candidate within h of the ENTRY position (sweep 2) and of the position after this
    // substep's integration (sweep 3).  The latter is predicted as entry + 0.995 v dt; what the forces of this substep
    // add to it is covered by eps and checked after integrate().  (A search heuristic, not part of the arithmetic contract.)
    const float eps = SPH_LIST_EPS * k.h;
    const float hp = k.h + eps;
    const float hp2 = hp * hp;
    const float qx = fmaf(0.995f * fmaf(k.gravx, k.dt, o.vx), k.dt, o.px), qy = fmaf(0.995f * fmaf(k.gravy, k.dt, o.vy), k.dt, o.py),
                qz = fmaf(0.995f * fmaf(k.gravz, k.dt, o.vz), k.dt, o.pz);      // free-flight prediction
    uint32_t qs[9], qe[9];                                   // all 18 run bounds first (independent loads in flight)
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
        const bool in = live && nz >= 0 && nz < k.gz && ny >= 0 && ny < k.gy;
        const int rowBase = in ? (nz * k.gy + ny) * k.gx : 0;
        const uint32_t a = cellStart[rowBase + xlo], b = cellStart[rowBase + xhi + 1];
        qs[r] = in ? a : 0u; qe[r] = in ? b : 0u;
    }
    int cnt = 0;
    bool listOk = !(dbg & 1);
    // ---- sweep 1: density over every candidate (branch-free: a rejected candidate adds +0), and the list ----
    // The two distances (to the entry and to the predicted position) are one packed-fp32 chain (same rounding as dot3).
    // The list is written unconditionally at the running count (a candidate that fails leaves the count alone).
    const v2f PX = {o.px, qx}, PY = {o.py, qy}, PZ = {o.pz, qz};
    // one candidate: density term and the two-ball list test; returns 1 when the candidate belongs on the list
    auto visit = [&](const float4& J) -> int {
        const v2f dx = PX - J.x, dyy = PY - J.y, dzz = PZ - J.z;
        const v2f d2 = __builtin_elementwise_fma(dzz, dzz, __builtin_elementwise_fma(dyy, dyy, dx * dx));   // (r2, p2)
        const float tt = fmaxf(k.h2 - d2.x, 0.0f);
        o.dsum = fmaf(tt * tt, tt, o.dsum);
        return (d2.x < k.h2 || d2.y < hp2) ? 1 : 0;
    };
    // candidates [m, m + UNROLL) of run r, all valid: the list rows are taken from one clamped running count
    // (rows MAXN .. MAXN + UNROLL - 1 absorb the writes of a full list; cnt keeps counting for the overflow test)
    auto group = [&](const float4 (&J)[UNROLL], uint32_t ebase) {   // ebase = (row << 12) | (slot of J[0] - rowA[row])
        int row = min(cnt, MAXN);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int p = visit(J[u]);
            nl[row][tid] = (uint16_t)(ebase + (uint32_t)u);
            row += p; cnt += p;
        }
    };
    auto single = [&](const float4& J, uint32_t e) {
        const int p = visit(J);
        nl[min(cnt, MAXN)][tid] = (uint16_t)e;
        cnt += p;
    };
    // The wave-uniform facts of a row (who has candidates, the union [A, B) of the lanes' runs, whether it fits the LDS
    // window) and the window's loads are formed ONE ROW AHEAD: the loads of row r + 1 are in flight while row r is walked
    // out of LDS (the staged walk issues no global loads, so nothing waits on them before the next row's LDS store).
    static_assert(CAP <= 128, "the window is staged with two loads per lane");
    unsigned long long mneN = 0ull;
    uint32_t aN = 0u, bN = 0u;
    bool stagedN = false;
    float4 pre0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pre1 = pre0;
    auto plan = [&](uint32_t q0, uint32_t q1) {
        // union of the wave's ranges of this row: the lanes are consecutive sorted slots, so their cells ascend with the
        // lane; the neighbour cell (x - 1 clamped, y + dy, z + dz) and with it cellStart[...] ascends too (the clamp at
        // x = 0 keeps the order, rows outside the grid are empty and excluded).  The union therefore runs from the first
        // non-empty lane's start to the last one's end: two readlanes.
        const bool ne = q1 > q0;
        mneN = __ballot(ne);
        aN = bN = 0u; stagedN = false;
        if (mneN == 0ull) return;                          // nobody has a candidate in this row
        const int lf = __ffsll((long long)mneN) - 1, ll = 63 - __clzll((long long)mneN);
        const uint32_t A = (uint32_t)__builtin_amdgcn_readlane((int)q0, lf);
        const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)q1, ll);
        aN = A; bN = B;
        stagedN = (B - A) <= (uint32_t)CAP && !(dbg & 4);  // wave-uniform
        if (stagedN) {                                     // clamped, unconditional: B - A >= 1 here
            pre0 = S.P(A + min((uint32_t)lane, B - A - 1u));
            if (CAP > 64) pre1 = S.P(A + min((uint32_t)lane + 64u, B - A - 1u));
        }
    };
    plan(qs[0], qe[0]);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const uint32_t q0 = qs[r], q1 = qe[r];
        const bool ne = q1 > q0;
        const unsigned long long mne = mneN;
        const uint32_t A = aN, B = bN;
        const bool staged = stagedN;
        if (mne != 0ull && staged) {                       // this row's window into LDS (lanes past the union store duplicates)
            stage[wv][lane] = pre0;
            if (CAP > 64 && lane < CAP - 64) stage[wv][lane + 64] = pre1;
        }
        if (r < 8) plan(qs[r + 1], qe[r + 1]);
        if (mne == 0ull) continue;
        if (lane == 0) rowA[wv][r] = A;
        if (B - A > 4095u) listOk = false;                 // offsets beyond the entry format (wave-uniform)
        const uint32_t len = q1 - q0;
        const uint32_t ebase = ((uint32_t)r << 12) | ((ne ? q0 - A : 0u) & 0xfffu);
        uint32_t m = 0;
        if (staged) {
            __builtin_amdgcn_wave_barrier();
            const float4* __restrict__ wp = &stage[wv][ne ? q0 - A : 0u];
            for (; m + UNROLL <= len; m += UNROLL) {       // full groups: no validity tests, immediate LDS offsets
                float4 J[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) J[u] = wp[m + (uint32_t)u];
                group(J, ebase + m);
            }
            for (; m < len; ++m) single(wp[m], ebase + m);
            __builtin_amdgcn_wave_barrier();
        } else {
            for (; m + UNROLL <= len; m += UNROLL) {
                float4 J[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) J[u] = S.P(q0 + m + (uint32_t)u);
                group(J, ebase + m);
            }
            for (; m < len; ++m) single(S.P(q0 + m), ebase + m);
        }
    }
    __builtin_amdgcn_wave_barrier();                       // rowA written by lane 0, read by every lane below
    listOk = (listOk && cnt <= MAXN) || !live;          // lanes without a target never take a fallback
    finish_density(k, o);

    // walk the first `count` list entries, SPH_LIST_LISTU loads in flight
    auto listed = [&](int count, auto&& f) {
        for (int e = 0; e < count; e += SPH_LIST_LISTU) {
            float4 J[SPH_LIST_LISTU], JV[SPH_LIST_LISTU];
            int32_t ok[SPH_LIST_LISTU];
#pragma unroll
            for (int u = 0; u < SPH_LIST_LISTU; ++u) {
                const uint32_t a = nl[min(e + u, count - 1)][tid];
                const uint32_t q = rowA[wv][a >> 12] + (a & 0xfffu);
                J[u] = S.P(q); JV[u] = S.V(q);
                ok[u] = (e + u < count && (int)q != s) ? -1 : 0;
            }
#pragma unroll
            for (int u = 0; u < SPH_LIST_LISTU; ++u) f(J[u], JV[u], ok[u]);
        }
    };
    // Exact fallback of a sweep (a list that did not fit: compressed fluid; sweep 3 after a long move): every
    // candidate again, in canonical order, in CHUNKS -- candidates within h of (cpx, cpy, cpz) are collected into the
    // list until some lane's list is full, then the wave walks what it has and starts over.  The accumulators see
    // the same operations in the same order as one uninterrupted sweep (a candidate outside h adds +0 either way).
    auto chunked = [&](float cpx, float cpy, float cpz, auto&& f) {
        int c = 0;
        for (int r = 0; r < 9; ++r) {
            const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
            if (nz < 0 || nz >= k.gz || ny < 0 || ny >= k.gy) continue;
            const int rowBase = (nz * k.gy + ny) * k.gx;
            const uint32_t a = cellStart[rowBase + xlo], b = cellStart[rowBase + xhi + 1];
            const uint32_t base = rowA[wv][r];                 // <= a: the wave's union of this row starts at or before this lane's run
            if (b - base > 4095u || base > a) {                // offsets beyond the entry format: plain sweep of this run
                listed(c, f); c = 0;
                for (uint32_t q = a; q < b; ++q) f(S.P(q), S.V(q), (int32_t)((int)q != s ? -1 : 0));
                continue;
            }
            constexpr int G = SPH_LIST_CHUNKG;                 // candidates per step: G loads in flight, one fullness test
            static_assert(UNROLL + MAXN - G >= 8, "chunk rows");
            for (uint32_t q = a; q < b; q += G) {
                float4 J[G];
#pragma unroll
                for (int u = 0; u < G; ++u) J[u] = S.P(min(q + (uint32_t)u, b - 1u));
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const float dx = cpx - J[u].x, dy = cpy - J[u].y, dz = cpz - J[u].z;
                    nl[c][tid] = (uint16_t)((r << 12) | (int)(q + (uint32_t)u - base));
                    c += (q + (uint32_t)u < b && dot3(dx, dy, dz, dx, dy, dz) < k.h2) ? 1 : 0;
                }
                if (__any(c > MAXN + UNROLL - 1 - G)) { listed(c, f); c = 0; }   // the next step may add G more rows
            }
        }
        listed(c, f);
    };
    auto force_at = [&](const float4& J, const float4& JV, int32_t ok) { pair_force(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, JV.w, J.w, ok); };
    auto xsph_at = [&](const float4& J, const float4& JV, int32_t ok) { pair_xsph(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, J.w, ok); };
#if defined(SPH_LIST_CUT) && SPH_LIST_CUT == 1   // timing experiment only (tools/ab_variants.sh): stop after sweep 1
    if (live) store_fields(k, out, s, fbits(O.z), fbits(O.w), o.px, o.py, o.pz, o.vx, o.vy, o.vz, (float)cnt, o.ay, o.az, o.rho, o.prs, O.y, cz);
    return;
#endif
    // ---- sweep 2 ----
    if (listOk) listed(cnt, force_at); else chunked(o.px, o.py, o.pz, force_at);
    integrate(k, o);
#if defined(SPH_LIST_CUT) && SPH_LIST_CUT == 2   // timing experiment only: stop after sweep 2
    if (live) store_fields(k, out, s, fbits(O.z), fbits(O.w), o.px, o.py, o.pz, o.vx, o.vy, o.vz, o.ax, o.ay, o.az, o.rho, o.prs, O.y, cz);
    return;
#endif
    // ---- sweep 3: the list stays a superset only while the displacement is inside its slack ----
    const float mx = o.px - qx, my = o.py - qy, mz = o.pz - qz;
    const float lim = 0.98f * eps;
    const bool near = (dot3(mx, my, mz, mx, my, mz) <= lim * lim && !(dbg & 2)) || !live;
    if (listOk && near) listed(cnt, xsph_at); else chunked(o.px, o.py, o.pz, xsph_at);
    const float foamOut = finish_particle(k, o, O.y);
    if (live) store_fields(k, out, s, fbits(O.z), fbits(O.w), o.px, o.py, o.pz, o.vx, o.vy, o.vz, o.ax, o.ay, o.az, o.rho, o.prs, foamOut, cz);
    if (dbg & 8) {   // diagnostics: [1] targets on an exact fallback sweep, [2] list entries, [4] lanes
        const unsigned long long slowT = (unsigned long long)__popcll(__ballot(live && !(listOk && near)));
        unsigned long long ents = (unsigned long long)(live ? cnt : 0);
        for (int d = 32; d >= 1; d >>= 1) ents += (unsigned long long)__shfl_xor((int)ents, d, 64);
        const unsigned long long ovf = (unsigned long long)__popcll(__ballot(live && !listOk)), far = (unsigned long long)__popcll(__ballot(live && listOk && !near));
        if (lane == 0) { atomicAdd(&stats[1], slowT); atomicAdd(&stats[2], ents & 0xffffffffull); atomicAdd(&stats[4], 64ull); atomicAdd(&stats[5], ovf); atomicAdd(&stats[6], far); atomicAdd(&stats[7], slowT ? 1ull : 0ull); }
    }
}

}  // namespace sph
