// sph_pair.h -- the SPH pass (SPHFluid.comp:66-221 + fused OBBConstraints.comp) of the engine.
//
//   k_sph_pair  (default)  two ADJACENT sorted targets per lane, 128 targets per wave.  The two targets of a
//               lane share one candidate stream: every candidate is loaded once and tested against both
//               targets with one packed-fp32 instruction per operation (the candidate is broadcast into both
//               halves through op_sel, the targets sit in the two halves).
//               LDS-staged stencil: the wave copies the 9 (dy,dz) candidate rows of its 128 targets -- each one
//               contiguous range of the sorted order -- into wave-private LDS windows (x, y, z, 1/rho, slot)
//               with coalesced loads ONCE; the candidate scan, the density sweep and the position part of the
//               force / XSPH sweeps then run out of LDS.  Sweep 1 only builds a neighbour list per lane
//               (16-bit window indices, canonical order) of everything inside a ball that contains both the
//               h-ball of the entry position and the h-ball of the predicted new position; density, forces and
//               XSPH walk that list.  No __syncthreads: windows and lists are private to the wave.
//   k_sph_slow  one target per thread, three full candidate sweeps straight from global memory: the plain
//               statement of the same arithmetic (A/B variant SPH_OPT_NEIGHBOR_KERNEL = 1) and the exact
//               fallback of k_sph_pair (window or list overflow, a target that moved further than the list's
//               slack).  Same functions, same candidate order, therefore the same bits.
#pragma once
#include <type_traits>

#include "sph_kernels.h"

namespace sph {

#ifndef SPH_PAIR_LN
#define SPH_PAIR_LN 32        // list entries per lane (two targets share a list)
#endif
#ifndef SPH_PAIR_WCAP
#define SPH_PAIR_WCAP 1792    // window capacity per wave, in candidates (all 9 rows of the wave's 128 targets)
#endif

struct SortedIn {
    const float4* __restrict__ posI;   // (x, y, z, 1/rho or 0)  in (cell, id) order of THIS substep
    const float4* __restrict__ velP;   // (vx, vy, vz, P)
    const float4* __restrict__ own;    // (bits(cx | cy << 10 | cz << 20), foam, bits(flags), bits(id))
};   // written by k_rank<true>

__device__ __forceinline__ void store_fields(const SimK& k, const StateOut& out, int s, uint32_t flags, uint32_t id, float px, float py,
                                             float pz, float vx, float vy, float vz, float ax, float ay, float az, float rho, float prs,
                                             float foamOut) {
    if (!(flags & F_GHOSTNZ)) obb_apply(k, px, py, pz, vx, vy, vz);   // OBBConstraints.comp:46
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
    const float4 P = S.posI[s], V = S.velP[s], O = S.own[s];
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
    rows([&](uint32_t q) { const float4 J = S.posI[q]; pair_density(k, o, J.x, J.y, J.z, (int32_t)-1); });
    finish_density(k, o);
    rows([&](uint32_t q) {
        const float4 J = S.posI[q], JV = S.velP[q];
        pair_force(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, JV.w, J.w, (int32_t)((int)q != s ? -1 : 0));
    });
    integrate(k, o);
    rows([&](uint32_t q) {
        const float4 J = S.posI[q], JV = S.velP[q];
        pair_xsph(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, J.w, (int32_t)((int)q != s ? -1 : 0));
    });
    const float foamOut = finish_particle(k, o, O.y);
    store_fields(k, out, s, fbits(O.z), fbits(O.w), o.px, o.py, o.pz, o.vx, o.vy, o.vz, o.ax, o.ay, o.az, o.rho, o.prs, foamOut);
}

__global__ __launch_bounds__(kBlock) void k_sph_slow(SimK k, SortedIn S, StateIn in, StateOut out, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ cellStart, int n) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n || (uint32_t)s >= cellStart[k.numCells]) return;
    sph_slow_one(k, S, in, out, order, cellStart, s);
}

__device__ __forceinline__ v2f bc(float x) { v2f r = {x, x}; return r; }

template <int LN, int WCAP>
__global__ __launch_bounds__(64) void k_sph_pair(SimK k, SortedIn S, StateIn in, StateOut out, const uint32_t* __restrict__ order,
                                                 const uint32_t* __restrict__ cellStart, const uint32_t* __restrict__ liveCount, int n, int dbg,
                                                 unsigned long long* __restrict__ stats) {
    static_assert(WCAP <= 16384, "list entries keep the window index in 14 bits");
    __shared__ float wx[WCAP], wy[WCAP], wz[WCAP], wi[WCAP];   // window: position and 1/rho of every staged candidate
    __shared__ uint32_t ws[WCAP];                              // its sorted slot (address of vel / P)
    __shared__ uint16_t nl[LN + 1][64];                        // entry e of lane l (row LN absorbs the writes of a full list): window index | in-stencil bits (14: target 0, 15: target 1)
    const int lane = threadIdx.x;
    // XCD-aware unit mapping: blocks b and b+8 run on the same XCD (round-robin dispatch); each XCD gets one
    // contiguous eighth of the sorted order so that its L2 holds that part of the sorted copy.  Any mapping is correct.
    const int nUnits = (n + 127) >> 7, perXcd = (nUnits + 7) >> 3;
    const int vb = ((int)blockIdx.x & 7) * perXcd + ((int)blockIdx.x >> 3);
    if (vb >= nUnits) return;
    const int bound = liveCount ? min(n, (int)*liveCount) : n;
    const int s0 = vb * 128 + 2 * lane, s1 = s0 + 1;
    bool live0 = s0 < bound, live1 = s1 < bound;
    const int c0 = live0 ? s0 : max(bound - 1, 0), c1 = live1 ? s1 : max(bound - 1, 0);
    const float4 P0 = S.posI[c0], P1 = S.posI[c1], V0 = S.velP[c0], V1 = S.velP[c1], O0 = S.own[c0], O1 = S.own[c1];
    if (live0 && special_slot(k, S, in, out, order, s0, P0, V0, O0)) live0 = false;
    if (live1 && special_slot(k, S, in, out, order, s1, P1, V1, O1)) live1 = false;

    OwnT<v2f> o;
    own_reset(o);
    { const v2f a = {P0.x, P1.x}, b = {P0.y, P1.y}, c = {P0.z, P1.z}; o.px = a; o.py = b; o.pz = c; }
    { const v2f a = {V0.x, V1.x}, b = {V0.y, V1.y}, c = {V0.z, V1.z}; o.vx = a; o.vy = b; o.vz = c; }
    o.rho = bc(0.0f); o.prs = bc(0.0f);
    const uint32_t cb0 = fbits(O0.x), cb1 = fbits(O1.x);
    const int cx0 = (int)(cb0 & 1023u), cy0 = (int)((cb0 >> 10) & 1023u), cz0 = (int)(cb0 >> 20);
    const int cx1 = (int)(cb1 & 1023u), cy1 = (int)((cb1 >> 10) & 1023u), cz1 = (int)(cb1 >> 20);
    const int xlo0 = max(cx0 - 1, 0), xhi0 = min(cx0 + 1, k.gx - 1), xlo1 = max(cx1 - 1, 0), xhi1 = min(cx1 + 1, k.gx - 1);

    // List ball (a search heuristic, not part of the arithmetic contract): centre = entry position + half the
    // predicted displacement d = 0.995 (v + g dt) dt, radius h + |d|/2 + eps.  It contains the h-ball of the
    // entry position (density, forces) and the h-ball of every position within eps of entry + d (XSPH); the
    // actual displacement is checked after integrate().
    const float eps = 0.08f * k.h;
    const v2f dt2 = bc(k.dt);
    const v2f ddx = (0.995f * t_fma(bc(k.gravx), dt2, o.vx)) * k.dt, ddy = (0.995f * t_fma(bc(k.gravy), dt2, o.vy)) * k.dt,
              ddz = (0.995f * t_fma(bc(k.gravz), dt2, o.vz)) * k.dt;
    const v2f qx = o.px + ddx, qy = o.py + ddy, qz = o.pz + ddz;                 // predicted new position
    const v2f mx = t_fma(bc(0.5f), ddx, o.px), my = t_fma(bc(0.5f), ddy, o.py), mz = t_fma(bc(0.5f), ddz, o.pz);
    const v2f dd2 = t_dot3(ddx, ddy, ddz, ddx, ddy, ddz);
    const v2f dlen = dd2 * t_rsqrt(t_max(dd2, bc(SPH_TINY)));
    const v2f Rl = t_fma(bc(0.5005f), dlen, bc(k.h + eps));
    const v2f R2 = Rl * Rl;

    // ---- run bounds of all 9 rows first (36 independent loads in flight), then the wave's windows ----
    uint32_t qa0[9], qb0[9], qa1[9], qb1[9];           // [row] begin / end of target 0's and target 1's run
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int dz = r / 3 - 1, dy = r % 3 - 1;
        {
            const int nz = cz0 + dz, ny = cy0 + dy;
            const bool in0 = live0 && nz >= 0 && nz < k.gz && ny >= 0 && ny < k.gy;
            const int rowBase = in0 ? (nz * k.gy + ny) * k.gx : 0;
            const uint32_t a = cellStart[rowBase + xlo0], b = cellStart[rowBase + xhi0 + 1];
            qa0[r] = in0 ? a : 0u; qb0[r] = in0 ? b : 0u;
        }
        {
            const int nz = cz1 + dz, ny = cy1 + dy;
            const bool in1 = live1 && nz >= 0 && nz < k.gz && ny >= 0 && ny < k.gy;
            const int rowBase = in1 ? (nz * k.gy + ny) * k.gx : 0;
            const uint32_t a = cellStart[rowBase + xlo1], b = cellStart[rowBase + xhi1 + 1];
            qa1[r] = in1 ? a : 0u; qb1[r] = in1 ? b : 0u;
        }
    }
    // The lane scans ONE contiguous span per row that covers both targets' runs (adjacent targets: the runs
    // coincide or overlap; across an x-row end they are adjacent in memory); in-stencil bits keep each target exact.
    // Union over the wave: lanes hold ascending sorted slots, so the bounds ascend with the lane.
    uint32_t wA[9], wL[9];                             // wave-uniform: first slot and length of each row's window
    uint32_t wtotal = 0;
    bool slowWave = (dbg & 4) != 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const bool ne0 = qb0[r] > qa0[r], ne1 = qb1[r] > qa1[r];
        const uint32_t start = ne0 ? (ne1 ? min(qa0[r], qa1[r]) : qa0[r]) : qa1[r];
        const uint32_t end = ne0 ? (ne1 ? max(qb0[r], qb1[r]) : qb0[r]) : qb1[r];
        const bool ne = ne0 || ne1;
        const unsigned long long mne = __ballot(ne);
        uint32_t A = 0, L = 0;
        if (mne != 0ull) {
            const int lf = __ffsll((long long)mne) - 1, ll = 63 - __clzll((long long)mne);
            A = (uint32_t)__builtin_amdgcn_readlane((int)start, lf);
            const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)end, ll);
            if (__any(ne && (start < A || end > B)) || B < A) slowWave = true;
            L = B - A;
        }
        wA[r] = A; wL[r] = L; wtotal += L;
    }
    if (wtotal > (uint32_t)WCAP) slowWave = true;      // wave-uniform

    // ---- sweep 1: stage the 9 candidate rows (row r+1's loads fly while row r is scanned), build the list ----
    int cnt = 0, wtot = 0;
    int self0 = -1, self1 = -1;
    if (!slowWave) {
        float4 J[4];
        auto issue = [&](int r) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t i = (uint32_t)lane + 64u * (uint32_t)c;
                J[c] = S.posI[wA[r] + min(i, wL[r] ? wL[r] - 1u : 0u)];
            }
        };
        issue(0);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const uint32_t A = wA[r], L = wL[r];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t i = (uint32_t)lane + 64u * (uint32_t)c;
                if (i < L) { wx[wtot + i] = J[c].x; wy[wtot + i] = J[c].y; wz[wtot + i] = J[c].z; wi[wtot + i] = J[c].w; ws[wtot + i] = A + i; }
            }
            for (uint32_t i = (uint32_t)lane + 256u; i < L; i += 64u) {      // rows longer than 256 candidates
                const float4 Jx = S.posI[A + i];
                wx[wtot + i] = Jx.x; wy[wtot + i] = Jx.y; wz[wtot + i] = Jx.z; wi[wtot + i] = Jx.w; ws[wtot + i] = A + i;
            }
            if (r < 8) issue(r + 1);
            __builtin_amdgcn_wave_barrier();
            if (r == 4) { self0 = wtot + (s0 - (int)A); self1 = wtot + (s1 - (int)A); }
            const bool ne0 = qb0[r] > qa0[r], ne1 = qb1[r] > qa1[r];
            const uint32_t start = ne0 ? (ne1 ? min(qa0[r], qa1[r]) : qa0[r]) : qa1[r];
            const uint32_t end = ne0 ? (ne1 ? max(qb0[r], qb1[r]) : qb0[r]) : qb1[r];
            const uint32_t span = (ne0 || ne1) ? end - start : 0u;
            const uint32_t base = (uint32_t)wtot + (start - A);
            const uint32_t mlo0 = qa0[r] - start, len0 = ne0 ? qb0[r] - qa0[r] : 0u, mlo1 = qa1[r] - start, len1 = ne1 ? qb1[r] - qa1[r] : 0u;
            if (span) {
                float x = wx[base], y = wy[base], z = wz[base];
#pragma unroll 2
                for (uint32_t m = 0; m < span; ++m) {
                    const uint32_t idx = base + m, idn = base + min(m + 1u, span - 1u);
                    const float xn = wx[idn], yn = wy[idn], zn = wz[idn];       // next candidate's reads fly during this one's arithmetic
                    const v2f ex = mx - x, ey = my - y, ez = mz - z;
                    const v2f d2 = t_dot3(ex, ey, ez, ex, ey, ez);
                    const bool v0 = (m - mlo0) < len0, v1 = (m - mlo1) < len1;   // candidate inside the target's own 3-cell run
                    const bool pass = (v0 && d2.x < R2.x) || (v1 && d2.y < R2.y);
                    nl[min(cnt, LN)][lane] = (uint16_t)(idx | (v0 ? 0x4000u : 0u) | (v1 ? 0x8000u : 0u));   // row LN: scratch
                    cnt += pass ? 1 : 0;
                    x = xn; y = yn; z = zn;
                }
            }
            wtot += (int)L;
            __builtin_amdgcn_wave_barrier();
        }
    }

    bool need0 = false, need1 = false;
    if (!slowWave) {
        const bool listOk = cnt <= LN && !(dbg & 1);
        const int nEnt = min(cnt, LN);
        // The walks are software-pipelined by hand (the wave runs at 1-2 waves per SIMD because of its LDS windows, so
        // nothing else hides latency): entry t+D's LDS reads and its velocity gather are issued while entry t is computed.
        constexpr int D = 4;
        auto walk = [&](auto gatherTag, auto&& f) {
            constexpr bool GATHER = decltype(gatherTag)::value;
            if (nEnt <= 0) return;
            uint32_t E[D]; float X[D], Y[D], Z[D], I[D]; float4 R[D];
            auto fetch = [&](int u, int t) {
                const uint32_t e = nl[min(t, nEnt - 1)][lane];
                const uint32_t idx = e & 0x3fffu;
                E[u] = e; X[u] = wx[idx]; Y[u] = wy[idx]; Z[u] = wz[idx];
                if (GATHER) { I[u] = wi[idx]; R[u] = S.velP[ws[idx]]; }
            };
#pragma unroll
            for (int u = 0; u < D; ++u) fetch(u, u);
            for (int t0 = 0; t0 < nEnt; t0 += D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    const uint32_t e = E[u];
                    const float x = X[u], y = Y[u], z = Z[u], iv = GATHER ? I[u] : 0.0f;
                    const float4 jv = GATHER ? R[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                    fetch(u, t0 + D + u);
                    f(e, x, y, z, iv, jv, t0 + u < nEnt);
                }
            }
        };
        // ---- density over the list (self included; entries outside h add +0).  Every sweep honours the in-stencil
        //      bits: on tiny grids a candidate can sit in the span of one row for target 1 and of another row for target 0 ----
        walk(std::false_type{}, [&](uint32_t e, float x, float y, float z, float, const float4&, bool valid) {
            const v2i ok = {((e & 0x4000u) && valid) ? -1 : 0, ((e & 0x8000u) && valid) ? -1 : 0};
            pair_density(k, o, bc(x), bc(y), bc(z), ok);
        });
        finish_density(k, o);
        // ---- sweep 2 ----
        walk(std::true_type{}, [&](uint32_t e, float x, float y, float z, float iv, const float4& JV, bool valid) {
            const int idx = (int)(e & 0x3fffu);
            const v2i ok = {((e & 0x4000u) && valid && idx != self0) ? -1 : 0, ((e & 0x8000u) && valid && idx != self1) ? -1 : 0};
            pair_force(k, o, bc(x), bc(y), bc(z), bc(JV.x), bc(JV.y), bc(JV.z), bc(JV.w), bc(iv), ok);
        });
        integrate(k, o);
        // the list stays a superset for sweep 3 only while the new position is inside the slack of the prediction
        const v2f ux = o.px - qx, uy = o.py - qy, uz = o.pz - qz;
        const v2f u2 = t_dot3(ux, uy, uz, ux, uy, uz);
        const float lim = 0.98f * eps;
        const bool far0 = !(u2.x <= lim * lim) || (dbg & 2), far1 = !(u2.y <= lim * lim) || (dbg & 2);
        // ---- sweep 3 (only candidates of the ENTRY cell's stencil: the in-stencil bits) ----
        walk(std::true_type{}, [&](uint32_t e, float x, float y, float z, float iv, const float4& JV, bool valid) {
            const int idx = (int)(e & 0x3fffu);
            const v2i ok = {((e & 0x4000u) && valid && idx != self0) ? -1 : 0, ((e & 0x8000u) && valid && idx != self1) ? -1 : 0};
            pair_xsph(k, o, bc(x), bc(y), bc(z), bc(JV.x), bc(JV.y), bc(JV.z), bc(iv), ok);
        });
        const v2f fin = {O0.y, O1.y};
        const v2f foamOut = finish_particle(k, o, fin);
        need0 = live0 && (!listOk || far0); need1 = live1 && (!listOk || far1);
        if (live0 && !need0) store_fields(k, out, s0, fbits(O0.z), fbits(O0.w), o.px.x, o.py.x, o.pz.x, o.vx.x, o.vy.x, o.vz.x, o.ax.x, o.ay.x, o.az.x, o.rho.x, o.prs.x, foamOut.x);
        if (live1 && !need1) store_fields(k, out, s1, fbits(O1.z), fbits(O1.w), o.px.y, o.py.y, o.pz.y, o.vx.y, o.vy.y, o.vz.y, o.ax.y, o.ay.y, o.az.y, o.rho.y, o.prs.y, foamOut.y);
        if (dbg & 8) {   // diagnostics: [1] targets recomputed by the exact fallback, [2] list entries, [3] staged candidates, [4] lanes
            const unsigned long long slowT = (unsigned long long)__popcll(__ballot(need0)) + (unsigned long long)__popcll(__ballot(need1));
            unsigned long long ents = (unsigned long long)cnt;
            for (int d = 32; d >= 1; d >>= 1) ents += (unsigned long long)__shfl_xor((int)ents, d, 64);
            if (lane == 0) { atomicAdd(&stats[1], slowT); atomicAdd(&stats[2], ents & 0xffffffffull); atomicAdd(&stats[3], (unsigned long long)wtot); atomicAdd(&stats[4], 64ull); }
        }
    } else {
        need0 = live0; need1 = live1;
        if ((dbg & 8) && lane == 0) atomicAdd(&stats[0], 1ull);   // [0] waves that fell back as a whole
    }
    // exact fallback (one copy of the code): window / list overflow, a target that left the list's slack
#pragma unroll 1
    for (int hh = 0; hh < 2; ++hh) {
        const bool need = hh ? need1 : need0;
        if (need) sph_slow_one(k, S, in, out, order, cellStart, hh ? s1 : s0);
    }
}

}  // namespace sph
