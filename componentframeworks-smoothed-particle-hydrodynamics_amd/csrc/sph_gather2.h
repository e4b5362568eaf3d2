// k_sph_gather2 (the default SPH pass): one thread per target over a physically SORTED copy of the
// entry state, wave-cooperative LDS windows for the candidate scan and a per-thread neighbour list in LDS.
//
// Each target walks the 9 contiguous (dy,dz) rows of its 27-cell neighbourhood once (density + a list
// of everything within h of its entry position or of its predicted new position), then sweeps 2 and 3
// touch only the listed neighbours.  The list is private to the thread (column `tid` of an LDS array)
// and the candidate windows are private to the wave, so the kernel has no __syncthreads at all.
// Arithmetic, candidate order and therefore the bits are those of sph_gather_one (and of the oracle).
//
// Why it replaced the LDS-tiled pass (sph_tile.h) as the default: the tiled pass spends about a third
// of a tile's cycles on staging and per-cell candidate lists for ~2 targets per cell, walks an
// inflated-radius mask in sweeps 2/3, and its fixed-size lists, masks and tile pools overflow into
// chunked / sub-box / slow paths when the fluid is compressed.  This kernel has no such capacity
// limits (a long list or a long row only costs that thread / that wave a slower, exact fallback).
#pragma once
#include "sph_kernels.h"

namespace sph {

// tuning knobs (values measured best on MI355X)
#ifndef SPH_G2_LISTU
#define SPH_G2_LISTU 4      // list entries fetched together in sweeps 2 / 3
#endif
#ifndef SPH_G2_WAVES
#define SPH_G2_WAVES 1      // __launch_bounds__ minimum waves per SIMD (1 = let the register allocator decide)
#endif
typedef float v2f __attribute__((ext_vector_type(2)));

struct SortedIn {
    const float4* __restrict__ posI;   // (x, y, z, 1/rho or 0)  in (cell, id) order of THIS substep
    const float4* __restrict__ velP;   // (vx, vy, vz, P)
    const float4* __restrict__ own;    // (rho, foam, bits(flags), bits(id))
};   // written by k_rank<true>

template <int MAXN, int UNROLL, int CAP>
__global__ __launch_bounds__(kBlock, SPH_G2_WAVES) void k_sph_gather2(SimK k, SortedIn S, StateOut out, const uint32_t* __restrict__ cellStart,
                                                        const uint32_t* __restrict__ liveCount, int n) {
    __shared__ uint16_t nl[MAXN][kBlock];      // entry e of thread t: (run << 12) | offset inside the run
    __shared__ uint32_t runLo[9][kBlock];      // first sorted slot of each of the 9 runs
    // CAP > 0: sweep 1 reads its candidates from a wave-private LDS window that the wave fills with
    // coalesced loads (the 64 targets of a wave are consecutive sorted slots, so their 64 candidate
    // ranges of one (dy,dz) row overlap heavily and their union is one short contiguous range).
    __shared__ float4 stage[CAP > 0 ? kBlock / 64 : 1][CAP > 0 ? CAP : 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    // XCD-aware block mapping: blocks b and b+8 run on the same XCD (round-robin dispatch); give each
    // XCD one contiguous eighth of the sorted order, so that its L2 holds only that part of the sorted
    // copy (plus the neighbouring rows) instead of streaming all of it.  Any mapping is correct.
    const int nBlocks = (n + kBlock - 1) / kBlock, perXcd = (nBlocks + 7) >> 3;
    const int vb = ((int)blockIdx.x & 7) * perXcd + ((int)blockIdx.x >> 3);
    if (vb >= nBlocks) return;                               // whole block, uniformly
    const int sRaw = vb * kBlock + tid;
    const int bound = liveCount ? min(n, (int)*liveCount) : n;
    // every lane stays in the kernel to the end (the staging is a wave-wide cooperation); lanes without
    // a target to compute (past the end, halo copies, ghosts) get empty candidate ranges
    bool live = sRaw < bound;
    const int s = live ? sRaw : max(bound - 1, 0);
    const float4 P = S.posI[s], V = S.velP[s], O = S.own[s];
    const uint32_t flags = fbits(O.z), id = fbits(O.w);
    const float foamIn = O.y;
    Own o;
    own_reset(o);
    o.px = P.x; o.py = P.y; o.pz = P.z; o.vx = V.x; o.vy = V.y; o.vz = V.z; o.rho = O.x; o.prs = V.w;
    if (live && (flags & F_HALO)) { out.pos[s] = make_float4(P.x, P.y, P.z, O.z); live = false; }
    if (live && (flags & F_GHOST1)) {                        // SPHFluid.comp:72-83
        float gvx = o.vx, gvy = o.vy, gvz = o.vz, grho = o.rho, gprs = o.prs;
        if (!(flags & F_INACTIVE)) { gvx = gvy = gvz = 0.0f; grho = k.rho0; gprs = 0.0f; }
        out.pos[s] = make_float4(P.x, P.y, P.z, O.z);
        out.vel[s] = make_float4(gvx, gvy, gvz, O.w);
        out.rp[s] = make_float2(grho, gprs);
        out.foam[s] = foamIn;
        if (out.aos) { if (!(flags & F_INACTIVE)) aos_write_active_ghost(out.aos, id - out.idBase, k.rho0); }
        else out.acc[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        live = false;
    }
    const int cx = cell_axis(P.x, k.gminx, k.cellSize, k.gx);
    const int cy = cell_axis(P.y, k.gminy, k.cellSize, k.gy);
    const int cz = cell_z_local(k, P.z);
    const int xlo = max(cx - 1, 0), xhi = min(cx + 1, k.gx - 1);
    // The list must hold every candidate within h of the ENTRY position (sweep 2) and of the position
    // after this substep's integration (sweep 3).  The latter is predicted as entry + 0.995 v dt; what
    // the forces of this substep add to it is covered by eps and checked after integrate().
    const float eps = 0.08f * k.h;
    const float hp = k.h + eps;
    const float hp2 = hp * hp;
    const float qx = fmaf(0.995f * o.vx, k.dt, o.px), qy = fmaf(0.995f * o.vy, k.dt, o.py), qz = fmaf(0.995f * o.vz, k.dt, o.pz);
    // ---- sweep 1: density over every candidate (branch-free: a rejected candidate adds +0), and the list ----
    // all 18 run bounds first (independent loads in flight), then the runs in canonical order
    uint32_t qs[9], qe[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
        const bool in = live && nz >= 0 && nz < k.gz && ny >= 0 && ny < k.gy;
        const int rowBase = in ? (nz * k.gy + ny) * k.gx : 0;
        const uint32_t a = cellStart[rowBase + xlo], b = cellStart[rowBase + xhi + 1];
        qs[r] = in ? a : 0u; qe[r] = in ? b : 0u;
    }
    int cnt = 0;
    bool listOk = true;
    // one candidate slot; an invalid slot (past the end of the run) is moved far away, so that it adds +0
    // to the density and fails both list tests: the loop body has a single, rarely taken branch
    // The two distances (to the entry and to the predicted position) are one packed-fp32 chain
    // (v_pk_add / v_pk_mul / v_pk_fma_f32: two IEEE fp32 lanes per instruction, same rounding as dot3).
    const v2f PX = {o.px, qx}, PY = {o.py, qy}, PZ = {o.pz, qz};
    const v2f LIM = {k.h2hi, hp2};
    auto visit = [&](const float4& J, bool valid, uint32_t q, uint32_t q0, int r) {
        const float jx = valid ? J.x : 3.0e30f;
        const v2f dx = PX - jx, dyy = PY - J.y, dzz = PZ - J.z;
        const v2f d2 = __builtin_elementwise_fma(dzz, dzz, __builtin_elementwise_fma(dyy, dyy, dx * dx));   // (r2, p2)
        const float tt = fmaxf(k.h2 - d2.x, 0.0f);
        o.dens = fmaf(k.mass, k.poly6C * ((tt * tt) * tt), o.dens);
        const v2f e = d2 - LIM;
        if ((fminf(e.x, e.y) < 0.0f) & ((int)q != s)) {
            nl[min(cnt, MAXN - 1)][tid] = (uint16_t)((r << 12) | (int)((q - q0) & 0xfffu));
            ++cnt;
        }
    };
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const uint32_t q0 = qs[r], q1 = qe[r];
        runLo[r][tid] = q0;
        if (q1 - q0 > 4096u) listOk = false;
        bool staged = false;
        uint32_t A = 0;
        if (CAP > 0) {
            // union of the wave's ranges of this row.  Lanes are consecutive sorted slots, so the row bases
            // ascend with the lane: the union runs from the first non-empty lane's start to the last one's
            // end (two readlanes instead of two wave reductions); checked below, with the direct loads as
            // the fallback.
            const bool ne = q1 > q0;
            const unsigned long long mne = __ballot(ne);
            if (mne == 0ull) continue;                         // nobody has a candidate in this row
            const int lf = __ffsll((long long)mne) - 1, ll = 63 - __clzll((long long)mne);
            A = (uint32_t)__builtin_amdgcn_readlane((int)q0, lf);
            const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)q1, ll);
            const bool inside = !__any(ne && (q0 < A || q1 > B));
            staged = inside && B > A && (B - A) <= (uint32_t)CAP;   // wave-uniform
            if (staged) {
                for (uint32_t i = (uint32_t)lane; i < B - A; i += 64u) stage[wv][i] = S.posI[A + i];
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (staged) {
            for (uint32_t q = q0; q < q1; q += UNROLL) {
                float4 J[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) J[u] = stage[wv][min(q + (uint32_t)u, q1 - 1u) - A];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) visit(J[u], q + (uint32_t)u < q1, q + (uint32_t)u, q0, r);
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            for (uint32_t q = q0; q < q1; q += UNROLL) {
                float4 J[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) J[u] = S.posI[min(q + (uint32_t)u, q1 - 1u)];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) visit(J[u], q + (uint32_t)u < q1, q + (uint32_t)u, q0, r);
            }
        }
    }
    listOk = listOk && cnt <= MAXN;
    finish_density(k, o);

    // every candidate again, in canonical order (lists that did not fit, sweep 3 after a long move)
    auto full = [&](auto&& f) {
        for (int r = 0; r < 9; ++r) {
            const int nz = cz + r / 3 - 1, ny = cy + r % 3 - 1;
            if (nz < 0 || nz >= k.gz || ny < 0 || ny >= k.gy) continue;
            const int rowBase = (nz * k.gy + ny) * k.gx;
            const uint32_t qs = cellStart[rowBase + xlo], qe = cellStart[rowBase + xhi + 1];
            for (uint32_t q = qs; q < qe; ++q) {
                if ((int)q == s) continue;
                f(S.posI[q], S.velP[q]);
            }
        }
    };
    auto listed = [&](auto&& f) {
        for (int e = 0; e < cnt; e += SPH_G2_LISTU) {
            float4 J[SPH_G2_LISTU], JV[SPH_G2_LISTU];
#pragma unroll
            for (int u = 0; u < SPH_G2_LISTU; ++u) {
                const uint32_t a = nl[min(e + u, cnt - 1)][tid];
                const uint32_t q = runLo[a >> 12][tid] + (a & 0xfffu);
                J[u] = S.posI[q]; JV[u] = S.velP[q];
            }
#pragma unroll
            for (int u = 0; u < SPH_G2_LISTU; ++u) if (e + u < cnt) f(J[u], JV[u]);
        }
    };
    auto force_at = [&](const float4& J, const float4& JV) {
        if (J.w > 0.0f) pair_force_pre(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, JV.w, J.w);
    };
    auto xsph_at = [&](const float4& J, const float4& JV) {
        if (J.w > 0.0f) pair_xsph_pre(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, J.w);
    };
    // ---- sweep 2 ----
    if (listOk) listed(force_at); else full(force_at);
    integrate(k, o);
    // ---- sweep 3: the list stays a superset only while the displacement is inside its slack ----
    const float mx = o.px - qx, my = o.py - qy, mz = o.pz - qz;
    const float lim = 0.98f * eps;
    if (listOk && dot3(mx, my, mz, mx, my, mz) <= lim * lim) listed(xsph_at); else full(xsph_at);
    const float foamOut = finish_particle(k, o, foamIn);
    if (live) store_particle(k, out, s, flags, id, o, foamOut);
}

}  // namespace sph
