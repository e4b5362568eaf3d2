// sph_tile.h -- k_sph_tile: the SPH pass (SPHFluid.comp:66-221 + fused OBBConstraints.comp), round-4 form.
//
// One WORKGROUP per compact block of TX x TY x TZ cells.  The block's candidates -- every particle of the (TX+2) x (TY+2) x (TZ+2)
// cell hull, (TY+2)(TZ+2) contiguous runs of the sorted copy -- are staged ONCE into LDS as (x, y, z, 1/rho) and (vx, vy, vz, P);
// each lane then takes one target of the block and runs the three sweeps of the shader entirely out of LDS:
//   sweep 1   the 9 candidate rows of the target in canonical order (density + the two-ball neighbour list, as k_sph_walk);
//   sweeps 2/3 per-lane walks over the list; an entry IS the candidate's byte offset in the LDS image (no row-base table, no
//             address arithmetic, no vector-memory instruction: k_sph_walk spent its time on the L1 tag lookups of its gathers).
// Same arithmetic contract, same candidate order, therefore the same bits as k_sph_walk / k_sph_slow.  A compact block re-reads
// 2.8 x its own particles (8 x 4 x 4 cells) where a wave of k_sph_walk staged nine row windows (9.3 x, 16-byte halves).
// Blocks whose hull or target count does not fit the LDS image (compressed fluid) are flagged in tileDone[] and left to
// k_sph_walk, which skips the targets of every block this kernel has done.
#pragma once
#include "sph_walk.h"

namespace sph {

#ifndef SPH_TILE_WAVES
#define SPH_TILE_WAVES 6      // waves per workgroup = 64 x this many targets per block at most
#endif
#ifndef SPH_TILE_CAP
#define SPH_TILE_CAP 896      // candidates in the LDS image
#endif
#ifndef SPH_TILE_MAXN
#define SPH_TILE_MAXN 24      // list entries per target
#endif
#ifndef SPH_TILE_UNROLL
#define SPH_TILE_UNROLL 3
#endif
#ifndef SPH_TILE_OCC
#define SPH_TILE_OCC 3        // __launch_bounds__ minimum waves per SIMD
#endif

template <int TX, int TY, int TZ, int NW, int CAP, int MAXN, int UNROLL, bool SMALLH>
__global__ __launch_bounds__(NW * 64, SPH_TILE_OCC) void k_sph_tile(SimK k, SortedIn S, StateIn in, StateOut out, const uint32_t* __restrict__ order,
                                                                      const uint32_t* __restrict__ cellStart, TileGeom tg, uint8_t* __restrict__ tileDone,
                                                                      uint32_t* __restrict__ nFallback, int dbg, unsigned long long* __restrict__ stats) {
    constexpr int kB = NW * 64;
    constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2, NR = HY * HZ, NT = TY * TZ;
    static_assert(NR <= 64, "the hull's rows are planned by the lanes of one wave");
    static_assert(CAP * 16 <= 65536, "a list entry is the 16-bit byte offset of a candidate in the LDS image");
    constexpr uint32_t kRowBytes = kB * 2;                 // one list row = one entry of every thread
    constexpr int kSpare = UNROLL > 2 ? UNROLL : 2;        // rows past MAXN: absorb the writes of a full list / the walks' look-ahead
    __shared__ float4 candP[CAP];                          // (x, y, z, 1 / rho) of the hull's particles, row after row of the hull
    __shared__ float4 candV[CAP];                          // (vx, vy, vz, P)
    __shared__ uint16_t nl[MAXN + kSpare][kB];             // entry e of thread t: byte offset of the candidate in candP / candV
    __shared__ uint32_t rowG[NR];                          // sorted slot of the first candidate of each hull row
    __shared__ uint32_t rowL[NR + 1];                      // index in the image of the first candidate of each hull row; [NR] = candidates
    __shared__ uint16_t cellL[NR][HX + 1];                 // index in the image of the first candidate of each hull cell; [HX] = end of the row
    __shared__ uint32_t tgtG[NT];                          // sorted slot of the first target of each row of the block
    __shared__ uint32_t tgtRow[NT + 1];                    // targets before each row of the block; [NT] = targets
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    // XCD-aware block mapping: workgroups b and b + 8 share an XCD; each XCD takes one contiguous eighth of the blocks (z-major),
    // so that its L2 holds one slab of the sorted copy.  Any mapping is correct.
    const int nTiles = tg.ntx * tg.nty * tg.ntz, perXcd = (nTiles + 7) >> 3;
    const int vb = ((int)blockIdx.x & 7) * perXcd + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= perXcd || vb >= nTiles) return;   // whole workgroup, uniformly
    const int x0 = (vb % tg.ntx) * TX, y0 = ((vb / tg.ntx) % tg.nty) * TY, z0 = (vb / (tg.ntx * tg.nty)) * TZ;

    // ---- plan: the runs of the sorted copy under the hull's rows and under the block's own rows (one lane per hull row).
    // Every thread also requests NOW the cellStart words its entries of the cell table need (they depend on the thread index only),
    // so that the image costs two round trips to memory (cellStart; the records) instead of three.
    constexpr int kCellEnts = NR * (HX + 1), kCellPer = (kCellEnts + kB - 1) / kB;
    uint32_t cellWord[kCellPer];
#pragma unroll
    for (int i = 0; i < kCellPer; ++i) {
        const int j = tid + i * kB;
        const int hr = j / (HX + 1), c = j - hr * (HX + 1);
        const int y = y0 - 1 + hr % HY, z = z0 - 1 + hr / HY;
        cellWord[i] = 0xffffffffu;                         // (row outside the grid: empty)
        if (j < kCellEnts && y >= 0 && y < k.gy && z >= 0 && z < k.gz) cellWord[i] = cellStart[(z * k.gy + y) * k.gx + min(max(x0 - 1 + c, 0), k.gx)];
    }
    if (wv == 0) {
        uint32_t start = 0, len = 0, tstart = 0, tlen = 0;
        const int hy = lane % HY, hz = lane / HY;
        const int y = y0 - 1 + hy, z = z0 - 1 + hz;
        const bool interior = lane < NR && hy >= 1 && hy <= TY && hz >= 1 && hz <= TZ;
        if (lane < NR && y >= 0 && y < k.gy && z >= 0 && z < k.gz) {
            const int rowBase = (z * k.gy + y) * k.gx;
            start = cellStart[rowBase + max(x0 - 1, 0)];
            len = cellStart[rowBase + min(x0 + TX + 1, k.gx)] - start;
            if (interior) {
                tstart = cellStart[rowBase + x0];
                tlen = cellStart[rowBase + min(x0 + TX, k.gx)] - tstart;
            }
        }
        const uint32_t inc = wave_incl_scan(len), tinc = wave_incl_scan(tlen);
        if (lane < NR) { rowG[lane] = start; rowL[lane] = inc - len; }
        if (lane == 63) { rowL[NR] = inc; tgtRow[NT] = tinc; }
        if (interior) { const int ir = (hz - 1) * TY + (hy - 1); tgtG[ir] = tstart; tgtRow[ir] = tinc - tlen; }
    }
    __syncthreads();
    const int T = (int)tgtRow[NT];
    const uint32_t M = rowL[NR];
    if (T > kB || M > (uint32_t)CAP || (dbg & 16)) {       // does not fit: k_sph_walk takes this block's targets
        if (tid == 0) { tileDone[vb] = 0; if (T > 0) { atomicAdd(nFallback, 1u); if (dbg & 8) atomicAdd(&stats[0], 1ull); } }
        return;
    }
    if (tid == 0) { tileDone[vb] = 1; if ((dbg & 8) && T > 0) atomicAdd(&stats[3], 1ull); }
    if (T == 0) return;

    // ---- this lane's target (the tid-th particle of the block's rows) and its own data, requested before the image is filled ----
    bool live = tid < T;
    uint32_t slot;
    {
        const uint32_t t = live ? (uint32_t)tid : 0u;
        int ir = 0;
#pragma unroll
        for (int j = 1; j < NT; ++j) ir += (t >= tgtRow[j]) ? 1 : 0;
        slot = tgtG[ir] + (t - tgtRow[ir]);
    }
    const float4 O = S.own[slot];

    // ---- the image: cell offsets, the candidates' records (coalesced 16-byte loads of the 32-byte records) ----
#pragma unroll
    for (int i = 0; i < kCellPer; ++i) {
        const int j = tid + i * kB;
        const int hr = j / (HX + 1), c = j - hr * (HX + 1);
        if (j < kCellEnts) cellL[hr][c] = (uint16_t)(cellWord[i] == 0xffffffffu ? rowL[hr] : rowL[hr] + (cellWord[i] - rowG[hr]));
    }
    {
        constexpr int RPW = (NR + NW - 1) / NW;            // hull rows per wave
        float4 v[RPW];
        uint32_t l0[RPW], len2[RPW], g2[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int hr = wv + i * NW;
            l0[i] = len2[i] = g2[i] = 0u;
            v[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (hr < NR) {
                g2[i] = 2u * rowG[hr]; l0[i] = rowL[hr]; len2[i] = 2u * (rowL[hr + 1] - l0[i]);
                if ((uint32_t)lane < len2[i]) v[i] = S.pv[g2[i] + (uint32_t)lane];
            }
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            if ((uint32_t)lane < len2[i]) ((lane & 1) ? candV : candP)[l0[i] + ((uint32_t)lane >> 1)] = v[i];
            for (uint32_t j = (uint32_t)lane + 64u; j < len2[i]; j += 64u) ((j & 1u) ? candV : candP)[l0[i] + (j >> 1)] = S.pv[g2[i] + j];
        }
    }
    __syncthreads();

    // ---- one target per lane ----
    const int s = (int)slot;
    const uint32_t cb = fbits(O.x);
    const int cx = (int)(cb & 1023u), cy = (int)((cb >> 10) & 1023u), cz = (int)(cb >> 20);
    const int hx = cx - x0 + 1, hy = cy - y0 + 1, hz = cz - z0 + 1;
    const int hr4 = hz * HY + hy;
    const uint32_t eSelf = (rowL[hr4] + (slot - rowG[hr4])) * 16u;   // the target's own place in the image
    const char* const pBytes = reinterpret_cast<const char*>(&candP[0]);
    const char* const vBytes = reinterpret_cast<const char*>(&candV[0]);
    const float4 P = *reinterpret_cast<const float4*>(pBytes + eSelf), V = *reinterpret_cast<const float4*>(vBytes + eSelf);
    if (live && special_slot(k, S, in, out, order, s, P, V, O)) live = false;
    if (!__any(live)) return;                              // (no workgroup barrier below)
#if defined(SPH_TILE_CUT) && SPH_TILE_CUT == 3   // timing experiment only: stop behind the image (the state passes through unchanged)
    if (live) store_fields(k, out, s, fbits(O.z), fbits(O.w), P.x, P.y, P.z, V.x, V.y, V.z, 0.0f, 0.0f, 0.0f, 1.0f / P.w, V.w, O.y, cz);
    return;
#endif
    Own o;
    own_reset(o);
    o.px = P.x; o.py = P.y; o.pz = P.z; o.vx = V.x; o.vy = V.y; o.vz = V.z; o.rho = 0.0f; o.prs = 0.0f;
    // byte offsets [qa, qb) of the target's candidates in each of its 9 rows (canonical order: dz outer, dy, dx inner)
    uint32_t qa[9], qb[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int hrr = (hz + r / 3 - 1) * HY + (hy + r % 3 - 1);
        const uint32_t a = cellL[hrr][hx - 1], b = cellL[hrr][hx + 2];
        qa[r] = live ? a * 16u : 0u; qb[r] = live ? b * 16u : 0u;
    }
    // The list: as k_sph_walk (two balls: within h of the entry position, or within h + eps of the free-flight prediction).
    const float eps = SPH_WALK_EPS * k.h;
    const float hp = k.h + eps;
    const float qx = fmaf(0.995f * fmaf(k.gravx, k.dt, o.vx), k.dt, o.px), qy = fmaf(0.995f * fmaf(k.gravy, k.dt, o.vy), k.dt, o.py),
                qz = fmaf(0.995f * fmaf(k.gravz, k.dt, o.vz), k.dt, o.pz);
    const float mvx = qx - o.px, mvy = qy - o.py, mvz = qz - o.pz;
    const float ex = mvx + mvx, ey = mvy + mvy, ez = mvz + mvz;
    const float mm = dot3(mvx, mvy, mvz, mvx, mvy, mvz);
    const float c0 = mm - (hp * hp) * 1.0001f;
    constexpr float kBig = 0x1p40f;
    // (the rounding of w grows with |m|^2: beyond 16 h of predicted move -- an uploaded or impulse-driven velocity far above the cap --
    //  the sign of s is no longer safe, and such a target takes the exact sweeps)
    bool listOk = !(dbg & 1) && !(mm > 256.0f * k.h2);
    uint32_t cur = (uint32_t)tid * 2u;
    const uint32_t curEnd = (uint32_t)tid * 2u + (uint32_t)MAXN * kRowBytes;
    const uint32_t adv = live ? kRowBytes : 0u;
    char* const nlBytes = reinterpret_cast<char*>(&nl[0][0]);
    auto visit = [&](const float4& J, uint32_t e, bool selfRow) {
        const float dx = o.px - J.x, dy = o.py - J.y, dz = o.pz - J.z;
        const float r2 = dot3(dx, dy, dz, dx, dy, dz);
        const float t = SMALLH ? __builtin_amdgcn_fmed3f(k.h2 - r2, 0.0f, 1.0f) : fmaxf(k.h2 - r2, 0.0f);
        o.dsum = fmaf(t * t, t, o.dsum);
        const float w = fmaf(ez, dz, fmaf(ey, dy, fmaf(ex, dx, r2 + c0)));
        float sg = fmaf(-kBig, t, w);                      // < 0 iff r2 < h2 or |d + m| < hp
        if (selfRow) sg = (e == eSelf) ? 1.0f : sg;        // the target itself: density only
        *reinterpret_cast<uint16_t*>(nlBytes + cur) = (uint16_t)e;
        cur += (uint32_t)((int32_t)fbits(sg) >> 31) & adv;
    };
    // ---- sweep 1: the candidates of a group are requested one group ahead (across the rows too), as whole 16-byte reads ----
    auto ldP = [&](uint32_t at) {
        const float4 J = *reinterpret_cast<const float4*>(pBytes + at);
        asm volatile("" ::"v"(J.w));                       // (keeps the read a ds_read_b128: 4 LDS cycles; a 12-byte read takes 8)
        return J;
    };
    float4 Jn[UNROLL];                                     // first group of the NEXT row
    bool haveN = qa[0] + 16u * UNROLL <= qb[0];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) Jn[u] = ldP(haveN ? qa[0] + 16u * (uint32_t)u : 0u);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const bool selfRow = (r == 4);
        uint32_t a = qa[r];
        const uint32_t b = qb[r];
        float4 Jc[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) Jc[u] = Jn[u];
        if (r < 8) {
            haveN = qa[r + 1] + 16u * UNROLL <= qb[r + 1];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) Jn[u] = ldP(haveN ? qa[r + 1] + 16u * (uint32_t)u : 0u);
        }
        while (a + 16u * UNROLL <= b) {
            float4 J[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) J[u] = Jc[u];
            const uint32_t a2 = a + 16u * UNROLL;
            const bool more = a2 + 16u * UNROLL <= b;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) Jc[u] = ldP(more ? a2 + 16u * (uint32_t)u : 0u);
            cur = min(cur, curEnd);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) visit(J[u], a + 16u * (uint32_t)u, selfRow);
            a = a2;
        }
        for (; a < b; a += 16u) {
            const float4 J = ldP(a);
            cur = min(cur, curEnd);
            visit(J, a, selfRow);
        }
    }
    listOk = (listOk && cur < curEnd) || !live;
    finish_density(k, o);

    // ---- walks of sweeps 2 / 3: per-lane loops over the list, two entries ahead, everything out of LDS ----
    auto fetch = [&](uint32_t at, float4& J, float4& JV) {
        const uint32_t ent = *reinterpret_cast<const uint16_t*>(nlBytes + at);
        J = *reinterpret_cast<const float4*>(pBytes + ent); JV = *reinterpret_cast<const float4*>(vBytes + ent);
    };
    auto listed = [&](auto&& f) {
        const uint32_t end = cur;
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
    // Exact fallback of a sweep for lanes whose list cannot be used: every candidate again, in canonical order, out of LDS.
    auto plain = [&](auto&& f) {
#pragma unroll 1
        for (int r = 0; r < 9; ++r) {
            const int hrr = (hz + r / 3 - 1) * HY + (hy + r % 3 - 1);
            const uint32_t b = (uint32_t)cellL[hrr][hx + 2] * 16u;
            auto within = [&](const float4& J) { const float dx = o.px - J.x, dy = o.py - J.y, dz = o.pz - J.z; return dot3(dx, dy, dz, dx, dy, dz) < k.h2; };
            for (uint32_t a = (uint32_t)cellL[hrr][hx - 1] * 16u; a < b; a += 16u) {
                const float4 J = *reinterpret_cast<const float4*>(pBytes + a);
                // (a candidate outside h of every lane that is here adds +-0 everywhere: the wave skips its pair arithmetic)
                if (__any(within(J))) f(J, *reinterpret_cast<const float4*>(vBytes + a), (int32_t)(a != eSelf ? -1 : 0));
            }
        }
    };
    auto force_at = [&](const float4& J, const float4& JV) { pair_force_other(k, o, J, JV); };
    auto xsph_at = [&](const float4& J, const float4& JV) { pair_xsph_other<SMALLH>(k, o, J, JV); };
    auto force_plain = [&](const float4& J, const float4& JV, int32_t ok) { pair_force(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, JV.w, J.w, ok); };
    auto xsph_plain = [&](const float4& J, const float4& JV, int32_t ok) { pair_xsph(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, J.w, ok); };
#if defined(SPH_TILE_CUT) && SPH_TILE_CUT == 1   // timing experiment only: stop after sweep 1
    if (live) store_fields(k, out, s, fbits(O.z), fbits(O.w), o.px, o.py, o.pz, o.vx, o.vy, o.vz, (float)cur, o.ay, o.az, o.rho, o.prs, O.y, cz);
    return;
#endif
    // ---- sweep 2 ----
    if (listOk) listed(force_at); else if (live) plain(force_plain);
    integrate(k, o);
#if defined(SPH_TILE_CUT) && SPH_TILE_CUT == 2   // timing experiment only: stop after sweep 2
    if (live) store_fields(k, out, s, fbits(O.z), fbits(O.w), o.px, o.py, o.pz, o.vx, o.vy, o.vz, o.ax, o.ay, o.az, o.rho, o.prs, O.y, cz);
    return;
#endif
    // ---- sweep 3: the list stays a superset only while the displacement is inside its slack ----
    const float mx = o.px - qx, my = o.py - qy, mz = o.pz - qz;
    const float lim = 0.98f * eps;
    const bool near = (dot3(mx, my, mz, mx, my, mz) <= lim * lim && !(dbg & 2)) || !live;
    if (listOk && near) listed(xsph_at); else if (live) plain(xsph_plain);
    const float foamOut = finish_particle(k, o, O.y);
    if (live) store_fields(k, out, s, fbits(O.z), fbits(O.w), o.px, o.py, o.pz, o.vx, o.vy, o.vz, o.ax, o.ay, o.az, o.rho, o.prs, foamOut, cz);
    if (dbg & 8) {   // diagnostics as k_sph_walk: [1] targets on an exact fallback sweep, [2] list entries, [4] lanes, [5] overflowed lists, [6] far targets, [7] waves with a fallback
        const unsigned long long slowT = (unsigned long long)__popcll(__ballot(live && !(listOk && near)));
        unsigned long long ents = (unsigned long long)((live && listOk) ? (cur - (uint32_t)tid * 2u) / kRowBytes : 0u);
        for (int d = 32; d >= 1; d >>= 1) ents += (unsigned long long)__shfl_xor((int)ents, d, 64);
        const unsigned long long ovf = (unsigned long long)__popcll(__ballot(live && !listOk)), far = (unsigned long long)__popcll(__ballot(live && listOk && !near));
        if (lane == 0) { atomicAdd(&stats[1], slowT); atomicAdd(&stats[2], ents & 0xffffffffull); atomicAdd(&stats[4], 64ull); atomicAdd(&stats[5], ovf); atomicAdd(&stats[6], far); atomicAdd(&stats[7], slowT ? 1ull : 0ull); }
    }
}

}  // namespace sph
