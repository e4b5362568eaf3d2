// sph_tile.h -- LDS-tiled 27-cell SPH pass for gfx950 (the engine's default neighbour kernel).
//
// One workgroup owns a tile of TX x TY x TZ grid cells.  It stages every particle of the
// tile plus a one-cell halo into LDS once (each (y,z) row of the halo box is ONE contiguous
// range of the cell-sorted order, so staging is row-wise coalesced), then each thread runs
// SPHFluid.comp main() (shaders/SPHFluid.comp:66-221) for one particle of the tile against
// LDS instead of chasing cellHead/particleNext through HBM:
//   sweep 1  walks the 9 contiguous 3-cell runs of the 27-cell stencil (ascending cell
//            index = canonical order), accumulates density, and records the candidates
//            inside an inflated radius in a per-thread LDS list;
//   sweep 2  (forces) and sweep 3 (XSPH) walk that list instead of re-scanning 27 cells.
// The exact accept tests of the shader are re-evaluated on every list entry, so the list is
// only a superset filter; results are bit-identical to k_sph_gather and to the oracle.
// Fallbacks keep every case exact: list overflow or a displacement larger than the list's
// slack -> full LDS re-scan for that thread; tile overflow (more halo particles than MAXC)
// -> per-particle global gather for that tile.
#pragma once
#include "sph_kernels.h"

namespace sph {

struct TileGeom {
    int tx, ty, tz;        // tile size in cells
    int ntx, nty, ntz;     // tiles per axis
    int numTiles;
    int debugFlags;        // test hooks: 1 = force list overflow, 2 = force displacement fallback, 4 = force tile overflow
};

struct TilePlan {
    int tx = 8, ty = 4, tz = 4;
    int debugFlags = 0;
};
inline void tile_free(TilePlan&) {}

constexpr int kTileThreads = 256;
constexpr int kMaxCand = 896;       // staged particles (tile + halo) per workgroup
constexpr int kMaxList = 32;        // per-thread neighbour list entries
constexpr int kMaxRows = 64;        // (TY+2)*(TZ+2) halo rows, one wave scans them
constexpr int kMaxHaloCells = 640;  // (TX+2)*(TY+2)*(TZ+2)

struct TileLds {
    float4 pos[kMaxCand];            // x, y, z, density(entry)
    float4 vel[kMaxCand];            // vx, vy, vz, pressure(entry)
    float2 aux[kMaxCand];            // mass/density, 1/(2*density)
    uint32_t rowG[kMaxRows + 1];     // global sorted index of each halo row's first particle
    uint32_t rowL[kMaxRows + 1];     // LDS index of each halo row's first particle
    uint32_t tgtStart[kMaxRows + 1]; // exclusive prefix of targets over interior rows
    uint16_t cellOff[kMaxHaloCells + 2];
    uint16_t list[kMaxList * kTileThreads];
};

// Walk the 9 runs (dz outer, dy inner; each run = cells hx-1..hx+1 of one halo row) of the
// target at halo cell (hx, hy, hz) in canonical order; f(ldsIndex) per candidate.
template <class F>
__device__ __forceinline__ void tile_scan(const TileLds& L, int HX, int HY, int hx, int hy, int hz, F&& f) {
    int off = ((hz - 1) * HY + (hy - 1)) * HX + hx - 1;
    for (int dz = 0; dz < 3; ++dz) {
        for (int dy = 0; dy < 3; ++dy) {
            const uint32_t qs = L.cellOff[off], qe = L.cellOff[off + 3];
            for (uint32_t q = qs; q < qe; ++q) f(q);
            off += HX;
        }
        off += (HY - 3) * HX;
    }
}

__device__ __forceinline__ int upper_row(const uint32_t* a, int n, uint32_t v) {
    // largest r in [0, n) with a[r] <= v   (a is non-decreasing, a[0] == 0)
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kTileThreads) void k_sph_tile(SimK k, TileGeom g, StateIn in, StateOut out,
                                                           const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ cellStart, int n) {
    __shared__ TileLds L;
    const int tid = threadIdx.x;
    // XCD-aware tile mapping: blocks b and b+8 share an XCD (round-robin dispatch), so give
    // each XCD one contiguous chunk of the tile order (neighbouring tiles share halo rows in
    // that XCD's L2).  Pure speed; any mapping is correct.
    int tile;
    {
        const int b = blockIdx.x, per = (g.numTiles + 7) >> 3;
        tile = (b & 7) * per + (b >> 3);
        if (tile >= g.numTiles) return;
    }
    const int tX = tile % g.ntx, tY = (tile / g.ntx) % g.nty, tZ = tile / (g.ntx * g.nty);
    const int x0 = tX * g.tx, y0 = tY * g.ty, z0 = tZ * g.tz;
    const int HX = g.tx + 2, HY = g.ty + 2, HZ = g.tz + 2;
    const int R = HY * HZ;
    const int xlo = max(x0 - 1, 0), xhi = min(x0 + g.tx, k.gx - 1);      // staged x range (inclusive)

    // ---- 1. halo rows: global ranges and LDS offsets (wave 0 scans <= 64 rows) ----
    if (tid < 64) {
        uint32_t gs = 0, cnt = 0;
        if (tid < R) {
            const int hy = tid % HY, hz = tid / HY;
            const int yy = y0 - 1 + hy, zz = z0 - 1 + hz;
            if (yy >= 0 && yy < k.gy && zz >= 0 && zz < k.gz) {
                const int base = (zz * k.gy + yy) * k.gx;
                gs = cellStart[base + xlo];
                cnt = cellStart[base + xhi + 1] - gs;
            }
        }
        const uint32_t inc = wave_incl_scan(cnt);
        if (tid < R) { L.rowG[tid] = gs; L.rowL[tid] = inc - cnt; }
        if (tid == R - 1) L.rowL[R] = inc;
    }
    __syncthreads();
    const uint32_t nC = L.rowL[R];
    // Interior rows of this tile may be empty: nothing to do.
    // (nC counts halo too, so test the targets after the cell table is built.)
    const bool overflow = (nC > (uint32_t)kMaxCand) || (g.debugFlags & 4);

    if (!overflow) {
        // ---- 2. per-cell LDS offsets of the halo box ----
        const int nHalo = HX * R;
        for (int ci = tid; ci < nHalo; ci += kTileThreads) {
            const int r = ci / HX, hx = ci - r * HX;
            const int xg = x0 - 1 + hx;
            const int hy = r % HY, hz = r / HY;
            const int yy = y0 - 1 + hy, zz = z0 - 1 + hz;
            uint32_t off;
            const bool rowIn = (yy >= 0 && yy < k.gy && zz >= 0 && zz < k.gz);
            if (!rowIn || xg < xlo) off = L.rowL[r];
            else if (xg > xhi) off = L.rowL[r + 1];
            else off = L.rowL[r] + (cellStart[(zz * k.gy + yy) * k.gx + xg] - L.rowG[r]);
            L.cellOff[ci] = (uint16_t)off;
        }
        if (tid == 0) { L.cellOff[nHalo] = (uint16_t)nC; L.cellOff[nHalo + 1] = (uint16_t)nC; }
        // ---- 3. stage particles: LDS slot i <- sorted slot rowG[r] + (i - rowL[r]) ----
        for (uint32_t i = tid; i < nC; i += kTileThreads) {
            const int r = upper_row(L.rowL, R + 1, i);
            const uint32_t q = L.rowG[r] + (i - L.rowL[r]);
            const uint32_t src = order[q];
            const float4 P = in.pos[src], V = in.vel[src];
            const float2 RP = in.rp[src];
            L.pos[i] = make_float4(P.x, P.y, P.z, RP.x);
            L.vel[i] = make_float4(V.x, V.y, V.z, RP.y);
            L.aux[i] = make_float2(k.mass / RP.x, 1.0f / (2.0f * RP.x));
        }
        __syncthreads();
        // ---- 4. target prefix over interior rows ----
        const int IR = g.ty * g.tz;
        if (tid < 64) {
            uint32_t cnt = 0;
            if (tid < IR) {
                const int iy = tid % g.ty, iz = tid / g.ty;
                const int r = (iz + 1) * HY + (iy + 1);
                cnt = (uint32_t)L.cellOff[r * HX + 1 + g.tx] - (uint32_t)L.cellOff[r * HX + 1];
            }
            const uint32_t inc = wave_incl_scan(cnt);
            if (tid < IR) L.tgtStart[tid] = inc - cnt;
            if (tid == IR - 1) L.tgtStart[IR] = inc;
        }
        __syncthreads();
        const uint32_t nT = L.tgtStart[IR];

        // ---- 5. one thread per target particle; no barrier below this line ----
        for (uint32_t t = tid; t < nT; t += kTileThreads) {
            const int ir = upper_row(L.tgtStart, IR + 1, t);
            const int iy = ir % g.ty, iz = ir / g.ty;
            const int hy = iy + 1, hz = iz + 1;
            const int r = hz * HY + hy;
            const uint32_t li = (uint32_t)L.cellOff[r * HX + 1] + (t - L.tgtStart[ir]);
            const int s = (int)(L.rowG[r] + (li - L.rowL[r]));
            const uint32_t src = order[s];
            const float4 P = in.pos[src], V = in.vel[src];
            const float foamIn = in.foam[src];
            const uint32_t flags = fbits(P.w), id = fbits(V.w);
            const float4 LP = L.pos[li], LV = L.vel[li];
            Own o;
            own_reset(o);
            o.px = LP.x; o.py = LP.y; o.pz = LP.z; o.vx = LV.x; o.vy = LV.y; o.vz = LV.z; o.rho = LP.w; o.prs = LV.w;
            if (flags & F_GHOST1) {                                    // SPHFluid.comp:72-83
                if (!(flags & F_INACTIVE)) { o.vx = o.vy = o.vz = 0.0f; o.rho = k.rho0; o.prs = 0.0f; }
                out.pos[s] = P;
                out.vel[s] = make_float4(o.vx, o.vy, o.vz, V.w);
                out.rp[s] = make_float2(o.rho, o.prs);
                out.foam[s] = foamIn;
                out.acc[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                continue;
            }
            const int cx = cell_axis(o.px, k.gminx, k.cellSize, k.gx);
            const int hx = cx - (x0 - 1);
            const float ex = o.px, ey = o.py, ez = o.pz;             // entry position

            // list radius: h + slack, slack covers this substep's own displacement
            const float slack = fmaf((fabsf(o.vx) + fabsf(o.vy) + fabsf(o.vz)) * k.dt, 1.25f, 0.05f * k.h);
            const float hl = k.h + slack;
            const float h2list = hl * hl;
            uint32_t cnt = (g.debugFlags & 1) ? (uint32_t)kMaxList + 1u : 0u;

            // sweep 1: density + list build
            tile_scan(L, HX, HY, hx, hy, hz, [&](uint32_t q) {
                const float4 J = L.pos[q];
                const float dx = o.px - J.x, dy = o.py - J.y, dz = o.pz - J.z;
                const float r2 = dot3(dx, dy, dz, dx, dy, dz);
                if (r2 < k.h2) {
                    const float tt = k.h2 - r2;
                    const float w = k.poly6C * ((tt * tt) * tt);
                    o.dens = fmaf(k.mass, w, o.dens);
                }
                if (r2 < h2list && q != li) {
                    if (cnt < (uint32_t)kMaxList) L.list[cnt * kTileThreads + tid] = (uint16_t)q;
                    ++cnt;
                }
            });
            finish_density(k, o);
            const bool listOk = cnt <= (uint32_t)kMaxList;

            // sweep 2: forces
            auto force_one = [&](uint32_t q) {
                const float4 J = L.pos[q], JV = L.vel[q];
                const float2 JA = L.aux[q];
                pair_force_pre(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, J.w, JV.w, JA.x, JA.y);
            };
            if (listOk) {
                for (uint32_t i = 0; i < cnt; ++i) force_one(L.list[i * kTileThreads + tid]);
            } else {
                tile_scan(L, HX, HY, hx, hy, hz, [&](uint32_t q) { if (q != li) force_one(q); });
            }
            integrate(k, o);

            // sweep 3: XSPH with the updated own state; the list is valid only if the
            // displacement stayed inside the slack it was built with.
            const float mx = o.px - ex, my = o.py - ey, mz = o.pz - ez;
            const float moved2 = dot3(mx, my, mz, mx, my, mz);
            const float lim = 0.98f * slack;
            const bool listOk3 = listOk && (moved2 <= lim * lim) && !(g.debugFlags & 2);
            auto xsph_one = [&](uint32_t q) {
                const float4 J = L.pos[q], JV = L.vel[q];
                const float2 JA = L.aux[q];
                pair_xsph_pre(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, J.w, JA.x);
            };
            if (listOk3) {
                for (uint32_t i = 0; i < cnt; ++i) xsph_one(L.list[i * kTileThreads + tid]);
            } else {
                tile_scan(L, HX, HY, hx, hy, hz, [&](uint32_t q) { if (q != li) xsph_one(q); });
            }
            const float foamOut = finish_particle(k, o, foamIn);
            store_particle(k, out, s, flags, id, o, foamOut);
        }
        return;
    }

    // ---- tile overflow: exact per-particle global gather for this tile's targets ----
    for (int ir = 0; ir < g.ty * g.tz; ++ir) {
        const int iy = ir % g.ty, iz = ir / g.ty;
        const int yy = y0 + iy, zz = z0 + iz;
        if (yy >= k.gy || zz >= k.gz) continue;
        const int base = (zz * k.gy + yy) * k.gx;
        const int xe = min(x0 + g.tx, k.gx);
        const uint32_t gs = cellStart[base + x0], ge = cellStart[base + xe];
        for (uint32_t s = gs + tid; s < ge; s += kTileThreads) sph_gather_one(k, in, out, order, cellStart, (int)s);
    }
}

// Host side: pick the tile geometry and launch.  Returns 0 or -(hipError_t).
template <class TimedFactory>
inline int tile_launch(TilePlan& plan, hipStream_t stream, const SimK& k, const StateIn& in, const StateOut& out,
                       const uint32_t* order, const uint32_t* cellStart, int n, TimedFactory&& timed) {
    TileGeom g;
    g.tx = plan.tx; g.ty = plan.ty; g.tz = plan.tz;
    g.debugFlags = plan.debugFlags;
    g.ntx = (k.gx + g.tx - 1) / g.tx; g.nty = (k.gy + g.ty - 1) / g.ty; g.ntz = (k.gz + g.tz - 1) / g.tz;
    g.numTiles = g.ntx * g.nty * g.ntz;
    if ((g.ty + 2) * (g.tz + 2) > kMaxRows || (g.tx + 2) * (g.ty + 2) * (g.tz + 2) > kMaxHaloCells) return -(int)hipErrorInvalidValue;
    const int per = (g.numTiles + 7) / 8;
    auto t = timed(SPH_K_SPH);
    hipLaunchKernelGGL(k_sph_tile, dim3(per * 8), dim3(kTileThreads), 0, stream, k, g, in, out, order, cellStart, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace sph
