// sph_tile.h -- LDS-tiled 27-cell SPH pass for gfx950 (the engine's default neighbour kernel).
//
// One workgroup owns a tile of TX x TY x TZ grid cells and walks it in z-slices thin enough
// for LDS.  Per slice:
//   stage    every particle of the slice plus a one-cell halo goes to LDS once; each (y,z) row
//            of the halo box is ONE contiguous range of the cell-sorted order, so staging is
//            row-wise coalesced.  32 B per particle: (x, y, z, 1/rho_entry) (vx, vy, vz, P_entry).
//   lists    per interior cell, the byte offsets of its 27-cell candidates in canonical order
//            (ascending cell index, then ascending particle id), padded with a far-away
//            sentinel, so the hot loop needs no bounds checks and no run bookkeeping.
//   sweep 1  one thread per target particle: 8 candidates per step (one b128 index read,
//            eight position reads in flight), branch-free density accumulation, and a
//            96-bit per-thread mask of the candidates inside an inflated radius.
//   sweep 2/3 (forces, XSPH) walk the set bits of that mask instead of re-scanning 27 cells;
//            the shader's exact accept tests are re-evaluated on every entry, so the mask is
//            only a superset filter.
// Results are bit-identical to k_sph_gather and to the oracle.  Targets the fast path cannot
// take -- a cell with more candidates than the mask holds, a displacement larger than the
// mask's slack, a slice whose halo exceeds LDS even at one cell layer -- are queued and
// finished by k_sph_slow with the exact per-particle global gather.
#pragma once
#include "sph_kernels.h"

namespace sph {

struct TileGeom {
    int tx, ty, tz;        // tile size in cells
    int ntx, nty, ntz;     // tiles per axis
    int numTiles;
    int debugFlags;        // test hooks: 1 = force list overflow, 2 = force sweep-3 re-scan, 4 = force slice overflow, 8 = stamps
};

struct TilePlan {
    int tx = 8, ty = 4, tz = 4;
    int debugFlags = 0;
    int config = 0;        // 0 = TileCfgA, 1 = TileCfgB, 2 = TileCfgC
};
inline void tile_free(TilePlan&) {}

constexpr int kMaxRows = 64;         // (TY+2)*(TZ+2) halo rows, one wave scans them
constexpr int kMaxHaloCells = 640;   // (TX+2)*(TY+2)*(TZ+2)
constexpr int kMaxCells = 128;       // interior cells per slice
constexpr int kMaskBits = 96;        // candidates a per-thread mask can describe

// LDS budget / workgroup shape of one build of the kernel.  160 KiB of LDS per CU: 53 KB
// per workgroup -> 3 workgroups per CU, 40 KB -> 4, 32 KB -> 5.
template <int THREADS, int MAXCAND, int POOL, int MINWAVES = 1>
struct TileCfg {
    static constexpr int kMinWaves = MINWAVES; // __launch_bounds__ waves per SIMD the register allocator must allow
    static constexpr int kThreads = THREADS;   // multiple of 64
    static constexpr int kMaxCand = MAXCAND;   // staged particles per slice; slot kMaxCand is the sentinel
    static constexpr int kListPool = POOL;     // u16 entries shared by the slice's per-cell lists
};
using TileCfgA = TileCfg<256, 864, 128 * 88>;  // 52.8 KB: tile 8x4x4, 3 workgroups / CU
using TileCfgB = TileCfg<256, 672, 96 * 72>;   // 38.0 KB: tile 8x4x3, 4 workgroups / CU
using TileCfgD = TileCfg<192, 576, 64 * 104>;  // 34.6 KB: tile 8x4x2, 3 waves, 4 workgroups / CU
using TileCfgE = TileCfg<128, 416, 32 * 104 + 1408>;  // 24.3 KB: tile 8x2x2, 2 waves, 6 workgroups / CU
using TileCfgC = TileCfg<320, 864, 128 * 88, 4>;  // as A with 5 waves (one round for <= 320 targets); 15 waves / CU need <= 128 VGPRs

template <class CFG>
struct TileLdsT {
    static constexpr int kMaxCand = CFG::kMaxCand;
    static constexpr int kListPool = CFG::kListPool;
    float4 pos[kMaxCand + 1];        // x, y, z, (rho > 0 ? 1/rho : 0)
    float4 vel[kMaxCand + 1];        // vx, vy, vz, pressure
    uint32_t rowG[kMaxRows + 1];     // global sorted index of each halo row's first particle
    uint32_t rowL[kMaxRows + 1];     // LDS index of each halo row's first particle
    uint32_t tgtStart[kMaxRows + 1]; // exclusive prefix of targets over interior rows
    uint16_t cellOff[kMaxHaloCells + 2];
    // candidate byte offsets (LDS index * 16) per interior cell; during the prologue the same
    // bytes hold the tile's cellStart values (u32 per halo cell + one row end per row)
    alignas(16) uint16_t cl[kListPool + 8];
    uint16_t clLen[kMaxCells];       // candidates of the cell, 0xFFFF = does not fit the mask
    uint16_t clSelf[kMaxCells];      // list position of the cell's own first particle
    int sliceTz;
    int maxLen;                      // longest 27-cell candidate list of the tile (slice planning)
    int tileOver;                    // the whole tile (plus halo) exceeds kMaxCand
    static_assert((kMaxHaloCells + kMaxRows) * 4 <= kListPool * 2, "prologue cellStart copy must fit the list pool");
};

// Diagnostic build only (SPH_OPT_DEBUG bit 3): per-tile shader-clock sums, reduced on the host.
enum TileStamp {
    TS_PROLOGUE = 0, TS_STAGE, TS_LISTS, TS_SCAN, TS_SWEEP2, TS_SWEEP3, TS_EPILOGUE, TS_TOTAL,   // cycles
    TS_TILES, TS_SLICES, TS_WAVEROUNDS, TS_SCANGROUPS, TS_OVERFLOW_SLICES, TS_SLOW_LANES, TS_TARGETS, TS_CANDIDATES,
    TS_WALK2MAX, TS_WALK2SUM, TS_RESCAN_LANES, TS_L_TGT, TS_L_BUILD,
    TS_COUNT
};
template <bool STAMP>
__device__ __forceinline__ unsigned long long stamp_now() {
    if (!STAMP) return 0ull;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): s_memtime returns through the scalar cache
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
template <bool STAMP>
__device__ __forceinline__ void stamp_add(unsigned long long* st, int which, unsigned long long v) {
    if (STAMP) atomicAdd(&st[which], v);         // st points at this tile's private row of counters
}

// Walk the 9 runs (dz outer, dy inner; each run = cells hx-1..hx+1 of one halo row) of the
// target at halo cell (hx, hy, hz) in canonical order; f(ldsIndex) per candidate.
template <class LDS, class F>
__device__ __forceinline__ void tile_scan(const LDS& L, int HX, int HY, int hx, int hy, int hz, F&& f) {
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

// x / d for 0 <= x < 1024, 1 <= d <= 1024 without the ~35-instruction runtime integer division:
// inv = floor(2^20 / d) + 1 is exact in that range (x * d < 2^20).
__device__ __forceinline__ int fdiv_inv(int d) { return (int)((1u << 20) / (uint32_t)d) + 1; }
__device__ __forceinline__ int fdiv(int x, int inv) { return (int)(((uint32_t)x * (uint32_t)inv) >> 20); }

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

__device__ __forceinline__ float4 lds_f4(const void* base, uint32_t byteOff) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byteOff);
}

// 8 candidates whose byte offsets are packed in `ix` against the query point (qx,qy,qz):
// 8 mask bits (first candidate = bit 7) for r2 < thr2 and, when DENS, the density sum of sweep 1.
template <bool DENS, class LDS>
__device__ __forceinline__ uint32_t scan8(const SimK& k, const LDS& L, Own& o, float qx, float qy, float qz, float thr2, uint4 ix) {
    const uint32_t o0 = ix.x & 0xffffu, o1 = ix.x >> 16, o2 = ix.y & 0xffffu, o3 = ix.y >> 16;
    const uint32_t o4 = ix.z & 0xffffu, o5 = ix.z >> 16, o6 = ix.w & 0xffffu, o7 = ix.w >> 16;
    const float4 J0 = lds_f4(L.pos, o0), J1 = lds_f4(L.pos, o1), J2 = lds_f4(L.pos, o2), J3 = lds_f4(L.pos, o3);
    const float4 J4 = lds_f4(L.pos, o4), J5 = lds_f4(L.pos, o5), J6 = lds_f4(L.pos, o6), J7 = lds_f4(L.pos, o7);
    uint32_t bits = 0;
#define SPH_SCAN1(J)                                                       \
    {                                                                      \
        const float dx = qx - J.x, dy = qy - J.y, dz = qz - J.z;           \
        const float r2 = dot3(dx, dy, dz, dx, dy, dz);                     \
        if (DENS) {                                                        \
            const float tt = fmaxf(k.h2 - r2, 0.0f);                       \
            const float w = k.poly6C * ((tt * tt) * tt);                   \
            o.dens = fmaf(k.mass, w, o.dens);                              \
        }                                                                  \
        bits = (bits << 1) | ((r2 < thr2) ? 1u : 0u);                      \
    }
    SPH_SCAN1(J0) SPH_SCAN1(J1) SPH_SCAN1(J2) SPH_SCAN1(J3) SPH_SCAN1(J4) SPH_SCAN1(J5) SPH_SCAN1(J6) SPH_SCAN1(J7)
#undef SPH_SCAN1
    return bits;
}

// Groups [g0, g1) (8 candidates each, g1 - g0 <= 12) of the list at clp: candidate kk of the
// chunk ends up at bit 8*(g1-g0)-1-kk of the 96-bit register m2:m1:m0.
template <bool DENS, class LDS>
__device__ __forceinline__ void scan_chunk(const SimK& k, const LDS& L, Own& o, const uint16_t* clp, int g0, int g1,
                                           float qx, float qy, float qz, float thr2, uint32_t& m0, uint32_t& m1, uint32_t& m2) {
    m0 = m1 = m2 = 0;
    uint4 ix = *reinterpret_cast<const uint4*>(clp + 8 * g0);
    for (int g = g0; g < g1; ++g) {
        const uint4 nx = *reinterpret_cast<const uint4*>(clp + 8 * (g + 1));   // padded / next list: always mapped
        const uint32_t b = scan8<DENS>(k, L, o, qx, qy, qz, thr2, ix);
        m2 = (m2 << 8) | (m1 >> 24);
        m1 = (m1 << 8) | (m0 >> 24);
        m0 = (m0 << 8) | b;
        ix = nx;
    }
}

// Visit the set bits of m2:m1:m0 from the top (= ascending candidate order); f(pos4, vel4) per entry.
template <class LDS, class F>
__device__ __forceinline__ int walk_chunk(const LDS& L, const uint16_t* clp, int g0, int g1, uint32_t m0, uint32_t m1, uint32_t m2, F&& f) {
    const uint32_t topBit = (uint32_t)(8 * (g1 - g0) - 1);
    const uint16_t* cp = clp + 8 * g0;
    int trips = 0;
    while (m2) {
        const int p = 31 - __clz((int)m2);
        m2 &= ~(1u << p);
        const uint32_t off = cp[topBit - (64u + (uint32_t)p)];
        f(lds_f4(L.pos, off), lds_f4(L.vel, off));
        ++trips;
    }
    uint64_t mm = ((uint64_t)m1 << 32) | (uint64_t)m0;
    while (mm) {
        const int p = 63 - __clzll((long long)mm);
        mm &= ~(1ull << p);
        const uint32_t off = cp[topBit - (uint32_t)p];
        f(lds_f4(L.pos, off), lds_f4(L.vel, off));
        ++trips;
    }
    return trips;
}

// A sub-box of a tile: cells [x0,x0+tx) x [y0,y0+ty) x [z0,z0+tz) (absolute), plus where it sits in
// the tile's LDS cellStart copy (row pitch csW, csHY rows per z layer, offset csX/csY/csZ in cells).
struct SliceGeo {
    int x0, y0, z0, tx, ty, tz;
    int csW, csHY, csX, csY, csZ;
};

// Targets the fast path does not take (cell list overflow, sweep-3 displacement beyond the
// mask's slack, slice overflow) are appended to a queue of sorted slots that k_sph_slow
// processes right after this kernel with the exact per-particle gather (sph_gather_one).
struct SlowQueue {
    uint32_t* count;
    uint32_t* slots;
};
__device__ __forceinline__ void slow_push(const SlowQueue& q, bool pred, uint32_t s) {
    const unsigned long long m = __ballot(pred);
    if (!pred) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(q.count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    q.slots[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = s;
}

__global__ __launch_bounds__(kBlock) void k_sph_slow(SimK k, StateIn in, StateOut out, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ cellStart, SlowQueue q) {
    const uint32_t n = *q.count;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        sph_gather_one(k, in, out, order, cellStart, (int)q.slots[i]);
}

// One sub-box of a tile.  Uniform control flow up to the target loop; contains __syncthreads().
// When csInLds, L.cl holds (as uint32_t cs[]) the tile's cellStart copy: cs[r * csW + hx] for
// whole-tile halo row r and halo column hx (the last column = one past the row's end).
template <bool STAMP, class CFG>
__device__ __forceinline__ void tile_slice(TileLdsT<CFG>& L, const SimK& k, int dbg, const SliceGeo& G, bool csInLds,
                                           const StateIn& in, const StateOut& out, const uint32_t* __restrict__ order,
                                           const uint32_t* __restrict__ cellStart, const SlowQueue& slowq, unsigned long long* st) {
    constexpr int kTileThreads = CFG::kThreads, kMaxCand = CFG::kMaxCand, kListPool = CFG::kListPool;
    const int tid = threadIdx.x;
    const int x0 = G.x0, y0 = G.y0, z0 = G.z0, tx = G.tx, ty = G.ty, tz = G.tz;
    const int HX = tx + 2, HY = ty + 2, HZ = tz + 2;
    const int R = HY * HZ;
    const int iHX = fdiv_inv(HX), iHY = fdiv_inv(HY), iTX = fdiv_inv(tx), iTY = fdiv_inv(ty);
    const int xlo = max(x0 - 1, 0), xhi = min(x0 + tx, k.gx - 1);      // staged x range (inclusive)
    unsigned long long c0 = stamp_now<STAMP>();
    const uint32_t* cs = reinterpret_cast<const uint32_t*>(L.cl);

    // cellStart of halo cell (hx, r) of this slice, hx in [0, HX]; cells outside the grid
    // collapse onto the nearest in-grid boundary so that differences give 0 particles.
    auto cs_at = [&](int r, int hx) -> uint32_t {
        const int hz = fdiv(r, iHY), hy = r - hz * HY;
        if (csInLds) return cs[((G.csZ + hz) * G.csHY + G.csY + hy) * G.csW + G.csX + hx];
        const int yy = y0 - 1 + hy, zz = z0 - 1 + hz;
        if (yy < 0 || yy >= k.gy || zz < 0 || zz >= k.gz) return 0u;
        const int xg = min(max(x0 - 1 + hx, xlo), xhi + 1);
        return cellStart[(zz * k.gy + yy) * k.gx + xg];
    };

    // ---- 1. halo rows: global ranges and LDS offsets (wave 0 scans <= 64 rows) ----
    if (tid < 64) {
        uint32_t gs = 0, cnt = 0;
        if (tid < R) {
            gs = cs_at(tid, 0);
            cnt = cs_at(tid, HX) - gs;
        }
        const uint32_t inc = wave_incl_scan(cnt);
        if (tid < R) { L.rowG[tid] = gs; L.rowL[tid] = inc - cnt; }
        if (tid == R - 1) L.rowL[R] = inc;
    }
    __syncthreads();
    const uint32_t nC = L.rowL[R];
    if (STAMP && tid == 0) { stamp_add<STAMP>(st, TS_SLICES, 1); stamp_add<STAMP>(st, TS_CANDIDATES, nC); }
    if (nC > (uint32_t)kMaxCand || (dbg & 4)) {
        // ---- slice overflow: every target of this slice goes to the slow queue ----
        if (STAMP && tid == 0) stamp_add<STAMP>(st, TS_OVERFLOW_SLICES, 1);
        for (int ir = 0; ir < ty * tz; ++ir) {
            const int iz = fdiv(ir, iTY), iy = ir - iz * ty;
            const int yy = y0 + iy, zz = z0 + iz;
            if (yy >= k.gy || zz >= k.gz || x0 >= k.gx) continue;
            const int base = (zz * k.gy + yy) * k.gx;
            const int xe = min(x0 + tx, k.gx);
            const uint32_t gs = cellStart[base + x0], ge = cellStart[base + xe];
            for (uint32_t s0 = gs; s0 < ge; s0 += kTileThreads) slow_push(slowq, s0 + tid < ge, s0 + tid);
        }
        return;
    }

    // ---- 2. per-cell LDS offsets of the halo box ----
    const int nHalo = HX * R;
    for (int ci = tid; ci < nHalo; ci += kTileThreads) {
        const int r = fdiv(ci, iHX), hx = ci - r * HX;
        L.cellOff[ci] = (uint16_t)(L.rowL[r] + (cs_at(r, hx) - L.rowG[r]));
    }
    if (tid == 0) { L.cellOff[nHalo] = (uint16_t)nC; L.cellOff[nHalo + 1] = (uint16_t)nC; }
    if (STAMP && tid == 0) { const unsigned long long c1 = stamp_now<STAMP>(); stamp_add<STAMP>(st, TS_PROLOGUE, c1 - c0); c0 = c1; }
    // ---- 3. stage particles: LDS slot i <- sorted slot rowG[r] + (i - rowL[r]); four
    //         independent gathers per thread in flight ----
    {
        constexpr int U = (kMaxCand + kTileThreads - 1) / kTileThreads;
        uint32_t src[U], slot[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            slot[u] = (uint32_t)tid + (uint32_t)u * kTileThreads;
            on[u] = slot[u] < nC;
            const uint32_t i = on[u] ? slot[u] : 0u;
            const int r = upper_row(L.rowL, R + 1, i);
            src[u] = on[u] ? order[L.rowG[r] + (i - L.rowL[r])] : 0u;
        }
        float4 P[U], V[U];
        float2 RP[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (on[u]) { P[u] = in.pos[src[u]]; V[u] = in.vel[src[u]]; RP[u] = in.rp[src[u]]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (on[u]) {
                L.pos[slot[u]] = make_float4(P[u].x, P[u].y, P[u].z, RP[u].x > 0.0f ? 1.0f / RP[u].x : 0.0f);
                L.vel[slot[u]] = make_float4(V[u].x, V[u].y, V[u].z, RP[u].y);
            }
        }
    }
    __syncthreads();      // cs[] (aliased on L.cl) is dead from here on
    if (STAMP && tid == 0) { const unsigned long long c1 = stamp_now<STAMP>(); stamp_add<STAMP>(st, TS_STAGE, c1 - c0); c0 = c1; }

    // ---- 4. target prefix over interior rows (wave 0) + per-cell candidate lists ----
    const int IR = ty * tz;
    const int nCells = tx * IR;
    // list stride: an odd number of 16-B slots (8 entries each) so that the b128 index reads of
    // neighbouring cells fall on different bank groups
    int CLs = min(kListPool / max(nCells, 1), 1016) & ~7;
    if (!((CLs >> 3) & 1)) CLs -= 8;
    const int cap = CLs;                                       // lists longer than one 96-bit mask are processed in chunks
    if (tid < 64) {
        uint32_t cnt = 0;
        if (tid < IR) {
            const int iz = fdiv(tid, iTY), iy = tid - iz * ty;
            const int r = (iz + 1) * HY + (iy + 1);
            cnt = (uint32_t)L.cellOff[r * HX + 1 + tx] - (uint32_t)L.cellOff[r * HX + 1];
        }
        const uint32_t inc = wave_incl_scan(cnt);
        if (tid < IR) L.tgtStart[tid] = inc - cnt;
        if (tid == IR - 1) L.tgtStart[IR] = inc;
    }
    const unsigned long long l0 = stamp_now<STAMP>();
    if (STAMP && tid == 0) stamp_add<STAMP>(st, TS_L_TGT, l0 - c0);
    {   // Wave-cooperative list build.  A wave takes one interior (y,z) row of cells at a time;
        // lane = (cell of the row) * 8 + j copies entry j, j+8, ... of each of the 9 runs, then
        // pads the cell's list with the sentinel up to the stride.  No cross-lane traffic: the
        // 8 lanes of a cell read the same cellOff words (LDS broadcast) and keep the same prefix.
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int kWaves = kTileThreads / 64;
        const int cpw = 8;                                           // cells per wave step
        const int j = lane & 7;
        const uint16_t sent = (uint16_t)(kMaxCand * 16);
        for (int c0 = wave * cpw; c0 < nCells; c0 += kWaves * cpw) {
            const int c = c0 + (lane >> 3);
            if (c >= nCells) continue;
            const int cq = fdiv(c, iTX), ix = c - cq * tx, iz = fdiv(cq, iTY), iy = cq - iz * ty;
            const int off0 = (iz * HY + iy) * HX + ix;               // halo cell (hx-1, hy-1, hz-1)
            int qs[9], ln[9];
#pragma unroll
            for (int run = 0; run < 9; ++run) {
                const int off = off0 + ((run / 3) * HY + (run % 3)) * HX;
                qs[run] = L.cellOff[off];
                ln[run] = (int)L.cellOff[off + 3] - qs[run];
            }
            const int mid = (int)L.cellOff[off0 + (HY + 1) * HX + 1] - qs[4];
            const int base = c * CLs;
            int pre = 0, selfK = 0;
            uint16_t* dst = &L.cl[base];
#pragma unroll
            for (int run = 0; run < 9; ++run) {
                if (run == 4) selfK = pre + mid;
                // entries j and j+8 of the run as straight-line predicated stores (runs of 3 cells
                // rarely hold more than 16 particles); anything longer in a plain scalar loop
                const int n = ln[run], q0 = qs[run];
                if (j < n && pre + j < cap) dst[pre + j] = (uint16_t)((q0 + j) * 16);
                if (j + 8 < n && pre + j + 8 < cap) dst[pre + j + 8] = (uint16_t)((q0 + j + 8) * 16);
                if (n > 16) {
#pragma clang loop vectorize(disable) unroll(disable)
                    for (int e = j + 16; e < n; e += 8)
                        if (pre + e < cap) dst[pre + e] = (uint16_t)((q0 + e) * 16);
                }
                pre += n;
            }
            {   // sentinel padding up to the stride, one 16-B slot (8 entries) per lane where possible
                const int p0 = min(pre, cap);
                const int pa = (p0 + 7) & ~7;                        // first whole slot
                if (p0 + j < pa) dst[p0 + j] = sent;                 // ragged head
                const uint32_t sv = (uint32_t)sent * 0x10001u;
#pragma clang loop vectorize(disable) unroll(disable)
                for (int p = pa + 8 * j; p < CLs; p += 64) *reinterpret_cast<uint4*>(dst + p) = make_uint4(sv, sv, sv, sv);
            }
            if (j == 0) {
                L.clSelf[c] = (uint16_t)selfK;
                L.clLen[c] = (pre <= cap && !(dbg & 1)) ? (uint16_t)pre : (uint16_t)0xFFFFu;
            }
        }
    }
    if (STAMP && (tid & 63) == 0) stamp_add<STAMP>(st, TS_L_BUILD, stamp_now<STAMP>() - l0);
    __syncthreads();
    const uint32_t nT = L.tgtStart[IR];
    if (STAMP && tid == 0) { const unsigned long long c1 = stamp_now<STAMP>(); stamp_add<STAMP>(st, TS_LISTS, c1 - c0); c0 = c1; stamp_add<STAMP>(st, TS_TARGETS, nT); }
    const bool wlead = (tid & 63) == 0;

    // ---- 5. one thread per target particle; no barrier below this line.  A wave whose 64
    //         slots are all past the last target skips the round; inside a live wave the lanes
    //         past the end redo the last target (uniform control flow) and skip the store. ----
    for (uint32_t t0 = 0; t0 < nT; t0 += kTileThreads) {
        if (t0 + (uint32_t)(tid & ~63) >= nT) continue;
        const unsigned long long w0 = stamp_now<STAMP>();
        const bool valid = (t0 + tid) < nT;
        const uint32_t t = valid ? (t0 + tid) : (nT - 1);
        const int ir = upper_row(L.tgtStart, IR + 1, t);
        const int iz = fdiv(ir, iTY), iy = ir - iz * ty;
        const int hy = iy + 1, hz = iz + 1;
        const int r = hz * HY + hy;
        const uint32_t li = (uint32_t)L.cellOff[r * HX + 1] + (t - L.tgtStart[ir]);
        const int s = (int)(L.rowG[r] + (li - L.rowL[r]));
        const uint32_t src = order[s];
        const float4 P = in.pos[src], V = in.vel[src];
        const float foamIn = in.foam[src];
        const float4 LP = L.pos[li], LV = L.vel[li];
        Own o;
        own_reset(o);
        o.px = LP.x; o.py = LP.y; o.pz = LP.z; o.vx = LV.x; o.vy = LV.y; o.vz = LV.z; o.rho = 0.0f; o.prs = LV.w;
        const int cx = cell_axis(o.px, k.gminx, k.cellSize, k.gx);
        const int hx = cx - (x0 - 1);
        const int c = (iz * ty + iy) * tx + (hx - 1);
        const uint32_t len = L.clLen[c];
        const bool useMask = (len != 0xFFFFu);
        const float ex = o.px, ey = o.py, ez = o.pz;             // entry position
        // mask radius: h + slack, slack covers this substep's own displacement
        const float slack = fmaf((fabsf(o.vx) + fabsf(o.vy) + fabsf(o.vz)) * k.dt, 1.25f, 0.05f * k.h);
        const float hl = k.h + slack;
        const float h2list = hl * hl;

        // ---- wave-uniform plan: groups of 8 candidates in the longest list of this wave ----
        const int groups = (__builtin_amdgcn_readfirstlane((int)wave_max_u32(useMask ? len : 0u)) + 7) >> 3;
        const uint16_t* clp = &L.cl[c * CLs];
        const bool dense = groups > 12;                            // more candidates than one 96-bit mask holds
        uint32_t m0 = 0, m1 = 0, m2 = 0;
        // sweep 1: density over every candidate; a short list also yields the neighbour mask
        if (!dense) {
            scan_chunk<true>(k, L, o, clp, 0, groups, o.px, o.py, o.pz, h2list, m0, m1, m2);
        } else {
            for (int g0 = 0; g0 < groups; g0 += 12) {
                uint32_t t0, t1, t2;
                scan_chunk<true>(k, L, o, clp, g0, min(g0 + 12, groups), o.px, o.py, o.pz, k.h2, t0, t1, t2);
            }
        }
        const unsigned long long w1 = stamp_now<STAMP>();
        if (STAMP && wlead) { stamp_add<STAMP>(st, TS_SCAN, w1 - w0); stamp_add<STAMP>(st, TS_WAVEROUNDS, 1); stamp_add<STAMP>(st, TS_SCANGROUPS, groups); }

        const uint32_t flags = fbits(P.w), id = fbits(V.w);
        if (flags & F_HALO) {                                      // neighbour rank's particle: candidate only
            if (valid) out.pos[s] = P;
            continue;
        }
        if (flags & F_GHOST1) {                                    // SPHFluid.comp:72-83
            if (valid) {
                float gvx = V.x, gvy = V.y, gvz = V.z, grho = in.rp[src].x, gprs = LV.w;
                if (!(flags & F_INACTIVE)) { gvx = gvy = gvz = 0.0f; grho = k.rho0; gprs = 0.0f; }
                out.pos[s] = P;
                out.vel[s] = make_float4(gvx, gvy, gvz, V.w);
                out.rp[s] = make_float2(grho, gprs);
                out.foam[s] = foamIn;
                if (out.aos) { if (!(flags & F_INACTIVE)) aos_write_active_ghost(out.aos, id - out.idBase, k.rho0); }
                else out.acc[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            continue;
        }
        if (!useMask) {                                            // list does not fit the pool: exact path in k_sph_slow
            if (STAMP && valid) stamp_add<STAMP>(st, TS_SLOW_LANES, 1);
            slow_push(slowq, valid, (uint32_t)s);
            continue;
        }
        finish_density(k, o);
        const uint32_t ks = (uint32_t)L.clSelf[c] + (li - (uint32_t)L.cellOff[r * HX + hx]);   // own position in the list
        auto drop_self = [&](int g0, int g1, uint32_t& a0, uint32_t& a1, uint32_t& a2) {
            const uint32_t lo = (uint32_t)(8 * g0), hi = (uint32_t)(8 * g1);
            if (ks >= lo && ks < hi) {
                const uint32_t B = (hi - lo - 1u) - (ks - lo);
                const uint32_t bit = 1u << (B & 31u);
                const uint32_t w = B >> 5;
                a0 &= ~(w == 0 ? bit : 0u); a1 &= ~(w == 1 ? bit : 0u); a2 &= ~(w == 2 ? bit : 0u);
            }
        };
        auto force_at = [&](const float4& J, const float4& JV) {
            if (J.w > 0.0f) pair_force_pre(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, JV.w, J.w);
        };
        auto xsph_at = [&](const float4& J, const float4& JV) {
            if (J.w > 0.0f) pair_xsph_pre(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, J.w);
        };
        int trips2 = 0;
        // sweep 2: forces, ascending candidate order
        if (!dense) {
            drop_self(0, groups, m0, m1, m2);
            trips2 = walk_chunk(L, clp, 0, groups, m0, m1, m2, force_at);
        } else {
            for (int g0 = 0; g0 < groups; g0 += 12) {
                const int g1 = min(g0 + 12, groups);
                uint32_t t0, t1, t2;
                scan_chunk<false>(k, L, o, clp, g0, g1, ex, ey, ez, k.h2hi, t0, t1, t2);
                drop_self(g0, g1, t0, t1, t2);
                trips2 += walk_chunk(L, clp, g0, g1, t0, t1, t2, force_at);
            }
        }
        integrate(k, o);
        const unsigned long long w2 = stamp_now<STAMP>();
        // sweep 3: XSPH with the updated own state.  The sweep-1 mask stays valid only if the
        // displacement stayed inside the slack it was built with; otherwise (and for long
        // lists) the candidates are re-scanned from LDS against the new position.
        const float mx = o.px - ex, my = o.py - ey, mz = o.pz - ez;
        const float moved2 = dot3(mx, my, mz, mx, my, mz);
        const float lim = 0.98f * slack;
        const bool maskOk = !dense && (moved2 <= lim * lim) && !(dbg & 2);
        if (!__any(!maskOk)) {                                     // uniform over the lanes still active
            (void)walk_chunk(L, clp, 0, groups, m0, m1, m2, xsph_at);
        } else {
            if (STAMP && valid && !maskOk) stamp_add<STAMP>(st, TS_RESCAN_LANES, 1);
            for (int g0 = 0; g0 < groups; g0 += 12) {
                const int g1 = min(g0 + 12, groups);
                uint32_t t0, t1, t2;
                scan_chunk<false>(k, L, o, clp, g0, g1, o.px, o.py, o.pz, k.h2, t0, t1, t2);
                drop_self(g0, g1, t0, t1, t2);
                (void)walk_chunk(L, clp, g0, g1, t0, t1, t2, xsph_at);
            }
        }
        const unsigned long long w3 = stamp_now<STAMP>();
        const float foamOut = finish_particle(k, o, foamIn);
        if (valid) store_particle(k, out, s, flags, id, o, foamOut);
        if (STAMP) {
            const unsigned long long w4 = stamp_now<STAMP>();
            const uint32_t tmax = wave_max_u32((uint32_t)trips2);
            if (wlead) {
                stamp_add<STAMP>(st, TS_SWEEP2, w2 - w1); stamp_add<STAMP>(st, TS_SWEEP3, w3 - w2); stamp_add<STAMP>(st, TS_EPILOGUE, w4 - w3);
                stamp_add<STAMP>(st, TS_WALK2MAX, tmax);
            }
            if (valid) stamp_add<STAMP>(st, TS_WALK2SUM, (unsigned long long)trips2);
        }
    }
}

template <bool STAMP, class CFG>
__global__ __launch_bounds__(CFG::kThreads, CFG::kMinWaves) void k_sph_tile(SimK k, TileGeom g, StateIn in, StateOut out,
                                                            const uint32_t* __restrict__ order,
                                                            const uint32_t* __restrict__ cellStart, SlowQueue slowq,
                                                            unsigned long long* stAll) {
    constexpr int kTileThreads = CFG::kThreads, kMaxCand = CFG::kMaxCand;
    __shared__ TileLdsT<CFG> L;
    const int tid = threadIdx.x;
    const unsigned long long k0 = stamp_now<STAMP>();
    // XCD-aware tile mapping: blocks b and b+8 share an XCD (round-robin dispatch), so give
    // each XCD one contiguous chunk of the tile order (neighbouring tiles share halo rows in
    // that XCD's L2).  Pure speed; any mapping is correct.
    int tile;
    {
        const int b = blockIdx.x, per = (g.numTiles + 7) >> 3;
        tile = (b & 7) * per + (b >> 3);
        if (tile >= g.numTiles) return;
    }
    unsigned long long* st = STAMP ? (stAll + (size_t)tile * TS_COUNT) : nullptr;
    const int tX = tile % g.ntx, tY = (tile / g.ntx) % g.nty, tZ = tile / (g.ntx * g.nty);
    const int x0 = tX * g.tx, y0 = tY * g.ty, z0 = tZ * g.tz;
    const int HX = g.tx + 2, HY = g.ty + 2, HZ = g.tz + 2;
    const int xlo = max(x0 - 1, 0), xhi = min(x0 + g.tx, k.gx - 1);

    // ---- one global round trip: cellStart of every halo cell of the whole tile (plus each
    //      row's end) into LDS, aliased on the list pool ----
    uint32_t* cs = reinterpret_cast<uint32_t*>(L.cl);
    const int W = HX + 1;
    const int iW = fdiv_inv(W), iHYt = fdiv_inv(HY), iTXt = fdiv_inv(g.tx), iTYt = fdiv_inv(g.ty);
    for (int ci = tid; ci < W * HY * HZ; ci += kTileThreads) {
        const int r = fdiv(ci, iW), hx = ci - r * W;
        const int hz = fdiv(r, iHYt), hy = r - hz * HY;
        const int yy = y0 - 1 + hy, zz = z0 - 1 + hz;
        uint32_t v = 0;
        if (yy >= 0 && yy < k.gy && zz >= 0 && zz < k.gz) {
            const int xg = min(max(x0 - 1 + hx, xlo), xhi + 1);
            v = cellStart[(zz * k.gy + yy) * k.gx + xg];
        }
        cs[ci] = v;
    }
    if (tid == 0) {
        L.pos[kMaxCand] = make_float4(1e18f, 1e18f, 1e18f, 0.0f);   // sentinel: never within any radius
        L.vel[kMaxCand] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        L.maxLen = 0;
        L.tileOver = 0;
    }
    __syncthreads();
    if (tid < 64) {   // staged particles of the whole tile
        const uint32_t cnt = (tid < HY * HZ) ? (cs[tid * W + HX] - cs[tid * W]) : 0u;
        const uint32_t total = (uint32_t)__shfl((int)wave_incl_scan(cnt), 63, 64);
        if (tid == 0 && total > (uint32_t)kMaxCand) L.tileOver = 1;
    }
    {   // longest candidate list among the tile's cells (from the cellStart copy)
        int mx = 0;
        for (int c = tid; c < g.tx * g.ty * g.tz; c += kTileThreads) {
            const int cq = fdiv(c, iTXt), ix = c - cq * g.tx, iz = fdiv(cq, iTYt), iy = cq - iz * g.ty;
            int len = 0;
            for (int dz = 0; dz < 3; ++dz)
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = (iz + dz) * HY + iy + dy;
                    len += (int)(cs[r * W + ix + 3] - cs[r * W + ix]);
                }
            mx = max(mx, len);
        }
        mx = (int)wave_max_u32((uint32_t)mx);
        if ((tid & 63) == 0 && mx > 0) atomicMax(&L.maxLen, mx);
    }
    __syncthreads();
    // ---- plan: the largest sub-box (halved in z, then y, then x) such that every sub-box of
    //      the tile fits: staged particles <= kMaxCand, cells <= kMaxCells, lists fit the pool.
    //      The common case (whole tile) needs one pass; dense tiles iterate, all threads in step. ----
    int sx = g.tx, sy = g.ty, sz = g.tz;
    {
        const int needStride = max(72, ((L.maxLen + 7) & ~7) + 16);
        const int maxCellsPerBox = max(1, min(kMaxCells, CFG::kListPool / needStride));
        const bool whole = !L.tileOver && (sx * sy * sz <= maxCellsPerBox);   // the common case: no further barrier
        if (!whole) for (;;) {
            while (sx * sy * sz > maxCellsPerBox) {
                if (sz > 1) sz = (sz + 1) >> 1; else if (sy > 1) sy = (sy + 1) >> 1; else if (sx > 1) sx = (sx + 1) >> 1; else break;
            }
            __syncthreads();
            if (tid == 0) L.sliceTz = 0;
            __syncthreads();
            const int nbx = (g.tx + sx - 1) / sx, nby = (g.ty + sy - 1) / sy, nbz = (g.tz + sz - 1) / sz;
            bool over = false;
            for (int b = tid; b < nbx * nby * nbz; b += kTileThreads) {
                const int bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
                const int xs = bx * sx, ys = by * sy, zs = bz * sz;
                const int ex = min(xs + sx, g.tx), ey = min(ys + sy, g.ty), ez = min(zs + sz, g.tz);
                uint32_t sum = 0;
                for (int hz = zs; hz < ez + 2; ++hz)
                    for (int hy = ys; hy < ey + 2; ++hy) sum += cs[(hz * HY + hy) * W + ex + 2] - cs[(hz * HY + hy) * W + xs];
                over = over || (sum > (uint32_t)kMaxCand);
            }
            if (over) L.sliceTz = 1;                                   // benign race: any writer writes 1
            __syncthreads();
            const bool bad = L.sliceTz != 0;
            if (!bad) break;
            if (sz > 1) sz = (sz + 1) >> 1; else if (sy > 1) sy = (sy + 1) >> 1; else if (sx > 1) sx = (sx + 1) >> 1;
            else break;                                                // single cells still overflow: those go to the slow queue
        }
    }
    if (STAMP && tid == 0) { stamp_add<STAMP>(st, TS_PROLOGUE, stamp_now<STAMP>() - k0); stamp_add<STAMP>(st, TS_TILES, 1); }
    bool first = true;
    for (int zs = 0; zs < g.tz; zs += sz)
        for (int ys = 0; ys < g.ty; ys += sy)
            for (int xs = 0; xs < g.tx; xs += sx) {
                SliceGeo G;
                G.x0 = x0 + xs; G.y0 = y0 + ys; G.z0 = z0 + zs;
                G.tx = min(sx, g.tx - xs); G.ty = min(sy, g.ty - ys); G.tz = min(sz, g.tz - zs);
                G.csW = W; G.csHY = HY; G.csX = xs; G.csY = ys; G.csZ = zs;
                if (G.x0 >= k.gx || G.y0 >= k.gy || G.z0 >= k.gz) continue;   // sub-box outside the grid (clipped tile)
                // the LDS cellStart copy is overwritten by the first sub-box's lists; later ones
                // (dense tiles only) read cellStart from global memory again
                tile_slice<STAMP, CFG>(L, k, g.debugFlags, G, first, in, out, order, cellStart, slowq, st);
                first = false;
                __syncthreads();
            }
    if (STAMP && tid == 0) stamp_add<STAMP>(st, TS_TOTAL, stamp_now<STAMP>() - k0);
}

inline int tile_count(const TilePlan& plan, const SimK& k) {
    return ((k.gx + plan.tx - 1) / plan.tx) * ((k.gy + plan.ty - 1) / plan.ty) * ((k.gz + plan.tz - 1) / plan.tz);
}

// Host side: pick the tile geometry and launch.  Returns 0 or -(hipError_t).
template <class TimedFactory>
inline int tile_launch(TilePlan& plan, hipStream_t stream, const SimK& k, const StateIn& in, const StateOut& out,
                       const uint32_t* order, const uint32_t* cellStart, const SlowQueue& slowq, unsigned long long* stamps,
                       TimedFactory&& timed) {
    TileGeom g;
    g.tx = plan.tx; g.ty = plan.ty; g.tz = plan.tz;
    g.debugFlags = plan.debugFlags;
    g.ntx = (k.gx + g.tx - 1) / g.tx; g.nty = (k.gy + g.ty - 1) / g.ty; g.ntz = (k.gz + g.tz - 1) / g.tz;
    g.numTiles = g.ntx * g.nty * g.ntz;
    if ((g.ty + 2) * (g.tz + 2) > kMaxRows || (g.tx + 2) * (g.ty + 2) * (g.tz + 2) > kMaxHaloCells ||
        g.tx * g.ty > kMaxCells) return -(int)hipErrorInvalidValue;
    const int per = (g.numTiles + 7) / 8;
    hipError_t e = hipMemsetAsync(slowq.count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return -(int)e;
    {
        auto t = timed(SPH_K_SPH);
        const bool stamp = (plan.debugFlags & 8) && stamps;
        unsigned long long* sp = stamp ? stamps : nullptr;
#define SPH_LAUNCH_TILE(CFG)                                                                                        \
        if (stamp) hipLaunchKernelGGL((k_sph_tile<true, CFG>), dim3(per * 8), dim3(CFG::kThreads), 0, stream, k, g, in, out, order, cellStart, slowq, sp); \
        else hipLaunchKernelGGL((k_sph_tile<false, CFG>), dim3(per * 8), dim3(CFG::kThreads), 0, stream, k, g, in, out, order, cellStart, slowq, sp);
        if (plan.config == 1) { SPH_LAUNCH_TILE(TileCfgB) }
        else if (plan.config == 3) { SPH_LAUNCH_TILE(TileCfgD) }
        else if (plan.config == 4) { SPH_LAUNCH_TILE(TileCfgE) }
        else if (plan.config == 2) { SPH_LAUNCH_TILE(TileCfgC) }
        else { SPH_LAUNCH_TILE(TileCfgA) }
#undef SPH_LAUNCH_TILE
    }
    {
        auto t = timed(SPH_K_OTHER);
        hipLaunchKernelGGL(k_sph_slow, dim3(512), dim3(kBlock), 0, stream, k, in, out, order, cellStart, slowq);
    }
    e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace sph
