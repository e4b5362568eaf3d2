// sph_kernels.h -- HIP kernels of the substep pipeline (gfx950, wave64).
//
// Pipeline of one sph_dispatch() (reference: SPHFluid3D.cpp:431-509):
//   k_bin      cell index + per-cell histogram      (BuildGrid.comp:21-31, without the list push)
//   k_scan_*   exclusive scan of the histogram, clears it for the next substep (ClearGrid.comp)
//   k_scatter  counting-sort scatter into cell-contiguous slots (replaces cellHead/particleNext)
//   k_rank     canonical order inside a cell: ascending particle id (makes fp32 sums reproducible)
//   k_sph_*    (sph_pass.h) 27-cell density -> pressure -> forces -> integrate -> XSPH -> cap -> foam, with
//              OBBConstraints.comp fused into the epilogue (legal: neighbours are read from the
//              entry snapshot, the own record is private to the thread)
//   k_writeback  update of the public 80-byte AoS in ORIGINAL particle order
//
// Internal state ("sorted SoA", double buffered; slot s of substep n's output is the s-th
// particle in (cell, id) order of substep n):
//   pos[s] = (x, y, z, flag bits)   vel[s] = (vx, vy, vz, particle id bits)
//   rp[s]  = (density, pressure)    foam[s] = padA           acc[s] = (ax, ay, az, 0)
#pragma once
#include "sph_device.h"

namespace sph {

constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __uint_as_float(u); }
}  // namespace sph
#include "sph_shapes_ext.h"
namespace sph {

// ---- AoS -> internal state (after upload / reset) ------------------------------------
__global__ __launch_bounds__(kBlock) void k_import(const SphParticle* __restrict__ aos, float4* __restrict__ pos,
                                                   float4* __restrict__ vel, float2* __restrict__ rp,
                                                   float* __restrict__ foam, uint32_t idBase, int n) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4* rec = reinterpret_cast<const float4*>(aos + i);
    float4 p = rec[0], v = rec[1], d = rec[3];
    int4 fl = *reinterpret_cast<const int4*>(rec + 4);
    uint32_t flags = (fl.x == 1 ? F_GHOST1 : 0u) | (fl.x != 0 ? F_GHOSTNZ : 0u) | (fl.y == 0 ? F_INACTIVE : 0u);
    pos[i] = make_float4(p.x, p.y, p.z, bitsf(flags));
    vel[i] = make_float4(v.x, v.y, v.z, bitsf(idBase + (uint32_t)i));
    rp[i] = make_float2(d.x, d.y);
    foam[i] = d.z;
}

// ---- exclusive scan over the histogram: tiles -------------------------------------------
#ifndef SPH_SCAN_ITEMS
#define SPH_SCAN_ITEMS 8     // (16: scan 16.8 us at config 3, 8: 14.4 us -- twice the blocks for the same bytes; gpurun_out/r05 A/B, profiles/r05_grid_build.txt)
#endif
constexpr int kScanItems = SPH_SCAN_ITEMS;           // 256 threads x 8 = 2048 cells per block
constexpr int kScanTile = kBlock * kScanItems;

// ---- BuildGrid.comp:21-31: cell of every particle + histogram -------------------------
// The state is (nearly) cell-sorted from the previous substep, so equal cells sit in adjacent
// lanes: each run of equal cells inside a wave issues ONE returning atomic (run leader) and
// hands out consecutive slots, instead of one contended atomic per particle.
// Round 5: (cell, slot) leave as ONE 8-byte key.  (Also tried in round 5: this pass adding its particles to the sum of the scan tile their cell lies in,
// to save k_scan_reduce's read of the histogram.  One global atomic per wave and tile: k_bin 31 -> 204 us; aggregated per block in LDS first, one or two
// global atomics per block: 31 -> 77 us.  Consecutive blocks add to the SAME word at the same time, and a device-scope atomic on one address costs
// hundreds of nanoseconds on this chip (eight L2s that are not coherent with each other): profiles/r05_grid_build.txt.)
__global__ __launch_bounds__(kBlock) void k_bin(SimK k, const float4* __restrict__ pos, uint2* __restrict__ binKey,
                                                uint32_t* __restrict__ cellCount, int n, const uint32_t* __restrict__ slotsInUse) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // z-slab mode without host round trips: n is only a launch bound, the slots that hold data are counted on the device
    if (slotsInUse && (uint32_t)(blockIdx.x * kBlock) >= *slotsInUse) return;      // whole block beyond the data (k_scatter skips it too)
    const bool inRange = i < n;
    const bool used = inRange && (!slotsInUse || (uint32_t)i < *slotsInUse);
    uint32_t cell = 0xFFFFFFFFu;              // also the key of dead / unused slots (z-slab mode): they get no slot
    if (used) {
        const float4 p = pos[i];
        const int cx = cell_axis(p.x, k.gminx, k.cellSize, k.gx);
        const int cy = cell_axis(p.y, k.gminy, k.cellSize, k.gy);
        const int cz = cell_z_local(k, p.z);
        if (!(fbits(p.w) & F_DEAD)) cell = (uint32_t)((cz * k.gy + cy) * k.gx + cx);   // flatten(), BuildGrid.comp:19
    }
    const unsigned long long upto = (2ull << lane) - 1ull;              // bits 0..lane
    const bool valid = cell != 0xFFFFFFFFu;
    const uint32_t prev = (uint32_t)__shfl_up((int)cell, 1, 64);
    const bool head = (lane == 0) || (cell != prev);
    const unsigned long long heads = __ballot(head);
    const int startLane = 63 - __clzll((long long)(heads & upto));
    const unsigned long long above = heads & ~upto;
    const int endLane = above ? (__ffsll((long long)above) - 1) : 64;
    uint32_t base = 0;
    if (head && valid) base = atomicAdd(&cellCount[cell], (uint32_t)(endLane - lane));
    base = (uint32_t)__shfl((int)base, startLane, 64);
    if (inRange) binKey[i] = make_uint2(cell, base + (uint32_t)(lane - startLane));
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads = 4 waves); returns the
// exclusive prefix and, through total, the block sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* sm, uint32_t& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    uint32_t w0 = sm[0], w1 = sm[1], w2 = sm[2], w3 = sm[3];
    uint32_t base = (w > 0 ? w0 : 0u) + (w > 1 ? w1 : 0u) + (w > 2 ? w2 : 0u);
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

// Each thread owns kScanItems CONSECUTIVE cells (16-byte loads), so a tile of 256 x kScanItems cells needs one block-wide scan.
__device__ __forceinline__ void scan_load16(const uint32_t* __restrict__ cnt, int c0, int numCells, uint32_t (&v)[kScanItems]) {
    if (c0 + kScanItems <= numCells) {
        const uint4* p = reinterpret_cast<const uint4*>(cnt + c0);
#pragma unroll
        for (int j = 0; j < kScanItems / 4; ++j) { const uint4 q = p[j]; v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w; }
    } else {
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) v[j] = (c0 + j < numCells) ? cnt[c0 + j] : 0u;
    }
}

__global__ __launch_bounds__(kBlock) void k_scan_reduce(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ blockSums, int numCells) {
    __shared__ uint32_t sm[4];
    const int c0 = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    scan_load16(cnt, c0, numCells, v);
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) s += v[j];
    uint32_t total;
    (void)block_excl_scan(s, sm, total);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
}

// single block: exclusive scan of the per-tile sums (any count, chunks of 256 with carry); grids beyond kScanFusedBlocks tiles only
__global__ __launch_bounds__(kBlock) void k_scan_blocksums(uint32_t* __restrict__ blockSums, int numBlocks) {
    __shared__ uint32_t sm[4];
    uint32_t carry = 0;
    for (int base = 0; base < numBlocks; base += kBlock) {
        int i = base + threadIdx.x;
        uint32_t v = (i < numBlocks) ? blockSums[i] : 0u;
        uint32_t total;
        uint32_t ex = block_excl_scan(v, sm, total);
        if (i < numBlocks) blockSums[i] = carry + ex;
        carry += total;
    }
}

// per-block scan + block offset -> cellStart[0..numCells]; clears the histogram (ClearGrid)
// rawSums (round 4, grids of up to kScanFusedBlocks tiles): blockSums still holds k_scan_reduce's per-tile totals and every block adds up the
// totals in front of its own (a few hundred words out of L2), which saves the single-block k_scan_blocksums launch in between.
constexpr int kScanFusedBlocks = 1024;
__global__ __launch_bounds__(kBlock) void k_scan_apply(uint32_t* __restrict__ cnt, const uint32_t* __restrict__ blockSums,
                                                       uint32_t* __restrict__ cellStart, int numCells, uint32_t nTotal, int rawSums) {
    __shared__ uint32_t sm[4];
    uint32_t before = 0u;
    if (rawSums) {
        uint32_t part = 0u;
        for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) part += blockSums[i];
        uint32_t tot;
        (void)block_excl_scan(part, sm, tot);
        before = tot;
    } else {
        before = blockSums[blockIdx.x];
    }
    const int c0 = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    scan_load16(cnt, c0, numCells, v);
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) { const uint32_t t = v[j]; v[j] = s; s += t; }       // exclusive prefix inside the thread
    uint32_t total;
    const uint32_t off = before + block_excl_scan(s, sm, total);
    if (c0 + kScanItems <= numCells) {
        uint4* ps = reinterpret_cast<uint4*>(cellStart + c0);
        uint4* pc = reinterpret_cast<uint4*>(cnt + c0);
#pragma unroll
        for (int j = 0; j < kScanItems / 4; ++j) {
            ps[j] = make_uint4(off + v[4 * j], off + v[4 * j + 1], off + v[4 * j + 2], off + v[4 * j + 3]);
            pc[j] = make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) if (c0 + j < numCells) { cellStart[c0 + j] = off + v[j]; cnt[c0 + j] = 0u; }
    }
    // total = live particles (in z-slab mode nTotal, the slot count, also covers dead slots)
    (void)nTotal;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) cellStart[numCells] = before + total;
}

// ---- counting-sort scatter: tmp[slot] = (source index, cell) ------------------------------
// (round 5: the particle id is no longer carried -- reading it cost a 16-byte vel record per particle for 4 bytes; k_rank gathers the ids of a
//  cell's few members itself, out of cache lines it loads anyway -- and the cell rides along instead, which saves k_rank a dependent gather)
__global__ __launch_bounds__(kBlock) void k_scatter(const uint2* __restrict__ binKey, const uint32_t* __restrict__ cellStart,
                                                    uint2* __restrict__ tmp, int n, const uint32_t* __restrict__ slotsInUse) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || (slotsInUse && (uint32_t)i >= *slotsInUse)) return;
    const uint2 key = binKey[i];
    if (key.x == 0xFFFFFFFFu) return;         // dead slot (z-slab mode)
    tmp[cellStart[key.x] + key.y] = make_uint2((uint32_t)i, key.x);
}

// ---- canonical order inside each cell: rank by ascending particle id ---------------------
// (atomic arrival order in k_bin is arbitrary, exactly like BuildGrid.comp's atomicExchange;
// this pass removes that freedom.)  order[s] = source index of the particle in sorted slot s.
// With COPY the pass also writes the physically sorted copy of the entry state that the SPH pass
// reads (sph_pass.h SortedIn: a 32-byte record + 16 bytes of own data per particle; 1/rho is the one correctly rounded division per
// neighbour of the numerics contract, item 9).
template <bool COPY>
__global__ __launch_bounds__(kBlock) void k_rank(const uint2* __restrict__ tmp, const uint32_t* __restrict__ cellStart, uint32_t* __restrict__ order, int n, int numCells,
                                                 const float4* __restrict__ pos, const float4* __restrict__ vel,
                                                 const float2* __restrict__ rp, const float* __restrict__ foam,
                                                 float4* __restrict__ pv, float4* __restrict__ own, int gx, int gy,
                                                 uint32_t* __restrict__ liveOut) {
    const int d = blockIdx.x * kBlock + threadIdx.x;
    // z-slab mode: the sorted output will hold exactly the live particles; their count replaces the slots-in-use count
    // of the exchange (k_bin / k_scatter, which read that count, have finished)
    const uint32_t nLive = cellStart[numCells];                  // (<= n)
    if (d == 0 && liveOut) *liveOut = nLive;
    if ((uint32_t)(blockIdx.x * kBlock) >= nLive) return;        // whole block beyond the live particles
    const bool act = d < n && (uint32_t)d < nLive;               // every lane of a wave stays to the end: the ids cross the lanes below
    const uint2 me = act ? tmp[d] : make_uint2(0u, 0u);          // (source index, cell)
    // everything that depends only on the source index is requested NOW, beside the cell lookups below
    float4 V = make_float4(0.0f, 0.0f, 0.0f, 0.0f), P = V;
    float2 RP = make_float2(0.0f, 0.0f);
    float F = 0.0f;
    if (act) {
        V = vel[me.x];
        if (COPY) { P = pos[me.x]; RP = rp[me.x]; F = foam[me.x]; }
    }
    const uint32_t c = me.y;
    const uint32_t s = act ? cellStart[c] : 0u, e = act ? cellStart[c + 1] : 0u;
    const uint32_t myId = fbits(V.w);
    uint32_t rank = 0;
    // The ids of the cell's other members: a cell's slots are consecutive, i.e. its members are NEIGHBOURING LANES of this wave, and every lane
    // holds its particle's id in V.w already -- a cross-lane read, no memory access.  Only members outside the wave's 64 slots (a cell that
    // straddles the wave's edge, a crowded cell of compressed fluid) are gathered from memory.
    const uint32_t d0 = (uint32_t)d - (uint32_t)(threadIdx.x & 63);                // slot of lane 0
    const uint32_t m = e - s;
    uint32_t mWave = m;
    for (int sh = 32; sh >= 1; sh >>= 1) mWave = max(mWave, (uint32_t)__shfl_xor((int)mWave, sh, 64));
    if (mWave > 1u) {                                                              // wave-uniform
        if (mWave <= 64u) {
            for (uint32_t j = 0; j < mWave; ++j) {                                 // wave-uniform trip count: the shuffle needs every lane
                const uint32_t src = s + j - d0;                                   // (a member in front of the wave wraps to a huge number)
                const uint32_t idq = (uint32_t)__shfl((int)myId, (int)(src & 63u), 64);
                rank += (j < m && src < 64u && idq < myId) ? 1u : 0u;
            }
        } else {                                                                   // a cell larger than a wave somewhere in it: every lane against every lane
            for (uint32_t L = 0; L < 64u; ++L) {
                const uint32_t idq = (uint32_t)__shfl((int)myId, (int)L, 64);
                const uint32_t q = d0 + L;
                rank += (q >= s && q < e && idq < myId) ? 1u : 0u;
            }
        }
        for (uint32_t q = s; q < min(e, d0); ++q) rank += (fbits(vel[tmp[q].x].w) < myId) ? 1u : 0u;             // members in front of the wave's first slot
        for (uint32_t q = max(s, d0 + 64u); q < e; ++q) rank += (fbits(vel[tmp[q].x].w) < myId) ? 1u : 0u;       // ... and behind its last one
    }
    if (!act) return;
    if (!COPY || (fbits(P.w) & F_GHOST1)) order[s + rank] = me.x;                  // (read only for ghosts: special_slot passes an inactive ghost's density through)
    if (COPY) {
        pv[2u * (s + rank)] = make_float4(P.x, P.y, P.z, RP.x > 0.0f ? 1.0f / RP.x : 0.0f);
        pv[2u * (s + rank) + 1u] = make_float4(V.x, V.y, V.z, RP.y);
        const uint32_t cxy = c % (uint32_t)(gx * gy);
        const uint32_t cellBits = (cxy % (uint32_t)gx) | ((cxy / (uint32_t)gx) << 10) | ((c / (uint32_t)(gx * gy)) << 20);   // dims <= 1024 (validate_params)
        own[s + rank] = make_float4(bitsf(cellBits), F, P.w, V.w);
    }
}

struct StateIn {
    const float4* __restrict__ pos;
    const float4* __restrict__ vel;
    const float2* __restrict__ rp;
    const float* __restrict__ foam;
};
struct StateOut {
    float4* __restrict__ pos;
    float4* __restrict__ vel;
    float2* __restrict__ rp;
    float* __restrict__ foam;
    float4* __restrict__ acc;
    // eager AoS mode: the public 80-byte array is updated straight from the SPH pass (the values
    // are in registers here); nullptr in lazy / z-slab mode, where k_writeback runs on demand
    SphParticle* __restrict__ aos;
    uint32_t idBase;
};

// Fields SPHFluid.comp / OBBConstraints.comp change, written into the record of particle `id`.
__device__ __forceinline__ void aos_write_fluid(SphParticle* __restrict__ aos, uint32_t id, float px, float py, float pz,
                                                float vx, float vy, float vz, float ax, float ay, float az,
                                                float rho, float prs, float foam) {
    float* rec = reinterpret_cast<float*>(aos + id);
    rec[0] = px; rec[1] = py; rec[2] = pz;
    rec[4] = vx; rec[5] = vy; rec[6] = vz;
    *reinterpret_cast<float4*>(rec + 8) = make_float4(ax, ay, az, 0.0f);
    rec[12] = rho; rec[13] = prs; rec[14] = foam;
}
// Ghost branch of SPHFluid.comp:72-83 for an ACTIVE ghost: vel = acc = vec4(0), density = rho0, pressure = 0.
__device__ __forceinline__ void aos_write_active_ghost(SphParticle* __restrict__ aos, uint32_t id, float rho0) {
    float* rec = reinterpret_cast<float*>(aos + id);
    *reinterpret_cast<float4*>(rec + 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    *reinterpret_cast<float4*>(rec + 8) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    rec[12] = rho0; rec[13] = 0.0f;
}

// OBBConstraints.comp for container shapes 7..14, applied to the SPH pass's output state
// (slot order).  liveCount (nullable) bounds the slots that hold particles in slab mode.
__global__ __launch_bounds__(kBlock) void k_obb_ext(SimK k, ShapeTab T, float4* __restrict__ pos, float4* __restrict__ vel,
                                                    const uint32_t* __restrict__ liveCount, int n, const float4* __restrict__ own) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    const int bound = liveCount ? min(n, (int)*liveCount) : n;
    if (s >= bound) return;
    float4 P = pos[s];
    if (fbits(P.w) & (F_GHOSTNZ | F_HALO)) return;          // OBBConstraints.comp:46; halo copies are never targets
    float4 V = vel[s];
    obb_apply_ext(k, T, P.x, P.y, P.z, V.x, V.y, V.z);
    pos[s] = P; vel[s] = V;
    if (k.slabFlags && own) slab_check_layer_move(k, (int)(fbits(own[s].x) >> 20), P.z);   // against the layer the substep began in (sorted copy, still in slot order)
}

// shaders/FountainRecycle.comp:24-54 on the SPH pass's output state.  The particle index of the
// shader is the (global) id kept in vel.w; cos / sin are the pinned routines.
struct FountainK {
    float ex, ey, ez;          // uEmitterPos
    float radius, spread, jet; // uEmitterRadius, uJetSpread, uJetSpeed
    float drainY, chance, rho0;
    uint32_t seedMul;          // uSeed * 747796405u
};
__device__ __forceinline__ float lcg_next(uint32_t& s) {
    s = s * 1664525u + 1013904223u;
    return (float)(s & 0xFFFFFFu) / 16777215.0f;
}
__global__ __launch_bounds__(kBlock) void k_fountain(FountainK f, float4* __restrict__ pos, float4* __restrict__ vel,
                                                     float2* __restrict__ rp, float4* __restrict__ acc, SphParticle* __restrict__ aos,
                                                     uint32_t idBase, const uint32_t* __restrict__ liveCount, int n) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    const int bound = liveCount ? min(n, (int)*liveCount) : n;
    if (s >= bound) return;
    const float4 P = pos[s];
    const uint32_t flags = fbits(P.w);
    if (flags & (F_GHOST1 | F_HALO)) return;                 // :34 (isGhost == 1 only)
    if (P.y >= f.drainY) return;                             // :35
    const float4 V = vel[s];
    const uint32_t id = fbits(V.w);
    uint32_t seed = (id ^ f.seedMul) + 2891336453u;
    if (lcg_next(seed) > f.chance) return;                   // :38
    const float r1 = lcg_next(seed), r2 = lcg_next(seed), r3 = lcg_next(seed), r4 = lcg_next(seed);
    const float ang = 6.2831853f * r1;
    const float rad = f.radius * sqrtf(r2);
    const float ca = sph_cosf(ang), sa = sph_sinf(ang);
    const float px = f.ex + ca * rad, py = f.ey + 0.2f * r3, pz = f.ez + sa * rad;
    const float sm = f.spread * r4;
    const float sx = ca * sm, sz = sa * sm;
    const float len = sqrtf(dot3(sx, 1.0f, sz, sx, 1.0f, sz));
    const float vx = f.jet * (sx / len), vy = f.jet * (1.0f / len), vz = f.jet * (sz / len);
    pos[s] = make_float4(px, py, pz, P.w);
    vel[s] = make_float4(vx, vy, vz, V.w);
    rp[s] = make_float2(f.rho0, 0.0f);
    if (acc) acc[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (aos) {
        float* rec = reinterpret_cast<float*>(aos + (id - idBase));
        rec[0] = px; rec[1] = py; rec[2] = pz;
        rec[4] = vx; rec[5] = vy; rec[6] = vz;
        *reinterpret_cast<float4*>(rec + 8) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        rec[12] = f.rho0; rec[13] = 0.0f;
    }
}

// River / stream mode: shaders/TerrainConstraints.comp, ChannelConstraint.comp and StreamEmit.comp (DispatchCompute
// step 5, SPHFluid3D.cpp:511-516) on the SPH pass's output state.  Three dispatches in the reference; each touches only
// its own particle, so one kernel applies them in the same order.  mix(a, b, t) = a (1 - t) + b t, normalize = v / sqrt(dot),
// sin / cos = sph_sinf / sph_cosf (the oracle's definitions).
struct RiverK {
    int W, H;
    float minX, minZ, sizeX, sizeZ;            // terrainMin / terrainSize
    float restitution, oneMinusFriction;       // 0.02, 1 - 0.05 (:554-555)
    float centerX, amp, freq, phase, width;    // boxCenterX, riverAmp, riverFreq, riverPhase, channelWidth
    float flowGravity, dt;                     // 80, param_timeStep (:572-573)
    float sinkY, sinkZMax, ex, ey, ez, evx, evy, evz, radius, spreadZ, rho0;   // :583-591
};
__device__ __forceinline__ float river_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ float river_height(const RiverK& r, const float* __restrict__ T, float wx, float wz) {   // TerrainConstraints.comp:21-34
    float u = (wx - r.minX) / r.sizeX * (float)(r.W - 1);
    float v = (wz - r.minZ) / r.sizeZ * (float)(r.H - 1);
    u = clampf(u, 0.0f, (float)(r.W - 2));
    v = clampf(v, 0.0f, (float)(r.H - 2));
    const int ix = (int)u, iz = (int)v;
    const float fx = u - (float)ix, fz = v - (float)iz;
    const float* a = T + iz * r.W + ix;
    const float* b = a + r.W;
    return river_mix(river_mix(a[0], a[1], fx), river_mix(b[0], b[1], fx), fz);
}
__global__ __launch_bounds__(kBlock) void k_river(RiverK r, const float* __restrict__ T, float4* __restrict__ pos, float4* __restrict__ vel,
                                                  float2* __restrict__ rp, float4* __restrict__ acc, SphParticle* __restrict__ aos,
                                                  SphParticle* __restrict__ aosW, uint32_t idBase, int n) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const float4 P = pos[s];
    if (fbits(P.w) & F_GHOST1) return;                       // flags.x == 1 in all three shaders
    const float4 V = vel[s];
    const uint32_t id = fbits(V.w);
    float px = P.x, py = P.y, pz = P.z, vx = V.x, vy = V.y, vz = V.z;
    // ---- TerrainConstraints.comp:50-80 ----
    if (!(px < r.minX || px > r.minX + r.sizeX || pz < r.minZ || pz > r.minZ + r.sizeZ)) {
        const float ty = river_height(r, T, px, pz);
        if (py < ty) {
            const float ddx = r.sizeX / (float)(r.W - 1), ddz = r.sizeZ / (float)(r.H - 1);
            const float hR = river_height(r, T, px + ddx, pz), hL = river_height(r, T, px - ddx, pz);
            const float hF = river_height(r, T, px, pz + ddz), hB = river_height(r, T, px, pz - ddz);
            const float nx = hL - hR, ny = 2.0f * ddx, nz = hB - hF;
            const float nl = sqrtf(dot3(nx, ny, nz, nx, ny, nz));
            const float Nx = nx / nl, Ny = ny / nl, Nz = nz / nl;
            py = ty + 0.001f;
            const float vN = dot3(vx, vy, vz, Nx, Ny, Nz);
            if (vN < 0.0f) {
                const float ax = vN * Nx, ay = vN * Ny, az = vN * Nz;
                const float tx = vx - ax, tyv = vy - ay, tz = vz - az;
                vx = -r.restitution * ax + r.oneMinusFriction * tx;
                vy = -r.restitution * ay + r.oneMinusFriction * tyv;
                vz = -r.restitution * az + r.oneMinusFriction * tz;
            }
        }
    }
    // ---- ChannelConstraint.comp:27-46 ----
    {
        const float arg = r.freq * pz + r.phase;
        const float cx = r.centerX + r.amp * sph_sinf(arg);
        const float dx = px - cx;
        const float tdx = r.amp * r.freq * sph_cosf(arg);
        const float tlen = sqrtf(tdx * tdx + 1.0f);
        vx = vx + (tdx / tlen) * r.flowGravity * r.dt;
        vz = vz + (1.0f / tlen) * r.flowGravity * r.dt;
        if (fabsf(dx) > r.width) {
            px = cx + signf(dx) * r.width;
            if (dx * vx > 0.0f) vx = 0.0f;
        }
    }
    // ---- StreamEmit.comp:31-60 ----
    const bool dead = (py < r.sinkY) || (pz > r.sinkZMax);
    if (dead) {
        uint32_t seed = id * 1664525u + 1013904223u;
        const float r1 = (float)(seed & 0xFFFFu) / 65535.0f;
        seed = seed * 1664525u + 1013904223u;                // r2: drawn, unused
        seed = seed * 1664525u + 1013904223u;
        const float r3 = (float)(seed & 0xFFFFu) / 65535.0f;
        seed = seed * 1664525u + 1013904223u;
        const float r4 = (float)(seed & 0xFFFFu) / 65535.0f;
        const float spawnZ = r.ez + r1 * r.spreadZ;
        const float cx = r.centerX + r.amp * sph_sinf(r.freq * spawnZ + r.phase);
        px = cx + (r4 - 0.5f) * 2.0f * r.radius;
        py = r.ey + r3 * 0.6f;
        pz = spawnZ;
        vx = r.evx; vy = r.evy; vz = r.evz;
        rp[s] = make_float2(r.rho0, 0.0f);
        if (acc) acc[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        reinterpret_cast<float*>(aosW + (id - idBase))[7] = 0.0f;             // p.vel = vec4(emitterVel, 0.0): the record's vel.w
    }
    pos[s] = make_float4(px, py, pz, P.w);
    vel[s] = make_float4(vx, vy, vz, V.w);
    if (aos) {
        float* rec = reinterpret_cast<float*>(aos + (id - idBase));
        rec[0] = px; rec[1] = py; rec[2] = pz;
        rec[4] = vx; rec[5] = vy; rec[6] = vz;
        if (dead) {
            *reinterpret_cast<float4*>(rec + 8) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            rec[12] = r.rho0; rec[13] = 0.0f;
        }
    }
}

// Render-side export: what the reference's particle renderers read from the SSBO
// (fluidDepth.vert / particleImpostor.vert: pos, padA foam, density, |vel|, padB dye), packed as one float4
// per particle in ORIGINAL order, so that a renderer can map its vertex buffer once and never touch the
// 80-byte records.  w: 0 = 1.0, 1 = density, 2 = foam (padA), 3 = speed, 4 = dye (padB).
__global__ __launch_bounds__(kBlock) void k_pack_render(const SphParticle* __restrict__ aos, float4* __restrict__ out, int wMode, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* rec = reinterpret_cast<const float*>(aos + i);
    float w = 1.0f;
    if (wMode == 1) w = rec[12];
    else if (wMode == 2) w = rec[14];
    else if (wMode == 3) w = sqrtf(dot3(rec[4], rec[5], rec[6], rec[4], rec[5], rec[6]));
    else if (wMode == 4) w = rec[15];
    out[i] = make_float4(rec[0], rec[1], rec[2], w);
}

// ---- public AoS update (original order; the array is never permuted) ---------------------
// Writes exactly the fields SPHFluid.comp/OBBConstraints.comp change: pos.xyz, vel.xyz,
// acc.xyzw, density, pressure, padA (fluid); vel/acc = 0, density = rho0, pressure = 0
// (active ghost, SPHFluid.comp:77-81); nothing for an inactive ghost (:73-76).
__global__ __launch_bounds__(kBlock) void k_writeback(SphParticle* __restrict__ aos, const float4* __restrict__ pos,
                                                      const float4* __restrict__ vel, const float2* __restrict__ rp,
                                                      const float* __restrict__ foam, const float4* __restrict__ acc,
                                                      uint32_t idBase, int n) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const float4 P = pos[s], V = vel[s];
    const uint32_t flags = fbits(P.w);
    const uint32_t id = fbits(V.w) - idBase;
    if ((flags & F_GHOST1) && (flags & F_INACTIVE)) return;
    float* rec = reinterpret_cast<float*>(aos + id);
    const float2 RP = rp[s];
    if (flags & F_GHOST1) {
        rec[4] = 0.0f; rec[5] = 0.0f; rec[6] = 0.0f; rec[7] = 0.0f;
        rec[8] = 0.0f; rec[9] = 0.0f; rec[10] = 0.0f; rec[11] = 0.0f;
        rec[12] = RP.x; rec[13] = RP.y;
        return;
    }
    const float4 A = acc[s];
    rec[0] = P.x; rec[1] = P.y; rec[2] = P.z;
    rec[4] = V.x; rec[5] = V.y; rec[6] = V.z;
    *reinterpret_cast<float4*>(rec + 8) = A;
    rec[12] = RP.x; rec[13] = RP.y; rec[14] = foam[s];
}

// ---- WaveImpulse.comp:30-46 on the internal state (and on the AoS when it is current) -----
struct WaveK {
    float amplitude, kk, phase, ndx, ndy, ndz, yMin, yMax;
};
__global__ __launch_bounds__(kBlock) void k_wave_impulse(WaveK w, float4* __restrict__ pos, float4* __restrict__ vel,
                                                         SphParticle* __restrict__ aosOrNull, uint32_t idBase, int n) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const float4 P = pos[s];
    if (fbits(P.w) & F_GHOSTNZ) return;
    if (P.y < w.yMin || P.y > w.yMax) return;
    float4 V = vel[s];
    const float theta = fmaf(w.kk, dot3(P.x, P.y, P.z, w.ndx, w.ndy, w.ndz), w.phase);
    const float kick = w.amplitude * sph_sinf(theta);
    V.x = fmaf(w.ndx, kick, V.x); V.y = fmaf(w.ndy, kick, V.y); V.z = fmaf(w.ndz, kick, V.z);
    vel[s] = V;
    if (aosOrNull) {
        float* rec = reinterpret_cast<float*>(aosOrNull + (fbits(V.w) - idBase));
        rec[4] = V.x; rec[5] = V.y; rec[6] = V.z;
    }
}

// ======================= per-frame impulse kernels ================
// VortexImpulse / AttractorImpulse / StencilAttract / CurlFlow.comp: velocity kicks with no
// neighbour access, applied between substeps (Scene0p.cpp:3479-3531).  Each functor returns the
// new velocity of one non-ghost particle; k_impulse updates the internal state and, when it is
// current, the public array.  Operation order matches the oracle (fma only inside dot3).
__device__ __forceinline__ float smoothstepf(float e0, float e1, float x) {
    const float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}

struct VortexK {                       // VortexImpulse.comp:25-30
    float cx, cy, cz, ax, ay, az, tangent, inward, e1;
    __device__ __forceinline__ bool apply(const float4& P, uint32_t, float& vx, float& vy, float& vz) const {
        const float rx = P.x - cx, ry = P.y - cy, rz = P.z - cz;
        const float d = dot3(rx, ry, rz, ax, ay, az);
        const float qx = rx - ax * d, qy = ry - ay * d, qz = rz - az * d;
        const float r = sqrtf(dot3(qx, qy, qz, qx, qy, qz));
        if (r < 1e-4f) return false;
        const float hx = qx / r, hy = qy / r, hz = qz / r;
        const float tx = ay * hz - az * hy, ty = az * hx - ax * hz, tz = ax * hy - ay * hx;
        const float fall = smoothstepf(0.0f, e1, r);
        const float kt = tangent * fall, ki = inward * fall;
        vx = vx + (tx * kt - hx * ki); vy = vy + (ty * kt - hy * ki); vz = vz + (tz * kt - hz * ki);
        return true;
    }
};
struct AttractorK {                    // AttractorImpulse.comp:23-27
    float px, py, pz, pullSoft, soften, radius, e0;
    __device__ __forceinline__ bool apply(const float4& P, uint32_t, float& vx, float& vy, float& vz) const {
        const float rx = px - P.x, ry = py - P.y, rz = pz - P.z;
        const float d = sqrtf(dot3(rx, ry, rz, rx, ry, rz));
        if (d < 1e-5f) return false;
        float pull = pullSoft / (d + soften);
        pull = pull * (1.0f - smoothstepf(e0, radius, d));
        vx = vx + (rx / d) * pull; vy = vy + (ry / d) * pull; vz = vz + (rz / d) * pull;
        return true;
    }
};
struct StencilK {                      // StencilAttract.comp:26-29
    const float4* __restrict__ targets;
    uint32_t nTargets;
    float pull, oneMinusDamp;
    __device__ __forceinline__ bool apply(const float4& P, uint32_t index, float& vx, float& vy, float& vz) const {
        const float4 t = targets[index % nTargets];
        vx = (vx + (t.x - P.x) * pull) * oneMinusDamp;
        vy = (vy + (t.y - P.y) * pull) * oneMinusDamp;
        vz = (vz + (t.z - P.z) * pull) * oneMinusDamp;
        return true;
    }
};
__device__ __forceinline__ float fractf(float x) { return x - floorf(x); }
__device__ __forceinline__ float hash13(float x, float y, float z) {            // CurlFlow.comp:30-34
    x = fractf(x * 0.1031f); y = fractf(y * 0.1031f); z = fractf(z * 0.1031f);
    const float dd = dot3(x, y, z, z + 31.32f, y + 31.32f, x + 31.32f);
    x = x + dd; y = y + dd; z = z + dd;
    return fractf((x + y) * z);
}
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ float vnoise(float x, float y, float z) {            // CurlFlow.comp:36-50
    const float ix = floorf(x), iy = floorf(y), iz = floorf(z);
    float fx = x - ix, fy = y - iy, fz = z - iz;
    fx = (fx * fx) * (3.0f - 2.0f * fx); fy = (fy * fy) * (3.0f - 2.0f * fy); fz = (fz * fz) * (3.0f - 2.0f * fz);
    const float n000 = hash13(ix, iy, iz), n100 = hash13(ix + 1.0f, iy, iz);
    const float n010 = hash13(ix, iy + 1.0f, iz), n110 = hash13(ix + 1.0f, iy + 1.0f, iz);
    const float n001 = hash13(ix, iy, iz + 1.0f), n101 = hash13(ix + 1.0f, iy, iz + 1.0f);
    const float n011 = hash13(ix, iy + 1.0f, iz + 1.0f), n111 = hash13(ix + 1.0f, iy + 1.0f, iz + 1.0f);
    return mixf(mixf(mixf(n000, n100, fx), mixf(n010, n110, fx), fy), mixf(mixf(n001, n101, fx), mixf(n011, n111, fx), fy), fz);
}
__device__ __forceinline__ float pot1(float x, float y, float z) { return vnoise(x, y, z); }
__device__ __forceinline__ float pot2(float x, float y, float z) { return vnoise(x + 31.416f, y + 47.853f, z + 12.793f); }
__device__ __forceinline__ float pot3(float x, float y, float z) { return vnoise(x + -233.145f, y + 93.912f, z + 55.121f); }
struct CurlK {                         // CurlFlow.comp:25-28
    float kick, scale, time;
    __device__ __forceinline__ bool apply(const float4& P, uint32_t, float& vx, float& vy, float& vz) const {
        const float h = 0.35f;
        const float qx = P.x * scale, qy = P.y * scale, qz = P.z * scale + time;
        const float dP3dy = pot3(qx, qy + h, qz) - pot3(qx, qy - h, qz);
        const float dP2dz = pot2(qx, qy, qz + h) - pot2(qx, qy, qz - h);
        const float dP1dz = pot1(qx, qy, qz + h) - pot1(qx, qy, qz - h);
        const float dP3dx = pot3(qx + h, qy, qz) - pot3(qx - h, qy, qz);
        const float dP2dx = pot2(qx + h, qy, qz) - pot2(qx - h, qy, qz);
        const float dP1dy = pot1(qx, qy + h, qz) - pot1(qx, qy - h, qz);
        const float inv = 2.0f * h;
        const float cx = (dP3dy - dP2dz) / inv, cy = (dP1dz - dP3dx) / inv, cz = (dP2dx - dP1dy) / inv;
        const float m = sqrtf(dot3(cx, cy, cz, cx, cy, cz));
        float dx = 0.0f, dy = 0.0f, dz = 0.0f;
        if (m > 1e-5f) { dx = cx / m; dy = cy / m; dz = cz / m; }
        const float mm = fminf(m, 1.0f);
        vx = vx + (dx * mm) * kick; vy = vy + (dy * mm) * kick; vz = vz + (dz * mm) * kick;
        return true;
    }
};

template <class K>
__global__ __launch_bounds__(kBlock) void k_impulse(K kk, const float4* __restrict__ pos, float4* __restrict__ vel,
                                                    SphParticle* __restrict__ aosOrNull, uint32_t idBase, int n) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const float4 P = pos[s];
    if (fbits(P.w) & (F_GHOSTNZ | F_DEAD)) return;             // `if (p.isGhost != 0) return;`
    float4 V = vel[s];
    const uint32_t index = fbits(V.w) - idBase;
    if (!kk.apply(P, index, V.x, V.y, V.z)) return;
    vel[s] = V;
    if (aosOrNull) {
        float* rec = reinterpret_cast<float*>(aosOrNull + index);
        rec[4] = V.x; rec[5] = V.y; rec[6] = V.z;
    }
}

// ---- test support: per-cell counts in the reference's cell indexing -------------------------
__global__ __launch_bounds__(kBlock) void k_debug_cells(const uint32_t* __restrict__ cellStart, int32_t* __restrict__ cellCount, int numCells) {
    int c = blockIdx.x * kBlock + threadIdx.x;
    if (c < numCells) cellCount[c] = (int32_t)(cellStart[c + 1] - cellStart[c]);
}
__global__ __launch_bounds__(kBlock) void k_debug_particle_cell(SimK k, const float4* __restrict__ pos, const float4* __restrict__ vel,
                                                                int32_t* __restrict__ particleCell, uint32_t idBase, int n) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    float4 p = pos[s];
    int cx = cell_axis(p.x, k.gminx, k.cellSize, k.gx);
    int cy = cell_axis(p.y, k.gminy, k.cellSize, k.gy);
    int cz = cell_z_local(k, p.z);
    particleCell[fbits(vel[s].w) - idBase] = (cz * k.gy + cy) * k.gx + cx;
}


// ======================= linked-list grid build (A/B variant, SPH_OPT_GRID_BUILD = 1) ==========
// The reference's own scheme, kept for the BASELINE configs[1] comparison: ClearGrid.comp writes
// cellHead = -1, BuildGrid.comp pushes every particle on its cell's list with an atomic exchange
// and also writes particleCell / cellKey; the SPH pass then chases cellHead -> particleNext.
// Particles are never sorted in this mode (slot i stays particle i).  List order is the atomic
// arrival order, so fp32 sums are NOT reproducible here -- exactly the reference's behaviour; the
// parity tests for this variant use a tolerance instead of bit equality.
__global__ __launch_bounds__(kBlock) void k_ll_clear(int32_t* __restrict__ cellHead, int numCells) {
    int c = blockIdx.x * kBlock + threadIdx.x;
    if (c < numCells) cellHead[c] = -1;                               // ClearGrid.comp:9
}
__global__ __launch_bounds__(kBlock) void k_ll_build(SimK k, const float4* __restrict__ pos, int32_t* __restrict__ cellHead,
                                                     int32_t* __restrict__ particleNext, int32_t* __restrict__ particleCell,
                                                     int32_t* __restrict__ cellKey, int n) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos[i];
    const int cx = cell_axis(p.x, k.gminx, k.cellSize, k.gx);
    const int cy = cell_axis(p.y, k.gminy, k.cellSize, k.gy);
    const int cz = cell_z_local(k, p.z);
    const int cell = (cz * k.gy + cy) * k.gx + cx;                    // BuildGrid.comp:19,27
    particleCell[i] = cell;                                           // :29
    cellKey[i] = cell;                                                // :30
    particleNext[i] = atomicExch(&cellHead[cell], i);                 // :31-32
}

template <class F>
__device__ __forceinline__ void for_each_listed(const SimK& k, int cx, int cy, int cz, const int32_t* __restrict__ cellHead,
                                                const int32_t* __restrict__ particleNext, F&& f) {
    for (int dx = -1; dx <= 1; ++dx)                                  // the shader's own nesting, SPHFluid.comp:91-93
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const int nx = cx + dx, ny = cy + dy, nz = cz + dz;
                if (nx < 0 || ny < 0 || nz < 0 || nx >= k.gx || ny >= k.gy || nz >= k.gz) continue;
                int j = cellHead[nx + k.gx * (ny + k.gy * nz)];
                while (j != -1) { f(j); j = particleNext[j]; }
            }
}

__global__ __launch_bounds__(kBlock) void k_sph_ll(SimK k, StateIn in, StateOut out, const int32_t* __restrict__ cellHead,
                                                   const int32_t* __restrict__ particleNext, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 P = in.pos[i], V = in.vel[i];
    const float2 RP = in.rp[i];
    const float foamIn = in.foam[i];
    const uint32_t flags = fbits(P.w), id = fbits(V.w);
    Own o;
    own_reset(o);
    o.px = P.x; o.py = P.y; o.pz = P.z; o.vx = V.x; o.vy = V.y; o.vz = V.z; o.rho = RP.x; o.prs = RP.y;
    if (flags & F_GHOST1) {
        if (!(flags & F_INACTIVE)) { o.vx = o.vy = o.vz = 0.0f; o.rho = k.rho0; o.prs = 0.0f; }
        out.pos[i] = P;
        out.vel[i] = make_float4(o.vx, o.vy, o.vz, V.w);
        out.rp[i] = make_float2(o.rho, o.prs);
        out.foam[i] = foamIn;
        if (out.aos) { if (!(flags & F_INACTIVE)) aos_write_active_ghost(out.aos, id - out.idBase, k.rho0); }
        else out.acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const int cx = cell_axis(P.x, k.gminx, k.cellSize, k.gx);
    const int cy = cell_axis(P.y, k.gminy, k.cellSize, k.gy);
    const int cz = cell_z_local(k, P.z);
    for_each_listed(k, cx, cy, cz, cellHead, particleNext, [&](int j) {
        const float4 J = in.pos[j];
        pair_density(k, o, J.x, J.y, J.z, (int32_t)-1);
    });
    finish_density(k, o);
    for_each_listed(k, cx, cy, cz, cellHead, particleNext, [&](int j) {
        const float4 J = in.pos[j], JV = in.vel[j];
        const float2 JR = in.rp[j];
        pair_force(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, JR.y, JR.x > 0.0f ? 1.0f / JR.x : 0.0f, (int32_t)(j != i ? -1 : 0));
    });
    integrate(k, o);
    for_each_listed(k, cx, cy, cz, cellHead, particleNext, [&](int j) {
        const float4 J = in.pos[j], JV = in.vel[j];
        const float2 JR = in.rp[j];
        pair_xsph(k, o, J.x, J.y, J.z, JV.x, JV.y, JV.z, JR.x > 0.0f ? 1.0f / JR.x : 0.0f, (int32_t)(j != i ? -1 : 0));
    });
    const float foamOut = finish_particle(k, o, foamIn);
    if (!(flags & F_GHOSTNZ)) obb_apply(k, o.px, o.py, o.pz, o.vx, o.vy, o.vz);   // OBBConstraints.comp:46
    out.pos[i] = make_float4(o.px, o.py, o.pz, bitsf(flags));
    out.vel[i] = make_float4(o.vx, o.vy, o.vz, bitsf(id));
    out.rp[i] = make_float2(o.rho, o.prs);
    out.foam[i] = foamOut;
    if (out.aos) aos_write_fluid(out.aos, id - out.idBase, o.px, o.py, o.pz, o.vx, o.vy, o.vz, o.ax, o.ay, o.az, o.rho, o.prs, foamOut);
    else out.acc[i] = make_float4(o.ax, o.ay, o.az, 0.0f);
}

// ======================= z-slab (multi-GPU) support ==========================================
// 64-byte record that crosses ranks: a migrant (flags without F_HALO: the receiver owns it)
// or a boundary-layer copy (F_HALO set: candidate only).  acc travels too: a migrant's record of THIS substep
// (the 80-byte contract includes acc) is complete on its new owner even though the new owner never computed it.
struct SlabRec {
    float px, py, pz, vx, vy, vz, rho, prs, foam;
    uint32_t id, flags, pad;
    float ax, ay, az, pad2;
};
static_assert(sizeof(SlabRec) == 64, "SlabRec is 64 bytes");

__global__ __launch_bounds__(kBlock) void k_slab_import(const SphParticle* __restrict__ aos, const uint32_t* __restrict__ ids,
                                                        float4* __restrict__ pos, float4* __restrict__ vel, float2* __restrict__ rp,
                                                        float* __restrict__ foam, int n) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4* rec = reinterpret_cast<const float4*>(aos + i);
    float4 p = rec[0], v = rec[1], d = rec[3];
    int4 fl = *reinterpret_cast<const int4*>(rec + 4);
    uint32_t flags = (fl.x == 1 ? F_GHOST1 : 0u) | (fl.x != 0 ? F_GHOSTNZ : 0u) | (fl.y == 0 ? F_INACTIVE : 0u);
    pos[i] = make_float4(p.x, p.y, p.z, bitsf(flags));
    vel[i] = make_float4(v.x, v.y, v.z, bitsf(ids[i]));
    rp[i] = make_float2(d.x, d.y);
    foam[i] = d.z;
}

__device__ __forceinline__ void slab_append(SlabRec* buf, uint32_t* counter, uint32_t cap, bool pred, const SlabRec& r) {
    const unsigned long long m = __ballot(pred);
    if (!pred) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (slot < cap) buf[slot] = r;
}

// Start of a substep on a slab rank [z0, z1): classify every local slot by its CURRENT position.
//   stale ghost                      -> dead
//   owned, still inside              -> stays; copied to the neighbour as a ghost if in a boundary layer
//   owned, crossed into a neighbour  -> record sent as a migrant; kept here as a ghost if it sits in the
//                                       adjacent layer, otherwise dead
// counters[0] = records for the lower neighbour, counters[1] = for the upper neighbour.
// COMPACT (round 4, the engine-owned face buffers): halo copies travel as 40-byte HaloRec (what a candidate needs: position, velocity,
// density, pressure, id, flags), only migrants as full 64-byte records; sendLo / sendHi then point at a face buffer (SlabFace layout
// below) and counters[8..11] count halo lo / hi, migrant lo / hi.
struct HaloRec {
    float px, py, pz, vx, vy, vz, rho, prs;
    uint32_t id, flags;
};
static_assert(sizeof(HaloRec) == 40, "HaloRec is 40 bytes");
struct SlabHdr {                                        // first 64 bytes of a face buffer / of the migrant message
    uint32_t magic, nHalo, nHaloTrue, nMig, nMigTrue, exchange;
    uint32_t msgHalo, msgMig;                           // round 5: records the SENDER's two messages of this exchange carry (the receiver compares them with what it sized its receives for: flag 32)
    uint32_t pad[8];
};
static_assert(sizeof(SlabHdr) == 64, "SlabHdr is 64 bytes");
// A face buffer of capacity cap: [SlabHdr][cap x SlabRec migrants][cap x HaloRec halo copies].  Two messages per direction and
// exchange: header + the migrants in use, and the halo copies in use.
__host__ __device__ __forceinline__ size_t slab_face_mig_off() { return sizeof(SlabHdr); }
__host__ __device__ __forceinline__ size_t slab_face_halo_off(uint32_t cap) { return sizeof(SlabHdr) + (size_t)cap * 64u; }
__host__ __device__ __forceinline__ size_t slab_face_bytes(uint32_t cap) { return sizeof(SlabHdr) + (size_t)cap * (64u + 40u); }

template <class R>
__device__ __forceinline__ void slab_append_t(R* buf, uint32_t* counter, uint32_t cap, bool pred, const R& r) {
    const unsigned long long m = __ballot(pred);
    if (!pred) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (slot < cap) buf[slot] = r;
}

template <bool COMPACT>
__global__ __launch_bounds__(kBlock) void k_slab_pack(SimK k, int z0, int z1, int hasLo, int hasHi, float4* __restrict__ pos,
                                                      const float4* __restrict__ vel, const float2* __restrict__ rp,
                                                      const float* __restrict__ foam, const float4* __restrict__ acc, int nBound, SlabRec* __restrict__ sendLo,
                                                      SlabRec* __restrict__ sendHi, uint32_t capLo, uint32_t capHi,
                                                      uint32_t* __restrict__ counters, const uint32_t* __restrict__ cellStart, int layerCells) {
    const int gzLocal = z1 - z0 + 2;                    // the rank's layers plus one ghost layer each side (k holds the GLOBAL grid here)
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int n = min(nBound, (int)counters[2]);       // counters[2] = slots that hold data (live count of the last sort)
    if ((int)(blockIdx.x * kBlock) >= n) return;       // whole block beyond the data
    // cellStart != nullptr: the slots are still in the order of the last counting sort (z-major) and nothing moved a
    // particle by more than kSlabJump cell layers since (unchanged container; a substep that did is reported, slab_check_layer_move).  Everything this pass acts on --
    // stale ghosts, face particles, migrants -- then entered that substep in one of the kSlabDepth lowest or highest
    // local layers, i.e. sits in two slot ranges at the ends; blocks in between have nothing to do.
    bool inEnds = true;
    if (cellStart && gzLocal > 2 * kSlabDepth) {
        const int endLo = (int)cellStart[kSlabDepth * layerCells], startHi = (int)cellStart[(gzLocal - kSlabDepth) * layerCells];
        const int b0 = (int)(blockIdx.x * kBlock);
        if (b0 >= endLo && b0 + kBlock <= startHi) return;
        // exactly the two ranges, slot by slot: the boundary-first substep (sph_slab_step_begin) runs this pass while the SPH
        // pass is still writing the slots in between
        inEnds = i < endLo || i >= startHi;
    }
    bool toLoBuf = false, toHiBuf = false;
    SlabRec r;
    r.px = r.py = r.pz = r.vx = r.vy = r.vz = r.rho = r.prs = r.foam = 0.0f; r.id = 0; r.flags = 0; r.pad = 0;
    r.ax = r.ay = r.az = r.pad2 = 0.0f;
    uint32_t fLo = 0, fHi = 0;
    if (i < n && inEnds) {
        const float4 P = pos[i];
        uint32_t flags = fbits(P.w);
        if (!(flags & F_DEAD)) {
            if (flags & F_HALO) {
                pos[i].w = bitsf(flags | F_DEAD);
            } else {
                const int cz = cell_z_global(k, P.z);
                const bool goLo = cz < z0, goHi = cz >= z1;
                const float4 V = vel[i];
                const float2 RP = rp[i];
                r.px = P.x; r.py = P.y; r.pz = P.z; r.vx = V.x; r.vy = V.y; r.vz = V.z;
                r.rho = RP.x; r.prs = RP.y; r.foam = foam[i]; r.id = fbits(V.w);
                if (acc) { const float4 A = acc[i]; r.ax = A.x; r.ay = A.y; r.az = A.z; }
                toLoBuf = hasLo && (goLo || cz == z0);
                toHiBuf = hasHi && (goHi || cz == z1 - 1);
                fLo = goLo ? flags : (flags | F_HALO);
                fHi = goHi ? flags : (flags | F_HALO);
                if (goLo) pos[i].w = bitsf(cz == z0 - 1 ? (flags | F_HALO) : (flags | F_DEAD));
                if (goHi) pos[i].w = bitsf(cz == z1 ? (flags | F_HALO) : (flags | F_DEAD));
            }
        }
    }
    if (!COMPACT) {
        r.flags = fLo;
        slab_append(sendLo, &counters[0], capLo, toLoBuf, r);
        r.flags = fHi;
        slab_append(sendHi, &counters[1], capHi, toHiBuf, r);
    } else {
        HaloRec hr;
        hr.px = r.px; hr.py = r.py; hr.pz = r.pz; hr.vx = r.vx; hr.vy = r.vy; hr.vz = r.vz; hr.rho = r.rho; hr.prs = r.prs; hr.id = r.id;
        char* const fl = reinterpret_cast<char*>(sendLo);
        char* const fh = reinterpret_cast<char*>(sendHi);
        r.flags = fLo; hr.flags = fLo;
        slab_append_t(reinterpret_cast<SlabRec*>(fl + slab_face_mig_off()), &counters[10], capLo, toLoBuf && !(fLo & F_HALO), r);
        slab_append_t(reinterpret_cast<HaloRec*>(fl + slab_face_halo_off(capLo)), &counters[8], capLo, toLoBuf && (fLo & F_HALO), hr);
        r.flags = fHi; hr.flags = fHi;
        slab_append_t(reinterpret_cast<SlabRec*>(fh + slab_face_mig_off()), &counters[11], capHi, toHiBuf && !(fHi & F_HALO), r);
        slab_append_t(reinterpret_cast<HaloRec*>(fh + slab_face_halo_off(capHi)), &counters[9], capHi, toHiBuf && (fHi & F_HALO), hr);
    }
}

// What a received record tells about the exchange's one assumption -- no particle crosses more than one cell layer in z per
// substep.  An OWNED record (a migrant) must land inside the receiver's layers, and
// not in the layer next to its OTHER face: that layer's particles are halo copies on a third rank, and this exchange is already
// past the point where the receiver could have made one.  Otherwise the run goes on, but no longer equals the single-domain run:
// counters[4] bit 4.  (fromLo: the record came from the lower neighbour.)
struct SlabGeom {
    float gminz, cellSize;
    int gzGlobal, z0, z1, hasLo, hasHi;
};
__device__ __forceinline__ bool slab_record_misplaced(const SlabGeom& g, const SlabRec& r, bool fromLo) {
    if (r.flags & (F_HALO | F_DEAD)) return false;
    const int cz = cell_axis(r.pz, g.gminz, g.cellSize, g.gzGlobal);
    if (cz < g.z0 || cz >= g.z1) return true;                               // belongs to a rank further on
    if (g.z1 - g.z0 < 2) return false;
    return fromLo ? (g.hasHi && cz == g.z1 - 1) : (g.hasLo && cz == g.z0);  // a third rank needed its halo copy in this very exchange
}

// Append received records behind the local slots.
__global__ __launch_bounds__(kBlock) void k_slab_unpack(const SlabRec* __restrict__ recv, int nRecv, float4* __restrict__ pos,
                                                        float4* __restrict__ vel, float2* __restrict__ rp, float* __restrict__ foam,
                                                        float4* __restrict__ acc, int dstBase, SlabGeom g, int fromLo, uint32_t* __restrict__ counters) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nRecv) return;
    const SlabRec r = recv[i];
    const int d = dstBase + i;
    pos[d] = make_float4(r.px, r.py, r.pz, bitsf(r.flags));
    vel[d] = make_float4(r.vx, r.vy, r.vz, bitsf(r.id));
    rp[d] = make_float2(r.rho, r.prs);
    foam[d] = r.foam;
    acc[d] = make_float4(r.ax, r.ay, r.az, 0.0f);
    if (slab_record_misplaced(g, r, fromLo != 0)) atomicOr(&counters[4], 16u);
}

// ---- exchange without host round trips: the record counts travel in a header in front of the payload ----
constexpr uint32_t kSlabMagic = 0x48414c4fu;            // "HALO": first word of a face's header
// counters[4], the exchange's flags -- bit 0: a send face overflowed, bit 1: the slab's slot capacity overflowed on unpack, bit 2: a received
// face did not start with a valid header, bit 3: the sender had more records than its message carried, bit 4 (16, a notice): a particle crossed more
// cell layers within one substep than the exchange follows (slab_check_layer_move, slab_record_misplaced), bit 5 (32, round 5): the two ends of a
// link sized this exchange's messages differently (the header says what the sender's messages carry; the receiver's sizes come out of its own plan).
// ---- round 4: the compact faces (SlabFace layout above).  counters: [8] halo copies for lo, [9] for hi, [10] migrants for lo, [11] for hi
// (running, reset here), [12..15] the same four of the last pack (sph_slab_status / the host's message sizing), [7] exchanges so far.
// msgHalo* / msgMig*: the records this engine's messages of THIS exchange will carry (the host's plan, slab_plan in sph_engine.hip).
__global__ void k_slab_headers2(uint32_t* __restrict__ counters, char* __restrict__ faceLo, char* __restrict__ faceHi, uint32_t cap,
                                uint32_t msgHaloLo, uint32_t msgMigLo, uint32_t msgHaloHi, uint32_t msgMigHi) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t ex = counters[7];
    for (int d = 0; d < 2; ++d) {
        char* f = d ? faceHi : faceLo;
        const uint32_t nh = counters[8 + d], nm = counters[10 + d];
        if (nh > cap || nm > cap) atomicOr(&counters[4], 1u);
        if (f) {
            SlabHdr h;
            h.magic = kSlabMagic; h.nHalo = min(nh, cap); h.nHaloTrue = nh; h.nMig = min(nm, cap); h.nMigTrue = nm; h.exchange = ex;
            h.msgHalo = d ? msgHaloHi : msgHaloLo; h.msgMig = d ? msgMigHi : msgMigLo;
            for (int i = 0; i < 8; ++i) h.pad[i] = 0u;
            *reinterpret_cast<SlabHdr*>(f) = h;
        }
        counters[12 + d] = nh; counters[14 + d] = nm;
        counters[8 + d] = 0u; counters[10 + d] = 0u;
    }
    counters[5] = counters[12] + counters[14]; counters[6] = counters[13] + counters[15];   // what sph_slab_status reports: records per direction
    counters[7] = ex + 1u;
}
// What of a received face may be used: the header must be one, and the counts are cut to what the MESSAGES carried
// (msgHalo / msgMig records: the receiver sized them from the sender's counts of two exchanges ago; more than that -> error bit 8).
struct FaceCounts { uint32_t nHalo, nMig; bool bad, more, sized; };
__device__ __forceinline__ FaceCounts slab_face_counts(const char* __restrict__ face, uint32_t cap, uint32_t msgHalo, uint32_t msgMig) {
    FaceCounts c{0u, 0u, false, false, false};
    if (!face) return c;
    const SlabHdr h = *reinterpret_cast<const SlabHdr*>(face);
    if (h.magic != kSlabMagic || h.nHalo > cap || h.nMig > cap) { c.bad = true; return c; }
    c.nHalo = min(h.nHalo, msgHalo); c.nMig = min(h.nMig, msgMig);
    c.more = h.nHaloTrue > c.nHalo || h.nMigTrue > c.nMig;
    c.sized = h.msgHalo != msgHalo || h.msgMig != msgMig;      // the sender's messages carried another number of records than this end received
    return c;
}
// Appends one received face behind slot counters[2] (+ `before` records of the face unpacked first): halo copies, then migrants.
__global__ __launch_bounds__(kBlock) void k_slab_unpack2(const char* __restrict__ face, const char* __restrict__ otherFirst, uint32_t cap, uint32_t msgHalo, uint32_t msgMig,
                                                         uint32_t otherMsgHalo, uint32_t otherMsgMig, float4* __restrict__ pos, float4* __restrict__ vel,
                                                         float2* __restrict__ rp, float* __restrict__ foam, float4* __restrict__ acc, uint32_t* __restrict__ counters,
                                                         uint32_t slotCap, SlabGeom g, int fromLo) {
    const FaceCounts c = slab_face_counts(face, cap, msgHalo, msgMig);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.nHalo + c.nMig) return;
    const FaceCounts o = slab_face_counts(otherFirst, cap, otherMsgHalo, otherMsgMig);
    const uint32_t d = counters[2] + o.nHalo + o.nMig + i;
    if (d >= slotCap) { atomicOr(&counters[4], 2u); return; }
    if (i < c.nHalo) {
        const HaloRec r = reinterpret_cast<const HaloRec*>(face + slab_face_halo_off(cap))[i];
        pos[d] = make_float4(r.px, r.py, r.pz, bitsf(r.flags));
        vel[d] = make_float4(r.vx, r.vy, r.vz, bitsf(r.id));
        rp[d] = make_float2(r.rho, r.prs);
        foam[d] = 0.0f;
        acc[d] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        const SlabRec r = reinterpret_cast<const SlabRec*>(face + slab_face_mig_off())[i - c.nHalo];
        pos[d] = make_float4(r.px, r.py, r.pz, bitsf(r.flags));
        vel[d] = make_float4(r.vx, r.vy, r.vz, bitsf(r.id));
        rp[d] = make_float2(r.rho, r.prs);
        foam[d] = r.foam;
        acc[d] = make_float4(r.ax, r.ay, r.az, 0.0f);
        if (slab_record_misplaced(g, r, fromLo != 0)) atomicOr(&counters[4], 16u);
    }
}
__global__ void k_slab_commit2(uint32_t* __restrict__ counters, const char* __restrict__ recvLo, const char* __restrict__ recvHi, uint32_t cap, uint32_t slotCap,
                               uint32_t msgHaloLo, uint32_t msgMigLo, uint32_t msgHaloHi, uint32_t msgMigHi) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t add = 0u, err = 0u;
    const FaceCounts a = slab_face_counts(recvLo, cap, msgHaloLo, msgMigLo), b = slab_face_counts(recvHi, cap, msgHaloHi, msgMigHi);
    if (a.bad || b.bad) err |= 4u;
    if (a.more || b.more) err |= 8u;
    if (a.sized || b.sized) err |= 32u;
    add = a.nHalo + a.nMig + b.nHalo + b.nMig;
    if (err) atomicOr(&counters[4], err);
    counters[2] = min(counters[2] + add, slotCap);
}

__global__ void k_set_u32(uint32_t* __restrict__ p, uint32_t v) { *p = v; }

// Owned particles of a slab rank as 64-byte records (pos3, vel3, acc3, rho, P, foam, id, flags, 2 pad),
// compacted; *count receives the number written.
struct SlabOut {
    float px, py, pz, vx, vy, vz, ax, ay, az, rho, prs, foam;
    uint32_t id, flags, pad0, pad1;
};
static_assert(sizeof(SlabOut) == 64, "SlabOut is 64 bytes");
__global__ __launch_bounds__(kBlock) void k_slab_download(const float4* __restrict__ pos, const float4* __restrict__ vel,
                                                          const float2* __restrict__ rp, const float* __restrict__ foam,
                                                          const float4* __restrict__ acc, int n, int accValid, SlabOut* __restrict__ out,
                                                          uint32_t cap, uint32_t* __restrict__ count) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    bool own = false;
    SlabOut o;
    o.px = o.py = o.pz = o.vx = o.vy = o.vz = o.ax = o.ay = o.az = o.rho = o.prs = o.foam = 0.0f; o.id = o.flags = o.pad0 = o.pad1 = 0;
    if (i < n) {
        const float4 P = pos[i];
        const uint32_t flags = fbits(P.w);
        if (!(flags & (F_DEAD | F_HALO))) {
            own = true;
            const float4 V = vel[i];
            const float2 RP = rp[i];
            o.px = P.x; o.py = P.y; o.pz = P.z; o.vx = V.x; o.vy = V.y; o.vz = V.z;
            if (accValid) { const float4 A = acc[i]; o.ax = A.x; o.ay = A.y; o.az = A.z; }
            o.rho = RP.x; o.prs = RP.y; o.foam = foam[i]; o.id = fbits(V.w); o.flags = flags;
        }
    }
    const unsigned long long m = __ballot(own);
    if (!own) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (slot < cap) out[slot] = o;
}

}  // namespace sph
