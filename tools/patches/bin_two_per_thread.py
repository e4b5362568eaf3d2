"""Variant patch: k_bin handles TWO particles per thread (slots i and i + 256 of a 512-slot block), both position loads and both returning atomics in
flight together: the pass is bound by the latency of load -> atomic -> store at full occupancy, not by its bytes.  usage: bin_two_per_thread.py <csrc dir>"""
import os, sys
d = sys.argv[1]
p = os.path.join(d, "sph_kernels.h")
s = open(p).read()
a = s.index("__global__ __launch_bounds__(kBlock) void k_bin(")
b = s.index("__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {")
new = r'''constexpr int kBinPerThread = BINPT;
__global__ __launch_bounds__(kBlock) void k_bin(SimK k, const float4* __restrict__ pos, uint2* __restrict__ binKey,
                                                uint32_t* __restrict__ cellCount, int n, const uint32_t* __restrict__ slotsInUse) {
    const int lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * (kBlock * kBinPerThread) + threadIdx.x;
    // z-slab mode without host round trips: n is only a launch bound, the slots that hold data are counted on the device
    const uint32_t inUse = slotsInUse ? *slotsInUse : 0xFFFFFFFFu;
    if ((uint32_t)(blockIdx.x * (kBlock * kBinPerThread)) >= inUse) return;        // whole block beyond the data (k_scatter skips it too)
    const unsigned long long upto = (2ull << lane) - 1ull;              // bits 0..lane
    uint32_t cell[kBinPerThread], base[kBinPerThread];
    int startLane[kBinPerThread];
    float4 p[kBinPerThread];
#pragma unroll
    for (int j = 0; j < kBinPerThread; ++j) {                            // both loads first
        const int i = i0 + j * kBlock;
        const bool used = i < n && (uint32_t)i < inUse;
        p[j] = used ? pos[i] : make_float4(0.0f, 0.0f, 0.0f, bitsf(F_DEAD));
    }
#pragma unroll
    for (int j = 0; j < kBinPerThread; ++j) {                            // both atomics next
        cell[j] = 0xFFFFFFFFu;                                          // also the key of dead / unused slots (z-slab mode): they get no slot
        const int cx = cell_axis(p[j].x, k.gminx, k.cellSize, k.gx);
        const int cy = cell_axis(p[j].y, k.gminy, k.cellSize, k.gy);
        const int cz = cell_z_local(k, p[j].z);
        if (!(fbits(p[j].w) & F_DEAD)) cell[j] = (uint32_t)((cz * k.gy + cy) * k.gx + cx);   // flatten(), BuildGrid.comp:19
        const bool valid = cell[j] != 0xFFFFFFFFu;
        const uint32_t prev = (uint32_t)__shfl_up((int)cell[j], 1, 64);
        const bool head = (lane == 0) || (cell[j] != prev);
        const unsigned long long heads = __ballot(head);
        startLane[j] = 63 - __clzll((long long)(heads & upto));
        const unsigned long long above = heads & ~upto;
        const int endLane = above ? (__ffsll((long long)above) - 1) : 64;
        base[j] = 0;
        if (head && valid) base[j] = atomicAdd(&cellCount[cell[j]], (uint32_t)(endLane - lane));
    }
#pragma unroll
    for (int j = 0; j < kBinPerThread; ++j) {
        const int i = i0 + j * kBlock;
        const uint32_t b = (uint32_t)__shfl((int)base[j], startLane[j], 64);
        if (i < n) binKey[i] = make_uint2(cell[j], b + (uint32_t)(lane - startLane[j]));
    }
}

'''
s = s[:a] + new.replace('BINPT', os.environ.get('BIN_PER_THREAD', '2')) + s[b:]
open(p, "w").write(s)
e = os.path.join(d, "sph_engine.hip")
t = open(e).read()
old = "hipLaunchKernelGGL(k_bin, dim3(nb), dim3(kBlock), 0, e->stream,"
assert t.count(old) == 1
t = t.replace(old, "hipLaunchKernelGGL(k_bin, dim3(blocks_for(n, kBlock * kBinPerThread)), dim3(kBlock), 0, e->stream,")
open(e, "w").write(t)
