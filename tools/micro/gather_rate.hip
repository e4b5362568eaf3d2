// TA cost of 16-byte / 4-byte per-lane gathers on gfx950 for the access patterns of the SPH list walks
// (how many cycles does one wave-level gather instruction keep a CU's address unit busy?).
// Build: hipcc -O3 --offload-arch=gfx950 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// PAT 0 coalesced, 1 permuted inside a 64-slot window, 2 lane + jitter(0..7) inside ~72 slots, 3 three rows x (lane + jitter),
//     4 random in 4 MB (L2), 5 pairs of adjacent lanes share a slot (+jitter)
template <int PAT, int W>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, float* __restrict__ out, int iters, unsigned mask) {
    const unsigned lane = threadIdx.x & 63, wv = (blockIdx.x * 4 + (threadIdx.x >> 6));
    unsigned base = (wv * 977u) & mask;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned h = hsh(lane * 131u + (unsigned)(it * 8 + u) * 2654435761u + wv);
            unsigned idx;
            if (PAT == 0) idx = base + lane;
            else if (PAT == 1) idx = base + (h & 63u);
            else if (PAT == 2) idx = base + lane + (h & 7u);
            else if (PAT == 3) idx = base + ((h >> 8) % 3u) * 4096u + lane + (h & 7u);
            else if (PAT == 4) idx = h;
            else idx = base + (lane >> 1) * 2u + (h & 7u);
            idx &= mask;
            if (W == 4) v[u] = src[idx];
            else { const float* s1 = reinterpret_cast<const float*>(src); v[u].x = s1[idx * 4]; v[u].y = v[u].z = v[u].w = 0.f; }
            base += 64u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// LDS: ds_read_b32 x6 per step with per-lane indices (lane + jitter) vs ds_read2_b32 x3 vs ds_read_b128 x2
template <int MODE>
__global__ __launch_bounds__(256) void kl(float* __restrict__ out, int iters) {
    __shared__ float sx[4][1024 * 4];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = lane; i < 4096; i += 64) sx[w][i] = (float)i;
    __builtin_amdgcn_wave_barrier();
    float acc = 0.f;
    unsigned j = hsh(threadIdx.x) & 7u;
    for (int it = 0; it < iters; ++it) {
        const unsigned i0 = (lane + j + (unsigned)it * 3u) & 1023u, i1 = (lane + 2u + ((j * 5u) & 7u) + (unsigned)it * 3u) & 1023u;
        if (MODE == 0) {        // SoA, two targets: 6 x b32
            acc += sx[w][i0] + sx[w][1024 + i0] + sx[w][2048 + i0] + sx[w][i1] + sx[w][1024 + i1] + sx[w][2048 + i1];
        } else if (MODE == 1) { // SoA, two consecutive candidates: 3 x read2_b32
            acc += sx[w][i0] + sx[w][i0 + 1] + sx[w][1024 + i0] + sx[w][1024 + i0 + 1] + sx[w][2048 + i0] + sx[w][2048 + i0 + 1];
        } else {                // AoS float4, two targets: 2 x b128
            const float4 a = reinterpret_cast<const float4*>(sx[w])[i0], b = reinterpret_cast<const float4*>(sx[w])[i1];
            acc += a.x + a.y + a.z + b.x + b.y + b.z;
        }
        j = (j * 5u + 1u) & 7u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    const unsigned n = 1u << 22;   // 4M float4 = 64 MB
    float4* src; float* out;
    hipMalloc(&src, (size_t)n * 16); hipMalloc(&out, 2048 * 256 * 4);
    hipMemset(src, 0, (size_t)n * 16);
    const int iters = 200, blocks = 2048;
    const double instr = (double)blocks * 4 * iters * 8;     // wave-level gather instructions
    const double cuCycles = 2.4e9 / 256.0;                   // per CU per second... printed as cycles per instruction per CU
#define RUN(P, W, MASK, NAME) { float ms = timeit([&] { hipLaunchKernelGGL((k<P, W>), dim3(blocks), dim3(256), 0, 0, src, out, iters, MASK); }); \
    printf("%-44s %7.3f ms  %6.1f cycles per wave-instr per CU (2.4 GHz)\n", NAME, ms, ms * 1e-3 * 2.4e9 * 256.0 / instr); }
    (void)cuCycles;
    const unsigned L2 = (1u << 18) - 1;   // 256K float4 = 4 MB footprint
    RUN(0, 4, L2, "x4 coalesced (4 MB)");
    RUN(1, 4, L2, "x4 permuted in 64-slot window");
    RUN(2, 4, L2, "x4 lane + jitter 0..7");
    RUN(3, 4, L2, "x4 three rows, lane + jitter");
    RUN(5, 4, L2, "x4 lane pairs share slot + jitter");
    RUN(4, 4, L2, "x4 random in 4 MB");
    RUN(4, 4, n - 1, "x4 random in 64 MB");
    RUN(0, 1, L2, "dword coalesced (stride 16 B)");
    RUN(2, 1, L2, "dword lane + jitter");
    RUN(3, 1, L2, "dword three rows, lane + jitter");
    RUN(4, 1, L2, "dword random in 4 MB");
    const int li = 4000;
#define RUNL(M, NAME) { float ms = timeit([&] { hipLaunchKernelGGL((kl<M>), dim3(1024), dim3(256), 0, 0, out, li); }); \
    printf("%-44s %7.3f ms  %6.1f cycles per step per CU\n", NAME, ms, ms * 1e-3 * 2.4e9 * 256.0 / ((double)1024 * 4 * li)); }
    RUNL(0, "LDS 6 x ds_read_b32 (SoA, two targets)");
    RUNL(1, "LDS 3 x ds_read2_b32 (SoA, adjacent pair)");
    RUNL(2, "LDS 2 x ds_read_b128 (AoS, two targets)");
    return 0;
}
