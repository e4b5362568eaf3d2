// Issue cost of the vector instructions the SPH pass is made of, on gfx950, as a function of the waves resident per
// SIMD.  Settles whether a wave64 VALU instruction costs 2 or 4 cycles of its SIMD (MI355X_MICROARCH.md says 2 with more
// than one wave resident; round 1's pk_rate said 4).  Every body is inline asm (no compiler re-association), 8 independent
// chains per lane, timed two ways: s_memtime inside the kernel (shader cycles per wave) and hipEvents (wall).
//   build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip     run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// one "round" = 8 independent instructions; ROUNDS rounds per loop iteration
#define K_FMA(i)    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
#define K_FMAC(i)   asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
#define K_MUL(i)    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(m));
#define K_ADD(i)    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define K_MAX(i)    asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define K_PKFMA(i)  asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
#define K_PKMUL(i)  asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
#define K_PKADD(i)  asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define K_PKADDC(i) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1] clamp" : "+v"(p[i]) : "v"(pc));
#define K_ADDU(i)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ui));
#define K_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(ui));
#define K_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(ui));
#define K_OR(i)     asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ui));
#define K_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ui));
#define K_CMP(i)    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(c) : "vcc");
#define K_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(qi));
#define K_RSQ(i)    asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
#define K_MIX(i)    asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %4, %5" : "+v"(x[i]), "+v"(p[i]) : "v"(m), "v"(c), "v"(pm), "v"(pc));

typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float a, float b, int iters) {
    float x[8]; v2f p[8]; uint32_t u[8]; unsigned long long q[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + i + threadIdx.x * 1e-3f; p[i] = v2f{a + i, b - i}; u[i] = threadIdx.x + i; q[i] = threadIdx.x + i; }
    const float m = a, c = b; const v2f pm = {a, b}, pc = {b, a}; const uint32_t ui = (uint32_t)iters | 1u; const unsigned long long qi = ui;
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[0]), "v"(c) : "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define ROUND(K) REP8(K) REP8(K) REP8(K) REP8(K)
        if (KIND == 0) { ROUND(K_FMA) }
        if (KIND == 1) { ROUND(K_FMAC) }
        if (KIND == 2) { ROUND(K_MUL) }
        if (KIND == 3) { ROUND(K_ADD) }
        if (KIND == 4) { ROUND(K_MAX) }
        if (KIND == 5) { ROUND(K_PKFMA) }
        if (KIND == 6) { ROUND(K_PKMUL) }
        if (KIND == 7) { ROUND(K_PKADD) }
        if (KIND == 8) { ROUND(K_PKADDC) }
        if (KIND == 9) { ROUND(K_ADDU) }
        if (KIND == 10) { ROUND(K_LSHLADD) }
        if (KIND == 11) { ROUND(K_ALIGNBIT) }
        if (KIND == 12) { ROUND(K_OR) }
        if (KIND == 13) { ROUND(K_CNDMASK) }
        if (KIND == 14) { ROUND(K_CMP) }
        if (KIND == 15) { ROUND(K_LSHLADD64) }
        if (KIND == 16) { ROUND(K_RSQ) }
        if (KIND == 17) { ROUND(K_MIX) }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y + (float)u[i] + (float)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* NAMES[] = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
                              "v_pk_add_f32 clamp", "v_add_u32", "v_lshl_add_u32", "v_alignbit_b32", "v_or_b32", "v_cndmask_b32", "v_cmp_lt_f32",
                              "v_lshl_add_u64", "v_rsq_f32", "v_fma_f32+v_pk_fma_f32 (pair = 2 instr)"};
typedef void (*KFn)(float*, unsigned long long*, float, float, int);
static KFn FNS[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>};

int main() {
    const int maxBlocks = 256 * 8;
    float* d; unsigned long long* dc;
    hipMalloc(&d, (size_t)maxBlocks * 256 * sizeof(float));
    hipMalloc(&dc, (size_t)maxBlocks * 4 * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    printf("# 256-thread blocks, blocks = 256 CUs x w  =>  w waves per SIMD; 32 instructions x %d iterations per wave\n", iters);
    printf("# cyc = median over waves of s_memtime ticks / instructions; 'SIMD cyc/instr' = cyc / w (waves sharing the SIMD)\n");
    printf("%-42s %3s %10s %12s %14s %12s\n", "instruction", "w", "ms", "cyc/instr", "SIMD cyc/instr", "GHz(eff)");
    for (int kind = 0; kind < 18; ++kind) {
        for (int w : {1, 2, 4, 8}) {
            const int blocks = 256 * w;
            double ms = 0; std::vector<unsigned long long> h(blocks * 4);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(FNS[kind], dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 0.5f, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1); ms = t;
            }
            hipMemcpy(h.data(), dc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double n = (double)iters * 32 * (kind == 17 ? 2 : 1);
            const double cyc = (double)h[h.size() / 2] / n;
            // s_memtime ticks at a fixed 100 MHz on gfx9?  report the ratio to wall time as an effective clock
            const double ghz = (double)h[h.size() / 2] / (ms * 1e-3) * 1e-9;
            printf("%-42s %3d %10.3f %12.3f %14.3f %12.3f\n", NAMES[kind], w, ms, cyc, cyc / w, ghz);
        }
    }
    return 0;
}
