// Issue rate of v_pk_fma_f32 vs v_fma_f32 on gfx950 (are packed fp32 ops double rate?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void k(float* out, float a, float b, int iters) {
    v2f x0 = {a + threadIdx.x, a}, x1 = {b, a}, x2 = {a, b + 1}, x3 = {b, b}, x4 = {a, 1}, x5 = {2, b}, x6 = {a, 3}, x7 = {4, b};
    const v2f m = {a, b}, c = {b, a};
    for (int i = 0; i < iters; ++i) {
        if (PK) {
            x0 = __builtin_elementwise_fma(x0, m, c); x1 = __builtin_elementwise_fma(x1, m, c); x2 = __builtin_elementwise_fma(x2, m, c); x3 = __builtin_elementwise_fma(x3, m, c);
            x4 = __builtin_elementwise_fma(x4, m, c); x5 = __builtin_elementwise_fma(x5, m, c); x6 = __builtin_elementwise_fma(x6, m, c); x7 = __builtin_elementwise_fma(x7, m, c);
        } else {
            x0.x = fmaf(x0.x, m.x, c.x); x1.x = fmaf(x1.x, m.x, c.x); x2.x = fmaf(x2.x, m.x, c.x); x3.x = fmaf(x3.x, m.x, c.x);
            x4.x = fmaf(x4.x, m.x, c.x); x5.x = fmaf(x5.x, m.x, c.x); x6.x = fmaf(x6.x, m.x, c.x); x7.x = fmaf(x7.x, m.x, c.x);
        }
    }
    v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 16 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    for (int pk = 0; pk < 2; ++pk) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d, 1.0001f, 0.5f, iters);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d, 1.0001f, 0.5f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts = (double)blocks * 8 /*waves*/ * iters * 8;
            if (rep) printf("%s: %.3f ms, %.3e wave-instr/s (%.2f instr/cycle/SIMD at 2.4 GHz)\n", pk ? "v_pk_fma_f32" : "v_fma_f32", ms, insts / (ms * 1e-3),
                            insts / (ms * 1e-3) / (1024 * 2.4e9));
        }
    }
    return 0;
}
