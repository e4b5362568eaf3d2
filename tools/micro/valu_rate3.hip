// Second pass of the VALU issue-cost table (gfx950): one row per instruction FORM the SPH pass could use, 8 waves per SIMD
// (2048 blocks x 256 threads), wall time only (hipEvents): ns per wave-instruction per SIMD and the same in cycles at the
// 2.4 GHz nominal clock.  8 independent chains per lane, bodies in inline asm.
//   build: hipcc -O3 --offload-arch=gfx950 -o valu_rate2 valu_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ROUND(K) REP8(K) REP8(K) REP8(K) REP8(K)

#define DEF(NAME, ASMSTR, ...) \
    struct NAME { static constexpr const char* name = ASMSTR; \
      template <class S> static __device__ __forceinline__ void run(S& s) { _Pragma("unroll") for (int r = 0; r < 4; ++r) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile(ASMSTR __VA_ARGS__); } } } };

struct St { float x[8]; v2f p[8]; uint32_t u[8]; unsigned long long q[8]; float m, c; v2f pm, pc; uint32_t ui; float sm, sc; };

// x: float chains; u: uint chains; p: packed chains.  s.m / s.c VGPR operands, s.sm / s.sc wave-uniform (SGPR) operands.
DEF(AddF,      "v_add_f32 %0, %0, %1",                 : "+v"(s.x[i]) : "v"(s.c))
DEF(SubF,      "v_sub_f32 %0, %0, %1",                 : "+v"(s.x[i]) : "v"(s.c))
DEF(MulF,      "v_mul_f32 %0, %0, %1",                 : "+v"(s.x[i]) : "v"(s.m))
DEF(AddFs,     "v_add_f32 %0, %1, %0",                 : "+v"(s.x[i]) : "s"(s.sc))
DEF(AddF64,    "v_add_f32_e64 %0, %0, %1",             : "+v"(s.x[i]) : "v"(s.c))
DEF(AddFclamp, "v_add_f32_e64 %0, %0, %1 clamp",       : "+v"(s.x[i]) : "v"(s.c))
DEF(SubFneg,   "v_sub_f32_e64 %0, %1, %0 clamp",       : "+v"(s.x[i]) : "s"(s.sc))
DEF(MaxF,      "v_max_f32 %0, %0, %1",                 : "+v"(s.x[i]) : "v"(s.c))
DEF(MaxF0,     "v_max_f32 %0, 0, %0",                  : "+v"(s.x[i]))
DEF(MinF,      "v_min_f32 %0, %0, %1",                 : "+v"(s.x[i]) : "v"(s.c))
DEF(Med3,      "v_med3_f32 %0, %0, %1, %2",            : "+v"(s.x[i]) : "v"(s.c), "v"(s.m))
DEF(FmaVVV,    "v_fma_f32 %0, %0, %1, %2",             : "+v"(s.x[i]) : "v"(s.m), "v"(s.c))
DEF(FmaVSV,    "v_fma_f32 %0, %0, %1, %2",             : "+v"(s.x[i]) : "s"(s.sm), "v"(s.c))
DEF(FmaVCC,    "v_fma_f32 %0, %0, 1.0, 0.5",           : "+v"(s.x[i]))
DEF(Fmac,      "v_fmac_f32 %0, %1, %2",                : "+v"(s.x[i]) : "v"(s.m), "v"(s.c))
DEF(FmacS,     "v_fmac_f32 %0, %1, %2",                : "+v"(s.x[i]) : "s"(s.sm), "v"(s.c))
DEF(FmaSq,     "v_fma_f32 %0, %1, %1, %0",             : "+v"(s.x[i]) : "v"(s.m))
DEF(PkFma,     "v_pk_fma_f32 %0, %0, %1, %2",          : "+v"(s.p[i]) : "v"(s.pm), "v"(s.pc))
DEF(PkMul,     "v_pk_mul_f32 %0, %0, %1",              : "+v"(s.p[i]) : "v"(s.pm))
DEF(PkAdd,     "v_pk_add_f32 %0, %0, %1",              : "+v"(s.p[i]) : "v"(s.pc))
DEF(CmpVcc,    "v_cmp_lt_f32 vcc, %0, %1",             : : "v"(s.x[i]), "v"(s.c) : "vcc")
DEF(CmpSgpr,   "v_cmp_lt_f32_e64 s[40:41], %0, %1",    : : "v"(s.x[i]), "v"(s.c) : "s40", "s41")
DEF(CmpU,      "v_cmp_lt_u32 vcc, %0, %1",             : : "v"(s.u[i]), "v"(s.ui) : "vcc")
DEF(CmpxF,     "v_cmp_class_f32 vcc, %0, %1",          : : "v"(s.x[i]), "v"(s.ui) : "vcc")
DEF(CndVcc,    "v_cndmask_b32 %0, %0, %1, vcc",        : "+v"(s.u[i]) : "v"(s.ui))
DEF(CndSgpr,   "v_cndmask_b32_e64 %0, %0, %1, s[42:43]", : "+v"(s.u[i]) : "v"(s.ui))
DEF(CndConst,  "v_cndmask_b32_e64 %0, 0, 1, s[42:43]", : "=v"(s.u[i]))
DEF(AddU,      "v_add_u32 %0, %0, %1",                 : "+v"(s.u[i]) : "v"(s.ui))
DEF(SubU,      "v_sub_u32 %0, %0, %1",                 : "+v"(s.u[i]) : "v"(s.ui))
DEF(AddCo,     "v_add_co_u32 %0, vcc, %0, %1",         : "+v"(s.u[i]) : "v"(s.ui) : "vcc")
DEF(AddcCo,    "v_addc_co_u32 %0, vcc, 0, %0, vcc",    : "+v"(s.u[i]) : : "vcc")
DEF(SubCoImm,  "v_subrev_co_u32 %0, vcc, 2, %0",       : "+v"(s.u[i]) : : "vcc")
DEF(AndB,      "v_and_b32 %0, %0, %1",                 : "+v"(s.u[i]) : "v"(s.ui))
DEF(OrB,       "v_or_b32 %0, %0, %1",                  : "+v"(s.u[i]) : "v"(s.ui))
DEF(Lshr,      "v_lshrrev_b32 %0, 1, %0",              : "+v"(s.u[i]))
DEF(Lshl,      "v_lshlrev_b32 %0, 1, %0",              : "+v"(s.u[i]))
DEF(Ashr,      "v_ashrrev_i32 %0, 31, %0",             : "+v"(s.u[i]))
DEF(Bfe,       "v_bfe_u32 %0, %0, 3, 5",               : "+v"(s.u[i]))
DEF(Add3,      "v_add3_u32 %0, %0, %1, %1",            : "+v"(s.u[i]) : "v"(s.ui))
DEF(LshlAdd,   "v_lshl_add_u32 %0, %0, 1, %1",         : "+v"(s.u[i]) : "v"(s.ui))
DEF(AddLshl,   "v_add_lshl_u32 %0, %0, %1, 1",         : "+v"(s.u[i]) : "v"(s.ui))
DEF(LshlOr,    "v_lshl_or_b32 %0, %0, 1, %1",          : "+v"(s.u[i]) : "v"(s.ui))
DEF(AndOr,     "v_and_or_b32 %0, %0, %1, %1",          : "+v"(s.u[i]) : "v"(s.ui))
DEF(Or3,       "v_or3_b32 %0, %0, %1, %1",             : "+v"(s.u[i]) : "v"(s.ui))
DEF(Alignbit,  "v_alignbit_b32 %0, %0, %1, 31",        : "+v"(s.u[i]) : "v"(s.ui))
DEF(MadU24,    "v_mad_u32_u24 %0, %0, %1, %1",         : "+v"(s.u[i]) : "v"(s.ui))
DEF(MulU24,    "v_mul_u32_u24 %0, %0, %1",             : "+v"(s.u[i]) : "v"(s.ui))
DEF(Ffbl,      "v_ffbl_b32 %0, %0",                    : "+v"(s.u[i]))
DEF(Bcnt,      "v_bcnt_u32_b32 %0, %0, %1",            : "+v"(s.u[i]) : "v"(s.ui))
DEF(Mov,       "v_mov_b32 %0, %1",                     : "=v"(s.u[i]) : "v"(s.ui))
DEF(MovS,      "v_mov_b32 %0, %1",                     : "=v"(s.u[i]) : "s"(s.sm))
DEF(Perm,      "v_perm_b32 %0, %0, %1, %1",            : "+v"(s.u[i]) : "v"(s.ui))
DEF(Rsq,       "v_rsq_f32 %0, %0",                     : "+v"(s.x[i]))
DEF(Rcp,       "v_rcp_f32 %0, %0",                     : "+v"(s.x[i]))
DEF(CvtI,      "v_cvt_i32_f32 %0, %0",                 : "+v"(s.u[i]))
DEF(Floor,     "v_floor_f32 %0, %0",                   : "+v"(s.x[i]))
DEF(LshlAdd64, "v_lshl_add_u64 %0, %0, 4, %1",         : "+v"(s.q[i]) : "v"(s.q[7 - (i & 3)]))
DEF(Bperm,     "ds_bpermute_b32 %0, %1, %0",           : "+v"(s.u[i]) : "v"(s.ui))
DEF(Readlane,  "v_readlane_b32 s44, %0, 3",            : : "v"(s.u[i]) : "s44")
DEF(MixAddFma, "v_add_f32 %0, %0, %2\n\tv_fma_f32 %1, %1, %3, %2", : "+v"(s.x[i]), "+v"(s.x[(i + 4) & 7]) : "v"(s.c), "v"(s.m))
DEF(MixAddPk,  "v_add_f32 %0, %0, %2\n\tv_pk_fma_f32 %1, %1, %3, %4", : "+v"(s.x[i]), "+v"(s.p[i]) : "v"(s.c), "v"(s.pm), "v"(s.pc))
DEF(MixAddMul, "v_add_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3", : "+v"(s.x[i]), "+v"(s.x[(i + 4) & 7]) : "v"(s.c), "v"(s.m))
DEF(MixAddU,   "v_add_f32 %0, %0, %2\n\tv_add_u32 %1, %1, %3", : "+v"(s.x[i]), "+v"(s.u[i]) : "v"(s.c), "v"(s.ui))

DEF(CndVccS,   "v_cndmask_b32 %0, %0, %1, vcc",        : "+v"(s.u[i]) : "v"(s.ui))
DEF(CndVcc64,  "v_cndmask_b32_e64 %0, %0, %1, vcc",    : "+v"(s.u[i]) : "v"(s.ui))
DEF(CmpCndVcc, "v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc", : "+v"(s.u[i]) : "v"(s.x[i]), "v"(s.c), "v"(s.ui) : "vcc")
DEF(CmpCndSg,  "v_cmp_lt_f32_e64 s[40:41], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[40:41]", : "+v"(s.u[i]) : "v"(s.x[i]), "v"(s.c), "v"(s.ui) : "s40", "s41")
DEF(AshrAnd,   "v_ashrrev_i32 %1, 31, %1\n\tv_and_b32 %0, %0, %1", : "+v"(s.u[i]), "+v"(s.u[(i + 4) & 7]))
DEF(MovDpp,    "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", : "=v"(s.u[i]) : "v"(s.ui))
DEF(AddDpp,    "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", : "+v"(s.x[i]) : "v"(s.c))

template <class OP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters, float sa, float sb) {
    St s;
    for (int i = 0; i < 8; ++i) { s.x[i] = a + i + threadIdx.x * 1e-3f; s.p[i] = v2f{a + i, b - i}; s.u[i] = threadIdx.x + i; s.q[i] = threadIdx.x + i; }
    s.m = a; s.c = b; s.pm = v2f{a, b}; s.pc = v2f{b, a}; s.ui = (uint32_t)iters | 1u;
    s.sm = __builtin_amdgcn_readfirstlane(sa); s.sc = __builtin_amdgcn_readfirstlane(sb);
    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_mov_b64 s[42:43], vcc" : : "v"(s.x[0]), "v"(s.c) : "vcc", "s42", "s43");
    for (int it = 0; it < iters; ++it) OP::run(s);
    float r = 0; for (int i = 0; i < 8; ++i) r += s.x[i] + s.p[i].x + s.p[i].y + (float)s.u[i] + (float)s.q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static float* d;
static hipEvent_t e0, e1;
template <class OP> void bench(int perRound) {
    const int iters = 2000, blocks = 2048;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float t; (void)hipEventElapsedTime(&t, e0, e1); if (rep == 0 || t < ms) ms = t;
    }
    const double perSimd = (double)iters * 32 * perRound * 8;     // 8 waves per SIMD
    const double ns = ms * 1e6 / perSimd;
    printf("%-58s %8.3f ms %7.3f ns %6.2f cyc@2.4\n", OP::name, ms, ns, ns * 2.4);
}

int main() {
    (void)hipMalloc(&d, (size_t)2048 * 256 * sizeof(float));
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("# 2048 blocks x 256 threads (8 waves per SIMD), 64000 x n instructions per wave; best of 3; per wave-instruction per SIMD\n");
#define B1(OP) bench<OP>(1);
#define B2(OP) bench<OP>(2);
    B1(AddF) B1(FmaVVV) B1(CndVcc) B1(CndVcc64) B1(CndSgpr) B2(CmpCndVcc) B2(CmpCndSg) B2(AshrAnd) B1(MovDpp) B1(AddDpp)
    return 0;
}
