// sph_device.h -- per-particle / per-pair device arithmetic of the SPH substep (gfx950).
//
// Hand-written restatement of shaders/SPHFluid.comp:42-221, OBBConstraints.comp:31-330
// and WaveImpulse.comp:30-46 (paths relative to /root/reference/ComponentFramework).
// Every fp32 operation here is a separately rounded IEEE op except the explicit
// fmaf()s (build with -ffp-contract=off); division and sqrt are hipcc's correctly
// rounded defaults.  The operation order is part of the engine's contract: it is what
// makes results independent of the kernel variant, the tile shape and the number of
// GPUs (DESIGN.md "Numerics").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sph {

// Per-dispatch constants, derived on the host exactly once per sph_dispatch()
// (the uniform uploads of SPHFluid3D.cpp:458-506).
struct SimK {
    // SPHFluid.comp uniforms + derived kernel coefficients (:42-64)
    float h, h2, poly6C, spikyC, viscC;
    float mp6;           // mass * poly6C (density = mp6 * sum (h2-r2)^3)
    float mass, negHalfMass, rho0, halfRho0, invRho0, kgas, visc, negSigma;
    float gravx, gravy, gravz;
    float dt, maxSpeed, maxSpeed2, foamGen, invFoamRef;
    // grid (BuildGrid.comp:14-19)
    float gminx, gminy, gminz, cellSize;
    int gx, gy, gz, numCells;   // LOCAL grid (a z-slab rank: its layers plus one ghost layer each side)
    int gzGlobal, zoff;         // global z extent and global index of local layer 0 (single GPU: gz, 0)
    // OBBConstraints.comp uniforms (:20-29)
    float R[9];
    float bcx, bcy, bcz, bhx, bhy, bhz, auxx, auxy, auxz;
    float negRest, oneMinusFric;
    int shape;
    int obbDeferred;   // shape 7..14: the SPH pass skips OBB, k_obb_ext applies it afterwards
    uint32_t* slabFlags;   // z-slab engines, substeps whose following pack may use the reduced scan: the exchange's error flags (slab_check_layer_move); else nullptr
};

// flag bits packed into pos.w of the internal state
enum : uint32_t {
    F_GHOST1 = 1u,    // isGhost == 1  (SPHFluid.comp:72)
    F_GHOSTNZ = 2u,   // isGhost != 0  (OBBConstraints.comp:46, WaveImpulse.comp:36)
    F_INACTIVE = 4u,  // isActive == 0 (SPHFluid.comp:73)
    F_HALO = 8u,      // z-slab mode: read-only copy of a neighbour rank's boundary particle (never a target)
    F_DEAD = 16u      // z-slab mode: slot no longer part of this rank's set (left or stale ghost); skipped by the sort
};

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float signf(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// BuildGrid.comp:25-26 / SPHFluid.comp:86-87: cell coordinate along one axis.
__device__ __forceinline__ int cell_axis(float p, float gmin, float cellSize, int dim) {
    float q = (p - gmin) / cellSize;
    float f = floorf(q);
    f = fminf(fmaxf(f, 0.0f), (float)(dim - 1));
    return (int)f;
}

// z cell coordinate in the LOCAL grid: the global coordinate exactly as the single-domain run
// computes it (same clamp), shifted by the slab's first layer.
__device__ __forceinline__ int cell_z_global(const SimK& k, float pz) { return cell_axis(pz, k.gminz, k.cellSize, k.gzGlobal); }
__device__ __forceinline__ int cell_z_local(const SimK& k, float pz) {
    return min(max(cell_z_global(k, pz) - k.zoff, 0), k.gz - 1);
}
// z-slab engines.  Round 4: the exchange FOLLOWS a particle across up to kSlabJump cell layers in z per substep (halo copies are one layer
// deep and migrants go to the adjacent rank -- which a jump of up to kSlabJump layers still reaches while every slab is thicker than
// that -- and while the container stays put the pack looks only at the slots of the kSlabDepth = kSlabJump + 2 lowest / highest local layers:
// whatever ends in a face layer or beyond it began the substep there).  SPHFluid.comp integrates the position with the UNCAPPED velocity (v + a dt) and
// OBBConstraints.comp projects, so a violent substep can do more.  Where that makes the decomposed run differ from the single-domain
// run it is REPORTED (error bit 16 of the exchange's flags), not followed:
//   * here, for the pack's reduced scan: a particle that began the substep OUTSIDE the slot ranges the next pack reads and ends in a
//     layer that pack would have acted on (a face layer: halo copy; beyond it: migrant);
//   * in k_slab_unpack*, for migrants that land on a rank that cannot place them (slab_record_misplaced).
// k.slabFlags is set only for substeps whose following pack may use the reduced scan.
constexpr int kSlabJump = 3;                  // cell layers in z a substep may carry a particle across and still be followed exactly
constexpr int kSlabDepth = kSlabJump + 2;     // local layers (ghost layer included) at each end of a slab that the pack and the face launches cover
__device__ __forceinline__ void slab_check_layer_move(const SimK& k, int czEntryLocal, float pzNew) {
    if (!k.slabFlags) return;                                      // (kernel-uniform)
    if (czEntryLocal < kSlabDepth || czEntryLocal > k.gz - kSlabDepth - 1) return;       // the pack reads this slot anyway
    // half a cell of margin around the exact test, so that nearly every particle leaves after two comparisons
    const float zLo = fmaf((float)(k.zoff + 2) + 0.5f, k.cellSize, k.gminz), zHi = fmaf((float)(k.zoff + k.gz - 2) - 0.5f, k.cellSize, k.gminz);
    if (pzNew > zLo && pzNew < zHi) return;
    const int czNew = cell_z_global(k, pzNew) - k.zoff;
    if (czNew <= 1 || czNew >= k.gz - 2) atomicOr(k.slabFlags, 16u);
}

// ======================= SPHFluid.comp main(), per-particle / per-pair arithmetic =======================
// Written once as templates over T = float (one target per thread: k_sph_slow, k_sph_ll, exact fallbacks)
// and T = v2f (two targets per lane, one v_pk_*_f32 per operation: k_sph_list).  Packed fp32 operations
// round each half exactly like the scalar instruction, so both instantiations give the same bits.
//
// ARITHMETIC CONTRACT (shared with oracle/sph_oracle.c o_sph_one, DESIGN.md "Numerics"): the sums, the
// candidate order and the accept tests are SPHFluid.comp's; a few quantities are formed differently so that
// the per-pair work is add / mul / fma only:
//   1/sqrt(x) = t_rsqrt(x): integer seed + three Newton steps (<= 1 ulp measured, GLSL allows 2 ULP);
//   sqrt(x) = x * t_rsqrt(x);  1/x = t_rsqrt(x)^2 for the particle's own density and the XSPH norm;
//   density = (mass*poly6C) * sum (h2 - r2)^3;  the XSPH quotient drops the poly6 coefficient;
//   spikyGrad = (spikyC (h-r)^2 / r) * rij;  viscosity term (vj - vi) * ((m/rho_j) * lapW);
//   `r < h` is tested as r2 < h2 and `length(v) > maxSpeed` as |v|^2 > maxSpeed^2.
// All pair functions are branch-free: a rejected candidate is evaluated with 1/rho_j := 0 (or weight 0) and
// adds exactly +-0, which never changes an accumulator (accumulators start at +0 and cannot reach -0).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef int32_t v2i __attribute__((ext_vector_type(2)));

template <class T> struct VecOf;
template <> struct VecOf<float> { typedef uint32_t U; typedef int32_t I; };
template <> struct VecOf<v2f> { typedef v2u U; typedef v2i I; };

template <class T> __device__ __forceinline__ T t_splat(float x);
template <> __device__ __forceinline__ float t_splat<float>(float x) { return x; }
template <> __device__ __forceinline__ v2f t_splat<v2f>(float x) { v2f r = {x, x}; return r; }
template <class T> __device__ __forceinline__ T t_fma(T a, T b, T c) { return __builtin_elementwise_fma(a, b, c); }
template <class T> __device__ __forceinline__ T t_max(T a, T b) { return __builtin_elementwise_max(a, b); }
template <class T> __device__ __forceinline__ T t_dot3(T ax, T ay, T az, T bx, T by, T bz) { return t_fma(az, bz, t_fma(ay, by, ax * bx)); }
// all-ones / all-zeros masks (per half for v2f)
__device__ __forceinline__ int32_t t_lt(float a, float b) { return a < b ? -1 : 0; }
__device__ __forceinline__ int32_t t_gt(float a, float b) { return a > b ? -1 : 0; }
__device__ __forceinline__ v2i t_lt(v2f a, v2f b) { v2i r = {a.x < b.x ? -1 : 0, a.y < b.y ? -1 : 0}; return r; }
__device__ __forceinline__ v2i t_gt(v2f a, v2f b) { v2i r = {a.x > b.x ? -1 : 0, a.y > b.y ? -1 : 0}; return r; }
__device__ __forceinline__ float t_and(float x, int32_t m) { return __uint_as_float(__float_as_uint(x) & (uint32_t)m); }
__device__ __forceinline__ v2f t_and(v2f x, v2i m) { v2f r = {t_and(x.x, m.x), t_and(x.y, m.y)}; return r; }
__device__ __forceinline__ float t_sel(int32_t m, float a, float b) { return m ? a : b; }
__device__ __forceinline__ v2f t_sel(v2i m, v2f a, v2f b) { v2f r = {m.x ? a.x : b.x, m.y ? a.y : b.y}; return r; }

// 1/sqrt(x) for normal x > 0: same operations as sph_oracle_rsqrt.
__device__ __forceinline__ float t_rsqrt(float x) {
    float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(x) >> 1));
    const float xh = 0.5f * x;
#pragma unroll
    for (int it = 0; it < 3; ++it) { const float t = y * y; const float e = fmaf(-xh, t, 0.5f); y = fmaf(y, e, y); }
    return y;
}
__device__ __forceinline__ v2f t_rsqrt(v2f x) {
    v2f y = {__uint_as_float(0x5f3759dfu - (__float_as_uint(x.x) >> 1)), __uint_as_float(0x5f3759dfu - (__float_as_uint(x.y) >> 1))};
    const v2f xh = 0.5f * x;
    const v2f half = {0.5f, 0.5f};
#pragma unroll
    for (int it = 0; it < 3; ++it) { const v2f t = y * y; const v2f e = t_fma(-xh, t, half); y = t_fma(y, e, y); }
    return y;
}
#define SPH_TINY 1e-30f

// Own state of one particle (T = float) or of the two particles of a lane (T = v2f).
template <class T>
struct OwnT {
    T px, py, pz;       // position (entry, then integrated)
    T vx, vy, vz;       // velocity
    T rho, prs;         // density / pressure of THIS substep (:106-111)
    T dsum;             // sweep-1 accumulator: sum of (h2 - r2)^3
    T fPx, fPy, fPz, fVx, fVy, fVz, gCx, gCy, gCz, lapC;   // sweep-2 accumulators
    T xsx, xsy, xsz, norm;                                 // sweep-3 accumulators
    T ax, ay, az;       // acceleration written to the record (:167)
};
typedef OwnT<float> Own;

template <class T>
__device__ __forceinline__ void own_reset(OwnT<T>& o) {
    const T z = t_splat<T>(0.0f);
    o.dsum = z;
    o.fPx = o.fPy = o.fPz = o.fVx = o.fVy = o.fVz = o.gCx = o.gCy = o.gCz = o.lapC = z;
    o.xsx = o.xsy = o.xsz = o.norm = z;
    o.ax = o.ay = o.az = z;
}

// ---- sweep 1, SPHFluid.comp:90-106: density (self included); r2 >= h2 adds +0 ----------
template <class T, class M>
__device__ __forceinline__ void pair_density(const SimK& k, OwnT<T>& o, T jx, T jy, T jz, M ok) {
    const T dx = o.px - jx, dy = o.py - jy, dz = o.pz - jz;
    const T r2 = t_dot3(dx, dy, dz, dx, dy, dz);
    const T t = t_and(t_max(t_splat<T>(k.h2) - r2, t_splat<T>(0.0f)), ok);
    o.dsum = t_fma(t * t, t, o.dsum);
}
template <class T>
__device__ __forceinline__ void finish_density(const SimK& k, OwnT<T>& o) {
    o.rho = t_max(k.mp6 * o.dsum, t_splat<T>(k.halfRho0));                       // :106
    o.prs = t_max(k.kgas * (o.rho - t_splat<T>(k.rho0)), t_splat<T>(0.0f));     // :111
}

// ---- sweep 2, :113-155: pressure / viscosity / colour field.  `ok` = candidate is another particle of
// the target's 27-cell stencil; invRho = 1 / rho_j (0 for rho_j <= 0), formed once per neighbour. ----------
template <class T, class M>
__device__ __forceinline__ void pair_force(const SimK& k, OwnT<T>& o, T jx, T jy, T jz, T jvx, T jvy, T jvz, T jprs, T invRho, M ok) {
    const T dx = o.px - jx, dy = o.py - jy, dz = o.pz - jz;
    const T r2 = t_dot3(dx, dy, dz, dx, dy, dz);
    const M acc = ok & t_lt(r2, t_splat<T>(k.h2)) & t_gt(invRho, t_splat<T>(0.0f));
    const T ie = t_and(invRho, acc);
    const T rinv = t_rsqrt(t_max(r2, t_splat<T>(SPH_TINY)));
    const T r = r2 * rinv;
    const T hr = t_splat<T>(k.h) - r;
    const T sr = (k.spikyC * (hr * hr)) * rinv;             // spikyGrad :50-57 = sr * rij
    const T gx = sr * dx, gy = sr * dy, gz = sr * dz;
    const T mor = k.mass * ie;
    const T pterm = ((o.prs + jprs) * k.negHalfMass) * ie;
    const T ml = mor * (k.viscC * hr);                      // viscLaplacian :58-64 times m / rho_j
    o.fPx = t_fma(gx, pterm, o.fPx); o.fPy = t_fma(gy, pterm, o.fPy); o.fPz = t_fma(gz, pterm, o.fPz);
    o.fVx = t_fma(jvx - o.vx, ml, o.fVx); o.fVy = t_fma(jvy - o.vy, ml, o.fVy); o.fVz = t_fma(jvz - o.vz, ml, o.fVz);
    o.gCx = t_fma(mor, gx, o.gCx); o.gCy = t_fma(mor, gy, o.gCy); o.gCz = t_fma(mor, gz, o.gCz);
    o.lapC = o.lapC + ml;
}

// ---- :157-171: surface tension, gravity, integrate -----------------------------------
template <class T>
__device__ __forceinline__ void integrate(const SimK& k, OwnT<T>& o) {
    const T gl2 = t_dot3(o.gCx, o.gCy, o.gCz, o.gCx, o.gCy, o.gCz);
    const T sc = t_and(((k.negSigma * o.lapC) * t_rsqrt(t_max(gl2, t_splat<T>(SPH_TINY)))), t_gt(gl2, t_splat<T>(1e-12f)));
    const T fSx = sc * o.gCx, fSy = sc * o.gCy, fSz = sc * o.gCz;
    const T rr = t_rsqrt(t_max(o.rho, t_splat<T>(SPH_TINY)));
    const T invRhoI = rr * rr;
    T t;
    t = t_fma(t_splat<T>(k.visc), o.fVx, o.fPx); t = t + k.gravx * o.rho; t = t + fSx; o.ax = t * invRhoI;
    t = t_fma(t_splat<T>(k.visc), o.fVy, o.fPy); t = t + k.gravy * o.rho; t = t + fSy; o.ay = t * invRhoI;
    t = t_fma(t_splat<T>(k.visc), o.fVz, o.fPz); t = t + k.gravz * o.rho; t = t + fSz; o.az = t * invRhoI;
    const T dt = t_splat<T>(k.dt);
    o.vx = t_fma(o.ax, dt, o.vx) * 0.995f; o.vy = t_fma(o.ay, dt, o.vy) * 0.995f; o.vz = t_fma(o.az, dt, o.vz) * 0.995f;
    o.px = t_fma(o.vx, dt, o.px); o.py = t_fma(o.vy, dt, o.py); o.pz = t_fma(o.vz, dt, o.pz);
}

// ---- sweep 3, :177-201: XSPH against the neighbours' ENTRY state ----------------------
template <class T, class M>
__device__ __forceinline__ void pair_xsph(const SimK& k, OwnT<T>& o, T jx, T jy, T jz, T jvx, T jvy, T jvz, T invRho, M ok) {
    const T dx = o.px - jx, dy = o.py - jy, dz = o.pz - jz;
    const T r2 = t_dot3(dx, dy, dz, dx, dy, dz);
    const M acc = ok & t_gt(invRho, t_splat<T>(0.0f));
    const T t = t_and(t_max(t_splat<T>(k.h2) - r2, t_splat<T>(0.0f)), acc);
    const T w3 = (t * t) * t;
    const T wm = w3 * (k.mass * invRho);
    o.xsx = t_fma(jvx - o.vx, wm, o.xsx); o.xsy = t_fma(jvy - o.vy, wm, o.xsy); o.xsz = t_fma(jvz - o.vz, wm, o.xsz);
    o.norm = o.norm + w3;
}

// ---- :200-217: XSPH blend, velocity cap, foam; returns the new padA -------------------
template <class T>
__device__ __forceinline__ T finish_particle(const SimK& k, OwnT<T>& o, T foamIn) {
    const T nr = t_rsqrt(t_max(o.norm, t_splat<T>(SPH_TINY)));
    const T ninv = t_sel(t_gt(o.norm, t_splat<T>(0.0f)), nr * nr, t_splat<T>(1.0f));   // norm == 0: the sums are 0 as well
    o.xsx = o.xsx * ninv; o.xsy = o.xsy * ninv; o.xsz = o.xsz * ninv;
    const T c = t_splat<T>(0.12f);
    o.vx = t_fma(c, o.xsx, o.vx); o.vy = t_fma(c, o.xsy, o.vy); o.vz = t_fma(c, o.xsz, o.vz);
    const T sp2 = t_dot3(o.vx, o.vy, o.vz, o.vx, o.vy, o.vz);
    const T f = t_sel(t_gt(sp2, t_splat<T>(k.maxSpeed2)), k.maxSpeed * t_rsqrt(t_max(sp2, t_splat<T>(SPH_TINY))), t_splat<T>(1.0f));
    o.vx = o.vx * f; o.vy = o.vy * f; o.vz = o.vz * f;
    const T s2 = t_dot3(o.vx, o.vy, o.vz, o.vx, o.vy, o.vz);
    const T speed = s2 * t_rsqrt(t_max(s2, t_splat<T>(SPH_TINY)));
    const T one = t_splat<T>(1.0f), zero = t_splat<T>(0.0f);
    const T a0 = __builtin_elementwise_min(t_max((t_splat<T>(k.rho0) - o.rho) * k.invRho0, zero), one);
    const T a1 = __builtin_elementwise_min(t_max(speed * k.invFoamRef, zero), one);
    return t_max((a0 * a1) * k.foamGen, foamIn * 0.995f);
}

// ---- OBBConstraints.comp ---------------------------------------------------------------
__device__ __forceinline__ void matvec(const float* R, float vx, float vy, float vz, float& ox, float& oy, float& oz) {
    ox = fmaf(R[6], vz, fmaf(R[3], vy, R[0] * vx));
    oy = fmaf(R[7], vz, fmaf(R[4], vy, R[1] * vx));
    oz = fmaf(R[8], vz, fmaf(R[5], vy, R[2] * vx));
}

// Shape projection (:60-143 shapes 1..6, :297-309 box).  Returns hit; q = projected
// point, n = outward normal, both in container-local space.
__device__ __forceinline__ bool shape_project(const SimK& k, float px, float py, float pz,
                                              float& qx, float& qy, float& qz, float& nx, float& ny, float& nz) {
    qx = px; qy = py; qz = pz; nx = ny = nz = 0.0f;
    switch (k.shape) {
    case 1: {                                               // sphere
        float R = k.bhx;
        float d = sqrtf(dot3(px, py, pz, px, py, pz));
        if (d > R) {
            if (d > 1e-6f) { nx = px / d; ny = py / d; nz = pz / d; } else { nx = 0.0f; ny = 1.0f; nz = 0.0f; }
            qx = nx * R; qy = ny * R; qz = nz * R;
            return true;
        }
        return false;
    }
    case 2: {                                               // cylinder
        float R = k.bhx, H = k.bhy;
        float rad = sqrtf(fmaf(pz, pz, px * px));
        float cxq = px, czq = pz;
        if (rad > R) { float s = R / fmaxf(rad, 1e-6f); cxq = px * s; czq = pz * s; }
        qx = cxq; qy = clampf(py, -H, H); qz = czq;
        float ex = px - qx, ey = py - qy, ez = pz - qz;
        float dl = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
        if (dl > 1e-6f) { nx = ex / dl; ny = ey / dl; nz = ez / dl; return true; }
        return false;
    }
    case 3: {                                               // torus
        float R = k.bhx, r = k.bhy;
        float lxz = sqrtf(fmaf(pz, pz, px * px));
        float rdx = 1.0f, rdz = 0.0f;
        if (lxz > 1e-6f) { rdx = px / lxz; rdz = pz / lxz; }
        float gxr = rdx * R, gyr = 0.0f, gzr = rdz * R;
        float ex = px - gxr, ey = py - gyr, ez = pz - gzr;
        float dl = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
        if (dl > r) {
            float m = fmaxf(dl, 1e-6f);
            nx = ex / m; ny = ey / m; nz = ez / m;
            qx = gxr + nx * r; qy = gyr + ny * r; qz = gzr + nz * r;
            return true;
        }
        return false;
    }
    case 4: {                                               // capsule
        float R = k.bhx, H = k.bhy;
        float sx = 0.0f, sy = clampf(py, -H, H), sz = 0.0f;
        float ex = px - sx, ey = py - sy, ez = pz - sz;
        float dl = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
        if (dl > R) {
            float m = fmaxf(dl, 1e-6f);
            nx = ex / m; ny = ey / m; nz = ez / m;
            qx = sx + nx * R; qy = sy + ny * R; qz = sz + nz * R;
            return true;
        }
        return false;
    }
    case 5: {                                               // hourglass
        float baseR = k.bhx, H = fmaxf(k.bhy, 1e-6f), neckR = fminf(k.bhz, baseR);
        float yC = clampf(py, -H, H);
        float rMax = neckR + ((baseR - neckR) * fabsf(yC)) / H;
        float lxz = sqrtf(fmaf(pz, pz, px * px));
        float cxq = px, czq = pz;
        if (lxz > rMax) { float s = rMax / fmaxf(lxz, 1e-6f); cxq = px * s; czq = pz * s; }
        qx = cxq; qy = yC; qz = czq;
        float ex = px - qx, ey = py - qy, ez = pz - qz;
        float dl = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
        if (dl > 1e-6f) { nx = ex / dl; ny = ey / dl; nz = ez / dl; return true; }
        return false;
    }
    case 6: {                                               // egg
        float a = fmaxf(k.bhx, 1e-6f), b = fmaxf(k.bhy, 1e-6f);
        float ux = px / a, uy = py / b, uz = pz / a;
        float d = sqrtf(dot3(ux, uy, uz, ux, uy, uz));
        if (d > 1.0f) {
            qx = (ux / d) * a; qy = (uy / d) * b; qz = (uz / d) * a;
            float gx = qx / (a * a), gy = qy / (b * b), gz = qz / (a * a);
            float gl = sqrtf(dot3(gx, gy, gz, gx, gy, gz));
            nx = gx / gl; ny = gy / gl; nz = gz / gl;
            return true;
        }
        return false;
    }
    default: {                                              // box
        qx = clampf(px, -k.bhx, k.bhx); qy = clampf(py, -k.bhy, k.bhy); qz = clampf(pz, -k.bhz, k.bhz);
        float ex = px - qx, ey = py - qy, ez = pz - qz;
        float ax = fabsf(ex), ay = fabsf(ey), az = fabsf(ez);
        if (ax > 0.0f || ay > 0.0f || az > 0.0f) {
            if (ax >= ay && ax >= az) nx = signf(ex);
            else if (ay >= ax && ay >= az) ny = signf(ey);
            else nz = signf(ez);
            return true;
        }
        return false;
    }
    }
}

// worldToLocal :32-36
__device__ __forceinline__ void obb_to_local(const SimK& k, float px, float py, float pz, float& lx, float& ly, float& lz) {
    float dx = px - k.bcx, dy = py - k.bcy, dz = pz - k.bcz;
    lx = dot3(dx, dy, dz, k.R[0], k.R[1], k.R[2]);
    ly = dot3(dx, dy, dz, k.R[3], k.R[4], k.R[5]);
    lz = dot3(dx, dy, dz, k.R[6], k.R[7], k.R[8]);
}
// Collision response :311-327 for a hit with local projected point q and local normal n.
__device__ __forceinline__ void obb_respond(const SimK& k, float qx, float qy, float qz, float nx, float ny, float nz,
                                            float& px, float& py, float& pz, float& vx, float& vy, float& vz) {
    float wx, wy, wz, tx, ty, tz;
    matvec(k.R, nx, ny, nz, wx, wy, wz);                // :313
    float len = sqrtf(dot3(wx, wy, wz, wx, wy, wz));
    wx = wx / len; wy = wy / len; wz = wz / len;
    matvec(k.R, qx, qy, qz, tx, ty, tz);                // :316
    px = k.bcx + tx; py = k.bcy + ty; pz = k.bcz + tz;
    float vn = dot3(vx, vy, vz, wx, wy, wz);            // :319-326
    float nvx = vn * wx, nvy = vn * wy, nvz = vn * wz;
    vx = k.negRest * nvx + k.oneMinusFric * (vx - nvx);
    vy = k.negRest * nvy + k.oneMinusFric * (vy - nvy);
    vz = k.negRest * nvz + k.oneMinusFric * (vz - nvz);
}

// OBBConstraints.comp main() for one non-ghost particle: updates pos / vel in place.
// Shapes 7..14 are applied by their own pass (sph_shapes_ext.h, k.obbDeferred).
__device__ __forceinline__ void obb_apply(const SimK& k, float& px, float& py, float& pz, float& vx, float& vy, float& vz) {
    if (k.obbDeferred) return;
    float lx, ly, lz, qx, qy, qz, nx, ny, nz;
    obb_to_local(k, px, py, pz, lx, ly, lz);
    if (shape_project(k, lx, ly, lz, qx, qy, qz, nx, ny, nz)) obb_respond(k, qx, qy, qz, nx, ny, nz, px, py, pz, vx, vy, vz);
}

// Fully specified fp32 sine shared with the parity oracle's definition
// (DESIGN.md "Numerics", item sin): Cody-Waite pi/2 reduction + fixed polynomials.
__device__ __forceinline__ float sph_sinf(float x) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float P1 = 1.5703125f, P2 = 4.837512969970703125e-4f, P3 = 7.549789948768648e-8f;
    float q = rintf(x * TWO_OVER_PI);
    float r = fmaf(q, -P1, x);
    r = fmaf(q, -P2, r);
    r = fmaf(q, -P3, r);
    int n = (int)(q - 4.0f * floorf(q * 0.25f));
    float r2 = r * r;
    float res;
    if (n & 1) {
        float c = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
        c = fmaf(c, r2, 4.166664568298827e-2f);
        c = fmaf(c, r2, -0.5f);
        res = fmaf(c, r2, 1.0f);
    } else {
        float s = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
        s = fmaf(s, r2, -1.6666654611e-1f);
        s = s * r2;
        res = fmaf(s, r, r);
    }
    return (n & 2) ? -res : res;
}

}  // namespace sph
