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
    float h2hi;   // h2 * (1 + 1e-6): r2 >= h2hi implies sqrt(r2) >= h, a cheap superset filter for the force sweep
    float mass, negMass, rho0, halfRho0, kgas, visc, negSigma;
    float gravx, gravy, gravz;
    float dt, maxSpeed, foamGen, foamVelRefMax;
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

// Own state of one particle while it runs through SPHFluid.comp main().
struct Own {
    float px, py, pz;       // position (entry, then integrated)
    float vx, vy, vz;       // velocity
    float rho, prs;         // density / pressure of THIS substep (:106-111)
    float dens;             // sweep-1 accumulator
    float fPx, fPy, fPz, fVx, fVy, fVz, gCx, gCy, gCz, lapC;   // sweep-2 accumulators
    float xsx, xsy, xsz, norm;                                 // sweep-3 accumulators
    float ax, ay, az;       // acceleration written to the record (:167)
};

// ---- sweep 1, SPHFluid.comp:90-106: density (self included) -------------------------
__device__ __forceinline__ void pair_density(const SimK& k, Own& o, float jx, float jy, float jz) {
    float dx = o.px - jx, dy = o.py - jy, dz = o.pz - jz;
    float r2 = dot3(dx, dy, dz, dx, dy, dz);
    if (r2 < k.h2) {
        float t = k.h2 - r2;
        float w = k.poly6C * ((t * t) * t);
        o.dens = fmaf(k.mass, w, o.dens);
    }
}
__device__ __forceinline__ void finish_density(const SimK& k, Own& o) {
    o.rho = fmaxf(o.dens, k.halfRho0);                      // :106
    o.prs = fmaxf(k.kgas * (o.rho - k.rho0), 0.0f);         // :111
}

// ---- sweep 2, :113-155: pressure / viscosity / colour field (self skipped by caller) -
// invRho = 1 / rho_j depends on the neighbour only, so callers may form it once per staged
// neighbour instead of once per pair (same bits either way).  Reciprocal forms fixed by the
// numerics contract: mass / rho_j = mass * invRho, x / (2 rho_j) = (x * 0.5) *
// invRho, rij / r = rij * (1/r).
__device__ __forceinline__ void pair_force_pre(const SimK& k, Own& o, float jx, float jy, float jz,
                                               float jvx, float jvy, float jvz, float jprs, float invRho) {
    float dx = o.px - jx, dy = o.py - jy, dz = o.pz - jz;
    float r = sqrtf(dot3(dx, dy, dz, dx, dy, dz));
    if (r < k.h) {
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;              // spikyGrad :50-57
        float hr = k.h - r;
        if (r > 0.0f) {
            float invr = 1.0f / r;
            float s = k.spikyC * (hr * hr);
            gx = s * (dx * invr); gy = s * (dy * invr); gz = s * (dz * invr);
        }
        float pterm = ((k.negMass * (o.prs + jprs)) * 0.5f) * invRho;
        float mor = k.mass * invRho;
        float lapW = k.viscC * hr;                          // viscLaplacian :58-64
        o.fPx = fmaf(gx, pterm, o.fPx); o.fPy = fmaf(gy, pterm, o.fPy); o.fPz = fmaf(gz, pterm, o.fPz);
        o.fVx = fmaf((jvx - o.vx) * mor, lapW, o.fVx);
        o.fVy = fmaf((jvy - o.vy) * mor, lapW, o.fVy);
        o.fVz = fmaf((jvz - o.vz) * mor, lapW, o.fVz);
        o.gCx = fmaf(mor, gx, o.gCx); o.gCy = fmaf(mor, gy, o.gCy); o.gCz = fmaf(mor, gz, o.gCz);
        o.lapC = fmaf(mor, lapW, o.lapC);
    }
}
__device__ __forceinline__ void pair_force(const SimK& k, Own& o, float jx, float jy, float jz,
                                           float jvx, float jvy, float jvz, float jrho, float jprs) {
    if (jrho > 0.0f) pair_force_pre(k, o, jx, jy, jz, jvx, jvy, jvz, jprs, 1.0f / jrho);
}

// ---- :157-171: surface tension, gravity, integrate -----------------------------------
__device__ __forceinline__ void integrate(const SimK& k, Own& o) {
    float fSx = 0.0f, fSy = 0.0f, fSz = 0.0f;
    float gl = sqrtf(dot3(o.gCx, o.gCy, o.gCz, o.gCx, o.gCy, o.gCz));
    if (gl > 1e-6f) {
        float sc = k.negSigma * o.lapC;
        fSx = sc * (o.gCx / gl); fSy = sc * (o.gCy / gl); fSz = sc * (o.gCz / gl);
    }
    float t;
    t = fmaf(k.visc, o.fVx, o.fPx); t = t + k.gravx * o.rho; t = t + fSx; o.ax = t / o.rho;
    t = fmaf(k.visc, o.fVy, o.fPy); t = t + k.gravy * o.rho; t = t + fSy; o.ay = t / o.rho;
    t = fmaf(k.visc, o.fVz, o.fPz); t = t + k.gravz * o.rho; t = t + fSz; o.az = t / o.rho;
    o.vx = fmaf(o.ax, k.dt, o.vx) * 0.995f; o.vy = fmaf(o.ay, k.dt, o.vy) * 0.995f; o.vz = fmaf(o.az, k.dt, o.vz) * 0.995f;
    o.px = fmaf(o.vx, k.dt, o.px); o.py = fmaf(o.vy, k.dt, o.py); o.pz = fmaf(o.vz, k.dt, o.pz);
}

// ---- sweep 3, :177-201: XSPH against the neighbours' ENTRY state ----------------------
__device__ __forceinline__ void pair_xsph_pre(const SimK& k, Own& o, float jx, float jy, float jz,
                                              float jvx, float jvy, float jvz, float invRho) {
    float dx = o.px - jx, dy = o.py - jy, dz = o.pz - jz;
    float r2 = dot3(dx, dy, dz, dx, dy, dz);
    if (r2 < k.h2) {
        float t = k.h2 - r2;
        float w = k.poly6C * ((t * t) * t);
        float mor = k.mass * invRho;
        o.xsx = fmaf((jvx - o.vx) * w, mor, o.xsx);
        o.xsy = fmaf((jvy - o.vy) * w, mor, o.xsy);
        o.xsz = fmaf((jvz - o.vz) * w, mor, o.xsz);
        o.norm = o.norm + w;
    }
}
__device__ __forceinline__ void pair_xsph(const SimK& k, Own& o, float jx, float jy, float jz,
                                          float jvx, float jvy, float jvz, float jrho) {
    if (jrho > 0.0f) pair_xsph_pre(k, o, jx, jy, jz, jvx, jvy, jvz, 1.0f / jrho);
}

// ---- :200-217: XSPH blend, velocity cap, foam; returns the new padA -------------------
__device__ __forceinline__ float finish_particle(const SimK& k, Own& o, float foamIn) {
    if (o.norm > 0.0f) { o.xsx = o.xsx / o.norm; o.xsy = o.xsy / o.norm; o.xsz = o.xsz / o.norm; }
    o.vx = fmaf(0.12f, o.xsx, o.vx); o.vy = fmaf(0.12f, o.xsy, o.vy); o.vz = fmaf(0.12f, o.xsz, o.vz);
    float sp = sqrtf(dot3(o.vx, o.vy, o.vz, o.vx, o.vy, o.vz));
    if (sp > k.maxSpeed) {
        float f = k.maxSpeed / sp;
        o.vx = o.vx * f; o.vy = o.vy * f; o.vz = o.vz * f;
    }
    float speed = sqrtf(dot3(o.vx, o.vy, o.vz, o.vx, o.vy, o.vz));
    float aer = clampf((k.rho0 - o.rho) / k.rho0, 0.0f, 1.0f) * clampf(speed / k.foamVelRefMax, 0.0f, 1.0f);
    return fmaxf(aer * k.foamGen, foamIn * 0.995f);
}

__device__ __forceinline__ void own_reset(Own& o) {
    o.dens = 0.0f;
    o.fPx = o.fPy = o.fPz = o.fVx = o.fVy = o.fVz = o.gCx = o.gCy = o.gCz = o.lapC = 0.0f;
    o.xsx = o.xsy = o.xsz = o.norm = 0.0f;
    o.ax = o.ay = o.az = 0.0f;
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
