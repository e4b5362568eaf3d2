// sph_host.h -- host-side scalar logic of SPHFluidGPU that needs no device:
// defaults, MakeRotationMat3XYZ, EffectiveHalf, ComputeGridExtents, the standard-fill
// spawn and the per-dispatch constant derivation.  Citations are relative to
// /root/reference/ComponentFramework.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/sph_abi.h"
#include "sph_device.h"

namespace sph {

inline void params_default(SphParams& p) {            // SPHFluid3D.h:94-124
    std::memset(&p, 0, sizeof(p));
    p.param_h = 0.28f;
    p.param_mass = 13.8f;
    p.param_restDensity = 1000.0f;
    p.param_gasConstant = 2000.0f;
    p.param_viscosity = 3.5f;
    p.param_gravityY = -980.0f;
    p.param_gravityX = 0.0f;
    p.param_gravityZ = 0.0f;
    p.param_surfaceTension = 0.0728f;
    p.param_timeStep = 0.001f;
    p.param_pause = 0;
    p.param_useJitter = 1;
    p.param_jitterAmp = 0.20f;
    p.param_foamGen = 1.0f;
    p.param_foamVelRef = 8.0f;
    p.param_boxHalf[0] = p.param_boxHalf[1] = p.param_boxHalf[2] = 7.0f;
    p.param_shapeType = 0;
    p.param_shapeAux[0] = 5.0f; p.param_shapeAux[1] = 0.35f; p.param_shapeAux[2] = 2.5f;
    p.param_wallRestitution = 0.15f;
    p.param_wallFriction = 0.02f;
    p.grid_cap = 160;                                  // SPHFluid3D.cpp:370
}

// SPHFluid3D.cpp:13-30.  outM is column-major world_from_box, R = Rz * Ry * Rx.
inline void rotation_mat3(const float e[3], float outM[9]) {
    const float d2r = float(3.14159265358979323846 / 180.0);
    const float rx = e[0] * d2r, ry = e[1] * d2r, rz = e[2] * d2r;
    const float cx = std::cos(rx), sx = std::sin(rx);
    const float cy = std::cos(ry), sy = std::sin(ry);
    const float cz = std::cos(rz), sz = std::sin(rz);
    const float Z[9] = {cz, sz, 0, -sz, cz, 0, 0, 0, 1};
    const float Y[9] = {cy, 0, -sy, 0, 1, 0, sy, 0, cy};
    const float X[9] = {1, 0, 0, 0, cx, sx, 0, -sx, cx};
    auto mul = [](const float* A, const float* B, float* Cm) {
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r)
                Cm[c * 3 + r] = A[r] * B[c * 3] + A[3 + r] * B[c * 3 + 1] + A[6 + r] * B[c * 3 + 2];
    };
    float ZY[9];
    mul(Z, Y, ZY);
    mul(ZY, X, outM);
}

// SPHFluid3D.h:127-158
inline void effective_half(const SphParams& p, float o[3]) {
    const float x = p.param_boxHalf[0], y = p.param_boxHalf[1], z = p.param_boxHalf[2];
    const float a0 = p.param_shapeAux[0], a1 = p.param_shapeAux[1];
    switch (p.param_shapeType) {
    case 1: case 13: o[0] = x; o[1] = x; o[2] = x; return;
    case 2: case 5: case 6: case 7: case 8: o[0] = x; o[1] = y; o[2] = x; return;
    case 3: o[0] = x + y; o[1] = y; o[2] = x + y; return;
    case 4: o[0] = x; o[1] = y + x; o[2] = x; return;
    case 9: o[0] = 3.0f * x + y; o[1] = 0.35f * x + y; o[2] = 3.0f * x + y; return;
    case 10: o[0] = x + y; o[1] = y + a0; o[2] = x + y; return;
    case 11: case 14: o[0] = x + y; o[1] = a1 + y; o[2] = x + y; return;
    case 12: o[0] = x + y; o[1] = 1.15f * x + y; o[2] = y + 0.3f; return;
    default: o[0] = x; o[1] = y; o[2] = z; return;
    }
}

// SPHFluid3D.cpp:354-376
inline void compute_grid_extents(const SphParams& p, SphGridInfo& g) {
    float R[9], half[3];
    g.cellSize = p.param_h;
    rotation_mat3(p.param_boxEulerDeg, R);
    effective_half(p, half);
    const int cap = p.grid_cap > 0 ? p.grid_cap : 160;
    for (int i = 0; i < 3; ++i) {
        float ext = std::fabs(R[i]) * half[0] + std::fabs(R[3 + i]) * half[1] + std::fabs(R[6 + i]) * half[2];
        ext = ext + g.cellSize;
        g.gridMin[i] = p.param_boxCenter[i] - ext;
        // clamp in float, then convert (an out-of-range float -> int conversion is undefined)
        const float df = std::ceil((2.0f * ext) / g.cellSize);
        g.dims[i] = (df >= float(cap)) ? cap : ((df >= 1.0f) ? int(df) : 1);
    }
    const long long nc = (long long)g.dims[0] * g.dims[1] * g.dims[2];
    g.numCells = nc < 1 ? 1 : (nc > 2147483647LL ? 2147483647 : (int)nc);   // engines refuse grids beyond kMaxCells
}

// Per-dispatch constants (uniform derivation of SPHFluid3D.cpp:458-506, incl. maxSpeed :488).
inline void make_simk(const SphParams& p, const SphGridInfo& g, float dt, SimK& k) {
    const float h = p.param_h;
    const float h2 = h * h, h3 = h2 * h, h6 = h3 * h3, h9 = h6 * h3;
    const float pi_f = 3.141592653589f;               // literal of SPHFluid.comp:45,53,60
    k.h = h; k.h2 = h2;
    k.poly6C = 315.0f / ((64.0f * pi_f) * h9);
    k.spikyC = -45.0f / (pi_f * h6);
    k.viscC = 45.0f / (pi_f * h6);
    k.mass = p.param_mass; k.negHalfMass = (-p.param_mass) * 0.5f;
    k.mp6 = p.param_mass * k.poly6C;
    k.rho0 = p.param_restDensity; k.halfRho0 = p.param_restDensity * 0.5f; k.invRho0 = 1.0f / p.param_restDensity;
    k.kgas = p.param_gasConstant; k.visc = p.param_viscosity; k.negSigma = -p.param_surfaceTension;
    k.gravx = p.param_gravityX; k.gravy = p.param_gravityY; k.gravz = p.param_gravityZ;
    k.dt = dt;
    k.maxSpeed = (0.4f * h) / std::fmax(dt, 1e-6f);
    k.maxSpeed2 = k.maxSpeed * k.maxSpeed;
    k.foamGen = p.param_foamGen;
    k.invFoamRef = 1.0f / std::fmax(p.param_foamVelRef, 1e-3f);
    k.gminx = g.gridMin[0]; k.gminy = g.gridMin[1]; k.gminz = g.gridMin[2];
    k.cellSize = g.cellSize;
    k.gx = g.dims[0]; k.gy = g.dims[1]; k.gz = g.dims[2]; k.numCells = g.numCells;
    k.gzGlobal = g.dims[2]; k.zoff = 0;
    rotation_mat3(p.param_boxEulerDeg, k.R);
    k.bcx = p.param_boxCenter[0]; k.bcy = p.param_boxCenter[1]; k.bcz = p.param_boxCenter[2];
    k.bhx = p.param_boxHalf[0]; k.bhy = p.param_boxHalf[1]; k.bhz = p.param_boxHalf[2];
    k.auxx = p.param_shapeAux[0]; k.auxy = p.param_shapeAux[1]; k.auxz = p.param_shapeAux[2];
    k.negRest = -p.param_wallRestitution;
    k.oneMinusFric = 1.0f - p.param_wallFriction;
    k.shape = p.param_shapeType;
    k.obbDeferred = (k.shape >= 7 && k.shape <= 14) ? 1 : 0;
    k.slabFlags = nullptr;
}

// ---- spawn -------------------------------------------------------------------------
// PCG32 (XSH-RR) replaces default_random_engine(time(nullptr)) of SPHFluid3D.cpp:99.
struct Pcg32 {
    uint64_t state = 0, inc = (54ull << 1) | 1ull;
    explicit Pcg32(uint64_t seed) { next(); state += seed; next(); }
    uint32_t next() {
        const uint64_t old = state;
        state = old * 6364136223846793005ull + inc;
        const uint32_t xs = uint32_t(((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = uint32_t(old >> 59u);
        return (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    float uniform(float lo, float hi) { return lo + (float(next() >> 8) * (1.0f / 16777216.0f)) * (hi - lo); }
};

// Sampled curves of container shapes 9 / 11 / 12 / 14: the shader evaluates them at fixed
// parameter values for every particle (OBBConstraints.comp:188-191, :232-236, :248-252,
// :288-290); here they are tabulated once per dispatch.  out: 3 floats per point, at most 128
// points; best0: the shader's initial nearest point.  Returns the point count.
inline int shape_table(const SphParams& p, float* out, float best0[3]) {
    const float hx = p.param_boxHalf[0], hy = p.param_boxHalf[1];
    int n = 0;
    best0[0] = best0[1] = best0[2] = 0.0f;
    switch (p.param_shapeType) {
    case 9: {
        const float S = hx;
        best0[0] = 3.0f * S;
        for (int k = 0; k < 48; ++k, ++n) {
            const float t = 6.2831853f * float(k) / 48.0f;
            out[3 * n + 0] = S * (sinf(t) + 2.0f * sinf(2.0f * t));
            out[3 * n + 1] = S * (0.35f * (-sinf(3.0f * t)));
            out[3 * n + 2] = S * (cosf(t) - 2.0f * cosf(2.0f * t));
        }
        break;
    }
    case 11: case 14: {
        const float R = hx, r = hy;
        const float turns = std::fmax(1.0f, p.param_shapeAux[0]), H = std::fmax(p.param_shapeAux[1], r);
        best0[0] = R; best0[1] = -H;
        for (int k = 0; k < 64; ++k) {
            const float f = float(k) / 63.0f;
            const float t = f * turns * 6.2831853f;
            const float y = (f - 0.5f) * 2.0f * H;
            out[3 * n + 0] = R * cosf(t); out[3 * n + 1] = y; out[3 * n + 2] = R * sinf(t); ++n;
            if (p.param_shapeType == 11) {
                out[3 * n + 0] = R * cosf(t + 3.14159265f); out[3 * n + 1] = y; out[3 * n + 2] = R * sinf(t + 3.14159265f); ++n;
            }
        }
        break;
    }
    case 12: {
        const float S = hx * 0.0625f;
        for (int k = 0; k < 64; ++k, ++n) {
            const float t = 6.2831853f * float(k) / 64.0f;
            const float st = sinf(t);
            const float hxx = 16.0f * st * st * st;
            const float hyy = 13.0f * cosf(t) - 5.0f * cosf(2.0f * t) - 2.0f * cosf(3.0f * t) - cosf(4.0f * t);
            out[3 * n + 0] = S * hxx; out[3 * n + 1] = S * hyy; out[3 * n + 2] = 0.0f;
        }
        break;
    }
    default: break;
    }
    return n;
}

// insideShape lambda of SPHFluid3D.cpp:167-289.
inline bool inside_shape(const SphParams& p, const float hf[3], float margin, float lx, float ly, float lz) {
    const float bx = p.param_boxHalf[0], by = p.param_boxHalf[1], bz = p.param_boxHalf[2];
    switch (p.param_shapeType) {
    case 1: { const float r = hf[0] - margin; return lx * lx + ly * ly + lz * lz <= r * r; }
    case 2: { const float r = hf[0] - margin; return lx * lx + lz * lz <= r * r && std::fabs(ly) <= hf[1] - margin; }
    case 3: { const float r = by - margin; const float dr = std::sqrt(lx * lx + lz * lz) - bx;
              return r > 0.0f && (dr * dr + ly * ly) <= r * r; }
    case 4: { const float r = bx - margin; const float cy = std::fmin(std::fmax(ly, -by), by); const float dy = ly - cy;
              return (lx * lx + lz * lz + dy * dy) <= r * r; }
    case 5: { const float H = std::fmax(by, 1e-6f), neck = std::fmin(bz, bx);
              if (std::fabs(ly) > H - margin) return false;
              const float rMax = neck + (bx - neck) * std::fabs(ly) / H - margin;
              return rMax > 0.0f && (lx * lx + lz * lz) <= rMax * rMax; }
    case 6: { const float a = std::fmax(bx - margin, 1e-4f), b = std::fmax(by - margin, 1e-4f);
              const float u = lx / a, v = ly / b, w = lz / a;
              return (u * u + v * v + w * w) <= 1.0f; }
    case 7: { const float pts = std::fmax(3.0f, p.param_shapeAux[0]), depth = std::fmin(std::fmax(p.param_shapeAux[1], 0.0f), 0.9f);
              if (std::fabs(ly) > by - margin) return false;
              const float ang = atan2f(lz, lx);
              const float rMax = bx * (1.0f - depth * (0.5f + 0.5f * cosf(pts * ang))) - margin;
              return rMax > 0.0f && (lx * lx + lz * lz) <= rMax * rMax; }
    case 8: { const float a = std::fmax(bx - margin, 1e-4f), b = std::fmax(by - margin, 1e-4f);
              const float n = std::fmin(std::fmax(p.param_shapeAux[2], 0.6f), 8.0f);
              const float F = powf(std::fabs(lx) / a, n) + powf(std::fabs(ly) / b, n) + powf(std::fabs(lz) / a, n);
              return F <= 1.0f; }
    case 9: case 11: case 12: case 14: {
              const float r = by - margin;
              if (r <= 0.0f) return false;
              float tab[384], b0[3];
              const int cnt = shape_table(p, tab, b0);
              float bestD2 = 1e30f;
              for (int k = 0; k < cnt; ++k) {
                  const float dx = lx - tab[3 * k], dy = ly - tab[3 * k + 1], dz = lz - tab[3 * k + 2];
                  bestD2 = std::fmin(bestD2, dx * dx + dy * dy + dz * dz);
              }
              return bestD2 <= r * r; }
    case 10: { const float wHalf = by - margin, tHalf = std::fmax(p.param_shapeAux[0], 0.05f) - margin;
              if (wHalf <= 0.0f || tHalf <= 0.0f) return false;
              const float phi = atan2f(lz, lx);
              const float erx = cosf(phi), erz = sinf(phi);
              const float ox = lx - bx * erx, oy = ly, oz = lz - bx * erz;
              const float psi = 0.5f * phi;
              const float cw = cosf(psi), sw = sinf(psi);
              const float du = ox * (cw * erx) + oy * (sw) + oz * (cw * erz);
              const float dv = ox * (-sw * erx) + oy * (cw) + oz * (-sw * erz);
              return std::fabs(du) <= wHalf && std::fabs(dv) <= tHalf; }
    case 13: { const float R = bx - margin;
              const float sc = std::fmax(p.param_shapeAux[0], 0.1f), th = std::fmin(std::fmax(p.param_shapeAux[1], 0.2f), 2.5f);
              if (lx * lx + ly * ly + lz * lz > R * R) return false;
              const float qx = lx * sc, qy = ly * sc, qz = lz * sc;
              const float g = sinf(qx) * cosf(qy) + sinf(qy) * cosf(qz) + sinf(qz) * cosf(qx);
              return std::fabs(g) <= th; }
    default: return true;
    }
}

// InitializeParticles, standard-fill branch (SPHFluid3D.cpp:85-102,159-332).
inline void spawn_particles(const SphParams& p, size_t nRequested, uint32_t seed,
                            std::vector<SphParticle>& out, float& massOut) {
    const float spacing = p.param_h * 0.85f;                                  // :89
    massOut = p.param_restDensity * spacing * spacing * spacing;              // :92
    const float fillFraction = 0.4f;                                          // :97
    Pcg32 rng(seed);
    const float jl = -spacing * p.param_jitterAmp, jh = spacing * p.param_jitterAmp;
    float hf[3];
    effective_half(p, hf);
    const float margin = spacing * 0.5f;
    const int layersY = std::max(1, int((2.0f * hf[1] * fillFraction) / spacing));  // :290-292
    const int sideX = std::max(1, int((hf[0] * 1.7f) / spacing));
    const int sideZ = std::max(1, int((hf[2] * 1.7f) / spacing));
    out.clear();
    for (int x = 0; x < sideX && out.size() < nRequested; ++x)
        for (int y = 0; y < layersY && out.size() < nRequested; ++y)
            for (int z = 0; z < sideZ && out.size() < nRequested; ++z) {
                const float jx = p.param_useJitter ? rng.uniform(jl, jh) : 0.0f;
                const float jy = p.param_useJitter ? rng.uniform(jl, jh) : 0.0f;
                const float jz = p.param_useJitter ? rng.uniform(jl, jh) : 0.0f;
                const float lx = -hf[0] * 0.85f + float(x) * spacing + jx;    // :296-298
                const float ly = -hf[1] + spacing + float(y) * spacing + jy;
                const float lz = -hf[2] * 0.85f + float(z) * spacing + jz;
                if (!inside_shape(p, hf, margin, lx, ly, lz)) continue;
                SphParticle q;
                std::memset(&q, 0, sizeof(q));
                q.pos[0] = p.param_boxCenter[0] + lx;
                q.pos[1] = p.param_boxCenter[1] + ly;
                q.pos[2] = p.param_boxCenter[2] + lz;
                if (p.param_mixPattern == 1) q.padC = (x + y + z) & 1;        // :307-311
                else if (p.param_mixPattern == 2) q.padC = int(rng.next() & 1u);
                else q.padC = (lx < 0.0f) ? 0 : 1;
                float d;                                                      // :316-329
                if (p.param_dyePattern == 1) d = (ly + hf[1]) / std::fmax(2.0f * hf[1], 1e-3f);
                else if (p.param_dyePattern == 2) {
                    const float nn = std::sin(lx * 1.3f) * std::cos(lz * 1.7f) + std::sin(ly * 1.1f + lx * 0.7f);
                    d = 0.5f + 0.5f * std::sin(nn * 2.3f);
                } else d = 0.5f + 0.5f * std::sin(lx * 1.5f);
                q.padB = std::fmin(std::fmax(d, 0.0f), 1.0f);
                out.push_back(q);
            }
}

// ---- river / stream mode, host side (SPHFluid3D.h:171-206) ---------------------------------------------------------
inline void river_default(SphRiver& r) {                                       // SPHFluid3D.h:172-196
    r = SphRiver{0, 64, 64, -7.0f, -10.0f, 14.0f, 20.0f, {0.0f, 3.0f, -9.0f}, {0.0f, -0.5f, 4.0f}, 1.5f, -8.5f, 9.0f,
                 2.0f, 0.25f, 0.0f, 3.0f, 3.5f, 0.3f};
}

// std::srand / std::rand of the Microsoft C runtime (the reference is built with Visual Studio): RAND_MAX = 0x7fff.
struct MsvcRand {
    uint32_t hold;
    explicit MsvcRand(unsigned seed) : hold(seed) {}
    int next() { hold = hold * 214013u + 2531011u; return int((hold >> 16) & 0x7fffu); }
    float frand() { return float(next()) / 32767.0f; }                        // std::rand() / float(RAND_MAX), :774
};

// GenerateRiverTerrain, SPHFluid3D.cpp:772-878 (without the GL upload).
inline void river_terrain(SphParams& p, int seed, SphRiver& r, float* heights) {
    MsvcRand rnd(static_cast<unsigned>(seed));
    r.riverAmp = 0.5f + rnd.frand() * 1.5f;                                    // :777-782
    r.riverFreq = 0.18f + rnd.frand() * 0.18f;
    r.riverPhase = rnd.frand() * 6.2831f;
    r.riverChannelWidth = 1.8f + rnd.frand() * 1.2f;
    r.riverChannelDepth = 3.5f + rnd.frand() * 1.0f;
    r.riverSlopeDrop = 0.3f + rnd.frand() * 0.5f;
    float ph[8];
    for (float& x : ph) x = rnd.frand() * 6.2831f;
    r.terrainWorldMinX = p.param_boxCenter[0] - p.param_boxHalf[0];            // :792-795
    r.terrainWorldMinZ = p.param_boxCenter[2] - p.param_boxHalf[2];
    r.terrainWorldSizeX = 2.0f * p.param_boxHalf[0];
    r.terrainWorldSizeZ = 2.0f * p.param_boxHalf[2];
    const float yBase = p.param_boxCenter[1] - p.param_boxHalf[1];
    const float depth = r.riverChannelDepth, drop = r.riverSlopeDrop;
    auto centerline = [&](float wz) { return p.param_boxCenter[0] + r.riverAmp * std::sin(r.riverFreq * wz + r.riverPhase); };
    for (int iz = 0; iz < r.terrainH; ++iz) {
        const float wz = r.terrainWorldMinZ + (float(iz) / float(r.terrainH - 1)) * r.terrainWorldSizeZ;
        const float tFlow = (wz - r.terrainWorldMinZ) / r.terrainWorldSizeZ;
        const float riverFloor = yBase + 1.0f - tFlow * drop;                  // :818
        const float channelEdge = riverFloor + depth;
        const float cX = centerline(wz);
        for (int ix = 0; ix < r.terrainW; ++ix) {
            const float wx = r.terrainWorldMinX + (float(ix) / float(r.terrainW - 1)) * r.terrainWorldSizeX;
            const float dist = std::fabs(wx - cX);
            float h = channelEdge + 3.0f;                                      // plateau :824
            h += 0.5f * std::sin(wx * 0.35f + ph[0]) * std::cos(wz * 0.28f + ph[1]);
            h += 0.25f * std::sin(wx * 0.70f + ph[2]) * std::sin(wz * 0.60f + ph[3]);
            h += 0.12f * std::sin(wx * 1.40f + ph[4]) * std::cos(wz * 1.20f + ph[5]);
            if (dist < r.riverChannelWidth) {                                  // trapezoidal channel :830-840
                const float u = dist / r.riverChannelWidth;
                const float floorFrac = 0.50f;
                if (u < floorFrac) h = riverFloor;
                else {
                    const float uw = (u - floorFrac) / (1.0f - floorFrac);
                    h = riverFloor + depth * uw * uw;
                }
            } else {
                h = std::fmax(h, channelEdge + 0.3f);
            }
            heights[iz * r.terrainW + ix] = std::fmax(h, yBase - 0.3f);         // :847
        }
    }
    const float emitterZ = r.terrainWorldMinZ + 0.5f;                          // :853-861
    r.riverEmitterPos[0] = centerline(emitterZ);
    r.riverEmitterPos[1] = (yBase + 1.0f) + depth * 0.5f;
    r.riverEmitterPos[2] = emitterZ;
    r.riverEmitterVel[0] = 0.0f; r.riverEmitterVel[1] = -0.5f; r.riverEmitterVel[2] = 0.5f;
    r.riverEmitterRadius = r.riverChannelWidth * 0.35f;
    r.riverSinkY = yBase + 0.3f;
    r.riverSinkZMax = p.param_boxCenter[2] + p.param_boxHalf[2] - 0.5f;
    p.param_gravityY = -120.0f;                                                // :864-865
    p.param_gravityZ = 0.0f;
}

// The sampleH lambda of the spawn (SPHFluid3D.cpp:113-126): four-product bilinear form.
inline float river_sample_host(const SphRiver& r, const float* T, float wx, float wz) {
    float u = (wx - r.terrainWorldMinX) / r.terrainWorldSizeX * float(r.terrainW - 1);
    float v = (wz - r.terrainWorldMinZ) / r.terrainWorldSizeZ * float(r.terrainH - 1);
    u = std::fmax(0.0f, std::fmin(float(r.terrainW - 2), u));
    v = std::fmax(0.0f, std::fmin(float(r.terrainH - 2), v));
    const int ix = int(u), iz = int(v);
    const float fx = u - float(ix), fz = v - float(iz);
    const float* row0 = T + iz * r.terrainW + ix;
    const float* row1 = row0 + r.terrainW;
    return row0[0] * (1 - fx) * (1 - fz) + row0[1] * fx * (1 - fz) + row1[0] * (1 - fx) * fz + row1[1] * fx * fz;
}

// InitializeParticles, river branch (SPHFluid3D.cpp:104-160): channel fill, then the rest at the emitter.
inline void spawn_river_particles(const SphParams& p, const SphRiver& r, const float* T, size_t nRequested, uint32_t seed,
                                  std::vector<SphParticle>& out, float& massOut) {
    const float spacing = p.param_h * 0.85f;
    massOut = p.param_restDensity * spacing * spacing * spacing;
    Pcg32 rng(seed);
    const float jl = -spacing * p.param_jitterAmp, jh = spacing * p.param_jitterAmp;
    auto jitter = [&]() { return p.param_useJitter ? rng.uniform(jl, jh) : 0.0f; };
    auto record = [&](float x, float y, float z, float vz) {
        SphParticle q;
        std::memset(&q, 0, sizeof(q));
        q.pos[0] = x; q.pos[1] = y; q.pos[2] = z;
        q.vel[2] = vz;
        q.padC = int(out.size() & 1u);                                         // isGhost = isActive = 0, padC = count & 1
        out.push_back(q);
    };
    out.clear();
    const float zEnd = r.terrainWorldMinZ + r.terrainWorldSizeZ - spacing;
    for (float wz = r.terrainWorldMinZ + spacing; wz < zEnd && out.size() < nRequested; wz += spacing) {
        const float cX = p.param_boxCenter[0] + r.riverAmp * std::sin(r.riverFreq * wz + r.riverPhase);
        for (float wx = cX - r.riverChannelWidth; wx <= cX + r.riverChannelWidth && out.size() < nRequested; wx += spacing) {
            const float ty = river_sample_host(r, T, wx, wz);
            for (float wy = ty + spacing; wy <= ty + 2.5f && out.size() < nRequested; wy += spacing) {
                const float jx = jitter(), jy = jitter(), jz = jitter();
                record(wx + jx, wy + jy, wz + jz, 0.5f);
            }
        }
    }
    const float hw = r.riverChannelWidth * 0.5f;
    while (out.size() < nRequested) {                                          // :144-159
        const float wx = r.riverEmitterPos[0] + rng.uniform(-hw, hw);
        const float wz = r.riverEmitterPos[2] + rng.uniform(-hw, hw);
        const float ty = river_sample_host(r, T, wx, wz);
        const float up = rng.uniform(0.0f, 1.5f);
        record(wx, ty + up, wz, 2.0f);
    }
}

}  // namespace sph
