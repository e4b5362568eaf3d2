/*
 * sph_oracle.c -- CPU restatement of the reference's SPH substep path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and its GLSL compute shaders cannot be
 * executed in the build container, so this restatement is pinned only by the
 * analytic known-answer tests in tests/ (derived from the shader formulas) and by
 * an independent brute-force numpy restatement (oracle/oracle.py).
 *
 * What is restated (paths relative to /root/reference/ComponentFramework):
 *   shaders/ClearGrid.comp:7-10, shaders/BuildGrid.comp:21-37   -> o_build_grid
 *   shaders/SPHFluid.comp:42-64 (kernels), :66-221 (main)       -> o_sph_one
 *   shaders/OBBConstraints.comp:31-39, :297-309, :311-330       -> o_obb_one (box)
 *   shaders/WaveImpulse.comp:30-46, SPHFluid3D.cpp:604-623      -> sph_oracle_wave_impulse
 *   SPHFluid3D.cpp:13-30 (MakeRotationMat3XYZ)                  -> o_rotation
 *   SPHFluid3D.cpp:354-376 (ComputeGridExtents), .h:127-158     -> sph_oracle_grid_extents
 *   SPHFluid3D.cpp:431-509 (DispatchCompute)                    -> sph_oracle_substep
 *   SPHFluid3D.cpp:85-102,159-332 (InitializeParticles std fill)-> sph_oracle_spawn
 *
 * Semantics this oracle FIXES because the reference leaves them open
 * (SURVEY.md section 8a "Semantics the oracle must fix"):
 *  1. Snapshot reads.  SPHFluid.comp reads neighbour records from the buffer every
 *     invocation overwrites (:99,:131,:190 vs :220).  Here every neighbour read
 *     sees the state at dispatch entry; a particle's own state evolves through the
 *     three sweeps exactly as the shader writes it.
 *  2. Neighbour order.  atomicExchange arrival order (BuildGrid.comp:36) makes the
 *     fp32 sums order-dependent and irreproducible.  Canonical order here: cells by
 *     ascending reference cell index ((cz*gy+cy)*gx+cx, i.e. dz outer, dy, dx inner),
 *     inside a cell ascending particle index.  The shader's own dx->dy->dz nesting
 *     is not observable in its output because list order inside a cell is already
 *     arbitrary.
 *  3. pow().  pow(h,9), pow(h,6), pow(x,3.0), pow(x,2.0) are exact products.
 *  4. Division, sqrt: IEEE-754 correctly rounded fp32.
 *  5. Contraction.  GLSL lets the compiler fuse a*b+c.  Fixed here: dot products are
 *     fma(z,z', fma(y,y', x*x')); accumulations `s += a*b` are fmaf(a,b,s); every
 *     other operation is a separately rounded fp32 op (build with -ffp-contract=off).
 *  6. sin() in WaveImpulse: a fully specified fp32 routine (o_sinf) so that the HIP
 *     kernel can match bit for bit; max error about 1 ulp for |x| < 1e4.
 *  7. RNG for spawn jitter: PCG32 (the reference seeds default_random_engine with
 *     time(nullptr), SPHFluid3D.cpp:99, which is neither reproducible nor portable).
 *  8. Float -> cell index: clamp in float then convert (identical for in-range
 *     values; avoids undefined float->int overflow for far-away particles).
 *  9. Reciprocal forms.  GLSL specifies a/b only to 2.5 ULP and drivers lower it to
 *     a*rcp(b).  Divisions by a NEIGHBOUR quantity are fixed in that form, with the
 *     reciprocal a correctly rounded fp32 division, so that it can be formed once per
 *     neighbour: invRho_j = 1/rho_j gives `mass / pj.density` (SPHFluid.comp:141,147,192)
 *     = mass * invRho_j and `x / (2.0 * pj.density)` (:137) = (x * 0.5) * invRho_j; and
 *     `rij / r` (:54) = rij * (1/r).  Every other division is written as in the shader.
 * 10. cos, atan(y,x) and pow of OBBConstraints.comp:144-296 (container shapes 7..14) are, like
 *     sin (5), fully specified fp32 routines: the sine's Cody-Waite reduction shifted by a
 *     quadrant; cephes atanf with a fixed operation order; pow(x,p) = exp2(p*log2 x) with an
 *     fdlibm-style log and a degree-6 exp2 polynomial (|rel. error| <= about 6e-8 * (1 + |p log2 x|)).
 *     The curves those shapes sample at fixed parameter values (trefoil, DNA, heart, coil) are
 *     tabulated once per dispatch with the host's libm; the spawn's insideShape is host code in
 *     the reference too and uses libm.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    float pos[4];
    float vel[4];
    float acc[4];
    float density, pressure, padA, padB;
    int32_t isGhost, isActive, padC, pad0;
} OParticle; /* 80 bytes, SPHFluid3D.h:12-24 */

typedef struct {
    float h, mass, restDensity, gasConstant, viscosity;      /* SPHFluid3D.h:94-98 */
    float gravity[3];                                        /* :99-101 (X,Y,Z)    */
    float surfaceTension, timeStep;                          /* :102-103           */
    int32_t pause;                                           /* :104               */
    int32_t useJitter; float jitterAmp;                      /* :106-107           */
    float foamGen, foamVelRef;                               /* :109-110           */
    float boxCenter[3], boxHalf[3], boxEulerDeg[3];          /* :112-116           */
    int32_t shapeType; float shapeAux[3];                    /* :117-119           */
    int32_t mixPattern, dyePattern;                          /* :121-122           */
    float wallRestitution, wallFriction;                     /* :123-124           */
    int32_t gridCap;                                         /* 160 in the reference, SPHFluid3D.cpp:370 */
} OParams;

typedef struct {
    int32_t dims[3];
    int32_t numCells;
    float gridMin[3];
    float cellSize;
} OGrid;

#define O_PI_F 3.141592653589f /* literal used by SPHFluid.comp:45,53,60 */

int sph_oracle_sizeof_particle(void) { return (int)sizeof(OParticle); }
int sph_oracle_sizeof_params(void) { return (int)sizeof(OParams); }

void sph_oracle_default_params(OParams* p) {
    memset(p, 0, sizeof(*p));
    p->h = 0.28f; p->mass = 13.8f; p->restDensity = 1000.0f; p->gasConstant = 2000.0f;
    p->viscosity = 3.5f; p->gravity[0] = 0.0f; p->gravity[1] = -980.0f; p->gravity[2] = 0.0f;
    p->surfaceTension = 0.0728f; p->timeStep = 0.001f; p->pause = 0;
    p->useJitter = 1; p->jitterAmp = 0.20f; p->foamGen = 1.0f; p->foamVelRef = 8.0f;
    p->boxHalf[0] = p->boxHalf[1] = p->boxHalf[2] = 7.0f;
    p->shapeType = 0; p->shapeAux[0] = 5.0f; p->shapeAux[1] = 0.35f; p->shapeAux[2] = 2.5f;
    p->mixPattern = 0; p->dyePattern = 0;
    p->wallRestitution = 0.15f; p->wallFriction = 0.02f;
    p->gridCap = 160;
}

/* ---------------------------------------------------------------- helpers */

static inline float o_dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
static inline float o_clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float o_signf(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

/* Fully specified fp32 sine (semantic 6).  Cody-Waite reduction by pi/2 in three
 * parts, then degree-7 / degree-6 minimax-style polynomials on [-pi/4, pi/4]. */
float sph_oracle_sinf(float x) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float P1 = 1.5703125f;                 /* pi/2 split, 8+ trailing zero bits each */
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.549789948768648e-8f;
    float q = rintf(x * TWO_OVER_PI);
    float r = fmaf(q, -P1, x);
    r = fmaf(q, -P2, r);
    r = fmaf(q, -P3, r);
    /* quadrant = q mod 4 computed in float: no float->int overflow for huge |x| */
    int n = (int)(q - 4.0f * floorf(q * 0.25f));
    float r2 = r * r;
    float res;
    if (n & 1) { /* cos(r) */
        float c = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
        c = fmaf(c, r2, 4.166664568298827e-2f);
        c = fmaf(c, r2, -0.5f);
        res = fmaf(c, r2, 1.0f);
    } else {     /* sin(r) */
        float s = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
        s = fmaf(s, r2, -1.6666654611e-1f);
        s = s * r2;
        res = fmaf(s, r, r);
    }
    return (n & 2) ? -res : res;
}

/* MakeRotationMat3XYZ, SPHFluid3D.cpp:13-30: column-major world_from_box R = Rz*Ry*Rx. */
static void o_rotation(const float eulerDeg[3], float outM[9]) {
    const float k = (float)(3.14159265358979323846 / 180.0); /* M_PI */
    const float rx = eulerDeg[0] * k, ry = eulerDeg[1] * k, rz = eulerDeg[2] * k;
    const float cx = cosf(rx), sx = sinf(rx);
    const float cy = cosf(ry), sy = sinf(ry);
    const float cz = cosf(rz), sz = sinf(rz);
    const float Rz[9] = { cz, sz, 0, -sz, cz, 0, 0, 0, 1 };
    const float Ry[9] = { cy, 0, -sy, 0, 1, 0, sy, 0, cy };
    const float Rx[9] = { 1, 0, 0, 0, cx, sx, 0, -sx, cx };
    float Rzy[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            Rzy[c * 3 + r] = Rz[0 * 3 + r] * Ry[c * 3 + 0] + Rz[1 * 3 + r] * Ry[c * 3 + 1] + Rz[2 * 3 + r] * Ry[c * 3 + 2];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            outM[c * 3 + r] = Rzy[0 * 3 + r] * Rx[c * 3 + 0] + Rzy[1 * 3 + r] * Rx[c * 3 + 1] + Rzy[2 * 3 + r] * Rx[c * 3 + 2];
}
void sph_oracle_rotation(const float eulerDeg[3], float outM[9]) { o_rotation(eulerDeg, outM); }

/* EffectiveHalf, SPHFluid3D.h:127-158 */
void sph_oracle_effective_half(const OParams* p, float out[3]) {
    const float bx = p->boxHalf[0], by = p->boxHalf[1], bz = p->boxHalf[2];
    const float ax = p->shapeAux[0], ay = p->shapeAux[1];
    switch (p->shapeType) {
    case 1: out[0] = bx; out[1] = bx; out[2] = bx; break;
    case 2: case 5: case 6: case 7: case 8: out[0] = bx; out[1] = by; out[2] = bx; break;
    case 3: out[0] = bx + by; out[1] = by; out[2] = bx + by; break;
    case 4: out[0] = bx; out[1] = by + bx; out[2] = bx; break;
    case 9: out[0] = 3.0f * bx + by; out[1] = 0.35f * bx + by; out[2] = 3.0f * bx + by; break;
    case 10: out[0] = bx + by; out[1] = by + ax; out[2] = bx + by; break;
    case 11: case 14: out[0] = bx + by; out[1] = ay + by; out[2] = bx + by; break;
    case 12: out[0] = bx + by; out[1] = 1.15f * bx + by; out[2] = by + 0.3f; break;
    case 13: out[0] = bx; out[1] = bx; out[2] = bx; break;
    default: out[0] = bx; out[1] = by; out[2] = bz; break;
    }
}

/* ComputeGridExtents, SPHFluid3D.cpp:354-376 */
void sph_oracle_grid_extents(const OParams* p, OGrid* g) {
    float R[9], half[3], ext[3];
    g->cellSize = p->h;
    o_rotation(p->boxEulerDeg, R);
    sph_oracle_effective_half(p, half);
    for (int i = 0; i < 3; ++i) {
        ext[i] = fabsf(R[i]) * half[0] + fabsf(R[3 + i]) * half[1] + fabsf(R[6 + i]) * half[2];
        ext[i] = ext[i] + g->cellSize;
        g->gridMin[i] = p->boxCenter[i] - ext[i];
        /* clamp in float, then convert (an out-of-range float -> int conversion is undefined; semantic 8) */
        const float df = ceilf((2.0f * ext[i]) / g->cellSize);
        g->dims[i] = (df >= (float)p->gridCap) ? p->gridCap : ((df >= 1.0f) ? (int)df : 1);
    }
    const long long nc = (long long)g->dims[0] * g->dims[1] * g->dims[2];
    g->numCells = nc < 1 ? 1 : (nc > 2147483647LL ? 2147483647 : (int)nc);
}

/* BuildGrid.comp:21-31: cell coordinate of a position */
static inline void o_cell_coord(const OGrid* g, const float pos[3], int c[3]) {
    for (int a = 0; a < 3; ++a) {
        float q = (pos[a] - g->gridMin[a]) / g->cellSize;
        float f = floorf(q);
        f = fminf(fmaxf(f, 0.0f), (float)(g->dims[a] - 1));
        c[a] = (int)f;
    }
}
static inline int o_flatten(const OGrid* g, const int c[3]) {
    return (c[2] * g->dims[1] + c[1]) * g->dims[0] + c[0];
}

/* ClearGrid.comp + BuildGrid.comp restated as a stable counting sort:
 * cellStart[C+1] exclusive prefix, sorted[N] = particle indices grouped by cell in
 * ascending particle index (canonical order 2), particleCell[N] = flattened cell.
 * Also (optionally) the shader's own linked list for ONE legal arrival order
 * (ascending i => each list is in descending index order). */
void sph_oracle_build_grid(const OParticle* P, int n, const OGrid* g,
                           int32_t* cellStart, int32_t* sorted, int32_t* particleCell,
                           int32_t* cellHead /*nullable*/, int32_t* particleNext /*nullable*/) {
    const int C = g->numCells;
    memset(cellStart, 0, sizeof(int32_t) * (size_t)(C + 1));
    for (int i = 0; i < n; ++i) {
        int c[3];
        o_cell_coord(g, P[i].pos, c);
        int cell = o_flatten(g, c);
        particleCell[i] = cell;
        cellStart[cell + 1]++;
    }
    for (int c = 0; c < C; ++c) cellStart[c + 1] += cellStart[c];
    int32_t* cursor = (int32_t*)malloc(sizeof(int32_t) * (size_t)(C > 0 ? C : 1));
    memcpy(cursor, cellStart, sizeof(int32_t) * (size_t)C);
    for (int i = 0; i < n; ++i) sorted[cursor[particleCell[i]]++] = i;
    free(cursor);
    if (cellHead && particleNext) {
        for (int c = 0; c < C; ++c) cellHead[c] = -1;            /* ClearGrid.comp:9 */
        for (int i = 0; i < n; ++i) {                            /* BuildGrid.comp:23-37 */
            int cell = particleCell[i];
            particleNext[i] = cellHead[cell];
            cellHead[cell] = i;
        }
    }
}

typedef struct {
    float h, h2, poly6C, spikyC, viscC;
    float mass, restDensity, gasConstant, viscosity, surfaceTension;
    float g[3];
    float dt, maxSpeed, foamGen, foamVelRef;
} OConsts;

static void o_consts(const OParams* p, float dt, OConsts* k) {
    const float h = p->h;
    const float h2 = h * h;
    const float h3 = h2 * h;
    const float h6 = h3 * h3;
    const float h9 = h6 * h3;
    k->h = h; k->h2 = h2;
    k->poly6C = 315.0f / ((64.0f * O_PI_F) * h9);   /* SPHFluid.comp:45 */
    k->spikyC = -45.0f / (O_PI_F * h6);             /* :53 */
    k->viscC = 45.0f / (O_PI_F * h6);               /* :60 */
    k->mass = p->mass; k->restDensity = p->restDensity; k->gasConstant = p->gasConstant;
    k->viscosity = p->viscosity; k->surfaceTension = p->surfaceTension;
    k->g[0] = p->gravity[0]; k->g[1] = p->gravity[1]; k->g[2] = p->gravity[2];
    k->dt = dt;
    k->maxSpeed = (0.4f * p->h) / fmaxf(dt, 1e-6f); /* SPHFluid3D.cpp:488 */
    k->foamGen = p->foamGen; k->foamVelRef = p->foamVelRef;
}

/* Stencil traversal of the literal restatement: canonical (dz outer, dy, dx inner; ascending index
 * inside a cell) or, with shaderOrder, the shader's own nesting dx -> dy -> dz (SPHFluid.comp:91-93)
 * with each cell's list in DESCENDING index order -- the list BuildGrid.comp's push-front produces
 * when invocations happen to run in index order, i.e. ONE legal order of the reference itself. */
#define O_STENCIL(it, shaderOrder) \
    const int dx = (shaderOrder) ? ((it) / 9 - 1) : ((it) % 3 - 1); \
    const int dy = ((it) / 3) % 3 - 1; \
    const int dz = (shaderOrder) ? ((it) % 3 - 1) : ((it) / 9 - 1)
/* SPHFluid.comp main(), :66-221, for particle i, snapshot semantics: the LITERAL restatement (contract 0:
 * IEEE sqrt and division exactly where the shader has them).  Kept as the yardstick the engine's
 * arithmetic contract (o_sph_one below) is measured against (tests/test_oracle_contract.py). */
static void o_sph_one_literal(int i, const OParticle* in, OParticle* out, const OGrid* g,
                      const int32_t* cellStart, const int32_t* sorted, const OConsts* k, int shaderOrder) {
    OParticle pi = in[i];
    if (pi.isGhost == 1) {                                   /* :72-83 */
        if (pi.isActive == 0) { out[i] = pi; return; }
        pi.vel[0] = pi.vel[1] = pi.vel[2] = pi.vel[3] = 0.0f;
        pi.acc[0] = pi.acc[1] = pi.acc[2] = pi.acc[3] = 0.0f;
        pi.density = k->restDensity;
        pi.pressure = 0.0f;
        out[i] = pi;
        return;
    }
    int cc[3];
    o_cell_coord(g, pi.pos, cc);                             /* :85-87 */
    const int gx = g->dims[0], gy = g->dims[1], gz = g->dims[2];
    const float h = k->h, h2 = k->h2, mass = k->mass;

    /* ---- sweep 1: density, :90-106 (self included) ---- */
    float density = 0.0f;
    for (int it = 0; it < 27; ++it) {
        O_STENCIL(it, shaderOrder);
        int nx = cc[0] + dx, ny = cc[1] + dy, nz = cc[2] + dz;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= gx || ny >= gy || nz >= gz) continue;
        int cell = (nz * gy + ny) * gx + nx;
        for (int qq = cellStart[cell]; qq < cellStart[cell + 1]; ++qq) {
            const int q = shaderOrder ? (cellStart[cell] + cellStart[cell + 1] - 1 - qq) : qq;
            const OParticle* pj = &in[sorted[q]];
            float ddx = pi.pos[0] - pj->pos[0], ddy = pi.pos[1] - pj->pos[1], ddz = pi.pos[2] - pj->pos[2];
            float r2 = o_dot3(ddx, ddy, ddz, ddx, ddy, ddz);
            if (r2 < h2) {
                float t = h2 - r2;
                float w = k->poly6C * ((t * t) * t);
                density = fmaf(mass, w, density);
            }
        }
    }
    density = fmaxf(density, k->restDensity * 0.5f);
    pi.density = density;
    pi.pressure = fmaxf(k->gasConstant * (pi.density - k->restDensity), 0.0f);   /* :111 */

    /* ---- sweep 2: forces, :113-155 (self skipped) ---- */
    float fP[3] = { 0, 0, 0 }, fV[3] = { 0, 0, 0 }, gradC[3] = { 0, 0, 0 };
    float lapC = 0.0f;
    for (int it = 0; it < 27; ++it) {
        O_STENCIL(it, shaderOrder);
        int nx = cc[0] + dx, ny = cc[1] + dy, nz = cc[2] + dz;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= gx || ny >= gy || nz >= gz) continue;
        int cell = (nz * gy + ny) * gx + nx;
        for (int qq = cellStart[cell]; qq < cellStart[cell + 1]; ++qq) {
            const int q = shaderOrder ? (cellStart[cell] + cellStart[cell + 1] - 1 - qq) : qq;
            int j = sorted[q];
            if (j == i) continue;
            const OParticle* pj = &in[j];
            float rij[3] = { pi.pos[0] - pj->pos[0], pi.pos[1] - pj->pos[1], pi.pos[2] - pj->pos[2] };
            float r = sqrtf(o_dot3(rij[0], rij[1], rij[2], rij[0], rij[1], rij[2]));
            if (r < h && pj->density > 0.0f) {
                float gW[3] = { 0, 0, 0 };                   /* spikyGrad :50-57 */
                if (r > 0.0f) {
                    float hr = h - r;
                    float s = k->spikyC * (hr * hr);
                    float invr = 1.0f / r;                   /* semantic 9: rij / r */
                    gW[0] = s * (rij[0] * invr); gW[1] = s * (rij[1] * invr); gW[2] = s * (rij[2] * invr);
                }
                float invRho = 1.0f / pj->density;           /* semantic 9 */
                float pterm = (((-mass) * (pi.pressure + pj->pressure)) * 0.5f) * invRho;
                float mor = mass * invRho;
                float lapW = k->viscC * (h - r);             /* viscLaplacian :58-64 */
                for (int a = 0; a < 3; ++a) {
                    fP[a] = fmaf(gW[a], pterm, fP[a]);
                    fV[a] = fmaf((pj->vel[a] - pi.vel[a]) * mor, lapW, fV[a]);
                    gradC[a] = fmaf(mor, gW[a], gradC[a]);
                }
                lapC = fmaf(mor, lapW, lapC);
                /* `c += mj_over_rhoj * w` (:148) feeds nothing and is omitted. */
            }
        }
    }
    float fS[3] = { 0, 0, 0 };                               /* :157-163 */
    float gl = sqrtf(o_dot3(gradC[0], gradC[1], gradC[2], gradC[0], gradC[1], gradC[2]));
    if (gl > 1e-6f) {
        float sc = (-k->surfaceTension) * lapC;
        for (int a = 0; a < 3; ++a) fS[a] = sc * (gradC[a] / gl);
    }
    for (int a = 0; a < 3; ++a) {                            /* :165-171 */
        float fG = k->g[a] * pi.density;
        float t = fmaf(k->viscosity, fV[a], fP[a]);
        t = t + fG;
        t = t + fS[a];
        float acc = t / pi.density;
        pi.acc[a] = acc;
        pi.vel[a] = fmaf(acc, k->dt, pi.vel[a]);
        pi.vel[a] = pi.vel[a] * 0.995f;
        pi.pos[a] = fmaf(pi.vel[a], k->dt, pi.pos[a]);
    }
    pi.acc[3] = 0.0f;

    /* ---- sweep 3: XSPH, :177-201 (own state updated, neighbours at entry,
     *      cell coordinate NOT recomputed) ---- */
    float xs[3] = { 0, 0, 0 };
    float norm = 0.0f;
    for (int it = 0; it < 27; ++it) {
        O_STENCIL(it, shaderOrder);
        int nx = cc[0] + dx, ny = cc[1] + dy, nz = cc[2] + dz;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= gx || ny >= gy || nz >= gz) continue;
        int cell = (nz * gy + ny) * gx + nx;
        for (int qq = cellStart[cell]; qq < cellStart[cell + 1]; ++qq) {
            const int q = shaderOrder ? (cellStart[cell] + cellStart[cell + 1] - 1 - qq) : qq;
            int j = sorted[q];
            if (j == i) continue;
            const OParticle* pj = &in[j];
            float ddx = pi.pos[0] - pj->pos[0], ddy = pi.pos[1] - pj->pos[1], ddz = pi.pos[2] - pj->pos[2];
            float r2 = o_dot3(ddx, ddy, ddz, ddx, ddy, ddz);
            if (r2 < h2 && pj->density > 0.0f) {
                float t = h2 - r2;
                float w = k->poly6C * ((t * t) * t);
                float mor = mass * (1.0f / pj->density);     /* semantic 9 */
                for (int a = 0; a < 3; ++a) xs[a] = fmaf((pj->vel[a] - pi.vel[a]) * w, mor, xs[a]);
                norm = norm + w;
            }
        }
    }
    if (norm > 0.0f) { xs[0] = xs[0] / norm; xs[1] = xs[1] / norm; xs[2] = xs[2] / norm; }
    for (int a = 0; a < 3; ++a) pi.vel[a] = fmaf(0.12f, xs[a], pi.vel[a]);

    {                                                        /* velocity cap :203-207 */
        float sp = sqrtf(o_dot3(pi.vel[0], pi.vel[1], pi.vel[2], pi.vel[0], pi.vel[1], pi.vel[2]));
        if (sp > k->maxSpeed) {
            float f = k->maxSpeed / sp;
            pi.vel[0] = pi.vel[0] * f; pi.vel[1] = pi.vel[1] * f; pi.vel[2] = pi.vel[2] * f;
        }
    }
    {                                                        /* foam :209-217 */
        float speed = sqrtf(o_dot3(pi.vel[0], pi.vel[1], pi.vel[2], pi.vel[0], pi.vel[1], pi.vel[2]));
        float aer = o_clampf((k->restDensity - pi.density) / k->restDensity, 0.0f, 1.0f)
                  * o_clampf(speed / fmaxf(k->foamVelRef, 1e-3f), 0.0f, 1.0f);
        pi.padA = fmaxf(aer * k->foamGen, pi.padA * 0.995f);
    }
    out[i] = pi;                                             /* :220 */
}

/* ---------------------------------------------------------------------------------------------
 * The engine's arithmetic contract (contract 1, the default; semantic 11 in the header).
 * Same sums, same candidate order, same accept tests as o_sph_one_literal; what changes is HOW a
 * few quantities are formed, so that the per-pair arithmetic is add / mul / fma only (it then packs
 * two pairs per v_pk_*_f32 instruction on gfx950) and costs no IEEE sqrt / division per pair:
 *   - 1/sqrt(x) is o_rsqrt(): integer seed + three Newton steps, |rel. error| <= 2 ulp -- inside
 *     GLSL's own tolerance for inversesqrt (2 ULP, GLSL 4.50 spec section 4.7.1);
 *     sqrt(x) = x * o_rsqrt(x), 1/x = o_rsqrt(x)^2 (own density, XSPH norm);
 *   - constant factors are taken out of the sums: density = (mass*poly6C) * sum (h2-r2)^3; the XSPH
 *     quotient sum(dv W m/rho) / sum(W) drops the poly6 coefficient from both sums;
 *   - spikyGrad = (spikyC (h-r)^2 / r) * rij instead of spikyC (h-r)^2 * (rij / r);
 *     viscosity term (vj-vi) * ((m/rho_j) lapW) instead of ((vj-vi) (m/rho_j)) * lapW;
 *   - `r < h` is tested as r2 < h2 and `length(v) > maxSpeed` as |v|^2 > maxSpeed^2.
 * Rejected candidates add exactly +0, so "skip" and "add zero" are the same bits (the HIP kernels
 * are branch-free).
 * ------------------------------------------------------------------------------------------- */
static inline uint32_t o_fbits(float f);
static inline float o_bitsf(uint32_t u);
float sph_oracle_rsqrt(float x) {                            /* x > 0, normal */
    uint32_t u; memcpy(&u, &x, 4);
    u = 0x5f3759dfu - (u >> 1);
    float y; memcpy(&y, &u, 4);
    const float xh = 0.5f * x;
    for (int it = 0; it < 3; ++it) {
        float t = y * y;
        float e = fmaf(-xh, t, 0.5f);
        y = fmaf(y, e, y);
    }
    return y;
}
#define O_TINY 1e-30f

static void o_sph_one(int i, const OParticle* in, OParticle* out, const OGrid* g,
                      const int32_t* cellStart, const int32_t* sorted, const OConsts* k) {
    OParticle pi = in[i];
    if (pi.isGhost == 1) {                                   /* :72-83 */
        if (pi.isActive == 0) { out[i] = pi; return; }
        pi.vel[0] = pi.vel[1] = pi.vel[2] = pi.vel[3] = 0.0f;
        pi.acc[0] = pi.acc[1] = pi.acc[2] = pi.acc[3] = 0.0f;
        pi.density = k->restDensity;
        pi.pressure = 0.0f;
        out[i] = pi;
        return;
    }
    int cc[3];
    o_cell_coord(g, pi.pos, cc);                             /* :85-87 */
    const int gx = g->dims[0], gy = g->dims[1], gz = g->dims[2];
    const float h = k->h, h2 = k->h2, mass = k->mass;
    const float mp6 = mass * k->poly6C;
    const float nhm = (-mass) * 0.5f;
    const float maxSpeed2 = k->maxSpeed * k->maxSpeed;
    const float invRho0 = 1.0f / k->restDensity;
    const float invFoamRef = 1.0f / fmaxf(k->foamVelRef, 1e-3f);

    /* ---- sweep 1: density, :90-106 (self included) ---- */
    float dsum = 0.0f;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        int nx = cc[0] + dx, ny = cc[1] + dy, nz = cc[2] + dz;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= gx || ny >= gy || nz >= gz) continue;
        int cell = (nz * gy + ny) * gx + nx;
        for (int q = cellStart[cell]; q < cellStart[cell + 1]; ++q) {
            const OParticle* pj = &in[sorted[q]];
            float ddx = pi.pos[0] - pj->pos[0], ddy = pi.pos[1] - pj->pos[1], ddz = pi.pos[2] - pj->pos[2];
            float r2 = o_dot3(ddx, ddy, ddz, ddx, ddy, ddz);
            float t = fmaxf(h2 - r2, 0.0f);                  /* r2 >= h2 adds +0 */
            dsum = fmaf(t * t, t, dsum);
        }
    }
    pi.density = fmaxf(mp6 * dsum, k->restDensity * 0.5f);   /* :106 */
    pi.pressure = fmaxf(k->gasConstant * (pi.density - k->restDensity), 0.0f);   /* :111 */

    /* ---- sweep 2: forces, :113-155 (self skipped) ---- */
    float fP[3] = { 0, 0, 0 }, fV[3] = { 0, 0, 0 }, gradC[3] = { 0, 0, 0 };
    float lapC = 0.0f;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        int nx = cc[0] + dx, ny = cc[1] + dy, nz = cc[2] + dz;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= gx || ny >= gy || nz >= gz) continue;
        int cell = (nz * gy + ny) * gx + nx;
        for (int q = cellStart[cell]; q < cellStart[cell + 1]; ++q) {
            int j = sorted[q];
            if (j == i) continue;
            const OParticle* pj = &in[j];
            float rij[3] = { pi.pos[0] - pj->pos[0], pi.pos[1] - pj->pos[1], pi.pos[2] - pj->pos[2] };
            float r2 = o_dot3(rij[0], rij[1], rij[2], rij[0], rij[1], rij[2]);
            if (r2 < h2 && pj->density > 0.0f) {
                float rinv = sph_oracle_rsqrt(fmaxf(r2, O_TINY));
                float r = r2 * rinv;
                float hr = h - r;
                float sr = (k->spikyC * (hr * hr)) * rinv;   /* spikyGrad :50-57 = sr * rij (0 for rij = 0) */
                float gW[3] = { sr * rij[0], sr * rij[1], sr * rij[2] };
                float invRho = 1.0f / pj->density;           /* semantic 9: once per neighbour */
                float mor = mass * invRho;
                float pterm = ((pi.pressure + pj->pressure) * nhm) * invRho;
                float ml = mor * (k->viscC * hr);            /* viscLaplacian :58-64 times m/rho_j */
                for (int a = 0; a < 3; ++a) {
                    fP[a] = fmaf(gW[a], pterm, fP[a]);
                    fV[a] = fmaf(pj->vel[a] - pi.vel[a], ml, fV[a]);
                    gradC[a] = fmaf(mor, gW[a], gradC[a]);
                }
                lapC = lapC + ml;
            }
        }
    }
    float fS[3];                                             /* :157-163 */
    float gl2 = o_dot3(gradC[0], gradC[1], gradC[2], gradC[0], gradC[1], gradC[2]);
    {
        float sc = (gl2 > 1e-12f) ? (((-k->surfaceTension) * lapC) * sph_oracle_rsqrt(gl2)) : 0.0f;
        for (int a = 0; a < 3; ++a) fS[a] = sc * gradC[a];   /* branch-free form of the kernels: sc = +0 below the threshold */
    }
    {
        float rr = sph_oracle_rsqrt(fmaxf(pi.density, O_TINY));
        float invRhoI = rr * rr;
        for (int a = 0; a < 3; ++a) {                        /* :165-171 */
            float fG = k->g[a] * pi.density;
            float t = fmaf(k->viscosity, fV[a], fP[a]);
            t = t + fG;
            t = t + fS[a];
            float acc = t * invRhoI;
            pi.acc[a] = acc;
            pi.vel[a] = fmaf(acc, k->dt, pi.vel[a]);
            pi.vel[a] = pi.vel[a] * 0.995f;
            pi.pos[a] = fmaf(pi.vel[a], k->dt, pi.pos[a]);
        }
    }
    pi.acc[3] = 0.0f;

    /* ---- sweep 3: XSPH, :177-201 (own state updated, neighbours at entry, cell NOT recomputed) ---- */
    float xs[3] = { 0, 0, 0 };
    float norm = 0.0f;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        int nx = cc[0] + dx, ny = cc[1] + dy, nz = cc[2] + dz;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= gx || ny >= gy || nz >= gz) continue;
        int cell = (nz * gy + ny) * gx + nx;
        for (int q = cellStart[cell]; q < cellStart[cell + 1]; ++q) {
            int j = sorted[q];
            if (j == i) continue;
            const OParticle* pj = &in[j];
            float ddx = pi.pos[0] - pj->pos[0], ddy = pi.pos[1] - pj->pos[1], ddz = pi.pos[2] - pj->pos[2];
            float r2 = o_dot3(ddx, ddy, ddz, ddx, ddy, ddz);
            if (r2 < h2 && pj->density > 0.0f) {
                float t = h2 - r2;
                float w3 = (t * t) * t;
                float wm = w3 * (mass * (1.0f / pj->density));
                for (int a = 0; a < 3; ++a) xs[a] = fmaf(pj->vel[a] - pi.vel[a], wm, xs[a]);
                norm = norm + w3;
            }
        }
    }
    if (norm > 0.0f) {
        float nr = sph_oracle_rsqrt(fmaxf(norm, O_TINY));
        float ninv = nr * nr;
        xs[0] = xs[0] * ninv; xs[1] = xs[1] * ninv; xs[2] = xs[2] * ninv;
    }
    for (int a = 0; a < 3; ++a) pi.vel[a] = fmaf(0.12f, xs[a], pi.vel[a]);

    {                                                        /* velocity cap :203-207 */
        float sp2 = o_dot3(pi.vel[0], pi.vel[1], pi.vel[2], pi.vel[0], pi.vel[1], pi.vel[2]);
        if (sp2 > maxSpeed2) {
            float f = k->maxSpeed * sph_oracle_rsqrt(sp2);
            pi.vel[0] = pi.vel[0] * f; pi.vel[1] = pi.vel[1] * f; pi.vel[2] = pi.vel[2] * f;
        }
    }
    {                                                        /* foam :209-217 */
        float sp2 = o_dot3(pi.vel[0], pi.vel[1], pi.vel[2], pi.vel[0], pi.vel[1], pi.vel[2]);
        float speed = sp2 * sph_oracle_rsqrt(fmaxf(sp2, O_TINY));
        float aer = o_clampf((k->restDensity - pi.density) * invRho0, 0.0f, 1.0f)
                  * o_clampf(speed * invFoamRef, 0.0f, 1.0f);
        pi.padA = fmaxf(aer * k->foamGen, pi.padA * 0.995f);
    }
    out[i] = pi;                                             /* :220 */
}

/* 1 = the engine's contract (o_sph_one); 0 = literal restatement, canonical order; 2 = literal
 * restatement in the shader's own traversal order (see O_STENCIL) */
static int g_contract = 1;
void sph_oracle_set_contract(int c) { g_contract = (c == 0 || c == 2) ? c : 1; }
int sph_oracle_get_contract(void) { return g_contract; }

/* OBBConstraints.comp, box branch :297-309 and response :311-330.  Other shape
 * types are handled by sph_oracle_obb_shape() below. */
typedef struct {
    float R[9];
    float c[3], half[3], aux[3];
    float e, f;
    int shape;
    float tab[384];                                          /* sampled curve of shapes 9/11/12/14 */
    int tabCount;
    float best0[3];
} OObb;

int sph_oracle_shape_table(const OParams* p, float* out, float best0[3]);


static void o_obb_setup(const OParams* p, OObb* b) {
    o_rotation(p->boxEulerDeg, b->R);                        /* SPHFluid3D.cpp:498-506 */
    for (int a = 0; a < 3; ++a) { b->c[a] = p->boxCenter[a]; b->half[a] = p->boxHalf[a]; b->aux[a] = p->shapeAux[a]; }
    b->e = p->wallRestitution; b->f = p->wallFriction; b->shape = p->shapeType;
    b->tabCount = sph_oracle_shape_table(p, b->tab, b->best0);
}

static inline void o_matvec(const float R[9], const float v[3], float out[3]) {
    for (int i = 0; i < 3; ++i) out[i] = fmaf(R[6 + i], v[2], fmaf(R[3 + i], v[1], R[i] * v[0]));
}


/* ====================================================================================
 * Container shapes 7..14 (OBBConstraints.comp:144-296) and the matching insideShape
 * cases of the spawn (SPHFluid3D.cpp:200-289).
 *
 * Semantic 10: the shader's cos / atan(y,x) / pow are fixed as fully specified fp32
 * routines (o_cosf, o_atan2f, o_powf), like o_sinf, so that the HIP kernels can match
 * bit for bit.  Curves that the shader samples at FIXED parameter values (trefoil, DNA,
 * heart, coil) are tabulated once per dispatch on the host with libm and handed to the
 * kernel; the oracle builds the identical table (o_shape_table).
 * ==================================================================================== */
float sph_oracle_cosf(float x) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float P1 = 1.5703125f, P2 = 4.837512969970703125e-4f, P3 = 7.549789948768648e-8f;
    float q = rintf(x * TWO_OVER_PI);
    float r = fmaf(q, -P1, x);
    r = fmaf(q, -P2, r);
    r = fmaf(q, -P3, r);
    int n = ((int)(q - 4.0f * floorf(q * 0.25f)) + 1) & 3;      /* cos x = sin(x + pi/2) */
    float r2 = r * r, res;
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

static float o_atanf(float x) {                              /* cephes atanf, fixed op order */
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    y = y + fmaf(p * z, x, x);
    return sign * y;
}
float sph_oracle_atan2f(float y, float x) {                  /* GLSL atan(y, x) */
    const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966f;
    if (x > 0.0f) return o_atanf(y / x);
    if (x < 0.0f) return (y >= 0.0f) ? (o_atanf(y / x) + PI) : (o_atanf(y / x) - PI);
    return (y > 0.0f) ? PIO2 : ((y < 0.0f) ? -PIO2 : 0.0f);
}

static inline uint32_t o_fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float o_bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float o_log2f(float x) {                              /* x > 0, normal; fdlibm-style, fixed op order */
    uint32_t ix = o_fbits(x);
    int e = (int)((int32_t)(ix - 0x3f3504f3u) >> 23);        /* floor(log2(x / sqrt(1/2))) */
    ix = ix - ((uint32_t)e << 23);
    float m = o_bitsf(ix);                                   /* in [sqrt(1/2), sqrt(2)) */
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s, w = z * z;
    float t1 = w * fmaf(w, 0.24279078841f, 0.40000972152f);
    float t2 = z * fmaf(w, 0.28498786688f, 0.66666662693f);
    float R = t2 + t1;
    float hfsq = 0.5f * (f * f);
    float ln1pf = f - (hfsq - s * (hfsq + R));
    return fmaf(ln1pf, 1.44269504088896341f, (float)e);
}
static float o_exp2f(float y) {                              /* fixed op order, |error| ~ 2e-7 relative */
    y = fminf(fmaxf(y, -126.0f), 127.0f);
    float n = rintf(y);
    float r = y - n;                                         /* [-0.5, 0.5] */
    float p = fmaf(r, 1.535336188319500e-4f, 1.339887440266574e-3f);
    p = fmaf(p, r, 9.618437357674640e-3f);
    p = fmaf(p, r, 5.550357186158072e-2f);
    p = fmaf(p, r, 2.402264791363012e-1f);
    p = fmaf(p, r, 6.931472028550421e-1f);
    p = fmaf(p, r, 1.0f);
    return p * o_bitsf((uint32_t)((int)n + 127) << 23);
}
float sph_oracle_powf(float x, float p) {                    /* GLSL pow for x >= 0 */
    if (!(x > 0.0f)) return (p > 0.0f) ? 0.0f : ((p == 0.0f) ? 1.0f : INFINITY);
    return o_exp2f(p * o_log2f(x));
}

/* Sampled curves of shapes 9 / 11 / 12 / 14: out[3*k..] (k < *count <= 128) and the initial
 * "best" point the shader starts its nearest-sample search from. */
int sph_oracle_shape_table(const OParams* p, float* out, float best0[3]) {
    const float hx = p->boxHalf[0], hy = p->boxHalf[1];
    int n = 0;
    best0[0] = best0[1] = best0[2] = 0.0f;
    switch (p->shapeType) {
    case 9: {                                                /* trefoil :180-202 */
        const float S = hx;
        best0[0] = 3.0f * S;
        for (int k = 0; k < 48; ++k) {
            float t = 6.2831853f * (float)k / 48.0f;
            out[3 * n + 0] = S * (sinf(t) + 2.0f * sinf(2.0f * t));
            out[3 * n + 1] = S * (0.35f * (-sinf(3.0f * t)));
            out[3 * n + 2] = S * (cosf(t) - 2.0f * cosf(2.0f * t));
            ++n;
        }
        break;
    }
    case 11: case 14: {                                      /* DNA :224-241, coil :282-296 */
        const float R = hx, r = hy;
        const float turns = fmaxf(1.0f, p->shapeAux[0]), H = fmaxf(p->shapeAux[1], r);
        best0[0] = R; best0[1] = -H;
        for (int k = 0; k < 64; ++k) {
            float f = (float)k / 63.0f;
            float t = f * turns * 6.2831853f;
            float y = (f - 0.5f) * 2.0f * H;
            out[3 * n + 0] = R * cosf(t); out[3 * n + 1] = y; out[3 * n + 2] = R * sinf(t); ++n;
            if (p->shapeType == 11) {
                out[3 * n + 0] = R * cosf(t + 3.14159265f); out[3 * n + 1] = y; out[3 * n + 2] = R * sinf(t + 3.14159265f); ++n;
            }
        }
        break;
    }
    case 12: {                                               /* heart :242-257 */
        const float S = hx * 0.0625f;
        for (int k = 0; k < 64; ++k) {
            float t = 6.2831853f * (float)k / 64.0f;
            float st = sinf(t);
            float hxx = 16.0f * st * st * st;
            float hyy = 13.0f * cosf(t) - 5.0f * cosf(2.0f * t) - 2.0f * cosf(3.0f * t) - cosf(4.0f * t);
            out[3 * n + 0] = S * hxx; out[3 * n + 1] = S * hyy; out[3 * n + 2] = 0.0f; ++n;
        }
        break;
    }
    default: break;
    }
    return n;
}

/* Shapes 7..14; returns hit, fills qL / nL (local space). */
static int o_shape_project_ext(const OObb* b, const float pL[3], float qL[3], float nL[3]) {
    const float hx = b->half[0], hy = b->half[1];
    switch (b->shape) {
    case 7: {                                                /* star prism :144-163 */
        float R = hx, H = hy;
        float pts = fmaxf(3.0f, b->aux[0]);
        float depth = o_clampf(b->aux[1], 0.0f, 0.9f);
        float yC = o_clampf(pL[1], -H, H);
        float ang = sph_oracle_atan2f(pL[2], pL[0]);
        float rMax = R * (1.0f - depth * (0.5f + 0.5f * sph_oracle_cosf(pts * ang)));
        float lxz = sqrtf(fmaf(pL[2], pL[2], pL[0] * pL[0]));
        float qx = pL[0], qz = pL[2];
        if (lxz > rMax) { float s = rMax / fmaxf(lxz, 1e-6f); qx = pL[0] * s; qz = pL[2] * s; }
        qL[0] = qx; qL[1] = yC; qL[2] = qz;
        float de[3] = { pL[0] - qL[0], pL[1] - qL[1], pL[2] - qL[2] };
        float dl = sqrtf(o_dot3(de[0], de[1], de[2], de[0], de[1], de[2]));
        if (dl > 1e-6f) { nL[0] = de[0] / dl; nL[1] = de[1] / dl; nL[2] = de[2] / dl; return 1; }
        return 0;
    }
    case 8: {                                                /* superellipsoid :164-179 */
        float a = fmaxf(hx, 1e-6f), bb = fmaxf(hy, 1e-6f);
        float n = o_clampf(b->aux[2], 0.6f, 8.0f);
        float e[3] = { a, bb, a };
        float u[3] = { fabsf(pL[0]) / e[0], fabsf(pL[1]) / e[1], fabsf(pL[2]) / e[2] };
        float F = (sph_oracle_powf(u[0], n) + sph_oracle_powf(u[1], n)) + sph_oracle_powf(u[2], n);
        if (F > 1.0f) {
            float s = sph_oracle_powf(F, -1.0f / n);
            float g[3];
            for (int k = 0; k < 3; ++k) {
                qL[k] = pL[k] * s;
                g[k] = (o_signf(pL[k]) * sph_oracle_powf(fmaxf(fabsf(qL[k]) / e[k], 1e-6f), n - 1.0f)) / e[k];
            }
            float gl = sqrtf(o_dot3(g[0], g[1], g[2], g[0], g[1], g[2]));
            for (int k = 0; k < 3; ++k) nL[k] = g[k] / gl;
            return 1;
        }
        return 0;
    }
    case 9: case 11: case 12: case 14: {                     /* nearest sample of a tabulated curve, then tube */
        float r = hy;
        float best[3] = { b->best0[0], b->best0[1], b->best0[2] };
        float bestD2 = 1e30f;
        for (int k = 0; k < b->tabCount; ++k) {
            const float* c = b->tab + 3 * k;
            float d0 = pL[0] - c[0], d1 = pL[1] - c[1], d2v = pL[2] - c[2];
            float d2 = o_dot3(d0, d1, d2v, d0, d1, d2v);
            if (d2 < bestD2) { bestD2 = d2; best[0] = c[0]; best[1] = c[1]; best[2] = c[2]; }
        }
        float d[3] = { pL[0] - best[0], pL[1] - best[1], pL[2] - best[2] };
        float dl = sqrtf(o_dot3(d[0], d[1], d[2], d[0], d[1], d[2]));
        if (dl > r) {
            float m = fmaxf(dl, 1e-6f);
            for (int k = 0; k < 3; ++k) { nL[k] = d[k] / m; qL[k] = best[k] + nL[k] * r; }
            return 1;
        }
        return 0;
    }
    case 10: {                                               /* Moebius band :203-223 */
        float R = hx, wHalf = hy, tHalf = fmaxf(b->aux[0], 0.05f);
        float phi = sph_oracle_atan2f(pL[2], pL[0]);
        float cp = sph_oracle_cosf(phi), sp = sph_oracle_sinf(phi);
        float c[3] = { R * cp, 0.0f, R * sp };
        float psi = 0.5f * phi;
        float cps = sph_oracle_cosf(psi), sps = sph_oracle_sinf(psi);
        float wA[3] = { cps * cp, sps, cps * sp };
        float tA[3] = { (-sps) * cp, cps, (-sps) * sp };
        float o[3] = { pL[0] - c[0], pL[1] - c[1], pL[2] - c[2] };
        float cu = o_clampf(o_dot3(o[0], o[1], o[2], wA[0], wA[1], wA[2]), -wHalf, wHalf);
        float cv = o_clampf(o_dot3(o[0], o[1], o[2], tA[0], tA[1], tA[2]), -tHalf, tHalf);
        for (int k = 0; k < 3; ++k) qL[k] = (c[k] + cu * wA[k]) + cv * tA[k];
        float de[3] = { pL[0] - qL[0], pL[1] - qL[1], pL[2] - qL[2] };
        float dl = sqrtf(o_dot3(de[0], de[1], de[2], de[0], de[1], de[2]));
        if (dl > 1e-5f) { nL[0] = de[0] / dl; nL[1] = de[1] / dl; nL[2] = de[2] / dl; return 1; }
        return 0;
    }
    case 13: {                                               /* gyroid :258-281 */
        float R = hx;
        float sc = fmaxf(b->aux[0], 0.1f);
        float th = o_clampf(b->aux[1], 0.2f, 2.5f);
        float lp = sqrtf(o_dot3(pL[0], pL[1], pL[2], pL[0], pL[1], pL[2]));
        if (lp > R) {
            float m = fmaxf(lp, 1e-6f);
            for (int k = 0; k < 3; ++k) { nL[k] = pL[k] / m; qL[k] = nL[k] * R; }
            return 1;
        }
        float qx = pL[0] * sc, qy = pL[1] * sc, qz = pL[2] * sc;
        float sx = sph_oracle_sinf(qx), cx = sph_oracle_cosf(qx), sy = sph_oracle_sinf(qy), cy = sph_oracle_cosf(qy);
        float sz = sph_oracle_sinf(qz), cz = sph_oracle_cosf(qz);
        float g = (sx * cy + sy * cz) + sz * cx;
        if (fabsf(g) > th) {
            float gr[3] = { sc * (cx * cy - sz * sx), sc * ((-sx) * sy + cy * cz), sc * ((-sy) * sz + cz * cx) };
            float gl = fmaxf(sqrtf(o_dot3(gr[0], gr[1], gr[2], gr[0], gr[1], gr[2])), 1e-5f);
            float sg = o_signf(g);
            float step = (fabsf(g) - th) / gl;
            for (int k = 0; k < 3; ++k) { nL[k] = (sg * gr[k]) / gl; qL[k] = pL[k] - nL[k] * step; }
            return 1;
        }
        return 0;
    }
    default: return 0;
    }
}

/* Shape projection: returns hit, fills qL and nL (local space). */
static int o_shape_project(const OObb* b, const float pL[3], float qL[3], float nL[3]);

static void o_obb_one(OParticle* p, const OObb* b) {
    if (p->isGhost != 0) return;                             /* :46 */
    float d[3] = { p->pos[0] - b->c[0], p->pos[1] - b->c[1], p->pos[2] - b->c[2] };
    float pL[3];                                             /* worldToLocal :32-36 */
    for (int a = 0; a < 3; ++a) pL[a] = o_dot3(d[0], d[1], d[2], b->R[3 * a], b->R[3 * a + 1], b->R[3 * a + 2]);
    float qL[3] = { pL[0], pL[1], pL[2] }, nL[3] = { 0, 0, 0 };
    int hit = o_shape_project(b, pL, qL, nL);
    if (hit) {                                               /* :311-327 */
        float nW[3], t[3];
        o_matvec(b->R, nL, nW);
        float len = sqrtf(o_dot3(nW[0], nW[1], nW[2], nW[0], nW[1], nW[2]));
        nW[0] = nW[0] / len; nW[1] = nW[1] / len; nW[2] = nW[2] / len;
        o_matvec(b->R, qL, t);
        for (int a = 0; a < 3; ++a) p->pos[a] = b->c[a] + t[a];
        float vn = o_dot3(p->vel[0], p->vel[1], p->vel[2], nW[0], nW[1], nW[2]);
        for (int a = 0; a < 3; ++a) {
            float vN = vn * nW[a];
            float vT = p->vel[a] - vN;
            float vNn = (-b->e) * vN;
            float vTn = (1.0f - b->f) * vT;
            p->vel[a] = vNn + vTn;
        }
    }
}

static int o_shape_project(const OObb* b, const float pL[3], float qL[3], float nL[3]) {
    const float hx = b->half[0], hy = b->half[1], hz = b->half[2];
    if (b->shape >= 7 && b->shape <= 14) return o_shape_project_ext(b, pL, qL, nL);
    switch (b->shape) {
    case 1: {                                                /* sphere :60-68 */
        float R = hx;
        float d = sqrtf(o_dot3(pL[0], pL[1], pL[2], pL[0], pL[1], pL[2]));
        if (d > R) {
            if (d > 1e-6f) { nL[0] = pL[0] / d; nL[1] = pL[1] / d; nL[2] = pL[2] / d; }
            else { nL[0] = 0; nL[1] = 1; nL[2] = 0; }
            qL[0] = nL[0] * R; qL[1] = nL[1] * R; qL[2] = nL[2] * R;
            return 1;
        }
        return 0;
    }
    case 2: {                                                /* cylinder :69-82 */
        float R = hx, H = hy;
        float rad = sqrtf(fmaf(pL[2], pL[2], pL[0] * pL[0]));
        float qx = pL[0], qz = pL[2];
        if (rad > R) { float s = R / fmaxf(rad, 1e-6f); qx = pL[0] * s; qz = pL[2] * s; }
        qL[0] = qx; qL[1] = o_clampf(pL[1], -H, H); qL[2] = qz;
        float de[3] = { pL[0] - qL[0], pL[1] - qL[1], pL[2] - qL[2] };
        float dl = sqrtf(o_dot3(de[0], de[1], de[2], de[0], de[1], de[2]));
        if (dl > 1e-6f) { nL[0] = de[0] / dl; nL[1] = de[1] / dl; nL[2] = de[2] / dl; return 1; }
        return 0;
    }
    case 3: {                                                /* torus :83-97 */
        float R = hx, r = hy;
        float lxz = sqrtf(fmaf(pL[2], pL[2], pL[0] * pL[0]));
        float rdx = 1.0f, rdz = 0.0f;
        if (lxz > 1e-6f) { rdx = pL[0] / lxz; rdz = pL[2] / lxz; }
        float ring[3] = { rdx * R, 0.0f, rdz * R };
        float d[3] = { pL[0] - ring[0], pL[1] - ring[1], pL[2] - ring[2] };
        float dl = sqrtf(o_dot3(d[0], d[1], d[2], d[0], d[1], d[2]));
        if (dl > r) {
            float m = fmaxf(dl, 1e-6f);
            for (int a = 0; a < 3; ++a) { nL[a] = d[a] / m; qL[a] = ring[a] + nL[a] * r; }
            return 1;
        }
        return 0;
    }
    case 4: {                                                /* capsule :98-110 */
        float R = hx, H = hy;
        float seg[3] = { 0.0f, o_clampf(pL[1], -H, H), 0.0f };
        float d[3] = { pL[0] - seg[0], pL[1] - seg[1], pL[2] - seg[2] };
        float dl = sqrtf(o_dot3(d[0], d[1], d[2], d[0], d[1], d[2]));
        if (dl > R) {
            float m = fmaxf(dl, 1e-6f);
            for (int a = 0; a < 3; ++a) { nL[a] = d[a] / m; qL[a] = seg[a] + nL[a] * R; }
            return 1;
        }
        return 0;
    }
    case 5: {                                                /* hourglass :111-129 */
        float baseR = hx, H = fmaxf(hy, 1e-6f), neckR = fminf(hz, baseR);
        float yC = o_clampf(pL[1], -H, H);
        float rMax = neckR + ((baseR - neckR) * fabsf(yC)) / H;
        float lxz = sqrtf(fmaf(pL[2], pL[2], pL[0] * pL[0]));
        float qx = pL[0], qz = pL[2];
        if (lxz > rMax) { float s = rMax / fmaxf(lxz, 1e-6f); qx = pL[0] * s; qz = pL[2] * s; }
        qL[0] = qx; qL[1] = yC; qL[2] = qz;
        float de[3] = { pL[0] - qL[0], pL[1] - qL[1], pL[2] - qL[2] };
        float dl = sqrtf(o_dot3(de[0], de[1], de[2], de[0], de[1], de[2]));
        if (dl > 1e-6f) { nL[0] = de[0] / dl; nL[1] = de[1] / dl; nL[2] = de[2] / dl; return 1; }
        return 0;
    }
    case 6: {                                                /* egg :130-143 */
        float a = fmaxf(hx, 1e-6f), bb = fmaxf(hy, 1e-6f);
        float e[3] = { a, bb, a };
        float u[3] = { pL[0] / e[0], pL[1] / e[1], pL[2] / e[2] };
        float d = sqrtf(o_dot3(u[0], u[1], u[2], u[0], u[1], u[2]));
        if (d > 1.0f) {
            float gq[3];
            for (int k = 0; k < 3; ++k) { qL[k] = (u[k] / d) * e[k]; gq[k] = qL[k] / (e[k] * e[k]); }
            float gl = sqrtf(o_dot3(gq[0], gq[1], gq[2], gq[0], gq[1], gq[2]));
            for (int k = 0; k < 3; ++k) nL[k] = gq[k] / gl;
            return 1;
        }
        return 0;
    }
    default: {                                               /* box :297-309 */
        qL[0] = o_clampf(pL[0], -hx, hx); qL[1] = o_clampf(pL[1], -hy, hy); qL[2] = o_clampf(pL[2], -hz, hz);
        float de[3] = { pL[0] - qL[0], pL[1] - qL[1], pL[2] - qL[2] };
        float d[3] = { fabsf(de[0]), fabsf(de[1]), fabsf(de[2]) };
        if (d[0] > 0.0f || d[1] > 0.0f || d[2] > 0.0f) {
            if (d[0] >= d[1] && d[0] >= d[2]) { nL[0] = o_signf(de[0]); nL[1] = 0; nL[2] = 0; }
            else if (d[1] >= d[0] && d[1] >= d[2]) { nL[0] = 0; nL[1] = o_signf(de[1]); nL[2] = 0; }
            else { nL[0] = 0; nL[1] = 0; nL[2] = o_signf(de[2]); }
            return 1;
        }
        return 0;
    }
    }
}

/* Shape types this oracle restates: 0 box, 1 sphere, 2 cylinder, 3 torus, 4 capsule,
 * 5 hourglass, 6 egg, 7 star, 8 superellipsoid, 9 trefoil, 10 Moebius, 11 DNA, 12 heart,
 * 13 gyroid, 14 coil (anything else is the box branch, as in the shader's final else). */
int sph_oracle_shape_supported(int shape) { return shape >= 0 && shape <= 14; }

void sph_oracle_obb(OParticle* P, int n, const OParams* p) {
    OObb b;
    o_obb_setup(p, &b);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) o_obb_one(&P[i], &b);
}

/* SPH pass only (no OBB): in -> out, snapshot semantics. */
void sph_oracle_sph_pass(const OParticle* in, OParticle* out, int n, const OParams* p, float overrideDt) {
    const float dt = (overrideDt > 0.0f) ? overrideDt : p->timeStep;   /* SPHFluid3D.cpp:434 */
    OGrid g;
    OConsts k;
    sph_oracle_grid_extents(p, &g);
    o_consts(p, dt, &k);
    int32_t* cellStart = (int32_t*)malloc(sizeof(int32_t) * (size_t)(g.numCells + 1));
    int32_t* sorted = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t* pcell = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    sph_oracle_build_grid(in, n, &g, cellStart, sorted, pcell, NULL, NULL);
    if (g_contract == 1) {
#pragma omp parallel for schedule(dynamic, 256)
        for (int i = 0; i < n; ++i) o_sph_one(i, in, out, &g, cellStart, sorted, &k);
    } else {
#pragma omp parallel for schedule(dynamic, 256)
        for (int i = 0; i < n; ++i) o_sph_one_literal(i, in, out, &g, cellStart, sorted, &k, g_contract == 2);
    }
    free(cellStart); free(sorted); free(pcell);
}

/* DispatchCompute, SPHFluid3D.cpp:431-509: ClearGrid -> BuildGrid -> SPHFluid -> OBB.
 * P is updated in place; scratch (n particles) is caller-provided. */
void sph_oracle_substep(OParticle* P, OParticle* scratch, int n, const OParams* p, float overrideDt) {
    if (p->pause) return;                                    /* :432 */
    sph_oracle_sph_pass(P, scratch, n, p, overrideDt);
    sph_oracle_obb(scratch, n, p);
    memcpy(P, scratch, sizeof(OParticle) * (size_t)n);
}

/* ApplyWaveImpulse + WaveImpulse.comp */
void sph_oracle_wave_impulse(OParticle* P, int n, float amplitude, float wavelength, float phase,
                             const float dir[3], float yMin, float yMax) {
    if (amplitude == 0.0f || wavelength <= 1e-6f) return;    /* SPHFluid3D.cpp:607 */
    float len = sqrtf(o_dot3(dir[0], dir[1], dir[2], dir[0], dir[1], dir[2]));
    float nd[3] = { 0.0f, 1.0f, 0.0f };
    if (len > 1e-6f) { nd[0] = dir[0] / len; nd[1] = dir[1] / len; nd[2] = dir[2] / len; }
    const float kk = 6.28318530718f / wavelength;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        OParticle* p = &P[i];
        if (p->isGhost != 0) continue;
        if (p->pos[1] < yMin || p->pos[1] > yMax) continue;
        float theta = fmaf(kk, o_dot3(p->pos[0], p->pos[1], p->pos[2], nd[0], nd[1], nd[2]), phase);
        float kick = amplitude * sph_oracle_sinf(theta);
        for (int a = 0; a < 3; ++a) p->vel[a] = fmaf(nd[a], kick, p->vel[a]);
    }
}

/* ------------------------------------------------------------------ spawn */

typedef struct { uint64_t state, inc; } OPcg;
static uint32_t o_pcg_next(OPcg* r) {
    uint64_t old = r->state;
    r->state = old * 6364136223846793005ULL + r->inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}
static void o_pcg_seed(OPcg* r, uint64_t seed) {
    r->state = 0u; r->inc = (54u << 1u) | 1u;
    o_pcg_next(r); r->state += seed; o_pcg_next(r);
}
/* ====================================================================================
 * Fountain recycle: DispatchCompute step 6 (SPHFluid3D.cpp:519, 526-544) and
 * shaders/FountainRecycle.comp:24-54.  cos / sin are the pinned routines (5, 10).
 * ==================================================================================== */
typedef struct {
    int32_t mode;                  /* fountainMode, SPHFluid3D.h:161 */
    float offset[3];               /* fountainOffset :162 */
    float radius, spread;          /* :163-164 */
    float jetSpeedLive;            /* :165 */
    float drainLevel, drainPerSec; /* :166-167 */
    uint32_t seed;                 /* fountainSeed :168 */
} OFountain;

static inline float o_lcg(uint32_t* s) {                     /* FountainRecycle.comp:24-27 */
    *s = *s * 1664525u + 1013904223u;
    return (float)(*s & 0xFFFFFFu) / 16777215.0f;
}

void sph_oracle_fountain(OParticle* P, int n, const OParams* p, const OFountain* f, float dt, uint32_t seed) {
    float half[3];
    sph_oracle_effective_half(p, half);
    const float emit[3] = { p->boxCenter[0] + f->offset[0], p->boxCenter[1] + f->offset[1], p->boxCenter[2] + f->offset[2] };
    const float drainY = (p->boxCenter[1] - half[1]) + f->drainLevel;          /* :536-537 */
    const float chance = fminf(1.0f, f->drainPerSec * dt);                     /* :538-539 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        OParticle* q = &P[i];
        if (q->isGhost == 1) continue;                       /* :34 */
        if (q->pos[1] >= drainY) continue;                   /* :35 */
        uint32_t s = ((uint32_t)i ^ (seed * 747796405u)) + 2891336453u;
        if (o_lcg(&s) > chance) continue;                    /* :38 */
        float r1 = o_lcg(&s), r2 = o_lcg(&s), r3 = o_lcg(&s), r4 = o_lcg(&s);
        float ang = 6.2831853f * r1;
        float rad = f->radius * sqrtf(r2);
        float ca = sph_oracle_cosf(ang), sa = sph_oracle_sinf(ang);
        q->pos[0] = emit[0] + ca * rad;
        q->pos[1] = emit[1] + 0.2f * r3;
        q->pos[2] = emit[2] + sa * rad;
        float sm = f->spread * r4;
        float sx = ca * sm, sz = sa * sm;
        float len = sqrtf(o_dot3(sx, 1.0f, sz, sx, 1.0f, sz));
        q->vel[0] = f->jetSpeedLive * (sx / len);
        q->vel[1] = f->jetSpeedLive * (1.0f / len);
        q->vel[2] = f->jetSpeedLive * (sz / len);
        q->acc[0] = q->acc[1] = q->acc[2] = q->acc[3] = 0.0f;
        q->density = p->restDensity;
        q->pressure = 0.0f;
    }
}

/* DispatchCompute with the fountain step: f->seed advances per dispatch as fountainSeed++ (:541). */
void sph_oracle_substep_fountain(OParticle* P, OParticle* scratch, int n, const OParams* p, float overrideDt, OFountain* f) {
    if (p->pause) return;
    sph_oracle_substep(P, scratch, n, p, overrideDt);
    if (f && f->mode) {
        const float dt = (overrideDt > 0.0f) ? overrideDt : p->timeStep;
        sph_oracle_fountain(P, n, p, f, dt, f->seed);
        f->seed = f->seed + 1u;
    }
}

static float o_pcg_uniform(OPcg* r, float lo, float hi) {
    float u = (float)(o_pcg_next(r) >> 8) * (1.0f / 16777216.0f);
    return lo + u * (hi - lo);
}

/* insideShape lambda, SPHFluid3D.cpp:167-289.  Host code in the reference too: libm here. */
static int o_inside_shape(const OParams* p, const float hf[3], float margin, float lx, float ly, float lz) {
    switch (p->shapeType) {
    case 1: { float r = hf[0] - margin; return lx * lx + ly * ly + lz * lz <= r * r; }
    case 2: { float r = hf[0] - margin; return lx * lx + lz * lz <= r * r && fabsf(ly) <= hf[1] - margin; }
    case 3: { float R = p->boxHalf[0], r = p->boxHalf[1] - margin;
              float dr = sqrtf(lx * lx + lz * lz) - R;
              return r > 0.0f && (dr * dr + ly * ly) <= r * r; }
    case 4: { float r = p->boxHalf[0] - margin, H = p->boxHalf[1];
              float dy = ly - o_clampf(ly, -H, H);
              return (lx * lx + lz * lz + dy * dy) <= r * r; }
    case 5: { float baseR = p->boxHalf[0], H = fmaxf(p->boxHalf[1], 1e-6f);
              float neckR = fminf(p->boxHalf[2], baseR);
              if (fabsf(ly) > H - margin) return 0;
              float rMax = neckR + (baseR - neckR) * fabsf(ly) / H - margin;
              return rMax > 0.0f && (lx * lx + lz * lz) <= rMax * rMax; }
    case 6: { float a = fmaxf(p->boxHalf[0] - margin, 1e-4f), b = fmaxf(p->boxHalf[1] - margin, 1e-4f);
              float u = lx / a, v = ly / b, w = lz / a;
              return (u * u + v * v + w * w) <= 1.0f; }
    case 7: { float R = p->boxHalf[0], H = p->boxHalf[1];
              float pts = fmaxf(3.0f, p->shapeAux[0]), depth = o_clampf(p->shapeAux[1], 0.0f, 0.9f);
              if (fabsf(ly) > H - margin) return 0;
              float ang = atan2f(lz, lx);
              float rMax = R * (1.0f - depth * (0.5f + 0.5f * cosf(pts * ang))) - margin;
              return rMax > 0.0f && (lx * lx + lz * lz) <= rMax * rMax; }
    case 8: { float a = fmaxf(p->boxHalf[0] - margin, 1e-4f), b = fmaxf(p->boxHalf[1] - margin, 1e-4f);
              float n = o_clampf(p->shapeAux[2], 0.6f, 8.0f);
              float F = powf(fabsf(lx) / a, n) + powf(fabsf(ly) / b, n) + powf(fabsf(lz) / a, n);
              return F <= 1.0f; }
    case 9: case 11: case 12: case 14: {
              float r = p->boxHalf[1] - margin;
              if (r <= 0.0f) return 0;
              float tab[384], b0[3];
              int cnt = sph_oracle_shape_table(p, tab, b0);   /* same samples as the shader (H, S unshrunk, :216-262) */
              float bestD2 = 1e30f;
              for (int k = 0; k < cnt; ++k) {
                  float dx = lx - tab[3 * k], dy = ly - tab[3 * k + 1], dz = lz - tab[3 * k + 2];
                  bestD2 = fminf(bestD2, dx * dx + dy * dy + dz * dz);
              }
              return bestD2 <= r * r; }
    case 10: { float R = p->boxHalf[0], wHalf = p->boxHalf[1] - margin, tHalf = fmaxf(p->shapeAux[0], 0.05f) - margin;
              if (wHalf <= 0.0f || tHalf <= 0.0f) return 0;
              float phi = atan2f(lz, lx);
              float erx = cosf(phi), erz = sinf(phi);
              float ox = lx - R * erx, oy = ly, oz = lz - R * erz;
              float psi = 0.5f * phi;
              float cw = cosf(psi), sw = sinf(psi);
              float du = ox * (cw * erx) + oy * (sw) + oz * (cw * erz);
              float dv = ox * (-sw * erx) + oy * (cw) + oz * (-sw * erz);
              return fabsf(du) <= wHalf && fabsf(dv) <= tHalf; }
    case 13: { float R = p->boxHalf[0] - margin;
              float sc = fmaxf(p->shapeAux[0], 0.1f), th = o_clampf(p->shapeAux[1], 0.2f, 2.5f);
              if (lx * lx + ly * ly + lz * lz > R * R) return 0;
              float qx = lx * sc, qy = ly * sc, qz = lz * sc;
              float g = sinf(qx) * cosf(qy) + sinf(qy) * cosf(qz) + sinf(qz) * cosf(qx);
              return fabsf(g) <= th; }
    default: return 1;
    }
}

/* InitializeParticles standard fill, SPHFluid3D.cpp:85-102,159-332.
 * Returns the particle count actually produced (<= nRequested) and writes
 * param_mass = rho0 * spacing^3 (:92) to *massOut.  Jitter draws: three per lattice
 * point in x,y,z order, whether or not the point is kept (as in :296-299). */
int sph_oracle_spawn(const OParams* p, int nRequested, uint32_t seed, OParticle* out, float* massOut) {
    const float h = p->h;
    const float spacing = h * 0.85f;
    *massOut = p->restDensity * spacing * spacing * spacing;
    const float fillFraction = 0.4f;
    OPcg rng; o_pcg_seed(&rng, seed);
    const float jlo = -spacing * p->jitterAmp, jhi = spacing * p->jitterAmp;
    float hf[3];
    sph_oracle_effective_half(p, hf);
    const float margin = spacing * 0.5f;
    int layersY = (int)((2.0f * hf[1] * fillFraction) / spacing); if (layersY < 1) layersY = 1;
    int sideX = (int)((hf[0] * 1.7f) / spacing); if (sideX < 1) sideX = 1;
    int sideZ = (int)((hf[2] * 1.7f) / spacing); if (sideZ < 1) sideZ = 1;
    int count = 0;
    for (int x = 0; x < sideX && count < nRequested; ++x)
        for (int y = 0; y < layersY && count < nRequested; ++y)
            for (int z = 0; z < sideZ && count < nRequested; ++z) {
                float jx = p->useJitter ? o_pcg_uniform(&rng, jlo, jhi) : 0.0f;
                float jy = p->useJitter ? o_pcg_uniform(&rng, jlo, jhi) : 0.0f;
                float jz = p->useJitter ? o_pcg_uniform(&rng, jlo, jhi) : 0.0f;
                float lx = -hf[0] * 0.85f + (float)x * spacing + jx;
                float ly = -hf[1] + spacing + (float)y * spacing + jy;
                float lz = -hf[2] * 0.85f + (float)z * spacing + jz;
                if (!o_inside_shape(p, hf, margin, lx, ly, lz)) continue;
                OParticle q; memset(&q, 0, sizeof(q));
                q.pos[0] = p->boxCenter[0] + lx; q.pos[1] = p->boxCenter[1] + ly; q.pos[2] = p->boxCenter[2] + lz;
                switch (p->mixPattern) {                     /* :307-311 */
                case 1: q.padC = (x + y + z) & 1; break;
                case 2: q.padC = (int)(o_pcg_next(&rng) & 1u); break;
                default: q.padC = (lx < 0.0f) ? 0 : 1; break;
                }
                float d;                                     /* :316-329 */
                switch (p->dyePattern) {
                case 1: d = (ly + hf[1]) / fmaxf(2.0f * hf[1], 1e-3f); break;
                case 2: { float nn = sinf(lx * 1.3f) * cosf(lz * 1.7f) + sinf(ly * 1.1f + lx * 0.7f);
                          d = 0.5f + 0.5f * sinf(nn * 2.3f); break; }
                default: d = 0.5f + 0.5f * sinf(lx * 1.5f); break;
                }
                q.padB = o_clampf(d, 0.0f, 1.0f);
                out[count++] = q;
            }
    return count;
}

/* --------------------------------------------------------- CPU baseline leg */

int sph_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void sph_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ====================================================================================
 * Per-frame impulse kernels.  Map-shaped, no neighbours.
 * Contraction policy as above: dot products are fma chains, everything else rounds
 * separately.  mix(a,b,t) = a*(1-t) + b*t, smoothstep = t*t*(3-2t) with
 * t = clamp((x-e0)/(e1-e0), 0, 1), fract(x) = x - floor(x) (GLSL definitions).
 * ==================================================================================== */
static inline float o_smoothstep(float e0, float e1, float x) {
    float t = o_clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}

/* ApplyVortexImpulse (SPHFluid3D.cpp:627-646) + shaders/VortexImpulse.comp:32-49 */
void sph_oracle_vortex_impulse(OParticle* P, int n, const OParams* prm, float tangentKick, float inwardKick) {
    if (fabsf(tangentKick) < 1e-6f && fabsf(inwardKick) < 1e-6f) return;
    float R[9], half[3];
    o_rotation(prm->boxEulerDeg, R);
    sph_oracle_effective_half(prm, half);
    const float ax = R[3], ay = R[4], az = R[5];             /* local +Y in world */
    const float radius = fmaxf(half[0], half[2]);
    const float e1 = 0.35f * fmaxf(radius, 1e-4f);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        OParticle* p = &P[i];
        if (p->isGhost != 0) continue;
        float rx = p->pos[0] - prm->boxCenter[0], ry = p->pos[1] - prm->boxCenter[1], rz = p->pos[2] - prm->boxCenter[2];
        float d = o_dot3(rx, ry, rz, ax, ay, az);
        float qx = rx - ax * d, qy = ry - ay * d, qz = rz - az * d;
        float r = sqrtf(o_dot3(qx, qy, qz, qx, qy, qz));
        if (r < 1e-4f) continue;
        float hx = qx / r, hy = qy / r, hz = qz / r;
        float tx = ay * hz - az * hy, ty = az * hx - ax * hz, tz = ax * hy - ay * hx;   /* cross(axis, rHat) */
        float fall = o_smoothstep(0.0f, e1, r);
        float kt = tangentKick * fall, ki = inwardKick * fall;
        p->vel[0] = p->vel[0] + (tx * kt - hx * ki);
        p->vel[1] = p->vel[1] + (ty * kt - hy * ki);
        p->vel[2] = p->vel[2] + (tz * kt - hz * ki);
    }
}

/* ApplyAttractorImpulse (SPHFluid3D.cpp:650-664) + shaders/AttractorImpulse.comp:29-45 */
void sph_oracle_attractor_impulse(OParticle* P, int n, const float point[3], float pullKick, float radius) {
    if (fabsf(pullKick) < 1e-6f) return;
    const float uRadius = fmaxf(radius, 0.1f);
    const float uSoften = fmaxf(0.15f * radius, 0.2f);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        OParticle* p = &P[i];
        if (p->isGhost != 0) continue;
        float rx = point[0] - p->pos[0], ry = point[1] - p->pos[1], rz = point[2] - p->pos[2];
        float d = sqrtf(o_dot3(rx, ry, rz, rx, ry, rz));
        if (d < 1e-5f) continue;
        float pull = (pullKick * uSoften) / (d + uSoften);
        pull = pull * (1.0f - o_smoothstep(0.6f * uRadius, uRadius, d));
        p->vel[0] = p->vel[0] + (rx / d) * pull;
        p->vel[1] = p->vel[1] + (ry / d) * pull;
        p->vel[2] = p->vel[2] + (rz / d) * pull;
    }
}

/* ApplyStencilAttract (SPHFluid3D.cpp:695-710) + shaders/StencilAttract.comp:31-44.
 * targets: nTargets x 4 floats (w unused), SetStencilTargets (:684-693). */
void sph_oracle_stencil_attract(OParticle* P, int n, const float* targets, int nTargets, float pullKick, float dampKick) {
    if (nTargets <= 0 || !targets) return;
    if (fabsf(pullKick) < 1e-6f && dampKick < 1e-6f) return;
    const float uDamp = fminf(dampKick, 0.5f);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        OParticle* p = &P[i];
        if (p->isGhost != 0) continue;
        const float* t = targets + 4 * ((unsigned)i % (unsigned)nTargets);
        for (int a = 0; a < 3; ++a) {
            float d = t[a] - p->pos[a];
            float v = p->vel[a] + d * pullKick;
            p->vel[a] = v * (1.0f - uDamp);
        }
    }
}

static inline float o_fract(float x) { return x - floorf(x); }
static float o_hash13(float x, float y, float z) {           /* CurlFlow.comp:30-34 */
    x = o_fract(x * 0.1031f); y = o_fract(y * 0.1031f); z = o_fract(z * 0.1031f);
    float dd = o_dot3(x, y, z, z + 31.32f, y + 31.32f, x + 31.32f);
    x = x + dd; y = y + dd; z = z + dd;
    return o_fract((x + y) * z);
}
static inline float o_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static float o_vnoise(float x, float y, float z) {           /* CurlFlow.comp:36-50 */
    float ix = floorf(x), iy = floorf(y), iz = floorf(z);
    float fx = x - ix, fy = y - iy, fz = z - iz;
    fx = (fx * fx) * (3.0f - 2.0f * fx); fy = (fy * fy) * (3.0f - 2.0f * fy); fz = (fz * fz) * (3.0f - 2.0f * fz);
    float n000 = o_hash13(ix, iy, iz), n100 = o_hash13(ix + 1.0f, iy, iz);
    float n010 = o_hash13(ix, iy + 1.0f, iz), n110 = o_hash13(ix + 1.0f, iy + 1.0f, iz);
    float n001 = o_hash13(ix, iy, iz + 1.0f), n101 = o_hash13(ix + 1.0f, iy, iz + 1.0f);
    float n011 = o_hash13(ix, iy + 1.0f, iz + 1.0f), n111 = o_hash13(ix + 1.0f, iy + 1.0f, iz + 1.0f);
    return o_mix(o_mix(o_mix(n000, n100, fx), o_mix(n010, n110, fx), fy),
                 o_mix(o_mix(n001, n101, fx), o_mix(n011, n111, fx), fy), fz);
}
static inline float o_p1(float x, float y, float z) { return o_vnoise(x, y, z); }
static inline float o_p2(float x, float y, float z) { return o_vnoise(x + 31.416f, y + 47.853f, z + 12.793f); }
static inline float o_p3(float x, float y, float z) { return o_vnoise(x + -233.145f, y + 93.912f, z + 55.121f); }

/* ApplyCurlFlow (SPHFluid3D.cpp:668-681) + shaders/CurlFlow.comp:57-80 */
void sph_oracle_curl_flow(OParticle* P, int n, float kick, float scale, float time) {
    if (fabsf(kick) < 1e-6f) return;
    const float uScale = fmaxf(scale, 1e-3f);
    const float h = 0.35f;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        OParticle* p = &P[i];
        if (p->isGhost != 0) continue;
        float qx = p->pos[0] * uScale, qy = p->pos[1] * uScale, qz = p->pos[2] * uScale + time;
        float dP3dy = o_p3(qx, qy + h, qz) - o_p3(qx, qy - h, qz);
        float dP2dz = o_p2(qx, qy, qz + h) - o_p2(qx, qy, qz - h);
        float dP1dz = o_p1(qx, qy, qz + h) - o_p1(qx, qy, qz - h);
        float dP3dx = o_p3(qx + h, qy, qz) - o_p3(qx - h, qy, qz);
        float dP2dx = o_p2(qx + h, qy, qz) - o_p2(qx - h, qy, qz);
        float dP1dy = o_p1(qx, qy + h, qz) - o_p1(qx, qy - h, qz);
        const float inv = 2.0f * h;
        float cx = (dP3dy - dP2dz) / inv, cy = (dP1dz - dP3dx) / inv, cz = (dP2dx - dP1dy) / inv;
        float m = sqrtf(o_dot3(cx, cy, cz, cx, cy, cz));
        float dx = 0.0f, dy = 0.0f, dz = 0.0f;
        if (m > 1e-5f) { dx = cx / m; dy = cy / m; dz = cz / m; }
        float mm = fminf(m, 1.0f);
        p->vel[0] = p->vel[0] + (dx * mm) * kick;
        p->vel[1] = p->vel[1] + (dy * mm) * kick;
        p->vel[2] = p->vel[2] + (dz * mm) * kick;
    }
}

/* ====================================================================================
 * River / stream mode (SPHFluid3D.h:171-206): GenerateRiverTerrain (SPHFluid3D.cpp:772-878), the river branch of
 * InitializeParticles (:104-160) and step 5 of DispatchCompute (:511-516: TerrainConstraints.comp,
 * ChannelConstraint.comp, StreamEmit.comp, in that order; the fountain step is skipped while riverMode is set, :519).
 * Dead code in the reference's scene (riverMode is never set, Scene0p.cpp:1660), restated for completeness.
 *
 * Semantic 11: std::rand() / RAND_MAX in GenerateRiverTerrain is the Microsoft C runtime's (the reference is a
 *   Visual Studio project, ComponentFramework.vcxproj): holdrand = holdrand * 214013 + 2531011, result
 *   (holdrand >> 16) & 0x7fff, RAND_MAX 0x7fff, srand(seed) sets holdrand = seed.  Host-side sinf / cosf are libm's
 *   (host code in the reference too).
 * Semantic 12: GLSL mix(a, b, t) = a * (1 - t) + b * t as the specification writes it; normalize(v) = v / sqrt(dot);
 *   sin / cos in the shaders = the specified sph_oracle_sinf / sph_oracle_cosf (semantics 6 and 10).
 * Semantic 13: the jitter / fill random numbers of the spawn come from the same PCG32 stream as the standard fill
 *   (semantic: default_random_engine(time(nullptr)) is not reproducible in the reference either).
 * ==================================================================================== */

typedef struct {
    int32_t riverMode;                                        /* SPHFluid3D.h:172 */
    int32_t terrainW, terrainH;                               /* :176-177 */
    float terrainWorldMinX, terrainWorldMinZ, terrainWorldSizeX, terrainWorldSizeZ;   /* :178-181 */
    float riverEmitterPos[3], riverEmitterVel[3];             /* :184-185 */
    float riverEmitterRadius, riverSinkY, riverSinkZMax;      /* :186-188 */
    float riverAmp, riverFreq, riverPhase, riverChannelWidth, riverChannelDepth, riverSlopeDrop;   /* :191-196 */
} ORiver;

int sph_oracle_sizeof_river(void) { return (int)sizeof(ORiver); }

void sph_oracle_river_default(ORiver* r) {
    memset(r, 0, sizeof(*r));
    r->terrainW = 64; r->terrainH = 64;
    r->terrainWorldMinX = -7.0f; r->terrainWorldMinZ = -10.0f; r->terrainWorldSizeX = 14.0f; r->terrainWorldSizeZ = 20.0f;
    r->riverEmitterPos[0] = 0.0f; r->riverEmitterPos[1] = 3.0f; r->riverEmitterPos[2] = -9.0f;
    r->riverEmitterVel[0] = 0.0f; r->riverEmitterVel[1] = -0.5f; r->riverEmitterVel[2] = 4.0f;
    r->riverEmitterRadius = 1.5f; r->riverSinkY = -8.5f; r->riverSinkZMax = 9.0f;
    r->riverAmp = 2.0f; r->riverFreq = 0.25f; r->riverPhase = 0.0f;
    r->riverChannelWidth = 3.0f; r->riverChannelDepth = 3.5f; r->riverSlopeDrop = 0.3f;
}

static float o_msvc_frand(uint32_t* hold) {
    *hold = *hold * 214013u + 2531011u;
    return (float)((*hold >> 16) & 0x7fffu) / 32767.0f;      /* std::rand() / float(RAND_MAX), :774 */
}

/* GenerateRiverTerrain(seed): fills the river members, the heightfield (terrainW * terrainH floats) and sets
 * param_gravityY = -120, param_gravityZ = 0 (:864-865). */
void sph_oracle_river_terrain(OParams* p, int seed, ORiver* r, float* heights) {
    uint32_t hold = (uint32_t)seed;                                          /* :773 */
    r->riverAmp = 0.5f + o_msvc_frand(&hold) * 1.5f;                         /* :777-782 */
    r->riverFreq = 0.18f + o_msvc_frand(&hold) * 0.18f;
    r->riverPhase = o_msvc_frand(&hold) * 6.2831f;
    r->riverChannelWidth = 1.8f + o_msvc_frand(&hold) * 1.2f;
    r->riverChannelDepth = 3.5f + o_msvc_frand(&hold) * 1.0f;
    r->riverSlopeDrop = 0.3f + o_msvc_frand(&hold) * 0.5f;
    const float channelDepth = r->riverChannelDepth, slopeDrop = r->riverSlopeDrop;
    float ph[8];
    for (int k = 0; k < 8; ++k) ph[k] = o_msvc_frand(&hold) * 6.2831f;       /* :787-788 */
    r->terrainWorldMinX = p->boxCenter[0] - p->boxHalf[0];                   /* :792-795 */
    r->terrainWorldMinZ = p->boxCenter[2] - p->boxHalf[2];
    r->terrainWorldSizeX = 2.0f * p->boxHalf[0];
    r->terrainWorldSizeZ = 2.0f * p->boxHalf[2];
    const float xMin = r->terrainWorldMinX, zMin = r->terrainWorldMinZ, xSize = r->terrainWorldSizeX, zSize = r->terrainWorldSizeZ;
    const float yBase = p->boxCenter[1] - p->boxHalf[1];
    const int W = r->terrainW, H = r->terrainH;
    for (int iz = 0; iz < H; ++iz)
        for (int ix = 0; ix < W; ++ix) {
            const float wx = xMin + ((float)ix / (float)(W - 1)) * xSize;   /* :807-808 */
            const float wz = zMin + ((float)iz / (float)(H - 1)) * zSize;
            const float tFlow = (wz - zMin) / zSize;
            const float centerX = p->boxCenter[0] + r->riverAmp * sinf(r->riverFreq * wz + r->riverPhase);
            const float distToRiver = fabsf(wx - centerX);
            const float riverFloor = yBase + 1.0f - tFlow * slopeDrop;      /* :818-819 */
            const float channelEdge = riverFloor + channelDepth;
            float h = channelEdge + 3.0f;                                    /* :824-828 */
            h += 0.5f * sinf(wx * 0.35f + ph[0]) * cosf(wz * 0.28f + ph[1]);
            h += 0.25f * sinf(wx * 0.70f + ph[2]) * sinf(wz * 0.60f + ph[3]);
            h += 0.12f * sinf(wx * 1.40f + ph[4]) * cosf(wz * 1.20f + ph[5]);
            if (distToRiver < r->riverChannelWidth) {                        /* :830-840 */
                const float u = distToRiver / r->riverChannelWidth;
                if (u < 0.50f) h = riverFloor;
                else { const float uw = (u - 0.50f) / (1.0f - 0.50f); h = riverFloor + channelDepth * uw * uw; }
            } else {
                h = fmaxf(h, channelEdge + 0.3f);                            /* :843 */
            }
            h = fmaxf(h, yBase - 0.3f);                                      /* :847 */
            heights[iz * W + ix] = h;
        }
    const float emitterZ = zMin + 0.5f;                                      /* :853-861 */
    r->riverEmitterPos[0] = p->boxCenter[0] + r->riverAmp * sinf(r->riverFreq * emitterZ + r->riverPhase);
    r->riverEmitterPos[1] = (yBase + 1.0f) + channelDepth * 0.5f;
    r->riverEmitterPos[2] = emitterZ;
    r->riverEmitterVel[0] = 0.0f; r->riverEmitterVel[1] = -0.5f; r->riverEmitterVel[2] = 0.5f;
    r->riverEmitterRadius = r->riverChannelWidth * 0.35f;
    r->riverSinkY = yBase + 0.3f;
    r->riverSinkZMax = p->boxCenter[2] + p->boxHalf[2] - 0.5f;
    p->gravity[1] = -120.0f; p->gravity[2] = 0.0f;                           /* :864-865 */
}

/* host-side bilinear sample of the spawn (the lambda at :113-126; NOT the shader's mix form) */
static float o_river_sample_host(const ORiver* r, const float* T, float wx, float wz) {
    float u = (wx - r->terrainWorldMinX) / r->terrainWorldSizeX * (float)(r->terrainW - 1);
    float v = (wz - r->terrainWorldMinZ) / r->terrainWorldSizeZ * (float)(r->terrainH - 1);
    u = fmaxf(0.0f, fminf((float)(r->terrainW - 2), u));
    v = fmaxf(0.0f, fminf((float)(r->terrainH - 2), v));
    const int ix = (int)u, iz = (int)v, W = r->terrainW;
    const float fx = u - (float)ix, fz = v - (float)iz;
    const float h00 = T[ix + iz * W], h10 = T[(ix + 1) + iz * W], h01 = T[ix + (iz + 1) * W], h11 = T[(ix + 1) + (iz + 1) * W];
    return h00 * (1 - fx) * (1 - fz) + h10 * fx * (1 - fz) + h01 * (1 - fx) * fz + h11 * fx * fz;
}

/* InitializeParticles, river branch (:104-160).  Returns the particle count (always nRequested). */
int sph_oracle_river_spawn(const OParams* p, const ORiver* r, const float* T, int nRequested, uint32_t seed, OParticle* out, float* massOut) {
    const float spacing = p->h * 0.85f;
    *massOut = p->restDensity * spacing * spacing * spacing;
    OPcg rng; o_pcg_seed(&rng, seed);
    const float jlo = -spacing * p->jitterAmp, jhi = spacing * p->jitterAmp;
    const float zMin = r->terrainWorldMinZ, zSize = r->terrainWorldSizeZ;
    int count = 0;
    for (float wz = zMin + spacing; wz < zMin + zSize - spacing && count < nRequested; wz += spacing) {
        const float centerX = p->boxCenter[0] + r->riverAmp * sinf(r->riverFreq * wz + r->riverPhase);
        for (float wx = centerX - r->riverChannelWidth; wx <= centerX + r->riverChannelWidth && count < nRequested; wx += spacing) {
            const float ty = o_river_sample_host(r, T, wx, wz);
            for (float wy = ty + spacing; wy <= ty + 2.5f && count < nRequested; wy += spacing) {
                OParticle q; memset(&q, 0, sizeof(q));
                const float jx = p->useJitter ? o_pcg_uniform(&rng, jlo, jhi) : 0.0f;   /* argument order of Vec4(wx + j(), wy + j(), wz + j()) */
                const float jy = p->useJitter ? o_pcg_uniform(&rng, jlo, jhi) : 0.0f;
                const float jz = p->useJitter ? o_pcg_uniform(&rng, jlo, jhi) : 0.0f;
                q.pos[0] = wx + jx; q.pos[1] = wy + jy; q.pos[2] = wz + jz;
                q.vel[2] = 0.5f;
                q.padC = count & 1;
                out[count++] = q;
            }
        }
    }
    while (count < nRequested) {                                             /* :144-159 */
        const float hw = r->riverChannelWidth * 0.5f;
        const float wx = r->riverEmitterPos[0] + o_pcg_uniform(&rng, -hw, hw);
        const float wz = r->riverEmitterPos[2] + o_pcg_uniform(&rng, -hw, hw);
        const float ty = o_river_sample_host(r, T, wx, wz);
        OParticle q; memset(&q, 0, sizeof(q));
        q.pos[0] = wx; q.pos[1] = ty + o_pcg_uniform(&rng, 0.0f, 1.5f); q.pos[2] = wz;
        q.vel[2] = 2.0f;
        q.padC = count & 1;
        out[count++] = q;
    }
    return count;
}

/* mix(): o_mix above (semantic 12 = the definition the curl-noise restatement already uses) */

/* TerrainConstraints.comp:21-34 */
static float o_terrain_height(const ORiver* r, const float* T, float wx, float wz) {
    float u = (wx - r->terrainWorldMinX) / r->terrainWorldSizeX * (float)(r->terrainW - 1);
    float v = (wz - r->terrainWorldMinZ) / r->terrainWorldSizeZ * (float)(r->terrainH - 1);
    u = o_clampf(u, 0.0f, (float)(r->terrainW - 2));
    v = o_clampf(v, 0.0f, (float)(r->terrainH - 2));
    const int ix = (int)u, iz = (int)v, W = r->terrainW;
    const float fx = u - (float)ix, fz = v - (float)iz;
    const float h00 = T[ix + iz * W], h10 = T[(ix + 1) + iz * W], h01 = T[ix + (iz + 1) * W], h11 = T[(ix + 1) + (iz + 1) * W];
    return o_mix(o_mix(h00, h10, fx), o_mix(h01, h11, fx), fz);
}

/* Step 5 of DispatchCompute (:511-516) on every particle: the three passes touch only their own particle, in order. */
void sph_oracle_river(OParticle* P, int n, const OParams* p, const ORiver* r, const float* T) {
    const float restitution = 0.02f, friction = 0.05f;                      /* :554-555 */
    const float flowGravity = 80.0f;                                        /* :572 */
    const float dt = p->timeStep;                                           /* :573: param_timeStep, not the override */
    const float spreadZ = r->riverSinkZMax - r->riverEmitterPos[2];         /* :590 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        OParticle* q = &P[i];
        if (q->isGhost == 1) continue;                                      /* all three shaders */
        /* ---- TerrainConstraints.comp:50-80 ---- */
        {
            const float wx = q->pos[0], wz = q->pos[2];
            const int inside = !(wx < r->terrainWorldMinX || wx > r->terrainWorldMinX + r->terrainWorldSizeX ||
                                 wz < r->terrainWorldMinZ || wz > r->terrainWorldMinZ + r->terrainWorldSizeZ);
            if (inside) {
                const float ty = o_terrain_height(r, T, wx, wz);
                if (q->pos[1] < ty) {
                    const float ddx = r->terrainWorldSizeX / (float)(r->terrainW - 1), ddz = r->terrainWorldSizeZ / (float)(r->terrainH - 1);   /* :38-39 */
                    const float hR = o_terrain_height(r, T, wx + ddx, wz), hL = o_terrain_height(r, T, wx - ddx, wz);
                    const float hF = o_terrain_height(r, T, wx, wz + ddz), hB = o_terrain_height(r, T, wx, wz - ddz);
                    const float nx = hL - hR, ny = 2.0f * ddx, nz = hB - hF;
                    const float nl = sqrtf(o_dot3(nx, ny, nz, nx, ny, nz));
                    const float Nx = nx / nl, Ny = ny / nl, Nz = nz / nl;
                    q->pos[1] = ty + 0.001f;                                /* :66 */
                    const float vN = o_dot3(q->vel[0], q->vel[1], q->vel[2], Nx, Ny, Nz);
                    if (vN < 0.0f) {                                        /* :71-76 */
                        const float ax = vN * Nx, ay = vN * Ny, az = vN * Nz;
                        const float tx = q->vel[0] - ax, tyv = q->vel[1] - ay, tz = q->vel[2] - az;
                        q->vel[0] = -restitution * ax + (1.0f - friction) * tx;
                        q->vel[1] = -restitution * ay + (1.0f - friction) * tyv;
                        q->vel[2] = -restitution * az + (1.0f - friction) * tz;
                    }
                }
            }
        }
        /* ---- ChannelConstraint.comp:27-46 ---- */
        {
            const float wz = q->pos[2];
            const float arg = r->riverFreq * wz + r->riverPhase;
            const float cx = p->boxCenter[0] + r->riverAmp * sph_oracle_sinf(arg);
            const float dx = q->pos[0] - cx;
            const float tdx = r->riverAmp * r->riverFreq * sph_oracle_cosf(arg);
            const float tlen = sqrtf(tdx * tdx + 1.0f);
            const float tX = tdx / tlen, tZ = 1.0f / tlen;
            q->vel[0] = q->vel[0] + tX * flowGravity * dt;
            q->vel[2] = q->vel[2] + tZ * flowGravity * dt;
            if (fabsf(dx) > r->riverChannelWidth) {
                q->pos[0] = cx + o_signf(dx) * r->riverChannelWidth;
                if (dx * q->vel[0] > 0.0f) q->vel[0] = 0.0f;
            }
        }
        /* ---- StreamEmit.comp:31-60 ---- */
        if (q->pos[1] < r->riverSinkY || q->pos[2] > r->riverSinkZMax) {
            uint32_t s = (uint32_t)i * 1664525u + 1013904223u;
            const float r1 = (float)(s & 0xFFFFu) / 65535.0f;
            s = s * 1664525u + 1013904223u;                                 /* r2 is drawn and unused, :40-41 */
            s = s * 1664525u + 1013904223u;
            const float r3 = (float)(s & 0xFFFFu) / 65535.0f;
            s = s * 1664525u + 1013904223u;
            const float r4 = (float)(s & 0xFFFFu) / 65535.0f;
            const float spawnZ = r->riverEmitterPos[2] + r1 * spreadZ;
            const float cx = p->boxCenter[0] + r->riverAmp * sph_oracle_sinf(r->riverFreq * spawnZ + r->riverPhase);
            q->pos[0] = cx + (r4 - 0.5f) * 2.0f * r->riverEmitterRadius;
            q->pos[1] = r->riverEmitterPos[1] + r3 * 0.6f;
            q->pos[2] = spawnZ;
            q->vel[0] = r->riverEmitterVel[0]; q->vel[1] = r->riverEmitterVel[1]; q->vel[2] = r->riverEmitterVel[2]; q->vel[3] = 0.0f;
            q->acc[0] = q->acc[1] = q->acc[2] = q->acc[3] = 0.0f;
            q->density = p->restDensity;
            q->pressure = 0.0f;
        }
    }
}

/* DispatchCompute in river mode (:431-522): the substep, then step 5; the fountain step does not run (:519). */
void sph_oracle_substep_river(OParticle* P, OParticle* scratch, int n, const OParams* p, float overrideDt, const ORiver* r, const float* T) {
    if (p->pause) return;
    sph_oracle_substep(P, scratch, n, p, overrideDt);
    if (r && r->riverMode && T) sph_oracle_river(P, n, p, r, T);
}
