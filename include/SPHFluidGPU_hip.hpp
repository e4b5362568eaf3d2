// SPHFluidGPU_hip.hpp -- header-only C++17 twin of the reference class `SPHFluidGPU`
// (/root/reference/ComponentFramework/SPHFluid3D.h:26-210) on top of the C-ABI in sph_abi.h.
//
// Scene0p-style code compiles against this header unchanged for the hot path: same class
// name, same public member names (`param_*`, `numParticles`, `particles`, `gridSizeX/Y/Z`,
// `numCells`, `gridMinV`, `cellSize`, `box`), same method names and argument meaning
// (`DispatchCompute`, `ResetSimulation`, `ApplyWaveImpulse`, `EffectiveHalf`,
// `ComputeGridExtents`, `GetNumFluids`, `GetFluidVBO`).  The GL object ids Scene0p touches on the
// simulation side (`ssbo`, `GetFluidVBO()`) and the dead `riverMode` switch exist as INERT members
// (0 / false) so that those lines compile unchanged; there is no GL buffer behind them: renderers
// take the device pointer from DeviceParticles() or a packed buffer from PackRenderBuffer()
// instead.  Error convention as the reference: methods return void and log
// (Debug::FatalError only logs, Debug.cpp:54); LastError() exposes the message.
//
// If the host project has MATH::Vec3 / Vec4 (its "MathLibrary"), define
// SPH_HIP_HAVE_MATHLIB before including this header; otherwise minimal PODs are provided.
#pragma once
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "sph_abi.h"

#ifndef SPH_HIP_HAVE_MATHLIB
namespace MATH {
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct Vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    Vec4() = default;
    Vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};
}  // namespace MATH
#endif

struct SPHParticle {              // SPHFluid3D.h:12-24, layout-identical to SphParticle
    MATH::Vec4 pos, vel, acc;
    float density, pressure, padA, padB;
    int isGhost, isActive, padC, pad0;
};
static_assert(sizeof(SPHParticle) == sizeof(SphParticle), "SPHParticle must stay 80 bytes");

class SPHFluidGPU {
public:
    explicit SPHFluidGPU(size_t numParticles_, uint32_t seed_ = 1, void* hipStream = nullptr)
        : numParticles(numParticles_), seed(seed_), stream(hipStream) {
        SphParams p;
        sph_params_default(&p);
        FromParams(p);
        Create();
    }
    ~SPHFluidGPU() { sph_destroy(engine); }
    SPHFluidGPU(const SPHFluidGPU&) = delete;
    SPHFluidGPU& operator=(const SPHFluidGPU&) = delete;

    // ---- methods Scene0p calls (Scene0p.cpp:83,1488,3739,1457,3623,1094,1467,2575,655) ----
    void DispatchCompute(float overrideDt = -1.0f) {                    // SPHFluid3D.cpp:431
        SphParams p = ToParams();                                       // members are re-read every dispatch (:458-506)
        if (Check(sph_set_params(engine, &p), "sph_set_params")) return;
        SphFountain f{fountainMode ? 1 : 0, {fountainOffset.x, fountainOffset.y, fountainOffset.z}, fountainRadius, fountainSpread,
                      fountainJetSpeedLive, fountainDrainLevel, fountainDrainPerSec, fountainSeed};
        if (Check(sph_set_fountain(engine, &f), "sph_set_fountain")) return;
        if (PushRiver()) return;                                        // river members, step 5 (:511-516)
        if (Check(sph_dispatch(engine, overrideDt), "sph_dispatch")) return;
        if (fountainMode && !riverMode && !param_pause) ++fountainSeed; // glUniform1ui("uSeed", fountainSeed++), :541 (fountain step only `!riverMode`, :519)
        RefreshGrid();
    }
    void SimulateSubstep(float overrideDt = -1.0f) { DispatchCompute(overrideDt); }   // BASELINE.json's name
    void ResetSimulation() {                                            // SPHFluid3D.cpp:713
        SphParams p = ToParams();
        if (Check(sph_set_params(engine, &p), "sph_set_params")) return;
        if (PushRiver()) return;                                        // riverMode && !terrainHeights.empty() selects the spawn branch (:104)
        if (Check(sph_reset(engine, numParticles, seed), "sph_reset")) return;
        AfterSpawn();
        std::printf("Reset: particles=%zu fluids=%zu grid=%dx%dx%d cells=%d\n", particles.size(), numFluids, gridSizeX, gridSizeY, gridSizeZ, numCells);
    }
    void ApplyWaveImpulse(float amplitude, float wavelength, float phase, const MATH::Vec3& dir,
                          float yMin = -FLT_MAX, float yMax = FLT_MAX) {   // SPHFluid3D.cpp:604
        const float d[3] = {dir.x, dir.y, dir.z};
        Check(sph_apply_wave_impulse(engine, amplitude, wavelength, phase, d, yMin, yMax), "sph_apply_wave_impulse");
    }
    void ApplyVortexImpulse(float tangentKick, float inwardKick) {       // SPHFluid3D.cpp:627
        SphParams p = ToParams();
        if (Check(sph_set_params(engine, &p), "sph_set_params")) return;
        Check(sph_apply_vortex_impulse(engine, tangentKick, inwardKick), "sph_apply_vortex_impulse");
    }
    void ApplyAttractorImpulse(const MATH::Vec3& point, float pullKick, float radius) {   // SPHFluid3D.cpp:650
        const float q[3] = {point.x, point.y, point.z};
        Check(sph_apply_attractor_impulse(engine, q, pullKick, radius), "sph_apply_attractor_impulse");
    }
    void ApplyCurlFlow(float kick, float scale, float time) {            // SPHFluid3D.cpp:668
        Check(sph_apply_curl_flow(engine, kick, scale, time), "sph_apply_curl_flow");
    }
    void SetStencilTargets(const std::vector<MATH::Vec4>& points) {      // SPHFluid3D.cpp:684
        stencilCount = int(points.size());
        Check(sph_set_stencil_targets(engine, points.empty() ? nullptr : &points[0].x, points.size()), "sph_set_stencil_targets");
    }
    void ApplyStencilAttract(float pullKick, float dampKick) {           // SPHFluid3D.cpp:695
        Check(sph_apply_stencil_attract(engine, pullKick, dampKick), "sph_apply_stencil_attract");
    }
    void GenerateRiverTerrain(int seed_) {                               // SPHFluid3D.cpp:772-878 (heightfield upload = sph_set_river)
        SphParams p = ToParams();
        SphRiver r = ToRiver();
        terrainHeights.assign(size_t(terrainW) * size_t(terrainH), 0.0f);
        if (Check(sph_generate_river_terrain(&p, seed_, &r, terrainHeights.data()), "sph_generate_river_terrain")) return;
        param_gravityY = p.param_gravityY; param_gravityZ = p.param_gravityZ;   // :864-865
        terrainWorldMinX = r.terrainWorldMinX; terrainWorldMinZ = r.terrainWorldMinZ;
        terrainWorldSizeX = r.terrainWorldSizeX; terrainWorldSizeZ = r.terrainWorldSizeZ;
        riverEmitterPos = MATH::Vec3(r.riverEmitterPos[0], r.riverEmitterPos[1], r.riverEmitterPos[2]);
        riverEmitterVel = MATH::Vec3(r.riverEmitterVel[0], r.riverEmitterVel[1], r.riverEmitterVel[2]);
        riverEmitterRadius = r.riverEmitterRadius; riverSinkY = r.riverSinkY; riverSinkZMax = r.riverSinkZMax;
        riverAmp = r.riverAmp; riverFreq = r.riverFreq; riverPhase = r.riverPhase;
        riverChannelWidth = r.riverChannelWidth; riverChannelDepth = r.riverChannelDepth; riverSlopeDrop = r.riverSlopeDrop;
        terrainDirty = true;
        std::printf("[River] seed=%d amp=%g freq=%g width=%g slope=%g\n", seed_, riverAmp, riverFreq, riverChannelWidth, riverSlopeDrop);
    }
    int stencilCount = 0;                                                // SPHFluid3D.h:55
    MATH::Vec3 EffectiveHalf() const {                                  // SPHFluid3D.h:127
        SphParams p = ToParams();
        float h[3];
        sph_effective_half(&p, h);
        return MATH::Vec3(h[0], h[1], h[2]);
    }
    void ComputeGridExtents() {                                         // SPHFluid3D.cpp:354
        SphParams p = ToParams();
        SphGridInfo g;
        sph_compute_grid_extents(&p, &g);
        SetGrid(g);
    }
    size_t GetNumFluids() const { return numFluids; }                   // SPHFluid3D.cpp:601
    // SPHFluid3D.h:37; Scene0p.cpp:85,1459,3625 only store the id.  Inert: no GL object exists (0 = "no buffer").
    unsigned int GetFluidVBO() const { return 0u; }

    // ---- what replaces the GL buffer ids -------------------------------------------------
    const SPHParticle* DeviceParticles() {     // device pointer of the 80-byte array in original order (the `ssbo`)
        const SphParticle* p = nullptr;
        Check(sph_device_particles(engine, &p), "sph_device_particles");
        return reinterpret_cast<const SPHParticle*>(p);
    }
    // packed (x, y, z, w) per particle in original order into a device buffer of the renderer (w: 0 one, 1 density,
    // 2 foam, 3 speed, 4 dye): the render-side replacement of binding 0 reads (fluidDepth.vert, particleImpostor.vert)
    void PackRenderBuffer(float* devOut4, int wMode = 0) { Check(sph_pack_render_buffer(engine, devOut4, sph_num_particles(engine), wMode), "sph_pack_render_buffer"); }
    bool Download(std::vector<SPHParticle>& out) {
        out.resize(sph_num_particles(engine));
        return !Check(sph_download_particles(engine, reinterpret_cast<SphParticle*>(out.data()), out.size()), "sph_download_particles");
    }
    void Sync() { Check(sph_sync(engine), "sph_sync"); }
    const std::string& LastError() const { return lastError; }
    SphEngine* Handle() { return engine; }

    // ---- public data members, names and defaults of SPHFluid3D.h:62-124 ------------------
    float box = 7.0f;
    float cellSize = 0.0f;
    int gridSizeX = 1, gridSizeY = 1, gridSizeZ = 1;
    int numCells = 1;
    MATH::Vec3 gridMinV = MATH::Vec3(-7, -7, -7);
    std::vector<SPHParticle> particles;        // initial state only, never refreshed (as in the reference)
    size_t numParticles;
    size_t numFluids = 0;

    float param_h = 0.28f;
    float param_mass = 13.8f;
    float param_restDensity = 1000.0f;
    float param_gasConstant = 2000.0f;
    float param_viscosity = 3.5f;
    float param_gravityY = -980.0f;
    float param_gravityX = 0.0f;
    float param_gravityZ = 0.0f;
    float param_surfaceTension = 0.0728f;
    float param_timeStep = 0.001f;
    bool param_pause = false;
    bool param_useJitter = true;
    float param_jitterAmp = 0.20f;
    float param_foamGen = 1.0f;
    float param_foamVelRef = 8.0f;
    MATH::Vec3 param_boxCenter = MATH::Vec3(0, 0, 0);
    MATH::Vec3 param_boxHalf = MATH::Vec3(7, 7, 7);
    MATH::Vec3 param_boxEulerDeg = MATH::Vec3(0, 0, 0);
    int param_shapeType = 0;
    MATH::Vec3 param_shapeAux = MATH::Vec3(5.0f, 0.35f, 2.5f);
    int param_mixPattern = 0;
    int param_dyePattern = 0;
    float param_wallRestitution = 0.15f;
    float param_wallFriction = 0.02f;
    // inert counterpart of a member Scene0p touches (SPHFluid3D.h:72 `ssbo`: bound as binding 0 by the GL renderers,
    // Scene0p.cpp:1625,2627,3065,3142 -- 0 binds nothing)
    unsigned int ssbo = 0;
    // river / stream mode, SPHFluid3D.h:171-196 (step 5 of DispatchCompute, :511-516; Scene0p only ever writes
    // riverMode = false, Scene0p.cpp:1660).  After editing terrainHeights by hand set terrainDirty.
    bool riverMode = false;
    std::vector<float> terrainHeights;
    int terrainW = 64, terrainH = 64;
    float terrainWorldMinX = -7.0f, terrainWorldMinZ = -10.0f, terrainWorldSizeX = 14.0f, terrainWorldSizeZ = 20.0f;
    MATH::Vec3 riverEmitterPos = MATH::Vec3(0.0f, 3.0f, -9.0f);
    MATH::Vec3 riverEmitterVel = MATH::Vec3(0.0f, -0.5f, 4.0f);
    float riverEmitterRadius = 1.5f, riverSinkY = -8.5f, riverSinkZMax = 9.0f;
    float riverAmp = 2.0f, riverFreq = 0.25f, riverPhase = 0.0f, riverChannelWidth = 3.0f, riverChannelDepth = 3.5f, riverSlopeDrop = 0.3f;
    bool terrainDirty = false;                 // engine extension: terrainHeights changed since the last upload
    // fountain members, SPHFluid3D.h:161-168 (step 6 of DispatchCompute, :519)
    bool fountainMode = false;
    MATH::Vec3 fountainOffset = MATH::Vec3(0.0f, -5.0f, 0.0f);
    float fountainRadius = 1.0f;
    float fountainSpread = 0.25f;
    float fountainJetSpeedLive = 25.0f;
    float fountainDrainLevel = 1.0f;
    float fountainDrainPerSec = 2.0f;
    unsigned fountainSeed = 0;
    int grid_cap = 160;                        // engine extension (SPHFluid3D.cpp:370 hard-codes 160)
    uint32_t seed;                             // engine extension (the reference seeds from time(nullptr), :99)

private:
    SphEngine* engine = nullptr;
    void* stream = nullptr;
    std::string lastError;

    SphRiver ToRiver() const {
        return SphRiver{riverMode ? 1 : 0, terrainW, terrainH, terrainWorldMinX, terrainWorldMinZ, terrainWorldSizeX, terrainWorldSizeZ,
                        {riverEmitterPos.x, riverEmitterPos.y, riverEmitterPos.z}, {riverEmitterVel.x, riverEmitterVel.y, riverEmitterVel.z},
                        riverEmitterRadius, riverSinkY, riverSinkZMax, riverAmp, riverFreq, riverPhase, riverChannelWidth, riverChannelDepth,
                        riverSlopeDrop};
    }
    bool PushRiver() {                         // true on error
        const SphRiver r = ToRiver();
        const bool send = terrainDirty && terrainHeights.size() == size_t(terrainW) * size_t(terrainH);
        if (Check(sph_set_river(engine, &r, send ? terrainHeights.data() : nullptr), "sph_set_river")) return true;
        if (send) terrainDirty = false;
        return false;
    }
    SphParams ToParams() const {
        SphParams p;
        sph_params_default(&p);
        p.param_h = param_h; p.param_mass = param_mass; p.param_restDensity = param_restDensity;
        p.param_gasConstant = param_gasConstant; p.param_viscosity = param_viscosity;
        p.param_gravityY = param_gravityY; p.param_gravityX = param_gravityX; p.param_gravityZ = param_gravityZ;
        p.param_surfaceTension = param_surfaceTension; p.param_timeStep = param_timeStep;
        p.param_pause = param_pause ? 1 : 0; p.param_useJitter = param_useJitter ? 1 : 0; p.param_jitterAmp = param_jitterAmp;
        p.param_foamGen = param_foamGen; p.param_foamVelRef = param_foamVelRef;
        const MATH::Vec3* v[4] = {&param_boxCenter, &param_boxHalf, &param_boxEulerDeg, &param_shapeAux};
        float* d[4] = {p.param_boxCenter, p.param_boxHalf, p.param_boxEulerDeg, p.param_shapeAux};
        for (int i = 0; i < 4; ++i) { d[i][0] = v[i]->x; d[i][1] = v[i]->y; d[i][2] = v[i]->z; }
        p.param_shapeType = param_shapeType; p.param_mixPattern = param_mixPattern; p.param_dyePattern = param_dyePattern;
        p.param_wallRestitution = param_wallRestitution; p.param_wallFriction = param_wallFriction;
        p.grid_cap = grid_cap;
        return p;
    }
    void FromParams(const SphParams& p) {
        param_h = p.param_h; param_mass = p.param_mass; param_restDensity = p.param_restDensity;
        param_gasConstant = p.param_gasConstant; param_viscosity = p.param_viscosity;
        param_gravityY = p.param_gravityY; param_gravityX = p.param_gravityX; param_gravityZ = p.param_gravityZ;
        param_surfaceTension = p.param_surfaceTension; param_timeStep = p.param_timeStep;
        param_pause = p.param_pause != 0; param_useJitter = p.param_useJitter != 0; param_jitterAmp = p.param_jitterAmp;
        param_foamGen = p.param_foamGen; param_foamVelRef = p.param_foamVelRef;
        param_boxCenter = MATH::Vec3(p.param_boxCenter[0], p.param_boxCenter[1], p.param_boxCenter[2]);
        param_boxHalf = MATH::Vec3(p.param_boxHalf[0], p.param_boxHalf[1], p.param_boxHalf[2]);
        param_boxEulerDeg = MATH::Vec3(p.param_boxEulerDeg[0], p.param_boxEulerDeg[1], p.param_boxEulerDeg[2]);
        param_shapeType = p.param_shapeType;
        param_shapeAux = MATH::Vec3(p.param_shapeAux[0], p.param_shapeAux[1], p.param_shapeAux[2]);
        param_mixPattern = p.param_mixPattern; param_dyePattern = p.param_dyePattern;
        param_wallRestitution = p.param_wallRestitution; param_wallFriction = p.param_wallFriction;
        grid_cap = p.grid_cap;
    }
    void SetGrid(const SphGridInfo& g) {
        gridSizeX = g.dims[0]; gridSizeY = g.dims[1]; gridSizeZ = g.dims[2]; numCells = g.numCells;
        gridMinV = MATH::Vec3(g.gridMin[0], g.gridMin[1], g.gridMin[2]); cellSize = g.cellSize;
    }
    void RefreshGrid() {
        SphGridInfo g;
        if (!Check(sph_grid_info(engine, &g), "sph_grid_info")) SetGrid(g);
    }
    void AfterSpawn() {
        SphParams p;
        if (!Check(sph_get_params(engine, &p), "sph_get_params")) param_mass = p.param_mass;   // SPHFluid3D.cpp:92
        particles.resize(sph_num_particles(engine));
        Check(sph_initial_particles(engine, reinterpret_cast<SphParticle*>(particles.data()), particles.size()), "sph_initial_particles");
        numFluids = 0;
        for (const auto& q : particles) if (q.isGhost == 0) ++numFluids;                       // SPHFluid3D.cpp:338-339
        box = param_boxHalf.x > param_boxHalf.y ? (param_boxHalf.x > param_boxHalf.z ? param_boxHalf.x : param_boxHalf.z)
                                                : (param_boxHalf.y > param_boxHalf.z ? param_boxHalf.y : param_boxHalf.z);
        RefreshGrid();
        std::printf("Fluid particles: %zu\n", particles.size());
    }
    void Create() {
        SphParams p = ToParams();
        if (Check(sph_create(&engine, numParticles, &p, seed, stream), "sph_create")) return;
        AfterSpawn();
    }
    bool Check(int rc, const char* what) {     // logs like Debug::FatalError (which only logs, Debug.cpp:54)
        if (rc == SPH_OK) return false;
        lastError = std::string(what) + ": " + sph_last_error();
        std::fprintf(stderr, "SPHFluidGPU(HIP) %s\n", lastError.c_str());
        return true;
    }
};
