// headless_scene.cpp -- the call pattern of Scene0p against the HIP engine, with no GL/SDL/ImGui:
// ctor (Scene0p.cpp:83), per frame: live parameter edits, optional ApplyWaveImpulse
// (Scene0p.cpp:1464-1468), fixed-dt substep loop capped at 16 per frame (:1482-1494), and a
// reel-export style burst of deterministic substeps (:3720-3739).
//
//   g++ -std=c++17 -I include examples/headless_scene.cpp -L <pkg dir> -lsph_hip -o headless_scene
#include <chrono>
#include <cmath>
#include <cstdio>

#include "SPHFluidGPU_hip.hpp"

using namespace MATH;

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)std::atol(argv[1]) : 50000;   // Scene0p.cpp:83 default
    SPHFluidGPU* fluidGPU = new SPHFluidGPU(n, /*seed=*/7);
    if (!fluidGPU->LastError().empty()) { delete fluidGPU; return 2; }
    std::printf("particles=%zu grid=%dx%dx%d mass=%g\n", fluidGPU->GetNumFluids(), fluidGPU->gridSizeX, fluidGPU->gridSizeY,
                fluidGPU->gridSizeZ, fluidGPU->param_mass);
    const float fixedDt = fluidGPU->param_timeStep;
    float wavePhase = 0.0f;
    int substeps = 0;
    for (int frame = 0; frame < 20; ++frame) {
        if (frame == 5) fluidGPU->param_viscosity = 5.0f;              // an ImGui slider edit (Scene0p.cpp:962-969)
        if (frame == 10) fluidGPU->param_boxEulerDeg = Vec3(0.0f, 15.0f, 0.0f);
        fluidGPU->ApplyWaveImpulse(1.5f, 3.0f, wavePhase, Vec3(0, 1, 0));   // continuousWave, Scene0p.h:144-147
        wavePhase += 4.0f * (1.0f / 60.0f);
        float dtAccumulator = 1.0f / 60.0f;
        int didSteps = 0;
        while (dtAccumulator >= fixedDt && didSteps < 16) {            // maxSubstepsPerFrame, Scene0p.h:48
            fluidGPU->DispatchCompute(fixedDt);
            dtAccumulator -= fixedDt;
            ++didSteps;
        }
        substeps += didSteps;
    }
    std::vector<SPHParticle> host;
    if (!fluidGPU->Download(host)) { delete fluidGPU; return 3; }
    double ymin = 1e30, ymax = -1e30, rho = 0;
    for (const auto& p : host) { ymin = std::fmin(ymin, p.pos.y); ymax = std::fmax(ymax, p.pos.y); rho += p.density; }
    std::printf("after %d substeps: y in [%.3f, %.3f], mean density %.1f\n", substeps, ymin, ymax, rho / host.size());
    // reel-export burst: nSub = ceil(frameDt / dt) back-to-back substeps, timed
    const int nSub = 34;
    fluidGPU->Sync();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nSub; ++i) fluidGPU->DispatchCompute(fixedDt);
    fluidGPU->Sync();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%d substeps in %.3f ms (%.1f substeps/s)\n", nSub, ms, nSub / ms * 1e3);
    fluidGPU->numParticles = 20000;
    fluidGPU->ResetSimulation();
    bool ok = fluidGPU->GetNumFluids() == 20000 && fluidGPU->LastError().empty() && std::isfinite(rho);
    // river / stream mode (SPHFluid3D.h:171-206; dead code in Scene0p, driven here the way its members are meant to be)
    fluidGPU->param_boxEulerDeg = Vec3(0.0f, 0.0f, 0.0f);
    fluidGPU->GenerateRiverTerrain(3);
    fluidGPU->riverMode = true;
    fluidGPU->ResetSimulation();
    for (int i = 0; i < 40; ++i) fluidGPU->DispatchCompute(fixedDt);
    if (!fluidGPU->Download(host)) { delete fluidGPU; return 3; }
    size_t inChannel = 0;
    for (const auto& p : host) {
        const float cx = fluidGPU->param_boxCenter.x + fluidGPU->riverAmp * std::sin(fluidGPU->riverFreq * p.pos.z + fluidGPU->riverPhase);
        if (std::fabs(p.pos.x - cx) <= fluidGPU->riverChannelWidth * 1.001f && std::isfinite(p.pos.y)) ++inChannel;
    }
    std::printf("river mode: %zu of %zu particles inside the channel after 40 substeps\n", inChannel, host.size());
    ok = ok && inChannel == host.size() && fluidGPU->LastError().empty() && fluidGPU->param_gravityY == -120.0f;
    delete fluidGPU;
    std::printf(ok ? "headless_scene OK\n" : "headless_scene FAILED\n");
    return ok ? 0 : 1;
}
