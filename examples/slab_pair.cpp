// slab_pair.cpp -- two z-slab engines driven through the C-ABI only (include/sph_abi.h), the way a C++ host such as
// Scene0p's owner would drive one rank per GPU, on the boundary-first schedule: sph_slab_step_begin (DispatchCompute with the
// slots next to the faces first + the pack of the next exchange on a second stream) and sph_slab_step_finish_local (copy of the
// neighbour's send face, unpack), record counts staying on the device, no host synchronisation per substep.  On ONE GPU the two
// "ranks" live in this process; with one process per GPU sph_slab_step_finish(engine, comm) puts RCCL in place of the copy.
// The result is compared with a single engine over the whole domain: bit for bit.
//
//   g++ -std=c++17 -I include examples/slab_pair.cpp -L <pkg dir> -lsph_hip -o slab_pair
#include <cstdio>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <vector>

#include "sph_abi.h"

#define CHECK(x) do { int _rc = (x); if (_rc) { std::printf("%s -> %d: %s\n", #x, _rc, sph_last_error()); return 1; } } while (0)

struct SlabOut { float pos[3], vel[3], acc[3], rho, prs, foam; uint32_t id, flags, pad[2]; };
static_assert(sizeof(SlabOut) == SPH_SLAB_OUT_BYTES, "64-byte download record");

int main(int argc, char** argv) {
    const size_t nReq = argc > 1 ? (size_t)std::atol(argv[1]) : 60000;
    const int steps = argc > 2 ? std::atoi(argv[2]) : 24;
    SphParams p;
    CHECK(sph_params_default(&p));
    std::vector<SphParticle> init(nReq);
    size_t n = 0;
    float mass = 0.f;
    CHECK(sph_spawn_particles(&p, nReq, 11u, init.data(), &n, &mass));
    init.resize(n);
    p.param_mass = mass;
    for (size_t i = 0; i < n; ++i) init[i].vel[2] = (float)((int)(i % 7) - 3) * 15.0f;   // traffic across the slab boundary
    SphGridInfo g;
    CHECK(sph_compute_grid_extents(&p, &g));
    const int gz = g.dims[2], zmid = gz / 2;
    // owner by cell layer (BuildGrid.comp:25-26)
    std::vector<SphParticle> part[2];
    std::vector<uint32_t> ids[2];
    for (size_t i = 0; i < n; ++i) {
        float q = std::floor((init[i].pos[2] - g.gridMin[2]) / g.cellSize);
        q = std::fmin(std::fmax(q, 0.0f), (float)(gz - 1));
        const int r = (int)q >= zmid ? 1 : 0;
        part[r].push_back(init[i]); ids[r].push_back((uint32_t)i);
    }
    const uint32_t faceCap = (uint32_t)(n / gz * 6 + 4096);
    SphEngine* slab[2] = {nullptr, nullptr};
    CHECK(sph_create_slab(&slab[0], part[0].data(), ids[0].data(), part[0].size(), &p, 0, zmid, 0, 1, part[0].size() + 6 * faceCap, nullptr));
    CHECK(sph_create_slab(&slab[1], part[1].data(), ids[1].data(), part[1].size(), &p, zmid, gz, 1, 0, part[1].size() + 6 * faceCap, nullptr));
    for (auto* e : slab) CHECK(sph_slab_alloc_faces(e, faceCap));
    SphEngine* one = nullptr;
    CHECK(sph_create_from_particles(&one, init.data(), n, &p, nullptr));
    void *sendHi0 = nullptr, *sendLo1 = nullptr;
    CHECK(sph_slab_face_buffer(slab[0], 1, &sendHi0));
    CHECK(sph_slab_face_buffer(slab[1], 0, &sendLo1));
    const float dir[3] = {0.f, 1.f, 0.f};
    // Prime: the halo records of the FIRST substep, by a plain exchange.  The two engines run on their own streams: the hand-off
    // needs each pack finished before the other side reads it.
    for (auto* e : slab) CHECK(sph_slab_pack_async(e));
    for (auto* e : slab) CHECK(sph_sync(e));
    CHECK(sph_slab_unpack_async(slab[0], nullptr, sendLo1, faceCap));
    CHECK(sph_slab_unpack_async(slab[1], sendHi0, nullptr, faceCap));
    for (auto* e : slab) CHECK(sph_sync(e));
    for (int s = 0; s < steps; ++s) {
        if (s % 16 == 0) {                                   // Scene0p.h:144-147 continuous wave
            for (auto* e : slab) CHECK(sph_apply_wave_impulse(e, 1.5f, 3.0f, 0.064f * (float)s, dir, -3.4e38f, 3.4e38f));
            CHECK(sph_apply_wave_impulse(one, 1.5f, 3.0f, 0.064f * (float)s, dir, -3.4e38f, 3.4e38f));
        }
        // Boundary-first substeps: each engine computes the slots next to its faces first, packs on its second stream, copies the
        // neighbour's send face device to device and unpacks -- beside the interior of its SPH pass, no host synchronisation.
        // With one process per GPU the second call is sph_slab_step_finish(engine, comm): RCCL in place of the copy.
        for (auto* e : slab) CHECK(sph_slab_step_begin(e, -1.0f));
        CHECK(sph_slab_step_finish_local(slab[0], nullptr, slab[1]));
        CHECK(sph_slab_step_finish_local(slab[1], slab[0], nullptr));
        CHECK(sph_dispatch(one, -1.0f));
    }
    for (auto* e : slab) CHECK(sph_sync(e));
    std::vector<SphParticle> want(n);
    CHECK(sph_download_particles(one, want.data(), n));
    size_t got = 0, bad = 0, moved = 0;
    for (int r = 0; r < 2; ++r) {
        uint32_t st[5];
        CHECK(sph_slab_status(slab[r], st));
        std::vector<SlabOut> out(part[r].size() + 6 * faceCap);
        size_t m = 0;
        CHECK(sph_slab_download(slab[r], out.data(), out.size(), &m));
        for (size_t i = 0; i < m; ++i) {
            const SphParticle& w = want[out[i].id];
            if (std::memcmp(out[i].pos, w.pos, 12) || std::memcmp(out[i].vel, w.vel, 12) || std::memcmp(&out[i].rho, &w.density, 4) ||
                std::memcmp(&out[i].prs, &w.pressure, 4)) ++bad;
            const float q = std::floor((w.pos[2] - g.gridMin[2]) / g.cellSize);
            if (((int)q >= zmid ? 1 : 0) != (ids[0].size() > out[i].id && std::binary_search(ids[0].begin(), ids[0].end(), out[i].id) ? 0 : 1)) ++moved;
        }
        got += m;
        uint64_t bytes[4];
        CHECK(sph_slab_message_bytes(slab[r], bytes));       // what the last exchange sent to the lower / upper neighbour, against a whole face
        if (st[4] & 16u) {                                    // notice: a particle crossed more cell layers in one substep than the exchange follows
            std::printf("slab %d: notice 16 (a particle crossed more than 3 cell layers in one substep)\n", r);
            CHECK(sph_slab_clear_flags(slab[r], 16u));
        }
        std::printf("slab %d: %zu owned particles, last pack %u / %u records, last messages %llu / %llu bytes (a face: %llu)\n", r, m, st[0], st[1],
                    (unsigned long long)bytes[0], (unsigned long long)bytes[1], (unsigned long long)(bytes[2] ? bytes[2] : bytes[3]));
    }
    std::printf("%zu of %zu particles, %zu differ from the single engine, %zu changed slab\n", got, n, bad, moved);
    // The error path (round 5): an impulse issued on ONE rank only.  That rank plans whole faces for its next exchanges, its neighbour does not --
    // over RCCL two sized ncclSend / ncclRecv that do not match (a hang, or records cut off).  Every sized exchange has a plan, and nothing moves
    // before both ends of a link have compared theirs: sph_slab_step_finish_local looks at the neighbour engine's plan, sph_slab_step_finish
    // sends the 64-byte plans across the link first.  Either way the step ends with SPH_ERR_STATE and the difference by name, on both ranks.
    bool refused = false;
    {
        CHECK(sph_apply_wave_impulse(slab[0], 40.0f, 50.0f, 1.0f, dir, -3.4e38f, 3.4e38f));      // slab 0 only
        for (auto* e : slab) CHECK(sph_slab_step_begin(e, -1.0f));
        const int r0 = sph_slab_step_finish_local(slab[0], nullptr, slab[1]);
        std::printf("one-sided impulse: finish(slab 0) -> %d: %s\n", r0, r0 ? sph_last_error() : "accepted");
        const int r1 = sph_slab_step_finish_local(slab[1], slab[0], nullptr);
        std::printf("one-sided impulse: finish(slab 1) -> %d: %s\n", r1, r1 ? sph_last_error() : "accepted");
        refused = r0 == SPH_ERR_STATE && r1 == SPH_ERR_STATE;
        SphSlabIntent a, b;
        CHECK(sph_slab_plan(slab[0], &a, nullptr));
        CHECK(sph_slab_plan(slab[1], &b, nullptr));
        std::printf("plans of exchange %u: slab 0 has seen %u hold events and would send %u + %u records up, slab 1 has seen %u and expects %u + %u\n",
                    a.exchangeNo, a.holdEvents, a.sendHalo[1], a.sendMig[1], b.holdEvents, b.recvHalo[0], b.recvMig[0]);
        for (auto* e : slab) { uint32_t st[5]; CHECK(sph_slab_status(e, st)); if (st[4] & ~16u) refused = false; }   // nothing was cut off: nothing moved
    }
    for (auto* e : slab) sph_destroy(e);
    sph_destroy(one);
    const bool ok = got == n && bad == 0 && refused;
    std::printf(ok ? "slab_pair OK\n" : "slab_pair FAILED\n");
    return ok ? 0 : 1;
}
