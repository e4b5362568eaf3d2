/*
 * sph_abi.h -- C-ABI of the MI355X-native SPH substep engine (libsph_hip.so).
 *
 * The reference has no plugin/FFI layer: its boundary is the concrete C++ class
 * SPHFluidGPU (ComponentFramework/SPHFluid3D.h:26-210) that Scene0p holds by raw
 * pointer (Scene0p.h:396).  Each entry point below names the reference member it
 * replaces.  Plain pointers and sizes only; no torch / STL types cross this line.
 * The header-only C++ shim include/SPHFluidGPU_hip.hpp re-creates the class surface
 * (same member names) on top of these calls; INTEGRATION.md shows the swap.
 *
 * Conventions
 *  - every function returns 0 on success, a negative SPH_ERR_* otherwise, and
 *    leaves a message for sph_last_error() (thread-local);
 *  - calls enqueue work on the engine's HIP stream and return without waiting;
 *    sph_download_particles() and sph_sync() synchronise;
 *  - one host thread per engine (as the reference: everything runs on the GL thread,
 *    SceneManager.cpp:69-93);
 *  - there is NO CPU fallback: without a HIP device every compute call fails with
 *    SPH_ERR_HIP.
 */
#ifndef SPH_ABI_H
#define SPH_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_ABI_VERSION 4   /* 4: the PLAN of a sized exchange (SphSlabIntent) compared between neighbours before any record moves: sph_slab_step_finish_local compares the
                               neighbour engines' plans, the RCCL transport sends them across each link first (sph_slab_set_verify), waits with a deadline (SPH_ERR_TIMEOUT,
                               sph_slab_set_deadline, sph_sync_deadline); flag 32; sph_slab_plan / _plans_agree, sph_comm_selftest_faces; SPH_OPT_NEIGHBOR_KERNEL 4 retired */
/* (3: compact halo faces (40-byte halo copies, count-sized messages), jumps of up to 3 cell layers followed, sph_slab_clear_flags / _message_bytes / _step_times / _face_bytes, flag 16 no longer an error, SPH_OPT_NEIGHBOR_KERNEL 4) */
/* (2: sph_slab_step_*, header validation of received halo messages, SPH_OPT_NEIGHBOR_KERNEL 3 (default), records on demand by default) */

enum {
    SPH_OK = 0,
    SPH_ERR_ARG = -1,      /* bad argument / null handle                    */
    SPH_ERR_HIP = -2,      /* HIP runtime error (message has hipGetErrorString) */
    SPH_ERR_STATE = -3,    /* call not valid in the engine's current state   */
    SPH_ERR_CAPACITY = -4, /* a fixed-capacity buffer (ghosts, migrants) overflowed */
    SPH_ERR_TIMEOUT = -5   /* a wait for a neighbour rank ran into its deadline (sph_slab_set_deadline): device work of this rank is still queued behind a
                              transfer that will never complete; print sph_last_error() and end the process */
};

/* 80-byte particle record: struct SPHParticle, SPHFluid3D.h:12-24 (std430 twin:
 * shaders/SPHFluid.comp:5-17).  Index i is the same particle forever (renderers
 * index by gl_InstanceID, Scene0p.cpp:1625-1637): the engine never permutes this
 * array, whatever it sorts internally. */
typedef struct SphParticle {
    float pos[4];
    float vel[4];
    float acc[4];
    float density;
    float pressure;
    float padA;      /* foam factor, SPHFluid.comp:209-217          */
    float padB;      /* dye, SPHFluid3D.cpp:316-329                 */
    int32_t isGhost;
    int32_t isActive;
    int32_t padC;    /* colour group, SPHFluid3D.cpp:307-311        */
    int32_t pad0;
} SphParticle;

/* The public param_* members of SPHFluidGPU, SPHFluid3D.h:94-124, field for field.
 * Sampled at every sph_dispatch(), like the per-dispatch uniform uploads at
 * SPHFluid3D.cpp:458-506, so edits take effect on the next substep. */
typedef struct SphParams {
    float param_h;                 /* :94  */
    float param_mass;              /* :95  (overwritten by spawn: rho0*(0.85h)^3, SPHFluid3D.cpp:92) */
    float param_restDensity;       /* :96  */
    float param_gasConstant;       /* :97  */
    float param_viscosity;         /* :98  */
    float param_gravityY;          /* :99  */
    float param_gravityX;          /* :100 */
    float param_gravityZ;          /* :101 */
    float param_surfaceTension;    /* :102 */
    float param_timeStep;          /* :103 */
    int32_t param_pause;           /* :104 (bool) */
    int32_t param_useJitter;       /* :106 (bool) */
    float param_jitterAmp;         /* :107 */
    float param_foamGen;           /* :109 */
    float param_foamVelRef;        /* :110 */
    float param_boxCenter[3];      /* :112 */
    float param_boxHalf[3];        /* :113 */
    float param_boxEulerDeg[3];    /* :116 */
    int32_t param_shapeType;       /* :117 */
    float param_shapeAux[3];       /* :119 */
    int32_t param_mixPattern;      /* :121 */
    int32_t param_dyePattern;      /* :122 */
    float param_wallRestitution;   /* :123 */
    float param_wallFriction;      /* :124 */
    /* engine extension, no reference member: per-axis cell-count cap of
     * ComputeGridExtents (hard-coded 160 at SPHFluid3D.cpp:370). */
    int32_t grid_cap;
} SphParams;

typedef struct SphGridInfo {       /* gridSizeX/Y/Z, numCells, gridMinV, cellSize: SPHFluid3D.h:63-66 */
    int32_t dims[3];
    int32_t numCells;
    float gridMin[3];
    float cellSize;
} SphGridInfo;

typedef struct SphFountain {       /* public fountain* members of SPHFluidGPU, SPHFluid3D.h:161-168 (same names) */
    int32_t fountainMode;          /* bool fountainMode = false */
    float fountainOffset[3];       /* nozzle, container-relative (0,-5,0) */
    float fountainRadius;          /* 1.0 */
    float fountainSpread;          /* 0.25 */
    float fountainJetSpeedLive;    /* 25.0, written per frame by the scene */
    float fountainDrainLevel;      /* 1.0 */
    float fountainDrainPerSec;     /* 2.0 */
    uint32_t fountainSeed;         /* advances by one per dispatch (SPHFluid3D.cpp:541) */
} SphFountain;

typedef struct SphRiver {          /* public river / terrain members of SPHFluidGPU, SPHFluid3D.h:171-196 (same names, same initialisers) */
    int32_t riverMode;             /* bool riverMode = false */
    int32_t terrainW, terrainH;    /* 64, 64: heightfield samples; the heights themselves travel beside this struct */
    float terrainWorldMinX, terrainWorldMinZ, terrainWorldSizeX, terrainWorldSizeZ;   /* -7, -10, 14, 20 */
    float riverEmitterPos[3];      /* (0, 3, -9) */
    float riverEmitterVel[3];      /* (0, -0.5, 4) */
    float riverEmitterRadius;      /* 1.5 */
    float riverSinkY, riverSinkZMax;   /* -8.5, 9 */
    float riverAmp, riverFreq, riverPhase, riverChannelWidth, riverChannelDepth, riverSlopeDrop;   /* 2, 0.25, 0, 3, 3.5, 0.3 */
} SphRiver;

typedef struct SphEngine SphEngine; /* opaque; owns every device buffer (as SPHFluidGPU owns its GL buffers, SPHFluid3D.cpp:61-83) */

/* ---- engine options (sph_set_option) ------------------------------------------- */
enum {
    SPH_OPT_NEIGHBOR_KERNEL = 1, /* SPH pass: 3 = k_sph_walk (default: one target per lane, LDS-staged candidate rows, neighbour lists walked per lane over 32-byte records), 2 = k_sph_list (round 2's form of the same plan), 1 = k_sph_slow (one target per thread, plain sweeps over global memory); same bits. 0 (round 1's tile pass) and 4 (round 4's k_sph_tile: measured slower everywhere, profiles/r04_tile_pass_experiment.txt) were retired and are refused */
    SPH_OPT_GRID_BUILD = 2,      /* 0 = counting sort (default), 1 = atomicExch linked list as BuildGrid.comp (A/B only; neighbour order then arbitrary) */
    SPH_OPT_AOS_MODE = 3,        /* 1 = lazy (default): the substep keeps its state in the engine's own arrays and the 80-byte records are brought up to date by sph_device_particles() / sph_download_particles() / sph_pack_render_buffer(), i.e. once per rendered frame instead of once per substep (the scattered 52-byte update of every record costs about 13 % of the SPH pass); 0 = eager: the SPH pass also updates the records, they are current after every dispatch. Same values either way. */
    SPH_OPT_GRAPH = 5,           /* 1 = sph_dispatch_n replays a hipGraph once the same call (same members, options, substep count) has been seen twice; default 0 */
    SPH_OPT_GRAPH_LAUNCHES = 6,  /* read-only: number of graph replays so far */
    SPH_OPT_TIMING = 4,          /* hipEvents around kernels for sph_kernel_times(): 1 = every kernel, 2 = only the SPH pass */
    /* test / tuning hooks */
    SPH_OPT_DEBUG = 100          /* test hooks of k_sph_walk / k_sph_list -- bit 0: treat every neighbour list as overflowed, bit 1: treat every target as
                                    outside the list's slack (sweep-3 fallback), bit 2: treat every window as overflowed (whole wave falls
                                    back), bit 3: count fallbacks / list entries / staged candidates for sph_debug_counters
                                    (bit 8 is used internally by the z-slab face launch) */
};

/* ---- host-only helpers (no device needed) --------------------------------------- */
int sph_abi_version(void);
/* Defaults of SPHFluid3D.h:94-124 (+ grid_cap = 160). */
int sph_params_default(SphParams* out);
/* MakeRotationMat3XYZ, SPHFluid3D.cpp:13-30: column-major world_from_box. */
int sph_rotation_mat3(const float eulerDeg[3], float outM[9]);
/* SPHFluidGPU::EffectiveHalf(), SPHFluid3D.h:127-158. */
int sph_effective_half(const SphParams* params, float outHalf[3]);
/* SPHFluidGPU::ComputeGridExtents(), SPHFluid3D.cpp:354-376. */
int sph_compute_grid_extents(const SphParams* params, SphGridInfo* out);
/* SPHFluidGPU::InitializeParticles() standard-fill branch, SPHFluid3D.cpp:85-102,159-332,
 * with an explicit seed (the reference seeds from time(nullptr), :99).  Writes at most
 * nRequested records, returns the count produced through *nOut and param_mass (:92)
 * through *massOut. */
int sph_spawn_particles(const SphParams* params, size_t nRequested, uint32_t seed,
                        SphParticle* out, size_t* nOut, float* massOut);
const char* sph_last_error(void);

/* ---- lifetime ------------------------------------------------------------------- */
/* SPHFluidGPU::SPHFluidGPU(size_t), SPHFluid3D.cpp:32-59: spawn + allocate + upload.
 * `stream` is a hipStream_t (or NULL for an engine-owned stream). */
int sph_create(SphEngine** out, size_t nRequested, const SphParams* params, uint32_t seed, void* stream);
/* Same, but with caller-provided initial records instead of the spawn (bench / tests). */
int sph_create_from_particles(SphEngine** out, const SphParticle* particles, size_t n,
                              const SphParams* params, void* stream);
/* SPHFluidGPU::~SPHFluidGPU(), SPHFluid3D.cpp:61-83. */
int sph_destroy(SphEngine* e);
/* SPHFluidGPU::ResetSimulation(), SPHFluid3D.cpp:713-731: respawn with numParticles
 * requested (Scene0p.cpp:1403-1408 writes numParticles before the reset). Invalidates
 * the pointer returned by sph_device_particles(). */
int sph_reset(SphEngine* e, size_t nRequested, uint32_t seed);

/* ---- parameters ----------------------------------------------------------------- */
int sph_set_params(SphEngine* e, const SphParams* params);
int sph_get_params(const SphEngine* e, SphParams* out);
int sph_set_option(SphEngine* e, int option, int value);
int sph_get_option(const SphEngine* e, int option, int* value);

/* ---- the hot path ---------------------------------------------------------------- */
/* SPHFluidGPU::DispatchCompute(float overrideDt = -1), SPHFluid3D.cpp:431-522:
 * ClearGrid -> BuildGrid -> SPHFluid -> OBBConstraints. No-op when param_pause. */
int sph_dispatch(SphEngine* e, float overrideDt);
/* n back-to-back substeps (the reel-export loop, Scene0p.cpp:3720-3739). */
int sph_dispatch_n(SphEngine* e, float overrideDt, int nSubsteps);
/* SPHFluidGPU::ApplyWaveImpulse, SPHFluid3D.cpp:604-623 + shaders/WaveImpulse.comp. */
int sph_apply_wave_impulse(SphEngine* e, float amplitude, float wavelength, float phase,
                           const float dir[3], float yMin, float yMax);

/* SPHFluidGPU::ApplyVortexImpulse, SPHFluid3D.cpp:627-646 + shaders/VortexImpulse.comp (kicks pre-multiplied by dt). */
int sph_apply_vortex_impulse(SphEngine* e, float tangentKick, float inwardKick);
/* SPHFluidGPU::ApplyAttractorImpulse, SPHFluid3D.cpp:650-664 + shaders/AttractorImpulse.comp. */
int sph_apply_attractor_impulse(SphEngine* e, const float point[3], float pullKick, float radius);
/* SPHFluidGPU::SetStencilTargets, SPHFluid3D.cpp:684-693: `count` points of 4 floats (w unused). */
int sph_set_stencil_targets(SphEngine* e, const float* points4, size_t count);
/* SPHFluidGPU::ApplyStencilAttract, SPHFluid3D.cpp:695-710 + shaders/StencilAttract.comp (target = points[i % count]). */
int sph_apply_stencil_attract(SphEngine* e, float pullKick, float dampKick);
/* SPHFluidGPU::ApplyCurlFlow, SPHFluid3D.cpp:668-681 + shaders/CurlFlow.comp. */
int sph_apply_curl_flow(SphEngine* e, float kick, float scale, float time);

/* Fountain recycle = DispatchCompute step 6 (SPHFluid3D.cpp:519, DispatchFountainRecycle :526-544,
 * shaders/FountainRecycle.comp): while fountainMode is set every dispatch ends with the recycle
 * pass and advances fountainSeed.  sph_get_fountain returns the current values (seed included). */
void sph_fountain_default(SphFountain* out);             /* SPHFluid3D.h:161-168 initialisers */
int sph_set_fountain(SphEngine* e, const SphFountain* f);
int sph_get_fountain(const SphEngine* e, SphFountain* out);

/* River / stream mode = DispatchCompute step 5 (SPHFluid3D.cpp:511-516: DispatchTerrainConstraints :546-560,
 * DispatchChannelConstraint :562-578, DispatchStreamEmit :580-602 with shaders/TerrainConstraints.comp,
 * ChannelConstraint.comp, StreamEmit.comp): while riverMode is set and a heightfield has been given, every dispatch
 * ends with terrain collision, channel confinement and recycling (one fused kernel: each pass touches only its own
 * particle); the fountain step is then skipped (:519) and sph_reset spawns along the channel (:104-160).  Dead code in
 * the reference's scene (Scene0p.cpp:1660 is the only writer of riverMode), provided for completeness.
 * Single-GPU engines only (recycled particles jump across slabs). */
void sph_river_default(SphRiver* out);                   /* SPHFluid3D.h:171-196 initialisers */
/* SPHFluidGPU::GenerateRiverTerrain(int seed), SPHFluid3D.cpp:772-878, as a pure host function: reads
 * params->param_boxCenter / param_boxHalf and river->terrainW / terrainH; writes every other member of *river,
 * terrainW * terrainH floats into `heights` (terrainHeights) and param_gravityY = -120, param_gravityZ = 0 (:864-865).
 * std::rand() is the Microsoft runtime's LCG (the reference is a Visual Studio project). */
int sph_generate_river_terrain(SphParams* params, int seed, SphRiver* river, float* heights);
/* The river branch of InitializeParticles (:104-160) as a pure host function; writes exactly nRequested records. */
int sph_spawn_river_particles(const SphParams* params, const SphRiver* river, const float* heights, size_t nRequested,
                              uint32_t seed, SphParticle* out, size_t* nOut, float* massOut);
/* Members + heightfield into the engine (the glBufferData of terrainSSBO, :868-873).  heights == NULL keeps the
 * heightfield given before (terrainW / terrainH must then be unchanged). */
int sph_set_river(SphEngine* e, const SphRiver* river, const float* heights);
int sph_get_river(const SphEngine* e, SphRiver* out);

/* ---- data ------------------------------------------------------------------------ */
size_t sph_num_particles(const SphEngine* e);            /* particles.size() / GetNumFluids() */
int sph_grid_info(const SphEngine* e, SphGridInfo* out); /* gridSize*, numCells, gridMinV, cellSize */
/* glBufferData of the particle SSBO, SPHFluid3D.cpp:417-429 (n must equal sph_num_particles). */
int sph_upload_particles(SphEngine* e, const SphParticle* host, size_t n);
/* Read-back of the SSBO (the reference never reads back; needed for parity tests). Synchronises. */
int sph_download_particles(SphEngine* e, SphParticle* host, size_t n);
/* Device pointer of the 80-byte AoS in original order: the `ssbo` renderers bind
 * (Scene0p.cpp:1625,2627,3065,3142). Borrowed, read-only, invalidated by reset/destroy.
 * With SPH_OPT_AOS_MODE 1 (default) this call is what brings the records up to date (one
 * streaming kernel on the engine's stream, no synchronisation): call it once per frame,
 * before the draw, exactly where Scene0p binds the buffer; the contents then stay valid
 * until the next dispatch. */
int sph_device_particles(SphEngine* e, const SphParticle** devPtr);
/* Render-side export: one float4 (x, y, z, w) per particle in original order into a DEVICE buffer the
 * caller owns (e.g. a GL vertex buffer mapped through HIP-GL interop), replacing the renderers' reads of
 * binding 0 (shaders/fluidDepth.vert:16-24, particleImpostor.vert; Scene0p.cpp:1625,2627,3065).
 * wMode: 0 = 1.0, 1 = density, 2 = foam (padA), 3 = speed |vel|, 4 = dye (padB).  Asynchronous on the
 * engine's stream (work the caller has queued on OTHER streams for devOut4, e.g. a fill, is not ordered
 * against it: finish that first, or hand the engine the caller's stream at creation); n must equal
 * sph_num_particles. */
int sph_pack_render_buffer(SphEngine* e, float* devOut4, size_t n, int wMode);
/* Initial host-side records (SPHFluidGPU::particles: initial state only, never refreshed). */
int sph_initial_particles(const SphEngine* e, SphParticle* host, size_t n);
/* Grid as BuildGrid.comp defines it, for tests: cellCount[numCells] and
 * particleCell[n] (binding 3) in the reference's cell indexing. Synchronises. */
int sph_download_grid(SphEngine* e, int32_t* cellCount, size_t nCells, int32_t* particleCell, size_t n);
int sph_sync(SphEngine* e);
/* Diagnostic counters of the list passes k_sph_walk / k_sph_list (SPH_OPT_DEBUG bit 3), summed over launches since the last
 * reset, 8 slots: [0] candidate rows walked from global memory because the wave's window did not fit (k_sph_walk only),
 * [1] targets recomputed by an exact fallback sweep, [2] neighbour-list entries, [3] candidate rows (k_sph_walk only),
 * [4] lanes (one target each), [5] targets whose list overflowed, [6] targets that left the list's slack (sweep-3
 * fallback), [7] waves with at least one fallback target.  Never used on a timed path. */
int sph_debug_counters(SphEngine* e, uint64_t* out, int count, int reset);

/* ---- multi-GPU: z-slab decomposition (no reference counterpart) ------------
 * One engine per rank owns the global cell layers [z0, z1) of ComputeGridExtents' grid plus one
 * read-only ghost layer per side.  Per substep the host calls pack -> (exchange) -> unpack ->
 * sph_dispatch.  Records crossing ranks are 64 bytes: float px,py,pz,vx,vy,vz,rho,prs,foam;
 * uint32 id, flags, pad; float ax,ay,az,pad (acc travels so that a migrant's 80-byte record is complete
 * on its new owner).  Buffers passed to pack/unpack are DEVICE pointers. */
#define SPH_SLAB_REC_BYTES 64
#define SPH_SLAB_OUT_BYTES 64
/* `ids` are global particle ids (they fix the summation order, so results do not depend on the
 * decomposition); `capacity` bounds owned + ghost + migrated-in slots. */
int sph_create_slab(SphEngine** out, const SphParticle* particles, const uint32_t* ids, size_t n,
                    const SphParams* params, int z0, int z1, int hasLo, int hasHi, size_t capacity, void* stream);
/* Classify by current position, emit records for the lower / upper neighbour (migrants + boundary
 * layer copies); countsOut = records written per direction.  Synchronises. */
int sph_slab_pack(SphEngine* e, void* sendLo, void* sendHi, uint32_t capLo, uint32_t capHi, uint32_t countsOut[2]);
/* Append the records received from the lower / upper neighbour. */
int sph_slab_unpack(SphEngine* e, const void* recvLo, uint32_t nLo, const void* recvHi, uint32_t nHi);
/* Owned particles as 64-byte records (pos3, vel3, acc3, rho, P, foam, uint32 id, flags, 2 pad) into host memory. */
int sph_slab_download(SphEngine* e, void* hostOut, size_t capRecords, size_t* nOut);

/* ---- the same exchange without host round trips, and its RCCL transport ------------------------------------------
 * The engine owns four device buffers (send lo / hi, receive lo / hi) of sph_slab_face_bytes(): a 64-byte header (magic, halo
 * copies, migrants, the sender's true counts, exchange number), then faceCap 64-byte records for MIGRANTS (the layout above),
 * then faceCap 40-byte records for HALO COPIES (float px,py,pz,vx,vy,vz,rho,prs; uint32 id, flags: what a neighbour candidate
 * needs -- round 4).  Counts never travel through the host.  Per substep a rank calls sph_slab_exchange
 * (pack -> per z-neighbour two grouped ncclSend / ncclRecv pairs over xGMI: header + migrants in use, halo copies in use ->
 * unpack, all on the engine's stream) and then sph_dispatch.  MESSAGE SIZES: the whole face, unless the face has been calm (its
 * record counts of the last two known exchanges within 3 % of each other, no impulse / container edit / re-priming in the last
 * three exchanges): then the records in use two exchanges ago + a quarter + 1024 (read back asynchronously into pinned memory,
 * so the path never waits for the device; both ends of a link derive the size from the same numbers: the sender from its own
 * counts, the receiver from the headers it received then).  A message that turns out too small sets error flag 8 on the
 * receiver (records were cut off).
 * sph_slab_pack_async / sph_slab_unpack_async are the two halves for hosts that move the faces themselves (several slab
 * engines in one process, another transport): move sph_slab_face_bytes() bytes, or the three parts in use.  Overflows (send
 * face, slot capacity) set a device-side flag that sph_slab_status / sph_slab_download report.  faceCap must be the same on
 * all ranks of a communicator: the first sph_slab_exchange of an engine on a communicator checks that with one ncclAllReduce
 * (and one stream synchronisation) and fails instead of hanging.  A `stream` of NULL at creation means an engine-owned stream:
 * everything above is ordered on THAT stream.  While param_pause is set sph_slab_exchange and the step calls below do nothing
 * (as the paused DispatchCompute, SPHFluid3D.cpp:432): the halo records in place stay valid; all ranks must pause together. */
#define SPH_COMM_ID_BYTES 128
#define SPH_SLAB_HALO_BYTES 40
typedef struct SphComm SphComm;
int sph_slab_alloc_faces(SphEngine* e, uint32_t faceCap);
int sph_slab_face_buffer(SphEngine* e, int which /* 0 send lo, 1 send hi, 2 recv lo, 3 recv hi */, void** devPtr);
int sph_slab_face_bytes(SphEngine* e, uint64_t* bytes);
int sph_slab_pack_async(SphEngine* e);
int sph_slab_unpack_async(SphEngine* e, const void* recvLo, const void* recvHi, uint32_t recvCap);
/* Synchronises; out = {records packed for lo, for hi, slots in use, -, flags}.  Flags (device side, sticky until
 * sph_slab_clear_flags).  ERRORS -- records were lost; this call and sph_slab_download then return an error: 1 a send face
 * overflowed, 2 the slot capacity overflowed while appending received records, 4 a received message did not start with a valid
 * header, 8 the neighbour had more records than its message carried.  NOTICE -- nothing lost, the call succeeds and out[4]
 * carries the bit: 16 a particle crossed MORE cell layers in z within one substep than the exchange follows.  The exchange
 * follows up to 3 layers per substep (the pack and the face launches of a boundary-first step cover the 5 lowest / highest
 * local layers; migrants go to the adjacent rank, which such a jump still reaches while every slab is at least 4 layers
 * thick); beyond that -- a particle placed far outside the container, a container that moved by cells under the fluid, a
 * slab thinner than the jump -- the decomposed run goes on but no longer equals the single-domain run, and says so here.
 * (SPHFluid.comp moves a particle with the uncapped velocity (v + a dt) dt, so this is a property of the scene, not a guarantee:
 * the BASELINE workloads stay far inside it through their whole collapse; the pack after a container change scans
 * every slot, so a change of shape alone is followed exactly as long as no particle has to cross a whole slab.) */
int sph_slab_status(SphEngine* e, uint32_t out[5]);
/* Clears the given flag bits (synchronises): a host acknowledges notice 16 and goes on. */
int sph_slab_clear_flags(SphEngine* e, uint32_t mask);
/* Bytes of the last exchange's messages to the lower / upper neighbour, and the bytes a whole face would be: {sent lo, sent hi,
 * face lo, face hi} (host-side bookkeeping, no synchronisation). */
int sph_slab_message_bytes(SphEngine* e, uint64_t out[4]);
/* Host-only (no device): the sizing rule itself -- records a message carries, given the face's record count two exchanges ago, three
 * exchanges ago, and the face capacity: the capacity unless the two counts are within 3 % + 64 of each other, else count + 25 % + 1024. */
int sph_slab_message_records(uint32_t seen, uint32_t before, uint32_t cap);
/* ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy: rank 0 creates the id and hands its 128 bytes to the other ranks
 * by any means (MPI, a file, torch.distributed); one rank per process, on the current HIP device. */
/* The nccl* entry points are loaded with dlopen ("librccl.so.1"); the environment variable SPH_RCCL_LIBRARY names another library to load them from.  The tests use
 * it to put a stand-in transport in RCCL's place that runs between processes on ONE GPU and refuses a receive whose size differs from its send's
 * (tests/fake_rccl/fake_rccl.c, tests/test_gpu_fake_rccl.py). */
int sph_comm_unique_id(void* out128);
int sph_comm_create(SphComm** out, const void* id128, int rank, int world);
int sph_comm_destroy(SphComm* comm);
/* Health check of the transport on this rank alone: one grouped ncclSend + ncclRecv of `bytes` bytes (a multiple of 4, at
 * most 2^30) from the rank to itself on a non-blocking stream, compared on the host.  Synchronises.  (The only way to
 * execute ncclSend / ncclRecv on a one-GPU box: RCCL refuses two ranks on one device.) */
int sph_comm_selftest(SphComm* comm, uint64_t bytes);
/* The same, also returning the hipEvent time of the grouped send + receive alone (ms). */
int sph_comm_selftest_timed(SphComm* comm, uint64_t bytes, float* msOut);
int sph_slab_exchange(SphEngine* e, SphComm* comm);

/* ---- agreement of the two ends of a link (round 5; no reference counterpart) -----------------------------------------
 * The sizes of an exchange's messages are derived on each rank by itself: the sender from its own record counts of two exchanges ago, the
 * receiver from the headers it received then, both from their own exchange number and from whether something stirred the fluid lately (an
 * impulse, a container / grid edit, a priming exchange: "hold", whole faces for three exchanges).  That only agrees while every rank makes the
 * same calls; ncclSend / ncclRecv with sizes that do not agree hang or cut records off.  So every sized exchange has a PLAN, 64 bytes: */
typedef struct SphSlabIntent {
    uint32_t magic;            /* "PLAN" */
    uint32_t exchangeNo;       /* sized exchanges this engine has enqueued before this one */
    uint32_t stepNo;           /* boundary-first steps begun */
    uint32_t faceCap;
    uint32_t sendHalo[2], sendMig[2];   /* records this engine's messages to the lower / upper neighbour carry */
    uint32_t recvHalo[2], recvMig[2];   /* records it posts receives for, from the lower / upper neighbour */
    uint32_t holdEvents;       /* impulses / container and grid edits / priming exchanges seen so far */
    uint32_t paramsHash;       /* FNV-1a of the SphParams the exchange is planned under */
    uint32_t flags;            /* 1 paused, 2 message test hook, 4 this exchange holds whole faces */
    uint32_t zRange;           /* z0 | z1 << 16: the cell layers this engine owns */
} SphSlabIntent;
/* ... and nothing moves before the plans of both ends of every link have been compared:
 *   - sph_slab_step_finish_local compares the neighbour ENGINES' plans directly;
 *   - sph_slab_exchange / sph_slab_step_finish send the plan across each link as a FIXED-SIZE message on a stream of its own, wait for the
 *     neighbours' plans on the host (polled, at most the deadline: SPH_ERR_TIMEOUT) and compare.  A difference is SPH_ERR_STATE on BOTH
 *     ranks of the link, with what differs by name ("this rank has seen 4 impulses ..., the upper neighbour 3"), before a sized message is
 *     posted.  Cost: one 64-byte send / receive per neighbour and exchange, and the host's look-ahead shrinks from two exchanges to about
 *     one (the device still holds more than a substep of queued work while the host waits).  sph_slab_set_verify(engine, 0) switches it
 *     off: then there is no host wait on the path beyond the pinned-memory read of counts that are two exchanges old, and agreement rests
 *     on the ranks' call sequences being the same.
 * Behind both there is a device-side check: a header names the sizes its sender's messages carry, the receiver's unpack compares them with the
 * sizes it received: flag 32 (an error of sph_slab_status / sph_slab_download). */
int sph_slab_set_verify(SphEngine* e, int mode /* 1 (default) | 0 */);
/* Limit of every host-side wait for a neighbour (default 30 s). */
int sph_slab_set_deadline(SphEngine* e, double seconds);
/* The plan of the last sized exchange, and (nullable) the host time its handshake waited, in ms. */
int sph_slab_plan(const SphEngine* e, SphSlabIntent* out, float* handshakeMsOut);
/* Host-only: 1 if `neighbour` (the plan of the engine on side 0 = below / 1 = above of `mine`) fits `mine`, else 0 and the reason in `why`. */
int sph_slab_plans_agree(const SphSlabIntent* mine, const SphSlabIntent* neighbour, int side, char* why, size_t whyBytes);
/* sph_sync that cannot hang: polls the engine's streams; SPH_ERR_TIMEOUT after `seconds` (<= 0: the engine's deadline) with where this engine stands. */
int sph_sync_deadline(SphEngine* e, double seconds);
/* Test hook (replaces round 4's environment variable): messages carry the count of two exchanges ago, no margin, no calm rule. */
int sph_slab_debug_tight_messages(SphEngine* e, int on);
/* The engine's exchange pattern on ONE rank, the rank being its own lower and upper neighbour: first the 64-byte plans (with the handshake's
 * polled wait), then the routine sph_slab_exchange itself posts the faces with -- per neighbour two ncclSend / ncclRecv pairs of UNEQUAL sizes
 * in one group: header + counts[2 + d] migrants, counts[d] halo copies -- over faces of capacity faceCap filled with a pattern; every byte is
 * compared (inside a message: arrived; beyond it: untouched).  msOut (nullable): hipEvent time of the faces' group. */
int sph_comm_selftest_faces(SphComm* comm, uint32_t faceCap, const uint32_t counts[4], float* msOut);
/* ---- boundary-first substep: the exchange hidden behind the interior of the SPH pass -----------------------------
 * sph_slab_step_begin = sph_dispatch, except that the SPH pass runs the slot ranges next to the slab's faces first (the
 * five lowest / five highest local cell layers: everything the next pack can touch as long as a substep moves a particle
 * across at most three layers; a substep that does not is reported, flag 16 above), and then, on a second stream of the engine, the pack of the exchange that prepares the
 * NEXT substep -- while the interior slots are still being computed on the engine's stream.  The second half moves the
 * faces and unpacks, still on the second stream; the engine's stream waits for it only at its end:
 *   sph_slab_step_finish(engine, comm)            one process per GPU: grouped ncclSend / ncclRecv (RCCL over xGMI)
 *   sph_slab_step_finish_local(engine, lo, hi)    several slab engines in ONE process: device-to-device copies of the
 *                                                 neighbours' send faces (call every engine's _begin before any _finish_local)
 * Both transports run the same stream / event schedule.  The state a step leaves behind already holds the halo records of
 * the next substep, so a run is: one plain exchange (sph_slab_exchange, or pack_async / unpack_async) to prime it, then
 * only steps; impulses go between steps as usual (they act on the halo copies as on their owners).  Members that move the
 * grid (box centre / half / angles, h, grid_cap) must not change between two steps: sph_slab_step_begin compares the grid with the
 * one the halo records in place were cut for and returns SPH_ERR_STATE (prime again with a plain exchange, then go on).  Every
 * engine of a group must begin a step before any of them finishes it (sph_slab_step_finish_local checks the neighbours' step
 * numbers).  Results are bit-identical to exchange + sph_dispatch. */
int sph_slab_step_begin(SphEngine* e, float overrideDt);
int sph_slab_step_finish(SphEngine* e, SphComm* comm);
int sph_slab_step_finish_local(SphEngine* e, SphEngine* lo, SphEngine* hi);
/* With SPH_OPT_TIMING on: hipEvent times of the LAST boundary-first step, in ms (synchronises): {pack, transfer, unpack} on the
 * exchange stream, then the end of the exchange and the end of the SPH pass (interior included), both measured from the start of
 * the step.  The transfer was hidden behind the interior iff out[3] <= out[4]. */
int sph_slab_step_times(SphEngine* e, float outMs[5]);

/* ---- measurement ----------------------------------------------------------------- */
enum {
    SPH_K_BIN = 0,      /* cell index + histogram   (BuildGrid.comp)            */
    SPH_K_SCAN = 1,     /* exclusive scan + clear   (ClearGrid.comp)            */
    SPH_K_SCATTER = 2,  /* counting-sort scatter + canonical rank               */
    SPH_K_SPH = 3,      /* 27-cell density+force+integrate+XSPH (+fused OBB)    */
    SPH_K_WRITEBACK = 4,/* 80-byte AoS update                                   */
    SPH_K_IMPULSE = 5,  /* WaveImpulse                                          */
    SPH_K_OTHER = 6,
    SPH_K_COUNT = 7
};
/* Accumulated hipEvent milliseconds and launch counts per kernel class since the
 * last reset (SPH_OPT_TIMING must be 1). Synchronises. */
int sph_kernel_times(SphEngine* e, double msOut[SPH_K_COUNT], int64_t launchesOut[SPH_K_COUNT], int reset);

#ifdef __cplusplus
}
#endif
#endif /* SPH_ABI_H */
