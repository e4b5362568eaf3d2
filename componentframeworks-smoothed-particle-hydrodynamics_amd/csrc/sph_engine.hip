// sph_engine.hip -- host driver + C-ABI of the MI355X SPH substep engine.
//
// Replaces the GL compute dispatch path of SPHFluidGPU
// (/root/reference/ComponentFramework/SPHFluid3D.cpp:431-522 DispatchCompute, :604-623
// ApplyWaveImpulse, :713-731 ResetSimulation, :32-83 ctor/dtor) behind include/sph_abi.h.
// No CPU fallback: every compute entry point fails with SPH_ERR_HIP when HIP does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sph_abi.h"
#include "sph_host.h"
#include "sph_kernels.h"
#include "sph_walk.h"

static_assert(sizeof(SphParticle) == 80, "SPHParticle must be 80 bytes (SPHFluid3D.h:12-24)");

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(SPH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

template <class T>
int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return SPH_OK;
}
template <class T>
void dev_free(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

inline int blocks_for(size_t n, int per = sph::kBlock) { return (int)((n + per - 1) / per); }

}  // namespace

struct SphEngine;
static std::set<SphEngine*> g_engines;      // live engines of this process (sph_destroy clears what neighbours hold of a destroyed one)
static std::mutex g_enginesMutex;           // (a host may drive its engines from one thread each)

struct SphEngine {
    hipStream_t stream = nullptr;
    bool ownStream = false;
    SphParams params{};
    SphGridInfo grid{};
    size_t n = 0;
    size_t cap = 0;
    int allocatedCells = 0;
    uint32_t idBase = 0;

    // options
    int optNeighbor = 3, optGridBuild = 0, optAos = 1, optTiming = 0, optGraph = 0;   // optAos: 1 = records materialised on demand (default)
    // hipGraph cache of sph_dispatch_n (SPH_OPT_GRAPH): one executable graph per distinct call
    struct GraphEntry {
        uint64_t key = 0;
        std::vector<unsigned char> material;   // full key material, compared on a hit
        hipGraphExec_t exec = nullptr;
        int postCur = 0;
        bool postAos = false, postAcc = false;
        uint64_t lastUse = 0;
    };
    std::vector<GraphEntry> graphs;
    uint64_t graphClock = 0, graphLaunches = 0, graphCaptures = 0;

    // public contract buffer
    SphParticle* d_aos = nullptr;
    // internal sorted SoA state, double buffered
    float4* d_pos[2] = {nullptr, nullptr};
    float4* d_vel[2] = {nullptr, nullptr};
    float2* d_rp[2] = {nullptr, nullptr};
    float* d_foam[2] = {nullptr, nullptr};
    float4* d_acc = nullptr;
    int cur = 0;
    bool internalValid = false, aosValid = false, accValid = false;
    // grid / sort scratch
    uint2* d_binKey = nullptr;              // k_bin: (cell, arrival slot inside the cell) of every state slot
    uint32_t* d_order = nullptr;
    float4* d_stencil = nullptr;            // SetStencilTargets points (binding 5 of StencilAttract.comp)
    size_t stencilCount = 0;
    int32_t *d_llNext = nullptr, *d_llCell = nullptr, *d_llKey = nullptr;   // linked-list A/B variant (particleNext, particleCell, cellKey)
    uint2* d_tmp = nullptr;
    uint32_t *d_cellCount = nullptr, *d_cellStart = nullptr, *d_blockSums = nullptr;
    int32_t* d_dbg = nullptr;
    size_t dbgCap = 0;
    // z-slab (multi-GPU) mode: this engine owns global cell layers [z0, z1) and keeps one ghost layer per side
    bool slab = false;
    int z0 = 0, z1 = 0, hasLo = 0, hasHi = 0;
    size_t nSlots = 0;                      // state slots in use (live + dead), upper bound of the live count (= cap once the async exchange is used)
    char* d_face[4] = {nullptr, nullptr, nullptr, nullptr};   // engine-owned halo buffers: send lo / hi, recv lo / hi (sph::slab_face_bytes(faceCap): header, migrants, halo copies)
    // message sizing (round 4): the two messages per direction carry the records in use, sized from the counts of TWO exchanges ago
    // (read back asynchronously into pinned memory; both ends of a link see the same numbers: the sender its own count, the receiver the
    // header it got), the face capacity until those exist
    uint32_t* h_cnt = nullptr;              // pinned, 4 slots x 8 words: own halo lo / hi, own migrants lo / hi, received (halo, migrants) from lo, from hi
    hipEvent_t evCnt[4] = {nullptr, nullptr, nullptr, nullptr};
    bool cntValid[4] = {false, false, false, false};
    uint32_t exchangeNo = 0;                // sized exchanges enqueued so far
    uint32_t msgSend[4] = {0, 0, 0, 0}, msgRecv[4] = {0, 0, 0, 0};   // records of the exchange being enqueued: halo lo / hi, migrants lo / hi
    uint64_t sentBytes[2] = {0, 0};         // bytes of the last exchange's messages to lo / hi
    int calmHold = 0;                       // exchanges that still send whole faces because something just changed under the fluid (impulse, container, re-prime)
    uint32_t stepNo = 0;                    // boundary-first steps begun
    // round 5: the PLAN of an exchange and its agreement with the neighbours' plans (slab_plan, slab_intent_mismatch, slab_handshake_rccl)
    SphSlabIntent intent{};                 // what this engine is about to do in exchange intent.exchangeNo: compared with the neighbours' plans before a sized message is posted
    bool intentValid = false;
    uint32_t holdEvents = 0;                // things that set calmHold so far (impulses, container / grid edits, priming exchanges): every rank must have seen the same ones
    int verifyMode = 1;                     // RCCL transport: 1 (default) = the plans cross each link as a fixed 64-byte message and are compared BEFORE the sized messages are posted
    double deadlineSec = 30.0;              // host-side limit of a wait for a neighbour (handshake) and of sph_sync_deadline's default
    float lastDt = 0.0f;                    // dt of the last unpaused dispatch (0: none yet)
    int tightMessages = 0;                  // test hook (sph_slab_debug_tight_messages): the count of two exchanges ago, no margin, no calm rule
    hipStream_t hstream = nullptr;          // the handshake's stream
    hipEvent_t evIntent = nullptr;
    uint32_t *d_intent = nullptr, *h_intent = nullptr;   // device / pinned host, 48 words: [0..15] this engine's plan, [16..31] the lower neighbour's, [32..47] the upper neighbour's
    float handshakeMs = 0.0f;               // host time the last handshake waited for its neighbours
    uint64_t handshakes = 0;
    bool stepPaused = false;                // the pending step is a paused one (nothing was enqueued)
    float packGrid[8] = {0};                // grid (gridMin, cellSize, dims) the halo records in place were cut for
    bool packGridValid = false;
    hipEvent_t evX[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // SPH_OPT_TIMING: pass start, pack start, pack end, transfer end, unpack end of the last step
    hipEvent_t evPassEnd = nullptr;         // SPH_OPT_TIMING: end of the SPH pass (interior included) of the last boundary-first step
    uint32_t faceCap = 0;
    const void* faceAgreedWith = nullptr;   // communicator whose ranks were checked to share faceCap (sph_slab_exchange)
    SphFountain fountain{};                 // fountain* members (SPHFluid3D.h:161-168)
    SphRiver river{};                       // river / terrain members (SPHFluid3D.h:171-196)
    std::vector<float> terrainHeights;      // CPU copy of the heightfield (:175); empty = river step off (:512)
    float* d_terrain = nullptr;             // terrainSSBO (binding 7 of TerrainConstraints.comp)
    size_t terrainCap = 0;
    float4 *d_sPV = nullptr, *d_sOwn = nullptr;   // sorted copy of the entry state: 32-byte records (pos, 1/rho | vel, P) + own data
    size_t sortedCap = 0;
    float4* d_shapeTab = nullptr;           // sampled curve of container shapes 9/11/12/14 (128 points)
    float shapeKey[8] = {-1.0f};            // parameters the uploaded table was built from
    sph::ShapeTab shapeTab{};
    // boundary-first substep (sph_slab_step_*): the halo exchange of the NEXT substep runs on xstream beside the interior of the SPH pass
    hipStream_t xstream = nullptr;
    hipStream_t bstream = nullptr;          // LOW priority: the interior launch of the SPH pass, beside the face launches on the engine's stream and the exchange on xstream
    hipEvent_t evBoundary = nullptr, evPacked = nullptr, evDone = nullptr, evSorted = nullptr, evInterior = nullptr;
    hipEvent_t peerDone[2] = {nullptr, nullptr};   // evDone of the lower / upper neighbour ENGINE of the last local transfer: it has read this engine's send face
    bool stepPending = false;               // sph_slab_step_begin ran, its finish has not yet
    bool slabOrderValid = false;            // slots are in the order of the last counting sort and no particle can have moved by more than a layer since
    float lastContainer[15] = {0};          // container / grid members of the last dispatch (a change may move particles by many cells, or the grid under them)
    uint32_t* d_slabCnt = nullptr;          // [0] lo records, [1] hi records, [2] live count, [3] download count, [4] flags, [5] / [6] counts of the last async pack
    int debugFlags = 0;
    unsigned long long* d_stats = nullptr;   // k_sph_walk / k_sph_list diagnostics (SPH_OPT_DEBUG bit 3), see sph_debug_counters

    std::vector<SphParticle> hostInit;   // SPHFluidGPU::particles: initial state only

    // timing
    struct Ev { int cls; hipEvent_t a, b; };
    std::vector<Ev> evLive;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evPool;
    double kms[SPH_K_COUNT] = {0};
    int64_t klaunch[SPH_K_COUNT] = {0};
};

namespace {

using namespace sph;

// Members that decide where the container walls and the grid are (for the slab exchange's reduced scan).
void container_key(const SphParams& q, float out[15]) {
    const float v[15] = {q.param_boxCenter[0], q.param_boxCenter[1], q.param_boxCenter[2], q.param_boxHalf[0], q.param_boxHalf[1], q.param_boxHalf[2],
                         q.param_boxEulerDeg[0], q.param_boxEulerDeg[1], q.param_boxEulerDeg[2], (float)q.param_shapeType,
                         q.param_shapeAux[0], q.param_shapeAux[1], q.param_shapeAux[2], q.param_h, (float)q.grid_cap};
    std::memcpy(out, v, sizeof(v));
}
// Something every rank knows to stir the fluid (an impulse, a container / grid edit, a priming exchange): whole faces for three exchanges, and one more
// event on the count every rank's plan carries (a rank that saw an event its neighbour did not is found by the plans' comparison, not by a hang).
void slab_hold(SphEngine* e) {
    e->calmHold = 3;
    e->holdEvents += 1u;
}
bool slab_ranges_usable(const SphEngine* e) {
    if (!e->slabOrderValid) return false;
    float cont[15];
    container_key(e->params, cont);
    return std::memcmp(cont, e->lastContainer, sizeof(cont)) == 0;   // same walls and same grid as the dispatch that sorted the slots
}

int flush_events(SphEngine* e) {
    if (e->evLive.empty()) return SPH_OK;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->xstream) HIP_TRY(hipStreamSynchronize(e->xstream));
    for (auto& ev : e->evLive) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        e->kms[ev.cls] += ms;
        e->klaunch[ev.cls] += 1;
        e->evPool.emplace_back(ev.a, ev.b);
    }
    e->evLive.clear();
    return SPH_OK;
}

struct Timed {   // RAII-ish bracket around one kernel launch when SPH_OPT_TIMING is on
    SphEngine* e;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    Timed(SphEngine* e_, int cls_, hipStream_t st_ = nullptr) : e(e_), cls(cls_), st(st_ ? st_ : e_->stream) {
        if (!e->optTiming || (e->optTiming == 2 && cls != SPH_K_SPH)) return;
        if (e->evPool.empty()) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        } else {
            a = e->evPool.back().first; b = e->evPool.back().second; e->evPool.pop_back();
        }
        (void)hipEventRecord(a, st);
    }
    ~Timed() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        e->evLive.push_back({cls, a, b});
        if (e->evLive.size() >= 8192) (void)flush_events(e);
    }
};

void free_particle_buffers(SphEngine* e) {
    dev_free(e->d_aos);
    for (int b = 0; b < 2; ++b) { dev_free(e->d_pos[b]); dev_free(e->d_vel[b]); dev_free(e->d_rp[b]); dev_free(e->d_foam[b]); }
    dev_free(e->d_acc);
    dev_free(e->d_binKey); dev_free(e->d_order); dev_free(e->d_tmp);
    dev_free(e->d_slabCnt); for (auto& f : e->d_face) dev_free(f);
    e->faceCap = 0; dev_free(e->d_shapeTab); dev_free(e->d_sPV); dev_free(e->d_sOwn);
    for (auto& g : e->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    e->graphs.clear();
    dev_free(e->d_llNext); dev_free(e->d_llCell); dev_free(e->d_llKey);
    e->cap = 0;
}
void free_grid_buffers(SphEngine* e) {
    if (!e->graphs.empty() && e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto& g : e->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);   // captured launches hold these addresses
    e->graphs.clear();
    dev_free(e->d_cellCount); dev_free(e->d_cellStart); dev_free(e->d_blockSums);
    e->allocatedCells = 0;
}

int alloc_particle_buffers(SphEngine* e, size_t n) {
    free_particle_buffers(e);
    int rc;
    if ((rc = dev_alloc(&e->d_aos, n))) return rc;
    for (int b = 0; b < 2; ++b) {
        if ((rc = dev_alloc(&e->d_pos[b], n))) return rc;
        if ((rc = dev_alloc(&e->d_vel[b], n))) return rc;
        if ((rc = dev_alloc(&e->d_rp[b], n))) return rc;
        if ((rc = dev_alloc(&e->d_foam[b], n))) return rc;
    }
    if ((rc = dev_alloc(&e->d_acc, n))) return rc;
    if ((rc = dev_alloc(&e->d_binKey, n))) return rc;
    if ((rc = dev_alloc(&e->d_order, n))) return rc;
    if ((rc = dev_alloc(&e->d_tmp, n))) return rc;
    if ((rc = dev_alloc(&e->d_slabCnt, 16))) return rc;
    HIP_TRY(hipMemsetAsync(e->d_slabCnt, 0, 16 * sizeof(uint32_t), e->stream));
    e->cap = n;
    return SPH_OK;
}

// cellHead realloc of SPHFluid3D.cpp:440-447 (only when the cell count changed)
int local_cells(const SphEngine* e) {
    return e->slab ? e->grid.dims[0] * e->grid.dims[1] * (e->z1 - e->z0 + 2) : e->grid.numCells;
}

int ensure_grid_buffers(SphEngine* e) {
    const int want = local_cells(e);
    if (want == e->allocatedCells && e->d_cellCount) return SPH_OK;
    free_grid_buffers(e);
    const size_t C = (size_t)want;
    int rc;
    if ((rc = dev_alloc(&e->d_cellCount, C))) return rc;
    if ((rc = dev_alloc(&e->d_cellStart, C + 1))) return rc;
    if ((rc = dev_alloc(&e->d_blockSums, (size_t)blocks_for(C, kScanTile) + 1))) return rc;
    HIP_TRY(hipMemsetAsync(e->d_cellCount, 0, C * sizeof(uint32_t), e->stream));
    e->allocatedCells = want;
    return SPH_OK;
}

constexpr long long kMaxCells = 1ll << 30;       // int32 cell indices and 4-byte-per-cell buffers stay far from overflow
constexpr size_t kMaxParticles = (size_t)1 << 30;
int validate_params(const SphParams& p) {
    if (!(p.param_h > 0.0f)) return fail(SPH_ERR_ARG, "param_h must be > 0");
    SphGridInfo g;
    compute_grid_extents(p, g);
    for (int a = 0; a < 3; ++a)
        if (g.dims[a] > 1024) return fail(SPH_ERR_CAPACITY, "grid axis %d has %d cells; the engine packs cell coordinates in 10 bits per axis (max 1024)", a, g.dims[a]);
    const long long nc = (long long)g.dims[0] * g.dims[1] * g.dims[2];
    if (nc > kMaxCells) return fail(SPH_ERR_CAPACITY, "grid of %d x %d x %d cells exceeds %lld cells (grid_cap %d)", g.dims[0], g.dims[1], g.dims[2], kMaxCells, p.grid_cap);
    return SPH_OK;
}

int set_particles(SphEngine* e, const SphParticle* host, size_t n) {
    int rc;
    if (n > kMaxParticles) return fail(SPH_ERR_CAPACITY, "%zu particles exceed the engine limit of %zu", n, kMaxParticles);
    if (n > e->cap || !e->d_aos) { if ((rc = alloc_particle_buffers(e, n))) return rc; }
    e->n = n;
    e->hostInit.assign(host, host + n);
    if (n) HIP_TRY(hipMemcpyAsync(e->d_aos, host, n * sizeof(SphParticle), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));   // `host` may be a temporary of the caller
    e->aosValid = true; e->internalValid = false; e->accValid = false; e->cur = 0;
    return SPH_OK;
}

int import_state(SphEngine* e) {
    if (e->internalValid) return SPH_OK;
    if (e->slab) return fail(SPH_ERR_STATE, "slab engine without a valid state");
    if (!e->aosValid) return fail(SPH_ERR_STATE, "neither the AoS nor the internal state is valid");
    if (e->n) {
        Timed t(e, SPH_K_OTHER);
        hipLaunchKernelGGL(k_import, dim3(blocks_for(e->n)), dim3(kBlock), 0, e->stream, e->d_aos, e->d_pos[e->cur],
                           e->d_vel[e->cur], e->d_rp[e->cur], e->d_foam[e->cur], e->idBase, (int)e->n);
    }
    HIP_TRY(hipGetLastError());
    e->internalValid = true;
    e->accValid = false;
    return SPH_OK;
}

// ClearGrid + BuildGrid as a counting sort: after this, d_cellStart/d_order describe the
// current state buffer.
// commitLive (z-slab dispatch only): k_rank also stores the live count as the new slots-in-use count of the exchange.
int build_grid(SphEngine* e, const SimK& k, bool commitLive = false) {
    const int n = (int)(e->slab ? e->nSlots : e->n), C = k.numCells;
    const int nb = blocks_for(n), sb = blocks_for((size_t)C, kScanTile);
    const bool sortedCopy = e->optGridBuild == 0;     // k_rank also writes the sorted copy the SPH pass reads
    if (sortedCopy && (e->sortedCap < e->cap || !e->d_sPV)) {
        int rc;
        dev_free(e->d_sPV); dev_free(e->d_sOwn);
        e->d_sPV = e->d_sOwn = nullptr; e->sortedCap = 0;
        if ((rc = dev_alloc(&e->d_sPV, 2 * e->cap)) || (rc = dev_alloc(&e->d_sOwn, e->cap))) return rc;
        e->sortedCap = e->cap;
    }
    if (n) {
        Timed t(e, SPH_K_BIN);
        hipLaunchKernelGGL(k_bin, dim3(nb), dim3(kBlock), 0, e->stream, k, e->d_pos[e->cur], e->d_binKey, e->d_cellCount, n,
                           e->slab ? e->d_slabCnt + 2 : (const uint32_t*)nullptr);
    }
    {
        Timed t(e, SPH_K_SCAN);
        const int rawSums = sb <= kScanFusedBlocks ? 1 : 0;
        hipLaunchKernelGGL(k_scan_reduce, dim3(sb), dim3(kBlock), 0, e->stream, e->d_cellCount, e->d_blockSums, C);
        if (!rawSums) hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(kBlock), 0, e->stream, e->d_blockSums, sb);
        hipLaunchKernelGGL(k_scan_apply, dim3(sb), dim3(kBlock), 0, e->stream, e->d_cellCount, e->d_blockSums, e->d_cellStart, C, (uint32_t)n, rawSums);
    }
    if (n) {
        Timed t(e, SPH_K_SCATTER);
        hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(kBlock), 0, e->stream, e->d_binKey, e->d_cellStart, e->d_tmp, n,
                           e->slab ? e->d_slabCnt + 2 : (const uint32_t*)nullptr);
        if (sortedCopy) {
            hipLaunchKernelGGL((k_rank<true>), dim3(nb), dim3(kBlock), 0, e->stream, e->d_tmp, e->d_cellStart, e->d_order, n, C,
                               e->d_pos[e->cur], e->d_vel[e->cur], e->d_rp[e->cur], e->d_foam[e->cur], e->d_sPV, e->d_sOwn, k.gx, k.gy,
                               (commitLive && e->slab) ? e->d_slabCnt + 2 : (uint32_t*)nullptr);
        } else {
            hipLaunchKernelGGL((k_rank<false>), dim3(nb), dim3(kBlock), 0, e->stream, e->d_tmp, e->d_cellStart, e->d_order, n, C,
                               nullptr, e->d_vel[e->cur], nullptr, nullptr, nullptr, nullptr, k.gx, k.gy,
                               (commitLive && e->slab) ? e->d_slabCnt + 2 : (uint32_t*)nullptr);
        }
    }
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}

int writeback(SphEngine* e) {
    if (e->slab) return fail(SPH_ERR_STATE, "a slab engine has no local 80-byte array: use sph_slab_download");
    if (e->aosValid) return SPH_OK;
    if (!e->internalValid) return fail(SPH_ERR_STATE, "no valid particle state");
    if (!e->accValid) return fail(SPH_ERR_STATE, "internal acc buffer is stale");
    if (e->n) {
        Timed t(e, SPH_K_WRITEBACK);
        hipLaunchKernelGGL(k_writeback, dim3(blocks_for(e->n)), dim3(kBlock), 0, e->stream, e->d_aos, e->d_pos[e->cur], e->d_vel[e->cur],
                           e->d_rp[e->cur], e->d_foam[e->cur], e->d_acc, e->idBase, (int)e->n);
    }
    HIP_TRY(hipGetLastError());
    e->aosValid = true;
    return SPH_OK;
}

// Table of the sampled-curve shapes for k_obb_ext, re-uploaded only when its inputs change.
int ensure_shape_table(SphEngine* e) {
    const SphParams& p = e->params;
    const float key[8] = {(float)p.param_shapeType, p.param_boxHalf[0], p.param_boxHalf[1], p.param_boxHalf[2],
                          p.param_shapeAux[0], p.param_shapeAux[1], p.param_shapeAux[2], 1.0f};
    if (e->d_shapeTab && std::memcmp(key, e->shapeKey, sizeof(key)) == 0) return SPH_OK;
    int rc;
    if (!e->d_shapeTab && (rc = dev_alloc(&e->d_shapeTab, 128))) return rc;
    float pts[384], b0[3];
    const int cnt = shape_table(p, pts, b0);
    float4 host[128];
    for (int i = 0; i < cnt; ++i) host[i] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], 0.0f);
    if (cnt) {
        // rare (parameter change): a synchronous copy keeps the pageable staging array safe
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipMemcpy(e->d_shapeTab, host, sizeof(float4) * (size_t)cnt, hipMemcpyHostToDevice));
    }
    e->shapeTab = sph::ShapeTab{e->d_shapeTab, cnt, b0[0], b0[1], b0[2]};
    std::memcpy(e->shapeKey, key, sizeof(key));
    return SPH_OK;
}

// boundaryFirst (z-slab engines, sph_slab_step_begin): the SPH pass runs the slot ranges next to the slab's faces first and
// records e->evBoundary behind them, so that the pack of the next exchange can start while the interior is computed.
int dispatch_one(SphEngine* e, float overrideDt, bool boundaryFirst = false) {
    if (e->params.param_pause) { if (boundaryFirst) HIP_TRY(hipEventRecord(e->evBoundary, e->stream)); return SPH_OK; }                               // SPHFluid3D.cpp:432
    int rc;
    if ((rc = validate_params(e->params))) return rc;
    const float dt = overrideDt > 0.0f ? overrideDt : e->params.param_timeStep;   // :434
    e->lastDt = dt;
    compute_grid_extents(e->params, e->grid);                               // :439
    if ((rc = ensure_grid_buffers(e))) return rc;                           // :440-447
    SimK k;
    make_simk(e->params, e->grid, dt, k);
    if (e->slab) {
        if (e->z1 > e->grid.dims[2] || e->z0 < 0) return fail(SPH_ERR_STATE, "slab [%d,%d) outside the %d-layer grid", e->z0, e->z1, e->grid.dims[2]);
        k.gz = e->z1 - e->z0 + 2; k.numCells = k.gx * k.gy * k.gz; k.zoff = e->z0 - 1;
        // the pack that follows this substep may restrict itself to the ends of the slot range (same container as the substep
        // before, sorted slots): a particle that this substep carries from the middle into a face layer is then reported
        float contNow[15];
        container_key(e->params, contNow);
        if (e->optGridBuild != 1 && std::memcmp(contNow, e->lastContainer, sizeof(contNow)) == 0) k.slabFlags = e->d_slabCnt + 4;
    }
    if ((rc = import_state(e))) return rc;
    const int n = (int)(e->slab ? e->nSlots : e->n);
    StateIn in{e->d_pos[e->cur], e->d_vel[e->cur], e->d_rp[e->cur], e->d_foam[e->cur]};
    const int nx = e->cur ^ 1;
    if (k.obbDeferred && (rc = ensure_shape_table(e))) return rc;
    // the array is current: keep it current from the SPH pass (not when OBB runs as its own pass afterwards)
    const bool fuseAos = !e->slab && e->optAos == 0 && e->aosValid && !k.obbDeferred;
    StateOut out{e->d_pos[nx], e->d_vel[nx], e->d_rp[nx], e->d_foam[nx], e->d_acc, fuseAos ? e->d_aos : nullptr, e->idBase};
    if (e->optGridBuild == 1) {
        // ---- A/B variant: the reference's atomicExchange linked lists (no sorting) ----
        if (e->slab) return fail(SPH_ERR_STATE, "the linked-list variant is single-GPU only");
        if (!e->d_llNext) {
            if ((rc = dev_alloc(&e->d_llNext, e->cap)) || (rc = dev_alloc(&e->d_llCell, e->cap)) || (rc = dev_alloc(&e->d_llKey, e->cap))) return rc;
        }
        int32_t* cellHead = reinterpret_cast<int32_t*>(e->d_cellStart);      // C+1 ints, reused as cellHead
        {
            Timed t(e, SPH_K_SCAN);
            hipLaunchKernelGGL(k_ll_clear, dim3(blocks_for((size_t)k.numCells)), dim3(kBlock), 0, e->stream, cellHead, k.numCells);
        }
        if (n) {
            {
                Timed t(e, SPH_K_BIN);
                hipLaunchKernelGGL(k_ll_build, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, k, in.pos, cellHead, e->d_llNext, e->d_llCell, e->d_llKey, n);
            }
            Timed t(e, SPH_K_SPH);
            hipLaunchKernelGGL(k_sph_ll, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, k, in, out, cellHead, e->d_llNext, n);
        }
    } else {
    if ((rc = build_grid(e, k, true))) return rc;                           // :449-468
    if (n) {                                                                // :470-509 (SPH + OBB fused)
        if (!e->d_sPV || e->sortedCap < (size_t)n) return fail(SPH_ERR_STATE, "sorted copy missing");
        const uint32_t* live = e->slab ? e->d_cellStart + k.numCells : nullptr;
        SortedIn S{e->d_sPV, e->d_sOwn};
        Timed t(e, SPH_K_SPH);
        if (e->optNeighbor >= 3 && (size_t)n < ((size_t)1 << 27)) {        // (buffer loads address the sorted copy with 32-bit byte offsets)
            const dim3 grid(8 * ((blocks_for(n, 256) + 7) / 8));
            auto walk = [&](hipStream_t st, const uint32_t* lo, const uint32_t* hi, bool bothEnds = false) {
                const int dbg = (e->debugFlags & ~256) | (bothEnds ? 256 : 0);    // (bit 8: the launch covers [0, *lo) and [*hi, end) instead of [*lo, *hi))
                if (k.h2 <= 1.0f)
                    hipLaunchKernelGGL((k_sph_walk<SPH_WALK_MAXN, SPH_WALK_UNROLL, SPH_WALK_CAP, true>), grid, dim3(256), 0, st, k, S, in, out,
                                       e->d_order, e->d_cellStart, live, n, dbg, e->d_stats, lo, hi);
                else
                    hipLaunchKernelGGL((k_sph_walk<SPH_WALK_MAXN, SPH_WALK_UNROLL, SPH_WALK_CAP, false>), grid, dim3(256), 0, st, k, S, in, out,
                                       e->d_order, e->d_cellStart, live, n, dbg, e->d_stats, lo, hi);
            };
            // The pack of the next exchange only reads the slots of the kSlabDepth lowest / highest local cell layers (k_slab_pack,
            // under the same conditions): those two slot ranges first, the event, then everything in between.
            float contNow[15];
            container_key(e->params, contNow);
            const bool split = boundaryFirst && e->slab && k.gz > 2 * kSlabDepth && !k.obbDeferred && std::memcmp(contNow, e->lastContainer, sizeof(contNow)) == 0;
            const bool timedStep = boundaryFirst && e->optTiming && e->evPassEnd;
            if (split) {
                const uint32_t* endLo = e->d_cellStart + kSlabDepth * (size_t)(k.gx * k.gy);
                const uint32_t* startHi = e->d_cellStart + (size_t)(k.gz - kSlabDepth) * (size_t)(k.gx * k.gy);
                // Both face ranges go FIRST, as ONE launch on the engine's stream (one tail instead of two) -- the exchange behind them needs them
                // early: the transfer has to fit beside the interior -- and the interior goes to a LOW-priority stream of its own that starts
                // with the face launch: beside it and, later, beside the pack / unpack kernels of the high-priority exchange stream, it fills
                // whatever they leave free.  Same inputs, disjoint output slots; the engine's stream joins the interior at the end of the pass.
                // Measured schedules (one rank's share of configs[4], bench.py --slab-path): this one 1.313-1.316 ms per substep; two face
                // launches + interior behind the first 1.351-1.355; merged faces + interior behind them 1.348-1.354
                // (profiles/r03_slab_face_launch_schedules.txt, profiles/r04_slab_face_launch_schedules.txt).
                HIP_TRY(hipEventRecord(e->evSorted, e->stream));
                walk(e->stream, endLo, startHi, true);
                HIP_TRY(hipEventRecord(e->evBoundary, e->stream));
                HIP_TRY(hipStreamWaitEvent(e->bstream, e->evSorted, 0));
                walk(e->bstream, endLo, startHi);
                HIP_TRY(hipEventRecord(e->evInterior, e->bstream));
                boundaryFirst = false;                                     // recorded
                HIP_TRY(hipStreamWaitEvent(e->stream, e->evInterior, 0));
            } else {
                walk(e->stream, nullptr, nullptr);
            }
            if (timedStep) HIP_TRY(hipEventRecord(e->evPassEnd, e->stream));
        } else if (e->optNeighbor >= 2) {
            hipLaunchKernelGGL((k_sph_list<SPH_LIST_MAXN, SPH_LIST_UNROLL, SPH_LIST_CAP>), dim3(8 * ((blocks_for(n, SPH_LIST_BLOCK) + 7) / 8)), dim3(SPH_LIST_BLOCK), 0, e->stream, k, S, in, out,
                               e->d_order, e->d_cellStart, live, n, e->debugFlags, e->d_stats);
        } else {
            hipLaunchKernelGGL(k_sph_slow, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, k, S, in, out, e->d_order, e->d_cellStart, n);
        }
    }
    }
    if (n && k.obbDeferred) {                                               // :495-509 for shapes 7..14
        Timed t(e, SPH_K_OTHER);
        const uint32_t* live = (e->slab && e->optGridBuild != 1) ? e->d_cellStart + k.numCells : nullptr;
        hipLaunchKernelGGL(k_obb_ext, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, k, e->shapeTab, out.pos, out.vel, live, n,
                           (e->slab && e->optGridBuild != 1) ? (const float4*)e->d_sOwn : (const float4*)nullptr);
    }
    const bool riverOn = e->river.riverMode && !e->terrainHeights.empty();   // :512
    if (riverOn) {                                                           // :511-516, DispatchTerrainConstraints / ChannelConstraint / StreamEmit
        if (e->slab) return fail(SPH_ERR_STATE, "riverMode on a z-slab engine: recycled particles jump across slabs (single-GPU engines only)");
        if (n) {
            const SphParams& p = e->params;
            const SphRiver& r = e->river;
            RiverK rk{r.terrainW, r.terrainH, r.terrainWorldMinX, r.terrainWorldMinZ, r.terrainWorldSizeX, r.terrainWorldSizeZ,
                      0.02f, 1.0f - 0.05f, p.param_boxCenter[0], r.riverAmp, r.riverFreq, r.riverPhase, r.riverChannelWidth,
                      80.0f, p.param_timeStep, r.riverSinkY, r.riverSinkZMax, r.riverEmitterPos[0], r.riverEmitterPos[1], r.riverEmitterPos[2],
                      r.riverEmitterVel[0], r.riverEmitterVel[1], r.riverEmitterVel[2], r.riverEmitterRadius,
                      r.riverSinkZMax - r.riverEmitterPos[2], p.param_restDensity};
            Timed t(e, SPH_K_OTHER);
            hipLaunchKernelGGL(k_river, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, rk, e->d_terrain, out.pos, out.vel, out.rp,
                               fuseAos ? (float4*)nullptr : e->d_acc, fuseAos ? e->d_aos : (SphParticle*)nullptr, e->d_aos, e->idBase, n);
        }
    }
    if (e->fountain.fountainMode && !e->river.riverMode) {                  // :519, :526-544
        if (e->slab) return fail(SPH_ERR_STATE, "fountainMode on a z-slab engine: recycled particles jump across slabs (single-GPU engines only)");
        if (n) {
            float half[3];
            effective_half(e->params, half);
            const SphParams& p = e->params;
            const SphFountain& f = e->fountain;
            FountainK fk{p.param_boxCenter[0] + f.fountainOffset[0], p.param_boxCenter[1] + f.fountainOffset[1],
                         p.param_boxCenter[2] + f.fountainOffset[2], f.fountainRadius, f.fountainSpread, f.fountainJetSpeedLive,
                         (p.param_boxCenter[1] - half[1]) + f.fountainDrainLevel, std::fmin(1.0f, f.fountainDrainPerSec * dt),
                         p.param_restDensity, f.fountainSeed * 747796405u};
            Timed t(e, SPH_K_OTHER);
            const uint32_t* live = (e->slab && e->optGridBuild != 1) ? e->d_cellStart + k.numCells : nullptr;
            hipLaunchKernelGGL(k_fountain, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, fk, out.pos, out.vel, out.rp,
                               fuseAos ? (float4*)nullptr : e->d_acc, fuseAos ? e->d_aos : (SphParticle*)nullptr, e->idBase, live, n);
        }
        e->fountain.fountainSeed++;
    }
    HIP_TRY(hipGetLastError());
    if (boundaryFirst) HIP_TRY(hipEventRecord(e->evBoundary, e->stream));   // no split launch: the pack waits for the whole pass
    e->cur = nx;
    e->accValid = !fuseAos;
    e->aosValid = fuseAos;
    if (e->slab) {   // the sorted output holds exactly the live particles: their count is on the device (slabCnt[2])
        const bool sorted = e->optGridBuild != 1 && n > 0;
        if (!sorted) HIP_TRY(hipMemcpyAsync(e->d_slabCnt + 2, e->d_cellStart + k.numCells, sizeof(uint32_t), hipMemcpyDeviceToDevice, e->stream));   // (k_rank stored it otherwise)
        // the next k_slab_pack may restrict itself to the ends of the slot range if this substep cannot have moved a
        // particle by more than one layer: a container that did not change under the fluid (and the SPH pass reports the particle that
        // jumps from the middle into a face layer all the same, slab_check_layer_move)
        float cont[15];
        container_key(e->params, cont);
        const bool sameContainer = std::memcmp(cont, e->lastContainer, sizeof(cont)) == 0;
        if (!sameContainer) slab_hold(e);                    // the walls moved under the fluid: whole faces until the counts have been seen calm again
        std::memcpy(e->lastContainer, cont, sizeof(cont));
        e->slabOrderValid = sorted && sameContainer;
        return SPH_OK;
    }
    if (e->optAos == 0 && !e->aosValid) return writeback(e);
    return SPH_OK;
}

}  // namespace

namespace {
using namespace sph;

// What the unpack kernels need to tell whether a received particle respects the exchange's one-layer assumption.
SlabGeom slab_geom(SphEngine* e) {
    sph::compute_grid_extents(e->params, e->grid);
    SlabGeom g;
    g.gminz = e->grid.gridMin[2]; g.cellSize = e->grid.cellSize; g.gzGlobal = e->grid.dims[2];
    g.z0 = e->z0; g.z1 = e->z1; g.hasLo = e->hasLo; g.hasHi = e->hasHi;
    return g;
}

// Device-side error flags of the exchange (slabCnt[4]) as a status: every entry point that synchronises anyway reports them.
int slab_flags_error(const SphEngine* e, uint32_t flags, uint32_t nLo, uint32_t nHi) {
    if (flags & 1u) return fail(SPH_ERR_CAPACITY, "halo send buffer overflowed (%u / %u records for a capacity of %u)", nLo, nHi, e->faceCap);
    if (flags & 2u) return fail(SPH_ERR_CAPACITY, "slab capacity %zu exceeded while appending halo records", e->cap);
    if (flags & 4u) return fail(SPH_ERR_HIP, "a received halo message did not start with a valid header (magic / count): failed or garbled receive");
    if (flags & 8u) return fail(SPH_ERR_CAPACITY, "a neighbour rank had more halo records than its message could carry (face capacity %u)", e->faceCap);
    if (flags & 32u) return fail(SPH_ERR_STATE, "the two ends of a link sized an exchange's messages differently (a received header names other message sizes than this engine received): "
                                                "the ranks' call sequences differ");
    // (flag 16 -- a particle crossed more cell layers in z within one substep than the exchange follows -- loses nothing: it is a notice
    //  that the decomposed run no longer equals the single-domain run, carried by sph_slab_status's out[4], cleared by sph_slab_clear_flags)
    return SPH_OK;
}

template <class K>
int launch_impulse(SphEngine* e, const K& kk) {
    int rc;
    if ((rc = import_state(e))) return rc;
    const size_t nw = e->slab ? e->nSlots : e->n;
    if (e->slab) slab_hold(e);                              // (z-slabs: an impulse may set a face's record count moving: whole faces for three exchanges)
    if (nw) {
        Timed t(e, SPH_K_IMPULSE);
        hipLaunchKernelGGL((k_impulse<K>), dim3(blocks_for(nw)), dim3(kBlock), 0, e->stream, kk, e->d_pos[e->cur], e->d_vel[e->cur],
                           (e->aosValid && !e->slab) ? e->d_aos : nullptr, e->idBase, (int)nw);
    }
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}
}  // namespace

// ======================================================================== C-ABI
extern "C" {

int sph_abi_version(void) { return SPH_ABI_VERSION; }
const char* sph_last_error(void) { return g_err.c_str(); }

int sph_params_default(SphParams* out) {
    if (!out) return fail(SPH_ERR_ARG, "null params");
    sph::params_default(*out);
    return SPH_OK;
}
int sph_rotation_mat3(const float eulerDeg[3], float outM[9]) {
    if (!eulerDeg || !outM) return fail(SPH_ERR_ARG, "null argument");
    sph::rotation_mat3(eulerDeg, outM);
    return SPH_OK;
}
int sph_effective_half(const SphParams* params, float outHalf[3]) {
    if (!params || !outHalf) return fail(SPH_ERR_ARG, "null argument");
    sph::effective_half(*params, outHalf);
    return SPH_OK;
}
int sph_compute_grid_extents(const SphParams* params, SphGridInfo* out) {
    if (!params || !out) return fail(SPH_ERR_ARG, "null argument");
    sph::compute_grid_extents(*params, *out);
    return SPH_OK;
}
int sph_spawn_particles(const SphParams* params, size_t nRequested, uint32_t seed, SphParticle* out, size_t* nOut, float* massOut) {
    if (!params || !out || !nOut || !massOut) return fail(SPH_ERR_ARG, "null argument");
    std::vector<SphParticle> v;
    float m;
    sph::spawn_particles(*params, nRequested, seed, v, m);
    std::memcpy(out, v.data(), v.size() * sizeof(SphParticle));
    *nOut = v.size();
    *massOut = m;
    return SPH_OK;
}

static int create_common(SphEngine** out, const SphParams* params, void* stream, SphEngine** made) {
    if (!out || !params) return fail(SPH_ERR_ARG, "null argument");
    *out = nullptr;
    int rc;
    if ((rc = validate_params(*params))) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(SPH_ERR_HIP, "no HIP device: this engine has no CPU fallback");
    SphEngine* e = new SphEngine();
    sph_fountain_default(&e->fountain);
    sph_river_default(&e->river);
    e->params = *params;
    if (stream) { e->stream = (hipStream_t)stream; e->ownStream = false; }
    else {
        hipError_t er = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
        if (er != hipSuccess) { delete e; return fail(SPH_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(er)); }
        e->ownStream = true;
    }
    *made = e;
    { std::lock_guard<std::mutex> lk(g_enginesMutex); g_engines.insert(e); }
    return SPH_OK;
}

int sph_create(SphEngine** out, size_t nRequested, const SphParams* params, uint32_t seed, void* stream) {
    SphEngine* e = nullptr;
    int rc = create_common(out, params, stream, &e);
    if (rc) return rc;
    std::vector<SphParticle> v;
    float m;
    sph::spawn_particles(e->params, nRequested, seed, v, m);          // SPHFluid3D.cpp:50
    e->params.param_mass = m;                                         // :92
    sph::compute_grid_extents(e->params, e->grid);                    // :53
    if ((rc = set_particles(e, v.data(), v.size())) || (rc = ensure_grid_buffers(e))) { sph_destroy(e); return rc; }
    *out = e;
    return SPH_OK;
}

int sph_create_from_particles(SphEngine** out, const SphParticle* particles, size_t n, const SphParams* params, void* stream) {
    if (!particles && n) return fail(SPH_ERR_ARG, "null particles");
    SphEngine* e = nullptr;
    int rc = create_common(out, params, stream, &e);
    if (rc) return rc;
    sph::compute_grid_extents(e->params, e->grid);
    if ((rc = set_particles(e, particles, n)) || (rc = ensure_grid_buffers(e))) { sph_destroy(e); return rc; }
    *out = e;
    return SPH_OK;
}

int sph_destroy(SphEngine* e) {
    if (!e) return SPH_OK;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->xstream) (void)hipStreamSynchronize(e->xstream);
    if (e->bstream) (void)hipStreamSynchronize(e->bstream);
    free_particle_buffers(e);
    free_grid_buffers(e);
    dev_free(e->d_dbg);
    dev_free(e->d_stencil);
    dev_free(e->d_terrain);
    dev_free(e->d_stats);
    for (auto& ev : e->evLive) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto& ev : e->evPool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (e->xstream) { (void)hipStreamSynchronize(e->xstream); (void)hipStreamDestroy(e->xstream); }
    if (e->bstream) { (void)hipStreamSynchronize(e->bstream); (void)hipStreamDestroy(e->bstream); }
    {
        std::lock_guard<std::mutex> lk(g_enginesMutex);
        for (SphEngine* o : g_engines)                                        // a neighbour engine of this process must not wait for an event that is gone
            for (auto& pd : o->peerDone) if (pd && pd == e->evDone) pd = nullptr;
        g_engines.erase(e);
    }
    for (hipEvent_t ev : {e->evBoundary, e->evPacked, e->evDone, e->evSorted, e->evInterior, e->evPassEnd}) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->evX) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->evCnt) if (ev) (void)hipEventDestroy(ev);
    if (e->h_cnt) (void)hipHostFree(e->h_cnt);
    if (e->hstream) { (void)hipStreamSynchronize(e->hstream); (void)hipStreamDestroy(e->hstream); }
    if (e->evIntent) (void)hipEventDestroy(e->evIntent);
    if (e->h_intent) (void)hipHostFree(e->h_intent);
    dev_free(e->d_intent);
    if (e->ownStream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return SPH_OK;
}

int sph_reset(SphEngine* e, size_t nRequested, uint32_t seed) {       // SPHFluid3D.cpp:713-731
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    int rc;
    if ((rc = validate_params(e->params))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<SphParticle> v;
    float m;
    if (e->river.riverMode && !e->terrainHeights.empty())             // :104
        sph::spawn_river_particles(e->params, e->river, e->terrainHeights.data(), nRequested, seed, v, m);
    else
        sph::spawn_particles(e->params, nRequested, seed, v, m);
    e->params.param_mass = m;
    sph::compute_grid_extents(e->params, e->grid);
    free_particle_buffers(e);                                         // :714-719 delete + recreate
    if ((rc = set_particles(e, v.data(), v.size()))) return rc;
    return ensure_grid_buffers(e);
}

int sph_set_params(SphEngine* e, const SphParams* params) {
    if (!e || !params) return fail(SPH_ERR_ARG, "null argument");
    int rc;
    if ((rc = validate_params(*params))) return rc;
    e->params = *params;
    return SPH_OK;
}
int sph_get_params(const SphEngine* e, SphParams* out) {
    if (!e || !out) return fail(SPH_ERR_ARG, "null argument");
    *out = e->params;
    return SPH_OK;
}
int sph_set_option(SphEngine* e, int option, int value) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    switch (option) {
    case SPH_OPT_NEIGHBOR_KERNEL: if (value < 1 || value > 3) return fail(SPH_ERR_ARG, "SPH pass %d: 3 = k_sph_walk, 2 = k_sph_list, 1 = k_sph_slow (0, the round-1 LDS tile pass, and 4, round 4's k_sph_tile, were retired)", value); e->optNeighbor = value; break;
    case SPH_OPT_GRID_BUILD: if (value < 0 || value > 1) return fail(SPH_ERR_ARG, "bad value"); e->optGridBuild = value; break;
    case SPH_OPT_AOS_MODE: if (value < 0 || value > 1) return fail(SPH_ERR_ARG, "bad value"); e->optAos = value; break;
    case SPH_OPT_TIMING: if (value < 0 || value > 2) return fail(SPH_ERR_ARG, "bad value"); e->optTiming = value; break;
    case SPH_OPT_GRAPH: if (value < 0 || value > 1) return fail(SPH_ERR_ARG, "bad value"); e->optGraph = value; break;
    case SPH_OPT_DEBUG:
        e->debugFlags = value;
        if ((value & 8) && !e->d_stats) {
            int rc;
            if ((rc = dev_alloc(&e->d_stats, 8))) return rc;
            HIP_TRY(hipMemsetAsync(e->d_stats, 0, 8 * sizeof(unsigned long long), e->stream));
        }
        break;
    default: return fail(SPH_ERR_ARG, "unknown option %d", option);
    }
    return SPH_OK;
}
int sph_get_option(const SphEngine* e, int option, int* value) {
    if (!e || !value) return fail(SPH_ERR_ARG, "null argument");
    switch (option) {
    case SPH_OPT_NEIGHBOR_KERNEL: *value = e->optNeighbor; break;
    case SPH_OPT_GRID_BUILD: *value = e->optGridBuild; break;
    case SPH_OPT_AOS_MODE: *value = e->optAos; break;
    case SPH_OPT_TIMING: *value = e->optTiming; break;
    case SPH_OPT_GRAPH: *value = e->optGraph; break;
    case SPH_OPT_GRAPH_LAUNCHES: *value = (int)e->graphLaunches; break;
    case SPH_OPT_DEBUG: *value = e->debugFlags; break;
    default: return fail(SPH_ERR_ARG, "unknown option %d", option);
    }
    return SPH_OK;
}

int sph_dispatch(SphEngine* e, float overrideDt) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    return dispatch_one(e, overrideDt);
}
// Key material of one sph_dispatch_n call for the graph cache: everything a captured launch sequence bakes in
// (uniforms, options, every device buffer a kernel gets, host-side validity state).  A hit compares the whole material,
// not only its hash.
static std::vector<unsigned char> graph_material(const SphEngine* e, float dt, int n) {
    std::vector<unsigned char> m;
    auto add = [&](const void* p, size_t len) { const unsigned char* b = static_cast<const unsigned char*>(p); m.insert(m.end(), b, b + len); };
    add(&e->params, sizeof(e->params));
    add(&dt, sizeof(dt)); add(&n, sizeof(n));
    // (whether the 80-byte array is current on entry only shapes the launches of the eager mode: fused update or write-back)
    const int opts[8] = {e->optNeighbor, e->optGridBuild, e->optAos, e->cur, ((e->aosValid && e->optAos == 0) ? 1 : 0) | (e->accValid ? 2 : 0) | (e->internalValid ? 4 : 0),
                         (int)e->idBase, e->allocatedCells, 0};
    add(opts, sizeof(opts));
    const void* ptrs[20] = {e->d_aos, e->d_pos[0], e->d_pos[1], e->d_vel[0], e->d_vel[1], e->d_rp[0], e->d_rp[1], e->d_foam[0], e->d_foam[1], e->d_acc,
                            e->d_binKey, e->d_binKey, e->d_order, e->d_tmp, e->d_cellCount, e->d_cellStart, e->d_blockSums, e->d_sPV, e->d_sPV, e->d_sOwn};
    add(ptrs, sizeof(ptrs));
    const void* more[3] = {e->d_llNext, e->d_shapeTab, e->d_stats};
    add(more, sizeof(more));
    const size_t sz[2] = {e->n, e->cap};
    add(sz, sizeof(sz));
    return m;
}
static uint64_t graph_hash(const std::vector<unsigned char>& m) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char b : m) { h ^= b; h *= 1099511628211ull; }
    return h ? h : 1;
}

int sph_dispatch_n(SphEngine* e, float overrideDt, int nSubsteps) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    // Scene0p runs up to 16 substeps per frame with unchanged members (Scene0p.cpp:1482-1494, :3720-3739);
    // at the default 50 000 particles that loop is launch-bound, so the second identical call is
    // captured into a hipGraph and replayed from then on.
    const bool graphable = e->optGraph && nSubsteps >= 2 && !e->slab && !e->optTiming && !e->debugFlags &&
                           !e->fountain.fountainMode && !e->river.riverMode && !e->params.param_pause && e->n > 0;
    SphEngine::GraphEntry* hit = nullptr;
    uint64_t key = 0;
    std::vector<unsigned char> material;
    if (graphable) {
        material = graph_material(e, overrideDt, nSubsteps);
        key = graph_hash(material);
        for (auto& g : e->graphs) if (g.key == key && g.material == material) { hit = &g; break; }
        if (hit && hit->exec) {
            sph::compute_grid_extents(e->params, e->grid);       // what an eager dispatch would have refreshed (sph_grid_info, RefreshGrid)
            HIP_TRY(hipGraphLaunch(hit->exec, e->stream));
            e->cur = hit->postCur; e->aosValid = hit->postAos; e->accValid = hit->postAcc; e->internalValid = true;
            hit->lastUse = ++e->graphClock;
            ++e->graphLaunches;
            return SPH_OK;
        }
    }
    const bool capture = graphable && hit;                   // seen once, run eagerly then: every buffer exists
    if (capture) HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    int rc = SPH_OK;
    for (int i = 0; i < nSubsteps && !rc; ++i) rc = dispatch_one(e, overrideDt);
    if (capture) {
        hipGraph_t graph = nullptr;
        hipError_t er = hipStreamEndCapture(e->stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (er != hipSuccess) return fail(SPH_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(er));
        er = hipGraphInstantiate(&hit->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (er != hipSuccess) { hit->exec = nullptr; return fail(SPH_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(er)); }
        hit->postCur = e->cur; hit->postAos = e->aosValid; hit->postAcc = e->accValid;
        hit->lastUse = ++e->graphClock;
        ++e->graphCaptures;
        HIP_TRY(hipGraphLaunch(hit->exec, e->stream));
        ++e->graphLaunches;
        return SPH_OK;
    }
    if (rc) return rc;
    if (graphable) {                                         // first sighting: remember it (at most 8 entries, LRU)
        if (e->graphs.size() >= 8) {
            size_t v = 0;
            for (size_t i = 1; i < e->graphs.size(); ++i) if (e->graphs[i].lastUse < e->graphs[v].lastUse) v = i;
            if (e->graphs[v].exec) {
                HIP_TRY(hipStreamSynchronize(e->stream));
                (void)hipGraphExecDestroy(e->graphs[v].exec);
            }
            e->graphs.erase(e->graphs.begin() + (long)v);
        }
        SphEngine::GraphEntry g;
        g.key = key; g.material = material; g.lastUse = ++e->graphClock;
        e->graphs.push_back(g);
    }
    return SPH_OK;
}

int sph_apply_wave_impulse(SphEngine* e, float amplitude, float wavelength, float phase, const float dir[3], float yMin, float yMax) {
    if (!e || !dir) return fail(SPH_ERR_ARG, "null argument");
    if (amplitude == 0.0f || wavelength <= 1e-6f) return SPH_OK;      // SPHFluid3D.cpp:607
    int rc;
    if ((rc = import_state(e))) return rc;
    WaveK w;
    const float len = std::sqrt(std::fma(dir[2], dir[2], std::fma(dir[1], dir[1], dir[0] * dir[0])));
    if (len > 1e-6f) { w.ndx = dir[0] / len; w.ndy = dir[1] / len; w.ndz = dir[2] / len; }   // WaveImpulse.comp:39
    else { w.ndx = 0.0f; w.ndy = 1.0f; w.ndz = 0.0f; }
    w.amplitude = amplitude; w.kk = 6.28318530718f / wavelength; w.phase = phase; w.yMin = yMin; w.yMax = yMax;
    const size_t nw = e->slab ? e->nSlots : e->n;
    // z-slabs: a kick of A changes a particle's step by at most A dt, i.e. a face layer's record count by about A dt / h per substep: a gentle wave (the
    // scene's A = 1.5: 0.5 % of h per substep) leaves a calm face calm; anything that could outgrow the messages' margin within two exchanges holds them whole
    // (the dt that is actually stepped: the last dispatch's overrideDt if there was one)
    if (e->slab && std::fabs(amplitude) * (e->lastDt > 0.0f ? e->lastDt : e->params.param_timeStep) > 0.02f * e->params.param_h) slab_hold(e);
    if (nw) {
        Timed t(e, SPH_K_IMPULSE);
        hipLaunchKernelGGL(k_wave_impulse, dim3(blocks_for(nw)), dim3(kBlock), 0, e->stream, w, e->d_pos[e->cur], e->d_vel[e->cur],
                           (e->aosValid && !e->slab) ? e->d_aos : nullptr, e->idBase, (int)nw);
    }
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}

// ---- per-frame impulses beyond WaveImpulse ------------------------

int sph_apply_vortex_impulse(SphEngine* e, float tangentKick, float inwardKick) {     // SPHFluid3D.cpp:627-646
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (std::fabs(tangentKick) < 1e-6f && std::fabs(inwardKick) < 1e-6f) return SPH_OK;
    float R[9], half[3];
    sph::rotation_mat3(e->params.param_boxEulerDeg, R);
    sph::effective_half(e->params, half);
    VortexK kk;
    kk.cx = e->params.param_boxCenter[0]; kk.cy = e->params.param_boxCenter[1]; kk.cz = e->params.param_boxCenter[2];
    kk.ax = R[3]; kk.ay = R[4]; kk.az = R[5];                                         // local +Y = column 1
    kk.tangent = tangentKick; kk.inward = inwardKick;
    kk.e1 = 0.35f * std::fmax(std::fmax(half[0], half[2]), 1e-4f);
    return launch_impulse(e, kk);
}

int sph_apply_attractor_impulse(SphEngine* e, const float point[3], float pullKick, float radius) {   // SPHFluid3D.cpp:650-664
    if (!e || !point) return fail(SPH_ERR_ARG, "null argument");
    if (std::fabs(pullKick) < 1e-6f) return SPH_OK;
    AttractorK kk;
    kk.px = point[0]; kk.py = point[1]; kk.pz = point[2];
    kk.radius = std::fmax(radius, 0.1f);
    kk.soften = std::fmax(0.15f * radius, 0.2f);
    kk.pullSoft = pullKick * kk.soften;
    kk.e0 = 0.6f * kk.radius;
    return launch_impulse(e, kk);
}

int sph_set_stencil_targets(SphEngine* e, const float* points4, size_t count) {       // SPHFluid3D.cpp:684-693
    if (!e || (!points4 && count)) return fail(SPH_ERR_ARG, "null argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    dev_free(e->d_stencil);
    e->stencilCount = 0;
    if (!count) return SPH_OK;
    int rc;
    if ((rc = dev_alloc(&e->d_stencil, count))) return rc;
    HIP_TRY(hipMemcpy(e->d_stencil, points4, count * sizeof(float4), hipMemcpyHostToDevice));
    e->stencilCount = count;
    return SPH_OK;
}

int sph_apply_stencil_attract(SphEngine* e, float pullKick, float dampKick) {         // SPHFluid3D.cpp:695-710
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (e->stencilCount == 0 || !e->d_stencil) return SPH_OK;
    if (std::fabs(pullKick) < 1e-6f && dampKick < 1e-6f) return SPH_OK;
    StencilK kk;
    kk.targets = e->d_stencil; kk.nTargets = (uint32_t)e->stencilCount;
    kk.pull = pullKick; kk.oneMinusDamp = 1.0f - std::fmin(dampKick, 0.5f);
    return launch_impulse(e, kk);
}

int sph_apply_curl_flow(SphEngine* e, float kick, float scale, float time) {          // SPHFluid3D.cpp:668-681
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (std::fabs(kick) < 1e-6f) return SPH_OK;
    CurlK kk;
    kk.kick = kick; kk.scale = std::fmax(scale, 1e-3f); kk.time = time;
    return launch_impulse(e, kk);
}

void sph_fountain_default(SphFountain* out) {               // SPHFluid3D.h:161-168
    if (!out) return;
    *out = SphFountain{0, {0.0f, -5.0f, 0.0f}, 1.0f, 0.25f, 25.0f, 1.0f, 2.0f, 0u};
}
int sph_set_fountain(SphEngine* e, const SphFountain* f) {
    if (!e || !f) return fail(SPH_ERR_ARG, "null argument");
    e->fountain = *f;
    return SPH_OK;
}
int sph_get_fountain(const SphEngine* e, SphFountain* out) {
    if (!e || !out) return fail(SPH_ERR_ARG, "null argument");
    *out = e->fountain;
    return SPH_OK;
}

void sph_river_default(SphRiver* out) {                     // SPHFluid3D.h:171-196
    if (out) sph::river_default(*out);
}
static int validate_river(const SphRiver& r) {
    if (r.terrainW < 2 || r.terrainH < 2 || r.terrainW > 4096 || r.terrainH > 4096) return fail(SPH_ERR_ARG, "terrainW / terrainH %d x %d: 2..4096 each", r.terrainW, r.terrainH);
    return SPH_OK;
}
int sph_generate_river_terrain(SphParams* params, int seed, SphRiver* river, float* heights) {
    if (!params || !river || !heights) return fail(SPH_ERR_ARG, "null argument");
    int rc;
    if ((rc = validate_river(*river))) return rc;
    sph::river_terrain(*params, seed, *river, heights);
    return SPH_OK;
}
int sph_spawn_river_particles(const SphParams* params, const SphRiver* river, const float* heights, size_t nRequested, uint32_t seed,
                              SphParticle* out, size_t* nOut, float* massOut) {
    if (!params || !river || !heights || !out || !nOut || !massOut) return fail(SPH_ERR_ARG, "null argument");
    int rc;
    if ((rc = validate_river(*river))) return rc;
    std::vector<SphParticle> v;
    float m;
    sph::spawn_river_particles(*params, *river, heights, nRequested, seed, v, m);
    std::memcpy(out, v.data(), v.size() * sizeof(SphParticle));
    *nOut = v.size();
    *massOut = m;
    return SPH_OK;
}
int sph_set_river(SphEngine* e, const SphRiver* river, const float* heights) {
    if (!e || !river) return fail(SPH_ERR_ARG, "null argument");
    int rc;
    if ((rc = validate_river(*river))) return rc;
    const size_t cells = (size_t)river->terrainW * (size_t)river->terrainH;
    if (heights) {                                                   // :868-873
        for (size_t i = 0; i < cells; ++i) if (!std::isfinite(heights[i])) return fail(SPH_ERR_ARG, "terrain height %zu is not finite", i);
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (cells > e->terrainCap) {
            dev_free(e->d_terrain);
            e->terrainCap = 0;
            if ((rc = dev_alloc(&e->d_terrain, cells))) return rc;
            e->terrainCap = cells;
        }
        e->terrainHeights.assign(heights, heights + cells);
        HIP_TRY(hipMemcpy(e->d_terrain, heights, cells * sizeof(float), hipMemcpyHostToDevice));
    } else if (!e->terrainHeights.empty() && e->terrainHeights.size() != cells) {
        return fail(SPH_ERR_ARG, "terrainW x terrainH changed to %d x %d without a new heightfield", river->terrainW, river->terrainH);
    }
    e->river = *river;
    return SPH_OK;
}
int sph_get_river(const SphEngine* e, SphRiver* out) {
    if (!e || !out) return fail(SPH_ERR_ARG, "null argument");
    *out = e->river;
    return SPH_OK;
}

size_t sph_num_particles(const SphEngine* e) { return e ? e->n : 0; }

int sph_grid_info(const SphEngine* e, SphGridInfo* out) {
    if (!e || !out) return fail(SPH_ERR_ARG, "null argument");
    *out = e->grid;
    return SPH_OK;
}

int sph_upload_particles(SphEngine* e, const SphParticle* host, size_t n) {
    if (!e || (!host && n)) return fail(SPH_ERR_ARG, "null argument");
    if (n != e->n) return fail(SPH_ERR_ARG, "upload of %zu records into an engine of %zu particles", n, e->n);
    if (n) HIP_TRY(hipMemcpyAsync(e->d_aos, host, n * sizeof(SphParticle), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->aosValid = true; e->internalValid = false; e->accValid = false;
    return SPH_OK;
}

int sph_download_particles(SphEngine* e, SphParticle* host, size_t n) {
    if (!e || (!host && n)) return fail(SPH_ERR_ARG, "null argument");
    if (n != e->n) return fail(SPH_ERR_ARG, "download of %zu records from an engine of %zu particles", n, e->n);
    int rc;
    if ((rc = writeback(e))) return rc;
    if (n) HIP_TRY(hipMemcpyAsync(host, e->d_aos, n * sizeof(SphParticle), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SPH_OK;
}

int sph_device_particles(SphEngine* e, const SphParticle** devPtr) {
    if (!e || !devPtr) return fail(SPH_ERR_ARG, "null argument");
    int rc;
    if ((rc = writeback(e))) return rc;
    *devPtr = e->d_aos;
    return SPH_OK;
}

int sph_pack_render_buffer(SphEngine* e, float* devOut4, size_t n, int wMode) {
    if (!e || (!devOut4 && n)) return fail(SPH_ERR_ARG, "null argument");
    if (e->slab) return fail(SPH_ERR_STATE, "a slab engine has no local 80-byte array");
    if (n != e->n) return fail(SPH_ERR_ARG, "size mismatch: %zu particles, engine has %zu", n, e->n);
    if (wMode < 0 || wMode > 4) return fail(SPH_ERR_ARG, "bad w mode %d", wMode);
    int rc;
    if ((rc = writeback(e))) return rc;
    if (n) {
        Timed t(e, SPH_K_OTHER);
        hipLaunchKernelGGL(k_pack_render, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, e->d_aos, reinterpret_cast<float4*>(devOut4), wMode, (int)n);
    }
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}

int sph_initial_particles(const SphEngine* e, SphParticle* host, size_t n) {
    if (!e || (!host && n)) return fail(SPH_ERR_ARG, "null argument");
    if (n != e->hostInit.size()) return fail(SPH_ERR_ARG, "size mismatch");
    std::memcpy(host, e->hostInit.data(), n * sizeof(SphParticle));
    return SPH_OK;
}

int sph_download_grid(SphEngine* e, int32_t* cellCount, size_t nCells, int32_t* particleCell, size_t n) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    int rc;
    sph::compute_grid_extents(e->params, e->grid);
    if ((rc = ensure_grid_buffers(e))) return rc;
    if (nCells != (size_t)e->grid.numCells || n != e->n) return fail(SPH_ERR_ARG, "size mismatch (cells %d, particles %zu)", e->grid.numCells, e->n);
    SimK k;
    make_simk(e->params, e->grid, e->params.param_timeStep, k);
    if ((rc = import_state(e))) return rc;
    if ((rc = build_grid(e, k))) return rc;
    const size_t need = std::max(nCells, n);
    if (need > e->dbgCap) { dev_free(e->d_dbg); if ((rc = dev_alloc(&e->d_dbg, need))) return rc; e->dbgCap = need; }
    if (cellCount) {
        hipLaunchKernelGGL(k_debug_cells, dim3(blocks_for(nCells)), dim3(kBlock), 0, e->stream, e->d_cellStart, e->d_dbg, (int)nCells);
        HIP_TRY(hipMemcpyAsync(cellCount, e->d_dbg, nCells * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    if (particleCell && n) {
        hipLaunchKernelGGL(k_debug_particle_cell, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, k, e->d_pos[e->cur], e->d_vel[e->cur], e->d_dbg, e->idBase, (int)n);
        HIP_TRY(hipMemcpyAsync(particleCell, e->d_dbg, n * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}

int sph_debug_counters(SphEngine* e, uint64_t* out, int count, int reset) {
    if (!e || !out) return fail(SPH_ERR_ARG, "null argument");
    if (count > 8) count = 8;
    for (int i = 0; i < count; ++i) out[i] = 0;
    if (!e->d_stats) return SPH_OK;
    HIP_TRY(hipStreamSynchronize(e->stream));
    unsigned long long host[8];
    HIP_TRY(hipMemcpy(host, e->d_stats, sizeof(host), hipMemcpyDeviceToHost));
    for (int i = 0; i < count; ++i) out[i] = host[i];
    if (reset) HIP_TRY(hipMemset(e->d_stats, 0, sizeof(host)));
    return SPH_OK;
}

// ---- z-slab (multi-GPU) entry points -----------------------------------------------------------
int sph_create_slab(SphEngine** out, const SphParticle* particles, const uint32_t* ids, size_t n, const SphParams* params,
                    int z0, int z1, int hasLo, int hasHi, size_t capacity, void* stream) {
    if ((!particles || !ids) && n) return fail(SPH_ERR_ARG, "null particles / ids");
    if (z1 <= z0 || z0 < 0) return fail(SPH_ERR_ARG, "bad slab range [%d, %d)", z0, z1);
    if (capacity < n) return fail(SPH_ERR_ARG, "capacity %zu < %zu particles", capacity, n);
    if (hasLo && hasHi && z1 - z0 < 2)
        return fail(SPH_ERR_ARG, "an inner slab needs at least 2 cell layers (a migrant from below would have to become the upper neighbour's ghost within the same substep)");
    SphEngine* e = nullptr;
    int rc = create_common(out, params, stream, &e);
    if (rc) return rc;
    e->slab = true; e->z0 = z0; e->z1 = z1; e->hasLo = hasLo ? 1 : 0; e->hasHi = hasHi ? 1 : 0;
    sph::compute_grid_extents(e->params, e->grid);
    uint32_t* d_ids = nullptr;
    auto cleanup = [&](int code) { if (d_ids) (void)hipFree(d_ids); sph_destroy(e); return code; };
    if ((rc = alloc_particle_buffers(e, capacity))) return cleanup(rc);
    if ((rc = ensure_grid_buffers(e))) return cleanup(rc);
    e->n = n; e->nSlots = n;
    if (n) {
        if ((rc = dev_alloc(&d_ids, n))) return cleanup(rc);
        hipError_t er = hipMemcpyAsync(e->d_aos, particles, n * sizeof(SphParticle), hipMemcpyHostToDevice, e->stream);
        if (er == hipSuccess) er = hipMemcpyAsync(d_ids, ids, n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream);
        if (er != hipSuccess) return cleanup(fail(SPH_ERR_HIP, "upload failed: %s", hipGetErrorString(er)));
        hipLaunchKernelGGL(k_slab_import, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, e->d_aos, d_ids, e->d_pos[0], e->d_vel[0], e->d_rp[0], e->d_foam[0], (int)n);
    }
    const uint32_t init[16] = {0u, 0u, (uint32_t)n, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    hipError_t er = hipMemcpyAsync(e->d_slabCnt, init, sizeof(init), hipMemcpyHostToDevice, e->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
    if (er != hipSuccess) return cleanup(fail(SPH_ERR_HIP, "slab init failed: %s", hipGetErrorString(er)));
    if (d_ids) (void)hipFree(d_ids);
    e->cur = 0; e->internalValid = true; e->aosValid = false; e->accValid = false;
    *out = e;
    return SPH_OK;
}

int sph_slab_pack(SphEngine* e, void* sendLo, void* sendHi, uint32_t capLo, uint32_t capHi, uint32_t countsOut[2]) {
    if (!e || !countsOut) return fail(SPH_ERR_ARG, "null argument");
    if (!e->slab) return fail(SPH_ERR_STATE, "not a slab engine");
    if ((e->hasLo && !sendLo) || (e->hasHi && !sendHi)) return fail(SPH_ERR_ARG, "missing send buffer");
    sph::compute_grid_extents(e->params, e->grid);
    SimK k;
    make_simk(e->params, e->grid, e->params.param_timeStep, k);
    HIP_TRY(hipMemsetAsync(e->d_slabCnt, 0, 2 * sizeof(uint32_t), e->stream));
    HIP_TRY(hipMemsetAsync(e->d_slabCnt + 5, 0, 2 * sizeof(uint32_t), e->stream));      // counts of an earlier async pack
    uint32_t host[3] = {0, 0, 0};
    // e->nSlots bounds the slots in use; the device-side live count (slabCnt[2]) trims it to the
    // slots that hold data, so no host round trip is needed before the launch
    if (e->nSlots) {
        Timed t(e, SPH_K_OTHER);
        hipLaunchKernelGGL((k_slab_pack<false>), dim3(blocks_for(e->nSlots)), dim3(kBlock), 0, e->stream, k, e->z0, e->z1, e->hasLo, e->hasHi,
                           e->d_pos[e->cur], e->d_vel[e->cur], e->d_rp[e->cur], e->d_foam[e->cur], e->accValid ? e->d_acc : (const float4*)nullptr, (int)e->nSlots,
                           (SlabRec*)sendLo, (SlabRec*)sendHi, capLo, capHi, e->d_slabCnt,
                           slab_ranges_usable(e) ? e->d_cellStart : (const uint32_t*)nullptr, k.gx * k.gy);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, e->d_slabCnt, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->nSlots = std::min<size_t>(e->nSlots, host[2]);
    // sph_slab_pack_async / sph_slab_exchange expect zeros in [0..1]; the counts of this pack stay visible to sph_slab_status in [5..6]
    HIP_TRY(hipMemsetAsync(e->d_slabCnt, 0, 2 * sizeof(uint32_t), e->stream));
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, e->stream, e->d_slabCnt + 5, host[0]);
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, e->stream, e->d_slabCnt + 6, host[1]);
    HIP_TRY(hipGetLastError());
    if (host[0] > capLo || host[1] > capHi) return fail(SPH_ERR_CAPACITY, "halo send buffer too small (%u/%u lo, %u/%u hi)", host[0], capLo, host[1], capHi);
    countsOut[0] = host[0]; countsOut[1] = host[1];
    return SPH_OK;
}

int sph_slab_unpack(SphEngine* e, const void* recvLo, uint32_t nLo, const void* recvHi, uint32_t nHi) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (!e->slab) return fail(SPH_ERR_STATE, "not a slab engine");
    if ((nLo && !recvLo) || (nHi && !recvHi)) return fail(SPH_ERR_ARG, "missing receive buffer");
    if (e->nSlots + nLo + nHi > e->cap) return fail(SPH_ERR_CAPACITY, "slab capacity %zu < %zu slots", e->cap, e->nSlots + nLo + nHi);
    const int c = e->cur;
    const SlabGeom geom = slab_geom(e);
    if (nLo) hipLaunchKernelGGL(k_slab_unpack, dim3(blocks_for(nLo)), dim3(kBlock), 0, e->stream, (const SlabRec*)recvLo, (int)nLo,
                                e->d_pos[c], e->d_vel[c], e->d_rp[c], e->d_foam[c], e->d_acc, (int)e->nSlots, geom, 1, e->d_slabCnt);
    e->nSlots += nLo;
    if (nHi) hipLaunchKernelGGL(k_slab_unpack, dim3(blocks_for(nHi)), dim3(kBlock), 0, e->stream, (const SlabRec*)recvHi, (int)nHi,
                                e->d_pos[c], e->d_vel[c], e->d_rp[c], e->d_foam[c], e->d_acc, (int)e->nSlots, geom, 0, e->d_slabCnt);
    e->nSlots += nHi;
    HIP_TRY(hipGetLastError());
    // until the next dispatch sorts again, every slot may hold data
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, e->stream, e->d_slabCnt + 2, (uint32_t)e->nSlots);
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}

int sph_slab_download(SphEngine* e, void* hostOut, size_t capRecords, size_t* nOut) {
    if (!e || !nOut || (!hostOut && capRecords)) return fail(SPH_ERR_ARG, "null argument");
    if (!e->slab) return fail(SPH_ERR_STATE, "not a slab engine");
    uint32_t cnts[8] = {0};
    HIP_TRY(hipMemcpyAsync(cnts, e->d_slabCnt, sizeof(cnts), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->xstream) HIP_TRY(hipStreamSynchronize(e->xstream));
    // a face or slot overflow dropped records: the owned set would come back short without a word.  (Flag 16 -- a particle crossed
    // more layers than the exchange follows -- drops nothing: the records are delivered, sph_slab_status carries the notice.)
    int frc = slab_flags_error(e, cnts[4], cnts[5], cnts[6]);
    if (frc) return frc;
    const size_t n = cnts[2];
    SlabOut* d_out = nullptr;
    int rc;
    if ((rc = dev_alloc(&d_out, n ? n : 1))) return rc;
    hipError_t er = hipMemsetAsync(e->d_slabCnt + 3, 0, sizeof(uint32_t), e->stream);
    if (er == hipSuccess && n)
        hipLaunchKernelGGL(k_slab_download, dim3(blocks_for(n)), dim3(kBlock), 0, e->stream, e->d_pos[e->cur], e->d_vel[e->cur], e->d_rp[e->cur],
                           e->d_foam[e->cur], e->d_acc, (int)n, e->accValid ? 1 : 0, d_out, (uint32_t)n, e->d_slabCnt + 3);
    uint32_t cnt = 0;
    if (er == hipSuccess) er = hipMemcpyAsync(&cnt, e->d_slabCnt + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
    if (er == hipSuccess && cnt > capRecords) { (void)hipFree(d_out); return fail(SPH_ERR_CAPACITY, "%u owned records > capacity %zu", cnt, capRecords); }
    if (er == hipSuccess && cnt) er = hipMemcpy(hostOut, d_out, (size_t)cnt * sizeof(SlabOut), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (er != hipSuccess) return fail(SPH_ERR_HIP, "slab download failed: %s", hipGetErrorString(er));
    *nOut = cnt;
    return SPH_OK;
}

// ---- exchange without host round trips (device-side counts) + RCCL transport -------------------------------------
int sph_slab_alloc_faces(SphEngine* e, uint32_t faceCap) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (!e->slab) return fail(SPH_ERR_STATE, "not a slab engine");
    if (faceCap == 0) return fail(SPH_ERR_ARG, "face capacity must be > 0");
    if (e->faceCap == faceCap && e->d_face[0]) return SPH_OK;
    HIP_TRY(hipStreamSynchronize(e->stream));
    int rc;
    for (int i = 0; i < 4; ++i) {
        dev_free(e->d_face[i]);
        if ((rc = dev_alloc(&e->d_face[i], slab_face_bytes(faceCap)))) return rc;
        HIP_TRY(hipMemsetAsync(e->d_face[i], 0, sizeof(SlabHdr), e->stream));      // no header yet
    }
    if (!e->h_cnt) {
        HIP_TRY(hipHostMalloc((void**)&e->h_cnt, 4 * 8 * sizeof(uint32_t), hipHostMallocDefault));
        for (auto& ev : e->evCnt) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    for (auto& v : e->cntValid) v = false;
    e->faceCap = faceCap;
    e->faceAgreedWith = nullptr;
    return SPH_OK;
}
int sph_slab_face_buffer(SphEngine* e, int which, void** devPtr) {
    if (!e || !devPtr) return fail(SPH_ERR_ARG, "null argument");
    if (which < 0 || which > 3 || !e->d_face[which]) return fail(SPH_ERR_STATE, "no face buffers: call sph_slab_alloc_faces first");
    *devPtr = e->d_face[which];
    return SPH_OK;
}
int sph_slab_face_bytes(SphEngine* e, uint64_t* bytes) {
    if (!e || !bytes) return fail(SPH_ERR_ARG, "null argument");
    if (!e->faceCap) return fail(SPH_ERR_STATE, "no face buffers: call sph_slab_alloc_faces first");
    *bytes = (uint64_t)slab_face_bytes(e->faceCap);
    return SPH_OK;
}
static void grid_key(const SphGridInfo& g, float out[8]) {
    const float v[8] = {g.gridMin[0], g.gridMin[1], g.gridMin[2], g.cellSize, (float)g.dims[0], (float)g.dims[1], (float)g.dims[2], 0.0f};
    std::memcpy(out, v, sizeof(v));
}
static int slab_pack_on(SphEngine* e, hipStream_t st) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (!e->slab || !e->d_face[0]) return fail(SPH_ERR_STATE, "slab engine with face buffers required");
    sph::compute_grid_extents(e->params, e->grid);
    SimK k;
    make_simk(e->params, e->grid, e->params.param_timeStep, k);
    grid_key(e->grid, e->packGrid);                         // the records this pack cuts are classified for THIS grid (sph_slab_step_begin checks it)
    e->packGridValid = true;
    e->nSlots = e->cap;                                     // from now on only a launch bound: slabCnt[2] counts the slots in use
    {                                                       // (slabCnt[8..11] are zero: set at creation, reset by k_slab_headers2)
        Timed t(e, SPH_K_OTHER, st);
        hipLaunchKernelGGL((k_slab_pack<true>), dim3(blocks_for(e->cap)), dim3(kBlock), 0, st, k, e->z0, e->z1, e->hasLo, e->hasHi,
                           e->d_pos[e->cur], e->d_vel[e->cur], e->d_rp[e->cur], e->d_foam[e->cur], e->accValid ? e->d_acc : (const float4*)nullptr, (int)e->cap,
                           (SlabRec*)e->d_face[0], (SlabRec*)e->d_face[1], e->faceCap, e->faceCap, e->d_slabCnt,
                           slab_ranges_usable(e) ? e->d_cellStart : (const uint32_t*)nullptr, k.gx * k.gy);
        hipLaunchKernelGGL(k_slab_headers2, dim3(1), dim3(1), 0, st, e->d_slabCnt, e->hasLo ? e->d_face[0] : (char*)nullptr,
                           e->hasHi ? e->d_face[1] : (char*)nullptr, e->faceCap, e->msgSend[0], e->msgSend[2], e->msgSend[1], e->msgSend[3]);
    }
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}
// msg[4]: records the messages carried (halo lo / hi, migrants lo / hi); the face capacity when whole faces were moved
static int slab_unpack_on(SphEngine* e, hipStream_t st, const void* recvLo, const void* recvHi, uint32_t recvCap, const uint32_t msg[4]) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (!e->slab) return fail(SPH_ERR_STATE, "not a slab engine");
    if (recvCap != e->faceCap || !e->faceCap) return fail(SPH_ERR_ARG, "received faces must have this engine's face capacity (%u, got %u)", e->faceCap, recvCap);
    const char* lo = e->hasLo ? (const char*)recvLo : nullptr;
    const char* hi = e->hasHi ? (const char*)recvHi : nullptr;
    if ((e->hasLo && !lo) || (e->hasHi && !hi)) return fail(SPH_ERR_ARG, "missing receive buffer");
    const int c = e->cur;
    e->nSlots = e->cap;
    const SlabGeom geom = slab_geom(e);
    Timed t(e, SPH_K_OTHER, st);
    if (lo) hipLaunchKernelGGL(k_slab_unpack2, dim3(blocks_for((size_t)msg[0] + msg[2])), dim3(kBlock), 0, st, lo, (const char*)nullptr, recvCap, msg[0], msg[2], 0u, 0u,
                               e->d_pos[c], e->d_vel[c], e->d_rp[c], e->d_foam[c], e->d_acc, e->d_slabCnt, (uint32_t)e->cap, geom, 1);
    if (hi) hipLaunchKernelGGL(k_slab_unpack2, dim3(blocks_for((size_t)msg[1] + msg[3])), dim3(kBlock), 0, st, hi, lo, recvCap, msg[1], msg[3], msg[0], msg[2],
                               e->d_pos[c], e->d_vel[c], e->d_rp[c], e->d_foam[c], e->d_acc, e->d_slabCnt, (uint32_t)e->cap, geom, 0);
    hipLaunchKernelGGL(k_slab_commit2, dim3(1), dim3(1), 0, st, e->d_slabCnt, lo, hi, recvCap, (uint32_t)e->cap, msg[0], msg[2], msg[1], msg[3]);
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}
int sph_slab_pack_async(SphEngine* e) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (e->slab) slab_hold(e);                              // an exchange outside the sized protocol (priming, re-priming after a grid change): no counts are noted for it
    for (int i = 0; i < 4; ++i) e->msgSend[i] = e->msgRecv[i] = e->faceCap;   // whole faces (the headers say so)
    e->intentValid = false;
    return slab_pack_on(e, e->stream);
}
int sph_slab_unpack_async(SphEngine* e, const void* recvLo, const void* recvHi, uint32_t recvCap) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    const uint32_t whole[4] = {e->faceCap, e->faceCap, e->faceCap, e->faceCap};
    return slab_unpack_on(e, e->stream, recvLo, recvHi, recvCap, whole);
}

// ---- message sizes.  A message carries whole capacity unless the face has been CALM: the counts of the last two known exchanges (e - 3, e - 2)
// within 3 % + 64 records of each other; then it carries the count of e - 2 + a quarter + 1024.  (A face whose count grows faster than that while it
// was calm two exchanges ago gets cut off: error flag 8 on the receiver, loud.  Random violent scenes showed that a margin alone is not enough:
// tools/fuzz_sweep.py, 2 of 240 slab runs with the plain "count + 25 % + 1024" rule of the first version.)
static uint32_t msg_records(uint32_t seen, uint32_t before, uint32_t cap, bool tight = false) {
    if (tight) return std::min(cap, seen);                                   // test hook (sph_slab_debug_tight_messages): no margin, no calm test
    const uint32_t hi = std::max(seen, before), lo = std::min(seen, before);
    if (hi - lo > hi / 32u + 64u) return cap;                                // not calm: the whole face
    return (uint32_t)std::min<uint64_t>(cap, (uint64_t)seen + seen / 4u + 1024u);
}
// host-only: the rule above as a function of (count two exchanges ago, count three exchanges ago, face capacity)
int sph_slab_message_records(uint32_t seen, uint32_t before, uint32_t cap) { return (int)msg_records(seen, before, cap); }

static uint32_t fnv32(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 16777619u; }
    return h;
}
constexpr uint32_t kIntentMagic = 0x504c414eu;             // "PLAN"
// THE PLAN of exchange e->exchangeNo: the records each of this engine's four messages will carry, the records it will post receives for, and the
// host-side state both derive from (exchange number, hold events, members, pause).  Everything a neighbour must agree with is in these 64 bytes:
// sph_slab_step_finish_local compares the neighbour ENGINES' plans, the RCCL transport sends the plan across each link as a fixed-size message and
// compares before a sized message is posted (slab_handshake_rccl).  The one place that reads the counts of two exchanges ago.
static int slab_plan(SphEngine* e) {
    // a member that moves the grid or the walls, noticed HERE, before the sizes (sph_slab_exchange sized its messages before anything had
    // noticed the edit; the pack that follows cuts the records for the NEW grid with a full-slot scan and may fill a face)
    {
        SphGridInfo g;
        sph::compute_grid_extents(e->params, g);
        float now[8], cont[15];
        grid_key(g, now);
        container_key(e->params, cont);
        const bool gridMoved = e->packGridValid && std::memcmp(now, e->packGrid, sizeof(now)) != 0;
        const bool wallsMoved = e->lastDt > 0.0f && std::memcmp(cont, e->lastContainer, sizeof(cont)) != 0;   // (before the first dispatch lastContainer is not a container yet)
        if (gridMoved || wallsMoved) slab_hold(e);
    }
    for (int i = 0; i < 4; ++i) e->msgSend[i] = e->msgRecv[i] = e->faceCap;
    const bool hold = e->calmHold > 0;                      // (host-side knowledge that every rank shares IF every rank makes the same calls: the plans' holdEvents say whether they did)
    if (hold) e->calmHold -= 1;
    if (e->exchangeNo >= 3 && !hold) {
        const int s2 = (int)((e->exchangeNo - 2u) & 3u), s3 = (int)((e->exchangeNo - 3u) & 3u);
        if (e->cntValid[s2] && e->cntValid[s3]) {
            HIP_TRY(hipEventSynchronize(e->evCnt[s2]));     // two exchanges back: long done unless the host is that far ahead of the device (the host's look-ahead is bounded by this wait)
            const uint32_t* c = e->h_cnt + 8 * s2;
            const uint32_t* b = e->h_cnt + 8 * s3;
            const bool tight = e->tightMessages != 0;
            for (int i = 0; i < 4; ++i) e->msgSend[i] = msg_records(c[i], b[i], e->faceCap, tight);
            e->msgRecv[0] = msg_records(c[4], b[4], e->faceCap, tight); e->msgRecv[2] = msg_records(c[5], b[5], e->faceCap, tight);   // from lo: its halo copies, its migrants
            e->msgRecv[1] = msg_records(c[6], b[6], e->faceCap, tight); e->msgRecv[3] = msg_records(c[7], b[7], e->faceCap, tight);   // from hi
        }
    }
    e->sentBytes[0] = e->hasLo ? sizeof(SlabHdr) + (uint64_t)e->msgSend[2] * 64u + (uint64_t)e->msgSend[0] * 40u : 0u;
    e->sentBytes[1] = e->hasHi ? sizeof(SlabHdr) + (uint64_t)e->msgSend[3] * 64u + (uint64_t)e->msgSend[1] * 40u : 0u;
    SphSlabIntent& I = e->intent;
    std::memset(&I, 0, sizeof(I));
    I.magic = kIntentMagic; I.exchangeNo = e->exchangeNo; I.stepNo = e->stepNo; I.faceCap = e->faceCap;
    for (int d = 0; d < 2; ++d) {
        const bool has = d ? e->hasHi : e->hasLo;
        I.sendHalo[d] = has ? e->msgSend[d] : 0u; I.sendMig[d] = has ? e->msgSend[2 + d] : 0u;
        I.recvHalo[d] = has ? e->msgRecv[d] : 0u; I.recvMig[d] = has ? e->msgRecv[2 + d] : 0u;
    }
    I.holdEvents = e->holdEvents;
    I.paramsHash = fnv32(&e->params, sizeof(e->params));
    I.flags = (e->params.param_pause ? 1u : 0u) | (e->tightMessages ? 2u : 0u) | (hold ? 4u : 0u);
    I.zRange = ((uint32_t)e->z0 & 0xffffu) | ((uint32_t)e->z1 << 16);
    e->intentValid = true;
    return SPH_OK;
}
// Does the plan `nb` of the neighbour on side d of `me` (0 = below, 1 = above) fit mine?  On a mismatch `why` says what differs.  Pure host logic: the
// same function judges a neighbour ENGINE's plan (one process, sph_slab_step_finish_local) and a plan received over RCCL.
static bool slab_intent_mismatch(const SphSlabIntent& me, const SphSlabIntent& nb, int d, char* why, size_t n) {
    const int o = 1 - d;                                    // the neighbour's side that faces me
    const char* side = d ? "upper" : "lower";
    if (nb.magic != kIntentMagic) { snprintf(why, n, "the %s neighbour sent no plan (magic %08x)", side, nb.magic); return true; }
    if (nb.exchangeNo != me.exchangeNo) { snprintf(why, n, "the %s neighbour is at exchange %u, this rank at %u (a rank skipped or repeated an exchange)", side, nb.exchangeNo, me.exchangeNo); return true; }
    if (nb.faceCap != me.faceCap) { snprintf(why, n, "face capacity %u here, %u on the %s neighbour", me.faceCap, nb.faceCap, side); return true; }
    const uint32_t nz0 = nb.zRange & 0xffffu, nz1 = nb.zRange >> 16, z0 = me.zRange & 0xffffu, z1 = me.zRange >> 16;
    if (d ? nz0 != z1 : nz1 != z0) { snprintf(why, n, "the %s neighbour owns layers [%u, %u), this rank [%u, %u): not adjacent", side, nz0, nz1, z0, z1); return true; }
    if (nb.holdEvents != me.holdEvents) {
        snprintf(why, n, "this rank has seen %u impulses / container edits / priming exchanges, the %s neighbour %u: one of them was issued on one rank only", me.holdEvents, side, nb.holdEvents);
        return true;
    }
    if (nb.paramsHash != me.paramsHash) { snprintf(why, n, "the members (SphParams) differ between this rank and the %s neighbour", side); return true; }
    if (nb.flags != me.flags) { snprintf(why, n, "pause / hold / message test hook differ (%u here, %u on the %s neighbour)", me.flags, nb.flags, side); return true; }
    if (nb.sendHalo[o] != me.recvHalo[d] || nb.sendMig[o] != me.recvMig[d]) {
        snprintf(why, n, "the %s neighbour will send %u halo copies + %u migrants, this rank expects %u + %u", side, nb.sendHalo[o], nb.sendMig[o], me.recvHalo[d], me.recvMig[d]);
        return true;
    }
    if (nb.recvHalo[o] != me.sendHalo[d] || nb.recvMig[o] != me.sendMig[d]) {
        snprintf(why, n, "this rank will send %u halo copies + %u migrants, the %s neighbour expects %u + %u", me.sendHalo[d], me.sendMig[d], side, nb.recvHalo[o], nb.recvMig[o]);
        return true;
    }
    return false;
}
// behind the transfer on `st`: this exchange's true counts (own: slabCnt[12..15]; the neighbours': their headers) on their way to the host
static int slab_note_counts(SphEngine* e, hipStream_t st) {
    const int slot = (int)(e->exchangeNo & 3u);
    uint32_t* c = e->h_cnt + 8 * slot;
    HIP_TRY(hipMemcpyAsync(c, e->d_slabCnt + 12, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    for (int d = 0; d < 2; ++d) {
        c[4 + 2 * d] = c[5 + 2 * d] = 0u;
        if (d ? e->hasHi : e->hasLo) {                      // SlabHdr words: [2] nHaloTrue, [4] nMigTrue
            HIP_TRY(hipMemcpyAsync(c + 4 + 2 * d, e->d_face[2 + d] + 2 * sizeof(uint32_t), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(c + 5 + 2 * d, e->d_face[2 + d] + 4 * sizeof(uint32_t), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        }
    }
    HIP_TRY(hipEventRecord(e->evCnt[slot], st));
    e->cntValid[slot] = true;
    e->exchangeNo += 1u;
    return SPH_OK;
}

// ---- boundary-first substep: the exchange of the next substep beside the interior of this one -------------------------
static int ensure_xstream(SphEngine* e) {
    if (e->xstream) return SPH_OK;
    {
        int least = 0, greatest = 0;                         // (numerically: greatest <= least)
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&e->xstream, hipStreamNonBlocking, greatest));   // pack / transfer / unpack: few small kernels that must not queue behind the interior's blocks
        HIP_TRY(hipStreamCreateWithPriority(&e->bstream, hipStreamNonBlocking, least));      // the interior of the SPH pass
        HIP_TRY(hipEventCreateWithFlags(&e->evSorted, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->evInterior, hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&e->evBoundary, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->evPacked, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&e->evDone));                    // (timed: sph_slab_step_times reads it as the end of the exchange)
    for (auto& ev : e->evX) HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(hipEventCreate(&e->evPassEnd));
    return SPH_OK;
}
int sph_slab_step_begin(SphEngine* e, float overrideDt) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (!e->slab || !e->d_face[0]) return fail(SPH_ERR_STATE, "slab engine with face buffers required");
    if (e->stepPending) return fail(SPH_ERR_STATE, "sph_slab_step_begin twice without sph_slab_step_finish / sph_slab_step_finish_local");
    int rc;
    if ((rc = ensure_xstream(e))) return rc;
    if (e->params.param_pause) {                             // SPHFluid3D.cpp:432: a paused DispatchCompute is a no-op, and so is its exchange --
        e->stepPending = true; e->stepPaused = true;         // the halo records in place stay valid (impulses act on the copies as on their owners)
        e->stepNo += 1u;
        return SPH_OK;
    }
    // The halo records in place were cut by the last pack for the grid of THAT moment; a member that moves the grid (box centre / half /
    // Euler angles, h, grid_cap) must not change between two steps: the records would be classified for the old grid without a word.
    if (e->packGridValid) {
        SphGridInfo g;
        sph::compute_grid_extents(e->params, g);
        float now[8];
        grid_key(g, now);
        if (std::memcmp(now, e->packGrid, sizeof(now)) != 0)
            return fail(SPH_ERR_STATE, "the grid changed since the halo records in place were cut (box centre / half / angles, h or grid_cap edited between two steps): "
                                       "prime again with sph_slab_exchange (or pack_async / unpack_async on every engine) before the next sph_slab_step_begin");
    }
    if (e->optTiming) HIP_TRY(hipEventRecord(e->evX[0], e->stream));
    e->stepNo += 1u;
    if ((rc = dispatch_one(e, overrideDt, true))) return rc;               // ... -> SPH (faces first, e->evBoundary, interior) on the engine's stream
    if ((rc = slab_plan(e))) return rc;                                      // the sizes of THIS step's exchange (the pack writes them into the headers)
    HIP_TRY(hipStreamWaitEvent(e->xstream, e->evBoundary, 0));
    for (hipEvent_t ev : e->peerDone)                                        // a neighbour engine of this process may still be copying the last send face
        if (ev) HIP_TRY(hipStreamWaitEvent(e->xstream, ev, 0));
    if (e->optTiming) HIP_TRY(hipEventRecord(e->evX[1], e->xstream));
    if ((rc = slab_pack_on(e, e->xstream))) return rc;
    HIP_TRY(hipEventRecord(e->evPacked, e->xstream));
    if (e->optTiming) HIP_TRY(hipEventRecord(e->evX[2], e->xstream));
    e->stepPending = true; e->stepPaused = false;
    return SPH_OK;
}
static int slab_step_join(SphEngine* e) {
    if (e->optTiming) HIP_TRY(hipEventRecord(e->evX[4], e->xstream));
    HIP_TRY(hipEventRecord(e->evDone, e->xstream));
    HIP_TRY(hipStreamWaitEvent(e->stream, e->evDone, 0));                    // the next substep's grid build sees the received records
    e->stepPending = false;
    return SPH_OK;
}
int sph_slab_step_finish_local(SphEngine* e, SphEngine* lo, SphEngine* hi) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (!e->stepPending) return fail(SPH_ERR_STATE, "sph_slab_step_finish_local without sph_slab_step_begin");
    if ((e->hasLo != 0) != (lo != nullptr) || (e->hasHi != 0) != (hi != nullptr)) return fail(SPH_ERR_ARG, "neighbour engines do not match the slab's faces");
    for (SphEngine* nb : {lo, hi}) {
        if (!nb) continue;
        if (!nb->evPacked || nb->stepNo != e->stepNo)       // (stepNo counts the begins: the neighbour's pack of THIS step is enqueued)
            return fail(SPH_ERR_STATE, "a neighbour engine has not begun THIS step (call every engine's sph_slab_step_begin before any sph_slab_step_finish_local)");
        if (nb->stepPaused != e->stepPaused) return fail(SPH_ERR_STATE, "param_pause differs between neighbouring slab engines");
        if (nb->faceCap != e->faceCap) return fail(SPH_ERR_ARG, "face capacities differ");
    }
    if (e->stepPaused) { e->stepPending = false; return SPH_OK; }   // (stepPaused keeps saying what this step was: the neighbours compare)
    int rc;
    // Both ends of a link must have planned the same messages (each derived its sizes on its own: the sender from its counts, the receiver
    // from the headers it got, both from their own exchange number and hold state).  Here the neighbour is an engine of this process, so its
    // plan is compared directly -- what sph_slab_step_finish does with the plans it receives over RCCL.  A call issued on one engine only
    // (an impulse, a member edit, a skipped step) ends HERE, with its name, not in a truncated copy.
    if (!e->intentValid) return fail(SPH_ERR_STATE, "no plan for this exchange (sph_slab_step_begin did not run to its end)");
    for (int d = 0; d < 2; ++d) {
        SphEngine* nb = d ? hi : lo;
        if (!nb) continue;
        char why[320];
        if (!nb->intentValid || slab_intent_mismatch(e->intent, nb->intent, d, why, sizeof(why))) {
            if (!nb->intentValid) snprintf(why, sizeof(why), "the %s neighbour engine has no plan for this exchange", d ? "upper" : "lower");
            e->stepPending = false;                          // (the step cannot be finished: destroy the group or prime it again)
            return fail(SPH_ERR_STATE, "slab exchange %u refused before any record moved: %s", e->intent.exchangeNo, why);
        }
    }
    // the neighbour's send face -> this engine's receive face: header + migrants in use, halo copies in use (what sph_slab_step_finish
    // sends with ncclSend / ncclRecv), behind the neighbour's pack
    for (int d = 0; d < 2; ++d) {
        SphEngine* nb = d ? hi : lo;
        if (!nb) continue;
        const char* src = nb->d_face[d ? 0 : 1];                             // its "send lo" is my "receive hi" and the other way round
        char* dst = e->d_face[2 + d];
        HIP_TRY(hipStreamWaitEvent(e->xstream, nb->evPacked, 0));
        HIP_TRY(hipMemcpyAsync(dst, src, sizeof(SlabHdr) + (size_t)e->msgRecv[2 + d] * 64u, hipMemcpyDeviceToDevice, e->xstream));
        HIP_TRY(hipMemcpyAsync(dst + slab_face_halo_off(e->faceCap), src + slab_face_halo_off(e->faceCap), (size_t)e->msgRecv[d] * 40u, hipMemcpyDeviceToDevice, e->xstream));
    }
    if (e->optTiming) HIP_TRY(hipEventRecord(e->evX[3], e->xstream));
    if ((rc = slab_unpack_on(e, e->xstream, e->d_face[2], e->d_face[3], e->faceCap, e->msgRecv))) return rc;
    if ((rc = slab_note_counts(e, e->xstream))) return rc;
    if ((rc = slab_step_join(e))) return rc;
    if (lo) lo->peerDone[1] = e->evDone;                                     // their next pack overwrites the face this engine has just read
    if (hi) hi->peerDone[0] = e->evDone;
    return SPH_OK;
}
int sph_slab_status(SphEngine* e, uint32_t out[5]) {
    if (!e || !out) return fail(SPH_ERR_ARG, "null argument");
    if (!e->slab) return fail(SPH_ERR_STATE, "not a slab engine");
    uint32_t host[8];
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->xstream) HIP_TRY(hipStreamSynchronize(e->xstream));
    HIP_TRY(hipMemcpyAsync(host, e->d_slabCnt, sizeof(host), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < 5; ++i) out[i] = host[i];
    out[0] += host[5]; out[1] += host[6];                   // the async pack keeps its counts there (the header kernels reset the running ones)
    return slab_flags_error(e, host[4], out[0], out[1]);    // (flag 16 alone is a notice, not an error: out[4] carries it)
}
int sph_slab_clear_flags(SphEngine* e, uint32_t mask) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (!e->slab) return fail(SPH_ERR_STATE, "not a slab engine");
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->xstream) HIP_TRY(hipStreamSynchronize(e->xstream));
    uint32_t f = 0;
    HIP_TRY(hipMemcpy(&f, e->d_slabCnt + 4, sizeof(f), hipMemcpyDeviceToHost));
    f &= ~mask;
    HIP_TRY(hipMemcpy(e->d_slabCnt + 4, &f, sizeof(f), hipMemcpyHostToDevice));
    return SPH_OK;
}
int sph_slab_message_bytes(SphEngine* e, uint64_t out[4]) {
    if (!e || !out) return fail(SPH_ERR_ARG, "null argument");
    if (!e->slab || !e->faceCap) return fail(SPH_ERR_STATE, "slab engine with face buffers required");
    out[0] = e->sentBytes[0]; out[1] = e->sentBytes[1];
    out[2] = e->hasLo ? (uint64_t)slab_face_bytes(e->faceCap) : 0u; out[3] = e->hasHi ? (uint64_t)slab_face_bytes(e->faceCap) : 0u;
    return SPH_OK;
}
int sph_slab_step_times(SphEngine* e, float outMs[5]) {
    if (!e || !outMs) return fail(SPH_ERR_ARG, "null argument");
    if (!e->slab || !e->xstream || !e->optTiming) return fail(SPH_ERR_STATE, "needs a slab engine that has run a boundary-first step with SPH_OPT_TIMING on");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipStreamSynchronize(e->xstream));
    // {pack, transfer, unpack} durations on the exchange stream, then when the exchange ended and when the SPH pass ended, both measured from the pass's start
    float v[5] = {0, 0, 0, 0, 0};
    hipError_t er = hipEventElapsedTime(&v[0], e->evX[1], e->evX[2]);
    if (er == hipSuccess) er = hipEventElapsedTime(&v[1], e->evX[2], e->evX[3]);
    if (er == hipSuccess) er = hipEventElapsedTime(&v[2], e->evX[3], e->evX[4]);
    if (er == hipSuccess) er = hipEventElapsedTime(&v[3], e->evX[0], e->evX[4]);
    if (er == hipSuccess && e->evPassEnd) er = hipEventElapsedTime(&v[4], e->evX[0], e->evPassEnd);
    if (er != hipSuccess) return fail(SPH_ERR_STATE, "no timed boundary-first step yet: %s", hipGetErrorString(er));
    for (int i = 0; i < 5; ++i) outMs[i] = v[i];
    return SPH_OK;
}

}  // extern "C"

// RCCL is loaded at run time (dlopen): the library has no link-time dependency on it, and a host without RCCL still loads
// the engine (the exchange then fails with a message).
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
int rccl_load() {
    if (g_rccl.lib) return SPH_OK;
    void* h = nullptr;
    // SPH_RCCL_LIBRARY names the library that provides the nccl* entry points (default: librccl.so.1).  The tests put a stand-in there that moves the messages
    // between PROCESSES ON ONE GPU through shared memory and REFUSES a receive whose size differs from its send's (tests/fake_rccl/): the engine's own multi-rank
    // code -- plans, grouped face messages, unpack -- then runs between real ranks on a one-GPU box, where RCCL itself refuses two ranks on one device.
    if (const char* over = std::getenv("SPH_RCCL_LIBRARY")) {
        h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(SPH_ERR_HIP, "SPH_RCCL_LIBRARY=%s cannot be loaded: %s", over, dlerror());
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { if (h) break; h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); }
    if (!h) return fail(SPH_ERR_HIP, "RCCL not available: %s", dlerror());
    Rccl r;
    r.lib = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.Send = (decltype(r.Send))dlsym(h, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(h, "ncclRecv");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Send || !r.Recv || !r.AllReduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        dlclose(h);
        return fail(SPH_ERR_HIP, "RCCL library lacks a required entry point");
    }
    g_rccl = r;
    return SPH_OK;
}
#define NCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess) return fail(SPH_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));   \
    } while (0)
}  // namespace

struct SphComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int sph_comm_unique_id(void* out128) {
    if (!out128) return fail(SPH_ERR_ARG, "null argument");
    static_assert(sizeof(ncclUniqueId) == SPH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    int rc;
    if ((rc = rccl_load())) return rc;
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    std::memcpy(out128, &id, sizeof(id));
    return SPH_OK;
}
int sph_comm_create(SphComm** out, const void* id128, int rank, int world) {
    if (!out || !id128) return fail(SPH_ERR_ARG, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(SPH_ERR_ARG, "bad rank %d of %d", rank, world);
    int rc;
    if ((rc = rccl_load())) return rc;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    SphComm* c = new SphComm();
    c->rank = rank; c->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);       // one rank per process, on the current HIP device
    if (r != ncclSuccess) { delete c; return fail(SPH_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
    *out = c;
    return SPH_OK;
}
int sph_comm_destroy(SphComm* c) {
    if (!c) return SPH_OK;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return SPH_OK;
}
// Health check of the transport on THIS rank alone: one grouped ncclSend + ncclRecv of `bytes` bytes from the rank to
// itself (RCCL serves a self send / recv with a device copy), issued the way slab_transfer_rccl issues the face messages
// (ncclUint8 counts, a non-blocking stream that is not the null stream), then compared word for word on the host.
// It is the only way to run ncclSend / ncclRecv on a box with one GPU (RCCL refuses two ranks on one device).
int sph_comm_selftest(SphComm* c, uint64_t bytes) { return sph_comm_selftest_timed(c, bytes, nullptr); }
// (msOut: hipEvent time of the grouped send + receive alone)
int sph_comm_selftest_timed(SphComm* c, uint64_t bytes, float* msOut) {
    if (!c || !c->comm) return fail(SPH_ERR_ARG, "null communicator");
    if (bytes == 0 || bytes > (1ull << 30) || (bytes & 3ull)) return fail(SPH_ERR_ARG, "bytes must be a multiple of 4 in (0, 2^30]");
    const size_t words = (size_t)(bytes / 4);
    uint32_t *src = nullptr, *dst = nullptr;
    hipStream_t st = nullptr;
    int rc = SPH_OK;
    std::vector<uint32_t> host(words);
    auto cleanup = [&]() { if (st) (void)hipStreamDestroy(st); if (src) (void)hipFree(src); if (dst) (void)hipFree(dst); };
    if ((rc = dev_alloc(&src, words)) || (rc = dev_alloc(&dst, words))) { cleanup(); return rc; }
    hipError_t er = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (size_t i = 0; i < words; ++i) host[i] = (uint32_t)i * 2654435761u + 12345u;
    if (er == hipSuccess) er = hipMemcpyAsync(src, host.data(), bytes, hipMemcpyHostToDevice, st);
    if (er == hipSuccess) er = hipMemsetAsync(dst, 0, bytes, st);
    if (er != hipSuccess) { cleanup(); return fail(SPH_ERR_HIP, "self-test setup failed: %s", hipGetErrorString(er)); }
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (msOut) { (void)hipEventCreate(&t0); (void)hipEventCreate(&t1); (void)hipEventRecord(t0, st); }
    ncclResult_t g0 = g_rccl.GroupStart();
    ncclResult_t r = g0;
    if (r == ncclSuccess) r = g_rccl.Send(src, (size_t)bytes, ncclUint8, c->rank, c->comm, st);
    if (r == ncclSuccess) r = g_rccl.Recv(dst, (size_t)bytes, ncclUint8, c->rank, c->comm, st);
    ncclResult_t g1 = g0 == ncclSuccess ? g_rccl.GroupEnd() : g0;
    if (msOut) {
        *msOut = 0.0f;
        (void)hipEventRecord(t1, st);
        if (hipEventSynchronize(t1) == hipSuccess) (void)hipEventElapsedTime(msOut, t0, t1);
        (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    }
    if (r != ncclSuccess || g1 != ncclSuccess) {
        (void)hipStreamSynchronize(st);
        cleanup();
        return fail(SPH_ERR_HIP, "self send / recv failed: %s", g_rccl.GetErrorString(r != ncclSuccess ? r : g1));
    }
    std::vector<uint32_t> back(words);
    er = hipMemcpyAsync(back.data(), dst, bytes, hipMemcpyDeviceToHost, st);
    if (er == hipSuccess) er = hipStreamSynchronize(st);
    cleanup();
    if (er != hipSuccess) return fail(SPH_ERR_HIP, "self-test copy back failed: %s", hipGetErrorString(er));
    for (size_t i = 0; i < words; ++i)
        if (back[i] != host[i]) return fail(SPH_ERR_HIP, "self send / recv delivered wrong data at word %zu of %zu", i, words);
    return SPH_OK;
}
// One halo exchange of a substep: pack -> one grouped ncclSend / ncclRecv per z-neighbour (fixed-size messages: header
// record + faceCap payload records) -> unpack, all enqueued on the engine's stream: no host synchronisation, and the
// stream order makes the unpack wait for the receives and the next pack wait for the sends.
static int slab_check_comm(SphEngine* e, SphComm* c) {
    if (!e || !c) return fail(SPH_ERR_ARG, "null argument");
    if (!e->slab || !e->d_face[0]) return fail(SPH_ERR_STATE, "slab engine with face buffers required");
    // both directions: a rank that waits for a neighbour the neighbour does not know about would hang in ncclRecv
    if ((e->hasLo != 0) != (c->rank > 0) || (e->hasHi != 0) != (c->rank < c->world - 1))
        return fail(SPH_ERR_ARG, "slab neighbours (lo %d, hi %d) do not match rank %d of %d", e->hasLo, e->hasHi, c->rank, c->world);
    int rc;
    if (e->faceAgreedWith != c) {
        // Once per (engine, communicator): the face capacity is the size of every message, so all ranks must have
        // allocated the same one -- a mismatch would otherwise show up as a hang inside ncclSend / ncclRecv.
        uint32_t* d = nullptr;
        if ((rc = dev_alloc(&d, 2))) return rc;
        const uint32_t mine[2] = {e->faceCap, ~e->faceCap};
        uint32_t agreed[2] = {0, 0};
        hipError_t er = hipMemcpyAsync(d, mine, sizeof(mine), hipMemcpyHostToDevice, e->stream);
        ncclResult_t nr = ncclSuccess;
        if (er == hipSuccess) nr = g_rccl.AllReduce(d, d, 2, ncclUint32, ncclMax, c->comm, e->stream);
        if (er == hipSuccess && nr == ncclSuccess) er = hipMemcpyAsync(agreed, d, sizeof(agreed), hipMemcpyDeviceToHost, e->stream);
        if (er == hipSuccess && nr == ncclSuccess) er = hipStreamSynchronize(e->stream);
        (void)hipFree(d);
        if (nr != ncclSuccess) return fail(SPH_ERR_HIP, "ncclAllReduce failed: %s", g_rccl.GetErrorString(nr));
        if (er != hipSuccess) return fail(SPH_ERR_HIP, "face capacity agreement failed: %s", hipGetErrorString(er));
        if (agreed[0] != e->faceCap || agreed[1] != ~e->faceCap)
            return fail(SPH_ERR_ARG, "face capacity %u differs between ranks (largest %u, smallest %u): sph_slab_alloc_faces must be given the same value everywhere",
                        e->faceCap, agreed[0], ~agreed[1]);
        e->faceAgreedWith = c;
    }
    return SPH_OK;
}
// The grouped ncclSend / ncclRecv of one exchange's faces: per neighbour TWO send / receive pairs in ONE group -- header + the migrants in use, and the
// halo copies in use -- of unequal sizes.  One routine for the engine's exchange (slab_transfer_rccl) and for the self-test that runs exactly this
// pattern on a one-GPU box with the rank itself as both neighbours (sph_comm_selftest_faces): ncclSend / ncclRecv to one peer match in issue order.
static int rccl_post_faces(SphComm* c, hipStream_t st, const int peer[2], const bool has[2], char* const sendFace[2], char* const recvFace[2],
                           const uint32_t msgSend[4], const uint32_t msgRecv[4], uint32_t faceCap) {
    if (!has[0] && !has[1]) return SPH_OK;
    const size_t hoff = slab_face_halo_off(faceCap);
    NCCL_TRY(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int d = 0; d < 2 && r == ncclSuccess; ++d) {
        if (!has[d]) continue;
        r = g_rccl.Send(sendFace[d], sizeof(SlabHdr) + (size_t)msgSend[2 + d] * 64u, ncclUint8, peer[d], c->comm, st);
        if (r == ncclSuccess) r = g_rccl.Send(sendFace[d] + hoff, (size_t)msgSend[d] * 40u, ncclUint8, peer[d], c->comm, st);
        if (r == ncclSuccess) r = g_rccl.Recv(recvFace[d], sizeof(SlabHdr) + (size_t)msgRecv[2 + d] * 64u, ncclUint8, peer[d], c->comm, st);
        if (r == ncclSuccess) r = g_rccl.Recv(recvFace[d] + hoff, (size_t)msgRecv[d] * 40u, ncclUint8, peer[d], c->comm, st);
    }
    ncclResult_t g = g_rccl.GroupEnd();
    if (r != ncclSuccess) return fail(SPH_ERR_HIP, "ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString(r));
    if (g != ncclSuccess) return fail(SPH_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(g));
    return SPH_OK;
}
static int slab_transfer_rccl(SphEngine* e, SphComm* c, hipStream_t st) {
    const int peer[2] = {c->rank - 1, c->rank + 1};
    const bool has[2] = {e->hasLo != 0, e->hasHi != 0};
    char* const sendFace[2] = {e->d_face[0], e->d_face[1]};
    char* const recvFace[2] = {e->d_face[2], e->d_face[3]};
    return rccl_post_faces(c, st, peer, has, sendFace, recvFace, e->msgSend, e->msgRecv, e->faceCap);
}
// A wait for a neighbour that cannot hang the host: the event is polled, and after `seconds` the call fails (SPH_ERR_TIMEOUT).  The work behind the
// event is then still queued on the device (a receive whose sender never came): the process has to end; nothing can be re-posted on this communicator.
static int wait_event_deadline(hipEvent_t ev, double seconds, float* waitedMs) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waitedMs) *waitedMs = (float)(el * 1e3);
        if (q == hipSuccess) return SPH_OK;
        if (q != hipErrorNotReady) return fail(SPH_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        if (el > seconds) return SPH_ERR_TIMEOUT;
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50)); else std::this_thread::yield();
    }
}
// The fixed-size message that precedes the sized ones: every rank sends its PLAN (64 bytes, slab_plan) to its neighbours and receives theirs.  Its size
// depends on nothing, so it cannot be mismatched.
static int rccl_post_plans(SphComm* c, hipStream_t st, const int peer[2], const bool has[2], uint32_t* d_plans /* 48 words: mine, from lo, from hi */) {
    if (!has[0] && !has[1]) return SPH_OK;
    NCCL_TRY(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int d = 0; d < 2 && r == ncclSuccess; ++d) {
        if (!has[d]) continue;
        r = g_rccl.Send(d_plans, sizeof(SphSlabIntent), ncclUint8, peer[d], c->comm, st);
        if (r == ncclSuccess) r = g_rccl.Recv(d_plans + 16 * (1 + d), sizeof(SphSlabIntent), ncclUint8, peer[d], c->comm, st);
    }
    ncclResult_t g = g_rccl.GroupEnd();
    if (r != ncclSuccess) return fail(SPH_ERR_HIP, "ncclSend / ncclRecv of the plans failed: %s", g_rccl.GetErrorString(r));
    if (g != ncclSuccess) return fail(SPH_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(g));
    return SPH_OK;
}
static int ensure_handshake(SphEngine* e) {
    if (e->hstream) return SPH_OK;
    int rc;
    if ((rc = dev_alloc(&e->d_intent, 48))) return rc;
    HIP_TRY(hipHostMalloc((void**)&e->h_intent, 48 * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&e->evIntent, hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&e->hstream, hipStreamNonBlocking));
    return SPH_OK;
}
// BEFORE a sized message of this exchange is posted: the plans cross each link and are compared on the host (SPH_SLAB_VERIFY 1, the default).  What this
// costs: one more grouped send / receive of 64 bytes per neighbour on a stream of its own, and a host wait until the neighbours have reached the same
// exchange -- which bounds the host's look-ahead to about one exchange (the device still holds more than a substep of queued work meanwhile).  What it
// buys: two ranks whose host-side state went apart (an impulse, a member edit, an upload or a skipped step on one rank only) get an error with the
// difference by name on BOTH ranks, where the sized ncclSend / ncclRecv would have hung or cut records off; and a neighbour that never arrives is a
// timeout with this rank's step and sizes, not a hang.
static int slab_handshake_rccl(SphEngine* e, SphComm* c) {
    if (!e->verifyMode || !(e->hasLo || e->hasHi)) return SPH_OK;
    if (!e->intentValid) return fail(SPH_ERR_STATE, "no plan for this exchange");
    int rc;
    if ((rc = ensure_handshake(e))) return rc;
    std::memcpy(e->h_intent, &e->intent, sizeof(SphSlabIntent));
    std::memset(e->h_intent + 16, 0, 2 * sizeof(SphSlabIntent));
    HIP_TRY(hipMemcpyAsync(e->d_intent, e->h_intent, 48 * sizeof(uint32_t), hipMemcpyHostToDevice, e->hstream));
    const int peer[2] = {c->rank - 1, c->rank + 1};
    const bool has[2] = {e->hasLo != 0, e->hasHi != 0};
    if ((rc = rccl_post_plans(c, e->hstream, peer, has, e->d_intent))) return rc;
    HIP_TRY(hipMemcpyAsync(e->h_intent + 16, e->d_intent + 16, 2 * sizeof(SphSlabIntent), hipMemcpyDeviceToHost, e->hstream));
    HIP_TRY(hipEventRecord(e->evIntent, e->hstream));
    rc = wait_event_deadline(e->evIntent, e->deadlineSec, &e->handshakeMs);
    e->handshakes += 1u;
    if (rc == SPH_ERR_TIMEOUT)
        return fail(SPH_ERR_TIMEOUT, "rank %d of %d: no plan from a neighbour within %.1f s at exchange %u (step %u; this rank would send %u + %u records down, %u + %u up): "
                                     "a neighbour rank is not making the same calls, or is gone.  The receive stays queued on the device: end this process.",
                    c->rank, c->world, e->deadlineSec, e->intent.exchangeNo, e->stepNo, e->intent.sendHalo[0], e->intent.sendMig[0], e->intent.sendHalo[1], e->intent.sendMig[1]);
    if (rc) return rc;
    for (int d = 0; d < 2; ++d) {
        if (!has[d]) continue;
        SphSlabIntent nb;
        std::memcpy(&nb, e->h_intent + 16 * (1 + d), sizeof(nb));
        char why[320];
        if (slab_intent_mismatch(e->intent, nb, d, why, sizeof(why)))
            return fail(SPH_ERR_STATE, "rank %d of %d: slab exchange %u refused before any record moved: %s", c->rank, c->world, e->intent.exchangeNo, why);
    }
    return SPH_OK;
}
// One halo exchange of a substep: plan -> (the plans cross the links and are compared) -> pack -> per z-neighbour two grouped ncclSend / ncclRecv pairs -> unpack,
// enqueued on the engine's stream; the stream order makes the unpack wait for the receives and the next pack wait for the sends.  With SPH_SLAB_VERIFY 0
// there is no host wait on the path at all (the sizes come from counts that are two exchanges old).
int sph_slab_exchange(SphEngine* e, SphComm* c) {
    int rc;
    if ((rc = slab_check_comm(e, c))) return rc;
    if (e->stepPending) return fail(SPH_ERR_STATE, "a boundary-first step is pending: finish it with sph_slab_step_finish");
    if (e->params.param_pause) return SPH_OK;                // the paused sph_dispatch that follows is a no-op: the halo records in place stay valid
    if ((rc = slab_plan(e))) return rc;
    if ((rc = slab_handshake_rccl(e, c))) return rc;
    if ((rc = slab_pack_on(e, e->stream))) return rc;
    if ((rc = slab_transfer_rccl(e, c, e->stream))) return rc;
    if ((rc = slab_unpack_on(e, e->stream, e->d_face[2], e->d_face[3], e->faceCap, e->msgRecv))) return rc;
    return slab_note_counts(e, e->stream);
}
// Second half of a boundary-first substep with RCCL as the transport: the transfer and the unpack follow the pack on the
// exchange stream; the engine's stream (the interior of the SPH pass) only waits for them at its end.
int sph_slab_step_finish(SphEngine* e, SphComm* c) {
    int rc;
    if ((rc = slab_check_comm(e, c))) return rc;
    if (!e->stepPending) return fail(SPH_ERR_STATE, "sph_slab_step_finish without sph_slab_step_begin");
    if (e->stepPaused) { e->stepPending = false; return SPH_OK; }
    if ((rc = slab_handshake_rccl(e, c))) { e->stepPending = false; return rc; }   // (the plan is sph_slab_step_begin's)
    if ((rc = slab_transfer_rccl(e, c, e->xstream))) return rc;
    if (e->optTiming) HIP_TRY(hipEventRecord(e->evX[3], e->xstream));
    if ((rc = slab_unpack_on(e, e->xstream, e->d_face[2], e->d_face[3], e->faceCap, e->msgRecv))) return rc;
    if ((rc = slab_note_counts(e, e->xstream))) return rc;
    return slab_step_join(e);
}

// ---- the agreement protocol's knobs and its view from outside ----
int sph_slab_set_verify(SphEngine* e, int mode) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (mode < 0 || mode > 1) return fail(SPH_ERR_ARG, "verify mode %d: 1 = the plans cross the links before every sized exchange (default), 0 = off", mode);
    e->verifyMode = mode;
    return SPH_OK;
}
int sph_slab_set_deadline(SphEngine* e, double seconds) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    if (!(seconds > 0.0)) return fail(SPH_ERR_ARG, "deadline must be > 0 s");
    e->deadlineSec = seconds;
    return SPH_OK;
}
int sph_slab_plan(const SphEngine* e, SphSlabIntent* out, float* handshakeMsOut) {
    if (!e || !out) return fail(SPH_ERR_ARG, "null argument");
    if (!e->slab) return fail(SPH_ERR_STATE, "not a slab engine");
    if (!e->intentValid) return fail(SPH_ERR_STATE, "no sized exchange has been planned yet");
    *out = e->intent;
    if (handshakeMsOut) *handshakeMsOut = e->handshakeMs;
    return SPH_OK;
}
int sph_slab_plans_agree(const SphSlabIntent* mine, const SphSlabIntent* neighbour, int side, char* why, size_t whyBytes) {
    if (!mine || !neighbour || side < 0 || side > 1) return fail(SPH_ERR_ARG, "bad argument");
    char buf[320];
    buf[0] = 0;
    const bool bad = slab_intent_mismatch(*mine, *neighbour, side, buf, sizeof(buf));
    if (why && whyBytes) { std::strncpy(why, buf, whyBytes - 1); why[whyBytes - 1] = 0; }
    return bad ? 0 : 1;
}
int sph_slab_debug_tight_messages(SphEngine* e, int on) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    e->tightMessages = on ? 1 : 0;
    return SPH_OK;
}
// Exactly the engine's exchange pattern over RCCL on ONE rank (the rank is its own lower and upper neighbour): the 64-byte plans first (with the
// polled, deadline-bounded wait of the handshake), then rccl_post_faces -- the routine slab_transfer_rccl calls -- with two send / receive pairs of
// UNEQUAL sizes per neighbour in one group, and every byte compared afterwards: what arrived is what was sent, nothing beyond a message was touched.
int sph_comm_selftest_faces(SphComm* c, uint32_t faceCap, const uint32_t counts[4] /* halo lo, halo hi, migrants lo, migrants hi */, float* msOut) {
    if (!c || !c->comm || !counts) return fail(SPH_ERR_ARG, "null argument");
    if (faceCap == 0 || faceCap > (1u << 24)) return fail(SPH_ERR_ARG, "face capacity %u out of range", faceCap);
    for (int i = 0; i < 4; ++i) if (counts[i] > faceCap) return fail(SPH_ERR_ARG, "count %u > face capacity %u", counts[i], faceCap);
    const size_t fb = slab_face_bytes(faceCap), hoff = slab_face_halo_off(faceCap);
    char* face[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t* d_plans = nullptr;
    hipStream_t st = nullptr, hs = nullptr;
    hipEvent_t ev = nullptr, t0 = nullptr, t1 = nullptr;
    auto cleanup = [&]() {
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        if (hs) { (void)hipStreamSynchronize(hs); (void)hipStreamDestroy(hs); }
        for (auto& f : face) if (f) (void)hipFree(f);
        if (d_plans) (void)hipFree(d_plans);
        for (hipEvent_t x : {ev, t0, t1}) if (x) (void)hipEventDestroy(x);
    };
    int rc = SPH_OK;
    for (int i = 0; i < 4 && !rc; ++i) rc = dev_alloc(&face[i], fb);
    if (!rc) rc = dev_alloc(&d_plans, 48);
    if (rc) { cleanup(); return rc; }
    hipError_t er = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (er == hipSuccess) er = hipStreamCreateWithFlags(&hs, hipStreamNonBlocking);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (er == hipSuccess) er = hipEventCreate(&t0);
    if (er == hipSuccess) er = hipEventCreate(&t1);
    std::vector<unsigned char> host[2];
    for (int d = 0; d < 2 && er == hipSuccess; ++d) {
        host[d].resize(fb);
        for (size_t i = 0; i < fb; ++i) host[d][i] = (unsigned char)((i * 2654435761u + 977u * (unsigned)d + (i >> 11)) >> 7);
        er = hipMemcpyAsync(face[d], host[d].data(), fb, hipMemcpyHostToDevice, st);
        if (er == hipSuccess) er = hipMemsetAsync(face[2 + d], 0, fb, st);
    }
    if (er == hipSuccess) er = hipStreamSynchronize(st);
    if (er != hipSuccess) { cleanup(); return fail(SPH_ERR_HIP, "self-test setup failed: %s", hipGetErrorString(er)); }
    const int peer[2] = {c->rank, c->rank};
    const bool has[2] = {true, true};
    // the plans
    uint32_t plans[48];
    for (int i = 0; i < 48; ++i) plans[i] = i < 16 ? 0x504c0000u + (uint32_t)i : 0u;
    er = hipMemcpyAsync(d_plans, plans, sizeof(plans), hipMemcpyHostToDevice, hs);
    if (er == hipSuccess) er = hipStreamSynchronize(hs);
    if (er != hipSuccess) { cleanup(); return fail(SPH_ERR_HIP, "self-test setup failed: %s", hipGetErrorString(er)); }
    if ((rc = rccl_post_plans(c, hs, peer, has, d_plans))) { cleanup(); return rc; }
    (void)hipEventRecord(ev, hs);
    float waited = 0.0f;
    rc = wait_event_deadline(ev, 20.0, &waited);
    if (rc) { cleanup(); return rc == SPH_ERR_TIMEOUT ? fail(SPH_ERR_TIMEOUT, "the plans' send / receive to self did not complete within 20 s") : rc; }
    uint32_t back[48];
    er = hipMemcpy(back, d_plans, sizeof(back), hipMemcpyDeviceToHost);
    if (er != hipSuccess) { cleanup(); return fail(SPH_ERR_HIP, "self-test copy back failed: %s", hipGetErrorString(er)); }
    for (int i = 0; i < 32; ++i)
        if (back[16 + i] != plans[i & 15]) { cleanup(); return fail(SPH_ERR_HIP, "a 64-byte plan sent to self arrived wrong at word %d", i); }
    // the faces: the sizes a receiver posts are the sizes the matching sender uses (what the plans' comparison guarantees between two ranks)
    (void)hipEventRecord(t0, st);
    if ((rc = rccl_post_faces(c, st, peer, has, face, face + 2, counts, counts, faceCap))) { cleanup(); return rc; }
    (void)hipEventRecord(t1, st);
    rc = wait_event_deadline(t1, 60.0, nullptr);
    if (rc) { cleanup(); return rc == SPH_ERR_TIMEOUT ? fail(SPH_ERR_TIMEOUT, "the face messages to self did not complete within 60 s") : rc; }
    if (msOut) { *msOut = 0.0f; (void)hipEventElapsedTime(msOut, t0, t1); }
    std::vector<unsigned char> got(fb);
    for (int d = 0; d < 2; ++d) {
        er = hipMemcpy(got.data(), face[2 + d], fb, hipMemcpyDeviceToHost);
        if (er != hipSuccess) { cleanup(); return fail(SPH_ERR_HIP, "self-test copy back failed: %s", hipGetErrorString(er)); }
        const size_t migEnd = sizeof(SlabHdr) + (size_t)counts[2 + d] * 64u, haloEnd = hoff + (size_t)counts[d] * 40u;
        for (size_t i = 0; i < fb; ++i) {
            const bool sent = i < migEnd || (i >= hoff && i < haloEnd);
            const unsigned char want = sent ? host[d][i] : (unsigned char)0;
            if (got[i] != want) { cleanup(); return fail(SPH_ERR_HIP, "face %d: byte %zu of %zu is %u, expected %u (%s a message)", d, i, fb, got[i], want, sent ? "inside" : "beyond"); }
        }
    }
    cleanup();
    return SPH_OK;
}

int sph_sync(SphEngine* e) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->xstream) HIP_TRY(hipStreamSynchronize(e->xstream));
    return SPH_OK;
}
// sph_sync that cannot hang: the engine's streams are polled, and after `seconds` (<= 0: the engine's deadline, sph_slab_set_deadline) the call fails with
// SPH_ERR_TIMEOUT and says where this engine stands.  For multi-rank hosts (bench.py --gpus N, halo.SlabSimulation): a transfer whose peer never posted
// its half keeps the stream busy for ever; the host prints the message and ends the process (a fresh process, never a re-exec).
int sph_sync_deadline(SphEngine* e, double seconds) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    const double limit = seconds > 0.0 ? seconds : e->deadlineSec;
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t sts[4] = {e->stream, e->xstream, e->bstream, e->hstream};
    for (unsigned spins = 0;; ++spins) {
        bool busy = false;
        for (hipStream_t st : sts) {
            if (!st) continue;
            const hipError_t q = hipStreamQuery(st);
            if (q == hipErrorNotReady) busy = true;
            else if (q != hipSuccess) return fail(SPH_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
        }
        if (!busy) return SPH_OK;
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > limit)
            return fail(SPH_ERR_TIMEOUT, "the engine's streams did not drain within %.1f s (boundary-first steps begun %u, sized exchanges enqueued %u, last plan: %u + %u records down, %u + %u up): "
                                         "a transfer is waiting for a neighbour rank that never posted its half.  End this process.",
                        limit, e->stepNo, e->exchangeNo, e->intent.sendHalo[0], e->intent.sendMig[0], e->intent.sendHalo[1], e->intent.sendMig[1]);
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(100)); else std::this_thread::yield();
    }
}

int sph_kernel_times(SphEngine* e, double msOut[SPH_K_COUNT], int64_t launchesOut[SPH_K_COUNT], int reset) {
    if (!e) return fail(SPH_ERR_ARG, "null engine");
    int rc;
    if ((rc = flush_events(e))) return rc;
    for (int i = 0; i < SPH_K_COUNT; ++i) {
        if (msOut) msOut[i] = e->kms[i];
        if (launchesOut) launchesOut[i] = e->klaunch[i];
        if (reset) { e->kms[i] = 0.0; e->klaunch[i] = 0; }
    }
    return SPH_OK;
}

}  // extern "C"
