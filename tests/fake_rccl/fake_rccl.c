/* fake_rccl.c -- TEST INFRASTRUCTURE: a stand-in for librccl.so that moves messages between PROCESSES ON ONE GPU through POSIX shared memory.
 *
 * The engine loads RCCL with dlopen (csrc/sph_engine.hip rccl_load); with SPH_RCCL_LIBRARY set it loads this library instead, so that the engine's own
 * multi-rank code -- the 64-byte plans crossing each link, the grouped face messages (two send / receive pairs of unequal sizes per neighbour), the unpack,
 * the flags -- runs between two or three real ranks on a one-GPU box (RCCL itself refuses two ranks on one device).  It is stricter than RCCL where that
 * helps a test: a receive whose size differs from the matching send's FAILS with both sizes (real RCCL hangs or cuts the message off), and a receive nobody
 * sends to fails after FAKE_RCCL_TIMEOUT_S seconds.  Semantics kept from NCCL: sends / receives to one peer match in issue order; a group's operations are
 * issued together at ncclGroupEnd (all sends, then all receives).  NOT kept: asynchrony -- every operation synchronises its stream and copies through the
 * host, so this says nothing about overlap or speed.  Never loaded by the product unless the environment variable names it. */
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3 } ncclDataType_t;   /* (the values of rccl.h for the types the engine uses) */
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;

#define MAX_RANKS 8
#define SLOTS 16
typedef struct {
    volatile uint64_t head, tail;            /* messages popped / pushed */
    volatile uint64_t size[SLOTS];
} Fifo;
typedef struct {
    volatile uint32_t ready;                 /* header initialised */
    volatile uint32_t attached, detached;
    uint64_t slotBytes;
    volatile uint32_t arEpoch[MAX_RANKS];    /* all-reduce: round each rank has arrived at */
    volatile uint32_t arDone[MAX_RANKS];     /* ... and has read */
    volatile uint32_t arVal[MAX_RANKS][8];
    Fifo fifo[MAX_RANKS][MAX_RANKS];         /* [src][dst] */
} Header;
typedef struct fakeComm {
    int rank, world;
    char name[160];
    Header* h;
    size_t mapBytes;
    uint32_t arCount;
} fakeComm;
typedef fakeComm* ncclComm_t;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double timeout_s(void) { const char* e = getenv("FAKE_RCCL_TIMEOUT_S"); return e ? atof(e) : 20.0; }
static uint64_t slot_bytes(void) { const char* e = getenv("FAKE_RCCL_SLOT_BYTES"); return e ? (uint64_t)atoll(e) : (uint64_t)(4u << 20); }
static char* payload(Header* h, int src, int dst, int slot) {
    const size_t base = (sizeof(Header) + 4095u) & ~(size_t)4095u;
    return (char*)h + base + (((size_t)src * MAX_RANKS + (size_t)dst) * SLOTS + (size_t)slot) * h->slotBytes;
}
static char g_err[256] = "no error";

const char* ncclGetErrorString(ncclResult_t r) { (void)r; return g_err; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/sph_fake_rccl_%d_%ld", (int)getpid(), (long)(now_s() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) { snprintf(g_err, sizeof(g_err), "fake rccl: bad rank %d of %d", rank, nranks); return ncclInvalidArgument; }
    fakeComm* c = (fakeComm*)calloc(1, sizeof(fakeComm));
    c->rank = rank; c->world = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    const uint64_t sb = slot_bytes();
    c->mapBytes = ((sizeof(Header) + 4095u) & ~(size_t)4095u) + (size_t)MAX_RANKS * MAX_RANKS * SLOTS * sb;   /* sparse: only touched pages exist */
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { snprintf(g_err, sizeof(g_err), "fake rccl: shm_open(%s): %s", c->name, strerror(errno)); free(c); return ncclSystemError; }
    if (ftruncate(fd, (off_t)c->mapBytes) != 0) { snprintf(g_err, sizeof(g_err), "fake rccl: ftruncate: %s", strerror(errno)); close(fd); free(c); return ncclSystemError; }
    c->h = (Header*)mmap(NULL, c->mapBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->h == MAP_FAILED) { snprintf(g_err, sizeof(g_err), "fake rccl: mmap: %s", strerror(errno)); free(c); return ncclSystemError; }
    if (rank == 0) { c->h->slotBytes = sb; __sync_synchronize(); c->h->ready = 1; }      /* (a fresh segment is zero-filled) */
    const double t0 = now_s();
    while (!c->h->ready) { if (now_s() - t0 > timeout_s()) { snprintf(g_err, sizeof(g_err), "fake rccl: rank 0 never initialised the segment"); return ncclSystemError; } usleep(200); }
    __sync_fetch_and_add(&c->h->attached, 1);
    while ((int)c->h->attached < nranks) { if (now_s() - t0 > timeout_s()) { snprintf(g_err, sizeof(g_err), "fake rccl: only %u of %d ranks attached", c->h->attached, nranks); return ncclSystemError; } usleep(200); }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (__sync_add_and_fetch(&c->h->detached, 1) == (uint32_t)c->world) shm_unlink(c->name);
    munmap(c->h, c->mapBytes);
    free(c);
    return ncclSuccess;
}

static size_t type_bytes(ncclDataType_t t) { return (t == ncclInt8 || t == ncclUint8) ? 1u : 4u; }

typedef struct { int isRecv; void* buf; size_t bytes; int peer; fakeComm* c; hipStream_t st; } Op;
static __thread Op g_ops[64];
static __thread int g_nops = 0, g_depth = 0;

static ncclResult_t do_send(const Op* o) {
    fakeComm* c = o->c;
    Header* h = c->h;
    if (o->bytes > h->slotBytes) { snprintf(g_err, sizeof(g_err), "fake rccl: message of %zu bytes exceeds FAKE_RCCL_SLOT_BYTES %llu", o->bytes, (unsigned long long)h->slotBytes); return ncclInvalidArgument; }
    if (hipStreamSynchronize(o->st) != hipSuccess) { snprintf(g_err, sizeof(g_err), "fake rccl: hipStreamSynchronize failed"); return ncclUnhandledCudaError; }
    Fifo* f = &h->fifo[c->rank][o->peer];
    const double t0 = now_s();
    while (f->tail - f->head >= SLOTS) { if (now_s() - t0 > timeout_s()) { snprintf(g_err, sizeof(g_err), "fake rccl: rank %d -> %d: %d messages nobody receives", c->rank, o->peer, SLOTS); return ncclSystemError; } usleep(50); }
    const int slot = (int)(f->tail % SLOTS);
    if (o->bytes && hipMemcpy(payload(h, c->rank, o->peer, slot), o->buf, o->bytes, hipMemcpyDeviceToHost) != hipSuccess) { snprintf(g_err, sizeof(g_err), "fake rccl: copy to host failed"); return ncclUnhandledCudaError; }
    f->size[slot] = o->bytes;
    __sync_synchronize();
    f->tail = f->tail + 1;
    return ncclSuccess;
}
static ncclResult_t do_recv(const Op* o) {
    fakeComm* c = o->c;
    Header* h = c->h;
    if (hipStreamSynchronize(o->st) != hipSuccess) { snprintf(g_err, sizeof(g_err), "fake rccl: hipStreamSynchronize failed"); return ncclUnhandledCudaError; }
    Fifo* f = &h->fifo[o->peer][c->rank];
    const double t0 = now_s();
    while (f->tail == f->head) {
        if (now_s() - t0 > timeout_s()) { snprintf(g_err, sizeof(g_err), "fake rccl: rank %d waited %.0f s for a message of %zu bytes from rank %d that was never sent", c->rank, timeout_s(), o->bytes, o->peer); return ncclSystemError; }
        usleep(50);
    }
    __sync_synchronize();
    const int slot = (int)(f->head % SLOTS);
    if (f->size[slot] != o->bytes) {          /* what real RCCL turns into a hang or a truncated message */
        snprintf(g_err, sizeof(g_err), "fake rccl: SIZE MISMATCH on the link %d -> %d: %llu bytes were sent, the matching receive was posted for %zu", o->peer, c->rank, (unsigned long long)f->size[slot], o->bytes);
        fprintf(stderr, "%s\n", g_err);
        return ncclInvalidArgument;
    }
    if (o->bytes && hipMemcpy(o->buf, payload(h, o->peer, c->rank, slot), o->bytes, hipMemcpyHostToDevice) != hipSuccess) { snprintf(g_err, sizeof(g_err), "fake rccl: copy to device failed"); return ncclUnhandledCudaError; }
    __sync_synchronize();
    f->head = f->head + 1;
    return ncclSuccess;
}
static ncclResult_t flush_ops(void) {
    ncclResult_t r = ncclSuccess;
    for (int i = 0; i < g_nops && r == ncclSuccess; ++i) if (!g_ops[i].isRecv) r = do_send(&g_ops[i]);
    for (int i = 0; i < g_nops && r == ncclSuccess; ++i) if (g_ops[i].isRecv) r = do_recv(&g_ops[i]);
    g_nops = 0;
    return r;
}
static ncclResult_t post(int isRecv, void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    if (!c || peer < 0 || peer >= c->world) { snprintf(g_err, sizeof(g_err), "fake rccl: bad peer %d", peer); return ncclInvalidArgument; }
    if (g_nops >= 64) { snprintf(g_err, sizeof(g_err), "fake rccl: more than 64 operations in a group"); return ncclInvalidUsage; }
    Op o = {isRecv, buf, count * type_bytes(t), peer, c, st};
    g_ops[g_nops++] = o;
    return g_depth ? ncclSuccess : flush_ops();
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) { return post(0, (void*)buf, count, t, peer, c, st); }
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) { return post(1, buf, count, t, peer, c, st); }
ncclResult_t ncclGroupStart(void) { g_depth += 1; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { if (g_depth > 0) g_depth -= 1; return g_depth ? ncclSuccess : flush_ops(); }

/* uint32 max / min / sum over up to 8 words (the engine's face-capacity agreement) */
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
    if (t != ncclUint32 || count > 8) { snprintf(g_err, sizeof(g_err), "fake rccl: all-reduce of this shape is not implemented"); return ncclInvalidArgument; }
    uint32_t mine[8] = {0}, out[8];
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(mine, send, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    Header* h = c->h;
    const uint32_t round = ++c->arCount;
    const double t0 = now_s();
    for (size_t i = 0; i < count; ++i) h->arVal[c->rank][i] = mine[i];
    __sync_synchronize();
    h->arEpoch[c->rank] = round;                            /* arrived */
    for (int r = 0; r < c->world; ++r)
        while (h->arEpoch[r] < round) { if (now_s() - t0 > timeout_s()) { snprintf(g_err, sizeof(g_err), "fake rccl: all-reduce: rank %d never arrived", r); return ncclSystemError; } usleep(50); }
    __sync_synchronize();
    for (size_t i = 0; i < count; ++i) {
        uint32_t v = h->arVal[0][i];
        for (int r = 1; r < c->world; ++r) { const uint32_t x = h->arVal[r][i]; v = op == ncclMax ? (x > v ? x : v) : op == ncclMin ? (x < v ? x : v) : v + x; }
        out[i] = v;
    }
    __sync_synchronize();
    h->arDone[c->rank] = round;                             /* read: nobody writes the next round's values before everybody has read this one's */
    for (int r = 0; r < c->world; ++r)
        while (h->arDone[r] < round) { if (now_s() - t0 > timeout_s()) { snprintf(g_err, sizeof(g_err), "fake rccl: all-reduce: rank %d stuck", r); return ncclSystemError; } usleep(50); }
    if (hipMemcpy(recv, out, count * 4, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
