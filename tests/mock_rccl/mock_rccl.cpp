/*
 * mock_rccl.cpp -- a stand-in for librccl.so over POSIX shared memory.  TEST INFRASTRUCTURE (never shipped, never loaded by
 * the product unless LH_RCCL_LIBRARY points at it).
 *
 * Why: the test box has ONE GPU and RCCL refuses two ranks on one device, so the RCCL branch of lucille_amd/csrc/lh_dist.hip
 * -- ncclBroadcast of the scene image, the grouped ncclSend / ncclRecv gather, the status words agreed before a payload moves,
 * the error returns -- never ran with real peers.  This library implements exactly the nine entry points lh_dist.hip
 * dlsym()s (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclBroadcast, ncclSend, ncclRecv, ncclGroupStart,
 * ncclGroupEnd, ncclGetErrorString) with the library's semantics as lh_dist.hip relies on them: stream order (the stream is
 * synchronised before bytes are read or written), grouped point-to-point operations issued at ncclGroupEnd, buffered sends,
 * receives matched per (source, destination) in posting order.  Any number of ranks may share a device.
 *
 * Test hooks (environment, read at ncclCommInitRank):
 *   MOCK_RCCL_LOG=prefix         every call is appended to prefix.rank<R> ("GroupStart", "Send 0 1024", ...)
 *   MOCK_RCCL_FAIL=op:rank:n     the (n+1)-th call of op (send | recv | broadcast) on that rank returns ncclSystemError
 *   MOCK_RCCL_TIMEOUT=seconds    a receive / broadcast whose peer never shows up fails after this long (default 30)
 *
 * Link model (round 6; tools/predict8.py): the bytes really move through host shared memory, far slower than xGMI, so wall time
 * here says nothing about a node.  Instead every rank keeps a MODEL CLOCK of what its calls would cost on one:
 *   MOCK_RCCL_LATENCY_US=us      per call that reaches the library's proxy: an ungrouped ncclSend / ncclRecv / ncclBroadcast, or
 *                                one ncclGroupEnd that closes a non-empty group
 *   MOCK_RCCL_GBPS=rate          per peer and direction (xGMI is point-to-point: the N - 1 receives of a gather cross N - 1 links
 *                                at once) -- a call costs latency + the LARGEST of its transfers / rate; a broadcast of B bytes
 *                                costs latency + B / rate (a pipelined ring moves every byte over every link once)
 * mock_rccl_model_seconds() / _calls() / _bytes() read this process's clock, mock_rccl_model_reset() zeroes it.  Nothing sleeps:
 * the clock is a count, read by the tool that adds it to kernel times measured on the GPU.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <stdarg.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#define MAXR 16

struct mock_ctl {
    volatile unsigned arrived, generation;
    volatile unsigned long long sent[MAXR][MAXR], consumed[MAXR][MAXR];      /* messages published / taken per (src, dst) */
    volatile unsigned long long bcast_pub, bcast_taken;                         /* broadcasts published by the root / copies taken by the others */
};

struct op_t { int kind; void *buf; size_t bytes; int peer; hipStream_t stream; };   /* kind 0 send, 1 recv */

struct ncclComm {
    int rank, nranks; mock_ctl *ctl; char name[96];
    unsigned long long nbcast;
    FILE *log; double timeout;
    int fail_op, fail_rank; long fail_after; long calls[3];
};

static double g_model_s = 0.0, g_lat_s = 0.0, g_rate = 0.0;          /* this process's model clock; seconds per call; bytes per second per link (0: free) */
static unsigned long long g_model_calls = 0, g_model_bytes = 0;
static void model_call(size_t largest_bytes, size_t all_bytes)
{
    g_model_s += g_lat_s + (g_rate > 0.0 ? (double)largest_bytes / g_rate : 0.0);
    g_model_calls++; g_model_bytes += all_bytes;
}
extern "C" double mock_rccl_model_seconds(void) { return g_model_s; }
extern "C" unsigned long long mock_rccl_model_calls(void) { return g_model_calls; }
extern "C" unsigned long long mock_rccl_model_bytes(void) { return g_model_bytes; }
extern "C" void mock_rccl_model_reset(void) { g_model_s = 0.0; g_model_calls = 0; g_model_bytes = 0; }

static thread_local int g_depth = 0;
static thread_local std::vector<std::pair<ncclComm *, op_t>> g_ops;

static double now_sec(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static size_t type_size(ncclDataType_t t)
{
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
                 case ncclInt64: case ncclUint64: case ncclFloat64: return 8; default: return 1; }
}

static void logf_(ncclComm *c, const char *fmt, ...)
{
    if (!c->log) return;
    va_list ap; va_start(ap, fmt); vfprintf(c->log, fmt, ap); va_end(ap); fputc('\n', c->log); fflush(c->log);
}

static int barrier(ncclComm *c)
{
    mock_ctl *k = c->ctl; const unsigned gen = k->generation;
    if (__atomic_add_fetch(&k->arrived, 1u, __ATOMIC_ACQ_REL) == (unsigned)c->nranks) {
        __atomic_store_n(&k->arrived, 0u, __ATOMIC_RELAXED); __atomic_store_n(&k->generation, gen + 1u, __ATOMIC_RELEASE); return 0;
    }
    const double t0 = now_sec();
    while (__atomic_load_n(&k->generation, __ATOMIC_ACQUIRE) == gen) { usleep(50); if (now_sec() - t0 > c->timeout) return -1; }
    return 0;
}

static bool fails(ncclComm *c, int op)
{
    const long n = c->calls[op]++;
    return c->fail_op == op && c->fail_rank == c->rank && n == c->fail_after;
}

extern "C" ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(id->internal, 1, 32, f) != 32) { unsigned long long t = (unsigned long long)(now_sec() * 1e9) ^ ((unsigned long long)getpid() << 32); memcpy(id->internal, &t, sizeof(t)); }
    if (f) fclose(f);
    memcpy(id->internal + 40, "MOCKRCCL", 8);
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm *c = (ncclComm *)calloc(1, sizeof(*c));
    unsigned long long h = 1469598103934665603ull;
    for (size_t k = 0; k < sizeof(id.internal); k++) { h ^= (unsigned char)id.internal[k]; h *= 1099511628211ull; }
    snprintf(c->name, sizeof(c->name), "/mock_rccl_%016llx", h);
    c->rank = rank; c->nranks = nranks; c->timeout = 30.0; c->fail_op = -1;
    if (const char *e = getenv("MOCK_RCCL_TIMEOUT")) c->timeout = atof(e);
    if (const char *e = getenv("MOCK_RCCL_LATENCY_US")) g_lat_s = atof(e) * 1e-6;
    if (const char *e = getenv("MOCK_RCCL_GBPS")) g_rate = atof(e) * 1e9;
    if (const char *e = getenv("MOCK_RCCL_LOG")) { char p[1200]; snprintf(p, sizeof(p), "%s.rank%d", e, rank); c->log = fopen(p, "a"); }
    if (const char *e = getenv("MOCK_RCCL_FAIL")) {
        char op[32] = ""; int r = -1; long n = 0;
        if (sscanf(e, "%31[^:]:%d:%ld", op, &r, &n) == 3) { c->fail_op = !strcmp(op, "send") ? 0 : !strcmp(op, "recv") ? 1 : !strcmp(op, "broadcast") ? 2 : -1; c->fail_rank = r; c->fail_after = n; }
    }
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(mock_ctl)) != 0) { if (fd >= 0) close(fd); free(c); return ncclSystemError; }
    c->ctl = (mock_ctl *)mmap(NULL, sizeof(mock_ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); close(fd);
    if ((void *)c->ctl == MAP_FAILED) { free(c); return ncclSystemError; }
    logf_(c, "CommInitRank %d %d", nranks, rank);
    if (barrier(c) != 0) { munmap((void *)c->ctl, sizeof(mock_ctl)); free(c); return ncclSystemError; }
    *comm = c;
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    logf_(c, "CommDestroy");
    (void)barrier(c);
    munmap((void *)c->ctl, sizeof(mock_ctl));
    if (c->rank == 0) shm_unlink(c->name);
    if (c->log) fclose(c->log);
    free(c);
    return ncclSuccess;
}

static void seg_name(ncclComm *c, char *out, size_t n, const char *kind, int a, int b, unsigned long long seq)
{
    snprintf(out, n, "%s_%s_%d_%d_%llu", c->name, kind, a, b, seq);
}

#define MOCK_WHY(c, ...) do { fprintf(stderr, "[mock_rccl rank %d] ", (c)->rank); fprintf(stderr, __VA_ARGS__); fprintf(stderr, " (errno %d: %s)\n", errno, strerror(errno)); } while (0)

static ncclResult_t put(ncclComm *c, const char *name, const void *dbuf, size_t bytes, hipStream_t s)
{
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)(bytes ? bytes : 1)) != 0) { MOCK_WHY(c, "put %s: shm_open / ftruncate of %zu bytes failed", name, bytes); if (fd >= 0) close(fd); return ncclSystemError; }
    void *seg = mmap(NULL, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); close(fd);
    if (seg == MAP_FAILED) { MOCK_WHY(c, "put %s: mmap of %zu bytes failed", name, bytes); return ncclSystemError; }
    if (hipStreamSynchronize(s) != hipSuccess || (bytes && hipMemcpy(seg, dbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess)) { munmap(seg, bytes ? bytes : 1); return ncclUnhandledCudaError; }
    munmap(seg, bytes ? bytes : 1);
    return ncclSuccess;
}

static ncclResult_t take(ncclComm *c, const char *name, void *dbuf, size_t bytes, hipStream_t s, bool unlink_it)
{
    int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) { MOCK_WHY(c, "take %s: shm_open failed", name); return ncclSystemError; }
    struct stat st; if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) { MOCK_WHY(c, "take %s: the segment holds %lld bytes, %zu wanted", name, (long long)st.st_size, bytes); close(fd); return ncclInvalidArgument; }      /* size mismatch between the peers */
    void *seg = mmap(NULL, bytes ? bytes : 1, PROT_READ, MAP_SHARED, fd, 0); close(fd);
    if (seg == MAP_FAILED) { MOCK_WHY(c, "take %s: mmap of %zu bytes failed", name, bytes); return ncclSystemError; }
    ncclResult_t rc = ncclSuccess;
    if (hipStreamSynchronize(s) != hipSuccess || (bytes && hipMemcpy(dbuf, seg, bytes, hipMemcpyHostToDevice) != hipSuccess)) rc = ncclUnhandledCudaError;
    munmap(seg, bytes ? bytes : 1);
    if (unlink_it) shm_unlink(name);
    return rc;
}

static ncclResult_t do_send(ncclComm *c, const op_t &o)
{
    if (fails(c, 0)) { logf_(c, "Send %d %zu FAILED (injected)", o.peer, o.bytes); return ncclSystemError; }
    char name[160]; const unsigned long long seq = c->ctl->sent[c->rank][o.peer];
    seg_name(c, name, sizeof(name), "p2p", c->rank, o.peer, seq);
    ncclResult_t rc = put(c, name, o.buf, o.bytes, o.stream);
    if (rc != ncclSuccess) return rc;
    __atomic_store_n(&c->ctl->sent[c->rank][o.peer], seq + 1, __ATOMIC_RELEASE);
    logf_(c, "Send %d %zu", o.peer, o.bytes);
    return ncclSuccess;
}

static ncclResult_t do_recv(ncclComm *c, const op_t &o)
{
    if (fails(c, 1)) { logf_(c, "Recv %d %zu FAILED (injected)", o.peer, o.bytes); return ncclSystemError; }
    const unsigned long long seq = c->ctl->consumed[o.peer][c->rank];
    const double t0 = now_sec();
    while (__atomic_load_n(&c->ctl->sent[o.peer][c->rank], __ATOMIC_ACQUIRE) <= seq) {
        usleep(50);
        if (now_sec() - t0 > c->timeout) { MOCK_WHY(c, "recv of %zu bytes from %d: nothing after %.0f s", o.bytes, o.peer, c->timeout); logf_(c, "Recv %d %zu TIMEOUT", o.peer, o.bytes); return ncclSystemError; }
    }
    char name[160]; seg_name(c, name, sizeof(name), "p2p", o.peer, c->rank, seq);
    ncclResult_t rc = take(c, name, o.buf, o.bytes, o.stream, true);
    c->ctl->consumed[o.peer][c->rank] = seq + 1;
    logf_(c, "Recv %d %zu", o.peer, o.bytes);
    return rc;
}

extern "C" ncclResult_t ncclGroupStart(void) { g_depth++; return ncclSuccess; }

extern "C" ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    /* buffered sends first, then the receives in posting order: no pattern of grouped operations can deadlock */
    for (auto &p : g_ops) if (p.second.kind == 0 && rc == ncclSuccess) rc = do_send(p.first, p.second);
    for (auto &p : g_ops) if (p.second.kind == 1 && rc == ncclSuccess) rc = do_recv(p.first, p.second);
    if (!g_ops.empty()) {
        size_t largest = 0, all = 0;
        for (auto &p : g_ops) { if (p.second.bytes > largest) largest = p.second.bytes; all += p.second.bytes; }
        model_call(largest, all);
        logf_(g_ops[0].first, "GroupEnd %zu", g_ops.size());
    }
    g_ops.clear();
    return rc;
}

extern "C" ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
    op_t o = {0, (void *)buf, count * type_size(t), peer, s};
    if (g_depth > 0) { g_ops.push_back({c, o}); return ncclSuccess; }
    model_call(o.bytes, o.bytes);
    return do_send(c, o);
}

extern "C" ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
    op_t o = {1, buf, count * type_size(t), peer, s};
    if (g_depth > 0) { g_ops.push_back({c, o}); return ncclSuccess; }
    model_call(o.bytes, o.bytes);
    return do_recv(c, o);
}

extern "C" ncclResult_t ncclBroadcast(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s)
{
    if (!c || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    const size_t bytes = count * type_size(t);
    const unsigned long long seq = c->nbcast++;
    if (fails(c, 2)) { logf_(c, "Broadcast %zu FAILED (injected)", bytes); return ncclSystemError; }
    char name[160]; seg_name(c, name, sizeof(name), "bc", root, 0, seq);
    if (c->nranks == 1) { logf_(c, "Broadcast %zu", bytes); return ncclSuccess; }
    model_call(bytes, bytes);
    if (c->rank == root) {
        ncclResult_t rc = put(c, name, sendbuf, bytes, s);
        if (rc != ncclSuccess) return rc;
        __atomic_store_n(&c->ctl->bcast_pub, seq + 1, __ATOMIC_RELEASE);
        const double t0 = now_sec();
        while (__atomic_load_n(&c->ctl->bcast_taken, __ATOMIC_ACQUIRE) < (seq + 1) * (unsigned long long)(c->nranks - 1)) {
            usleep(50);
            if (now_sec() - t0 > c->timeout) { MOCK_WHY(c, "broadcast %llu of %zu bytes: %llu of %llu copies taken after %.0f s", seq, bytes, (unsigned long long)c->ctl->bcast_taken, (seq + 1) * (unsigned long long)(c->nranks - 1), c->timeout); shm_unlink(name); logf_(c, "Broadcast %zu TIMEOUT", bytes); return ncclSystemError; }
        }
        shm_unlink(name);
    } else {
        const double t0 = now_sec();
        while (__atomic_load_n(&c->ctl->bcast_pub, __ATOMIC_ACQUIRE) <= seq) {
            usleep(50);
            if (now_sec() - t0 > c->timeout) { MOCK_WHY(c, "broadcast %llu of %zu bytes: the root published %llu after %.0f s", seq, bytes, (unsigned long long)c->ctl->bcast_pub, c->timeout); logf_(c, "Broadcast %zu TIMEOUT", bytes); return ncclSystemError; }
        }
        ncclResult_t rc = take(c, name, recvbuf, bytes, s, false);
        __atomic_add_fetch(&c->ctl->bcast_taken, 1ull, __ATOMIC_ACQ_REL);
        if (rc != ncclSuccess) return rc;
    }
    logf_(c, "Broadcast %zu", bytes);
    return ncclSuccess;
}

extern "C" const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "mock rccl: unhandled HIP error";
        case ncclSystemError: return "mock rccl: system error (peer missing, timeout or injected failure)";
        case ncclInvalidArgument: return "mock rccl: invalid argument (size mismatch between the peers?)";
        case ncclInvalidUsage: return "mock rccl: invalid usage";
        default: return "mock rccl: error";
    }
}
