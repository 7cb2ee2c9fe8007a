/*
 * lh_dist.hip -- one process per GPU, from plain C (include/lucille_hip.h "lh_dist_*").
 *
 * What it stands in for: lucille's compiled-out MPI layer -- ri_parallel_init / _barrier / _bcast / _gather / _send /
 * _recv (src/base/parallel.c:62-232) and the frame protocol built on it, "every rank renders, rank 0 owns the display"
 * (src/render/render.c:468-514).  Here the ranks are GPUs and the transport is RCCL over xGMI:
 *
 *   * scene load: ONE host build (rank 0), then ncclBroadcast of the flattened arrays -- traversal nodes, triangle
 *     records, lucille's own tree, per-primitive attributes -- into every rank's HBM (lh_dist_broadcast_scene); the
 *     other ranks never build and never hold a host copy;
 *   * frames: image space sharded over the ranks in interleaved full-width bands, a rank's bands rendered as ONE device
 *     batch (lh_render_ao_bands), ONE exchange step -- the gather of the ranks' slabs to rank 0 with ncclGroupStart /
 *     ncclSend / ncclRecv / ncclGroupEnd (each peer's slab crosses its own xGMI link) -- and a placement kernel
 *     (bucket_write's row order) on rank 0;
 *   * ray dumps: lh_dist_gather of hit-record slices, same primitive.
 *
 * There is no per-ray communication and no all-reduce.  RCCL is loaded at run time (dlopen "librccl.so.1") so that the
 * single-GPU entry points carry no dependency on it.  RCCL refuses two ranks on one device; for that case -- how the
 * N > 1 code path is exercised on a one-GPU test box -- LH_DIST_SHM selects a transport with the same interface over a
 * POSIX shared-memory segment (device -> host -> device), chosen automatically by lh_dist_init_file when ranks share a
 * device.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "lh_internal.h"

#define DFAIL(...) lh_fail(__VA_ARGS__)
#define LH_DIST_BAND_ROWS 16         /* lines per band of a sharded AO frame unless the caller says (render.py DEFAULT_BAND_ROWS; profiles/r05_shard_cost_table.md) */


struct rccl_api {
    void *lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
};
static rccl_api g_rccl;

static int rccl_load(void)
{
    if (g_rccl.lib) return 0;
    const char *names[] = {getenv("LH_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = NULL;
    for (size_t k = 0; k < sizeof(names) / sizeof(names[0]) && !lib; k++) if (names[k]) lib = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
    if (!lib) return DFAIL("lh_dist: RCCL is not loadable (%s)", dlerror());
#define SYM(F) do { *(void **)&g_rccl.F = dlsym(lib, "nccl" #F); if (!g_rccl.F) { dlclose(lib); return DFAIL("lh_dist: librccl lacks nccl" #F); } } while (0)
    SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(Broadcast); SYM(Send); SYM(Recv); SYM(GroupStart); SYM(GroupEnd); SYM(GetErrorString);
#undef SYM
    g_rccl.lib = lib;
    return 0;
}

#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return DFAIL("%s failed: %s", #x, g_rccl.GetErrorString(r_)); } while (0)

/* ---- shared-memory transport (ranks on one device: tests) ------------------------------------------------------- */
struct shm_ctl {                 /* control block of the job's segment */
    volatile unsigned arrived, generation;
    volatile unsigned long long data_bytes;       /* size of the data segment of the current operation */
};

struct lh_dist {
    int rank, world, device, transport;
    ncclComm_t comm;
    hipStream_t stream;          /* collectives without a caller stream */
    /* shm transport */
    char shm_name[96]; shm_ctl *ctl; unsigned op;
    /* frame assembly on rank 0 */
    lh_buf slab, mono, all, frame, bands, agree;
};

static double now_sec(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static int shm_barrier(lh_dist_t *d)
{
    shm_ctl *c = d->ctl;
    const unsigned gen = c->generation;
    if (__atomic_add_fetch(&c->arrived, 1u, __ATOMIC_ACQ_REL) == (unsigned)d->world) {
        __atomic_store_n(&c->arrived, 0u, __ATOMIC_RELAXED);
        __atomic_store_n(&c->generation, gen + 1u, __ATOMIC_RELEASE);
        return 0;
    }
    /* the others are microseconds away when the ranks run in step (the two barriers around a frame): watch the word for a while
     * before sleeping in 50 us slices -- a sleeping waiter leaves the barrier up to a slice late, and that spread is time a
     * sharded frame pays (tools/skew_probe.py) */
    const double t0 = now_sec();
    for (int spins = 0; __atomic_load_n(&c->generation, __ATOMIC_ACQUIRE) == gen; spins++) {
        if ((spins & 255) != 255) { LH_CPU_RELAX(); continue; }
        const double dt = now_sec() - t0;
        if (dt < 3.0e-3) continue;                    /* 3 ms of watching: ranks that render shares of one frame arrive within that */
        usleep(20);
        if (dt > 120.0) return DFAIL("lh_dist (shm): a rank did not reach the barrier within 120 s");
    }
    return 0;
}

/* the data segment of operation `op`: created by rank 0, mapped by everybody between two barriers */
static int shm_data(lh_dist_t *d, size_t bytes, void **out)
{
    char name[128];
    snprintf(name, sizeof(name), "%s_d%u", d->shm_name, d->op);
    if (d->rank == 0) {
        int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)(bytes ? bytes : 1)) != 0) { if (fd >= 0) close(fd); return DFAIL("lh_dist (shm): cannot create %s: %s", name, strerror(errno)); }
        *out = mmap(NULL, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); close(fd);
        if (*out == MAP_FAILED) return DFAIL("lh_dist (shm): mmap failed");
        if (shm_barrier(d) != 0) return -1;
    } else {
        if (shm_barrier(d) != 0) return -1;
        int fd = shm_open(name, O_RDWR, 0600);
        if (fd < 0) return DFAIL("lh_dist (shm): cannot open %s: %s", name, strerror(errno));
        *out = mmap(NULL, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); close(fd);
        if (*out == MAP_FAILED) return DFAIL("lh_dist (shm): mmap failed");
    }
    return 0;
}

static int shm_release(lh_dist_t *d, void *p, size_t bytes)
{
    char name[128];
    snprintf(name, sizeof(name), "%s_d%u", d->shm_name, d->op);
    if (shm_barrier(d) != 0) return -1;          /* everybody is done with the segment */
    munmap(p, bytes ? bytes : 1);
    if (d->rank == 0) shm_unlink(name);
    d->op++;
    return 0;
}

/* ---- lifetime ---------------------------------------------------------------------------------------------------- */
extern "C" int lh_dist_unique_id(void *id128)
{
    if (!id128) return DFAIL("lh_dist_unique_id: NULL");
    if (rccl_load() != 0) return -1;
    ncclUniqueId id;
    NCHK(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, LH_DIST_ID_BYTES);
    return 0;
}

static void name_from_id(const void *id128, char *out, size_t n)
{
    unsigned long long h = 1469598103934665603ull;
    for (int k = 0; k < LH_DIST_ID_BYTES; k++) { h ^= ((const unsigned char *)id128)[k]; h *= 1099511628211ull; }
    snprintf(out, n, "/lh_dist_%016llx", h);
}

/* an id of the host this process runs on: ranks with the same id can meet in one POSIX shared-memory segment */
static unsigned long long host_id(void)
{
    char buf[320]; memset(buf, 0, sizeof(buf));
    (void)gethostname(buf, 255);
    FILE *f = fopen("/proc/sys/kernel/random/boot_id", "r");          /* two containers may share a hostname, never a boot id + hostname + /dev/shm */
    if (f) { if (!fgets(buf + 256, 63, f)) buf[256] = 0; fclose(f); }
    unsigned long long h = 1469598103934665603ull;
    for (size_t k = 0; k < sizeof(buf); k++) { h ^= (unsigned char)buf[k]; h *= 1099511628211ull; }
    return h ? h : 1ull;
}

static void dist_teardown(lh_dist_t *d)
{
    if (d->ctl) munmap((void *)d->ctl, 4096);
    if (d->transport == LH_DIST_RCCL && d->comm) (void)g_rccl.CommDestroy(d->comm);
    lh_free_buf(&d->agree);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    free(d);
}

extern "C" int lh_dist_init(lh_dist_t **out, const void *id128, int rank, int world, int device, int transport)
{
    if (!out || !id128) return DFAIL("lh_dist_init: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return DFAIL("lh_dist_init: rank %d of %d", rank, world);
    if (transport != LH_DIST_RCCL && transport != LH_DIST_SHM) return DFAIL("lh_dist_init: unknown transport %d", transport);
    if (device < 0 || device >= lh_device_count()) return DFAIL("lh_dist_init: device %d out of range", device);
    lh_dist_t *d = (lh_dist_t *)calloc(1, sizeof(*d));
    if (!d) return DFAIL("out of memory");
    d->rank = rank; d->world = world; d->device = device; d->transport = transport;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) { d->stream = NULL; dist_teardown(d); return DFAIL("lh_dist_init: cannot use device %d", device); }
    bool one_host = true;                /* the shared-memory transport is one host by construction */
    if (transport == LH_DIST_RCCL) {
        if (rccl_load() != 0) { dist_teardown(d); return -1; }
        ncclUniqueId id; memcpy(&id, id128, LH_DIST_ID_BYTES);
        ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, id, rank);
        if (r != ncclSuccess) { d->comm = NULL; dist_teardown(d); return DFAIL("ncclCommInitRank failed: %s (two ranks on one device? use LH_DIST_SHM)", g_rccl.GetErrorString(r)); }
        /* do all ranks run on ONE host?  (RCCL itself spans nodes; the shared control block below does not: ranks of a second node
         * would wait in its barrier for arrivals that never come.)  Every rank's host id to rank 0 and the table back, over the
         * communicator that has just come up */
        if (world > 1) {
            const size_t tb = sizeof(unsigned long long) * (size_t)world;
            std::vector<unsigned long long> ids((size_t)world, 0ull);
            const unsigned long long mine = host_id();
            int rc = lh_ensure_buf(&d->agree, 2 * tb + 64);
            char *base = (char *)d->agree.p;
            if (rc == 0) rc = hipMemcpyAsync(base, &mine, sizeof(mine), hipMemcpyHostToDevice, d->stream) == hipSuccess ? 0 : -1;
            if (rc == 0) rc = lh_dist_gather(d, base, sizeof(mine), base + 64, (void *)d->stream);
            if (rc == 0) rc = lh_dist_broadcast(d, base + 64, tb, (void *)d->stream);
            if (rc == 0) rc = hipMemcpyAsync(ids.data(), base + 64, tb, hipMemcpyDeviceToHost, d->stream) == hipSuccess ? 0 : -1;
            if (rc == 0) rc = hipStreamSynchronize(d->stream) == hipSuccess ? 0 : -1;
            if (rc != 0) { dist_teardown(d); return DFAIL("lh_dist_init: the ranks could not exchange their host ids over RCCL"); }
            for (int r2 = 0; r2 < world; r2++) if (ids[(size_t)r2] != mine) one_host = false;
        }
    }
    if (one_host) {
        /* the job's control block in shared memory: the shm transport's barrier -- and, for BOTH transports, the host barrier of
         * the ranks of one node (lh_dist_host_barrier: what brackets a timed frame).  Ranks on more than one host (RCCL only) have
         * none: lh_dist_host_barrier falls back to the transport's own barrier there */
        name_from_id(id128, d->shm_name, sizeof(d->shm_name));
        int fd = shm_open(d->shm_name, O_CREAT | O_RDWR, 0600);       /* a fresh segment reads as zeros */
        if (fd < 0 || ftruncate(fd, 4096) != 0) { if (fd >= 0) close(fd); dist_teardown(d); return DFAIL("lh_dist (shm): cannot create %s: %s", d->shm_name, strerror(errno)); }
        d->ctl = (shm_ctl *)mmap(NULL, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); close(fd);
        if ((void *)d->ctl == MAP_FAILED) { d->ctl = NULL; dist_teardown(d); return DFAIL("lh_dist (shm): mmap failed"); }
        if (shm_barrier(d) != 0) { dist_teardown(d); return -1; }
    }
    *out = d;
    return 0;
}

/* plain-C hosts without a launcher (lsh_hip --rank / --world): rank 0 writes the id to `path`, the others poll it; every rank
 * adds a line with its device's bus id, and if two ranks share a device the shared-memory transport is chosen */
extern "C" int lh_dist_init_file(lh_dist_t **out, const char *path, int rank, int world, int device)
{
    if (!out || !path) return DFAIL("lh_dist_init_file: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return DFAIL("lh_dist_init_file: rank %d of %d", rank, world);
    unsigned char id[LH_DIST_ID_BYTES];
    char p[1200];
    if (rank == 0) {
        memset(id, 0, sizeof(id));
        if (rccl_load() == 0) { if (lh_dist_unique_id(id) != 0) return -1; }
        else {                      /* no RCCL on this box: an id for the shared-memory transport */
            unsigned long long t = (unsigned long long)(now_sec() * 1e9) ^ ((unsigned long long)getpid() << 32);
            memcpy(id, &t, sizeof(t));
        }
        snprintf(p, sizeof(p), "%s.tmp", path);
        FILE *f = fopen(p, "wb");
        if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) { if (f) fclose(f); return DFAIL("lh_dist_init_file: cannot write %s", p); }
        fclose(f);
        if (rename(p, path) != 0) return DFAIL("lh_dist_init_file: cannot publish %s", path);
    } else {
        const double t0 = now_sec();
        for (;;) {
            FILE *f = fopen(path, "rb");
            if (f) { const size_t n = fread(id, 1, sizeof(id), f); fclose(f); if (n == sizeof(id)) break; }
            if (now_sec() - t0 > 120.0) return DFAIL("lh_dist_init_file: no id in %s after 120 s", path);
            usleep(2000);
        }
    }
    /* bus ids: one small file per rank next to the id; every rank reads them all and takes the same decision */
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) snprintf(bus, sizeof(bus), "device%d", device);
    snprintf(p, sizeof(p), "%s.rank%d.tmp", path, rank);
    { FILE *f = fopen(p, "w"); if (!f) return DFAIL("lh_dist_init_file: cannot write %s", p); fprintf(f, "%s\n", bus); fclose(f); }
    { char q[1200]; snprintf(q, sizeof(q), "%s.rank%d", path, rank); if (rename(p, q) != 0) return DFAIL("lh_dist_init_file: cannot publish %s", q); }
    std::vector<std::string> ids((size_t)world);
    for (int r = 0; r < world; r++) {
        char other[64] = ""; const double t0 = now_sec();
        snprintf(p, sizeof(p), "%s.rank%d", path, r);
        for (;;) {
            FILE *f = fopen(p, "r");
            if (f) { const int ok = fscanf(f, "%63s", other) == 1; fclose(f); if (ok) break; }
            if (now_sec() - t0 > 120.0) return DFAIL("lh_dist_init_file: rank %d did not show up", r);
            usleep(2000);
        }
        ids[(size_t)r] = other;
    }
    int shared = 0;
    for (int r = 0; r < world; r++) for (int q = r + 1; q < world; q++) if (ids[(size_t)r] == ids[(size_t)q]) shared = 1;
    const char *force = getenv("LH_DIST_TRANSPORT");
    int transport = (shared || rccl_load() != 0) ? LH_DIST_SHM : LH_DIST_RCCL;
    if (force && strcmp(force, "shm") == 0) transport = LH_DIST_SHM;
    if (force && strcmp(force, "rccl") == 0) transport = LH_DIST_RCCL;
    return lh_dist_init(out, id, rank, world, device, transport);
}

extern "C" void lh_dist_destroy(lh_dist_t *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    lh_free_buf(&d->slab); lh_free_buf(&d->mono); lh_free_buf(&d->all); lh_free_buf(&d->frame); lh_free_buf(&d->bands); lh_free_buf(&d->agree);
    if (d->transport == LH_DIST_RCCL && d->comm) (void)g_rccl.CommDestroy(d->comm);
    if (d->ctl) { (void)shm_barrier(d); munmap((void *)d->ctl, 4096); if (d->rank == 0) shm_unlink(d->shm_name); }
    if (d->stream) (void)hipStreamDestroy(d->stream);
    free(d);
}

extern "C" int lh_dist_rank(const lh_dist_t *d) { return d ? d->rank : -1; }
extern "C" int lh_dist_world(const lh_dist_t *d) { return d ? d->world : 0; }
extern "C" int lh_dist_transport(const lh_dist_t *d) { return d ? d->transport : -1; }

/* ---- primitives: ri_parallel_bcast / ri_parallel_gather / ri_parallel_barrier (parallel.c:101-232) ---------------- */
/* device buffer, root 0 */
extern "C" int lh_dist_broadcast(lh_dist_t *d, void *d_buf, size_t bytes, void *stream)
{
    if (!d || (!d_buf && bytes)) return DFAIL("lh_dist_broadcast: NULL argument");
    HIPCHK(hipSetDevice(d->device));
    hipStream_t s = stream ? (hipStream_t)stream : d->stream;
    if (bytes == 0) return 0;
    if (d->transport == LH_DIST_RCCL) {
        NCHK(g_rccl.Broadcast(d_buf, d_buf, bytes, ncclUint8, 0, d->comm, s));
        return 0;
    }
    void *seg = NULL;
    if (shm_data(d, bytes, &seg) != 0) return -1;
    if (d->rank == 0) { HIPCHK(hipMemcpyAsync(seg, d_buf, bytes, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); }
    if (shm_barrier(d) != 0) return -1;
    if (d->rank != 0) { HIPCHK(hipMemcpyAsync(d_buf, seg, bytes, hipMemcpyHostToDevice, s)); HIPCHK(hipStreamSynchronize(s)); }
    return shm_release(d, seg, bytes);
}

/* equal-sized device slabs to rank 0: d_recv (rank 0 only) holds world * bytes, rank r's slab at r * bytes.  N - 1 point-to-point
 * transfers in one group: every peer's slab crosses its own xGMI link, in parallel */
extern "C" int lh_dist_gather(lh_dist_t *d, const void *d_send, size_t bytes, void *d_recv, void *stream)
{
    if (!d || (!d_send && bytes) || (d->rank == 0 && !d_recv && bytes)) return DFAIL("lh_dist_gather: NULL argument");
    HIPCHK(hipSetDevice(d->device));
    hipStream_t s = stream ? (hipStream_t)stream : d->stream;
    if (bytes == 0) return 0;
    if (d->transport == LH_DIST_RCCL) {
        NCHK(g_rccl.GroupStart());
        ncclResult_t rc = ncclSuccess;
        if (d->rank == 0) {
            for (int r = 1; r < d->world && rc == ncclSuccess; r++) rc = g_rccl.Recv((char *)d_recv + (size_t)r * bytes, bytes, ncclUint8, r, d->comm, s);
        } else rc = g_rccl.Send(d_send, bytes, ncclUint8, 0, d->comm, s);
        const ncclResult_t rce = g_rccl.GroupEnd();                     /* the group is closed whatever happened inside it */
        if (rc != ncclSuccess) return DFAIL("lh_dist_gather: ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString(rc));
        if (rce != ncclSuccess) return DFAIL("lh_dist_gather: ncclGroupEnd failed: %s", g_rccl.GetErrorString(rce));
        if (d->rank == 0 && d_recv != d_send) HIPCHK(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, s));     /* the owner's own slab */
        return 0;
    }
    void *seg = NULL;
    if (shm_data(d, bytes * (size_t)d->world, &seg) != 0) return -1;
    HIPCHK(hipMemcpyAsync((char *)seg + (size_t)d->rank * bytes, d_send, bytes, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
    if (shm_barrier(d) != 0) return -1;
    if (d->rank == 0) { HIPCHK(hipMemcpyAsync(d_recv, seg, bytes * (size_t)d->world, hipMemcpyHostToDevice, s)); HIPCHK(hipStreamSynchronize(s)); }
    return shm_release(d, seg, bytes * (size_t)d->world);
}

/* hit records for the wire (lucille_hip.h): one 16-byte store per ray */
__global__ __launch_bounds__(256) void k_pack_records16(size_t n, const uint32_t *__restrict__ prim, const double *__restrict__ t,
                                                        const double *__restrict__ u, const double *__restrict__ v, uint4 *__restrict__ rec)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint4 r;
    r.x = prim[i]; r.y = __float_as_uint((float)t[i]); r.z = __float_as_uint((float)u[i]); r.w = __float_as_uint((float)v[i]);
    rec[i] = r;
}

extern "C" int lh_dist_pack_records16(size_t n, const void *d_prim, const void *d_t, const void *d_u, const void *d_v, void *d_rec16, void *stream)
{
    if (n == 0) return 0;
    if (!d_prim || !d_t || !d_u || !d_v || !d_rec16) return DFAIL("lh_dist_pack_records16: NULL argument");
    if (((uintptr_t)d_rec16 & 15u) != 0) return DFAIL("lh_dist_pack_records16: the record buffer is not 16-byte aligned");
    hipLaunchKernelGGL(k_pack_records16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, (const uint32_t *)d_prim,
                       (const double *)d_t, (const double *)d_u, (const double *)d_v, (uint4 *)d_rec16);
    HIPCHK(hipGetLastError());
    return 0;
}

/* the ranks of ONE node (the contract of bench.py --gpus N and of lsh_hip --world N) meet in shared memory: microseconds, where a
 * gloo barrier between eight processes costs 1.3 ms and lets them out up to 0.4 ms apart (tools/skew_probe.py) -- a tenth of a
 * rank's 12 ms share of the config-5 frame.  Host only: the caller synchronises its device first. */
extern "C" int lh_dist_host_barrier(lh_dist_t *d)
{
    if (!d) return DFAIL("lh_dist_host_barrier: NULL");
    if (!d->ctl) return lh_dist_barrier(d);          /* RCCL ranks on more than one host: the transport's own barrier */
    return shm_barrier(d);
}

extern "C" int lh_dist_barrier(lh_dist_t *d)
{
    if (!d) return DFAIL("lh_dist_barrier: NULL");
    if (d->transport == LH_DIST_SHM) return shm_barrier(d);
    HIPCHK(hipSetDevice(d->device));
    if (lh_ensure_buf(&d->bands, 64 * (size_t)d->world + 64)) return -1;
    /* everybody reports to rank 0 (a one-byte gather), rank 0 answers (a one-byte broadcast) */
    if (lh_dist_gather(d, d->bands.p, 1, (char *)d->bands.p + 64, (void *)d->stream) != 0) return -1;
    NCHK(g_rccl.Broadcast(d->bands.p, d->bands.p, 1, ncclUint8, 0, d->comm, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

/* Every rank contributes ok (1) or not (0); every rank learns whether ALL were ok: a 4-byte gather to rank 0 and a 4-byte
 * broadcast back.  A collective of its own -- every rank must reach it -- placed BEFORE a payload moves, so that a rank that
 * cannot go on (rank 0's commit failed, a receiver is out of memory) tells the others instead of leaving them blocked in the
 * payload's collective: RCCL has no timeout.  -> 1 all ok, 0 somebody failed, -1 the exchange itself failed */
static int dist_agree(lh_dist_t *d, int ok)
{
    if (d->world == 1) return ok ? 1 : 0;
    if (lh_ensure_buf(&d->agree, 64 * (size_t)d->world + 128)) return -1;
    unsigned int mine = ok ? 1u : 0u;
    char *base = (char *)d->agree.p;
    HIPCHK(hipMemcpyAsync(base, &mine, sizeof(mine), hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    if (lh_dist_gather(d, base, 64, d->rank == 0 ? base + 64 : NULL, (void *)d->stream) != 0) return -1;
    unsigned int all = 1u;
    if (d->rank == 0) {
        std::vector<unsigned int> w(16 * (size_t)d->world);
        HIPCHK(hipMemcpyAsync(w.data(), base + 64, 64 * (size_t)d->world, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        for (int r = 0; r < d->world; r++) all &= (w[16 * (size_t)r] != 0u) ? 1u : 0u;
        HIPCHK(hipMemcpyAsync(base, &all, sizeof(all), hipMemcpyHostToDevice, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
    }
    if (lh_dist_broadcast(d, base, 64, (void *)d->stream) != 0) return -1;
    HIPCHK(hipMemcpyAsync(&all, base, sizeof(all), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    return all ? 1 : 0;
}

/* ---- scene load: one build, one broadcast -------------------------------------------------------------- */
/* EVERY rank calls this, rank 0 too when its commit failed (its accel is then not committed): the first thing that travels
 * is a status word, and every rank returns -1 together instead of the receivers waiting in ncclBroadcast for a scene that
 * never comes.  The same after the receivers have allocated: one that is out of memory says so before the arrays move. */
extern "C" int lh_dist_broadcast_scene(lh_dist_t *d, lh_accel_t *accel)
{
    if (!d || !accel) return DFAIL("lh_dist_broadcast_scene: NULL argument");
    HIPCHK(hipSetDevice(d->device));
    lh_scene_image_t h; memset(&h, 0, sizeof(h));
    char why[512]; why[0] = 0;
    int ok = 1;
    if (d->rank == 0 && lh_scene_image_header(accel, &h) != 0) { ok = 0; snprintf(why, sizeof(why), "%s", lh_last_error()); }
    int all = dist_agree(d, ok);
    if (all < 0) return -1;
    if (!all) return ok ? DFAIL("lh_dist_broadcast_scene: rank 0 has no committed scene to send (its commit failed); nothing was broadcast")
                        : DFAIL("lh_dist_broadcast_scene: rank 0 cannot send its scene (%s); the other ranks were told", why);
    /* the header and the two host arrays travel through a device staging buffer */
    if (lh_ensure_buf(&d->bands, sizeof(h) > 256 ? sizeof(h) : 256)) return -1;
    if (d->rank == 0) HIPCHK(hipMemcpy(d->bands.p, &h, sizeof(h), hipMemcpyHostToDevice));
    if (lh_dist_broadcast(d, d->bands.p, sizeof(h), NULL) != 0) return -1;
    HIPCHK(hipStreamSynchronize(d->stream));
    ok = 1;
    if (d->rank != 0) {
        HIPCHK(hipMemcpy(&h, d->bands.p, sizeof(h), hipMemcpyDeviceToHost));
        if (h.magic != 0x4C48494Du) { ok = 0; snprintf(why, sizeof(why), "the scene header arrived damaged"); }
        else if (lh_scene_image_alloc(accel, &h) != 0) { ok = 0; snprintf(why, sizeof(why), "%s", lh_last_error()); }
    }
    all = dist_agree(d, ok);
    if (all < 0) return -1;
    if (!all) return ok ? DFAIL("lh_dist_broadcast_scene: another rank could not allocate the scene; nothing was broadcast")
                        : DFAIL("lh_dist_broadcast_scene: rank %d cannot receive the scene (%s)", d->rank, why);
    void *ptr[32]; size_t bytes[32];
    const int n = lh_scene_image_arrays(accel, &h, ptr, bytes, 32);
    if (n > 32) return DFAIL("lh_dist_broadcast_scene: image table overflow");
    for (int k = 0; k < n; k++) if (lh_dist_broadcast(d, ptr[k], bytes[k], NULL) != 0) return -1;
    if (h.ntris) {
        const size_t b = sizeof(uint32_t) * (size_t)h.ntris;
        if (lh_ensure_buf(&d->slab, 2 * b)) return -1;
        if (d->rank == 0) {
            HIPCHK(hipMemcpy(d->slab.p, lh_scene_image_prim_geom(accel), b, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy((char *)d->slab.p + b, lh_scene_image_prim_index(accel), b, hipMemcpyHostToDevice));
        }
        if (lh_dist_broadcast(d, d->slab.p, 2 * b, NULL) != 0) return -1;
        HIPCHK(hipStreamSynchronize(d->stream));
        if (d->rank != 0) {
            HIPCHK(hipMemcpy(lh_scene_image_prim_geom(accel), d->slab.p, b, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(lh_scene_image_prim_index(accel), (char *)d->slab.p + b, b, hipMemcpyDeviceToHost));
        }
    }
    HIPCHK(hipStreamSynchronize(d->stream));
    if (d->rank != 0 && lh_scene_image_finish(accel) != 0) return -1;
    return 0;
}

/* ---- frames: every rank renders its bands, rank 0 owns the display (render.c:468-514) ----------------------------- */
/* slabs [world][per][rows][W][3] (image orientation inside) -> frame [H][W][3], top row first: band b covers frame lines
 * b * rows ... from the BOTTOM of the image (bucket_write's y flip, render.c:962-964).  The bands are dealt out in groups of
 * `world`; band g * world + pos is the g-th band of the rank whose position in group g is pos (band_pos: rank order, reversed,
 * shifted, shifted and reversed).  Plain interleaving hands the last rank the lower band of EVERY group: where the cost of a
 * line changes steadily down the image that rank carries the whole slope (config 5, 64-line bands: 7 %); this order cancels a
 * linear slope exactly and most of the curvature, so the bands can be tall -- and coherent.  Same rule: lucille_amd/shard.py */
/* the position of rank r's band inside group g, and its inverse: rank order, reversed, shifted by half the world, shifted and
 * reversed, and again (lucille_amd/shard.py _band_pos: rank order + reversed cancel a steady change of cost down the image,
 * the shifted pair most of its curvature) */
__host__ __device__ static inline int band_pos(int g, int r, int world)
{
    const int s = (r + world / 2) % world;
    switch (g & 3) { case 0: return r; case 1: return world - 1 - r; case 2: return s; default: return world - 1 - s; }
}
__host__ __device__ static inline int band_rank(int g, int pos, int world)
{
    const int q = (g & 1) ? world - 1 - pos : pos;                 /* undo the reversal */
    return (g & 2) ? (q + world - world / 2) % world : q;          /* undo the shift */
}

/* ch: floats per pixel of the slabs -- 3 (RGB), or 1: an AO frame is grey (Lo = (N - occluded) / N in every channel,
 * ambientocclusion.c:383-401), so the exchange step moves ONE float per pixel and the owner of the display writes it three times (k_take_count8 below: one BYTE
 * when the frame has one sample per pixel) */
__global__ void k_take_channel0(size_t npix, const float *__restrict__ rgb, float *__restrict__ mono)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < npix) mono[i] = rgb[3 * i];
}

/* ... and with ONE sample per pixel that float is (N - occluded) / N for an integer N - occluded <= N (k_ao_resolve, lh_render.hip:
 * `(float)(1.0 * (ns - occlusion) / ns)`, 0 for a miss): ONE BYTE per pixel travels when N <= 255 -- the count, recovered exactly from the float
 * (it is within 1e-7 N of an integer) -- and the owner of the display evaluates the same expression on it: the same bits (round 6:
 * the exchange of a 4096^2 frame at 8 ranks 58.7 -> 14.7 MB into rank 0) */
__global__ void k_take_count8(size_t npix, const float *__restrict__ rgb, uint8_t *__restrict__ cnt, float N)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < npix) cnt[i] = (uint8_t)(int)(rgb[3 * i] * N + 0.5f);
}

/* ch: floats per pixel of the slabs (3, 1), or 0: one byte per pixel, the count of k_take_count8 over N = nsamp */
__global__ void k_place_bands(const float *__restrict__ slabs, float *__restrict__ frame, int world, int per, int rows, int W, int H, int ch, int nsamp = 0)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const int x = (int)(i % W), row = (int)(i / W);            /* row 0 = top of the image = frame line H - 1 */
    const int line = H - 1 - row, band = line / rows, k = band / world, pos = band % world, r = band_rank(k, pos, world);
    const int y0 = band * rows, h = (y0 + rows <= H) ? rows : H - y0;
    /* inside a band slab the first frame line of the band is the LAST of its h lines; a clipped band keeps them at the bottom */
    const int srow = (rows - h) + (h - 1 - (line - y0));
    float *dst = frame + i * 3;
    if (ch == 0) {
        const uint8_t c = ((const uint8_t *)slabs)[(((size_t)r * per + k) * rows + srow) * W + x];
        const double ns = (double)(uint32_t)nsamp;
        const float f = (float)(1.0 * (double)c / ns);                 /* k_ao_resolve's expression for ns - occlusion = c */
        dst[0] = f; dst[1] = f; dst[2] = f;
        return;
    }
    const float *src = slabs + ((((size_t)r * per + k) * rows + srow) * W + x) * (size_t)ch;
    dst[0] = src[0]; dst[1] = ch == 3 ? src[1] : src[0]; dst[2] = ch == 3 ? src[2] : src[0];
}

extern "C" int lh_dist_render_ao_frame_host(lh_dist_t *d, lh_accel_t *accel, const lh_camera_t *cam, int pixel_samples,
                                            int gather_nsamples, uint64_t seed, int band_rows, float *rgb, lh_tile_stats_t *stats)
{
    if (!d || !accel || !cam) return DFAIL("lh_dist_render_ao_frame_host: NULL argument");
    if (d->rank == 0 && !rgb) return DFAIL("lh_dist_render_ao_frame_host: rank 0 needs the frame buffer");
    const int W = cam->width, H = cam->height;
    if (W <= 0 || H <= 0) return DFAIL("lh_dist_render_ao_frame_host: bad resolution");
    HIPCHK(hipSetDevice(d->device));
    if (band_rows <= 0) band_rows = LH_DIST_BAND_ROWS;
    if (d->world == 1) band_rows = H;
    if (band_rows > H) band_rows = H;
    const int nbands = (H + band_rows - 1) / band_rows, per = (nbands + d->world - 1) / d->world;
    std::vector<int> y0;
    for (int k = 0; k < per; k++) {             /* this rank's band of every group (k_place_bands) */
        const int b = k * d->world + band_pos(k, d->rank, d->world);
        if (b < nbands) y0.push_back(b * band_rows);
    }
    const size_t slab_bytes = (size_t)per * band_rows * W * 3 * sizeof(float);
    lh_tile_stats_t st; memset(&st, 0, sizeof(st));
    /* a rank that cannot render its bands (or rank 0 without room for the slabs) says so before the gather: the others return
     * with it instead of waiting in ncclRecv / ncclSend */
    int ok = lh_ensure_buf(&d->slab, slab_bytes) == 0 && hipMemsetAsync(d->slab.p, 0, slab_bytes, d->stream) == hipSuccess;
    if (!ok) lh_fail("lh_dist_render_ao_frame_host: no room for this rank's slab (%zu bytes)", slab_bytes);
    if (ok) ok = lh_render_ao_bands(accel, cam, (int)y0.size(), y0.data(), band_rows, pixel_samples, gather_nsamples, seed, d->slab.p, &st, (void *)d->stream) == 0;
    char why[512]; why[0] = 0;
    if (!ok) snprintf(why, sizeof(why), "%s", lh_last_error());
    /* what travels: one float per pixel (k_take_channel0) -- or, with one sample per pixel and at most 255 AO rays per hit, one BYTE: the
     * number of unoccluded rays (k_take_count8); LH_DIST_AO_BYTES=4 keeps the float */
    const int nphi_ = (int)sqrt((double)gather_nsamples), nsamp = nphi_ * nphi_;          /* lh_tile.hip ao_region, ambientocclusion.c:378-380 */
    static const bool want8 = !(getenv("LH_DIST_AO_BYTES") && atoi(getenv("LH_DIST_AO_BYTES")) == 4);
    const bool count8 = want8 && pixel_samples == 1 && nsamp >= 1 && nsamp <= 255;
    const size_t slab_px = (size_t)per * band_rows * W, mono_bytes = slab_px * (count8 ? 1 : sizeof(float));
    if (ok && lh_ensure_buf(&d->mono, mono_bytes)) { ok = 0; snprintf(why, sizeof(why), "%s", lh_last_error()); }
    if (ok) {
        if (count8) hipLaunchKernelGGL(k_take_count8, dim3((unsigned)((slab_px + 255) / 256)), dim3(256), 0, d->stream, slab_px, (const float *)d->slab.p, (uint8_t *)d->mono.p, (float)nsamp);
        else hipLaunchKernelGGL(k_take_channel0, dim3((unsigned)((slab_px + 255) / 256)), dim3(256), 0, d->stream, slab_px, (const float *)d->slab.p, (float *)d->mono.p);
        if (hipGetLastError() != hipSuccess) { ok = 0; snprintf(why, sizeof(why), "k_take_channel0 / k_take_count8 launch failed"); }
    }
    if (ok && d->rank == 0 && lh_ensure_buf(&d->all, mono_bytes * (size_t)d->world)) { ok = 0; snprintf(why, sizeof(why), "%s", lh_last_error()); }
    const int all_ok = dist_agree(d, ok);
    if (all_ok < 0) return -1;
    if (!all_ok) return ok ? DFAIL("lh_dist_render_ao_frame_host: another rank could not render its bands; no frame") : DFAIL("lh_dist_render_ao_frame_host: rank %d: %s", d->rank, why);
    if (lh_dist_gather(d, d->mono.p, mono_bytes, d->rank == 0 ? d->all.p : NULL, (void *)d->stream) != 0) return -1;
    /* statistics: the sum over the ranks (four 64-bit counters through the same gather) -- ALWAYS, whether or not this rank's
     * caller asked for them: a collective every rank must enter */
    {
        if (lh_ensure_buf(&d->bands, 64 * (size_t)d->world + 64)) return -1;
        unsigned long long mine[4] = {st.primary_rays, st.primary_hits, st.ao_rays, st.ao_occluded};
        HIPCHK(hipMemcpyAsync(d->bands.p, mine, sizeof(mine), hipMemcpyHostToDevice, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        if (lh_dist_gather(d, d->bands.p, 32, d->rank == 0 ? (char *)d->bands.p + 64 : NULL, (void *)d->stream) != 0) return -1;
        if (d->rank == 0) {
            std::vector<unsigned long long> all(4 * (size_t)d->world);
            HIPCHK(hipMemcpyAsync(all.data(), (char *)d->bands.p + 64, 32 * (size_t)d->world, hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipStreamSynchronize(d->stream));
            if (stats) {
                memset(stats, 0, sizeof(*stats));
                for (int r = 0; r < d->world; r++) {
                    stats->primary_rays += all[4 * r]; stats->primary_hits += all[4 * r + 1]; stats->ao_rays += all[4 * r + 2]; stats->ao_occluded += all[4 * r + 3];
                }
            }
        } else if (stats) *stats = st;
    }
    if (d->rank == 0) {
        const size_t fb = (size_t)W * H * 3 * sizeof(float);
        if (lh_ensure_buf(&d->frame, fb)) return -1;
        const size_t px = (size_t)W * H;
        hipLaunchKernelGGL(k_place_bands, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, d->stream, (const float *)d->all.p, (float *)d->frame.p,
                           d->world, per, band_rows, W, H, count8 ? 0 : 1, nsamp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(rgb, d->frame.p, fb, hipMemcpyDeviceToHost, d->stream));
    }
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}
