/*
 * lh_query.hip -- the ray queries of the C ABI: accel_intersect_func (src/render/accel.h:30-34) as device-resident, host
 * and pipelined host batches, one synchronous ray, traversal statistics (src/render/bvh.c:669-706) and beam visibility
 * (ri_bvh_intersect_beam_visibility, bvh.c:612-667).  The kernels are in lh_kernels.hip / lh_beam.hip.
 */
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "lh_internal.h"

/* fill miss results without touching the scene (empty accel) */
__global__ void k_fill_miss(size_t n, uint32_t *prim, double *t, double *u, double *v, uint8_t *occ)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (prim) prim[i] = LH_MISS_PRIM;
    if (t) t[i] = LH_T_INF;
    if (u) u[i] = 0.0;
    if (v) v[i] = 0.0;
    if (occ) occ[i] = 0;
}

/* the fix-up queue for launches on `s`: launches on one stream are ordered, so they share a slot (queue, second stream,
 * events); different streams (replicas' tile loops, the pipelined host path) get their own */
int lh_aoq_slot(lh_accel_t *a, hipStream_t s)
{
    int k, free_k = -1;
    for (k = 0; k < LH_AOQ_SLOTS; k++) {
        if (a->aoq[k].used && a->aoq[k].stream == s) return k;
        if (!a->aoq[k].used && free_k < 0) free_k = k;
    }
    if (free_k < 0) {
        /* more concurrent streams than slots: wait for the device, then recycle slot 0 */
        HIPCHK(hipDeviceSynchronize());
        free_k = 0;
    }
    k = free_k;
    lh_fixq_t *q = &a->aoq[k].q;
    if (!q->ev_done) {              /* ev_done is set last: a slot whose set-up failed half-way is not "ready" */
        void *queue = NULL; uint32_t *qcount = NULL; hipStream_t aux = NULL; hipEvent_t e0 = NULL, e1 = NULL;
        bool ok = hipMalloc(&queue, (size_t)LH_AO_QCAP * sizeof(unsigned long long)) == hipSuccess &&
                  hipMalloc((void **)&qcount, (4 + 4096) * sizeof(uint32_t)) == hipSuccess &&       /* counters + the consumer groups' heads (LH_Q_GROUPS) */
                  hipMemset(queue, 0, (size_t)LH_AO_QCAP * sizeof(unsigned long long)) == hipSuccess &&
                  hipMemset(qcount, 0, (4 + 4096) * sizeof(uint32_t)) == hipSuccess;
        /* the consumer's stream gets the highest priority: streams of one priority share a handful of hardware queues round
         * robin (four by default), and a consumer that lands on its producer's queue runs BEHIND it instead of next to it
         * (bench.py's second accelerator: config-5 frame 89 -> 122 ms); priority streams have queues of their own */
        if (ok) {
            int prio_lo = 0, prio_hi = 0;
            if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess || prio_hi == prio_lo ||
                hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, prio_hi) != hipSuccess)
                ok = hipStreamCreateWithFlags(&aux, hipStreamNonBlocking) == hipSuccess;
        }
        ok = ok && hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess;
        if (!ok) {                  /* nothing half-built stays behind */
            const hipError_t e = hipGetLastError();
            if (e1) (void)hipEventDestroy(e1);
            if (e0) (void)hipEventDestroy(e0);
            if (aux) (void)hipStreamDestroy(aux);
            if (qcount) (void)hipFree(qcount);
            if (queue) (void)hipFree(queue);
            return fail("the fix-up queue of a launch stream could not be set up: %s", hipGetErrorString(e));
        }
        q->queue = queue; q->qcount = qcount; q->qcap = LH_AO_QCAP;
        q->aux_stream = (void *)aux; q->ev_ready = (void *)e0; q->ev_done = (void *)e1;
    }
    a->aoq[k].stream = s; a->aoq[k].used = 1;
    return k;
}

/* hot set of a ray dump (4-wide nodes + 48-byte triangle records) against the Infinity Cache */
static bool wide8_pays(const lh_accel_t *a)
{
    const lh_bvh_t *b = &a->hs->bvh;
    return sizeof(lh_q4node_t) * (size_t)b->nq4nodes + sizeof(lh_tri32_t) * (size_t)b->ntris > ((size_t)256 << 20);
}

/* the 8-wide nodes exist or can be made: always for a host-built tree (lh_bvh_ensure_q8), by a second run of the build for a
 * tree built on this device (lh_ensure_formats), never for a scene received from another rank without them */
static bool q8_available(const lh_accel_t *a)
{
    if (a->d_q8nodes || !a->hs->device_built) return true;
    return !a->hs->received && a->hs->bvh.nq4nodes > 1;
}

int lh_launch(lh_accel_t *a, size_t n, const void *d_org, const void *d_dir, void *d_prim,
              void *d_t, void *d_u, void *d_v, void *d_occ, int mode, int variant,
              unsigned long long *d_counters, hipStream_t s, bool dump)
{
    if (!a || !a->committed) return fail("intersect: accel not committed");
    if (n == 0) return 0;
    if ((!d_org || !d_dir) && !a->dev.cam_src) return fail("intersect: NULL ray arrays");
    if (mode == LH_MODE_CLOSEST && (!d_prim || !d_t || !d_u || !d_v)) return fail("intersect: closest mode needs prim,t,u,v outputs");
    if (mode == LH_MODE_ANY && !d_occ) return fail("intersect: any mode needs the occluded output");
    if (mode != LH_MODE_CLOSEST && mode != LH_MODE_ANY) return fail("intersect: unknown mode %d", mode);
    HIPCHK(hipSetDevice(a->device));
    if (a->hs->bvh.ntris == 0) {
        size_t blocks = (n + 255) / 256;
        hipLaunchKernelGGL(k_fill_miss, dim3((unsigned)blocks), dim3(256), 0, s, n,
                           mode == LH_MODE_CLOSEST ? (uint32_t *)d_prim : NULL, (double *)(mode == LH_MODE_CLOSEST ? d_t : NULL),
                           (double *)(mode == LH_MODE_CLOSEST ? d_u : NULL), (double *)(mode == LH_MODE_CLOSEST ? d_v : NULL),
                           mode == LH_MODE_ANY ? (uint8_t *)d_occ : NULL);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (dump && a->poison_outputs) {          /* LH_POISON_OUTPUTS: every answer slot must be written by the launch */
        if (mode == LH_MODE_CLOSEST) {
            HIPCHK(hipMemsetAsync(d_prim, 0x77, n * sizeof(uint32_t), s)); HIPCHK(hipMemsetAsync(d_t, 0x77, n * sizeof(double), s));
            HIPCHK(hipMemsetAsync(d_u, 0x77, n * sizeof(double), s)); HIPCHK(hipMemsetAsync(d_v, 0x77, n * sizeof(double), s));
        } else HIPCHK(hipMemsetAsync(d_occ, 0x77, n, s));
    }
    /* a device-built scene: lucille's own tree (exact-t tie winners, fragile hits) is built by a background host thread.
     * Queries are exact by default -- the first launch waits for it; set_param("fast_start", 1) / LH_FAST_START=1 launches
     * at once and attaches the tree when it is ready (until then ties resolve to the larger primitive id) */
    if (a->hs->device_built && !a->d_ref_nodes && lh_sync_ref(a, !a->fast_start) != 0) return -1;
    if (variant == LH_VARIANT_DEFAULT) variant = a->default_variant;
    if (variant != LH_VARIANT_DIRECT && variant != LH_VARIANT_SPEC)
        return fail("intersect: unknown variant %d (%d: the textbook reference walk, %d: the default)", variant, LH_VARIANT_DIRECT, LH_VARIANT_SPEC);
    /* a tree built on the device exists only as 4-wide nodes: the textbook walk runs as the default walk there */
    if (a->hs->device_built) variant = LH_VARIANT_SPEC;
    if (lh_ensure_formats(a, lh_trace_formats_needed(&a->dev, variant)) != 0) return -1;     /* the textbook walk's nodes: uploaded on first use */
    /* ray dumps (incoherent by assumption) over a scene whose hot set does not fit the 256 MiB Infinity Cache walk the 8-wide
     * nodes: each record then costs a 128-byte line of HBM traffic whatever its size, and an 8-wide record uses all of it
     * (S-soup-10M: 57 -> 40 records per ray).  The tile pipelines' coherent rays stay on the 4-wide nodes. */
    a->dev.prefer_q8 = 0;
    if (dump && variant == LH_VARIANT_SPEC && q8_available(a) && (a->wide8 == 1 || (a->wide8 == -1 && wide8_pays(a)))) {
        if (lh_ensure_formats(a, LH_FMT_Q8) != 0) return -1;
        a->dev.prefer_q8 = 1;
    }
    /* ray dumps are incoherent by assumption: a wave's iterations serve 64 unrelated rays, a ray's age in iterations runs to
     * several times its own steps -- the tile pipelines' budget (128) would send half the batch to the cooperative walk */
    const uint32_t budget_keep = a->dev.ray_budget, chunk_keep = a->dev.ray_chunk;
    if (dump) a->dev.ray_budget = a->dump_budget;
    else if (a->dev.ray_chunk < LH_TILE_CHUNK) a->dev.ray_chunk = LH_TILE_CHUNK;       /* the tile pipelines' batches are coherent in batch order */
    const uint32_t cap_keep = a->dev.stack_cap; const int grid = a->grid_blocks;
    const int qk = lh_aoq_slot(a, s);             /* the stream's fix-up queue (rays out of visit budget -> the cooperative walk) */
    if (qk < 0) { a->dev.ray_budget = budget_keep; a->dev.ray_chunk = chunk_keep; a->dev.stack_cap = cap_keep; return -1; }
    /* a ray dump over the 4-wide nodes regroups a little later and passes over parked leaves a little sooner than the tile
     * pipelines' coherent batches want (tools/experiments/knob_sweep3.py / knob_sweep4.py, r05: S-soup-1M 2 233 -> 2 266 Mrays/s
     * closest hit, 2 735 -> 2 772 any hit; the 8-wide walk and the AO stage are best where they are) */
    const bool dump4 = dump && !a->dev.prefer_q8 && !a->knobs_user;
    const bool dump8 = dump && a->dev.prefer_q8 && !a->knobs_user;
    int rc = lh_launch_trace(&a->dev, n, (const double *)d_org, (const double *)d_dir, (uint32_t *)d_prim,
                             (double *)d_t, (double *)d_u, (double *)d_v, mode == LH_MODE_ANY,
                             (uint8_t *)d_occ, d_counters, (unsigned long long *)((uint32_t *)a->d_cursor + (size_t)LH_CURSOR_WORDS * (a->cursor_next++ % LH_NCURSOR)), variant, grid,
                             dump4 ? LH_DUMP_MIN_ACTIVE : dump8 ? LH_DUMP8_MIN_ACTIVE : a->min_active, dump4 ? LH_DUMP_TRI_BATCH : dump8 ? LH_DUMP8_TRI_BATCH : a->tri_batch,
                             &a->aoq[qk].q, a->ncus, (void *)s);
    a->dev.ray_budget = budget_keep; a->dev.ray_chunk = chunk_keep; a->dev.stack_cap = cap_keep;
    if (rc != 0) return fail("kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int lh_accel_intersect_device(lh_accel_t *a, size_t n, const void *d_org, const void *d_dir,
                                         void *d_prim, void *d_t, void *d_u, void *d_v, void *d_occ,
                                         int mode, int variant, void *stream)
{
    lh_guard guard(a);
    return lh_launch(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occ, mode, variant, NULL, (hipStream_t)stream, true);
}

extern "C" int lh_accel_intersect_device_counted(lh_accel_t *a, size_t n, const void *d_org, const void *d_dir,
                                                 void *d_prim, void *d_t, void *d_u, void *d_v, void *d_occ,
                                                 int mode, int variant, uint64_t counters[4])
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("intersect: accel not committed");
    if (!counters) return fail("intersect_counted: counters is NULL");
    HIPCHK(hipSetDevice(a->device));
    HIPCHK(hipMemsetAsync(a->d_counters, 0, sizeof(unsigned long long) * LH_CNT_DEV, a->stream));
    HIPCHK(hipDeviceSynchronize());
    if (a->hs->bvh.ntris == 0) { counters[0] = counters[1] = counters[2] = 0; counters[3] = n; }
    int rc = lh_launch(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occ, mode, variant, a->d_counters, a->stream, true);
    if (rc != 0) return rc;
    HIPCHK(hipStreamSynchronize(a->stream));
    if (a->hs->bvh.ntris) {
        unsigned long long h[LH_CNT_DEV];
        HIPCHK(hipMemcpy(h, a->d_counters, sizeof(h), hipMemcpyDeviceToHost));
        for (int k = 0; k < LH_CNT_N; k++) counters[k] = h[k];
        a->last_retraced = h[LH_CNT_RETRACED];
        if (getenv("LH_DEBUG_COUNTERS"))
            fprintf(stderr, "[lucille_hip] lane slots: node steps %llu of %llu, triangle steps %llu of %llu, regroup iterations %llu; "
                            "rays through the reference walk %llu\n",
                    h[LH_CNT_NODES], h[LH_CNT_NODE_SLOTS], h[LH_CNT_TRIS], h[LH_CNT_TRI_SLOTS], h[LH_CNT_REGROUP_SLOTS], h[LH_CNT_RETRACED]);
    }
    return 0;
}

/* bytes of the node record a ray dump walks on this scene: 128 when the 8-wide nodes are in use (hot set beyond the
 * Infinity Cache, or "wide8" forced), else the default format's */
extern "C" int lh_accel_dump_node_bytes(const lh_accel_t *a)
{
    if (!a || !a->committed || a->hs->bvh.ntris == 0) return 0;
    if (a->default_variant == LH_VARIANT_SPEC && q8_available(a) &&
        (a->wide8 == 1 || (a->wide8 == -1 && wide8_pays(a)))) return (int)sizeof(lh_q8node_t);
    return 64;
}

extern "C" uint64_t lh_accel_last_retraced(const lh_accel_t *a) { return a ? a->last_retraced : 0; }

int lh_ensure_stage(lh_accel_t *a, size_t bytes)
{
    if (a->stage_bytes >= bytes) return 0;
    if (a->d_stage) { (void)hipFree(a->d_stage); a->d_stage = NULL; a->stage_bytes = 0; }
    HIPCHK(hipMalloc(&a->d_stage, bytes));
    a->stage_bytes = bytes;
    return 0;
}

/* ---- large host batches: chunks pipelined through pinned staging -----------------------------------
 * A pageable hipMemcpy moves ~12 GB/s; the link does ~50.  Rays are cut into chunks of `cap` rays; chunk k is copied into pinned
 * block k % depth by a pool of host threads, sent, traced and brought back on stream k & 1 while the host stages the next chunks
 * and un-stages the finished ones.
 *
 * Rounds 2-5: two blocks of 2 M rays, whole chunks on two alternating streams, eight threads spawned per copy: 520 Mrays/s closest hit,
 * 640 any hit (S-soup-1M, 20 M rays) where the link alone does 57 GB/s each way = 1 190 Mrays/s of 48-byte rays.  Round 6 looked at the
 * timeline (profiles/r06_hostpath.txt): the host's copies were NOT the bound (3 .. 12 threads, memcpy or streaming stores: the same) --
 * the two streams ran in step, and streams that share one of the runtime's four hardware queues run one after the other.  Now: a stream
 * per direction of the link, a ring of `depth` blocks, the pipeline's streams on hardware queues of their own (the low-priority pool),
 * a pool of copy threads that lives with the process (stage + un-stage of an iteration as one set of slices): 845 / 1 030 Mrays/s.
 * LH_PIPE_CHUNK (rays), LH_PIPE_DEPTH (2 .. 8), LH_COPY_THREADS, LH_PIPE_PRIORITY, LH_PIPE_DIAG=1 (the calling thread's time). */
#define LH_PIPE_CHUNK_DEFAULT ((size_t)1 << 21)
#define LH_PIPE_DEPTH_DEFAULT 3
#define LH_PIPE_MIN   ((size_t)1 << 21)   /* below it the plain path (pageable copies, one launch) is as fast or faster: profiles/r06_hostpath.txt 7 */
#define LH_COPY_SLICE ((size_t)1 << 20)

namespace {
struct CopyJob { char *dst; const char *src; size_t bytes; std::atomic<size_t> *left; };
struct CopyPool {
    std::mutex mu; std::condition_variable cv;
    std::deque<CopyJob> q;
    int nthreads;
};
CopyPool *g_pool;
std::once_flag g_pool_once;

bool pool_take(CopyPool *P, CopyJob &j, bool wait)
{
    std::unique_lock<std::mutex> lk(P->mu);
    if (wait) P->cv.wait(lk, [&] { return !P->q.empty(); });
    if (P->q.empty()) return false;
    j = P->q.front(); P->q.pop_front();
    return true;
}

void pool_worker(CopyPool *P)
{
    for (;;) {
        CopyJob j;
        pool_take(P, j, true);
        memcpy(j.dst, j.src, j.bytes);
        j.left->fetch_sub(1, std::memory_order_release);
    }
}

/* the pool lives as long as the process: its threads sleep on the queue and are never joined (a library that may be dlclose()d
 * at exit must not run thread joins from a static destructor) */
CopyPool *copy_pool()
{
    std::call_once(g_pool_once, [] {
        CopyPool *P = new CopyPool();
        const unsigned hw = std::thread::hardware_concurrency();
        int nt = hw >= 16 ? 8 : (hw >= 4 ? 3 : 0);             /* + the calling thread */
        const char *e = getenv("LH_COPY_THREADS");
        if (e && atoi(e) >= 0 && atoi(e) <= 64) nt = atoi(e);
        P->nthreads = nt;
        for (int k = 0; k < nt; k++) std::thread(pool_worker, P).detach();
        g_pool = P;
    });
    return g_pool;
}

struct CopySet {                      /* copies submitted together; run() returns when all of them are done */
    std::vector<CopyJob> jobs;
    std::atomic<size_t> left{0};
    void add(void *dst, const void *src, size_t bytes)
    {
        for (size_t b = 0; b < bytes; b += LH_COPY_SLICE)
            jobs.push_back({(char *)dst + b, (const char *)src + b, bytes - b < LH_COPY_SLICE ? bytes - b : LH_COPY_SLICE, &left});
    }
    void run()
    {
        if (jobs.empty()) return;
        CopyPool *P = copy_pool();
        left.store(jobs.size(), std::memory_order_relaxed);
        if (P->nthreads == 0) { for (auto &j : jobs) memcpy(j.dst, j.src, j.bytes); jobs.clear(); return; }
        { std::lock_guard<std::mutex> lk(P->mu); for (auto &j : jobs) P->q.push_back(j); }
        P->cv.notify_all();
        CopyJob j;                    /* the caller copies too (any set's slices), then waits for its own set's stragglers */
        while (left.load(std::memory_order_acquire) != 0 && pool_take(P, j, false)) { memcpy(j.dst, j.src, j.bytes); j.left->fetch_sub(1, std::memory_order_release); }
        while (left.load(std::memory_order_acquire) != 0) LH_CPU_RELAX();
        jobs.clear();
    }
};
}  // namespace

static int pipe_init(lh_accel_t *a)
{
    if (a->pipe.ready) return 0;
    size_t C = LH_PIPE_CHUNK_DEFAULT; int depth = LH_PIPE_DEPTH_DEFAULT;
    const char *e = getenv("LH_PIPE_CHUNK");
    if (e && atoll(e) >= (1 << 16) && atoll(e) <= (1 << 24)) C = (size_t)atoll(e);
    e = getenv("LH_PIPE_DEPTH");
    if (e && atoi(e) >= 2 && atoi(e) <= LH_PIPE_DEPTH_MAX) depth = atoi(e);
    const size_t in_b = sizeof(double) * 6 * C, out_b = (sizeof(double) * 3 + sizeof(uint32_t)) * C;
    for (int b = 0; b < depth; b++) {
        HIPCHK(hipHostMalloc(&a->pipe.h_in[b], in_b, hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&a->pipe.h_out[b], out_b, hipHostMallocDefault));
        HIPCHK(hipMalloc(&a->pipe.d_in[b], in_b));
        HIPCHK(hipMalloc(&a->pipe.d_out[b], out_b));
        HIPCHK(hipEventCreateWithFlags(&a->pipe.in_done[b], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&a->pipe.done[b], hipEventDisableTiming));
    }
    /* the three streams come from the LOW-priority pool of hardware queues: the runtime maps the streams of one priority onto four
     * hardware queues (GPU_MAX_HW_QUEUES), this process has more than four at normal priority by now (the caller's, the accelerator's,
     * other accelerators') and the fix-up queues' consumer streams fill the high-priority pool (lh_aoq_slot) -- two of the pipeline's
     * stages on ONE in-order hardware queue run one after the other whatever the streams say (closest hit 660 Mrays/s; with the stages
     * on queues of their own 850: profiles/r06_hostpath.txt).  LH_PIPE_PRIORITY = -1 / 0 / 1: high / normal / low */
    int prio_least = 0, prio_greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    int prio = prio_least;
    e = getenv("LH_PIPE_PRIORITY");
    if (e) prio = atoi(e) < 0 ? prio_greatest : (atoi(e) > 0 ? prio_least : 0);
    for (int b = 0; b < 3; b++) HIPCHK(hipStreamCreateWithPriority(&a->pipe.s[b], hipStreamNonBlocking, prio));
    a->pipe.cap = C; a->pipe.depth = depth; a->pipe.ready = 1;
    return 0;
}

static int intersect_host_pipelined(lh_accel_t *a, size_t n, const double *org, const double *dir,
                                    uint32_t *prim, double *t, double *u, double *v, uint8_t *occ, int mode)
{
    if (pipe_init(a) != 0) return -1;
    /* CAP: rays a staging block holds (the arrays inside it are CAP apart); C: rays per chunk of THIS batch -- a batch of less than four
     * blocks is cut in four, so that its upload, launch and download overlap too (2 / 3 / 6 pieces: no better, tools/hostpath_sizes.py) */
    const size_t CAP = a->pipe.cap, D = (size_t)a->pipe.depth;
    size_t C = CAP;
    if (n < 4 * CAP) { C = ((n + 3) / 4 + 4095) & ~(size_t)4095; if (C < ((size_t)1 << 16)) C = (size_t)1 << 16; if (C > CAP) C = CAP; }
    const size_t nchunks = (n + C - 1) / C;
    CopySet cs;
    auto unstage = [&](size_t k) {
        const size_t b = k % D, first = k * C, m = (first + C <= n) ? C : n - first;
        const char *ho = (const char *)a->pipe.h_out[b];
        if (mode == LH_MODE_CLOSEST) {
            if (t) cs.add(t + first, ho, sizeof(double) * m);
            if (u) cs.add(u + first, ho + sizeof(double) * CAP, sizeof(double) * m);
            if (v) cs.add(v + first, ho + 2 * sizeof(double) * CAP, sizeof(double) * m);
            if (prim) cs.add(prim + first, ho + 3 * sizeof(double) * CAP, sizeof(uint32_t) * m);
        } else if (occ) cs.add(occ + first, ho, m);
    };
    size_t unstaged = 0;              /* chunks [0, unstaged) are back in the caller's arrays */
    static const bool diag = getenv("LH_PIPE_DIAG") != NULL;          /* where the calling thread's time goes: waits / copies / enqueues, ms */
    double t_wait = 0, t_copy = 0, t_enq = 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (size_t k = 0; k < nchunks; k++) {
        const size_t b = k % D, first = k * C, m = (first + C <= n) ? C : n - first;
        /* block b is free once chunk k - depth has been un-staged: wait for that one; take along every later chunk that has come back */
        double c0 = diag ? now() : 0;
        if (k >= D) { HIPCHK(hipEventSynchronize(a->pipe.done[b])); }
        double c1 = diag ? now() : 0; t_wait += c1 - c0;
        while (unstaged < k && (unstaged + D <= k || hipEventQuery(a->pipe.done[unstaged % D]) == hipSuccess)) unstage(unstaged++);
        (void)hipGetLastError();          /* a query that says "not ready" must not be the next launch check's last error */
        char *hi = (char *)a->pipe.h_in[b], *di = (char *)a->pipe.d_in[b], *dout = (char *)a->pipe.d_out[b];
        cs.add(hi, org + 3 * first, sizeof(double) * 3 * m);
        cs.add(hi + sizeof(double) * 3 * CAP, dir + 3 * first, sizeof(double) * 3 * m);
        cs.run();
        double c2 = diag ? now() : 0; t_copy += c2 - c1;
        /* the rays go up on a stream of their own; trace + records down alternate between two more: chunk k + 1's rays cross the link while
         * chunk k is traced, and its launch may start while chunk k's records go back (the runtime moves those with a copy KERNEL, which
         * shares the chip with the launch).  Rounds 2-5 ran whole chunks -- up, trace, down -- on two alternating streams: both fell into
         * step (two uploads sharing the link, then two launches sharing the chip, then two downloads) and nothing overlapped:
         * profiles/r06_hostpath.txt */
        hipStream_t s_in = a->pipe.s[0], s_tr = a->pipe.s[1 + (k & 1)];
        if (m == CAP) HIPCHK(hipMemcpyAsync(di, hi, sizeof(double) * 6 * CAP, hipMemcpyHostToDevice, s_in));
        else {
            HIPCHK(hipMemcpyAsync(di, hi, sizeof(double) * 3 * m, hipMemcpyHostToDevice, s_in));
            HIPCHK(hipMemcpyAsync(di + sizeof(double) * 3 * CAP, hi + sizeof(double) * 3 * CAP, sizeof(double) * 3 * m, hipMemcpyHostToDevice, s_in));
        }
        HIPCHK(hipEventRecord(a->pipe.in_done[b], s_in));
        HIPCHK(hipStreamWaitEvent(s_tr, a->pipe.in_done[b], 0));
        double *d_t = (double *)dout, *d_u = d_t + CAP, *d_v = d_u + CAP; uint32_t *d_prim = (uint32_t *)(d_v + CAP);
        const int rc = lh_launch(a, m, di, di + sizeof(double) * 3 * CAP, d_prim, d_t, d_u, d_v, (uint8_t *)dout, mode, LH_VARIANT_DEFAULT, NULL, s_tr, true);
        if (rc != 0) return rc;
        char *ho = (char *)a->pipe.h_out[b];
        if (mode == LH_MODE_CLOSEST) {
            if (t) HIPCHK(hipMemcpyAsync(ho, d_t, sizeof(double) * m, hipMemcpyDeviceToHost, s_tr));
            if (u) HIPCHK(hipMemcpyAsync(ho + sizeof(double) * CAP, d_u, sizeof(double) * m, hipMemcpyDeviceToHost, s_tr));
            if (v) HIPCHK(hipMemcpyAsync(ho + 2 * sizeof(double) * CAP, d_v, sizeof(double) * m, hipMemcpyDeviceToHost, s_tr));
            if (prim) HIPCHK(hipMemcpyAsync(ho + 3 * sizeof(double) * CAP, d_prim, sizeof(uint32_t) * m, hipMemcpyDeviceToHost, s_tr));
        } else if (occ) HIPCHK(hipMemcpyAsync(ho, dout, m, hipMemcpyDeviceToHost, s_tr));
        HIPCHK(hipEventRecord(a->pipe.done[b], s_tr));
        if (diag) t_enq += now() - c2;
    }
    const double c3 = diag ? now() : 0;
    for (; unstaged < nchunks; unstaged++) {
        HIPCHK(hipEventSynchronize(a->pipe.done[unstaged % D])); unstage(unstaged); cs.run();
    }
    if (diag) fprintf(stderr, "[lucille_hip] pipelined host batch: %zu chunks of %zu rays, ring of %zu: waits %.2f ms, copies %.2f, enqueues %.2f, tail %.2f\n", nchunks, C, D, t_wait, t_copy, t_enq, now() - c3);
    return 0;
}

extern "C" int lh_accel_intersect_host(lh_accel_t *a, size_t n, const double *org, const double *dir,
                                       uint32_t *prim, double *t, double *u, double *v, uint8_t *occ, int mode)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("intersect: accel not committed");
    if (n == 0) return 0;
    if (!org || !dir) return fail("intersect: NULL ray arrays");
    if (mode != LH_MODE_CLOSEST && mode != LH_MODE_ANY) return fail("intersect: unknown mode %d", mode);
    HIPCHK(hipSetDevice(a->device));
    static const size_t pipe_min = getenv("LH_PIPE_MIN") && atoll(getenv("LH_PIPE_MIN")) > 0 ? (size_t)atoll(getenv("LH_PIPE_MIN")) : LH_PIPE_MIN;
    if (n >= pipe_min && !a->stat_on && !getenv("LH_HOST_SIMPLE"))
        return intersect_host_pipelined(a, n, org, dir, prim, t, u, v, occ, mode);
    /* layout of the staging block: org | dir | t | u | v | prim | occ */
    const size_t b_ray = sizeof(double) * 3 * n, b_d = sizeof(double) * n;
    const size_t total = 2 * b_ray + 3 * b_d + sizeof(uint32_t) * n + n + 64;
    if (lh_ensure_stage(a, total) != 0) return -1;
    char *base = (char *)a->d_stage;
    double *d_org = (double *)base, *d_dir = (double *)(base + b_ray);
    double *d_t = (double *)(base + 2 * b_ray), *d_u = d_t + n, *d_v = d_u + n;
    uint32_t *d_prim = (uint32_t *)(d_v + n);
    uint8_t *d_occ = (uint8_t *)(d_prim + n);
    HIPCHK(hipMemcpyAsync(d_org, org, b_ray, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_dir, dir, b_ray, hipMemcpyHostToDevice, a->stream));
    if (a->stat_on) HIPCHK(hipMemsetAsync(a->d_counters, 0, sizeof(unsigned long long) * LH_CNT_DEV, a->stream));
    int rc = lh_launch(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occ, mode, LH_VARIANT_DEFAULT,
                    a->stat_on ? a->d_counters : NULL, a->stream, true);
    if (rc != 0) return rc;
    if (a->stat_on) {
        /* hits are counted from the device outputs whatever the caller asked to copy back */
        std::vector<uint32_t> hp; std::vector<uint8_t> ho; unsigned long long h[LH_CNT_N] = {0, 0, 0, 0}, nh = 0;
        if (mode == LH_MODE_CLOSEST) {
            hp.resize(n); HIPCHK(hipMemcpyAsync(hp.data(), d_prim, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, a->stream));
        } else {
            ho.resize(n); HIPCHK(hipMemcpyAsync(ho.data(), d_occ, n, hipMemcpyDeviceToHost, a->stream));
        }
        if (a->hs->bvh.ntris) HIPCHK(hipMemcpyAsync(h, a->d_counters, sizeof(h), hipMemcpyDeviceToHost, a->stream));
        HIPCHK(hipStreamSynchronize(a->stream));
        for (size_t i = 0; i < n; i++) nh += (mode == LH_MODE_CLOSEST) ? (hp[i] != LH_MISS_PRIM) : (ho[i] != 0);
        a->stat[0] += h[LH_CNT_NODES]; a->stat[1] += h[LH_CNT_TRIS]; a->stat[2] += h[LH_CNT_EXACT];
        a->stat[3] += n; a->stat[4] += nh;
    }
    if (mode == LH_MODE_CLOSEST) {
        if (prim) HIPCHK(hipMemcpyAsync(prim, d_prim, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, a->stream));
        if (t) HIPCHK(hipMemcpyAsync(t, d_t, b_d, hipMemcpyDeviceToHost, a->stream));
        if (u) HIPCHK(hipMemcpyAsync(u, d_u, b_d, hipMemcpyDeviceToHost, a->stream));
        if (v) HIPCHK(hipMemcpyAsync(v, d_v, b_d, hipMemcpyDeviceToHost, a->stream));
    } else {
        if (occ) HIPCHK(hipMemcpyAsync(occ, d_occ, n, hipMemcpyDeviceToHost, a->stream));
    }
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}

extern "C" int lh_accel_trace_statistics(lh_accel_t *a, int enable)
{
    lh_guard guard(a);
    if (!a) return fail("lh_accel_trace_statistics: NULL accel");
    a->stat_on = enable != 0;
    return 0;
}

extern "C" int lh_accel_statistics(lh_accel_t *a, uint64_t counters[5], int clear)
{
    lh_guard guard(a);
    if (!a) return fail("lh_accel_statistics: NULL accel");
    if (counters) for (int k = 0; k < 5; k++) counters[k] = a->stat[k];
    if (clear) for (int k = 0; k < 5; k++) a->stat[k] = 0;
    return 0;
}

extern "C" int lh_accel_slot_statistics(lh_accel_t *a, uint64_t slots[3], int clear)
{
    lh_guard guard(a);
    if (!a) return fail("lh_accel_slot_statistics: NULL accel");
    if (slots) for (int k = 0; k < 3; k++) slots[k] = a->stat_slots[k];
    if (clear) for (int k = 0; k < 3; k++) a->stat_slots[k] = 0;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* one synchronous ray: accel_intersect_func as lucille calls it              */
/* ------------------------------------------------------------------------ */
/* lucille calls accel->intersect(accel, ray, state, user) for ONE ray from up to 16 render threads at once (raytrace.c:31-69,
 * render.c:1043-1105; 18 call sites in whitted.c, shader.c, ibl.c).  A launch per call is ~100 us of copies, launch and
 * synchronisation for ~1 us of work, and the calls serialise on the accelerator's lock: ~10 000 rays/s whatever the thread
 * count (rounds 1-3).  Here concurrent callers are COALESCED (flat combining): a caller writes its ray into the open batch --
 * a slot of a pinned, device-visible block -- and either finds a batch in flight (it sleeps until its own batch is done) or
 * becomes the leader: it waits a few microseconds for the other render threads (which come back from their previous ray
 * at about the same time), closes the batch, launches it straight out of the pinned block (no copies: the kernel reads the
 * rays and writes the records over the link), and wakes the batch's owners.  While a batch runs the next one fills (two
 * blocks).  Records are the batch path's, bit for bit.  set_param("combine", 0) restores one launch per call. */
#define LH_COMB_CAP 64
struct lh_combiner {
    pthread_mutex_t mu; pthread_cond_t cv;
    unsigned long long open_gen, done_gen;      /* the batch that accepts rays; batches finished (done_gen > g: batch g is done) */
    int leader_active;
    volatile int n_pending;                     /* rays in the open batch */
    int readers_left[2], rc[2], expect;
    char err[2][256];
    void *block; double *org[2], *dir[2], *t[2], *u[2], *v[2]; uint32_t *prim[2];       /* pinned host memory the device reads / writes */
    void *d_block; double *d_org[2], *d_dir[2], *d_t[2], *d_u[2], *d_v[2]; uint32_t *d_prim[2];
    hipStream_t stream;
    unsigned long long batches, rays;
};

static int comb_create(lh_accel_t *a)
{
    lh_combiner *c = (lh_combiner *)calloc(1, sizeof(*c));
    if (!c) return fail("out of memory");
    const size_t per = sizeof(double) * (3 + 3 + 1 + 1 + 1) * LH_COMB_CAP + sizeof(uint32_t) * LH_COMB_CAP;
    if (hipHostMalloc(&c->block, 2 * per, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&c->d_block, c->block, 0) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        const hipError_t e = hipGetLastError();
        if (c->block) (void)hipHostFree(c->block);
        free(c);
        return fail("lh_accel_intersect1: the pinned block of the combiner could not be set up: %s", hipGetErrorString(e));
    }
    for (int b = 0; b < 2; b++) {
        char *h = (char *)c->block + (size_t)b * per, *d = (char *)c->d_block + (size_t)b * per;
        c->org[b] = (double *)h; c->dir[b] = c->org[b] + 3 * LH_COMB_CAP; c->t[b] = c->dir[b] + 3 * LH_COMB_CAP; c->u[b] = c->t[b] + LH_COMB_CAP;
        c->v[b] = c->u[b] + LH_COMB_CAP; c->prim[b] = (uint32_t *)(c->v[b] + LH_COMB_CAP);
        c->d_org[b] = (double *)d; c->d_dir[b] = c->d_org[b] + 3 * LH_COMB_CAP; c->d_t[b] = c->d_dir[b] + 3 * LH_COMB_CAP; c->d_u[b] = c->d_t[b] + LH_COMB_CAP;
        c->d_v[b] = c->d_u[b] + LH_COMB_CAP; c->d_prim[b] = (uint32_t *)(c->d_v[b] + LH_COMB_CAP);
    }
    pthread_mutex_init(&c->mu, NULL); pthread_cond_init(&c->cv, NULL);
    c->expect = 1;
    __atomic_store_n(&a->comb, c, __ATOMIC_RELEASE);          /* published last: a caller that sees the pointer (acquire) sees the mutex and the block pointers */
    return 0;
}

void lh_comb_destroy(lh_accel_t *a)
{
    lh_combiner *c = a->comb;
    if (!c) return;
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->block) (void)hipHostFree(c->block);
    pthread_mutex_destroy(&c->mu); pthread_cond_destroy(&c->cv);
    free(c); a->comb = NULL;
}

static int comb_run(lh_accel_t *a, lh_combiner *c, int b, int n)
{
    lh_guard guard(a);                       /* the accelerator's device state: one launch at a time */
    HIPCHK(hipSetDevice(a->device));
    if (lh_launch(a, (size_t)n, c->d_org[b], c->d_dir[b], c->d_prim[b], c->d_t[b], c->d_u[b], c->d_v[b], NULL, LH_MODE_CLOSEST,
                  LH_VARIANT_DEFAULT, NULL, c->stream, true) != 0) return -1;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int lh_accel_combine_statistics(lh_accel_t *a, uint64_t out[2], int clear)
{
    if (!a || !out) return fail("lh_accel_combine_statistics: NULL argument");
    out[0] = out[1] = 0;
    lh_combiner *c = a->comb;
    if (!c) return 0;
    pthread_mutex_lock(&c->mu);
    out[0] = c->batches; out[1] = c->rays;
    if (clear) c->batches = c->rays = 0;
    pthread_mutex_unlock(&c->mu);
    return 0;
}

extern "C" int lh_accel_intersect1(lh_accel_t *a, const double org[3], const double dir[3],
                                   uint32_t *prim, double *t, double *u, double *v)
{
    uint32_t p = LH_MISS_PRIM; double tt = LH_T_INF, uu = 0.0, vv = 0.0;
    if (!a || !a->committed) return fail("lh_accel_intersect1: accel not committed");
    if (!org || !dir) return fail("lh_accel_intersect1: NULL ray");
    /* One ray, on the calling thread, over the host copy of the trees (lh_hostwalk.c): a device answers a
     * single ray in ~20 us whatever its kernel does, the host walks lucille's example scenes in 0.2-0.3 us -- and sixteen render
     * threads walk side by side instead of queueing for one launch.  Same filter, same fp64 test, same tie and fragile-hit rules:
     * the same record.  For scenes whose trees live on the host (host-built commits; a device-built or received scene has none),
     * while a walk stays cheaper than the device's amortised answer (LH_HOST_WALK_MAX_NS, 8 us: the incoherent rays of a
     * million-triangle soup cost the host 5-10 us of cache misses each); statistics and per-ray diagnostics stay on the device. */
    if (__atomic_load_n(&a->host_walk, __ATOMIC_RELAXED) && !__atomic_load_n(&a->stat_on, __ATOMIC_RELAXED)) {
        const lh_host_scene *hs = a->hs;
        const bool trees_here = !hs->device_built && !hs->received && hs->bvh.q4nodes && hs->bvh.tri32 && hs->bvh.tri64 &&
                                (!hs->have_ref || (!hs->ref_on_device && __atomic_load_n(&hs->ref_state, __ATOMIC_ACQUIRE) == 2 && hs->ref.nodes));
        static const bool cpu_fma = __builtin_cpu_supports("fma");          /* lh_hostwalk.o is compiled with -mfma */
        if (trees_here && cpu_fma && (hs->bvh.ntris == 0 || __atomic_load_n(&a->hw_gpu_left, __ATOMIC_RELAXED) <= 0)) {
            static const double max_ns = getenv("LH_HOST_WALK_MAX_NS") ? atof(getenv("LH_HOST_WALK_MAX_NS")) : 8000.0;
            static thread_local unsigned long long tl_calls = 0;          /* per thread: sixteen render threads must not share a counter's cache line */
            const unsigned long long k = tl_calls++;
            const bool timed = (k & 63ull) == 63ull;         /* one call in 64 is timed (never the first: cold caches, page faults) */
            if (timed) __atomic_fetch_add(&a->hw_calls, 64ull, __ATOMIC_RELAXED);
            const double t0 = timed ? lh_now_s() : 0.0;
            const int hit = lh_host_walk_closest(&hs->bvh, hs->have_ref ? &hs->ref : NULL, org, dir, &p, &tt, &uu, &vv);
            if (timed) {
                const double ns = (lh_now_s() - t0) * 1e9, old = a->hw_ns;
                const double mean = old > 0.0 ? 0.75 * old + 0.25 * ns : ns;
                a->hw_ns = mean;                              /* racy by design: a statistic */
                if (mean > max_ns && k >= 255ull) __atomic_store_n(&a->hw_gpu_left, 65536, __ATOMIC_RELAXED);       /* long walks (four samples or more agree): the device for a while, then another look */
            }
            if (hit != -2) {
                if (prim) *prim = p;
                if (t) *t = tt;
                if (u) *u = uu;
                if (v) *v = vv;
                return hit;
            }
            p = LH_MISS_PRIM; tt = LH_T_INF; uu = vv = 0.0;          /* the host walk ran out of stack rows: this ray through the device */
        }
        if (trees_here) __atomic_fetch_sub(&a->hw_gpu_left, 1, __ATOMIC_RELAXED);
    }
    if (!__atomic_load_n(&a->combine, __ATOMIC_RELAXED) || __atomic_load_n(&a->stat_on, __ATOMIC_RELAXED)) {         /* statistics are per launch: counted launches stay one ray each */
        if (lh_accel_intersect_host(a, 1, org, dir, &p, &tt, &uu, &vv, NULL, LH_MODE_CLOSEST) != 0) return -1;
    } else {
        lh_combiner *c = __atomic_load_n(&a->comb, __ATOMIC_ACQUIRE);
        if (!c) {
            lh_guard guard(a);
            if (!a->comb) { HIPCHK(hipSetDevice(a->device)); if (comb_create(a) != 0) return -1; }
            c = a->comb;
        }
        int rc = 0;
        pthread_mutex_lock(&c->mu);
        while (c->n_pending >= LH_COMB_CAP) pthread_cond_wait(&c->cv, &c->mu);
        const unsigned long long g = c->open_gen;
        const int b = (int)(g & 1ull), slot = c->n_pending;
        for (int k = 0; k < 3; k++) { c->org[b][3 * slot + k] = org[k]; c->dir[b][3 * slot + k] = dir[k]; }
        __atomic_store_n(&c->n_pending, slot + 1, __ATOMIC_RELEASE);
        for (;;) {
            if (__atomic_load_n(&c->done_gen, __ATOMIC_ACQUIRE) > g) break;      /* somebody ran my batch */
            if (!c->leader_active && c->open_gen == g) {
                if (c->readers_left[b] != 0) { pthread_cond_wait(&c->cv, &c->mu); continue; }     /* owners of batch g - 2 still copy their records out of this block */
                c->leader_active = 1;
                /* the other render threads come back from their previous ray about now: wait for as many as the last batches
                 * held, a few microseconds at most */
                static const double gather_us = getenv("LH_COMB_GATHER_US") ? atof(getenv("LH_COMB_GATHER_US")) : 50.0;
                if (gather_us > 0.0 && c->expect > 1 && c->n_pending < c->expect) {
                    pthread_mutex_unlock(&c->mu);
                    const double t0 = lh_now_s();
                    while (__atomic_load_n(&c->n_pending, __ATOMIC_ACQUIRE) < c->expect && lh_now_s() - t0 < gather_us * 1e-6) { }
                    pthread_mutex_lock(&c->mu);
                }
                const int n = c->n_pending;
                c->n_pending = 0; c->open_gen = g + 1;                   /* closed: arrivals fill the other block */
                pthread_cond_broadcast(&c->cv);                           /* callers waiting for room */
                pthread_mutex_unlock(&c->mu);
                const int r = comb_run(a, c, b, n);
                pthread_mutex_lock(&c->mu);
                c->rc[b] = r;
                if (r != 0) { snprintf(c->err[b], sizeof(c->err[b]), "%s", lh_last_error()); }
                c->readers_left[b] = n; c->leader_active = 0;
                __atomic_store_n(&c->done_gen, g + 1, __ATOMIC_RELEASE);
                c->batches++; c->rays += (unsigned long long)n;
                c->expect = n > c->expect ? n : (n + 3 * c->expect) / 4;     /* follows the callers' concurrency, decays slowly */
                if (c->expect < 1) c->expect = 1;
                pthread_cond_broadcast(&c->cv);
                break;
            }
            /* a batch is ~100 us away at most: watch for it without the mutex for a while (a condition variable's wake-up costs
             * tens of microseconds and sixteen owners take the mutex one after the other), then sleep */
            pthread_mutex_unlock(&c->mu);
            {
                static const double spin_us = getenv("LH_COMB_SPIN_US") ? atof(getenv("LH_COMB_SPIN_US")) : 300.0;
                const double t0 = lh_now_s(); int spins = 0;
                if (spin_us > 0.0) while (__atomic_load_n(&c->done_gen, __ATOMIC_ACQUIRE) <= g && !(__atomic_load_n(&c->leader_active, __ATOMIC_RELAXED) == 0 && __atomic_load_n(&c->open_gen, __ATOMIC_RELAXED) == g)) {
                    LH_CPU_RELAX();
                    if ((++spins & 255) == 0 && lh_now_s() - t0 > spin_us * 1e-6) break;
                }
            }
            pthread_mutex_lock(&c->mu);
            if (__atomic_load_n(&c->done_gen, __ATOMIC_ACQUIRE) > g) break;
            if (!c->leader_active && c->open_gen == g) continue;          /* the leader's seat is free: take it */
            pthread_cond_wait(&c->cv, &c->mu);
        }
        rc = c->rc[b];
        if (rc == 0) { p = c->prim[b][slot]; tt = c->t[b][slot]; uu = c->u[b][slot]; vv = c->v[b][slot]; }
        else lh_fail("%s", c->err[b]);
        if (--c->readers_left[b] == 0) pthread_cond_broadcast(&c->cv);
        pthread_mutex_unlock(&c->mu);
        if (rc != 0) return -1;
    }
    if (prim) *prim = p;
    if (t) *t = tt;
    if (u) *u = uu;
    if (v) *v = vv;
    return p != LH_MISS_PRIM;
}


/* ------------------------------------------------------------------------ */
/* per-ray traversal diagnostics (ri_bvh_diag_t, bvh.h:103-110)               */
/* ------------------------------------------------------------------------ */
/* ri_bvh_intersect zeroes and (built with RI_BVH_ENABLE_DIAGNOSTICS) fills a caller-supplied ri_bvh_diag_t through `user`
 * (bvh.c:451-456,1127,1146,827): inner-node visits, leaf visits, leaf tests of THAT ray -- the testbed's heat maps
 * (simplerender.cpp:202-218).  Here: the same three numbers for THIS build's tree, per ray, from the sequential walk
 * (k_trace_small; the numbers of a speculative walk would depend on its wave-mates): diag = n x 4 u32 -- 4-wide node visits,
 * leaf visits, triangle records through the fp32 filter, fp64 tests.  Records are the batch path's.  A diagnostic path: a
 * lane per ray, no regrouping. */
extern "C" int lh_accel_intersect_diag_host(lh_accel_t *a, size_t n, const double *org, const double *dir, uint32_t *prim,
                                            double *t, double *u, double *v, uint32_t *diag)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("intersect_diag: accel not committed");
    if (n == 0) return 0;
    if (!org || !dir || !diag) return fail("intersect_diag: NULL argument");
    if (n > 0x7fffffffull) return fail("intersect_diag: more than 2^31 rays");
    HIPCHK(hipSetDevice(a->device));
    if (a->hs->bvh.ntris == 0) {
        memset(diag, 0, sizeof(uint32_t) * 4 * n);
        for (size_t i = 0; i < n; i++) { if (prim) prim[i] = LH_MISS_PRIM; if (t) t[i] = LH_T_INF; if (u) u[i] = 0.0; if (v) v[i] = 0.0; }
        return 0;
    }
    const size_t b_ray = sizeof(double) * 3 * n, b_d = sizeof(double) * n;
    const size_t total = 2 * b_ray + 3 * b_d + sizeof(uint32_t) * n + sizeof(uint32_t) * 4 * n + 64;
    if (lh_ensure_stage(a, total) != 0) return -1;
    char *base = (char *)a->d_stage;
    double *d_org = (double *)base, *d_dir = (double *)(base + b_ray);
    double *d_t = (double *)(base + 2 * b_ray), *d_u = d_t + n, *d_v = d_u + n;
    uint32_t *d_prim = (uint32_t *)(d_v + n), *d_diag = d_prim + n;
    HIPCHK(hipMemcpyAsync(d_org, org, b_ray, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_dir, dir, b_ray, hipMemcpyHostToDevice, a->stream));
    if (a->stat_on) HIPCHK(hipMemsetAsync(a->d_counters, 0, sizeof(unsigned long long) * LH_CNT_DEV, a->stream));
    a->dev.diag_out = d_diag;
    const int rc = lh_launch(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, NULL, LH_MODE_CLOSEST, LH_VARIANT_SPEC, a->stat_on ? a->d_counters : NULL, a->stream, true);
    a->dev.diag_out = NULL;
    if (rc != 0) return rc;
    HIPCHK(hipMemcpyAsync(diag, d_diag, sizeof(uint32_t) * 4 * n, hipMemcpyDeviceToHost, a->stream));
    std::vector<uint32_t> hp;
    if (a->stat_on || prim) { hp.resize(n); HIPCHK(hipMemcpyAsync(hp.data(), d_prim, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, a->stream)); }
    if (t) HIPCHK(hipMemcpyAsync(t, d_t, b_d, hipMemcpyDeviceToHost, a->stream));
    if (u) HIPCHK(hipMemcpyAsync(u, d_u, b_d, hipMemcpyDeviceToHost, a->stream));
    if (v) HIPCHK(hipMemcpyAsync(v, d_v, b_d, hipMemcpyDeviceToHost, a->stream));
    unsigned long long h[LH_CNT_N] = {0, 0, 0, 0};
    if (a->stat_on) HIPCHK(hipMemcpyAsync(h, a->d_counters, sizeof(h), hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    if (prim) memcpy(prim, hp.data(), sizeof(uint32_t) * n);
    if (a->stat_on) {
        unsigned long long nh = 0;
        for (size_t i = 0; i < n; i++) nh += hp[i] != LH_MISS_PRIM;
        a->stat[0] += h[LH_CNT_NODES]; a->stat[1] += h[LH_CNT_TRIS]; a->stat[2] += h[LH_CNT_EXACT]; a->stat[3] += n; a->stat[4] += nh;
    }
    return 0;
}

/* the same for rays resident on the device, closest- or any-hit: d_diag n x 4 u32 (see lh_accel_intersect_diag_host); the hit
 * records go to the accelerator's staging block and are dropped.  What a cost map of a frame is made of (tools/ao_cost_map.py). */
extern "C" int lh_accel_intersect_diag_device(lh_accel_t *a, size_t n, const void *d_org, const void *d_dir, int mode, void *d_diag, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("intersect_diag_device: accel not committed");
    if (n == 0) return 0;
    if (!d_org || !d_dir || !d_diag) return fail("intersect_diag_device: NULL argument");
    if (mode != LH_MODE_CLOSEST && mode != LH_MODE_ANY) return fail("intersect_diag_device: unknown mode %d", mode);
    if (n > 0x7fffffffull) return fail("intersect_diag_device: more than 2^31 rays");
    HIPCHK(hipSetDevice(a->device));
    hipStream_t s = (hipStream_t)stream;
    if (a->hs->bvh.ntris == 0) { HIPCHK(hipMemsetAsync(d_diag, 0, sizeof(uint32_t) * 4 * n, s)); return 0; }
    const size_t b_d = sizeof(double) * n;
    if (lh_ensure_stage(a, 3 * b_d + sizeof(uint32_t) * n + n + 64) != 0) return -1;
    double *d_t = (double *)a->d_stage, *d_u = d_t + n, *d_v = d_u + n;
    uint32_t *d_prim = (uint32_t *)(d_v + n); uint8_t *d_occ = (uint8_t *)(d_prim + n);
    a->dev.diag_out = (uint32_t *)d_diag;
    const int rc = lh_launch(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occ, mode, LH_VARIANT_SPEC, NULL, s, true);
    a->dev.diag_out = NULL;
    return rc;
}

/* ------------------------------------------------------------------------ */
/* beam visibility                                                          */
/* ------------------------------------------------------------------------ */
extern "C" int lh_launch_beam_visibility(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dirs,
                                         int32_t *d_result, void *stream, const lh_beam_set_t *d_preset);

__global__ void k_fill_i32(size_t n, int32_t *p, int32_t v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

static int beam_visibility_launch(lh_accel_t *a, size_t n, const void *d_org, const void *d_dirs, void *d_result, void *stream, const lh_beam_set_t *d_preset);

extern "C" int lh_accel_beam_visibility_device(lh_accel_t *a, size_t n, const void *d_org, const void *d_dirs,
                                               void *d_result, void *stream)
{
    return beam_visibility_launch(a, n, d_org, d_dirs, d_result, stream, NULL);
}

static int beam_visibility_launch(lh_accel_t *a, size_t n, const void *d_org, const void *d_dirs, void *d_result, void *stream, const lh_beam_set_t *d_preset)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("beam_visibility: accel not committed");
    if (n == 0) return 0;
    if ((!d_preset && (!d_org || !d_dirs)) || !d_result) return fail("beam_visibility: NULL argument");
    if (!a->hs->have_ref) return fail("beam_visibility: the reference-order tree was disabled (LH_REFTREE=0)");
    if (lh_sync_ref(a, true) != 0) return -1;
    HIPCHK(hipSetDevice(a->device));
    lh_dev_scene_t sc = a->dev;
    if (a->hs->bvh.ntris == 0) { sc.ref_empty = 1; }
    if (lh_launch_beam_visibility(&sc, n, (const double *)d_org, (const double *)d_dirs, (int32_t *)d_result, stream, d_preset) != 0)
        return fail("beam kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int lh_accel_beam_visibility_host(lh_accel_t *a, size_t n, const double *org, const double *dirs, int32_t *result)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("beam_visibility: accel not committed");
    if (n == 0) return 0;
    if (!org || !dirs || !result) return fail("beam_visibility: NULL argument");
    HIPCHK(hipSetDevice(a->device));
    const size_t bo = sizeof(double) * 3 * n, bd = sizeof(double) * 12 * n, br = sizeof(int32_t) * n;
    if (lh_ensure_stage(a, bo + bd + br + 64) != 0) return -1;
    char *base = (char *)a->d_stage;
    HIPCHK(hipMemcpyAsync(base, org, bo, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(base + bo, dirs, bd, hipMemcpyHostToDevice, a->stream));
    if (lh_accel_beam_visibility_device(a, n, base, base + bo, base + bo + bd, a->stream) != 0) return -1;
    HIPCHK(hipMemcpyAsync(result, base + bo + bd, br, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}

/* beams the caller's own ri_beam_set has set up (lh_beam_set_t: ri_bvh_intersect_beam_visibility's argument, bvh.h:208-221) */
extern "C" int lh_accel_beam_visibility_set_host(lh_accel_t *a, size_t n, const lh_beam_set_t *beams, int32_t *result)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("beam_visibility: accel not committed");
    if (n == 0) return 0;
    if (!beams || !result) return fail("beam_visibility: NULL argument");
    HIPCHK(hipSetDevice(a->device));
    const size_t bb = sizeof(lh_beam_set_t) * n, br = sizeof(int32_t) * n;
    if (lh_ensure_stage(a, bb + br + 64) != 0) return -1;
    char *base = (char *)a->d_stage;
    HIPCHK(hipMemcpyAsync(base, beams, bb, hipMemcpyHostToDevice, a->stream));
    if (beam_visibility_launch(a, n, NULL, NULL, base + bb, a->stream, (const lh_beam_set_t *)base) != 0) return -1;
    HIPCHK(hipMemcpyAsync(result, base + bb, br, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}

/* ------------------------------------------------------------------------ */
/* the beam-raster path (ri_bvh_intersect_beam, bvh.c:544-609)               */
/* ------------------------------------------------------------------------ */
extern "C" int lh_launch_beam_raster(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dirs, const double *d_corner,
                                     const lh_raster_plane_t *plane, double ktan, double *d_t, int32_t *d_status, unsigned long long *d_flags,
                                     void *stream, const lh_beam_set_t *d_preset);

static int beam_raster_launch(lh_accel_t *a, size_t n, const void *d_org, const void *d_dirs, const void *d_corner, const lh_raster_plane_t *plane,
                              void *d_t, void *d_status, void *d_flags, void *stream, const lh_beam_set_t *d_preset);

extern "C" int lh_accel_beam_raster_device(lh_accel_t *a, size_t n, const void *d_org, const void *d_dirs, const void *d_corner,
                                           const lh_raster_plane_t *plane, void *d_t, void *d_status, void *d_flags, void *stream)
{
    return beam_raster_launch(a, n, d_org, d_dirs, d_corner, plane, d_t, d_status, d_flags, stream, NULL);
}

static int beam_raster_launch(lh_accel_t *a, size_t n, const void *d_org, const void *d_dirs, const void *d_corner, const lh_raster_plane_t *plane,
                              void *d_t, void *d_status, void *d_flags, void *stream, const lh_beam_set_t *d_preset)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("beam_raster: accel not committed");
    if (n == 0) return 0;
    if ((!d_preset && (!d_org || !d_dirs)) || !d_corner || !plane || !d_t || !d_status) return fail("beam_raster: NULL argument");
    if (plane->width <= 0 || plane->height <= 0) return fail("beam_raster: the raster window is %d x %d", plane->width, plane->height);
    if (!a->hs->have_ref) return fail("beam_raster: the reference-order tree was disabled (LH_REFTREE=0)");
    if (lh_sync_ref(a, true) != 0) return -1;
    HIPCHK(hipSetDevice(a->device));
    lh_dev_scene_t sc = a->dev;
    if (a->hs->bvh.ntris == 0) { sc.ref_empty = 1; }
    /* (1.0 / tan(0.5 * fov_rad)) exactly as raster.c:120,135-136 evaluates it, on the host */
    const double fov_rad = plane->fov * M_PI / 180.0;
    const double ktan = 1.0 / tan(0.5 * fov_rad);
    if (lh_launch_beam_raster(&sc, n, (const double *)d_org, (const double *)d_dirs, (const double *)d_corner, plane, ktan, (double *)d_t,
                              (int32_t *)d_status, (unsigned long long *)d_flags, stream, d_preset) != 0)
        return fail("beam raster kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int lh_accel_beam_raster_host(lh_accel_t *a, size_t n, const double *org, const double *dirs, const double *corner,
                                         const lh_raster_plane_t *plane, double *t_out, int32_t *status, uint64_t *flags)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("beam_raster: accel not committed");
    if (n == 0) return 0;
    if (!org || !dirs || !corner || !plane || !t_out || !status) return fail("beam_raster: NULL argument");
    if (plane->width <= 0 || plane->height <= 0) return fail("beam_raster: the raster window is %d x %d", plane->width, plane->height);
    HIPCHK(hipSetDevice(a->device));
    const size_t px = (size_t)plane->width * (size_t)plane->height;
    const size_t bo = sizeof(double) * 3 * n, bd = sizeof(double) * 12 * n, bt = sizeof(double) * px * n, bs = (sizeof(int32_t) * n + 7) & ~(size_t)7,
                 bf = sizeof(uint64_t) * 4 * n;
    if (lh_ensure_stage(a, 2 * bo + bd + bt + bs + bf + 64) != 0) return -1;
    char *base = (char *)a->d_stage;
    char *d_org = base, *d_dirs = base + bo, *d_corner = d_dirs + bd, *d_t = d_corner + bo, *d_st = d_t + bt, *d_fl = d_st + bs;
    HIPCHK(hipMemcpyAsync(d_org, org, bo, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_dirs, dirs, bd, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_corner, corner, bo, hipMemcpyHostToDevice, a->stream));
    /* planes that are not traced keep the caller's contents: the staging copy starts as the caller's */
    HIPCHK(hipMemcpyAsync(d_t, t_out, bt, hipMemcpyHostToDevice, a->stream));
    if (lh_accel_beam_raster_device(a, n, d_org, d_dirs, d_corner, plane, d_t, d_st, d_fl, a->stream) != 0) return -1;
    HIPCHK(hipMemcpyAsync(t_out, d_t, bt, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipMemcpyAsync(status, d_st, sizeof(int32_t) * n, hipMemcpyDeviceToHost, a->stream));
    if (flags) HIPCHK(hipMemcpyAsync(flags, d_fl, bf, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}

extern "C" int lh_accel_beam_raster_set_host(lh_accel_t *a, size_t n, const lh_beam_set_t *beams, const double *corner,
                                             const lh_raster_plane_t *plane, double *t_out, int32_t *status, uint64_t *flags)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("beam_raster: accel not committed");
    if (n == 0) return 0;
    if (!beams || !corner || !plane || !t_out || !status) return fail("beam_raster: NULL argument");
    if (plane->width <= 0 || plane->height <= 0) return fail("beam_raster: the raster window is %d x %d", plane->width, plane->height);
    HIPCHK(hipSetDevice(a->device));
    const size_t px = (size_t)plane->width * (size_t)plane->height;
    const size_t bb = (sizeof(lh_beam_set_t) * n + 15) & ~(size_t)15, bo = sizeof(double) * 3 * n, bt = sizeof(double) * px * n,
                 bs = (sizeof(int32_t) * n + 7) & ~(size_t)7, bf = sizeof(uint64_t) * 4 * n;
    if (lh_ensure_stage(a, bb + bo + bt + bs + bf + 64) != 0) return -1;
    char *base = (char *)a->d_stage;
    char *d_beams = base, *d_corner = base + bb, *d_t = d_corner + bo, *d_st = d_t + bt, *d_fl = d_st + bs;
    HIPCHK(hipMemcpyAsync(d_beams, beams, sizeof(lh_beam_set_t) * n, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_corner, corner, bo, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_t, t_out, bt, hipMemcpyHostToDevice, a->stream));         /* planes that are not traced keep the caller's contents */
    if (beam_raster_launch(a, n, NULL, NULL, d_corner, plane, d_t, d_st, d_fl, a->stream, (const lh_beam_set_t *)d_beams) != 0) return -1;
    HIPCHK(hipMemcpyAsync(t_out, d_t, bt, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipMemcpyAsync(status, d_st, sizeof(int32_t) * n, hipMemcpyDeviceToHost, a->stream));
    if (flags) HIPCHK(hipMemcpyAsync(flags, d_fl, bf, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}
