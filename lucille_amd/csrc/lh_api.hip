/*
 * lh_api.hip -- implementation of the C ABI in include/lucille_hip.h over the
 * HIP runtime.  Owns the host BVH (lh_bvh.c), its device copies, and a small
 * amount of per-accel device scratch (work cursor, counters, staging for the
 * host-batch entry point).
 */
#include <hip/hip_runtime.h>
#include <thread>
#include <vector>

#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../../include/lucille_hip.h"
#include "lh_bvh.h"
#include "lh_refbvh.h"
#include "lh_device.h"

static thread_local char g_err[512] = "";

extern "C" void lh_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

static int fail(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return -1;
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail("%s failed: %s", #x, hipGetErrorString(e_)); } while (0)

/* attribute kinds of lh_accel_set_attribute: per-vertex xyz (colour, tangent, binormal), per-vertex st,
 * per-index st (texcoords_unshared) -- the optional members of ri_geom_t that ri_intersection_state_build reads */
struct lh_mesh_copy { uint32_t npos, nidx; double *pos; uint32_t *idx; double *nrm; int two_side;
                      double *attr[5]; };

/* launchers in lh_render.hip */
extern "C" int lh_render_launch_primary(const lh_camera_t *cam, int x0, int y0, int w, int h, int xs, int ys,
                                        double *d_org, double *d_dir, void *stream);
extern "C" int lh_render_launch_compact(const lh_dev_scene_t *sc, const double *d_nrm9, size_t n, const double *d_org,
                                        const double *d_dir, const uint32_t *d_prim, const double *d_t,
                                        const double *d_u, const double *d_v, uint32_t *d_block_counts,
                                        uint32_t *d_slot_of_sample, double *d_hitrec,
                                        unsigned long long *d_slot_key, int x0, int w, int nbands, int band_rows,
                                        const int *d_band_y0, int y0, int spp, int full_width,
                                        unsigned long long *d_total, void *stream);
extern "C" int lh_render_launch_primary_region(const lh_camera_t *cam, int x0, int w, int nbands, int band_rows, const int *d_band_y0,
                                               int y0, int height_limit, int xs, int ys, double *d_org, double *d_dir, void *stream);
extern "C" int lh_render_launch_ao_rays(size_t nslots, int ntheta, int nphi, unsigned long long seed,
                                        const double *d_hitrec, const double *d_rnd,
                                        const unsigned long long *d_slot_key, double *d_org, double *d_dir, void *stream);
extern "C" int lh_render_launch_resolve(int w, int h, int band_rows, int xs, int ys, int N, const uint32_t *d_slot_of_sample,
                                        const uint8_t *d_occ, const unsigned int *d_occ_count, float *d_rgb,
                                        unsigned long long *d_occ_total, void *stream);

struct lh_buf { void *p; size_t cap; };
#define LH_T2_SLOTS 4

/* the host side of a committed scene: ONE build, any number of device replicas (lh_multi.hip
 * uploads it to every GPU of the node; SURVEY.md 8e "replicated BVH") */
struct lh_host_scene {
    int refs;                 /* guarded by g_scene_mu */
    lh_bvh_t bvh;
    lh_refbvh_t ref;          /* reference-order tree (ties, beams, the reference walk) */
    int have_ref;
    double ref_build_seconds;
    double *nrm9;             /* per-primitive vertex normals (9 doubles, NaN = none) or NULL */
    double *attr9[3];         /* colour / tangent / binormal per primitive (9 doubles, NaN = none) or NULL */
    double *st6;              /* texture coordinates per primitive (6 doubles, NaN = none) or NULL */
    uint8_t *inside;          /* per primitive: the back half of a two-sided mesh (intersection_state.c:233-241) or NULL */
    uint32_t nmeshes;         /* meshes the scene was committed with */
    /* device build (lh_build.hip): the host holds the flattened primitives only; the reference-order tree is built by a
     * background thread and attached to the replicas when it is ready (ref_state: 0 none, 1 building, 2 ready, -1 failed) */
    int device_built;
    int ref_state; pthread_t ref_thread; int ref_thread_live; int ref_threads;
};
static pthread_mutex_t g_scene_mu = PTHREAD_MUTEX_INITIALIZER;

struct lh_accel {
    int device;
    int committed;
    int commit_failed;        /* a commit that failed half-way: device memory is released by destroy, a retry is refused */
    /* staged meshes (host copies, packed xyz) */
    lh_mesh_copy *meshes; uint32_t nmeshes;
    lh_host_scene *hs;        /* never NULL after create */
    void *d_ref_lca, *d_prim_leafpos, *d_ref_nodes, *d_ref_leaf_prims;
    /* device */
    lh_dev_scene_t dev;
    void *d_nodes, *d_tri32, *d_tri64, *d_qnodes, *d_q4nodes, *d_q4tnodes, *d_q8nodes, *d_c8nodes, *d_tri32_c8;
    int ncus, grid_forced_pt;          /* compute units of the device; "pt_grid": workgroups of the fused path-tracing kernel */
    int pt_fused;                      /* path-tracing passes inside the walk (ray source 2 of the trace kernel); 0: wavefront passes */
    int wide8;                         /* ray dumps walk the 8-wide nodes: -1 when the hot set exceeds the Infinity Cache (default), 0 never, 1 always */
    int quad_grid;                     /* workgroups of the quad-per-ray walk (variant 7) */
    unsigned long long *d_cursor, *d_counters;   /* d_cursor: LH_NCURSOR slots, one per launch in flight */
    unsigned cursor_next;
    pthread_mutex_t mu;                /* serialises the entry points of ONE accelerator (recursive) */
    int stat_on;                       /* lh_accel_trace_statistics */
    unsigned long long stat[5];        /* nodes, filter tests, fp64 tests, rays, hits */
    hipStream_t stream;
    uint64_t device_bytes;
    double upload_seconds;
    int grid_blocks;
    int min_active;
    uint32_t ray_chunk;                /* rays reserved per cursor atomic (LH_RAY_CHUNK) */
    int tri_batch;
    int default_variant;
    /* lean walk (lh_trace2.hip): persistent grid + per-stream scratch (spill strips, pending queue) */
    int t2_grid;
    struct { hipStream_t stream; int used; int *spill; uint32_t *queue; uint32_t *qcount; } t2[LH_T2_SLOTS];
    /* staging for host batches */
    void *d_stage; size_t stage_bytes;
    /* pipelined host batches: two pinned in/out staging pairs, two device pairs, two streams */
    struct { void *h_in[2], *h_out[2], *d_in[2], *d_out[2]; hipStream_t s[2]; hipEvent_t done[2]; size_t cap; int ready; } pipe;
    void *d_nrm9;                      /* hs->nrm9 on the device */
    void *d_attr9[3], *d_st6, *d_inside;            /* colour / tangent / binormal, st, inside flags (uploaded at commit if present) */
    void *d_prim_mesh;                 /* mesh ordinal per primitive (materials; uploaded on first use) */
    lh_material_t *materials; uint32_t nmaterials; void *d_materials; int materials_dirty;
    lh_environment_t env; void *d_env_map;
    lh_buf r_state;                    /* lh_accel_state_build_host staging */
    lh_buf r_uni;                      /* lh_render_ao_tile_host: caller uniforms on the device */
    lh_buf r_bands;                    /* lh_render_ao_bands: first line of every band */
    /* tile-render scratch (lh_render_ao_tile) */
    lh_buf r_org, r_dir, r_prim, r_t, r_u, r_v, r_slot, r_hitrec, r_aorg, r_adir, r_occ, r_blocks, r_key, r_frame, r_occcount;
    uint64_t last_retraced;            /* rays the last counted launch finished outside the main kernel */
    int ao_fused;                      /* AO rays generated inside the any-hit kernel (default); 0: materialised in HBM */
    lh_buf p_org2, p_dir2, p_path, p_path2, p_thr, p_thr2, p_rad, p_alive;   /* path tracer */
    unsigned long long *d_total;
    size_t r_nsamples, r_nslots, r_nao;
};

#define LH_NCURSOR 64
#define LH_T2_QCAP  (1u << 20)       /* pending any-hit rays per launch (24 B each); beyond: the reference walk */

/* lucille calls accel->intersect from up to 16 render threads at once (render.c:1043-1105): every
 * entry point that touches the accelerator's buffers holds its lock */
struct lh_guard {
    pthread_mutex_t *m;
    explicit lh_guard(const lh_accel_t *a) : m(a ? (pthread_mutex_t *)&a->mu : NULL) { if (m) pthread_mutex_lock(m); }
    ~lh_guard() { if (m) pthread_mutex_unlock(m); }
};

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

extern "C" const char *lh_last_error(void) { return g_err; }

extern "C" int lh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int lh_accel_create(lh_accel_t **out, int device)
{
    if (!out) return fail("lh_accel_create: out is NULL");
    int n = lh_device_count();
    if (n <= 0) return fail("lh_accel_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail("lh_accel_create: device %d out of range [0,%d)", device, n);
    lh_accel_t *a = (lh_accel_t *)calloc(1, sizeof(*a));
    if (!a) return fail("out of memory");
    a->hs = (lh_host_scene *)calloc(1, sizeof(lh_host_scene));
    if (!a->hs) { free(a); return fail("out of memory"); }
    a->hs->refs = 1;
    a->device = device;
    {
        pthread_mutexattr_t at; pthread_mutexattr_init(&at); pthread_mutexattr_settype(&at, PTHREAD_MUTEX_RECURSIVE);
        pthread_mutex_init(&a->mu, &at); pthread_mutexattr_destroy(&at);
    }
    a->default_variant = LH_VARIANT_SPEC;
    a->ao_fused = 1;
    a->wide8 = -1;
    { const char *e = getenv("LH_WIDE8"); if (e) a->wide8 = atoi(e); }
    a->pt_fused = 0;
    { const char *e = getenv("LH_PT_FUSED"); if (e) a->pt_fused = atoi(e) != 0; }
    { const char *e = getenv("LH_AO_FUSED"); if (e) a->ao_fused = atoi(e) != 0; }
    const char *env = getenv("LH_VARIANT");
    if (env) a->default_variant = atoi(env);
    a->min_active = 32;
    a->tri_batch = 8;
    env = getenv("LH_TRI_BATCH");
    if (env && atoi(env) > 0 && atoi(env) <= 64) a->tri_batch = atoi(env);
    env = getenv("LH_MIN_ACTIVE");
    if (env && atoi(env) > 0 && atoi(env) <= 64) a->min_active = atoi(env);
    a->ray_chunk = 256;
    env = getenv("LH_RAY_CHUNK");
    if (env && atoi(env) > 0 && atoi(env) <= (1 << 20)) a->ray_chunk = (uint32_t)atoi(env);
    a->dev.ray_chunk = a->ray_chunk;
    *out = a;
    return 0;
}

extern "C" int lh_accel_add_mesh(lh_accel_t *a, uint32_t npos, const double *pos, size_t stride,
                                 uint32_t nidx, const uint32_t *idx)
{
    lh_guard guard(a);
    if (!a) return fail("lh_accel_add_mesh: accel is NULL");
    if (a->committed) return fail("lh_accel_add_mesh: accel already committed");
    if ((npos && !pos) || (nidx && !idx)) return fail("lh_accel_add_mesh: NULL array");
    if (stride < 3 * sizeof(double) || (stride % sizeof(double)) != 0) return fail("lh_accel_add_mesh: bad stride %zu", stride);
    for (uint32_t i = 0; i < nidx - (nidx % 3); i++)
        if (idx[i] >= npos) return fail("lh_accel_add_mesh: index %u out of range (npositions %u)", idx[i], npos);
    lh_mesh_copy *nm = (lh_mesh_copy *)realloc(a->meshes, sizeof(lh_mesh_copy) * (a->nmeshes + 1));
    if (!nm) return fail("out of memory");
    a->meshes = nm;
    lh_mesh_copy *m = &a->meshes[a->nmeshes];
    m->npos = npos; m->nidx = nidx; m->nrm = NULL; m->two_side = 0;
    for (int k = 0; k < 5; k++) m->attr[k] = NULL;
    m->pos = (double *)malloc(sizeof(double) * 3 * (size_t)(npos ? npos : 1));
    m->idx = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(nidx ? nidx : 1));
    if (!m->pos || !m->idx) return fail("out of memory");
    for (uint32_t i = 0; i < npos; i++) {
        const double *p = (const double *)((const char *)pos + (size_t)i * stride);
        m->pos[3 * (size_t)i] = p[0]; m->pos[3 * (size_t)i + 1] = p[1]; m->pos[3 * (size_t)i + 2] = p[2];
    }
    memcpy(m->idx, idx, sizeof(uint32_t) * nidx);
    a->nmeshes++;
    return 0;
}

extern "C" int lh_accel_set_normals(lh_accel_t *a, uint32_t mesh, const double *nrm, size_t stride, int two_side)
{
    lh_guard guard(a);
    if (!a) return fail("lh_accel_set_normals: accel is NULL");
    if (a->committed) return fail("lh_accel_set_normals: accel already committed");
    if (mesh >= a->nmeshes) return fail("lh_accel_set_normals: mesh %u out of range", mesh);
    if (nrm && (stride < 3 * sizeof(double) || (stride % sizeof(double)) != 0)) return fail("lh_accel_set_normals: bad stride");
    lh_mesh_copy *m = &a->meshes[mesh];
    free(m->nrm); m->nrm = NULL; m->two_side = two_side;
    if (nrm) {
        m->nrm = (double *)malloc(sizeof(double) * 3 * (size_t)(m->npos ? m->npos : 1));
        if (!m->nrm) return fail("out of memory");
        for (uint32_t i = 0; i < m->npos; i++) {
            const double *p = (const double *)((const char *)nrm + (size_t)i * stride);
            m->nrm[3 * (size_t)i] = p[0]; m->nrm[3 * (size_t)i + 1] = p[1]; m->nrm[3 * (size_t)i + 2] = p[2];
        }
    }
    return 0;
}

extern "C" int lh_accel_set_attribute(lh_accel_t *a, uint32_t mesh, int kind, const double *data, size_t stride, uint32_t count)
{
    lh_guard guard(a);
    if (!a) return fail("lh_accel_set_attribute: accel is NULL");
    if (a->committed) return fail("lh_accel_set_attribute: accel already committed");
    if (mesh >= a->nmeshes) return fail("lh_accel_set_attribute: mesh %u out of range", mesh);
    if (kind < LH_ATTR_COLOR || kind > LH_ATTR_TEXCOORD_UNSHARED) return fail("lh_accel_set_attribute: unknown attribute kind %d", kind);
    lh_mesh_copy *m = &a->meshes[mesh];
    const int ncomp = kind <= LH_ATTR_BINORMAL ? 3 : 2;
    const uint32_t need = kind == LH_ATTR_TEXCOORD_UNSHARED ? m->nidx : m->npos;
    free(m->attr[kind]); m->attr[kind] = NULL;
    if (!data) return 0;
    if (count != need) return fail("lh_accel_set_attribute: %u values given, the mesh needs %u (one per %s)", count, need,
                                   kind == LH_ATTR_TEXCOORD_UNSHARED ? "index" : "vertex");
    if (stride < (size_t)ncomp * sizeof(double) || (stride % sizeof(double)) != 0) return fail("lh_accel_set_attribute: bad stride %zu", stride);
    m->attr[kind] = (double *)malloc(sizeof(double) * ncomp * (size_t)(need ? need : 1));
    if (!m->attr[kind]) return fail("out of memory");
    for (uint32_t i = 0; i < need; i++) {
        const double *q = (const double *)((const char *)data + (size_t)i * stride);
        for (int k = 0; k < ncomp; k++) m->attr[kind][(size_t)ncomp * i + k] = q[k];
    }
    return 0;
}

static void free_buf(lh_buf *b) { if (b->p) (void)hipFree(b->p); b->p = NULL; b->cap = 0; }

static void release_device(lh_accel_t *a)
{
    lh_buf *bufs[] = {&a->r_org, &a->r_dir, &a->r_prim, &a->r_t, &a->r_u, &a->r_v, &a->r_slot, &a->r_hitrec,
                      &a->r_aorg, &a->r_adir, &a->r_occ, &a->r_blocks, &a->r_key, &a->r_frame, &a->r_occcount,
                      &a->p_org2, &a->p_dir2, &a->p_path, &a->p_path2, &a->p_thr, &a->p_thr2, &a->p_rad, &a->p_alive};
    for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); i++) free_buf(bufs[i]);
    if (a->d_total) (void)hipFree(a->d_total);
    if (a->d_nrm9) (void)hipFree(a->d_nrm9);
    for (int k = 0; k < 3; k++) { if (a->d_attr9[k]) (void)hipFree(a->d_attr9[k]); a->d_attr9[k] = NULL; }
    if (a->d_st6) (void)hipFree(a->d_st6);
    if (a->d_inside) (void)hipFree(a->d_inside);
    if (a->d_prim_mesh) (void)hipFree(a->d_prim_mesh);
    if (a->d_materials) (void)hipFree(a->d_materials);
    if (a->d_env_map) (void)hipFree(a->d_env_map);
    a->d_st6 = a->d_inside = a->d_prim_mesh = a->d_materials = a->d_env_map = NULL;
    free_buf(&a->r_state); free_buf(&a->r_uni); free_buf(&a->r_bands);
    a->d_total = NULL; a->d_nrm9 = NULL;
    if (a->d_nodes) (void)hipFree(a->d_nodes);
    if (a->d_tri32) (void)hipFree(a->d_tri32);
    if (a->d_tri64) (void)hipFree(a->d_tri64);
    if (a->d_qnodes) (void)hipFree(a->d_qnodes);
    a->d_qnodes = NULL;
    if (a->d_q4nodes) (void)hipFree(a->d_q4nodes);
    if (a->d_q4tnodes) (void)hipFree(a->d_q4tnodes);
    if (a->d_q8nodes) (void)hipFree(a->d_q8nodes);
    a->d_q4nodes = a->d_q4tnodes = a->d_q8nodes = NULL; a->dev.q4tnodes = NULL; a->dev.q8nodes = NULL;
    if (a->d_c8nodes) (void)hipFree(a->d_c8nodes);
    if (a->d_tri32_c8) (void)hipFree(a->d_tri32_c8);
    a->d_c8nodes = a->d_tri32_c8 = NULL;
    if (a->d_ref_lca) (void)hipFree(a->d_ref_lca);
    if (a->d_prim_leafpos) (void)hipFree(a->d_prim_leafpos);
    if (a->d_ref_nodes) (void)hipFree(a->d_ref_nodes);
    if (a->d_ref_leaf_prims) (void)hipFree(a->d_ref_leaf_prims);
    a->d_ref_lca = a->d_prim_leafpos = a->d_ref_nodes = a->d_ref_leaf_prims = NULL;
    if (a->d_cursor) (void)hipFree(a->d_cursor);
    if (a->d_counters) (void)hipFree(a->d_counters);
    for (int k = 0; k < LH_T2_SLOTS; k++) {
        if (a->t2[k].spill) (void)hipFree(a->t2[k].spill);
        if (a->t2[k].queue) (void)hipFree(a->t2[k].queue);
        a->t2[k].spill = NULL; a->t2[k].queue = NULL; a->t2[k].qcount = NULL; a->t2[k].used = 0;
    }
    if (a->pipe.ready) {
        for (int b = 0; b < 2; b++) {
            (void)hipHostFree(a->pipe.h_in[b]); (void)hipHostFree(a->pipe.h_out[b]);
            (void)hipFree(a->pipe.d_in[b]); (void)hipFree(a->pipe.d_out[b]);
            (void)hipStreamDestroy(a->pipe.s[b]); (void)hipEventDestroy(a->pipe.done[b]);
        }
        a->pipe.ready = 0;
    }
    if (a->d_stage) (void)hipFree(a->d_stage);
    if (a->stream) (void)hipStreamDestroy(a->stream);
    a->d_nodes = a->d_tri32 = a->d_tri64 = NULL; a->d_cursor = a->d_counters = NULL;
    a->d_stage = NULL; a->stage_bytes = 0; a->stream = NULL;
}

static void *ref_thread_main(void *arg)
{
    lh_host_scene *hs = (lh_host_scene *)arg;
    const double t0 = now_s();
    const int rc = lh_refbvh_build(&hs->ref, hs->bvh.tri64, hs->bvh.ntris, hs->ref_threads);
    hs->ref_build_seconds = now_s() - t0;
    __atomic_store_n(&hs->ref_state, rc == 0 ? 2 : -1, __ATOMIC_RELEASE);
    return NULL;
}

/* lh_bvh_build_hook's callback on the host path: the triangles are flattened, start lucille's own tree */
static void start_ref_thread(void *arg)
{
    lh_host_scene *hs = (lh_host_scene *)arg;
    if (hs->bvh.ntris == 0) return;
    hs->ref_state = 1;
    if (pthread_create(&hs->ref_thread, NULL, ref_thread_main, hs) != 0) { hs->ref_state = 0; return; }
    hs->ref_thread_live = 1;
}

/* ---- host build (once per scene) ------------------------------------------------------------ */
static int host_build(lh_accel_t *a, int build_threads, bool on_device, bool keep_meshes)
{
    lh_host_scene *hs = a->hs;
    if (build_threads <= 0) {
        /* LH_BUILD_THREADS: how a multi-process launcher (one rank per GPU) keeps N ranks from
         * oversubscribing the host N times over (lucille_amd/shard.py sets cores / world) */
        const char *e = getenv("LH_BUILD_THREADS");
        long nc = (e && atoi(e) > 0) ? atoi(e) : sysconf(_SC_NPROCESSORS_ONLN);
        build_threads = nc > 0 ? (int)nc : 1;
    }
    lh_mesh_view_t *views = (lh_mesh_view_t *)calloc(a->nmeshes ? a->nmeshes : 1, sizeof(*views));
    if (!views) return fail("out of memory");
    for (uint32_t g = 0; g < a->nmeshes; g++) {
        views[g].npositions = a->meshes[g].npos; views[g].positions = a->meshes[g].pos;
        views[g].stride_bytes = 3 * sizeof(double);
        views[g].nindices = a->meshes[g].nidx; views[g].indices = a->meshes[g].idx;
    }
    /* host build: lucille's own tree (needs only the flattened triangles) is built next to the traversal tree */
    {
        const char *e = getenv("LH_REFTREE");
        hs->have_ref = !(e && atoi(e) == 0);
        hs->ref_threads = build_threads;
    }
    int rc = on_device ? lh_bvh_flatten(&hs->bvh, views, a->nmeshes)
                       : lh_bvh_build_hook(&hs->bvh, views, a->nmeshes, build_threads, hs->have_ref ? start_ref_thread : NULL, hs);
    free(views);
    if (!on_device && hs->ref_thread_live) {           /* the host path returns with both trees */
        pthread_join(hs->ref_thread, NULL); hs->ref_thread_live = 0;
        if (rc == 0 && hs->ref_state != 2) return fail("lh_accel_commit: reference-order tree build failed (out of memory)");
    }
    hs->device_built = on_device ? 1 : 0;
    if (rc == -2) return fail("lh_accel_commit: a vertex coordinate is NaN, infinite or beyond 1e30");
    if (rc != 0) return fail("lh_accel_commit: BVH build failed (bad input or out of memory)");
    /* the reference-order tree: exact-t tie winners, the reference walk for fragile hits, beam
     * visibility (LH_REFTREE=0 skips it: ties then fall back to "larger primitive id wins",
     * fragile hits are not re-traced and beam queries are refused) */
    {
        const double t0 = now_s();
        if (hs->have_ref && on_device && hs->bvh.ntris) {
            /* not in front of the first frame: a background thread builds it, launch() attaches it when it is ready */
            hs->ref_threads = build_threads; hs->ref_state = 1;
            if (pthread_create(&hs->ref_thread, NULL, ref_thread_main, hs) != 0) { hs->ref_state = 0; return fail("lh_accel_commit: cannot start the reference-tree thread"); }
            hs->ref_thread_live = 1;
        } else if (hs->ref_state != 2) {
            /* not started next to the tree build (an empty scene, or the thread could not be created): now */
            if (hs->have_ref && lh_refbvh_build(&hs->ref, hs->bvh.tri64, hs->bvh.ntris, build_threads) != 0)
                return fail("lh_accel_commit: reference-order tree build failed (out of memory)");
            hs->ref_build_seconds = now_s() - t0;
            hs->ref_state = hs->have_ref ? 2 : 0;
        }
    }
    /* per-primitive normals in primitive-id order, if any mesh carries normals */
    {
        bool any = false;
        for (uint32_t g = 0; g < a->nmeshes; g++) any = any || a->meshes[g].nrm != NULL;
        if (any && hs->bvh.ntris) {
            hs->nrm9 = (double *)malloc(sizeof(double) * 9 * (size_t)hs->bvh.ntris);
            if (!hs->nrm9) return fail("out of memory");
            for (uint32_t p = 0; p < hs->bvh.ntris; p++) {
                const lh_mesh_copy *m = &a->meshes[hs->bvh.prim_geom[p]];
                double *o = hs->nrm9 + 9 * (size_t)p;
                if (!m->nrm) { for (int k = 0; k < 9; k++) o[k] = NAN; continue; }
                for (int c = 0; c < 3; c++) {
                    uint32_t vi = m->idx[hs->bvh.prim_index[p] + c];
                    for (int k = 0; k < 3; k++) o[3 * c + k] = m->nrm[3 * (size_t)vi + k];
                }
            }
        }
    }
    /* the packed mesh copies are no longer needed: the BVH holds tri64 */
    /* the other per-vertex attributes ri_intersection_state_build reads, flattened the same way */
    {
        const uint32_t n = hs->bvh.ntris;
        hs->nmeshes = a->nmeshes;
        for (int kind = 0; kind < 3 && n; kind++) {
            bool anyk = false;
            for (uint32_t g = 0; g < a->nmeshes; g++) anyk = anyk || a->meshes[g].attr[kind] != NULL;
            if (!anyk) continue;
            hs->attr9[kind] = (double *)malloc(sizeof(double) * 9 * (size_t)n);
            if (!hs->attr9[kind]) return fail("out of memory");
            for (uint32_t p = 0; p < n; p++) {
                const lh_mesh_copy *m = &a->meshes[hs->bvh.prim_geom[p]];
                double *o = hs->attr9[kind] + 9 * (size_t)p;
                if (!m->attr[kind]) { for (int k = 0; k < 9; k++) o[k] = NAN; continue; }
                for (int c = 0; c < 3; c++) {
                    const uint32_t vi = m->idx[hs->bvh.prim_index[p] + c];
                    for (int k = 0; k < 3; k++) o[3 * c + k] = m->attr[kind][3 * (size_t)vi + k];
                }
            }
        }
        bool any_st = false, any_two = false;
        for (uint32_t g = 0; g < a->nmeshes; g++) { any_st = any_st || a->meshes[g].attr[3] || a->meshes[g].attr[4]; any_two = any_two || a->meshes[g].two_side; }
        if (any_st && n) {
            hs->st6 = (double *)malloc(sizeof(double) * 6 * (size_t)n);
            if (!hs->st6) return fail("out of memory");
            for (uint32_t p = 0; p < n; p++) {
                const lh_mesh_copy *m = &a->meshes[hs->bvh.prim_geom[p]];
                double *o = hs->st6 + 6 * (size_t)p;
                const uint32_t first = hs->bvh.prim_index[p];
                if (m->attr[3]) {                 /* shared: geom->texcoords[2 * i_c] (intersection_state.c:210-216) */
                    for (int c = 0; c < 3; c++) { const uint32_t vi = m->idx[first + c]; o[2 * c] = m->attr[3][2 * (size_t)vi]; o[2 * c + 1] = m->attr[3][2 * (size_t)vi + 1]; }
                } else if (m->attr[4]) {          /* unshared: geom->texcoords_unshared[2 * (index + c)] (:218-224) */
                    for (int c = 0; c < 3; c++) { o[2 * c] = m->attr[4][2 * (size_t)(first + c)]; o[2 * c + 1] = m->attr[4][2 * (size_t)(first + c) + 1]; }
                } else for (int k = 0; k < 6; k++) o[k] = NAN;
            }
        }
        if (any_two && n) {
            hs->inside = (uint8_t *)calloc(n, 1);
            if (!hs->inside) return fail("out of memory");
            for (uint32_t p = 0; p < n; p++) {
                const lh_mesh_copy *m = &a->meshes[hs->bvh.prim_geom[p]];
                hs->inside[p] = (m->two_side && hs->bvh.prim_index[p] >= m->nidx / 2) ? 1 : 0;
            }
        }
    }
    if (!keep_meshes) {
        for (uint32_t g = 0; g < a->nmeshes; g++) {
            free(a->meshes[g].pos); free(a->meshes[g].idx); free(a->meshes[g].nrm);
            for (int k = 0; k < 5; k++) free(a->meshes[g].attr[k]);
        }
        free(a->meshes); a->meshes = NULL; a->nmeshes = 0;
    }
    if (!on_device && hs->bvh.ntris && hs->bvh.max_depth + 1 > 64) return fail("lh_accel_commit: tree depth %u exceeds the kernel stack", hs->bvh.max_depth);
    return 0;
}

/* node formats a walk can read; only the one the default kernel uses is uploaded at commit, the
 * others (A/B variants, the deep-tree fallback) on first use */
enum { LH_FMT_F32 = 1, LH_FMT_Q16 = 2, LH_FMT_Q16X4 = 4, LH_FMT_C8 = 8, LH_FMT_Q4T = 16, LH_FMT_Q8 = 32 };
extern "C" int lh_trace_formats_needed(const lh_dev_scene_t *sc, int variant);      /* lh_kernels.hip */
extern "C" int lh_device_build(uint32_t ntris, const double *d_tri64, void **d_q4nodes, uint32_t *nq4, uint32_t *q4_depth,
                               void **d_tri32, float bmin[3], float bmax[3], float grid_lo[3], float grid_step[3],
                               void *stream, char *err, size_t errlen);                /* lh_build.hip */



static int ensure_formats(lh_accel_t *a, int mask)
{
    const lh_bvh_t *b = &a->hs->bvh;
    if ((mask & LH_FMT_Q4T) && !a->d_q4tnodes && (a->d_q4nodes || !a->hs->device_built)) {
        /* the quad-per-ray walk's child-major copy of the 4-wide nodes, made on the device */
        if (!a->d_q4nodes && ensure_formats(a, LH_FMT_Q16X4) != 0) return -1;
        const size_t q4b = sizeof(lh_q4node_t) * (size_t)b->nq4nodes;
        HIPCHK(hipMalloc(&a->d_q4tnodes, q4b));
        if (lh_quad_make_nodes(b->nq4nodes, a->d_q4nodes, a->d_q4tnodes, (void *)a->stream) != 0) return fail("lh_quad_make_nodes failed");
        HIPCHK(hipStreamSynchronize(a->stream));
        a->dev.q4tnodes = a->d_q4tnodes; a->device_bytes += q4b;
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, a->device));
        uint32_t need = 3 * b->q4_depth + 5;
        if (need > 64) need = 64;
        if (need < 16) need = 16;
        a->quad_grid = prop.multiProcessorCount * lh_quad_blocks_per_cu(need);
        const char *e = getenv("LH_QUAD_GRID");
        if (e && atoi(e) > 0) a->quad_grid = atoi(e);
    }
    mask &= ~LH_FMT_Q4T;
    if ((mask & LH_FMT_Q8) && !a->d_q8nodes && !a->hs->device_built) {
        /* the 8-wide 16-bit-grid nodes for ray dumps over scenes larger than the Infinity Cache: built on first use */
        pthread_mutex_lock(&g_scene_mu);
        const int rc8 = lh_bvh_ensure_q8(&a->hs->bvh);
        pthread_mutex_unlock(&g_scene_mu);
        if (rc8 != 0) return fail("building the 8-wide tree failed (out of memory)");
        const size_t q8b = sizeof(lh_q8node_t) * (size_t)b->nq8nodes;
        HIPCHK(hipMalloc(&a->d_q8nodes, q8b));
        HIPCHK(hipMemcpy(a->d_q8nodes, b->q8nodes, q8b, hipMemcpyHostToDevice));
        a->dev.q8nodes = a->d_q8nodes; a->dev.nq8nodes = b->nq8nodes; a->dev.q8_depth = b->q8_depth; a->device_bytes += q8b;
    }
    mask &= ~LH_FMT_Q8;
    if (a->hs->device_built) {
        if (mask & ~LH_FMT_Q16X4) return fail("this scene's tree was built on the device: only the 4-wide walks are available (variants 4, 6 and 7)");
        return 0;
    }
    if ((mask & LH_FMT_F32) && !a->d_nodes) {
        const size_t nb = sizeof(lh_node_t) * (size_t)b->nnodes;
        HIPCHK(hipMalloc(&a->d_nodes, nb));
        HIPCHK(hipMemcpy(a->d_nodes, b->nodes, nb, hipMemcpyHostToDevice));
        a->dev.nodes = a->d_nodes; a->device_bytes += nb;
    }
    if ((mask & LH_FMT_Q16) && !a->d_qnodes) {
        pthread_mutex_lock(&g_scene_mu);
        const int rcq = lh_bvh_ensure_qnodes(&a->hs->bvh);
        pthread_mutex_unlock(&g_scene_mu);
        if (rcq != 0) return fail("building the 2-wide 16-bit grid nodes failed");
        const size_t qb = sizeof(lh_qnode_t) * (size_t)b->nnodes;
        HIPCHK(hipMalloc(&a->d_qnodes, qb));
        HIPCHK(hipMemcpy(a->d_qnodes, b->qnodes, qb, hipMemcpyHostToDevice));
        a->dev.qnodes = a->d_qnodes; a->device_bytes += qb;
    }
    if ((mask & LH_FMT_Q16X4) && !a->d_q4nodes) {
        const size_t q4b = sizeof(lh_q4node_t) * (size_t)b->nq4nodes;
        HIPCHK(hipMalloc(&a->d_q4nodes, q4b));
        HIPCHK(hipMemcpy(a->d_q4nodes, b->q4nodes, q4b, hipMemcpyHostToDevice));
        a->dev.q4nodes = a->d_q4nodes; a->device_bytes += q4b;
    }
    if ((mask & LH_FMT_C8) && !a->d_c8nodes) {
        pthread_mutex_lock(&g_scene_mu);
        const int rc8 = lh_bvh_ensure_c8(&a->hs->bvh);
        pthread_mutex_unlock(&g_scene_mu);
        if (rc8 != 0) return fail("building the 8-wide compressed tree failed (out of memory)");
        a->dev.nc8nodes = b->nc8nodes; a->dev.c8_depth = b->c8_depth;
        const char *es = getenv("LH_C8_STRIDE");
        const size_t stride = (es && atoi(es) == 128) ? 128 : sizeof(lh_c8node_t);
        const size_t c8b = stride * (size_t)b->nc8nodes, t32 = sizeof(lh_tri32_t) * (size_t)b->ntris;
        a->dev.c8_stride = (uint32_t)stride;
        HIPCHK(hipMalloc(&a->d_c8nodes, c8b + 128));
        HIPCHK(hipMalloc(&a->d_tri32_c8, t32 + 64));
        if (stride == sizeof(lh_c8node_t)) { HIPCHK(hipMemcpy(a->d_c8nodes, b->c8nodes, c8b, hipMemcpyHostToDevice)); }
        else HIPCHK(hipMemcpy2D(a->d_c8nodes, stride, b->c8nodes, sizeof(lh_c8node_t), sizeof(lh_c8node_t), b->nc8nodes, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(a->d_tri32_c8, b->tri32_c8, t32, hipMemcpyHostToDevice));
        a->dev.c8nodes = a->d_c8nodes; a->dev.tri32_c8 = a->d_tri32_c8; a->device_bytes += c8b + t32;
    }
    return 0;
}

/* the reference-order tree of the host scene onto this replica's device (exact-t tie winners, the reference walk for
 * fragile hits, beams).  With a device-built traversal tree this happens when the background build has finished. */
static int attach_ref(lh_accel_t *a)
{
    lh_host_scene *hs = a->hs;
    if (a->d_ref_nodes || !hs->have_ref || hs->bvh.ntris == 0) return 0;
    {
        const uint32_t rn = hs->ref.nnodes;
        int *lca = (int *)malloc(sizeof(int) * 4 * (size_t)rn);
        uint32_t *lp = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (size_t)hs->bvh.ntris);
        if (!lca || !lp) { free(lca); free(lp); return fail("out of memory"); }
        for (uint32_t i = 0; i < rn; i++) {
            lca[4 * i] = hs->ref.nodes[i].parent; lca[4 * i + 1] = hs->ref.nodes[i].depth;
            lca[4 * i + 2] = hs->ref.nodes[i].axis; lca[4 * i + 3] = hs->ref.nodes[i].child[0];
        }
        for (uint32_t p = 0; p < hs->bvh.ntris; p++) { lp[2 * p] = hs->ref.prim_leaf[p]; lp[2 * p + 1] = hs->ref.prim_pos[p]; }
        hipError_t e1 = hipMalloc(&a->d_ref_lca, sizeof(int) * 4 * (size_t)rn);
        hipError_t e2 = hipMalloc(&a->d_prim_leafpos, sizeof(uint32_t) * 2 * (size_t)hs->bvh.ntris);
        hipError_t e3 = hipMalloc(&a->d_ref_nodes, sizeof(lh_refnode_t) * (size_t)rn);
        hipError_t e4 = hipMalloc(&a->d_ref_leaf_prims, sizeof(uint32_t) * (size_t)hs->bvh.ntris);
        if (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && e4 == hipSuccess) {
            e1 = hipMemcpy(a->d_ref_lca, lca, sizeof(int) * 4 * (size_t)rn, hipMemcpyHostToDevice);
            e2 = hipMemcpy(a->d_prim_leafpos, lp, sizeof(uint32_t) * 2 * (size_t)hs->bvh.ntris, hipMemcpyHostToDevice);
            e3 = hipMemcpy(a->d_ref_nodes, hs->ref.nodes, sizeof(lh_refnode_t) * (size_t)rn, hipMemcpyHostToDevice);
            e4 = hipMemcpy(a->d_ref_leaf_prims, hs->ref.leaf_prims, sizeof(uint32_t) * (size_t)hs->bvh.ntris, hipMemcpyHostToDevice);
        }
        free(lca); free(lp);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) return fail("reference-order tree upload failed");
        a->dev.ref_lca = a->d_ref_lca; a->dev.prim_leafpos = a->d_prim_leafpos;
        a->dev.ref_nodes = a->d_ref_nodes; a->dev.ref_leaf_prims = a->d_ref_leaf_prims;
        a->dev.ref_nnodes = rn; a->dev.ref_empty = hs->ref.empty;
        for (int k = 0; k < 3; k++) { a->dev.ref_bmin[k] = hs->ref.bmin[k]; a->dev.ref_bmax[k] = hs->ref.bmax[k]; }
        a->device_bytes += sizeof(int) * 4 * (size_t)rn + sizeof(uint32_t) * 3 * (size_t)hs->bvh.ntris + sizeof(lh_refnode_t) * (size_t)rn;
    }
    return 0;
}

/* the background build of the reference-order tree: attach it if it has finished (wait: block until it has) */
static int sync_ref(lh_accel_t *a, bool wait)
{
    lh_host_scene *hs = a->hs;
    if (!hs->have_ref || a->d_ref_nodes || hs->bvh.ntris == 0) return 0;
    if (wait) {
        pthread_mutex_lock(&g_scene_mu);
        if (hs->ref_thread_live) { pthread_join(hs->ref_thread, NULL); hs->ref_thread_live = 0; }
        pthread_mutex_unlock(&g_scene_mu);
    }
    const int st = __atomic_load_n(&hs->ref_state, __ATOMIC_ACQUIRE);
    if (st == -1) return fail("the reference-order tree build failed (out of memory)");
    if (st != 2) return 0;
    HIPCHK(hipSetDevice(a->device));
    return attach_ref(a);
}

/* ---- device replica of the host scene (once per GPU) ---------------------------------------- */
static int device_upload(lh_accel_t *a)
{
    lh_host_scene *hs = a->hs;
    HIPCHK(hipSetDevice(a->device));
    double t0 = now_s();
    HIPCHK(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc((void **)&a->d_cursor, sizeof(unsigned long long) * LH_NCURSOR));
    HIPCHK(hipMalloc((void **)&a->d_counters, sizeof(unsigned long long) * LH_CNT_DEV));
    HIPCHK(hipMalloc((void **)&a->d_total, sizeof(unsigned long long) * 4));
    a->device_bytes = 0;
    if (hs->nrm9) {
        HIPCHK(hipMalloc(&a->d_nrm9, sizeof(double) * 9 * (size_t)hs->bvh.ntris));
        HIPCHK(hipMemcpy(a->d_nrm9, hs->nrm9, sizeof(double) * 9 * (size_t)hs->bvh.ntris, hipMemcpyHostToDevice));
        a->device_bytes += sizeof(double) * 9 * (size_t)hs->bvh.ntris;
    }
    for (int kind = 0; kind < 3; kind++) if (hs->attr9[kind]) {
        const size_t b = sizeof(double) * 9 * (size_t)hs->bvh.ntris;
        HIPCHK(hipMalloc(&a->d_attr9[kind], b));
        HIPCHK(hipMemcpy(a->d_attr9[kind], hs->attr9[kind], b, hipMemcpyHostToDevice));
        a->device_bytes += b;
    }
    if (hs->st6) {
        const size_t b = sizeof(double) * 6 * (size_t)hs->bvh.ntris;
        HIPCHK(hipMalloc(&a->d_st6, b)); HIPCHK(hipMemcpy(a->d_st6, hs->st6, b, hipMemcpyHostToDevice));
        a->device_bytes += b;
    }
    if (hs->inside) {
        HIPCHK(hipMalloc(&a->d_inside, hs->bvh.ntris)); HIPCHK(hipMemcpy(a->d_inside, hs->inside, hs->bvh.ntris, hipMemcpyHostToDevice));
        a->device_bytes += hs->bvh.ntris;
    }
    if (hs->bvh.ntris) {
        size_t t32 = sizeof(lh_tri32_t) * (size_t)hs->bvh.ntris;
        size_t t64 = sizeof(lh_tri64_t) * (size_t)hs->bvh.ntris;
        HIPCHK(hipMalloc(&a->d_tri64, t64));
        HIPCHK(hipMemcpy(a->d_tri64, hs->bvh.tri64, t64, hipMemcpyHostToDevice));
        if (hs->device_built) {
            /* the traversal tree is built here, on this device (lh_build.hip): LBVH -> the same 4-wide nodes */
            char berr[256] = "";
            uint32_t nq4 = 0, d4 = 0; float bmin[3], bmax[3], glo[3], gst[3];
            const double tb = now_s();
            const int rcb = lh_device_build(hs->bvh.ntris, (const double *)a->d_tri64, &a->d_q4nodes, &nq4, &d4, &a->d_tri32, bmin, bmax, glo, gst,
                                            (void *)a->stream, berr, sizeof(berr));
            if (rcb == -2) return fail("lh_accel_commit: a vertex coordinate is NaN, infinite or beyond 1e30");
            if (rcb != 0) return fail("device BVH build failed: %s", berr);
            pthread_mutex_lock(&g_scene_mu);
            hs->bvh.nq4nodes = nq4; hs->bvh.q4_depth = d4; hs->bvh.nnodes = nq4; hs->bvh.max_depth = d4; hs->bvh.build_seconds = now_s() - tb;
            for (int k = 0; k < 3; k++) { hs->bvh.bmin[k] = bmin[k]; hs->bvh.bmax[k] = bmax[k]; hs->bvh.grid_lo[k] = glo[k]; hs->bvh.grid_step[k] = gst[k]; }
            pthread_mutex_unlock(&g_scene_mu);
            a->dev.q4nodes = a->d_q4nodes;
            a->device_bytes += sizeof(lh_q4node_t) * (size_t)nq4;
            if (3 * d4 + 5 > 264) return -3;          /* deeper than k_overflow_fix's private stack: the caller falls back to the host builder */
        } else {
            HIPCHK(hipMalloc(&a->d_tri32, t32 + 64));   /* the unified walk reads 16 B past a record */
            HIPCHK(hipMemcpy(a->d_tri32, hs->bvh.tri32, t32, hipMemcpyHostToDevice));
        }
        a->device_bytes += t32 + t64;
        float r = 0.0f;
        for (int k = 0; k < 3; k++) { r = fmaxf(r, fabsf(hs->bvh.bmin[k])); r = fmaxf(r, fabsf(hs->bvh.bmax[k])); }
        a->dev.tri32 = a->d_tri32; a->dev.tri64 = a->d_tri64;
        a->dev.ntris = hs->bvh.ntris; a->dev.nnodes = hs->bvh.nnodes;
        a->dev.max_depth = hs->bvh.max_depth; a->dev.scene_r = r; a->dev.ray_chunk = a->ray_chunk;
        for (int k = 0; k < 3; k++) { a->dev.grid_lo[k] = hs->bvh.grid_lo[k]; a->dev.grid_step[k] = hs->bvh.grid_step[k]; }
        if (hs->have_ref && __atomic_load_n(&hs->ref_state, __ATOMIC_ACQUIRE) == 2 && attach_ref(a) != 0) return -1;
        a->dev.nq4nodes = hs->bvh.nq4nodes; a->dev.q4_depth = hs->bvh.q4_depth;
        a->dev.nc8nodes = hs->bvh.nc8nodes; a->dev.c8_depth = hs->bvh.c8_depth;
        a->dev.use_qnodes = 2;
        {
            const char *fmt = getenv("LH_NODE_FORMAT");
            if (fmt && strcmp(fmt, "f32") == 0) a->dev.use_qnodes = 0;
            if (fmt && strcmp(fmt, "q16") == 0) a->dev.use_qnodes = 1;
            if (3 * hs->bvh.q4_depth + 5 > 64 && !hs->device_built) a->dev.use_qnodes = 1;     /* pathological depth: 2-wide walk */
            a->dev.nodes_2wide_available = !hs->device_built;
            /* 8-wide compressed nodes: stack overflow is handed to the reference walk, so that tree is needed */
            if (fmt && strcmp(fmt, "c8") == 0 && hs->have_ref) a->dev.use_qnodes = 3;
        }
        /* resident from the start: what the default kernel reads (everything else on first use) */
        if (ensure_formats(a, lh_trace_formats_needed(&a->dev, a->default_variant)) != 0) return -1;
    }
    a->upload_seconds = now_s() - t0;
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, a->device));
        uint32_t need = hs->bvh.max_depth + 1;
        if (a->dev.use_qnodes == 2) need = 3 * hs->bvh.q4_depth + 5;
        if (need > 64) need = 64;
        uint32_t stack = (need + 1u) & ~1u;
        if (stack < 16) stack = 16;
        int per_cu = (int)(160u / stack);                     /* LDS: stack KiB per 256-thread workgroup */
        if (per_cu > 5) per_cu = 5;
        if (per_cu < 1) per_cu = 1;
        a->grid_blocks = prop.multiProcessorCount * per_cu; a->ncus = prop.multiProcessorCount;
        const char *env = getenv("LH_GRID_BLOCKS");
        if (env && atoi(env) > 0) a->grid_blocks = atoi(env);
        a->t2_grid = prop.multiProcessorCount * lh_trace2_blocks_per_cu();
        env = getenv("LH_T2_GRID_BLOCKS");
        if (env && atoi(env) > 0) a->t2_grid = atoi(env);
    }
    a->committed = 1;
    return 0;
}

extern "C" int lh_accel_commit(lh_accel_t *a, int build_threads)
{
    lh_guard guard(a);
    if (!a) return fail("lh_accel_commit: accel is NULL");
    if (a->committed) return fail("lh_accel_commit: already committed");
    if (a->commit_failed) return fail("lh_accel_commit: an earlier commit of this accelerator failed; create a new one");
    a->commit_failed = 1;                       /* cleared on success */
    bool on_device = build_threads == LH_BUILD_ON_DEVICE;
    { const char *e = getenv("LH_BUILD"); if (e && strcmp(e, "device") == 0) on_device = true; if (e && strcmp(e, "host") == 0) on_device = false; }
    { const char *f = getenv("LH_NODE_FORMAT"); if (f && strcmp(f, "q16x4") != 0) on_device = false; }   /* the A/B formats come from the host builder */
    if (build_threads < 0) build_threads = 0;
    if (on_device) {
        if (host_build(a, build_threads, true, true) != 0) return -1;
        const int rc = device_upload(a);
        if (rc == -3) {
            /* an LBVH deeper than the kernel's stack bound (degenerate distributions): build on the host after all */
            (void)hipSetDevice(a->device); release_device(a);
            lh_host_scene *hs = a->hs;
            pthread_mutex_lock(&g_scene_mu);
            if (hs->ref_thread_live) { pthread_join(hs->ref_thread, NULL); hs->ref_thread_live = 0; }
            pthread_mutex_unlock(&g_scene_mu);
            free(hs->nrm9); free(hs->attr9[0]); free(hs->attr9[1]); free(hs->attr9[2]); free(hs->st6); free(hs->inside);
            hs->nrm9 = NULL; hs->attr9[0] = hs->attr9[1] = hs->attr9[2] = NULL; hs->st6 = NULL; hs->inside = NULL;
            lh_bvh_release(&hs->bvh); lh_refbvh_release(&hs->ref); hs->ref_state = 0;
            on_device = false;
        } else {
            for (uint32_t g = 0; g < a->nmeshes; g++) {
                free(a->meshes[g].pos); free(a->meshes[g].idx); free(a->meshes[g].nrm);
                for (int k = 0; k < 5; k++) free(a->meshes[g].attr[k]);
            }
            free(a->meshes); a->meshes = NULL; a->nmeshes = 0;
            if (rc != 0) return -1;
            a->commit_failed = 0;
            return 0;
        }
    }
    if (host_build(a, build_threads, false, false) != 0) return -1;
    if (device_upload(a) != 0) return -1;
    a->commit_failed = 0;
    return 0;
}

/* blocks until the reference-order tree of a device-built scene is attached: from then on exact-t ties and fragile
 * hits follow the reference's tree (before: ties fall back to "larger primitive id wins", as with LH_REFTREE=0) */
extern "C" int lh_accel_wait_exact(lh_accel_t *a)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_accel_wait_exact: accel not committed");
    return sync_ref(a, true);
}

/* lh_multi.hip: `dst` (created, nothing added) becomes a replica of `src`'s committed scene on its own device */
extern "C" int lh_accel_commit_replica(lh_accel_t *dst, lh_accel_t *src)
{
    if (!dst || !src || !src->committed) return fail("lh_accel_commit_replica: source not committed");
    lh_guard guard(dst);
    if (dst->committed || dst->commit_failed || dst->nmeshes) return fail("lh_accel_commit_replica: destination is not a fresh accelerator");
    pthread_mutex_lock(&g_scene_mu);
    lh_host_scene *old = dst->hs;
    dst->hs = src->hs; dst->hs->refs++;
    pthread_mutex_unlock(&g_scene_mu);
    free(old);                                  /* a fresh accelerator's scene holds nothing */
    dst->commit_failed = 1;
    if (device_upload(dst) != 0) return -1;         /* a device-built scene is built again on this replica's device */
    dst->commit_failed = 0;
    return 0;
}

extern "C" void lh_accel_destroy(lh_accel_t *a)
{
    if (!a) return;
    if (a->committed || a->commit_failed) { (void)hipSetDevice(a->device); release_device(a); }
    for (uint32_t g = 0; g < a->nmeshes; g++) {
        free(a->meshes[g].pos); free(a->meshes[g].idx); free(a->meshes[g].nrm);
        for (int k = 0; k < 5; k++) free(a->meshes[g].attr[k]);
    }
    free(a->meshes);
    pthread_mutex_lock(&g_scene_mu);
    const int last = (--a->hs->refs == 0);
    pthread_mutex_unlock(&g_scene_mu);
    if (last) {
        if (a->hs->ref_thread_live) { pthread_join(a->hs->ref_thread, NULL); a->hs->ref_thread_live = 0; }
        free(a->hs->nrm9); free(a->hs->attr9[0]); free(a->hs->attr9[1]); free(a->hs->attr9[2]); free(a->hs->st6); free(a->hs->inside);
        lh_bvh_release(&a->hs->bvh);
        lh_refbvh_release(&a->hs->ref);
        free(a->hs);
    }
    pthread_mutex_destroy(&a->mu);
    free(a);
}

extern "C" int lh_accel_info(const lh_accel_t *a, lh_accel_info_t *o)
{
    if (!a || !o) return fail("lh_accel_info: NULL argument");
    if (!a->committed) return fail("lh_accel_info: accel not committed");
    o->ntriangles = a->hs->bvh.ntris; o->nnodes = a->hs->bvh.nnodes; o->nleaves = a->hs->bvh.nleaves;
    o->max_depth = a->hs->bvh.max_depth; o->device_bytes = a->device_bytes;
    o->build_seconds = a->hs->bvh.build_seconds; o->upload_seconds = a->upload_seconds;
    o->device = a->device;
    o->ref_build_seconds = a->hs->ref_build_seconds;
    o->nnodes_traversal = a->hs->bvh.nq4nodes;
    return 0;
}

extern "C" int lh_accel_prim_lookup(const lh_accel_t *a, uint32_t prim, uint32_t *mesh, uint32_t *index)
{
    if (!a || !a->committed) return fail("lh_accel_prim_lookup: accel not committed");
    if (prim >= a->hs->bvh.ntris) return fail("lh_accel_prim_lookup: prim %u out of range", prim);
    if (mesh) *mesh = a->hs->bvh.prim_geom[prim];
    if (index) *index = a->hs->bvh.prim_index[prim];
    return 0;
}

extern "C" int lh_accel_set_grid(lh_accel_t *a, int blocks)
{
    lh_guard guard(a);
    if (!a || blocks <= 0) return fail("lh_accel_set_grid: bad argument");
    a->grid_blocks = blocks;
    return 0;
}

/* tuning knobs of the traversal kernels (A/B sweeps, tools/): "grid" / "t2_grid" persistent workgroups of the
 * r01 / lean walk, "min_active" regroup threshold, "tri_batch" parked leaves per triangle pass, "ray_chunk"
 * rays per cursor atomic, "variant" default kernel variant */
extern "C" int lh_accel_set_param(lh_accel_t *a, const char *name, int value)
{
    lh_guard guard(a);
    if (!a || !name) return fail("lh_accel_set_param: NULL argument");
    if (!strcmp(name, "grid") && value > 0) a->grid_blocks = value;
    else if (!strcmp(name, "t2_grid") && value > 0) {
        if (value > a->t2_grid) {            /* spill strips are sized by the grid: drop them, they are re-made */
            HIPCHK(hipSetDevice(a->device)); HIPCHK(hipDeviceSynchronize());
            for (int k = 0; k < LH_T2_SLOTS; k++) {
                if (a->t2[k].spill) (void)hipFree(a->t2[k].spill);
                if (a->t2[k].queue) (void)hipFree(a->t2[k].queue);
                a->t2[k].spill = NULL; a->t2[k].queue = NULL; a->t2[k].qcount = NULL; a->t2[k].used = 0;
            }
        }
        a->t2_grid = value;
    }
    else if (!strcmp(name, "min_active") && value > 0 && value <= 64) a->min_active = value;
    else if (!strcmp(name, "tri_batch") && value > 0 && value <= 64) a->tri_batch = value;
    else if (!strcmp(name, "ray_chunk") && value > 0 && value <= (1 << 20)) { a->ray_chunk = (uint32_t)value; a->dev.ray_chunk = (uint32_t)value; }
    else if (!strcmp(name, "variant") && value >= 0 && value <= LH_VARIANT_QUAD) a->default_variant = value;
    else if (!strcmp(name, "ao_fused")) a->ao_fused = value != 0;
    else if (!strcmp(name, "wide8") && value >= -1 && value <= 1) a->wide8 = value;
    else if (!strcmp(name, "pt_fused")) a->pt_fused = value != 0;
    else if (!strcmp(name, "pt_grid") && value >= 0) a->grid_forced_pt = value;
    else if (!strcmp(name, "quad_grid") && value > 0) a->quad_grid = value;
    else if (!strcmp(name, "stack_cap") && (value == 0 || (value >= 8 && value <= 64 && value % 2 == 0))) a->dev.stack_cap = (uint32_t)value;
    else return fail("lh_accel_set_param: unknown parameter or bad value: %s = %d", name, value);
    return 0;
}

extern "C" int lh_accel_export(const lh_accel_t *a, void *nodes, void *tri32)
{
    if (!a || !a->committed) return fail("lh_accel_export: accel not committed");
    if (a->hs->device_built) return fail("lh_accel_export: the tree was built on the device; there is no host copy");
    if (nodes && a->hs->bvh.nnodes) memcpy(nodes, a->hs->bvh.nodes, sizeof(lh_node_t) * (size_t)a->hs->bvh.nnodes);
    if (tri32 && a->hs->bvh.ntris) memcpy(tri32, a->hs->bvh.tri32, sizeof(lh_tri32_t) * (size_t)a->hs->bvh.ntris);
    return 0;
}

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

/* scratch of the lean walk for launches on `s`: launches on one stream are ordered, so they share a slot;
 * different streams (the two pipeline streams of large host batches) get their own */
static int t2_slot(lh_accel_t *a, hipStream_t s, bool need_spill)
{
    int k, free_k = -1;
    for (k = 0; k < LH_T2_SLOTS; k++) {
        if (a->t2[k].used && a->t2[k].stream == s) return k;
        if (!a->t2[k].used && free_k < 0) free_k = k;
    }
    if (free_k < 0) {
        /* more concurrent streams than slots: wait for the device, then recycle slot 0 */
        HIPCHK(hipDeviceSynchronize());
        free_k = 0;
    }
    k = free_k;
    if (need_spill && !a->t2[k].spill) {
        const size_t lanes = (size_t)a->t2_grid * LH_BLOCK;
        HIPCHK(hipMalloc((void **)&a->t2[k].spill, lanes * 64 * sizeof(int)));
    }
    if (!a->t2[k].queue) {
        HIPCHK(hipMalloc((void **)&a->t2[k].queue, (size_t)LH_T2_QCAP * 6 * sizeof(uint32_t) + 2 * sizeof(uint32_t)));
        a->t2[k].qcount = a->t2[k].queue + (size_t)LH_T2_QCAP * 6;
    }
    a->t2[k].stream = s; a->t2[k].used = 1;
    return k;
}

/* hot set of a ray dump (4-wide nodes + 48-byte triangle records) against the Infinity Cache */
static bool wide8_pays(const lh_accel_t *a)
{
    const lh_bvh_t *b = &a->hs->bvh;
    return sizeof(lh_q4node_t) * (size_t)b->nq4nodes + sizeof(lh_tri32_t) * (size_t)b->ntris > ((size_t)256 << 20);
}

static int launch(lh_accel_t *a, size_t n, const void *d_org, const void *d_dir, void *d_prim,
                  void *d_t, void *d_u, void *d_v, void *d_occ, int mode, int variant,
                  unsigned long long *d_counters, hipStream_t s, bool dump = false)
{
    if (!a || !a->committed) return fail("intersect: accel not committed");
    if (n == 0) return 0;
    if (!d_org || !d_dir) return fail("intersect: NULL ray arrays");
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
    if (a->hs->device_built && !a->d_ref_nodes && sync_ref(a, false) != 0) return -1;     /* background reference tree: attach when ready */
    if (variant == LH_VARIANT_DEFAULT) variant = a->default_variant;
    if (variant < 0 || variant > LH_VARIANT_QUAD) return fail("intersect: unknown variant %d", variant);
    /* a tree built on the device exists only as 4-wide nodes: the A/B walks over other formats run as the default walk */
    if (a->hs->device_built && variant != LH_VARIANT_SPEC && variant != LH_VARIANT_LEAN && variant != LH_VARIANT_QUAD) variant = LH_VARIANT_SPEC;
    if (variant == LH_VARIANT_LEAN) {
        /* the lean walk reads the 4-wide 16-bit-grid nodes; scenes it cannot take (another format forced by
         * LH_NODE_FORMAT, a tree too deep for its 64-entry logical stack) go through the r01 walk */
        if (a->dev.use_qnodes != 2 || 3 * a->dev.q4_depth + 5 > 64) variant = LH_VARIANT_SPEC;
        else {
            int k = t2_slot(a, s, true);
            if (k < 0) return -1;
            int rc = lh_launch_trace2(&a->dev, n, (const double *)d_org, (const double *)d_dir, (uint32_t *)d_prim, (double *)d_t,
                                      (double *)d_u, (double *)d_v, mode == LH_MODE_ANY, (uint8_t *)d_occ, d_counters,
                                      a->d_cursor + (a->cursor_next++ % LH_NCURSOR), a->t2_grid, a->min_active, a->tri_batch,
                                      a->t2[k].spill, a->t2[k].queue, a->t2[k].qcount, LH_T2_QCAP, (void *)s);
            if (rc != 0) return fail("kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
            return 0;
        }
    }
    if (ensure_formats(a, lh_trace_formats_needed(&a->dev, variant)) != 0) return -1;     /* A/B formats: uploaded on first use */
    /* ray dumps (incoherent by assumption) over a scene whose hot set does not fit the 256 MiB Infinity Cache walk the 8-wide
     * nodes: each record then costs a 128-byte line of HBM traffic whatever its size, and an 8-wide record uses all of it
     * (S-soup-10M: 57 -> 40 records per ray).  The tile pipelines' coherent rays stay on the 4-wide nodes. */
    a->dev.prefer_q8 = 0;
    if (dump && variant == LH_VARIANT_SPEC && a->dev.use_qnodes == 2 && !a->hs->device_built &&
        (a->wide8 == 1 || (a->wide8 == -1 && wide8_pays(a)))) {
        if (ensure_formats(a, LH_FMT_Q8) != 0) return -1;
        a->dev.prefer_q8 = 1;
    }
    int rc = lh_launch_trace(&a->dev, n, (const double *)d_org, (const double *)d_dir, (uint32_t *)d_prim,
                             (double *)d_t, (double *)d_u, (double *)d_v, mode == LH_MODE_ANY,
                             (uint8_t *)d_occ, d_counters, a->d_cursor + (a->cursor_next++ % LH_NCURSOR), variant, (variant == LH_VARIANT_QUAD && a->dev.q4tnodes) ? a->quad_grid : a->grid_blocks, a->min_active, a->tri_batch, (void *)s);
    if (rc != 0) return fail("kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int lh_accel_intersect_device(lh_accel_t *a, size_t n, const void *d_org, const void *d_dir,
                                         void *d_prim, void *d_t, void *d_u, void *d_v, void *d_occ,
                                         int mode, int variant, void *stream)
{
    lh_guard guard(a);
    return launch(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occ, mode, variant, NULL, (hipStream_t)stream, true);
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
    int rc = launch(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occ, mode, variant, a->d_counters, a->stream, true);
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
    if (a->default_variant == LH_VARIANT_SPEC && a->dev.use_qnodes == 2 && !a->hs->device_built &&
        (a->wide8 == 1 || (a->wide8 == -1 && wide8_pays(a)))) return (int)sizeof(lh_q8node_t);
    return a->dev.use_qnodes == 0 ? 64 : a->dev.use_qnodes == 1 ? 32 : a->dev.use_qnodes == 3 ? 80 : 64;
}

extern "C" uint64_t lh_accel_last_retraced(const lh_accel_t *a) { return a ? a->last_retraced : 0; }

static int ensure_stage(lh_accel_t *a, size_t bytes)
{
    if (a->stage_bytes >= bytes) return 0;
    if (a->d_stage) { (void)hipFree(a->d_stage); a->d_stage = NULL; a->stage_bytes = 0; }
    HIPCHK(hipMalloc(&a->d_stage, bytes));
    a->stage_bytes = bytes;
    return 0;
}

/* ---- large host batches: chunks pipelined through pinned staging -----------------------------------
 * A pageable hipMemcpy moves ~12 GB/s; the link does ~50.  Rays are cut into chunks of LH_PIPE_CHUNK;
 * chunk k is copied into pinned memory by a few host threads, sent, traced and brought back on stream
 * k & 1 while the host stages chunk k+1 and un-stages chunk k-1. */
#define LH_PIPE_CHUNK ((size_t)1 << 21)
#define LH_PIPE_MIN   ((size_t)1 << 20)

static void par_copy(void *dst, const void *src, size_t bytes)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = bytes < ((size_t)8 << 20) ? 1 : (hw >= 16 ? 8 : (hw >= 4 ? 4 : 1));
    if (nt == 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
    for (size_t k = 0; k < nt; k++) {
        const size_t b = k * per; if (b >= bytes) break;
        const size_t e = (b + per < bytes) ? b + per : bytes;
        th.emplace_back([=] { memcpy((char *)dst + b, (const char *)src + b, e - b); });
    }
    for (auto &t : th) t.join();
}

static int pipe_init(lh_accel_t *a)
{
    if (a->pipe.ready) return 0;
    const size_t C = LH_PIPE_CHUNK;
    const size_t in_b = sizeof(double) * 6 * C, out_b = (sizeof(double) * 3 + sizeof(uint32_t)) * C;
    for (int b = 0; b < 2; b++) {
        HIPCHK(hipHostMalloc(&a->pipe.h_in[b], in_b, hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&a->pipe.h_out[b], out_b, hipHostMallocDefault));
        HIPCHK(hipMalloc(&a->pipe.d_in[b], in_b));
        HIPCHK(hipMalloc(&a->pipe.d_out[b], out_b));
        HIPCHK(hipStreamCreateWithFlags(&a->pipe.s[b], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&a->pipe.done[b], hipEventDisableTiming));
    }
    a->pipe.cap = C; a->pipe.ready = 1;
    return 0;
}

static int intersect_host_pipelined(lh_accel_t *a, size_t n, const double *org, const double *dir,
                                    uint32_t *prim, double *t, double *u, double *v, uint8_t *occ, int mode)
{
    if (pipe_init(a) != 0) return -1;
    const size_t C = a->pipe.cap, nchunks = (n + C - 1) / C;
    auto unstage = [&](size_t k) {
        const int b = (int)(k & 1); const size_t first = k * C, m = (first + C <= n) ? C : n - first;
        const char *ho = (const char *)a->pipe.h_out[b];
        if (mode == LH_MODE_CLOSEST) {
            if (t) par_copy(t + first, ho, sizeof(double) * m);
            if (u) par_copy(u + first, ho + sizeof(double) * C, sizeof(double) * m);
            if (v) par_copy(v + first, ho + 2 * sizeof(double) * C, sizeof(double) * m);
            if (prim) par_copy(prim + first, ho + 3 * sizeof(double) * C, sizeof(uint32_t) * m);
        } else if (occ) par_copy(occ + first, ho, m);
    };
    for (size_t k = 0; k < nchunks; k++) {
        const int b = (int)(k & 1); const size_t first = k * C, m = (first + C <= n) ? C : n - first;
        if (k >= 2) { HIPCHK(hipEventSynchronize(a->pipe.done[b])); unstage(k - 2); }
        char *hi = (char *)a->pipe.h_in[b], *di = (char *)a->pipe.d_in[b], *dout = (char *)a->pipe.d_out[b];
        par_copy(hi, org + 3 * first, sizeof(double) * 3 * m);
        par_copy(hi + sizeof(double) * 3 * C, dir + 3 * first, sizeof(double) * 3 * m);
        hipStream_t s = a->pipe.s[b];
        HIPCHK(hipMemcpyAsync(di, hi, sizeof(double) * 3 * m, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(di + sizeof(double) * 3 * C, hi + sizeof(double) * 3 * C, sizeof(double) * 3 * m, hipMemcpyHostToDevice, s));
        double *d_t = (double *)dout, *d_u = d_t + C, *d_v = d_u + C; uint32_t *d_prim = (uint32_t *)(d_v + C);
        const int rc = launch(a, m, di, di + sizeof(double) * 3 * C, d_prim, d_t, d_u, d_v, (uint8_t *)dout, mode, LH_VARIANT_DEFAULT, NULL, s, true);
        if (rc != 0) return rc;
        char *ho = (char *)a->pipe.h_out[b];
        if (mode == LH_MODE_CLOSEST) {
            if (t) HIPCHK(hipMemcpyAsync(ho, d_t, sizeof(double) * m, hipMemcpyDeviceToHost, s));
            if (u) HIPCHK(hipMemcpyAsync(ho + sizeof(double) * C, d_u, sizeof(double) * m, hipMemcpyDeviceToHost, s));
            if (v) HIPCHK(hipMemcpyAsync(ho + 2 * sizeof(double) * C, d_v, sizeof(double) * m, hipMemcpyDeviceToHost, s));
            if (prim) HIPCHK(hipMemcpyAsync(ho + 3 * sizeof(double) * C, d_prim, sizeof(uint32_t) * m, hipMemcpyDeviceToHost, s));
        } else if (occ) HIPCHK(hipMemcpyAsync(ho, dout, m, hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(a->pipe.done[b], s));
    }
    for (size_t k = (nchunks >= 2 ? nchunks - 2 : 0); k < nchunks; k++) {
        HIPCHK(hipEventSynchronize(a->pipe.done[k & 1])); unstage(k);
    }
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
    if (n >= LH_PIPE_MIN && !a->stat_on && !getenv("LH_HOST_SIMPLE"))
        return intersect_host_pipelined(a, n, org, dir, prim, t, u, v, occ, mode);
    /* layout of the staging block: org | dir | t | u | v | prim | occ */
    const size_t b_ray = sizeof(double) * 3 * n, b_d = sizeof(double) * n;
    const size_t total = 2 * b_ray + 3 * b_d + sizeof(uint32_t) * n + n + 64;
    if (ensure_stage(a, total) != 0) return -1;
    char *base = (char *)a->d_stage;
    double *d_org = (double *)base, *d_dir = (double *)(base + b_ray);
    double *d_t = (double *)(base + 2 * b_ray), *d_u = d_t + n, *d_v = d_u + n;
    uint32_t *d_prim = (uint32_t *)(d_v + n);
    uint8_t *d_occ = (uint8_t *)(d_prim + n);
    HIPCHK(hipMemcpyAsync(d_org, org, b_ray, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_dir, dir, b_ray, hipMemcpyHostToDevice, a->stream));
    if (a->stat_on) HIPCHK(hipMemsetAsync(a->d_counters, 0, sizeof(unsigned long long) * LH_CNT_DEV, a->stream));
    int rc = launch(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occ, mode, LH_VARIANT_DEFAULT,
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

extern "C" int lh_accel_intersect1(lh_accel_t *a, const double org[3], const double dir[3],
                                   uint32_t *prim, double *t, double *u, double *v)
{
    uint32_t p = LH_MISS_PRIM; double tt = LH_T_INF, uu = 0.0, vv = 0.0;
    if (!org || !dir) return fail("lh_accel_intersect1: NULL ray");
    if (lh_accel_intersect_host(a, 1, org, dir, &p, &tt, &uu, &vv, NULL, LH_MODE_CLOSEST) != 0) return -1;
    if (prim) *prim = p;
    if (t) *t = tt;
    if (u) *u = uu;
    if (v) *v = vv;
    return p != LH_MISS_PRIM;
}

/* ------------------------------------------------------------------------ */
/* tile rendering                                                           */
/* ------------------------------------------------------------------------ */

static int ensure_buf(lh_buf *b, size_t bytes)
{
    if (b->cap >= bytes && b->p) return 0;
    if (b->p) { (void)hipFree(b->p); b->p = NULL; b->cap = 0; }
    if (bytes == 0) bytes = 16;
    HIPCHK(hipMalloc(&b->p, bytes));
    b->cap = bytes;
    return 0;
}

extern "C" int lh_render_primary_rays(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h,
                                      int ps, void *d_org, void *d_dir, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_primary_rays: accel not committed");
    if (!cam || !d_org || !d_dir) return fail("lh_render_primary_rays: NULL argument");
    if (w < 0 || h < 0 || ps < 1) return fail("lh_render_primary_rays: bad tile");
    HIPCHK(hipSetDevice(a->device));
    if (lh_render_launch_primary(cam, x0, y0, w, h, ps, ps, (double *)d_org, (double *)d_dir, stream) != 0)
        return fail("primary ray kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int lh_accel_add_rib_scene(lh_accel_t *a, const lh_rib_scene_t *scene)
{
    lh_guard guard(a);
    lh_rib_info_t info;
    if (!a || !scene) return fail("lh_accel_add_rib_scene: NULL argument");
    if (lh_rib_info(scene, &info) != 0) return fail("lh_accel_add_rib_scene: %s", lh_rib_last_error());
    for (uint32_t m = 0; m < info.nmeshes; m++) {
        uint32_t npos = 0, nidx = 0; const double *pos = NULL, *nrm = NULL; const uint32_t *idx = NULL; int two = 0;
        if (lh_rib_mesh(scene, m, &npos, &pos, &nidx, &idx, &nrm, &two) != 0) return fail("lh_accel_add_rib_scene: %s", lh_rib_last_error());
        const uint32_t ord = a->nmeshes;          /* add_mesh returns 0 / -1, not the ordinal */
        if (lh_accel_add_mesh(a, npos, pos, 4 * sizeof(double), nidx, idx) != 0) return -1;
        if (nrm && lh_accel_set_normals(a, ord, nrm, 4 * sizeof(double), two) != 0) return -1;
    }
    return 0;
}

/* one device batch of the AO pipeline over a Region (lh_render.hip): a rectangle, or nbands full-width bands */
static int ao_region(lh_accel_t *a, const lh_camera_t *cam, int x0, int w, int nbands, int band_rows, const int *d_band_y0, int y0,
                     uint64_t valid_pixels, int ps, int gather_nsamples, uint64_t seed, const void *d_uniforms, void *d_rgb,
                     lh_tile_stats_t *stats, void *stream)
{
    const int h = nbands * band_rows;              /* lines of the batch */
    HIPCHK(hipSetDevice(a->device));
    hipStream_t s = (hipStream_t)stream;
    const int nphi = (int)sqrt((double)gather_nsamples), ntheta = nphi, N = nphi * ntheta;   /* ambientocclusion.c:378-380 */
    const size_t S = (size_t)w * h * ps * ps;
    if ((unsigned long long)cam->width * (unsigned long long)cam->height * (unsigned long long)(ps * ps) >= (1ull << 34))
        return fail("AO pipeline: more than 2^34 samples in the frame (slot keys carry 34 bits)");
    const unsigned nb = (unsigned)((S + 255) / 256);
    if (ensure_buf(&a->r_org, S * 24) || ensure_buf(&a->r_dir, S * 24) || ensure_buf(&a->r_prim, S * 4) ||
        ensure_buf(&a->r_t, S * 8) || ensure_buf(&a->r_u, S * 8) || ensure_buf(&a->r_v, S * 8) ||
        ensure_buf(&a->r_slot, S * 4) || ensure_buf(&a->r_blocks, (size_t)nb * 4)) return -1;
    /* lh_accel_trace_statistics: the counting instantiations of the same kernels, accumulated over the batch */
    unsigned long long *cnt = (a->stat_on && a->hs->bvh.ntris) ? a->d_counters : NULL;
    if (cnt) HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * LH_CNT_DEV, s));
    /* 1. camera rays */
    if (lh_render_launch_primary_region(cam, x0, w, nbands, band_rows, d_band_y0, y0, cam->height, ps, ps,
                                        (double *)a->r_org.p, (double *)a->r_dir.p, s) != 0)
        return fail("primary ray kernel launch failed");
    /* 2. closest hit */
    if (launch(a, S, a->r_org.p, a->r_dir.p, a->r_prim.p, a->r_t.p, a->r_u.p, a->r_v.p, NULL, LH_MODE_CLOSEST,
               LH_VARIANT_DEFAULT, cnt, s) != 0) return -1;
    /* 3. count hits (deterministic compaction needs the total before sizing the AO batch) */
    unsigned long long nhit = 0;
    if (a->hs->bvh.ntris) {
        if (ensure_buf(&a->r_hitrec, S * 96) || ensure_buf(&a->r_key, S * 8)) return -1;   /* worst case: every sample hits */
        if (lh_render_launch_compact(&a->dev, (const double *)a->d_nrm9, S, (const double *)a->r_org.p,
                                     (const double *)a->r_dir.p, (const uint32_t *)a->r_prim.p, (const double *)a->r_t.p,
                                     (const double *)a->r_u.p, (const double *)a->r_v.p, (uint32_t *)a->r_blocks.p,
                                     (uint32_t *)a->r_slot.p, (double *)a->r_hitrec.p, (unsigned long long *)a->r_key.p,
                                     x0, w, nbands, band_rows, d_band_y0, y0, ps * ps, cam->width, a->d_total, s) != 0)
            return fail("compaction kernels failed: %s", hipGetErrorString(hipGetLastError()));
        HIPCHK(hipMemcpyAsync(&nhit, a->d_total, sizeof(nhit), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    } else {
        HIPCHK(hipMemsetAsync(a->r_slot.p, 0xFF, S * 4, s));
    }
    const size_t nao = (size_t)nhit * N;
    unsigned long long nocc = 0;
    /* AO stage.  Fused (default): the any-hit kernel generates ray (slot, r) in its refill (lh_ao.h) and counts
     * the occluded rays per slot -- nothing per AO ray goes through HBM.  Materialised: caller uniforms (the parity
     * replay), LH_AO_FUSED=0, a scene the lean walk cannot take, or a pending-queue overflow of the fused launch. */
    bool fused = a->ao_fused && !d_uniforms && nao && nao < ((size_t)1 << 32) && a->dev.use_qnodes == 2 &&
                 ((3 * a->dev.q4_depth + 5 <= 64 && !a->dev.stack_cap) || a->dev.ref_nodes != NULL);     /* deeper: rays whose stack would overflow go to the reference walk */
    const bool fused_tried = fused;
    if (fused) {
        if (ensure_buf(&a->r_occcount, (size_t)nhit * sizeof(unsigned int))) return -1;
        const int k = t2_slot(a, s, a->default_variant == LH_VARIANT_LEAN);
        if (k < 0) return -1;
        int rcf;
        if (a->default_variant == LH_VARIANT_LEAN)      /* A/B: the lean walk's AO source (its pending queue fills up on large tiles) */
            rcf = lh_launch_trace2_ao(&a->dev, (size_t)nhit, ntheta, nphi, seed, (const double *)a->r_hitrec.p, (const unsigned long long *)a->r_key.p,
                                      (unsigned int *)a->r_occcount.p, NULL, a->d_cursor + (a->cursor_next++ % LH_NCURSOR), a->t2_grid,
                                      a->min_active, a->tri_batch, a->t2[k].spill, a->t2[k].queue, a->t2[k].qcount, LH_T2_QCAP, (void *)s);
        else {
            rcf = lh_launch_trace_ao(&a->dev, (size_t)nhit, ntheta, nphi, seed, (const double *)a->r_hitrec.p, (const unsigned long long *)a->r_key.p,
                                     (unsigned int *)a->r_occcount.p, cnt, a->d_cursor + (a->cursor_next++ % LH_NCURSOR), a->grid_blocks,
                                     a->min_active, a->tri_batch, a->t2[k].queue, a->t2[k].qcount, LH_T2_QCAP, (void *)s);
            if (rcf == 0) rcf = lh_launch_ao_queue(&a->dev, ntheta, nphi, seed, (const double *)a->r_hitrec.p, (const unsigned long long *)a->r_key.p,
                                                   (unsigned int *)a->r_occcount.p, NULL, a->t2[k].queue, a->t2[k].qcount, LH_T2_QCAP, (void *)s);
        }
        if (rcf != 0) return fail("fused AO launch failed: %s", hipGetErrorString(hipGetLastError()));
        uint32_t qc[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(qc, a->t2[k].qcount, sizeof(qc), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (qc[1] != 0) fused = false;             /* more than LH_T2_QCAP uncertain AO rays: redo the stage materialised */
    }
    if (nao && !fused) {
        if (ensure_buf(&a->r_aorg, nao * 24) || ensure_buf(&a->r_adir, nao * 24) || ensure_buf(&a->r_occ, nao)) return -1;
        /* 4. AO rays */
        if (lh_render_launch_ao_rays(nhit, ntheta, nphi, seed, (const double *)a->r_hitrec.p, (const double *)d_uniforms,
                                     (const unsigned long long *)a->r_key.p, (double *)a->r_aorg.p, (double *)a->r_adir.p, s) != 0)
            return fail("AO ray kernel launch failed");
        /* 5. any-hit */
        if (cnt && fused_tried) HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * LH_CNT_DEV, s));   /* the abandoned fused pass is not counted (nor are the camera rays then) */
        if (launch(a, nao, a->r_aorg.p, a->r_adir.p, NULL, NULL, NULL, NULL, a->r_occ.p, LH_MODE_ANY,
                   LH_VARIANT_DEFAULT, cnt, s) != 0) return -1;
    }
    /* 6. radiance */
    HIPCHK(hipMemsetAsync(a->d_total, 0, sizeof(unsigned long long), s));
    if (lh_render_launch_resolve(w, h, band_rows, ps, ps, N, (const uint32_t *)a->r_slot.p, (const uint8_t *)a->r_occ.p,
                                 fused ? (const unsigned int *)a->r_occcount.p : NULL, (float *)d_rgb, a->d_total, s) != 0)
        return fail("resolve kernel launch failed");
    HIPCHK(hipMemcpyAsync(&nocc, a->d_total, sizeof(nocc), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    a->r_nsamples = S; a->r_nslots = (size_t)nhit; a->r_nao = fused ? 0 : nao;
    if (cnt) {
        unsigned long long hc[LH_CNT_N];
        HIPCHK(hipMemcpy(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost));
        a->stat[0] += hc[LH_CNT_NODES]; a->stat[1] += hc[LH_CNT_TRIS]; a->stat[2] += hc[LH_CNT_EXACT];
        a->stat[3] += hc[LH_CNT_RAYS]; a->stat[4] += nhit + nocc;
    }
    if (stats) {
        stats->primary_rays = valid_pixels * (uint64_t)(ps * ps); stats->primary_hits = nhit; stats->ao_rays = nao; stats->ao_occluded = nocc;
    }
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

extern "C" int lh_render_ao_tile(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int ps,
                                 int gather_nsamples, uint64_t seed, const void *d_uniforms, void *d_rgb,
                                 lh_tile_stats_t *stats, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_ao_tile: accel not committed");
    if (!cam || !d_rgb) return fail("lh_render_ao_tile: NULL argument");
    if (w <= 0 || h <= 0 || ps < 1 || gather_nsamples < 1) return fail("lh_render_ao_tile: bad tile/sample counts");
    return ao_region(a, cam, x0, w, 1, h, NULL, y0, (uint64_t)w * h, ps, gather_nsamples, seed, d_uniforms, d_rgb, stats, stream);
}

/* nbands full-width bands of band_rows lines (band b = frame lines band_y0[b] ...; a band that runs past the frame is
 * clipped) as ONE device batch: how a rank renders all of its interleaved shards of a frame with one set of launches.
 * d_rgb: float[nbands][band_rows][width][3], every band in image orientation (top line first) like a tile. */
extern "C" int lh_render_ao_bands(lh_accel_t *a, const lh_camera_t *cam, int nbands, const int *band_y0, int band_rows, int ps,
                                  int gather_nsamples, uint64_t seed, void *d_rgb, lh_tile_stats_t *stats, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_ao_bands: accel not committed");
    if (!cam || !d_rgb || (nbands > 0 && !band_y0)) return fail("lh_render_ao_bands: NULL argument");
    if (nbands < 0 || band_rows <= 0 || ps < 1 || gather_nsamples < 1) return fail("lh_render_ao_bands: bad band/sample counts");
    if (nbands == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return 0; }
    HIPCHK(hipSetDevice(a->device));
    uint64_t valid = 0;
    for (int b = 0; b < nbands; b++) {
        if (band_y0[b] < 0 || band_y0[b] >= cam->height) return fail("lh_render_ao_bands: band %d starts at line %d outside the frame", b, band_y0[b]);
        const int rows = (band_y0[b] + band_rows <= cam->height) ? band_rows : cam->height - band_y0[b];
        valid += (uint64_t)rows * cam->width;
    }
    if (ensure_buf(&a->r_bands, sizeof(int) * (size_t)nbands)) return -1;
    HIPCHK(hipMemcpyAsync(a->r_bands.p, band_y0, sizeof(int) * (size_t)nbands, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));          /* band_y0 is the caller's memory */
    return ao_region(a, cam, 0, cam->width, nbands, band_rows, (const int *)a->r_bands.p, 0, valid, ps, gather_nsamples, seed, NULL,
                     d_rgb, stats, stream);
}

/* the same tile for a plain-C host program: uniforms (optional) come from and the tile goes to HOST memory */
extern "C" int lh_render_ao_tile_host(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int ps,
                                      int gather_nsamples, uint64_t seed, const double *uniforms, size_t nuniforms,
                                      float *rgb, lh_tile_stats_t *stats)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_ao_tile_host: accel not committed");
    if (!cam || !rgb) return fail("lh_render_ao_tile_host: NULL argument");
    if (w <= 0 || h <= 0) return fail("lh_render_ao_tile_host: bad tile");
    HIPCHK(hipSetDevice(a->device));
    const size_t fb = (size_t)w * h * 3 * sizeof(float);
    if (ensure_buf(&a->r_frame, fb)) return -1;
    void *d_uni = NULL;
    if (uniforms) {
        const int nphi = (int)sqrt((double)gather_nsamples);
        const size_t need = (size_t)2 * nphi * nphi * w * h * ps * ps;       /* worst case: every sample hits */
        if (nuniforms < need) return fail("lh_render_ao_tile_host: %zu uniforms given, the tile may consume %zu", nuniforms, need);
        if (ensure_buf(&a->r_uni, need * sizeof(double))) return -1;
        HIPCHK(hipMemcpyAsync(a->r_uni.p, uniforms, need * sizeof(double), hipMemcpyHostToDevice, a->stream));
        d_uni = a->r_uni.p;
    }
    if (lh_render_ao_tile(a, cam, x0, y0, w, h, ps, gather_nsamples, seed, d_uni, a->r_frame.p, stats, a->stream) != 0) return -1;
    HIPCHK(hipMemcpyAsync(rgb, a->r_frame.p, fb, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}

extern "C" int lh_render_scratch(lh_accel_t *a, int which, void **d_ptr, size_t *count)
{
    lh_guard guard(a);
    if (!a || !a->committed || !d_ptr || !count) return fail("lh_render_scratch: bad argument");
    lh_buf *b[] = {&a->r_org, &a->r_dir, &a->r_prim, &a->r_t, &a->r_u, &a->r_v, &a->r_slot, &a->r_hitrec,
                   &a->r_aorg, &a->r_adir, &a->r_occ};
    if (which < 0 || which > 10) return fail("lh_render_scratch: unknown buffer %d", which);
    *d_ptr = b[which]->p;
    *count = which <= 6 ? a->r_nsamples : (which == 7 ? a->r_nslots : a->r_nao);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* hit epilogue for a batch (ri_intersection_state_build)                   */
/* ------------------------------------------------------------------------ */
extern "C" int lh_render_launch_state_build(size_t n, const lh_dev_scene_t *sc, const double *d_nrm9, const double *d_col9,
                                            const double *d_tan9, const double *d_bin9, const double *d_st6, const uint8_t *d_inside,
                                            const double *d_org, const double *d_dir, const uint32_t *d_prim, const double *d_t,
                                            const double *d_u, const double *d_v, double *d_state, void *stream);

extern "C" int lh_accel_state_build_device(lh_accel_t *a, size_t n, const void *d_org, const void *d_dir, const void *d_prim,
                                           const void *d_t, const void *d_u, const void *d_v, void *d_state, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_accel_state_build_device: accel not committed");
    if (n == 0 || a->hs->bvh.ntris == 0) return 0;
    if (!d_org || !d_dir || !d_prim || !d_t || !d_u || !d_v || !d_state) return fail("lh_accel_state_build_device: NULL argument");
    HIPCHK(hipSetDevice(a->device));
    if (lh_render_launch_state_build(n, &a->dev, (const double *)a->d_nrm9, (const double *)a->d_attr9[0], (const double *)a->d_attr9[1],
                                     (const double *)a->d_attr9[2], (const double *)a->d_st6, (const uint8_t *)a->d_inside,
                                     (const double *)d_org, (const double *)d_dir, (const uint32_t *)d_prim, (const double *)d_t,
                                     (const double *)d_u, (const double *)d_v, (double *)d_state, stream) != 0)
        return fail("state-build kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int lh_accel_state_build_host(lh_accel_t *a, size_t n, const double *org, const double *dir, const uint32_t *prim,
                                         const double *t, const double *u, const double *v, double *state)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_accel_state_build_host: accel not committed");
    if (n == 0) return 0;
    if (!org || !dir || !prim || !t || !u || !v || !state) return fail("lh_accel_state_build_host: NULL argument");
    HIPCHK(hipSetDevice(a->device));
    const size_t b_ray = sizeof(double) * 3 * n, b_d = sizeof(double) * n, b_state = sizeof(double) * LH_STATE_DOUBLES * n;
    if (ensure_buf(&a->r_state, 2 * b_ray + 3 * b_d + sizeof(uint32_t) * n + 8 + b_state)) return -1;
    char *base = (char *)a->r_state.p;
    double *d_state = (double *)base, *d_org = (double *)(base + b_state), *d_dir = d_org + 3 * n, *d_t = d_dir + 3 * n, *d_u = d_t + n, *d_v = d_u + n;
    uint32_t *d_prim = (uint32_t *)(d_v + n);
    HIPCHK(hipMemcpyAsync(d_org, org, b_ray, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_dir, dir, b_ray, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_t, t, b_d, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_u, u, b_d, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_v, v, b_d, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_prim, prim, sizeof(uint32_t) * n, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemsetAsync(d_state, 0, b_state, a->stream));
    if (lh_accel_state_build_device(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_state, a->stream) != 0) return -1;
    HIPCHK(hipMemcpyAsync(state, d_state, b_state, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}

/* ------------------------------------------------------------------------ */
/* materials / environment of the path tracer                               */
/* ------------------------------------------------------------------------ */
extern "C" int lh_accel_set_material(lh_accel_t *a, uint32_t mesh, const lh_material_t *mat)
{
    lh_guard guard(a);
    if (!a || !mat) return fail("lh_accel_set_material: NULL argument");
    const uint32_t nm = a->committed ? a->hs->nmeshes : a->nmeshes;
    if (mesh != LH_ALL_MESHES && mesh >= nm) return fail("lh_accel_set_material: mesh %u out of range", mesh);
    for (int k = 0; k < 3; k++) {
        if (!(mat->kd[k] >= 0.0f && mat->ks[k] >= 0.0f && mat->kt[k] >= 0.0f)) return fail("lh_accel_set_material: negative or NaN reflectance");
    }
    const double sum = (mat->kd[0] + mat->kd[1] + mat->kd[2] + mat->ks[0] + mat->ks[1] + mat->ks[2] + mat->kt[0] + mat->kt[1] + mat->kt[2]) / 3.0;
    if (sum > 1.0 + 1e-6) return fail("lh_accel_set_material: kd + ks + kt averages exceed 1 (pathtrace.c:419 asserts d + s + t <= 1)");
    if (!(mat->ior > 0.0f)) return fail("lh_accel_set_material: ior must be positive");
    if (a->nmaterials < nm) {
        lh_material_t *nmats = (lh_material_t *)realloc(a->materials, sizeof(lh_material_t) * (nm ? nm : 1));
        if (!nmats) return fail("out of memory");
        for (uint32_t k = a->nmaterials; k < nm; k++) {        /* ri_material_new (material.c:20-40): kd 1, ks 0, kt 0, ior 1 */
            memset(&nmats[k], 0, sizeof(lh_material_t));
            nmats[k].kd[0] = nmats[k].kd[1] = nmats[k].kd[2] = 1.0f; nmats[k].ior = 1.0f;
        }
        a->materials = nmats; a->nmaterials = nm;
    }
    for (uint32_t k = 0; k < a->nmaterials; k++) if (mesh == LH_ALL_MESHES || mesh == k) a->materials[k] = *mat;
    a->materials_dirty = 1;
    return 0;
}

extern "C" int lh_accel_set_environment(lh_accel_t *a, const lh_environment_t *env)
{
    lh_guard guard(a);
    if (!a || !env) return fail("lh_accel_set_environment: NULL argument");
    if (!a->committed) return fail("lh_accel_set_environment: accel not committed");
    if (env->map_rgba && (env->width < 1 || env->height < 1)) return fail("lh_accel_set_environment: bad map size");
    HIPCHK(hipSetDevice(a->device));
    if (a->d_env_map) { (void)hipFree(a->d_env_map); a->d_env_map = NULL; }
    a->env = *env; a->env.map_rgba = NULL;
    if (env->map_rgba) {
        const size_t b = sizeof(float) * 4 * (size_t)env->width * env->height;
        HIPCHK(hipMalloc(&a->d_env_map, b));
        HIPCHK(hipMemcpy(a->d_env_map, env->map_rgba, b, hipMemcpyHostToDevice));
    }
    return 0;
}

static int sync_materials(lh_accel_t *a)
{
    const uint32_t nm = a->hs->nmeshes ? a->hs->nmeshes : 1;
    if (a->nmaterials < nm) {
        lh_material_t def; memset(&def, 0, sizeof(def)); def.kd[0] = def.kd[1] = def.kd[2] = 1.0f; def.ior = 1.0f;
        lh_material_t *nmats = (lh_material_t *)realloc(a->materials, sizeof(lh_material_t) * nm);
        if (!nmats) return fail("out of memory");
        for (uint32_t k = a->nmaterials; k < nm; k++) nmats[k] = def;
        a->materials = nmats; a->nmaterials = nm; a->materials_dirty = 1;
    }
    if (a->materials_dirty || !a->d_materials) {
        if (a->d_materials) { (void)hipFree(a->d_materials); a->d_materials = NULL; }
        HIPCHK(hipMalloc(&a->d_materials, sizeof(lh_material_t) * a->nmaterials));
        HIPCHK(hipMemcpy(a->d_materials, a->materials, sizeof(lh_material_t) * a->nmaterials, hipMemcpyHostToDevice));
        a->materials_dirty = 0;
    }
    if (!a->d_prim_mesh && a->hs->bvh.ntris) {
        HIPCHK(hipMalloc(&a->d_prim_mesh, sizeof(uint32_t) * (size_t)a->hs->bvh.ntris));
        HIPCHK(hipMemcpy(a->d_prim_mesh, a->hs->bvh.prim_geom, sizeof(uint32_t) * (size_t)a->hs->bvh.ntris, hipMemcpyHostToDevice));
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* beam visibility                                                          */
/* ------------------------------------------------------------------------ */
extern "C" int lh_launch_beam_visibility(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dirs,
                                         int32_t *d_result, void *stream);

__global__ void k_fill_i32(size_t n, int32_t *p, int32_t v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

extern "C" int lh_accel_beam_visibility_device(lh_accel_t *a, size_t n, const void *d_org, const void *d_dirs,
                                               void *d_result, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("beam_visibility: accel not committed");
    if (n == 0) return 0;
    if (!d_org || !d_dirs || !d_result) return fail("beam_visibility: NULL argument");
    if (!a->hs->have_ref) return fail("beam_visibility: the reference-order tree was disabled (LH_REFTREE=0)");
    if (sync_ref(a, true) != 0) return -1;
    HIPCHK(hipSetDevice(a->device));
    lh_dev_scene_t sc = a->dev;
    if (a->hs->bvh.ntris == 0) { sc.ref_empty = 1; }
    if (lh_launch_beam_visibility(&sc, n, (const double *)d_org, (const double *)d_dirs, (int32_t *)d_result, stream) != 0)
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
    if (ensure_stage(a, bo + bd + br + 64) != 0) return -1;
    char *base = (char *)a->d_stage;
    HIPCHK(hipMemcpyAsync(base, org, bo, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(base + bo, dirs, bd, hipMemcpyHostToDevice, a->stream));
    if (lh_accel_beam_visibility_device(a, n, base, base + bo, base + bo + bd, a->stream) != 0) return -1;
    HIPCHK(hipMemcpyAsync(result, base + bo + bd, br, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}

/* ------------------------------------------------------------------------ */
/* wavefront path tracer                                                    */
/* ------------------------------------------------------------------------ */
extern "C" int lh_pt_launch_primary(const lh_camera_t *cam, int x0, int y0, int w, int h, int spp, int s0,
                                    unsigned long long seed, double *d_org, double *d_dir, uint32_t *d_path_of,
                                    float *d_thr, void *stream);
extern "C" int lh_pt_launch_shade(size_t n, const lh_dev_scene_t *sc, const double *d_nrm9, const double *d_col9,
                                  const uint32_t *d_prim_mesh, const void *d_materials, const lh_material_t *override_mat,
                                  const float env_rgb[3], const void *d_env_map, int env_w, int env_h, int ref_weights,
                                  int depth, int max_depth, unsigned long long seed, int s0, int spp, int x0, int y0, int w,
                                  int full_width, double *d_org, double *d_dir, const uint32_t *d_prim,
                                  const double *d_t, const double *d_u, const double *d_v, uint32_t *d_path_of,
                                  float *d_thr, float *d_radiance, uint8_t *d_alive, uint32_t *d_blocks,
                                  unsigned long long *d_total, double *d_org2, double *d_dir2, uint32_t *d_path_of2,
                                  float *d_thr2, void *stream);
extern "C" int lh_pt_launch_resolve(int w, int h, int spp, float inv_total_spp, const float *d_radiance, float *d_rgb, void *stream);

static int pt_tile(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int s0, int spp, int spp_total, int max_vertices,
                   const lh_material_t *override_mat, const float env_rgb[3], const void *d_env_map, int env_w, int env_h, int flags,
                   uint64_t seed, void *d_rgb, lh_pt_stats_t *stats, void *stream)
{
    if (w <= 0 || h <= 0 || spp < 1 || spp_total < spp || max_vertices < 2) return fail("lh_render_pt_tile: bad arguments");
    HIPCHK(hipSetDevice(a->device));
    if (sync_materials(a) != 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    const size_t S = (size_t)w * h * spp;
    if (S >= ((size_t)1 << 31)) return fail("lh_render_pt_tile: more than 2^31 paths in one pass; lower spp_count or the tile size");
    unsigned long long *cnt = (a->stat_on && a->hs->bvh.ntris) ? a->d_counters : NULL;      /* lh_accel_trace_statistics */
    if (cnt) HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * LH_CNT_DEV, s));
    if (a->pt_fused && a->hs->bvh.ntris && a->dev.use_qnodes == 2) {
        /* the pass inside the walk: camera ray to last vertex per lane, only the radiance goes through HBM (lh_pt.h) */
        if (a->hs->device_built && !a->d_ref_nodes && sync_ref(a, false) != 0) return -1;
        if (ensure_buf(&a->p_rad, S * 12)) return -1;
        HIPCHK(hipMemsetAsync(a->d_total + 1, 0, 2 * sizeof(unsigned long long), s));
        const int grid = a->grid_forced_pt > 0 ? a->grid_forced_pt : a->ncus * 2;
        const int rcf = lh_launch_trace_pt(&a->dev, S, cam, x0, y0, w, spp, s0, cam->width, max_vertices, seed, (const double *)a->d_nrm9,
                                           (const double *)a->d_attr9[0], (const uint32_t *)a->d_prim_mesh, a->d_materials, override_mat, env_rgb,
                                           d_env_map, env_w, env_h, (flags & LH_PT_REFERENCE_WEIGHTS) != 0, (float *)a->p_rad.p, a->d_total + 1,
                                           (unsigned int *)(a->d_total + 2), cnt, a->d_cursor + (a->cursor_next++ % LH_NCURSOR), grid,
                                           a->min_active, a->tri_batch, (void *)s);
        if (rcf == 0) {
            unsigned long long hv[2] = {0, 0};
            if (lh_pt_launch_resolve(w, h, spp, 1.0f / (float)spp_total, (const float *)a->p_rad.p, (float *)d_rgb, s) != 0)
                return fail("pt resolve launch failed");
            HIPCHK(hipMemcpyAsync(hv, a->d_total + 1, sizeof(hv), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (cnt) {
                unsigned long long hc[LH_CNT_N];
                HIPCHK(hipMemcpy(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost));
                a->stat[0] += hc[LH_CNT_NODES]; a->stat[1] += hc[LH_CNT_TRIS]; a->stat[2] += hc[LH_CNT_EXACT]; a->stat[3] += hc[LH_CNT_RAYS];
            }
            if (stats) { stats->paths = S; stats->rays = hv[0]; stats->max_depth_reached = (uint64_t)(unsigned int)hv[1] + 1u; }
            return 0;
        }
        /* a tree the fused walk cannot take (deeper than the LDS rows and lucille's own tree not there yet): wavefront passes */
    }
    const unsigned nb = (unsigned)((S + 255) / 256);
    if (ensure_buf(&a->r_org, S * 24) || ensure_buf(&a->r_dir, S * 24) || ensure_buf(&a->p_org2, S * 24) ||
        ensure_buf(&a->p_dir2, S * 24) || ensure_buf(&a->r_prim, S * 4) || ensure_buf(&a->r_t, S * 8) ||
        ensure_buf(&a->r_u, S * 8) || ensure_buf(&a->r_v, S * 8) || ensure_buf(&a->p_path, S * 4) ||
        ensure_buf(&a->p_path2, S * 4) || ensure_buf(&a->p_thr, S * 12) || ensure_buf(&a->p_thr2, S * 12) ||
        ensure_buf(&a->p_rad, S * 12) || ensure_buf(&a->p_alive, S) || ensure_buf(&a->r_blocks, ((size_t)nb + nb / 1024 + 4) * 4)) return -1;
    HIPCHK(hipMemsetAsync(a->p_rad.p, 0, S * 12, s));
    double *org = (double *)a->r_org.p, *dir = (double *)a->r_dir.p, *org2 = (double *)a->p_org2.p, *dir2 = (double *)a->p_dir2.p;
    uint32_t *path = (uint32_t *)a->p_path.p, *path2 = (uint32_t *)a->p_path2.p;
    float *thr = (float *)a->p_thr.p, *thr2 = (float *)a->p_thr2.p;
    if (lh_pt_launch_primary(cam, x0, y0, w, h, spp, s0, seed, org, dir, path, thr, s) != 0) return fail("pt primary launch failed");
    size_t n = S; uint64_t rays = 0; int depth = 0;
    while (n > 0) {
        if (launch(a, n, org, dir, a->r_prim.p, a->r_t.p, a->r_u.p, a->r_v.p, NULL, LH_MODE_CLOSEST, LH_VARIANT_DEFAULT, cnt, s) != 0) return -1;
        rays += n;
        if (lh_pt_launch_shade(n, &a->dev, (const double *)a->d_nrm9, (const double *)a->d_attr9[0], (const uint32_t *)a->d_prim_mesh,
                               a->d_materials, override_mat, env_rgb, d_env_map, env_w, env_h, (flags & LH_PT_REFERENCE_WEIGHTS) != 0,
                               depth, max_vertices, seed, s0, spp, x0, y0, w, cam->width, org, dir, (const uint32_t *)a->r_prim.p,
                               (const double *)a->r_t.p, (const double *)a->r_u.p, (const double *)a->r_v.p, path, thr, (float *)a->p_rad.p,
                               (uint8_t *)a->p_alive.p, (uint32_t *)a->r_blocks.p, a->d_total, org2, dir2, path2, thr2, s) != 0)
            return fail("pt shade launch failed: %s", hipGetErrorString(hipGetLastError()));
        unsigned long long alive = 0;
        HIPCHK(hipMemcpyAsync(&alive, a->d_total, sizeof(alive), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        n = (size_t)alive; depth++;
        { double *t1 = org; org = org2; org2 = t1; t1 = dir; dir = dir2; dir2 = t1; }
        { uint32_t *t2 = path; path = path2; path2 = t2; float *t3 = thr; thr = thr2; thr2 = t3; }
    }
    if (lh_pt_launch_resolve(w, h, spp, 1.0f / (float)spp_total, (const float *)a->p_rad.p, (float *)d_rgb, s) != 0)
        return fail("pt resolve launch failed");
    HIPCHK(hipStreamSynchronize(s));
    if (cnt) {
        unsigned long long hc[LH_CNT_N];
        HIPCHK(hipMemcpy(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost));
        a->stat[0] += hc[LH_CNT_NODES]; a->stat[1] += hc[LH_CNT_TRIS]; a->stat[2] += hc[LH_CNT_EXACT]; a->stat[3] += hc[LH_CNT_RAYS];
    }
    if (stats) { stats->paths = S; stats->rays = rays; stats->max_depth_reached = (uint64_t)depth; }
    return 0;
}

/* round-1 entry point: one diffuse reflectance for every mesh, constant environment */
extern "C" int lh_render_pt_tile(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int s0, int spp,
                                 int spp_total, int max_vertices, float kd, const float env[3], uint64_t seed,
                                 void *d_rgb, lh_pt_stats_t *stats, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_pt_tile: accel not committed");
    if (!cam || !d_rgb || !env) return fail("lh_render_pt_tile: NULL argument");
    if (!(kd > 0.0f) || kd > 1.0f) return fail("lh_render_pt_tile: bad arguments");
    lh_material_t m; memset(&m, 0, sizeof(m)); m.kd[0] = m.kd[1] = m.kd[2] = kd; m.ior = 1.0f;
    return pt_tile(a, cam, x0, y0, w, h, s0, spp, spp_total, max_vertices, &m, env, NULL, 0, 0, 0, seed, d_rgb, stats, stream);
}

/* per-mesh materials (lh_accel_set_material) and the accelerator's environment (lh_accel_set_environment) */
extern "C" int lh_render_pt_tile2(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int s0, int spp,
                                  int spp_total, int max_vertices, int flags, uint64_t seed, void *d_rgb, lh_pt_stats_t *stats, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_pt_tile2: accel not committed");
    if (!cam || !d_rgb) return fail("lh_render_pt_tile2: NULL argument");
    float one[3] = {1.0f, 1.0f, 1.0f};
    const float *rgb = (a->env.rgb[0] != 0.0f || a->env.rgb[1] != 0.0f || a->env.rgb[2] != 0.0f || a->d_env_map) ? a->env.rgb : one;
    return pt_tile(a, cam, x0, y0, w, h, s0, spp, spp_total, max_vertices, NULL, rgb, a->d_env_map, a->env.width, a->env.height, flags,
                   seed, d_rgb, stats, stream);
}

/* ------------------------------------------------------------------------ */
/* whole frame into host memory (render_frame_controller + bucket_write)     */
/* ------------------------------------------------------------------------ */

extern "C" int lh_render_ao_frame_host(lh_accel_t *a, const lh_camera_t *cam, int ps, int gather_nsamples,
                                       uint64_t seed, int tile, float *rgb, lh_tile_stats_t *stats)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_ao_frame_host: accel not committed");
    if (!cam || !rgb) return fail("lh_render_ao_frame_host: NULL argument");
    if (cam->width <= 0 || cam->height <= 0) return fail("lh_render_ao_frame_host: bad resolution");
    if (tile <= 0) {
        /* default tile: the largest power of two (<= 4096) whose scratch stays under ~6 GB.  Per sub-sample: ~200 B of ray /
         * hit / epilogue records, plus 49 B per AO ray when the AO stage has to materialise its rays (LH_AO_FUSED=0):
         * ambient_occlusion.rib's own 3 x 3 pixel samples x 64 AO rays would otherwise ask for 30 GB per 1024^2 tile */
        const int N = gather_nsamples > 0 ? gather_nsamples : 1;
        const double per_pixel = (double)(ps > 0 ? ps : 1) * (ps > 0 ? ps : 1) * (200.0 + (a->ao_fused ? 0.0 : 49.0 * N));
        tile = 4096;
        while (tile > 64 && (double)tile * tile * per_pixel > 6.0e9) tile /= 2;
    }
    HIPCHK(hipSetDevice(a->device));
    const int W = cam->width, H = cam->height;
    if (ensure_buf(&a->r_frame, (size_t)tile * tile * 3 * sizeof(float))) return -1;
    std::vector<float> host((size_t)tile * tile * 3);
    lh_tile_stats_t tot = {0, 0, 0, 0};
    for (int y0 = 0; y0 < H; y0 += tile)
        for (int x0 = 0; x0 < W; x0 += tile) {
            const int w = (x0 + tile <= W) ? tile : W - x0, h = (y0 + tile <= H) ? tile : H - y0;
            lh_tile_stats_t st;
            if (lh_render_ao_tile(a, cam, x0, y0, w, h, ps, gather_nsamples, seed, NULL, a->r_frame.p, &st, a->stream) != 0) return -1;
            HIPCHK(hipMemcpyAsync(host.data(), a->r_frame.p, (size_t)w * h * 3 * sizeof(float), hipMemcpyDeviceToHost, a->stream));
            HIPCHK(hipStreamSynchronize(a->stream));
            /* the tile comes back with its rows already flipped (row 0 = pixel row y0+h-1) */
            for (int r = 0; r < h; r++)
                memcpy(rgb + ((size_t)(H - (y0 + h) + r) * W + x0) * 3, host.data() + (size_t)r * w * 3, (size_t)w * 3 * sizeof(float));
            tot.primary_rays += st.primary_rays; tot.primary_hits += st.primary_hits;
            tot.ao_rays += st.ao_rays; tot.ao_occluded += st.ao_occluded;
        }
    if (stats) *stats = tot;
    return 0;
}
