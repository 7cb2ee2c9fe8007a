/*
 * lh_commit.hip -- lifetime of an accelerator behind the C ABI (include/lucille_hip.h): staging of the meshes, the host or
 * device build, the device replica of the scene, parameters.  Reference: accel_build_func / accel_free_func
 * (src/render/accel.h:24-28), ri_bvh_build (src/render/bvh.c:276-379), ri_bvh_free (bvh.c:381-387).
 */
#include <thread>
#include <vector>

#include "lh_internal.h"

static thread_local char g_err[512] = "";

extern "C" void lh_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

int lh_fail(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return -1;
}

pthread_mutex_t g_scene_mu = PTHREAD_MUTEX_INITIALIZER;

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
    a->combine = 1;
    { const char *e = getenv("LH_COMBINE"); if (e) a->combine = atoi(e) != 0; }
    a->host_walk = 1;
    { const char *e = getenv("LH_HOST_WALK"); if (e) a->host_walk = atoi(e) != 0; }
    a->ao_fused = 1;
    a->wide8 = -1;
    { const char *e = getenv("LH_WIDE8"); if (e) a->wide8 = atoi(e); }
    { const char *e = getenv("LH_AO_FUSED"); if (e) a->ao_fused = atoi(e) != 0; }
    { const char *e = getenv("LH_FAST_START"); if (e) a->fast_start = atoi(e) != 0; }
    { const char *e = getenv("LH_POISON_OUTPUTS"); a->poison_outputs = e && atoi(e) != 0; }
    const char *env;
    a->min_active = 24;         /* the tile pipelines' regroup threshold (tools/experiments/knob_sweep6.py, r05: config 4 108.9 -> 107.4 ms, config 5 47.95 -> 47.67 ms against 32); ray dumps: LH_DUMP*_MIN_ACTIVE */
    a->tri_batch = 12;          /* parked leaves a triangle pass waits for (tools/experiments/knob_sweep2.py, r03: S-soup-1M 2119 -> 2142 Mrays/s, config-5 AO frame 87.0 -> 85.8 ms against 8) */
    env = getenv("LH_TRI_BATCH");
    if (env && atoi(env) > 0 && atoi(env) <= 64) { a->tri_batch = atoi(env); a->knobs_user = 1; }
    env = getenv("LH_MIN_ACTIVE");
    if (env && atoi(env) > 0 && atoi(env) <= 64) { a->min_active = atoi(env); a->knobs_user = 1; }
    a->ray_chunk = 256;
    env = getenv("LH_RAY_CHUNK");
    if (env && atoi(env) > 0 && atoi(env) <= (1 << 20)) a->ray_chunk = (uint32_t)atoi(env);
    a->dev.ray_chunk = a->ray_chunk;
    a->dev.ray_budget = LH_RAY_BUDGET; a->dump_budget = LH_DUMP_BUDGET;
    a->ao_budget = LH_AO_BUDGET;
    env = getenv("LH_AO_BUDGET");
    if (env && atoi(env) > 0) { a->ao_budget = (uint32_t)atoi(env); a->ao_budget_user = 1; }
    env = getenv("LH_DUMP_BUDGET");
    if (env && atoi(env) > 0) a->dump_budget = (uint32_t)atoi(env);
    env = getenv("LH_RAY_BUDGET");
    if (env && atoi(env) > 0) a->dev.ray_budget = (uint32_t)atoi(env);
    a->dev.top_nodes = LH_TOP_AUTO;
    a->dev.deg_dcap = INFINITY; a->dev.cap_srcs = 0u; a->dev.ndanger = LH_DANGER_ALL;
    a->dev.coop_patience = 0;
    a->dev.ao_group = 0;            /* measured: the grouped order is SLOWER on the config-5 frame (84.6 -> 95.1 ms, tools/ao_group_probe.py): a slot's own rays share their first levels */
    env = getenv("LH_AO_GROUP");
    if (env && atoi(env) >= 0 && atoi(env) <= 4096) a->dev.ao_group = (uint32_t)atoi(env);
    env = getenv("LH_TOP_NODES");
    if (env && atoi(env) >= 0 && atoi(env) <= (int)LH_TOP_NODES_MAX) a->dev.top_nodes = (uint32_t)atoi(env);
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

void lh_free_buf(lh_buf *b) { if (b->p) (void)hipFree(b->p); b->p = NULL; b->cap = 0; }
#define free_buf lh_free_buf

int lh_ensure_buf(lh_buf *b, size_t bytes)
{
    if (b->cap >= bytes && b->p) return 0;
    if (b->p) { (void)hipFree(b->p); b->p = NULL; b->cap = 0; }
    if (bytes == 0) bytes = 16;
    HIPCHK(hipMalloc(&b->p, bytes));
    b->cap = bytes;
    return 0;
}

static void release_device(lh_accel_t *a)
{
    lh_buf *bufs[] = {&a->r_org, &a->r_dir, &a->r_prim, &a->r_t, &a->r_u, &a->r_v, &a->r_slot, &a->r_hitrec,
                      &a->r_aorg, &a->r_adir, &a->r_occ, &a->r_blocks, &a->r_key, &a->r_frame, &a->r_occcount,
                      &a->p_org2, &a->p_dir2, &a->p_path, &a->p_path2, &a->p_thr, &a->p_thr2, &a->p_rad, &a->p_counts};
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
    free_buf(&a->r_state); free_buf(&a->r_uni); free_buf(&a->r_bands); free_buf(&a->r_diag);
    if (a->h_read) { (void)hipHostFree(a->h_read); a->h_read = NULL; }
    a->d_total = NULL; a->d_nrm9 = NULL;
    if (a->d_nodes) (void)hipFree(a->d_nodes);
    if (a->d_tri32) (void)hipFree(a->d_tri32);
    if (a->d_tri64) (void)hipFree(a->d_tri64);
    if (a->d_q4nodes) (void)hipFree(a->d_q4nodes);
    if (a->d_q8nodes) (void)hipFree(a->d_q8nodes);
    a->d_q4nodes = a->d_q8nodes = NULL; a->dev.q4nodes = NULL; a->dev.q8nodes = NULL; a->dev.nodes = NULL;
    if (a->d_ref_lca) (void)hipFree(a->d_ref_lca);
    if (a->d_prim_leafpos) (void)hipFree(a->d_prim_leafpos);
    if (a->d_ref_nodes) (void)hipFree(a->d_ref_nodes);
    if (a->d_ref_leaf_prims) (void)hipFree(a->d_ref_leaf_prims);
    a->d_ref_lca = a->d_prim_leafpos = a->d_ref_nodes = a->d_ref_leaf_prims = NULL;
    if (a->d_danger) (void)hipFree(a->d_danger);
    a->d_danger = NULL; a->dev.ndanger = LH_DANGER_ALL;
    if (a->d_cursor) (void)hipFree(a->d_cursor);
    if (a->d_counters) (void)hipFree(a->d_counters);
    for (int k = 0; k < LH_AOQ_SLOTS; k++) {
        lh_fixq_t *q = &a->aoq[k].q;
        if (q->queue) {
            (void)hipFree(q->queue); (void)hipFree(q->qcount);
            (void)hipStreamDestroy((hipStream_t)q->aux_stream); (void)hipEventDestroy((hipEvent_t)q->ev_ready); (void)hipEventDestroy((hipEvent_t)q->ev_done);
        }
        memset(q, 0, sizeof(*q)); a->aoq[k].used = 0;
    }
    if (a->pipe.ready) {
        for (int b = 0; b < a->pipe.depth; b++) {
            (void)hipHostFree(a->pipe.h_in[b]); (void)hipHostFree(a->pipe.h_out[b]);
            (void)hipFree(a->pipe.d_in[b]); (void)hipFree(a->pipe.d_out[b]);
            (void)hipEventDestroy(a->pipe.in_done[b]); (void)hipEventDestroy(a->pipe.done[b]);
        }
        for (int b = 0; b < 3; b++) (void)hipStreamDestroy(a->pipe.s[b]);
        a->pipe.ready = 0;
    }
    if (a->d_stage) (void)hipFree(a->d_stage);
    if (a->stream) (void)hipStreamDestroy(a->stream);
    a->d_nodes = a->d_tri32 = a->d_tri64 = NULL; a->d_cursor = a->d_counters = NULL;
    a->d_stage = NULL; a->stage_bytes = 0; a->stream = NULL;
}

static void free_trash(lh_host_scene *hs)
{
    if (!hs->trash) return;
    for (uint32_t k = 0; k < hs->ntrash; k++) free(hs->trash[k]);
    free(hs->trash); hs->trash = NULL; hs->ntrash = 0;
}

static void *ref_thread_main(void *arg)
{
    lh_host_scene *hs = (lh_host_scene *)arg;
    const double t0 = now_s();
    free_trash(hs);
    if (hs->ref_on_device) { __atomic_store_n(&hs->ref_state, 0, __ATOMIC_RELEASE); return NULL; }       /* only the mesh copies to return */
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
    const double tf = now_s();
    int rc = on_device ? lh_bvh_flatten(&hs->bvh, views, a->nmeshes, build_threads)
                       : lh_bvh_build_hook(&hs->bvh, views, a->nmeshes, build_threads, hs->have_ref ? start_ref_thread : NULL, hs);
    free(views);
    if (on_device && getenv("LH_BUILD_TIMING")) fprintf(stderr, "[lucille_hip] commit: host flatten                 %8.2f ms\n", (now_s() - tf) * 1e3);
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
            /* not in front of the first frame: a background thread builds it, launch() attaches it when it is ready.  It is
             * started by lh_accel_commit AFTER the device has its scene (start_ref_background): its first phase streams the
             * same 1.5 GB the upload reads and would take a third of the host's memory bandwidth from it */
            hs->ref_threads = build_threads; hs->ref_state = 1;
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
    return 0;
}

extern "C" int lh_device_build(uint32_t ntris, const double *d_tri64, void **d_q4nodes, uint32_t *nq4, uint32_t *q4_depth, uint32_t *q4_stack,
                               int want_q8, void **d_q8nodes, uint32_t *nq8, uint32_t *q8_depth,
                               void **d_tri32, float bmin[3], float bmax[3], float grid_lo[3], float grid_step[3],
                               uint32_t *nlive, double *deg_dcap, void *stream, char *err, size_t errlen);                /* lh_build.hip */

/* flag = 1 if the two word arrays differ anywhere */
__global__ void k_words_differ(size_t nwords, const uint32_t *__restrict__ x, const uint32_t *__restrict__ y, int *__restrict__ flag)
{
    bool diff = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) diff |= x[i] != y[i];
    if (__ballot(diff) != 0ull && (threadIdx.x & 63) == 0) atomicExch(flag, 1);
}

/* the 8-wide nodes of a scene whose tree was built on this device without them: the build is run once more (it is
 * deterministic: stable sort, stable partitions, a layout of the top that does not depend on its threads) and only its 8-wide
 * collapse is kept.  The leaves of those nodes index the resident triangle records, so the second run's records must be the
 * resident ones word for word -- checked, not assumed */
static int device_rebuild_q8(lh_accel_t *a)
{
    lh_host_scene *hs = a->hs;
    char berr[256] = "";
    void *q4 = NULL, *q8 = NULL, *t32 = NULL; int *flag = NULL, h_flag = 0;
    uint32_t nq4 = 0, d4 = 0, st4 = 0, nq8 = 0, d8 = 0; float bmin[3], bmax[3], glo[3], gst[3];
    const int rcb = lh_device_build(hs->bvh.ntris, (const double *)a->d_tri64, &q4, &nq4, &d4, &st4, 1, &q8, &nq8, &d8, &t32, bmin, bmax, glo, gst,
                                    NULL, NULL, (void *)a->stream, berr, sizeof(berr));
    if (rcb != 0) return fail("building the 8-wide nodes on the device failed: %s", berr);
    int rc = 0;
    if (q8) {
        const size_t nwords = sizeof(lh_tri32_t) / 4 * (size_t)hs->bvh.ntris;
        if (hipMalloc((void **)&flag, sizeof(int)) != hipSuccess || hipMemsetAsync(flag, 0, sizeof(int), a->stream) != hipSuccess) rc = fail("out of device memory");
        if (rc == 0) {
            hipLaunchKernelGGL(k_words_differ, dim3(2048), dim3(256), 0, a->stream, nwords, (const uint32_t *)t32, (const uint32_t *)a->d_tri32, flag);
            if (hipMemcpyAsync(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, a->stream) != hipSuccess || hipStreamSynchronize(a->stream) != hipSuccess)
                rc = fail("comparing the rebuilt triangle order failed: %s", hipGetErrorString(hipGetLastError()));
        }
        if (rc == 0 && (h_flag || nq4 != hs->bvh.nq4nodes))
            rc = fail("the device build did not repeat itself (%u nodes, then %u): the 8-wide nodes are not usable; set \"wide8\" = 1 before the commit", hs->bvh.nq4nodes, nq4);
    }
    if (flag) (void)hipFree(flag);
    if (q4) (void)hipFree(q4);
    if (t32) (void)hipFree(t32);
    if (rc != 0) { if (q8) (void)hipFree(q8); return -1; }
    if (q8) { a->d_q8nodes = q8; a->dev.q8nodes = q8; a->dev.nq8nodes = nq8; a->dev.q8_depth = d8; a->device_bytes += sizeof(lh_q8node_t) * (size_t)nq8; }
    return 0;
}

extern "C" int lh_device_ref_build(uint32_t ntris, const double *d_tri64, void **d_nodes, uint32_t *nnodes, uint32_t *max_depth,
                                   void **d_leaf_prims, void **d_lca, void **d_leafpos, double scene6[6], void *stream, char *err, size_t errlen);   /* lh_refbuild.hip */

/* lucille's own tree of a device-built scene, on this device (LH_REF_BUILD=host: the background host thread instead).
 * 0: attached; 1: not built here (the host thread will); -1: error */
/* ---- the leaves of lucille's own tree that hold a zero-area triangle (round 6; lh_bvh.h ndanger / danger, lh_walk.h danger_hit) ----
 * A numerically collinear triangle that stays in the traversal tree caps the directions the tree can vouch for at 1 / s2 (lh_bvh.c
 * tri_zero_area_s2) -- below 1 as soon as |e1|_1 |e2|_1 > 1, i.e. for ANY such sliver in a scene modelled in small units -- and every
 * ray beyond the cap used to take the single-lane reference walk: correct, and a hundred times slower.  The reference
 * can "hit" such a triangle only on rays that reach its leaf, so once lucille's own tree is on the device this scan lists the
 * boxes of those leaves (the child box the leaf's parent holds: what test_ray_node tests, bvh.c:938-1083); rays that miss them all
 * walk the traversal tree like any other.  More than LH_DANGER_MAX of them: every ray, as before. */
__global__ __launch_bounds__(256) void k_danger_scan(uint32_t n, const double *__restrict__ tri64, const uint2 *__restrict__ leafpos, const int4 *__restrict__ lca,
                                                     const lh_refnode_t *__restrict__ nodes, double s0, double s1, double s2_, double s3, double s4, double s5,
                                                     unsigned long long *__restrict__ count, double *__restrict__ boxes)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const double *t = tri64 + 9 * (size_t)p;
    /* tri_dead_class (lh_bvh.c): out of the traversal tree, never reported below LH_DEG_DCAP_ALL */
    if ((t[0] == t[3] && t[1] == t[4] && t[2] == t[5]) || (t[0] == t[6] && t[1] == t[7] && t[2] == t[8])) return;
    if (t[3] == t[6] && t[4] == t[7] && t[5] == t[8]) {
        const double sN = fabs(t[3] - t[0]) + fabs(t[4] - t[1]) + fabs(t[5] - t[2]);
        if (sN * sN * (1.0 + 1e-9) <= 1.0e-14 / (1.0e-15 * LH_DEG_DCAP_ALL)) return;
    }
    if (!(lh_zero_area_weight(t, t + 3, t + 6) > 0.0)) return;          /* lh_bvh.h: the triangles lh_bvh.c tri_zero_area_s2 / lh_build.hip k_prim_boxes bound deg_dcap with */
    const unsigned long long slot = atomicAdd(count, 1ull);
    if (slot >= LH_DANGER_MAX) return;
    const int leaf = (int)leafpos[p].x, parent = lca[leaf].x;
    double *o = boxes + 6 * slot;
    if (parent < 0) { o[0] = s0; o[1] = s1; o[2] = s2_; o[3] = s3; o[4] = s4; o[5] = s5; return; }       /* the root is the leaf: the scene box (bvh.c:325-340) */
    const int k = (lca[parent].w == leaf) ? 0 : 1;          /* children are allocated adjacently: child[1] = child[0] + 1 */
    for (int q = 0; q < 6; q++) o[q] = nodes[parent].box[k][q];
}

static int lh_danger_scan(lh_accel_t *a)
{
    lh_host_scene *hs = a->hs;
    a->dev.ndanger = LH_DANGER_ALL;
    if (!a->d_ref_nodes || !a->d_tri64 || hs->bvh.ntris == 0) return 0;
    if (!(hs->bvh.deg_dcap < LH_DEG_DCAP_ALL)) return 0;              /* no zero-area triangle in the tree caps the directions: nothing to list */
    if (getenv("LH_DANGER_BOXES") && atoi(getenv("LH_DANGER_BOXES")) == 0) return 0;       /* A/B: round 5's rule (every ray beyond the cap) */
    HIPCHK(hipSetDevice(a->device));
    if (!a->d_danger) HIPCHK(hipMalloc(&a->d_danger, sizeof(double) * (8 + 6 * LH_DANGER_MAX)));
    HIPCHK(hipMemsetAsync(a->d_danger, 0, sizeof(double) * (8 + 6 * LH_DANGER_MAX), a->stream));
    const uint32_t n = hs->bvh.ntris;
    hipLaunchKernelGGL(k_danger_scan, dim3((n + 255u) / 256u), dim3(256), 0, a->stream, n, (const double *)a->d_tri64, (const uint2 *)a->d_prim_leafpos,
                       (const int4 *)a->d_ref_lca, (const lh_refnode_t *)a->d_ref_nodes, a->dev.ref_bmin[0], a->dev.ref_bmin[1], a->dev.ref_bmin[2],
                       a->dev.ref_bmax[0], a->dev.ref_bmax[1], a->dev.ref_bmax[2], (unsigned long long *)a->d_danger, (double *)a->d_danger + 8);
    HIPCHK(hipGetLastError());
    double h[8 + 6 * LH_DANGER_MAX];
    HIPCHK(hipMemcpyAsync(h, a->d_danger, sizeof(h), hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    unsigned long long cnt; memcpy(&cnt, h, sizeof(cnt));
    if (cnt == 0 || cnt > LH_DANGER_MAX) return 0;                      /* none found (the cap came from somewhere else: keep round 5's rule) or too many */
    pthread_mutex_lock(&g_scene_mu);
    if (hs->bvh.ndanger == LH_DANGER_ALL) {            /* the first replica of a shared host scene writes the list; the others found the same one */
        memcpy(hs->bvh.danger, h + 8, sizeof(double) * 6 * (size_t)cnt);
        __atomic_store_n(&hs->bvh.ndanger, (uint32_t)cnt, __ATOMIC_RELEASE);      /* the one-ray host walk reads it (lh_hostwalk.c) */
    }
    pthread_mutex_unlock(&g_scene_mu);
    {   /* the union of the listed boxes on the scene's 16-bit grid, a cell wider on every side (the nodes' own boxes are rounded outward
         * the same way: lh_bvh.c).  A union that leaves the grid (a leaf that also holds a triangle outside the traversal tree's bounds):
         * round 5's rule */
        double u[6] = {1.0e308, 1.0e308, 1.0e308, -1.0e308, -1.0e308, -1.0e308};
        for (unsigned long long i = 0; i < cnt; i++)
            for (int k = 0; k < 3; k++) { u[k] = fmin(u[k], h[8 + 6 * i + k]); u[3 + k] = fmax(u[3 + k], h[8 + 6 * i + 3 + k]); }
        uint32_t w[3];
        for (int k = 0; k < 3; k++) {
            const double g0 = (double)hs->bvh.grid_lo[k], st = (double)hs->bvh.grid_step[k];          /* the host scene's: a->dev's copy is set later in a device-built commit */
            const double qlo = floor((u[k] - g0) / st) - 1.0, qhi = ceil((u[3 + k] - g0) / st) + 1.0;
            if (!(st > 0.0) || !(qlo >= -2.0) || !(qhi <= 65537.0)) return 0;          /* (also a NaN) */
            const uint32_t lo = qlo < 0.0 ? 0u : (uint32_t)qlo, hi = qhi > 65535.0 ? 65535u : (uint32_t)qhi;
            w[k] = lo | hi << 16;
        }
        for (int k = 0; k < 3; k++) a->dev.danger[k] = w[k];
    }
    a->dev.ndanger = (uint32_t)cnt;
    return 0;
}

static int device_ref_tree(lh_accel_t *a)
{
    lh_host_scene *hs = a->hs;
    const char *e = getenv("LH_REF_BUILD");
    if (e && strcmp(e, "host") == 0 && !hs->ref_on_device) return 1;
    char rerr[256] = ""; uint32_t rn = 0, rdepth = 0; double sc6[6];
    const double t0 = now_s();
    if (lh_device_ref_build(hs->bvh.ntris, (const double *)a->d_tri64, &a->d_ref_nodes, &rn, &rdepth, &a->d_ref_leaf_prims, &a->d_ref_lca,
                            &a->d_prim_leafpos, sc6, (void *)a->stream, rerr, sizeof(rerr)) != 0) {
        a->d_ref_nodes = a->d_ref_leaf_prims = a->d_ref_lca = a->d_prim_leafpos = NULL;
        if (hs->ref_on_device) return fail("building lucille's own tree on the device failed: %s", rerr);       /* a replica: there is no host copy to fall back on */
        if (getenv("LH_BUILD_TIMING")) fprintf(stderr, "[lucille_hip] commit: lucille's own tree not built on the device (%s): host thread\n", rerr);
        return 1;
    }
    a->dev.ref_lca = a->d_ref_lca; a->dev.prim_leafpos = a->d_prim_leafpos;
    a->dev.ref_nodes = a->d_ref_nodes; a->dev.ref_leaf_prims = a->d_ref_leaf_prims;
    a->dev.ref_nnodes = rn; a->dev.ref_empty = 0;
    for (int k = 0; k < 3; k++) { a->dev.ref_bmin[k] = sc6[k]; a->dev.ref_bmax[k] = sc6[3 + k]; }
    a->device_bytes += sizeof(int) * 4 * (size_t)rn + sizeof(uint32_t) * 3 * (size_t)hs->bvh.ntris + sizeof(lh_refnode_t) * (size_t)rn;
    pthread_mutex_lock(&g_scene_mu);
    hs->ref_on_device = 1; hs->ref_build_seconds = now_s() - t0;
    pthread_mutex_unlock(&g_scene_mu);
    return lh_danger_scan(a);
}

int lh_ensure_formats(lh_accel_t *a, int mask)
{
    const lh_bvh_t *b = &a->hs->bvh;
    if ((mask & LH_FMT_Q8) && !a->d_q8nodes && a->hs->device_built && !a->hs->received && b->nq4nodes > 1 && device_rebuild_q8(a) != 0) return -1;
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
    if (a->hs->device_built) return 0;         /* the tree exists only as 4-wide nodes on the device */
    if ((mask & LH_FMT_F32) && !a->d_nodes) {
        const size_t nb = sizeof(lh_node_t) * (size_t)b->nnodes;
        HIPCHK(hipMalloc(&a->d_nodes, nb));
        HIPCHK(hipMemcpy(a->d_nodes, b->nodes, nb, hipMemcpyHostToDevice));
        a->dev.nodes = a->d_nodes; a->device_bytes += nb;
    }
    if ((mask & LH_FMT_Q16X4) && !a->d_q4nodes) {
        const size_t q4b = sizeof(lh_q4node_t) * (size_t)b->nq4nodes;
        HIPCHK(hipMalloc(&a->d_q4nodes, q4b));
        HIPCHK(hipMemcpy(a->d_q4nodes, b->q4nodes, q4b, hipMemcpyHostToDevice));
        a->dev.q4nodes = a->d_q4nodes; a->device_bytes += q4b;
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
    return lh_danger_scan(a);
}

/* the background build of the reference-order tree: attach it if it has finished (wait: block until it has) */
int lh_sync_ref(lh_accel_t *a, bool wait)
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

/* persistent workgroups per launch: as many as the LDS stack rows of this scene let a CU hold -- at most 4: the walk's 128 VGPRs */
static int size_grid(lh_accel_t *a)
{
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, a->device));
    const uint32_t stack = (uint32_t)lh_trace_rows(&a->dev);
    int per_cu = (int)(160u / stack);                     /* LDS: stack KiB per 256-thread workgroup */
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    a->grid_blocks = prop.multiProcessorCount * per_cu; a->ncus = prop.multiProcessorCount;
    const char *env = getenv("LH_GRID_BLOCKS");
    if (env && atoi(env) > 0) { a->grid_blocks = atoi(env); a->grid_user = 1; }
    return 0;
}

/* ---- device replica of the host scene (once per GPU) ---------------------------------------- */
static int device_upload(lh_accel_t *a)
{
    lh_host_scene *hs = a->hs;
    HIPCHK(hipSetDevice(a->device));
    double t0 = now_s();
    HIPCHK(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc((void **)&a->d_cursor, sizeof(uint32_t) * LH_CURSOR_WORDS * LH_NCURSOR));
    HIPCHK(hipMalloc((void **)&a->d_counters, sizeof(unsigned long long) * LH_CNT_DEV));
    HIPCHK(hipMalloc((void **)&a->d_total, sizeof(unsigned long long) * 72));      /* the hit count of a batch; then k_ao_resolve's 64 occlusion counters; [64]: the hit count kept for the fused AO stage and the batch's one read-back */
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
        const double tu = now_s();
        HIPCHK(hipMalloc(&a->d_tri64, t64));
        HIPCHK(hipMemcpy(a->d_tri64, hs->bvh.tri64, t64, hipMemcpyHostToDevice));
        if (getenv("LH_BUILD_TIMING")) fprintf(stderr, "[lucille_hip] commit: tri64 upload (%.0f MB)      %8.2f ms\n", t64 / 1e6, (now_s() - tu) * 1e3);
        if (hs->device_built) {
            /* the traversal tree is built here, on this device (lh_build.hip): LBVH -> the same 4-wide nodes */
            char berr[256] = "";
            uint32_t nq4 = 0, d4 = 0, st4 = 0, nq8 = 0, d8 = 0; float bmin[3], bmax[3], glo[3], gst[3];
            /* "wide8" = 1: the 8-wide nodes of ray dumps are collapsed from the same binary tree now.  Left to itself (-1) a scene
             * gets them when its first dump asks (lh_ensure_formats: the build is run again, ~40 ms per 10 M triangles) -- a renderer
             * that re-commits every frame and never dumps does not pay for them */
            const int want_q8 = a->wide8 == 1;
            const double tb = now_s();
            uint32_t nlive = hs->bvh.ntris; double dcap = INFINITY;
            const int rcb = lh_device_build(hs->bvh.ntris, (const double *)a->d_tri64, &a->d_q4nodes, &nq4, &d4, &st4, want_q8, &a->d_q8nodes, &nq8, &d8, &a->d_tri32, bmin, bmax, glo, gst,
                                            &nlive, &dcap, (void *)a->stream, berr, sizeof(berr));
            if (rcb == -2) return fail("lh_accel_commit: a vertex coordinate is NaN, infinite or beyond 1e30");
            if (rcb != 0 && a->build_auto && !hs->ref_on_device) {
                if (getenv("LH_BUILD_TIMING")) fprintf(stderr, "[lucille_hip] commit: device build failed (%s): host builders\n", berr);
                return -3;                              /* nobody asked for the device: the caller builds on the host */
            }
            if (rcb != 0) return fail("device BVH build failed: %s", berr);
            pthread_mutex_lock(&g_scene_mu);
            hs->bvh.nq4nodes = nq4; hs->bvh.q4_depth = d4; hs->bvh.q4_stack = st4; hs->bvh.nnodes = nq4; hs->bvh.max_depth = d4; hs->bvh.build_seconds = now_s() - tb;
            hs->bvh.nlive = nlive; hs->bvh.deg_dcap = dcap;
            for (int k = 0; k < 3; k++) { hs->bvh.bmin[k] = bmin[k]; hs->bvh.bmax[k] = bmax[k]; hs->bvh.grid_lo[k] = glo[k]; hs->bvh.grid_step[k] = gst[k]; }
            pthread_mutex_unlock(&g_scene_mu);
            a->dev.q4nodes = a->d_q4nodes;
            a->device_bytes += sizeof(lh_q4node_t) * (size_t)nq4;
            if (a->d_q8nodes) { a->dev.q8nodes = a->d_q8nodes; a->dev.nq8nodes = nq8; a->dev.q8_depth = d8; a->device_bytes += sizeof(lh_q8node_t) * (size_t)nq8; }
            if (3 * d4 + 5 > 264) return -3;          /* deeper than k_overflow_fix's private stack: the caller falls back to the host builder */
            if (hs->have_ref && (hs->ref_on_device || hs->ref_state == 1) && device_ref_tree(a) < 0) return -1;
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
        a->dev.deg_dcap = hs->bvh.deg_dcap < 3.0e38 ? (float)hs->bvh.deg_dcap : INFINITY;
        a->dev.cap_srcs = (a->dev.deg_dcap < 3.0e38f ? 1u : 0u) | (a->dev.deg_dcap < 1.0f ? 6u : 0u);
        for (int k = 0; k < 3; k++) { a->dev.grid_lo[k] = hs->bvh.grid_lo[k]; a->dev.grid_step[k] = hs->bvh.grid_step[k]; }
        if (hs->have_ref && __atomic_load_n(&hs->ref_state, __ATOMIC_ACQUIRE) == 2 && attach_ref(a) != 0) return -1;
        a->dev.nq4nodes = hs->bvh.nq4nodes; a->dev.q4_depth = hs->bvh.q4_depth; a->dev.q4_stack = hs->bvh.q4_stack;
        /* resident from the start: what the default kernel reads (everything else on first use) */
        if (lh_ensure_formats(a, LH_FMT_Q16X4) != 0) return -1;
    }
    a->upload_seconds = now_s() - t0;
    if (size_grid(a) != 0) return -1;
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
    /* where the trees are built: said by the caller (LH_BUILD_ON_DEVICE, LH_BUILD_ON_HOST, or a thread count = the host), by
     * LH_BUILD in the environment, or -- build_threads == 0 -- by the size of the scene: from LH_AUTO_DEVICE_TRIANGLES on the
     * device builders (0.3 s instead of 4 s for 21 M triangles, frames ~2 % slower, HISTORY.md 15), below it the host builders
     * (milliseconds either way, and the better tree).  An automatic device build that fails falls back to the host. */
    bool on_device = build_threads == LH_BUILD_ON_DEVICE;
    bool chosen = build_threads == LH_BUILD_ON_DEVICE || build_threads == LH_BUILD_ON_HOST || build_threads > 0;
    { const char *e = getenv("LH_BUILD"); if (e && strcmp(e, "device") == 0) { on_device = true; chosen = true; } if (e && strcmp(e, "host") == 0) { on_device = false; chosen = true; } }
    if (!chosen) {
        unsigned long long ntri = 0;
        for (uint32_t g = 0; g < a->nmeshes; g++) ntri += a->meshes[g].nidx / 3;
        unsigned long long at = LH_AUTO_DEVICE_TRIANGLES;
        { const char *e = getenv("LH_AUTO_DEVICE_TRIANGLES"); if (e && atoll(e) > 0) at = (unsigned long long)atoll(e); }
        on_device = ntri >= at;
    }
    a->build_auto = !chosen;
    if (build_threads < 0) build_threads = 0;
    if (on_device) {
        const bool timing = getenv("LH_BUILD_TIMING") != NULL;
        const double tc0 = now_s();
        if (host_build(a, build_threads, true, true) != 0) return -1;
        const double tc1 = now_s();
        const int rc = device_upload(a);
        if (timing) fprintf(stderr, "[lucille_hip] commit: host side %.2f ms, device side %.2f ms\n", (tc1 - tc0) * 1e3, (now_s() - tc1) * 1e3);
        {
            lh_host_scene *hs = a->hs;
            if (hs->ref_state == 1 && !hs->ref_thread_live && rc != 0) hs->ref_state = 0;      /* the device side failed (or the tree is too deep: -3): no background
                                                                                                  * build of lucille's tree for a scene that is about to be rebuilt on the host, or dropped */
            if (hs->ref_state == 1 && !hs->ref_thread_live) {
                if (rc == 0 && a->nmeshes) {
                    /* the mesh copies are no longer needed: the background thread returns them to the system */
                    hs->trash = (void **)calloc((size_t)a->nmeshes * 8, sizeof(void *));
                    if (hs->trash) {
                        for (uint32_t g = 0; g < a->nmeshes; g++) {
                            hs->trash[hs->ntrash++] = a->meshes[g].pos; hs->trash[hs->ntrash++] = a->meshes[g].idx; hs->trash[hs->ntrash++] = a->meshes[g].nrm;
                            for (int k = 0; k < 5; k++) hs->trash[hs->ntrash++] = a->meshes[g].attr[k];
                            a->meshes[g].pos = NULL; a->meshes[g].idx = NULL; a->meshes[g].nrm = NULL;
                            for (int k = 0; k < 5; k++) a->meshes[g].attr[k] = NULL;
                        }
                    }
                }
                if (pthread_create(&hs->ref_thread, NULL, ref_thread_main, hs) != 0) { hs->ref_state = 0; free_trash(hs); return fail("lh_accel_commit: cannot start the reference-tree thread"); }
                hs->ref_thread_live = 1;
            }
        }
        if (rc == -3) {
            /* an LBVH deeper than the kernel's stack bound (degenerate distributions): build on the host after all */
            (void)hipSetDevice(a->device); release_device(a);
            lh_host_scene *hs = a->hs;
            pthread_mutex_lock(&g_scene_mu);
            if (hs->ref_thread_live) { pthread_join(hs->ref_thread, NULL); hs->ref_thread_live = 0; }
            pthread_mutex_unlock(&g_scene_mu);
            free(hs->nrm9); free(hs->attr9[0]); free(hs->attr9[1]); free(hs->attr9[2]); free(hs->st6); free(hs->inside);
            hs->nrm9 = NULL; hs->attr9[0] = hs->attr9[1] = hs->attr9[2] = NULL; hs->st6 = NULL; hs->inside = NULL;
            lh_bvh_release(&hs->bvh); lh_refbvh_release(&hs->ref); hs->ref_state = 0; hs->ref_on_device = 0;
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
    return lh_sync_ref(a, true);
}

/* lh_multi.hip: `dst` (created, nothing added) becomes a replica of `src`'s committed scene on its own device */
extern "C" int lh_accel_commit_replica(lh_accel_t *dst, lh_accel_t *src)
{
    if (!dst || !src || !src->committed) return fail("lh_accel_commit_replica: source not committed");
    if (src->hs->received) return fail("lh_accel_commit_replica: the source scene was received from another rank (no host copy to replicate)");
    lh_guard guard(dst);
    if (dst->committed || dst->commit_failed || dst->nmeshes) return fail("lh_accel_commit_replica: destination is not a fresh accelerator");
    pthread_mutex_lock(&g_scene_mu);
    lh_host_scene *old = dst->hs;
    dst->hs = src->hs; dst->hs->refs++;
    pthread_mutex_unlock(&g_scene_mu);
    free(old);                                  /* a fresh accelerator's scene holds nothing */
    dst->commit_failed = 1;
    {
        const int rc = device_upload(dst);              /* a device-built scene is built again on this replica's device */
        if (rc == -3) return fail("lh_accel_commit_replica: the device-built tree is deeper than the traversal kernels' stacks (the source fell back to the host builder?)");
        if (rc != 0) return -1;
    }
    dst->commit_failed = 0;
    return 0;
}

extern "C" int lh_accel_ref_tree(lh_accel_t *a, uint32_t *nnodes, void *nodes_out, uint32_t *leaf_prims_out)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_accel_ref_tree: accel not committed");
    if (lh_sync_ref(a, true) != 0) return -1;
    if (!a->d_ref_nodes) return fail("lh_accel_ref_tree: the reference-order tree is not attached (empty scene or LH_REFTREE=0)");
    HIPCHK(hipSetDevice(a->device));
    if (nnodes) *nnodes = a->dev.ref_nnodes;
    if (nodes_out) HIPCHK(hipMemcpy(nodes_out, a->d_ref_nodes, sizeof(lh_refnode_t) * (size_t)a->dev.ref_nnodes, hipMemcpyDeviceToHost));
    if (leaf_prims_out) HIPCHK(hipMemcpy(leaf_prims_out, a->d_ref_leaf_prims, sizeof(uint32_t) * (size_t)a->hs->bvh.ntris, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void lh_accel_destroy(lh_accel_t *a)
{
    if (!a) return;
    if (a->committed || a->commit_failed) { (void)hipSetDevice(a->device); lh_comb_destroy(a); release_device(a); }
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
        free_trash(a->hs);
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
    o->ntriangles_in_tree = a->hs->bvh.ntris ? a->hs->bvh.nlive : 0u;
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

/* tuning knobs of the traversal kernel (sweeps, tools/): "grid" persistent workgroups, "min_active" regroup threshold,
 * "tri_batch" parked leaves per triangle pass, "ray_chunk" rays per cursor atomic, "variant" default kernel variant,
 * "ao_fused", "wide8" (-1 auto / 0 / 1), "stack_cap" (tests of the overflow path),
 * "coop_patience" (10 ns ticks the pass beside a launch waits for progress; 0: half a second) */
extern "C" int lh_accel_set_param(lh_accel_t *a, const char *name, int value)
{
    lh_guard guard(a);
    if (!a || !name) return fail("lh_accel_set_param: NULL argument");
    if (!strcmp(name, "grid") && value > 0) { a->grid_blocks = value; a->grid_user = 1; }
    else if (!strcmp(name, "min_active") && value > 0 && value <= 64) { a->min_active = value; a->knobs_user = 1; }
    else if (!strcmp(name, "tri_batch") && value > 0 && value <= 64) { a->tri_batch = value; a->knobs_user = 1; }
    else if (!strcmp(name, "ray_chunk") && value > 0 && value <= (1 << 20)) { a->ray_chunk = (uint32_t)value; a->dev.ray_chunk = (uint32_t)value; }
    else if (!strcmp(name, "variant") && (value == LH_VARIANT_DIRECT || value == LH_VARIANT_SPEC)) a->default_variant = value;
    else if (!strcmp(name, "ray_budget") && value > 0) { a->dev.ray_budget = (uint32_t)value; a->dump_budget = (uint32_t)value; a->ao_budget = (uint32_t)value; a->ao_budget_user = 1; }
    else if (!strcmp(name, "dump_budget") && value > 0) a->dump_budget = (uint32_t)value;
    else if (!strcmp(name, "ao_budget") && value >= 0) { a->ao_budget = (uint32_t)value; a->ao_budget_user = 1; }
    else if (!strcmp(name, "coop_patience") && value >= 0) a->dev.coop_patience = (uint32_t)value;
    else if (!strcmp(name, "ao_fused")) a->ao_fused = value != 0;
    else if (!strcmp(name, "combine")) a->combine = value != 0;
    else if (!strcmp(name, "host_walk")) { a->host_walk = value != 0; a->hw_gpu_left = 0; a->hw_ns = 0.0; }
    else if (!strcmp(name, "ao_group") && value >= 0 && value <= 4096) a->dev.ao_group = (uint32_t)value;
    else if (!strcmp(name, "fast_start")) a->fast_start = value != 0;
    else if (!strcmp(name, "wide8") && value >= -1 && value <= 1) a->wide8 = value;
    else if (!strcmp(name, "top_nodes") && value >= -1 && value <= (int)LH_TOP_NODES_MAX) a->dev.top_nodes = value < 0 ? LH_TOP_AUTO : (uint32_t)value;
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

/* ------------------------------------------------------------------------ */
/* the scene image: what one rank's commit hands the other ranks (lh_dist.hip) */
/* ------------------------------------------------------------------------ */
/* ONE host build, then a broadcast of the flattened arrays into every GPU's HBM.  The image is a header
 * (lh_scene_image_t), the device arrays in a fixed order (lh_scene_image_arrays) and two host arrays (primitive ->
 * mesh ordinal / index, for lh_accel_prim_lookup and the materials).  A receiver has no host tree and no host
 * triangles: it cannot spawn replicas or upload other node formats, everything else works. */
int lh_scene_image_header(lh_accel_t *a, lh_scene_image_t *h)
{
    if (!a || !a->committed) return fail("scene image: accel not committed");
    if (lh_sync_ref(a, true) != 0) return -1;           /* a device-built scene: lucille's own tree must be attached */
    /* a scene whose ray dumps walk the 8-wide nodes (hot set beyond the Infinity Cache) is sent WITH them: rank 0 builds them
     * lazily on its first dump, a receiver has no host tree to build them from and would walk the slower 4-wide nodes --
     * and the slowest rank sets the time of a sharded dump */
    if (lh_accel_dump_node_bytes(a) == (int)sizeof(lh_q8node_t) && !a->d_q8nodes && lh_ensure_formats(a, LH_FMT_Q8) != 0) return -1;
    const lh_host_scene *hs = a->hs;
    memset(h, 0, sizeof(*h));
    h->magic = 0x4C48494Du;
    h->ntris = hs->bvh.ntris; h->nnodes = hs->bvh.nnodes; h->max_depth = hs->bvh.max_depth; h->nleaves = hs->bvh.nleaves;
    h->nq4 = hs->bvh.nq4nodes; h->q4_depth = hs->bvh.q4_depth; h->q4_stack = hs->bvh.q4_stack;
    h->nq8 = a->d_q8nodes ? a->dev.nq8nodes : 0; h->q8_depth = a->d_q8nodes ? a->dev.q8_depth : 0;
    h->nmeshes = hs->nmeshes;
    h->have_ref = a->d_ref_nodes != NULL; h->ref_nnodes = a->dev.ref_nnodes; h->ref_empty = a->dev.ref_empty;
    h->has_nrm = a->d_nrm9 != NULL; h->has_st = a->d_st6 != NULL; h->has_inside = a->d_inside != NULL;
    for (int k = 0; k < 3; k++) {
        h->has_attr[k] = a->d_attr9[k] != NULL;
        h->bmin[k] = hs->bvh.bmin[k]; h->bmax[k] = hs->bvh.bmax[k]; h->grid_lo[k] = hs->bvh.grid_lo[k]; h->grid_step[k] = hs->bvh.grid_step[k];
        h->ref_bmin[k] = a->dev.ref_bmin[k]; h->ref_bmax[k] = a->dev.ref_bmax[k];
    }
    h->build_seconds = hs->bvh.build_seconds; h->ref_build_seconds = hs->ref_build_seconds;
    h->nlive = hs->bvh.nlive; h->deg_dcap = hs->bvh.deg_dcap;
    return 0;
}

/* the device arrays of the image, same order on the sender and on a receiver that ran lh_scene_image_alloc */
int lh_scene_image_arrays(lh_accel_t *a, const lh_scene_image_t *h, void **ptr, size_t *bytes, int cap)
{
    int n = 0;
    const size_t nt = h->ntris;
#define LH_IMG(P, B) do { if (n < cap) { ptr[n] = (P); bytes[n] = (B); } n++; } while (0)
    if (nt) {
        LH_IMG(a->d_tri32, sizeof(lh_tri32_t) * nt + 64); LH_IMG(a->d_tri64, sizeof(lh_tri64_t) * nt);
        LH_IMG(a->d_q4nodes, sizeof(lh_q4node_t) * (size_t)h->nq4);
        if (h->nq8) LH_IMG(a->d_q8nodes, sizeof(lh_q8node_t) * (size_t)h->nq8);
        if (h->have_ref) {
            LH_IMG(a->d_ref_lca, sizeof(int) * 4 * (size_t)h->ref_nnodes); LH_IMG(a->d_prim_leafpos, sizeof(uint32_t) * 2 * nt);
            LH_IMG(a->d_ref_nodes, sizeof(lh_refnode_t) * (size_t)h->ref_nnodes); LH_IMG(a->d_ref_leaf_prims, sizeof(uint32_t) * nt);
        }
        if (h->has_nrm) LH_IMG(a->d_nrm9, sizeof(double) * 9 * nt);
        for (int k = 0; k < 3; k++) if (h->has_attr[k]) LH_IMG(a->d_attr9[k], sizeof(double) * 9 * nt);
        if (h->has_st) LH_IMG(a->d_st6, sizeof(double) * 6 * nt);
        if (h->has_inside) LH_IMG(a->d_inside, nt);
    }
#undef LH_IMG
    return n;
}

/* host side of the image: prim -> mesh ordinal, prim -> 3 * i (ntris uint32 each) */
uint32_t *lh_scene_image_prim_geom(lh_accel_t *a) { return a->hs->bvh.prim_geom; }
uint32_t *lh_scene_image_prim_index(lh_accel_t *a) { return a->hs->bvh.prim_index; }

/* a fresh accelerator becomes the receiving end: device arrays allocated as the header says */
int lh_scene_image_alloc(lh_accel_t *a, const lh_scene_image_t *h)
{
    if (!a || a->committed || a->commit_failed || a->nmeshes) return fail("scene image: the receiver must be a fresh accelerator");
    if (h->magic != 0x4C48494Du) return fail("scene image: bad header");
    lh_guard guard(a);
    lh_host_scene *hs = a->hs;
    a->commit_failed = 1;
    HIPCHK(hipSetDevice(a->device));
    HIPCHK(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc((void **)&a->d_cursor, sizeof(uint32_t) * LH_CURSOR_WORDS * LH_NCURSOR));
    HIPCHK(hipMalloc((void **)&a->d_counters, sizeof(unsigned long long) * LH_CNT_DEV));
    HIPCHK(hipMalloc((void **)&a->d_total, sizeof(unsigned long long) * 72));      /* the hit count of a batch; then k_ao_resolve's 64 occlusion counters; [64]: the hit count kept for the fused AO stage and the batch's one read-back */
    hs->received = 1; hs->device_built = 1;              /* no host tree: the walks over other node formats are not available */
    hs->bvh.ntris = h->ntris; hs->bvh.nnodes = h->nnodes; hs->bvh.max_depth = h->max_depth; hs->bvh.nleaves = h->nleaves;
    hs->bvh.nq4nodes = h->nq4; hs->bvh.q4_depth = h->q4_depth; hs->bvh.q4_stack = h->q4_stack; hs->nmeshes = h->nmeshes;
    hs->bvh.build_seconds = h->build_seconds; hs->ref_build_seconds = h->ref_build_seconds;
    hs->bvh.nlive = h->nlive; hs->bvh.deg_dcap = h->deg_dcap;
    a->dev.deg_dcap = h->deg_dcap < 3.0e38 ? (float)h->deg_dcap : INFINITY;
    a->dev.cap_srcs = (a->dev.deg_dcap < 3.0e38f ? 1u : 0u) | (a->dev.deg_dcap < 1.0f ? 6u : 0u);
    hs->have_ref = h->have_ref; hs->ref_state = h->have_ref ? 2 : 0;
    float r = 0.0f;
    for (int k = 0; k < 3; k++) {
        hs->bvh.bmin[k] = h->bmin[k]; hs->bvh.bmax[k] = h->bmax[k]; hs->bvh.grid_lo[k] = h->grid_lo[k]; hs->bvh.grid_step[k] = h->grid_step[k];
        a->dev.grid_lo[k] = h->grid_lo[k]; a->dev.grid_step[k] = h->grid_step[k];
        a->dev.ref_bmin[k] = h->ref_bmin[k]; a->dev.ref_bmax[k] = h->ref_bmax[k];
        r = fmaxf(r, fabsf(h->bmin[k])); r = fmaxf(r, fabsf(h->bmax[k]));
    }
    const size_t nt = h->ntris;
    a->device_bytes = 0;
    if (nt) {
        hs->bvh.prim_geom = (uint32_t *)malloc(sizeof(uint32_t) * nt); hs->bvh.prim_index = (uint32_t *)malloc(sizeof(uint32_t) * nt);
        if (!hs->bvh.prim_geom || !hs->bvh.prim_index) return fail("out of memory");
        HIPCHK(hipMalloc(&a->d_tri32, sizeof(lh_tri32_t) * nt + 64)); HIPCHK(hipMalloc(&a->d_tri64, sizeof(lh_tri64_t) * nt));
        HIPCHK(hipMalloc(&a->d_q4nodes, sizeof(lh_q4node_t) * (size_t)h->nq4));
        if (h->nq8) HIPCHK(hipMalloc(&a->d_q8nodes, sizeof(lh_q8node_t) * (size_t)h->nq8));
        if (h->have_ref) {
            HIPCHK(hipMalloc(&a->d_ref_lca, sizeof(int) * 4 * (size_t)h->ref_nnodes)); HIPCHK(hipMalloc(&a->d_prim_leafpos, sizeof(uint32_t) * 2 * nt));
            HIPCHK(hipMalloc(&a->d_ref_nodes, sizeof(lh_refnode_t) * (size_t)h->ref_nnodes)); HIPCHK(hipMalloc(&a->d_ref_leaf_prims, sizeof(uint32_t) * nt));
        }
        if (h->has_nrm) HIPCHK(hipMalloc(&a->d_nrm9, sizeof(double) * 9 * nt));
        for (int k = 0; k < 3; k++) if (h->has_attr[k]) HIPCHK(hipMalloc(&a->d_attr9[k], sizeof(double) * 9 * nt));
        if (h->has_st) HIPCHK(hipMalloc(&a->d_st6, sizeof(double) * 6 * nt));
        if (h->has_inside) HIPCHK(hipMalloc(&a->d_inside, nt));
        a->dev.tri32 = a->d_tri32; a->dev.tri64 = a->d_tri64; a->dev.q4nodes = a->d_q4nodes;
        a->dev.q8nodes = a->d_q8nodes; a->dev.nq8nodes = h->nq8; a->dev.q8_depth = h->q8_depth;
        a->dev.ntris = h->ntris; a->dev.nnodes = h->nnodes; a->dev.max_depth = h->max_depth; a->dev.scene_r = r; a->dev.ray_chunk = a->ray_chunk;
        a->dev.nq4nodes = h->nq4; a->dev.q4_depth = h->q4_depth; a->dev.q4_stack = h->q4_stack;
        if (h->have_ref) {
            a->dev.ref_lca = a->d_ref_lca; a->dev.prim_leafpos = a->d_prim_leafpos; a->dev.ref_nodes = a->d_ref_nodes;
            a->dev.ref_leaf_prims = a->d_ref_leaf_prims; a->dev.ref_nnodes = h->ref_nnodes; a->dev.ref_empty = h->ref_empty;
        }
        void *ptr[32]; size_t bytes[32];
        const int n = lh_scene_image_arrays(a, h, ptr, bytes, 32);
        for (int k = 0; k < n; k++) a->device_bytes += bytes[k];
    }
    return 0;
}

/* ... and, once the arrays have arrived, a committed replica */
int lh_scene_image_finish(lh_accel_t *a)
{
    if (!a || !a->hs->received || a->committed) return fail("scene image: not a receiving accelerator");
    HIPCHK(hipSetDevice(a->device));
    if (size_grid(a) != 0) return -1;
    if (lh_danger_scan(a) != 0) return -1;          /* the arrays have arrived: this replica lists its own boxes */
    a->upload_seconds = 0.0;
    a->committed = 1; a->commit_failed = 0;
    return 0;
}


/* fill miss results without touching the scene (empty accel) */
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

