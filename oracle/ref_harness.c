/*
 * ref_harness.c -- thin ctypes-friendly driver around the COMPILED REFERENCE.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is this repository's own code; it is
 * compiled by oracle/Makefile together with the reference's sources *where
 * they lie* under /root/reference (never copied) into
 * oracle/_ref/liblucille_ref.so.  It includes the reference's public headers
 * at build time and calls the reference's own entry points:
 *
 *   ri_parallel_init / ri_render_init / ri_render_get   (base/parallel.h, render/render.h:104-105)
 *   ri_geom_new / ri_geom_add_positions / _add_indices  (render/geom.h)
 *   ri_scene_new / ri_scene_add_geom / ri_scene_build_accel (render/scene.h:60-96)
 *   ri_accel_bind(RI_ACCEL_BVH)                          (render/accel.h)
 *   ri_raytrace                                          (render/raytrace.h:46-49)
 *
 * It replaces exactly one reference object, render/accel.o, with the
 * ri_accel_new/free/bind below so that the accelerator's intersect() can be
 * wrapped by a recorder (the (org,dir)->(hit) stream the reference's own AO
 * transport produces is what tests/golden/ao_c1_* holds).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "ri.h"
#include "parallel.h"
#include "log.h"
#include "memory.h"
#include "render.h"
#include "scene.h"
#include "geom.h"
#include "accel.h"
#include "bvh.h"
#include "ugrid.h"
#include "raytrace.h"
#include "ray.h"
#include "intersection_state.h"
#include "beam.h"
#include "raster.h"
#include "context.h"
#include "option.h"
#include "camera.h"
#include "display.h"
#include "hash.h"
#include "list.h"

#define LREF_MISS 0xFFFFFFFFu

#ifdef LREF_WITH_HIP
extern int ri_accel_bind_hip(ri_accel_t *accel);   /* integration/ri_accel_hip.c */
extern int ri_hipbvh_intersect_beam(void *accel, ri_beam_t *beam, ri_raster_plane_t *raster_out, void *user);
extern int ri_hipbvh_intersect_beam_visibility(void *accel, ri_beam_t *beam, void *user);
extern int ri_hipbvh_intersect_beam_visibility_n(void *accel, size_t n, ri_beam_t *beams, int *result);
#endif

/* ---------------------------------------------------------------------- */
/* replacement for render/accel.c (ri_accel_new/free/bind, accel.c:28-109) */
/* with a recording wrapper around ri_bvh_intersect                        */
/* ---------------------------------------------------------------------- */

typedef struct {
    double   org[3], dir[3];
    double   t, u, v;
    uint32_t hit, geom, index, pad;
} lref_record_t;   /* 88 bytes */

static lref_record_t *g_rec      = NULL;
static size_t         g_rec_n    = 0, g_rec_cap = 0;
static int            g_rec_on   = 0;

#define LREF_MAX_GEOMS 4096
static ri_geom_t *g_geoms[LREF_MAX_GEOMS];
static uint32_t   g_geom_base[LREF_MAX_GEOMS + 1];
static uint32_t   g_ngeoms = 0;

static uint32_t geom_ordinal(const ri_geom_t *g)
{
    uint32_t i;
    for (i = 0; i < g_ngeoms; i++) if (g_geoms[i] == g) return i;
    return LREF_MISS;
}

/* camera of the frame being rendered, captured at the first query (it is set up
 * after the accelerator is built, render.c:335-336) */
static int g_cam_ortho;
static double g_cam[16 + 4];   /* camera_to_world row-major, flength, w, h, is_rh */
static int    g_cam_valid = 0;
static accel_intersect_func g_inner_intersect = NULL;

static void capture_camera(void)
{
    ri_camera_t *c = ri_render_get()->context->option->camera;
    int i, j;
    for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) g_cam[4 * i + j] = c->camera_to_world.f[i][j];
    g_cam[16] = c->flength; g_cam[17] = c->horizontal_resolution; g_cam[18] = c->vertical_resolution;
    g_cam[19] = c->is_rh;
    g_cam_ortho = (c->camera_projection == RI_ORTHOGRAPHIC);
    g_cam_valid = 1;
}

static int recording_intersect(void *accel, ri_ray_t *ray,
                               ri_intersection_state_t *state, void *user)
{
    double o[3], d[3];
    int hit;
    if (!g_cam_valid) capture_camera();
    o[0] = ray->org[0]; o[1] = ray->org[1]; o[2] = ray->org[2];
    d[0] = ray->dir[0]; d[1] = ray->dir[1]; d[2] = ray->dir[2];
    hit = g_inner_intersect(accel, ray, state, user);
    if (g_rec_on) {
        lref_record_t *r;
        if (g_rec_n == g_rec_cap) {
            g_rec_cap = g_rec_cap ? g_rec_cap * 2 : (1u << 16);
            g_rec = (lref_record_t *)realloc(g_rec, g_rec_cap * sizeof(*g_rec));
        }
        r = &g_rec[g_rec_n++];
        memcpy(r->org, o, sizeof(o)); memcpy(r->dir, d, sizeof(d));
        r->hit = (uint32_t)hit; r->pad = 0;
        if (hit) { r->t = state->t; r->u = state->u; r->v = state->v;
                   r->geom = geom_ordinal(state->geom); r->index = state->index; }
        else     { r->t = 1.0e38; r->u = r->v = 0.0; r->geom = LREF_MISS; r->index = 0; }
    }
    return hit;
}

/* geometry of the scene the renderer hands to build(): captured so fixtures can
 * hold the triangles the reference actually traced (after its own RIB ingest,
 * polygon.c:494-1001) */
static void capture_scene(const ri_scene_t *scene)
{
    ri_list_t *itr;
    g_ngeoms = 0; g_geom_base[0] = 0;
    for (itr = ri_list_first((ri_list_t *)scene->geom_list); itr != NULL; itr = ri_list_next(itr)) {
        ri_geom_t *g = (ri_geom_t *)itr->data;
        if (g_ngeoms >= LREF_MAX_GEOMS) break;
        g_geoms[g_ngeoms] = g;
        g_geom_base[g_ngeoms + 1] = g_geom_base[g_ngeoms] + g->nindices / 3;
        g_ngeoms++;
    }
}

static accel_build_func g_inner_build = NULL;
static void *capturing_build(const void *data)
{
    capture_scene((const ri_scene_t *)data);
    g_cam_valid = 0;
    return g_inner_build(data);
}

uint32_t lref_scene_ngeoms(void) { return g_ngeoms; }
void lref_scene_geom_sizes(uint32_t g, uint32_t *npos, uint32_t *nidx, uint32_t *has_normals, uint32_t *two_side)
{
    *npos = g_geoms[g]->npositions; *nidx = g_geoms[g]->nindices;
    *has_normals = g_geoms[g]->normals != NULL; *two_side = (uint32_t)g_geoms[g]->two_side;
}
void lref_scene_geom_copy(uint32_t g, double *pos_xyz, uint32_t *idx, double *nrm_xyz)
{
    uint32_t i; ri_geom_t *G = g_geoms[g];
    for (i = 0; i < G->npositions; i++) { pos_xyz[3*i] = G->positions[i][0]; pos_xyz[3*i+1] = G->positions[i][1]; pos_xyz[3*i+2] = G->positions[i][2]; }
    memcpy(idx, G->indices, sizeof(uint32_t) * G->nindices);
    if (nrm_xyz && G->normals)
        for (i = 0; i < G->nnormals; i++) { nrm_xyz[3*i] = G->normals[i][0]; nrm_xyz[3*i+1] = G->normals[i][1]; nrm_xyz[3*i+2] = G->normals[i][2]; }
}
int lref_camera_get(double out[20]) { memcpy(out, g_cam, sizeof(g_cam)); return g_cam_valid; }
int lref_camera_is_ortho(void) { return g_cam_ortho; }

ri_accel_t *ri_accel_new()
{
    ri_accel_t *p = (ri_accel_t *)ri_mem_alloc(sizeof(ri_accel_t));
    memset(p, 0, sizeof(ri_accel_t));
    return p;
}

void ri_accel_free(ri_accel_t *accel)
{
    if (accel) {
        if (accel->free) accel->free(accel->data);
        ri_mem_free(accel);
    }
}

int ri_accel_bind(ri_accel_t *accel, int method)
{
    switch (method) {
    case RI_ACCEL_UGRID:
        accel->build = ri_ugrid_build; accel->free = ri_ugrid_free;
        accel->intersect = ri_ugrid_intersect;
        break;
    case RI_ACCEL_BVH:
        accel->build = ri_bvh_build; accel->free = ri_bvh_free;
        accel->intersect = ri_bvh_intersect;
        break;
#ifdef LREF_WITH_HIP
    case 2:   /* RI_ACCEL_HIP: the glue of integration/ri_accel_hip.c */
        if (ri_accel_bind_hip(accel) != 0) return -1;
        break;
#endif
    default:
        return -1;
    }
    /* wrap whatever was bound with the scene/camera capture and the ray recorder */
    g_inner_build = accel->build;         accel->build = capturing_build;
    g_inner_intersect = accel->intersect; accel->intersect = recording_intersect;
    return 0;
}

/* ---------------------------------------------------------------------- */
/* direct-scene driver (pattern of src/testbed/main.cpp:53-65)             */
/* ---------------------------------------------------------------------- */

static int g_inited = 0;

int lref_init(void)
{
    if (!g_inited) {
        int argc = 1; char *argv0 = (char *)"lref"; char **argv = &argv0;
        ri_parallel_init(&argc, &argv);
        ri_render_init();
        ri_log_set_level(RI_LOG_LEVEL_ERROR);
        g_inited = 1;
    }
    return 0;
}

/* start a fresh scene on the global renderer (old one is leaked on purpose:
 * ri_bvh_free only frees the header, bvh.c:381-387) */
int lref_scene_reset(void)
{
    lref_init();
    ri_render_get()->scene = ri_scene_new();
    g_ngeoms = 0; g_geom_base[0] = 0;
    return 0;
}

int lref_scene_add_mesh(uint32_t npos, const double *pos_xyz, uint32_t nidx,
                        const uint32_t *idx)
{
    ri_geom_t *g; ri_vector_t *P; uint32_t i;
    if (g_ngeoms >= LREF_MAX_GEOMS) return -1;
    g = ri_geom_new();
    P = (ri_vector_t *)malloc(sizeof(ri_vector_t) * (npos ? npos : 1));
    for (i = 0; i < npos; i++) {
        P[i][0] = pos_xyz[3 * i + 0]; P[i][1] = pos_xyz[3 * i + 1];
        P[i][2] = pos_xyz[3 * i + 2]; P[i][3] = 0.0;
    }
    if (npos) ri_geom_add_positions(g, npos, (const ri_vector_t *)P);
    if (nidx) ri_geom_add_indices(g, nidx, idx);
    free(P);
    ri_scene_add_geom(ri_render_get()->scene, g);
    g_geoms[g_ngeoms] = g;
    g_geom_base[g_ngeoms + 1] = g_geom_base[g_ngeoms] + nidx / 3;
    g_ngeoms++;
    return 0;
}

/* optional attributes of geom `mesh`, through the reference's own ri_geom_add_* (geom.c:102-290).
 * kind -1: normals (+ two_side); 0 colors, 1 tangents, 2 binormals (xyz per vertex); 3 texcoords (st per vertex);
 * 4 texcoords_unshared (st per index). */
int lref_scene_set_attribute(uint32_t mesh, int kind, const double *data, uint32_t count, int two_side)
{
    ri_geom_t *g; ri_vector_t *V; uint32_t i;
    if (mesh >= g_ngeoms) return -1;
    g = g_geoms[mesh];
    if (kind == -1) g->two_side = two_side;
    if (!data) return 0;
    if (kind >= 3) {
        if (kind == 3) ri_geom_add_texcoords(g, count, data);
        else ri_geom_add_texcoords_unshared(g, count, data);
        return 0;
    }
    V = (ri_vector_t *)malloc(sizeof(ri_vector_t) * (count ? count : 1));
    for (i = 0; i < count; i++) { V[i][0] = data[3 * i]; V[i][1] = data[3 * i + 1]; V[i][2] = data[3 * i + 2]; V[i][3] = 0.0; }
    if (kind == -1) ri_geom_add_normals(g, count, (const ri_vector_t *)V);
    else if (kind == 0) ri_geom_add_colors(g, count, (const ri_vector_t *)V);
    else if (kind == 1) ri_geom_add_tangents(g, count, (const ri_vector_t *)V);
    else if (kind == 2) ri_geom_add_binormals(g, count, (const ri_vector_t *)V);
    free(V);
    return 0;
}

/* ri_raytrace + the whole ri_intersection_state_t of every hit: state24 = P Ng Ns tangent binormal color(3)
 * st(2) I inside; zeros for a miss */
void lref_state_batch(size_t n, const double *org, const double *dir, uint32_t *prim, double *state24)
{
    size_t i; ri_render_t *render = ri_render_get();
    for (i = 0; i < n; i++) {
        ri_ray_t ray; ri_intersection_state_t st; int hit, k; double *o = state24 + 24 * i;
        memset(&ray, 0, sizeof(ray)); memset(&st, 0, sizeof(st));
        for (k = 0; k < 3; k++) { ray.org[k] = org[3 * i + k]; ray.dir[k] = dir[3 * i + k]; }
        ray.thread_num = 0;
        hit = ri_raytrace(render, &ray, &st);
        memset(o, 0, 24 * sizeof(double));
        if (!hit) { prim[i] = LREF_MISS; continue; }
        prim[i] = g_geom_base[geom_ordinal(st.geom)] + st.index / 3;
        for (k = 0; k < 3; k++) {
            o[k] = st.P[k]; o[3 + k] = st.Ng[k]; o[6 + k] = st.Ns[k]; o[9 + k] = st.tangent[k]; o[12 + k] = st.binormal[k];
            o[15 + k] = st.color[k]; o[20 + k] = st.I[k];
        }
        o[18] = st.stqr[0]; o[19] = st.stqr[1]; o[23] = (double)st.inside;
    }
}

int lref_scene_build(void)
{
    ri_scene_t *scene = ri_render_get()->scene;
    if (ri_accel_bind(scene->accel, RI_ACCEL_BVH) != 0) return -1;
    return ri_scene_build_accel(scene);
}

/* closest hit through the reference's own façade ri_raytrace().
 * state9 (optional): P, Ng, Ns of the hit (9 doubles per ray) as produced by
 * ri_intersection_state_build. */
void lref_intersect_batch(size_t n, const double *org, const double *dir,
                          uint32_t *prim, double *t, double *u, double *v,
                          double *state9)
{
    size_t i; ri_render_t *render = ri_render_get();
    for (i = 0; i < n; i++) {
        ri_ray_t ray; ri_intersection_state_t st; int hit;
        memset(&ray, 0, sizeof(ray)); memset(&st, 0, sizeof(st));
        ray.org[0] = org[3 * i]; ray.org[1] = org[3 * i + 1]; ray.org[2] = org[3 * i + 2];
        ray.dir[0] = dir[3 * i]; ray.dir[1] = dir[3 * i + 1]; ray.dir[2] = dir[3 * i + 2];
        ray.thread_num = 0;
        hit = ri_raytrace(render, &ray, &st);
        if (hit) {
            uint32_t g = geom_ordinal(st.geom);
            prim[i] = g_geom_base[g] + st.index / 3;
            t[i] = st.t; u[i] = st.u; v[i] = st.v;
            if (state9) {
                int k;
                for (k = 0; k < 3; k++) {
                    state9[9 * i + k] = st.P[k]; state9[9 * i + 3 + k] = st.Ng[k];
                    state9[9 * i + 6 + k] = st.Ns[k];
                }
            }
        } else {
            prim[i] = LREF_MISS; t[i] = 1.0e38; u[i] = 0.0; v[i] = 0.0;
            if (state9) memset(&state9[9 * i], 0, 9 * sizeof(double));
        }
    }
}

/* tree shape by walking the reference's own node structs (bvh.h:73-94) */
static void walk(const ri_qbvh_node_t *n, uint64_t depth, uint64_t out[5])
{
    if (depth > out[2]) out[2] = depth;
    if (n->is_leaf) {
        uint32_t cnt = *((const uint32_t *)&n->bbox[0]);
        out[1]++; if (cnt > out[3]) out[3] = cnt; out[4] += cnt;
    } else {
        out[0]++;
        walk(n->child[0], depth + 1, out); walk(n->child[1], depth + 1, out);
    }
}

/* out = {ninner, nleaf, max_depth, max_leaf_tris, ntriangles} */
void lref_tree_stats(uint64_t out[5])
{
    ri_bvh_t *bvh = (ri_bvh_t *)ri_render_get()->scene->accel->data;
    memset(out, 0, 5 * sizeof(uint64_t));
    if (!bvh || bvh->empty) return;
    walk(bvh->root, 0, out);
}

void lref_scene_bbox(double bmin[3], double bmax[3])
{
    ri_bvh_t *bvh = (ri_bvh_t *)ri_render_get()->scene->accel->data;
    int k; for (k = 0; k < 3; k++) { bmin[k] = bvh->bmin[k]; bmax[k] = bvh->bmax[k]; }
}

/* traversal counters: meaningful only in the -DRI_BVH_TRACE_STATISTICS build
 * (liblucille_ref_stat.so); g_stattrav is the reference's global (bvh.c:146) */
#ifdef RI_BVH_TRACE_STATISTICS
extern ri_bvh_stat_traversal_t g_stattrav;
void lref_counters_clear(void) { memset(&g_stattrav, 0, sizeof(g_stattrav)); }
void lref_counters_get(uint64_t out[5])
{
    out[0] = g_stattrav.ninner_node_traversals; out[1] = g_stattrav.nleaf_node_traversals;
    out[2] = g_stattrav.ntested_triangles; out[3] = g_stattrav.nactually_hit_triangles;
    out[4] = g_stattrav.nrays;
}
int lref_has_counters(void) { return 1; }
#else
void lref_counters_clear(void) {}
void lref_counters_get(uint64_t out[5]) { memset(out, 0, 5 * sizeof(uint64_t)); }
int lref_has_counters(void) { return 0; }
#endif

/* ---------------------------------------------------------------------- */
/* frame capture: replaces the functions of the "file" display driver      */
/* (render.c:259-268, driver interface src/ri/display.h:72-81) so a        */
/* Display "x.hdr" "file" "rgb" render lands in memory as float RGB        */
/* ---------------------------------------------------------------------- */
static float *g_img = NULL; static int g_img_w = 0, g_img_h = 0;
static int cap_open(const char *name, int w, int h, int bits, RtToken comp, const char *fmt)
{
    (void)name; (void)bits; (void)comp; (void)fmt;
    free(g_img); g_img = (float *)calloc((size_t)w * h * 3, sizeof(float)); g_img_w = w; g_img_h = h;
    return 1;
}
static int cap_write(int x, int y, const void *pixel)
{
    const float *p = (const float *)pixel; size_t i;
    if (x < 0 || y < 0 || x >= g_img_w || y >= g_img_h) return 0;
    i = 3 * ((size_t)x + (size_t)y * g_img_w);
    g_img[i] += p[0] < 0 ? 0 : p[0]; g_img[i + 1] += p[1] < 0 ? 0 : p[1]; g_img[i + 2] += p[2] < 0 ? 0 : p[2];  /* hdrdrv.c:84-96 */
    return 1;
}
static int cap_close(void) { return 1; }
static int cap_progress(void) { return 1; }

int lref_capture_display(void)
{
    ri_display_drv_t *drv;
    lref_init();
    drv = (ri_display_drv_t *)ri_hash_lookup(ri_render_get()->display_drvs, RI_FILE);
    if (!drv) return -1;
    drv->open = cap_open; drv->write = cap_write; drv->close = cap_close; drv->progress = cap_progress;
    return 0;
}
int lref_image_size(int *w, int *h) { *w = g_img_w; *h = g_img_h; return g_img != NULL; }
void lref_image_copy(float *dst) { memcpy(dst, g_img, sizeof(float) * 3 * (size_t)g_img_w * g_img_h); }

/* option overrides the lsh CLI would apply (lsh/main.c:213-241) */
void lref_set_options(int accel_method, int nthreads, int gather_nsamples)
{
    ri_option_t *o = ri_render_get()->context->option;
    if (accel_method >= 0) o->accel_method = accel_method;
    if (nthreads > 0) o->nthreads = nthreads;
    if (gather_nsamples > 0) o->gather_nsamples = gather_nsamples;
}

/* beam visibility through the reference's own ri_beam_set + ri_bvh_intersect_beam_visibility
 * (beam.c:331-465, bvh.c:612-667); dirs: n x 4 x 3; -1 where ri_beam_set refuses */
void lref_beam_visibility_batch(size_t n, const double *org, const double *dirs, int32_t *result)
{
    size_t r; void *accel = ri_render_get()->scene->accel->data;
    FILE *saved = stderr;
    for (r = 0; r < n; r++) {
        ri_beam_t beam; ri_vector_t o, d[4]; int i, k;
        memset(&beam, 0, sizeof(beam));
        for (k = 0; k < 3; k++) o[k] = org[3 * r + k];
        o[3] = 0.0;
        for (i = 0; i < 4; i++) { for (k = 0; k < 3; k++) d[i][k] = dirs[12 * r + 3 * i + k]; d[i][3] = 0.0; }
        stderr = fopen("/dev/null", "w");          /* ri_beam_set prints a TODO message on refusal */
        i = ri_beam_set(&beam, o, d);
        fclose(stderr); stderr = saved;
        if (i != 0) { result[r] = -1; continue; }
        result[r] = ri_bvh_intersect_beam_visibility(accel, &beam, NULL);
    }
}

/* the beam-raster path through the reference's own ri_beam_set (beam.c:331-465), ri_raster_plane_setup (raster.c:42-147),
 * ri_bvh_invalidate_cache (bvh.c:420-428; the testbed's "MUST CALL", simplerender.cpp:693) and ri_bvh_intersect_beam
 * (bvh.c:544-609 -> bvh_traverse_beam :2547-2643 -> bvh_intersect_leaf_node_beam :2315-2426 -> project_triangles :2751-2820,
 * ri_beam_clip_by_triangle2d beam.c:469-730, ri_rasterize_beam raster.c:333-382).  The debug printf()s of that path go to
 * /dev/null (stdout swapped for the call, the reference is not edited).  frame9: du dv dw; t_out: w * h doubles = plane->t.
 * Returns -1 where ri_beam_set refuses the beam, else ri_bvh_intersect_beam's return value (always 0).
 * The reference writes plane->t[t * width + s] without a bounds check (raster.c:300-316): a beam whose footprint leaves the
 * raster window corrupts the heap -- callers run this in a child process and keep to beams inside the window. */
int lref_beam_raster(const double *org, const double *dirs, int w, int h, const double *frame9, const double *corner,
                     const double *eye, double fov, int invalidate, double *t_out)
{
    void *accel = ri_render_get()->scene->accel->data;
    static ri_raster_plane_t *plane = NULL;
    ri_beam_t beam; ri_vector_t o, d[4], fr[3], cn, ey; int i, k, rc;
    FILE *saved_err = stderr, *saved_out = stdout, *nul = fopen("/dev/null", "w");
    memset(&beam, 0, sizeof(beam));
    for (k = 0; k < 3; k++) { o[k] = org[k]; cn[k] = corner[k]; ey[k] = eye[k]; }
    o[3] = cn[3] = ey[3] = 0.0;
    for (i = 0; i < 4; i++) { for (k = 0; k < 3; k++) d[i][k] = dirs[3 * i + k]; d[i][3] = 0.0; }
    for (i = 0; i < 3; i++) { for (k = 0; k < 3; k++) fr[i][k] = frame9[3 * i + k]; fr[i][3] = 0.0; }
    stderr = nul; stdout = nul;
    rc = ri_beam_set(&beam, o, d);
    if (rc == 0) {
        if (!plane) plane = ri_raster_plane_new();
        ri_raster_plane_setup(plane, w, h, fr, cn, ey, fov);
        if (invalidate) ri_bvh_invalidate_cache(accel);
        rc = ri_bvh_intersect_beam(accel, &beam, plane, NULL);
        memcpy(t_out, plane->t, sizeof(double) * (size_t)w * (size_t)h);
    } else rc = -1;
    fflush(nul);
    stderr = saved_err; stdout = saved_out;
    fclose(nul);
    return rc;
}

#ifdef LREF_WITH_HIP
/* the scene built with the reference's CPU BVH (method 1) or through the RI_ACCEL_HIP glue (method 2) */
int lref_scene_build_with(int method)
{
    ri_scene_t *scene = ri_render_get()->scene;
    if (ri_accel_bind(scene->accel, method) != 0) return -1;
    return ri_scene_build_accel(scene);
}

/* lref_beam_raster with the scene's accelerator bound to RI_ACCEL_HIP: the reference's own ri_beam_set and
 * ri_raster_plane_setup, then the glue's ri_hipbvh_intersect_beam in place of ri_bvh_intersect_beam */
int lref_beam_raster_hip(const double *org, const double *dirs, int w, int h, const double *frame9, const double *corner,
                         const double *eye, double fov, double *t_out)
{
    void *accel = ri_render_get()->scene->accel->data;
    static ri_raster_plane_t *plane = NULL;
    ri_beam_t beam; ri_vector_t o, d[4], fr[3], cn, ey; int i, k, rc;
    FILE *saved_err = stderr, *saved_out = stdout, *nul = fopen("/dev/null", "w");
    memset(&beam, 0, sizeof(beam));
    for (k = 0; k < 3; k++) { o[k] = org[k]; cn[k] = corner[k]; ey[k] = eye[k]; }
    o[3] = cn[3] = ey[3] = 0.0;
    for (i = 0; i < 4; i++) { for (k = 0; k < 3; k++) d[i][k] = dirs[3 * i + k]; d[i][3] = 0.0; }
    for (i = 0; i < 3; i++) { for (k = 0; k < 3; k++) fr[i][k] = frame9[3 * i + k]; fr[i][3] = 0.0; }
    stderr = nul; stdout = nul;
    rc = ri_beam_set(&beam, o, d);
    if (rc == 0) {
        if (!plane) plane = ri_raster_plane_new();
        ri_raster_plane_setup(plane, w, h, fr, cn, ey, fov);
        rc = ri_hipbvh_intersect_beam(accel, &beam, plane, NULL);
        memcpy(t_out, plane->t, sizeof(double) * (size_t)w * (size_t)h);
    } else rc = -1;
    fflush(nul);
    stderr = saved_err; stdout = saved_out;
    fclose(nul);
    return rc;
}

/* lref_beam_visibility_batch with the scene's accelerator bound to RI_ACCEL_HIP: the reference's own ri_beam_set on the
 * reference's own ri_beam_t, then the glue's ri_hipbvh_intersect_beam_visibility (integration/ri_accel_hip.c) in place of
 * ri_bvh_intersect_beam_visibility -- one call per beam (together 0) or every accepted beam in one launch (together 1) */
void lref_beam_visibility_hip_batch(size_t n, const double *org, const double *dirs, int32_t *result, int together)
{
    size_t r, m = 0; void *accel = ri_render_get()->scene->accel->data;
    FILE *saved = stderr;
    ri_beam_t *beams = (ri_beam_t *)calloc(n ? n : 1, sizeof(ri_beam_t)); size_t *slot = (size_t *)calloc(n ? n : 1, sizeof(size_t));
    for (r = 0; r < n; r++) {
        ri_beam_t *beam = &beams[m]; ri_vector_t o, d[4]; int i, k;
        memset(beam, 0, sizeof(*beam));
        for (k = 0; k < 3; k++) o[k] = org[3 * r + k];
        o[3] = 0.0;
        for (i = 0; i < 4; i++) { for (k = 0; k < 3; k++) d[i][k] = dirs[12 * r + 3 * i + k]; d[i][3] = 0.0; }
        stderr = fopen("/dev/null", "w");
        i = ri_beam_set(beam, o, d);
        fclose(stderr); stderr = saved;
        if (i != 0) { result[r] = -1; continue; }
        if (!together) result[r] = ri_hipbvh_intersect_beam_visibility(accel, beam, NULL);
        else slot[m++] = r;
    }
    if (together && m) {
        int *cls = (int *)calloc(m, sizeof(int));
        if (ri_hipbvh_intersect_beam_visibility_n(accel, m, beams, cls) != 0) for (r = 0; r < m; r++) cls[r] = -2;
        for (r = 0; r < m; r++) result[slot[r]] = cls[r];
        free(cls);
    }
    free(beams); free(slot);
}
#endif

/* recorder control */
void   lref_record_start(void) { g_rec_n = 0; g_rec_on = 1; }
size_t lref_record_stop(void)  { g_rec_on = 0; return g_rec_n; }
size_t lref_record_size(void)  { return sizeof(lref_record_t); }
void   lref_record_copy(void *dst, size_t first, size_t count)
{
    memcpy(dst, g_rec + first, count * sizeof(lref_record_t));
}
