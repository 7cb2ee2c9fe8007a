/*
 * lh_host.c -- host side (plain C) above the C ABI: the reference's plugin
 * interface for the ray-query path, bound to the HIP accelerator.
 * See include/lucille_accel.h for the reference file:line of every function.
 */
#include "lucille_accel.h"
#include "lucille_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void vcross(ri_vector_t d, const ri_vector_t a, const ri_vector_t b);

/* ---------------------------------------------------------------- geom ---- */

ri_geom_t *ri_geom_new(void) { return (ri_geom_t *)calloc(1, sizeof(ri_geom_t)); }

void ri_geom_free(ri_geom_t *g)
{
    if (!g) return;
    free(g->positions); free(g->normals); free(g->indices);
    free(g->tangents); free(g->binormals); free(g->colors); free(g->texcoords); free(g->texcoords_unshared);
    free(g);
}

static void *dup_bytes(const void *src, size_t n)
{
    void *p = malloc(n ? n : 1);
    if (p && n) memcpy(p, src, n);
    return p;
}

/* geom.c:78-99: silently ignores empty input */
void ri_geom_add_positions(ri_geom_t *g, unsigned int n, const ri_vector_t *p)
{
    if (!g || n == 0 || !p) return;
    free(g->positions);
    g->positions = (ri_vector_t *)dup_bytes(p, sizeof(ri_vector_t) * n); g->npositions = n;
}

void ri_geom_add_normals(ri_geom_t *g, unsigned int n, const ri_vector_t *p)
{
    if (!g || n == 0 || !p) return;
    free(g->normals);
    g->normals = (ri_vector_t *)dup_bytes(p, sizeof(ri_vector_t) * n); g->nnormals = n;
}

/* geom.c:123-290: the optional attributes ri_intersection_state_build interpolates */
void ri_geom_add_tangents(ri_geom_t *g, unsigned int n, const ri_vector_t *p)
{
    if (!g || n == 0 || !p) return;
    free(g->tangents);
    g->tangents = (ri_vector_t *)dup_bytes(p, sizeof(ri_vector_t) * n); g->ntangents = n;
}

void ri_geom_add_binormals(ri_geom_t *g, unsigned int n, const ri_vector_t *p)
{
    if (!g || n == 0 || !p) return;
    free(g->binormals);
    g->binormals = (ri_vector_t *)dup_bytes(p, sizeof(ri_vector_t) * n); g->nbinormals = n;
}

void ri_geom_add_colors(ri_geom_t *g, unsigned int n, const ri_vector_t *p)
{
    if (!g || n == 0 || !p) return;
    free(g->colors);
    g->colors = (ri_vector_t *)dup_bytes(p, sizeof(ri_vector_t) * n); g->ncolors = n;
}

void ri_geom_add_texcoords(ri_geom_t *g, unsigned int n, const ri_float_t *st)
{
    if (!g || n == 0 || !st) return;
    free(g->texcoords);
    g->texcoords = (ri_float_t *)dup_bytes(st, sizeof(ri_float_t) * 2 * n); g->ntexcoords = n;
}

void ri_geom_add_texcoords_unshared(ri_geom_t *g, unsigned int n, const ri_float_t *st)
{
    if (!g || n == 0 || !st) return;
    free(g->texcoords_unshared);
    g->texcoords_unshared = (ri_float_t *)dup_bytes(st, sizeof(ri_float_t) * 2 * n); g->ntexcoords = n;
}

void ri_geom_add_indices(ri_geom_t *g, unsigned int n, const unsigned int *idx)
{
    if (!g || n == 0 || !idx) return;
    free(g->indices);
    g->indices = (unsigned int *)dup_bytes(idx, sizeof(unsigned int) * n); g->nindices = n;
}

/* --------------------------------------------------------------- scene ---- */

ri_scene_t *ri_scene_new(void)
{
    ri_scene_t *s = (ri_scene_t *)calloc(1, sizeof(ri_scene_t));
    if (s) s->accel = ri_accel_new();            /* scene.c:43 */
    return s;
}

void ri_scene_free(ri_scene_t *s)
{
    if (!s) return;
    ri_accel_free(s->accel);                     /* scene.c:73 */
    free(s->geom_list);
    free(s);
}

void ri_scene_add_geom(ri_scene_t *s, const ri_geom_t *g)
{
    ri_geom_t **nl;
    if (!s || !g) return;
    nl = (ri_geom_t **)realloc(s->geom_list, sizeof(ri_geom_t *) * (s->ngeoms + 1));
    if (!nl) return;
    s->geom_list = nl;
    s->geom_list[s->ngeoms++] = (ri_geom_t *)g;
}

int ri_scene_build_accel(ri_scene_t *s)
{
    if (!s || !s->accel || !s->accel->build) {
        fprintf(stderr, "[lucille_hip] FATAL : No spatial accelerator is assigned to the scene.\n");
        return -1;                               /* scene.c:158-161 */
    }
    /* the reference overwrites ->data and leaks the old tree when a scene is rebuilt
     * (scene.c:97,164); device memory is not something to leak: release it first */
    if (s->accel->data && s->accel->free) { s->accel->free(s->accel->data); s->accel->data = NULL; }
    s->accel->data = s->accel->build((const void *)s);
    return s->accel->data ? 0 : -1;
}

/* --------------------------------------------------------------- accel ---- */

ri_accel_t *ri_accel_new(void) { return (ri_accel_t *)calloc(1, sizeof(ri_accel_t)); }

void ri_accel_free(ri_accel_t *a)
{
    if (!a) return;
    if (a->free && a->data) a->free(a->data);    /* accel.c:45-50 */
    free(a);
}

int ri_accel_bind(ri_accel_t *a, int method)
{
    if (!a) return -1;
    switch (method) {
    case RI_ACCEL_HIP:
        a->build = ri_hipbvh_build; a->free = ri_hipbvh_free; a->intersect = ri_hipbvh_intersect;
        return 0;
    case RI_ACCEL_UGRID:
    case RI_ACCEL_BVH:
        fprintf(stderr, "[lucille_hip] ERROR : (Accel ) CPU accel method %d lives in lucille itself; "
                        "this library provides RI_ACCEL_HIP only (no CPU fallback)\n", method);
        return -1;
    default:
        fprintf(stderr, "[lucille_hip] ERROR : (Accel ) Unknown accel method\n");   /* accel.c:104 */
        return -1;
    }
}

/* what `void *accel` points to for RI_ACCEL_HIP */
typedef struct {
    lh_accel_t  *lh;
    ri_geom_t  **geoms; unsigned int ngeoms;     /* borrowed back-pointers, as in bvh.c:1808 */
} hipbvh_t;

static int g_device = 0;

void *ri_hipbvh_build(const void *data)
{
    const ri_scene_t *scene = (const ri_scene_t *)data;
    hipbvh_t *h; unsigned int g;
    if (!scene) return NULL;
    h = (hipbvh_t *)calloc(1, sizeof(*h));
    if (!h) return NULL;
    if (lh_accel_create(&h->lh, g_device) != 0) {
        fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error());
        free(h); return NULL;
    }
    h->ngeoms = scene->ngeoms;
    h->geoms = (ri_geom_t **)dup_bytes(scene->geom_list, sizeof(ri_geom_t *) * scene->ngeoms);
    for (g = 0; g < scene->ngeoms; g++) {
        const ri_geom_t *geom = scene->geom_list[g];
        if (lh_accel_add_mesh(h->lh, geom->npositions, (const double *)geom->positions,
                              sizeof(ri_vector_t), geom->nindices, geom->indices) != 0) {
            fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error());
            ri_hipbvh_free(h); return NULL;
        }
        /* the attributes of the geom go along, so that the device-side epilogue (lh_accel_state_build_*, the tile
         * pipelines) interpolates what ri_intersection_state_build would */
        if (geom->normals || geom->two_side) lh_accel_set_normals(h->lh, g, (const double *)geom->normals, sizeof(ri_vector_t), geom->two_side);
        if (geom->colors) lh_accel_set_attribute(h->lh, g, LH_ATTR_COLOR, (const double *)geom->colors, sizeof(ri_vector_t), geom->ncolors);
        if (geom->tangents) lh_accel_set_attribute(h->lh, g, LH_ATTR_TANGENT, (const double *)geom->tangents, sizeof(ri_vector_t), geom->ntangents);
        if (geom->binormals) lh_accel_set_attribute(h->lh, g, LH_ATTR_BINORMAL, (const double *)geom->binormals, sizeof(ri_vector_t), geom->nbinormals);
        if (geom->texcoords) lh_accel_set_attribute(h->lh, g, LH_ATTR_TEXCOORD, geom->texcoords, 2 * sizeof(ri_float_t), geom->npositions);
        else if (geom->texcoords_unshared) lh_accel_set_attribute(h->lh, g, LH_ATTR_TEXCOORD_UNSHARED, geom->texcoords_unshared, 2 * sizeof(ri_float_t), geom->nindices);
    }
    if (lh_accel_commit(h->lh, 0) != 0) {
        fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error());
        ri_hipbvh_free(h); return NULL;
    }
    return h;
}

void ri_hipbvh_free(void *accel)
{
    hipbvh_t *h = (hipbvh_t *)accel;
    if (!h) return;
    lh_accel_destroy(h->lh);
    free(h->geoms);
    free(h);
}

/* -------------------------------------------- traversal statistics ---- */
/* ri_bvh_clear_stat_traversal / ri_bvh_report_stat_traversal (bvh.c:669-706): process-wide
 * totals like the reference's g_stattrav; filled from the accelerator's counting kernel. */
static int      g_trace_stats = -1;          /* -1: read RI_BVH_TRACE_STATISTICS on first use */
static uint64_t g_stat[5];                   /* node visits, filter tests, fp64 tests, rays, hits */

static int trace_stats_on(void)
{
    if (g_trace_stats < 0) {
        const char *e = getenv("RI_BVH_TRACE_STATISTICS");
        g_trace_stats = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return g_trace_stats;
}

static void stat_before(hipbvh_t *h) { lh_accel_trace_statistics(h->lh, trace_stats_on()); }

static void stat_after(hipbvh_t *h)
{
    uint64_t c[5]; int k;
    if (!trace_stats_on()) return;
    if (lh_accel_statistics(h->lh, c, 1) == 0) for (k = 0; k < 5; k++) g_stat[k] += c[k];
}

void ri_hipbvh_trace_statistics(int enable) { g_trace_stats = enable != 0; }
void ri_hipbvh_clear_stat_traversal(void) { memset(g_stat, 0, sizeof(g_stat)); }
void ri_hipbvh_get_stat_traversal(uint64_t out[5])
{
    out[0] = g_stat[3]; out[1] = g_stat[0]; out[2] = g_stat[1]; out[3] = g_stat[2]; out[4] = g_stat[4];
}

void ri_hipbvh_report_stat_traversal(void)
{   /* the reference's layout (bvh.c:681-706); leaf visits are not a unit of this kernel
     * (leaves are parked and drained in batches), fp64 re-tests are listed instead */
    const double nrays = g_stat[3] ? (double)g_stat[3] : 1.0;     /* no rays traced yet: print zeros, not NaN */
    const double tested = g_stat[1] / nrays, hit = g_stat[4] / nrays;
    printf("== BVH traversal statistiscs ==================================================\n");
    printf("# of rays                    %llu\n", (unsigned long long)g_stat[3]);
    printf("# of inner node travs        %llu\n", (unsigned long long)g_stat[0]);
    printf("  Per ray                    %f\n", g_stat[0] / nrays);
    printf("# of tested triangles        %llu\n", (unsigned long long)g_stat[1]);
    printf("  Per ray                    %f\n", tested);
    printf("# of fp64 re-tested tris     %llu\n", (unsigned long long)g_stat[2]);
    printf("  Per ray                    %f\n", g_stat[2] / nrays);
    printf("# of actually hit triangles  %llu\n", (unsigned long long)g_stat[4]);
    printf("  Per ray                    %f\n", hit);
    printf("  Hit rate                   %f %%\n", tested > 0.0 ? 100.0 * (hit / tested) : 0.0);
    printf("===============================================================================\n");
}

static void fill_state(hipbvh_t *h, ri_intersection_state_t *st, uint32_t prim, double t, double u, double v,
                       const ri_vector_t org, const ri_vector_t dir)
{
    uint32_t mesh = 0, index = 0;
    lh_accel_prim_lookup(h->lh, prim, &mesh, &index);
    st->t = t; st->u = u; st->v = v;
    st->geom = h->geoms[mesh]; st->index = index;
    ri_intersection_state_build(st, org, dir);   /* bvh.c:537-539 */
}

int ri_hipbvh_intersect(void *accel, ri_ray_t *ray, ri_intersection_state_t *state, void *user)
{
    hipbvh_t *h = (hipbvh_t *)accel;
    uint32_t prim; double t, u, v; int hit, k;
    if (!h || !ray || !state) return 0;
    /* the scratch members the reference writes into the caller's ray (bvh.c:473-497) */
    for (k = 0; k < 3; k++) {
        ray->dir_sign[k] = (ray->dir[k] < 0.0) ? 1 : 0;
        ray->invdir[k] = (fabs(ray->dir[k]) > 1.0e-14) ? 1.0 / ray->dir[k]
                                                       : ((ray->dir[k] < 0.0) ? -1.7976931348623157e308 : 1.7976931348623157e308);
    }
    stat_before(h);
    if (user) {          /* a ri_bvh_diag_t (bvh.c:451-456): this ray's own numbers, from the sequential walk */
        ri_bvh_diag_t *dg = (ri_bvh_diag_t *)user; uint32_t d4[4] = {0, 0, 0, 0};
        memset(dg, 0, sizeof(*dg));
        hit = lh_accel_intersect_diag_host(h->lh, 1, ray->org, ray->dir, &prim, &t, &u, &v, d4);
        if (hit == 0) { hit = prim != 0xFFFFFFFFu; dg->ninner_node_traversals = d4[0]; dg->nleaf_node_traversals = d4[1]; dg->ntriangle_isects = d4[2]; }
    } else hit = lh_accel_intersect1(h->lh, ray->org, ray->dir, &prim, &t, &u, &v);
    stat_after(h);
    if (hit < 0) { fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error()); return 0; }
    /* bvh_traverse initialises these whether or not there is a hit (bvh.c:1111-1115) */
    state->t = 1.0e38; state->u = 0.0; state->v = 0.0; state->geom = NULL; state->index = 0;
    if (hit) fill_state(h, state, prim, t, u, v, ray->org, ray->dir);
    return hit;
}

/* -------------------------------------------------------------- render ---- */

static ri_render_t *g_render = NULL;

void ri_render_init(void)
{
    if (g_render) return;                        /* render.c:177-179 */
    g_render = (ri_render_t *)calloc(1, sizeof(ri_render_t));
    g_render->scene = ri_scene_new();
    g_render->device = g_device;
}

ri_render_t *ri_render_get(void) { return g_render; }

void ri_render_free(void)
{
    if (!g_render) return;
    ri_scene_free(g_render->scene);
    free(g_render); g_render = NULL;
}

/* ------------------------------------------------------------- queries ---- */

int ri_raytrace(ri_render_t *render, ri_ray_t *ray, ri_intersection_state_t *state_out)
{
    int hit; ri_intersection_state_t state;
    if (!render || !render->scene || !render->scene->accel || !render->scene->accel->intersect) return 0;
    render->stat.nrays++;                        /* raytrace.c:43 */
    memset(&state, 0, sizeof(state));
    state.inside = 0; ray->t = 0.0f;             /* raytrace.c:48-49 */
    hit = render->scene->accel->intersect(render->scene->accel->data, ray, &state, NULL);
    if (hit) memcpy(state_out, &state, sizeof(state));   /* raytrace.c:64-66 */
    return hit;
}

int ri_accel_intersect_batch(void *accel, size_t n, const double *org, const double *dir, uint32_t *prim,
                             double *t, double *u, double *v, uint8_t *occluded, int mode)
{
    hipbvh_t *h = (hipbvh_t *)accel;
    if (!h) return -1;
    stat_before(h);
    if (lh_accel_intersect_host(h->lh, n, org, dir, prim, t, u, v, occluded, mode) != 0) {
        fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error());
        return -1;
    }
    stat_after(h);
    return 0;
}

int ri_render_tile_ao(void *accel, const ri_tile_camera_t *camera, int x0, int y0, int w, int h,
                      int pixel_samples, int gather_nsamples, uint64_t seed, float *rgb)
{
    hipbvh_t *hb = (hipbvh_t *)accel;
    lh_camera_t cam; int k;
    if (!hb || !camera || !rgb) return -1;
    cam.width = camera->width; cam.height = camera->height; cam.rh = camera->rh; cam.ortho = camera->ortho;
    cam.flength = camera->flength;
    for (k = 0; k < 16; k++) cam.cam2world[k] = camera->cam2world[k];
    if (lh_render_ao_tile_host(hb->lh, &cam, x0, y0, w, h, pixel_samples, gather_nsamples, seed, NULL, 0, rgb, NULL) != 0) {
        fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error());
        return -1;
    }
    return 0;
}

/* ---------------------------------------------------------------- beams ---- */

/* ri_beam_set, src/render/beam.c:331-465 (incl. the maxval re-assignment at :387-390) */
int ri_beam_set(ri_beam_t *beam, ri_vector_t org, ri_vector_t dir[4])
{
    int i, j, dom; double maxval, normal[3] = {0.0, 0.0, 0.0};
    if (!beam || !org || !dir) return -1;
    beam->d = 1024.0; beam->t_max = 1.0e38;
    for (i = 0; i < 3; i++) {
        int zeros = 0, mask = 0;
        for (j = 0; j < 4; j++) {
            if (fabs(dir[j][i]) < 1.0e-14) zeros++;
            else mask += (dir[j][i] < 0.0) ? 1 : -1;
        }
        if ((mask != -(4 - zeros)) && (mask != (4 - zeros))) {
            fprintf(stderr, "TODO: Beam's dir does not have same sign.\n");
            return -1;
        }
    }
    memcpy(beam->org, org, sizeof(ri_vector_t));
    maxval = fabs(dir[0][0]); dom = 0;
    if (maxval < fabs(dir[0][1])) { maxval = fabs(dir[0][0]); dom = 1; }
    if (maxval < fabs(dir[0][2])) { maxval = fabs(dir[0][2]); dom = 2; }
    beam->dominant_axis = dom;
    for (i = 0; i < 3; i++) beam->dirsign[i] = (dir[0][i] < 0.0) ? 1 : 0;
    normal[dom] = beam->dirsign[dom] ? -1.0 : 1.0;
    for (i = 0; i < 4; i++) {
        const double t = dir[i][0] * normal[0] + dir[i][1] * normal[1] + dir[i][2] * normal[2];
        const double k = (fabs(t) > 1.0e-14) ? beam->d / t : 1.0;
        for (j = 0; j < 3; j++) {
            beam->dir[i][j] = k * dir[i][j];
            beam->invdir[i][j] = (fabs(beam->dir[i][j]) > 1.0e-14) ? 1.0 / beam->dir[i][j] : 1.7976931348623157e308;
        }
    }
    vcross(beam->normal[0], beam->dir[1], beam->dir[0]);
    vcross(beam->normal[1], beam->dir[2], beam->dir[1]);
    vcross(beam->normal[2], beam->dir[3], beam->dir[2]);
    vcross(beam->normal[3], beam->dir[0], beam->dir[3]);
    beam->is_tetrahedron = 0;
    return 0;
}

int ri_hipbvh_intersect_beam_visibility_batch(void *accel, size_t n, const double *org, const double *dirs, int32_t *result)
{
    hipbvh_t *h = (hipbvh_t *)accel;
    if (!h) return -1;
    if (lh_accel_beam_visibility_host(h->lh, n, org, dirs, result) != 0) {
        fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error());
        return -1;
    }
    return 0;
}

/* the members of a set-up ri_beam_t the beam walks read (lh_beam_set_t, lucille_hip.h) */
static void beam_fields(lh_beam_set_t *o, const ri_beam_t *beam)
{
    int i, k;
    for (k = 0; k < 3; k++) { o->org[k] = beam->org[k]; o->dirsign[k] = beam->dirsign[k]; }
    for (i = 0; i < 4; i++) for (k = 0; k < 3; k++) { o->dir[i][k] = beam->dir[i][k]; o->normal[i][k] = beam->normal[i][k]; }
    o->dominant_axis = beam->dominant_axis;
}

/* ri_bvh_intersect_beam_visibility, bvh.c:612-667: the beam was accepted by ri_beam_set */
int ri_hipbvh_intersect_beam_visibility(void *accel, ri_beam_t *beam, void *user)
{
    hipbvh_t *h = (hipbvh_t *)accel; lh_beam_set_t b; int32_t cls = 0;
    (void)user;
    if (!h || !beam) return 0;
    beam_fields(&b, beam);
    if (lh_accel_beam_visibility_set_host(h->lh, 1, &b, &cls) != 0) {
        fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error());
        return 0;
    }
    return cls < 0 ? 0 : (int)cls;
}

/* ---------------------------------------------------------- beam raster ---- */

/* ri_raster_plane_new / _setup / _free, src/render/raster.c:24-160 */
ri_raster_plane_t *ri_raster_plane_new(void)
{
    return (ri_raster_plane_t *)calloc(1, sizeof(ri_raster_plane_t));
}

int ri_raster_plane_setup(ri_raster_plane_t *pl, int width, int height, ri_vector_t frame[3], ri_vector_t corner, ri_vector_t org,
                          ri_float_t fov)
{
    size_t sz; double fov_rad, w[3], p[2];
    if (!pl || width <= 0 || height <= 0) return -1;
    sz = (size_t)width * (size_t)height;
    free(pl->t); free(pl->u); free(pl->v); free(pl->geom); free(pl->index);
    pl->width = width; pl->height = height;
    pl->t = (ri_float_t *)calloc(sz, sizeof(ri_float_t)); pl->u = (ri_float_t *)calloc(sz, sizeof(ri_float_t));
    pl->v = (ri_float_t *)calloc(sz, sizeof(ri_float_t)); pl->geom = (ri_geom_t **)calloc(sz, sizeof(ri_geom_t *));
    pl->index = (uint32_t *)calloc(sz, sizeof(uint32_t));
    if (!pl->t || !pl->u || !pl->v || !pl->geom || !pl->index) return -1;
    memcpy(pl->frame, frame, sizeof(ri_vector_t) * 3);
    memcpy(pl->corner, corner, sizeof(ri_vector_t)); memcpy(pl->org, org, sizeof(ri_vector_t));
    pl->fov = fov;
    /* the lower-left corner in NDC (raster.c:114-144) */
    fov_rad = pl->fov * M_PI / 180.0;
    w[0] =  pl->frame[0][0] * corner[0] + pl->frame[0][1] * corner[1] + pl->frame[0][2] * corner[2];
    w[1] =  pl->frame[1][0] * corner[0] + pl->frame[1][1] * corner[1] + pl->frame[1][2] * corner[2];
    w[2] = -pl->frame[2][0] * corner[0] - pl->frame[2][1] * corner[1] - pl->frame[2][2] * corner[2];
    p[0] = (1.0 / tan(0.5 * fov_rad)) * w[0]; p[1] = (1.0 / tan(0.5 * fov_rad)) * w[1];
    p[0] /= -w[2]; p[1] /= -w[2];
    pl->offset[0] = p[0]; pl->offset[1] = p[1];
    return 0;
}

int ri_raster_plane_free(ri_raster_plane_t *pl)
{
    if (pl == NULL) return -1;
    free(pl->t); free(pl->u); free(pl->v); free(pl->geom); free(pl->index);
    pl->t = pl->u = pl->v = NULL; pl->geom = NULL; pl->index = NULL;
    return 0;
}

int ri_hipbvh_intersect_beam_batch(void *accel, size_t n, const double *org, const double *dirs, const double *corners, int width,
                                   int height, const double *frame9, const double *eye, double fov, double *t_out, int32_t *status,
                                   uint64_t *flags)
{
    hipbvh_t *h = (hipbvh_t *)accel; lh_raster_plane_t pl; int k;
    if (!h || !frame9 || !eye) return -1;
    pl.width = width; pl.height = height; pl.fov = fov;
    for (k = 0; k < 9; k++) pl.frame[k] = frame9[k];
    for (k = 0; k < 3; k++) pl.eye[k] = eye[k];
    if (lh_accel_beam_raster_host(h->lh, n, org, dirs, corners, &pl, t_out, status, flags) != 0) {
        fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error());
        return -1;
    }
    return 0;
}

/* ri_bvh_intersect_beam, bvh.c:544-609: the beam was accepted by ri_beam_set; always returns 0 */
int ri_hipbvh_intersect_beam(void *accel, ri_beam_t *beam, ri_raster_plane_t *raster_out, void *user)
{
    hipbvh_t *h = (hipbvh_t *)accel; lh_beam_set_t b; lh_raster_plane_t pl; double corner[3]; int32_t st = 0; int i, k;
    (void)user;
    if (!h || !beam || !raster_out || !raster_out->t) return 0;
    beam_fields(&b, beam);
    pl.width = raster_out->width; pl.height = raster_out->height; pl.fov = raster_out->fov;
    for (i = 0; i < 3; i++) for (k = 0; k < 3; k++) pl.frame[3 * i + k] = raster_out->frame[i][k];
    for (k = 0; k < 3; k++) { corner[k] = raster_out->corner[k]; pl.eye[k] = raster_out->org[k]; }
    if (lh_accel_beam_raster_set_host(h->lh, 1, &b, corner, &pl, raster_out->t, &st, NULL) != 0)
        fprintf(stderr, "[lucille_hip] ERROR : (HIPBVH) %s\n", lh_last_error());
    return 0;
}

void ri_hipbvh_invalidate_cache(void *accel) { (void)accel; }

int ri_accel_prim_lookup(void *accel, uint32_t prim, ri_geom_t **geom, uint32_t *index)
{
    hipbvh_t *h = (hipbvh_t *)accel; uint32_t mesh = 0, idx = 0;
    if (!h || lh_accel_prim_lookup(h->lh, prim, &mesh, &idx) != 0) return -1;
    if (geom) *geom = h->geoms[mesh];
    if (index) *index = idx;
    return 0;
}

long ri_raytrace_batch(ri_render_t *render, size_t n, ri_ray_t *rays, ri_intersection_state_t *states, int *hit)
{
    hipbvh_t *h; double *org, *dir, *t, *u, *v; uint32_t *prim; size_t i; long nhit = 0; int k;
    if (!render || !render->scene || !render->scene->accel) return -1;
    if (render->scene->accel->intersect != ri_hipbvh_intersect) {
        /* any other accelerator: the reference's own per-ray loop */
        for (i = 0; i < n; i++) { hit[i] = ri_raytrace(render, &rays[i], &states[i]); nhit += hit[i]; }
        return nhit;
    }
    h = (hipbvh_t *)render->scene->accel->data;
    if (!h) return -1;
    if (n == 0) return 0;
    org = (double *)malloc(sizeof(double) * 3 * n); dir = (double *)malloc(sizeof(double) * 3 * n);
    t = (double *)malloc(sizeof(double) * n); u = (double *)malloc(sizeof(double) * n); v = (double *)malloc(sizeof(double) * n);
    prim = (uint32_t *)malloc(sizeof(uint32_t) * n);
    if (!org || !dir || !t || !u || !v || !prim) { nhit = -1; goto done; }
    for (i = 0; i < n; i++)
        for (k = 0; k < 3; k++) { org[3 * i + k] = rays[i].org[k]; dir[3 * i + k] = rays[i].dir[k]; }
    if (ri_accel_intersect_batch(h, n, org, dir, prim, t, u, v, NULL, LH_MODE_CLOSEST) != 0) { nhit = -1; goto done; }
    render->stat.nrays += n;
    for (i = 0; i < n; i++) {
        rays[i].t = 0.0f;
        hit[i] = prim[i] != LH_MISS;
        if (hit[i]) {
            memset(&states[i], 0, sizeof(states[i]));
            fill_state(h, &states[i], prim[i], t[i], u[i], v[i], rays[i].org, rays[i].dir);
            nhit++;
        }
    }
done:
    free(org); free(dir); free(t); free(u); free(v); free(prim);
    return nhit;
}

/* --------------------------------------------------- hit epilogue (host) ---- */

static void vnormalize(ri_vector_t d)
{   /* ri_vector_normalize, src/base/vector.h:75-86 (threshold is the FLOAT 1.0e-17f) */
    double norm2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (norm2 > 1.0e-17f) { double rsq = 1.0 / sqrt(norm2); d[0] *= rsq; d[1] *= rsq; d[2] *= rsq; }
}

static void vcross(ri_vector_t d, const ri_vector_t a, const ri_vector_t b)
{
    d[0] = a[1] * b[2] - a[2] * b[1]; d[1] = a[2] * b[0] - a[0] * b[2]; d[2] = a[0] * b[1] - a[1] * b[0];
}

/* ri_ortho_basis, src/render/reflection.c:311-333 */
static void ortho_basis(ri_vector_t basis[3], const ri_vector_t n)
{
    int i;
    memcpy(basis[2], n, sizeof(ri_vector_t));
    basis[1][0] = basis[1][1] = basis[1][2] = basis[1][3] = 0.0;
    for (i = 0; i < 3; i++) if (basis[2][i] < 0.6 && basis[2][i] > -0.6) break;
    if (i >= 3) i = 0;
    basis[1][i] = 1.0;
    vcross(basis[0], basis[1], basis[2]); vnormalize(basis[0]);
    vcross(basis[1], basis[2], basis[0]); vnormalize(basis[1]);
}

/* intersection_state.c:99-248, for the members this struct carries */
void ri_intersection_state_build(ri_intersection_state_t *st, const ri_vector_t eye, const ri_vector_t dir)
{
    const ri_geom_t *geom = st->geom; uint32_t index = st->index;
    const double t = st->t, u = st->u, v = st->v;
    uint32_t i0, i1, i2; ri_vector_t v01, v02, basis[3]; int k;
    st->P[0] = eye[0] + dir[0] * t; st->P[1] = eye[1] + dir[1] * t; st->P[2] = eye[2] + dir[2] * t; st->P[3] = 0.0;
    memcpy(st->I, dir, sizeof(ri_vector_t)); vnormalize(st->I);
    memcpy(st->E, eye, sizeof(ri_vector_t));
    i0 = geom->indices[index + 0]; i1 = geom->indices[index + 1]; i2 = geom->indices[index + 2];
    /* ri_normal_of_triangle, src/base/geometric.c:31-44 */
    for (k = 0; k < 3; k++) { v01[k] = geom->positions[i1][k] - geom->positions[i0][k]; v02[k] = geom->positions[i2][k] - geom->positions[i0][k]; }
    vcross(st->Ng, v01, v02); vnormalize(st->Ng);
    if (geom->normals) {
        /* ri_lerp_vector, geometric.c:51-72: (1-u-v) n0 + u n1 + v n2, in that order */
        const double w = 1.0 - u - v;
        for (k = 0; k < 3; k++) {
            double a = geom->normals[i0][k] * w, b = geom->normals[i1][k] * u, c = geom->normals[i2][k] * v;
            st->Ns[k] = (a + b) + c;
        }
    } else {
        memcpy(st->Ns, st->Ng, sizeof(ri_vector_t));
    }
    if (geom->normals && geom->tangents && geom->binormals) {     /* :161-176: only looked at next to normals */
        const double w = 1.0 - u - v;
        for (k = 0; k < 3; k++) {
            double a = geom->tangents[i0][k] * w, b = geom->tangents[i1][k] * u, c = geom->tangents[i2][k] * v;
            st->tangent[k] = (a + b) + c;
            a = geom->binormals[i0][k] * w; b = geom->binormals[i1][k] * u; c = geom->binormals[i2][k] * v;
            st->binormal[k] = (a + b) + c;
        }
    } else {
        ortho_basis(basis, st->Ng);
        memcpy(st->tangent, basis[0], sizeof(ri_vector_t));
        memcpy(st->binormal, basis[1], sizeof(ri_vector_t));
    }
    if (geom->colors) {                                            /* :188-200 */
        const double w = 1.0 - u - v;
        for (k = 0; k < 3; k++) { double a = geom->colors[i0][k] * w, b = geom->colors[i1][k] * u, c = geom->colors[i2][k] * v; st->color[k] = (a + b) + c; }
    } else { st->color[0] = st->color[1] = st->color[2] = 1.0; }
    st->color[3] = geom->colors ? 0.0 : 1.0;                       /* ri_vector_set1(defcol, 1.0) sets all four */
    if (geom->texcoords) {                                         /* :210-231, lerp_uv :266-280 */
        const ri_float_t *a = &geom->texcoords[2 * i0], *b = &geom->texcoords[2 * i1], *c = &geom->texcoords[2 * i2];
        st->stqr[0] = (1 - u - v) * a[0] + u * b[0] + v * c[0]; st->stqr[1] = (1 - u - v) * a[1] + u * b[1] + v * c[1];
    } else if (geom->texcoords_unshared) {
        const ri_float_t *a = &geom->texcoords_unshared[2 * (index + 0)], *b = &geom->texcoords_unshared[2 * (index + 1)],
                         *c = &geom->texcoords_unshared[2 * (index + 2)];
        st->stqr[0] = (1 - u - v) * a[0] + u * b[0] + v * c[0]; st->stqr[1] = (1 - u - v) * a[1] + u * b[1] + v * c[1];
    } else st->stqr[0] = st->stqr[1] = 0.0;
    st->inside = (geom->two_side && index >= geom->nindices / 2) ? 1 : 0;
}
