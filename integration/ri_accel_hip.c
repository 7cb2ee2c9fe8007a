/*
 * ri_accel_hip.c -- the reference-side glue a lucille maintainer adds to make
 * the MI355X accelerator a third accel method next to RI_ACCEL_UGRID/BVH.
 *
 * This file is compiled INSIDE lucille (it includes lucille's own headers:
 * src/render/accel.h, scene.h, geom.h, intersection_state.h, src/base/list.h)
 * and talks to liblucille_hip.so only through the flat C ABI of
 * include/lucille_hip.h.  Together with a three-line case in ri_accel_bind
 * (accel.c:72-109) and an "hip" string in the Option "raytrace" "accel_method"
 * parser (src/ri/option.c:453-462) it is the whole integration -- see
 * INTEGRATION.md.  oracle/Makefile builds it against the real reference
 * (oracle/_ref/liblucille_ref_hip.so) so tests can drive the reference's OWN
 * renderer through the GPU accelerator.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "accel.h"
#include "scene.h"
#include "geom.h"
#include "list.h"
#include "log.h"
#include "intersection_state.h"
#include "bvh.h"                 /* ri_bvh_diag_t: what `user` points at (bvh.h:103-110) */
#include "beam.h"
#include "raster.h"

#include "lucille_hip.h"

#ifndef RI_ACCEL_HIP
#define RI_ACCEL_HIP 2
#endif

#define RI_HIPBVH_MAGIC 0x48495042u   /* "HIPB": ri_hipbvh_handle is also asked about CPU accelerators */

typedef struct {
    unsigned    magic;
    lh_accel_t *lh;
    ri_geom_t **geoms;          /* back-pointers in geom_list order (cf. bvh.c:1808) */
    unsigned    ngeoms;
} ri_hipbvh_t;

static int g_ri_hip_device = 0;
void ri_hipbvh_set_device(int device) { g_ri_hip_device = device; }

/* RI_HIP_STATS_FILE=<path> (diagnostics, tests/test_gpu_dropin.py): how the one-ray calls of this accelerator were answered --
 * "<device batches> <rays in them>" of the coalesced device path (lh_accel_combine_statistics); 0 0 = every ray was answered by
 * the host walk.  Written when the accelerator is freed, or at exit if the renderer never frees it. */
static ri_hipbvh_t *g_ri_hip_last = NULL;
static void ri_hipbvh_dump_stats(void)
{
    const char *path = getenv("RI_HIP_STATS_FILE"); uint64_t st[2] = {0, 0}; FILE *f;
    if (!path || !g_ri_hip_last) return;
    if (lh_accel_combine_statistics(g_ri_hip_last->lh, st, 0) != 0) return;
    f = fopen(path, "w");
    if (f) { fprintf(f, "%llu %llu\n", (unsigned long long)st[0], (unsigned long long)st[1]); fclose(f); }
}

/* accel_build_func: walks scene->geom_list exactly like create_triangle_list
 * (bvh.c:1758-1821) so primitive ids match the CPU BVH's numbering */
void *ri_hipbvh_build(const void *data)
{
    const ri_scene_t *scene = (const ri_scene_t *)data;
    ri_hipbvh_t *h;
    ri_list_t *itr;
    unsigned n = 0;

    for (itr = ri_list_first((ri_list_t *)scene->geom_list); itr != NULL; itr = ri_list_next(itr)) n++;

    h = (ri_hipbvh_t *)calloc(1, sizeof(*h));
    h->magic = RI_HIPBVH_MAGIC;
    h->geoms = (ri_geom_t **)calloc(n ? n : 1, sizeof(ri_geom_t *));
    if (lh_accel_create(&h->lh, g_ri_hip_device) != 0) {
        ri_log(LOG_ERROR, "(HIPBVH) %s", lh_last_error());
        free(h->geoms); free(h);
        return NULL;
    }
    for (itr = ri_list_first((ri_list_t *)scene->geom_list); itr != NULL; itr = ri_list_next(itr)) {
        ri_geom_t *geom = (ri_geom_t *)itr->data;
        h->geoms[h->ngeoms++] = geom;
        /* positions are ri_vector_t = double[4]: stride 32 bytes */
        if (lh_accel_add_mesh(h->lh, geom->npositions, (const double *)geom->positions,
                              sizeof(ri_vector_t), geom->nindices, geom->indices) != 0) {
            /* a geom the accelerator refuses (an out-of-range index): no accelerator at all, as lh_host.c does -- skipping it
             * would let every later geom's ordinal run ahead of the mesh ordinal the hit records carry */
            ri_log(LOG_ERROR, "(HIPBVH) %s", lh_last_error());
            lh_accel_destroy(h->lh); free(h->geoms); free(h);
            return NULL;
        }
        /* what ri_intersection_state_build reads besides positions (geom.h:29-65): used by the device-side
         * epilogue of the batched frame loop (integration/ri_render_hip.c); the one-ray path below keeps calling
         * lucille's own ri_intersection_state_build on the host */
        {
            const uint32_t m = h->ngeoms - 1;
            if (getenv("RI_HIP_DEBUG"))
                fprintf(stderr, "(HIPBVH) geom %u: %u positions, %u indices, normals %p (%u), two_side %d, colors %p, tangents %p, texcoords %p / %p\n", m,
                        geom->npositions, geom->nindices, (void *)geom->normals, geom->nnormals, geom->two_side, (void *)geom->colors,
                        (void *)geom->tangents, (void *)geom->texcoords, (void *)geom->texcoords_unshared);
            int bad = 0;
            if (geom->normals || geom->two_side)
                bad |= lh_accel_set_normals(h->lh, m, (const double *)geom->normals, sizeof(ri_vector_t), geom->two_side);
            if (geom->colors) bad |= lh_accel_set_attribute(h->lh, m, LH_ATTR_COLOR, (const double *)geom->colors, sizeof(ri_vector_t), geom->ncolors);
            if (geom->tangents) bad |= lh_accel_set_attribute(h->lh, m, LH_ATTR_TANGENT, (const double *)geom->tangents, sizeof(ri_vector_t), geom->ntangents);
            if (geom->binormals) bad |= lh_accel_set_attribute(h->lh, m, LH_ATTR_BINORMAL, (const double *)geom->binormals, sizeof(ri_vector_t), geom->nbinormals);
            if (geom->texcoords) bad |= lh_accel_set_attribute(h->lh, m, LH_ATTR_TEXCOORD, geom->texcoords, 2 * sizeof(ri_float_t), geom->npositions);
            else if (geom->texcoords_unshared) bad |= lh_accel_set_attribute(h->lh, m, LH_ATTR_TEXCOORD_UNSHARED, geom->texcoords_unshared, 2 * sizeof(ri_float_t), geom->nindices);
            if (bad) {
                ri_log(LOG_ERROR, "(HIPBVH) geom %u: %s", m, lh_last_error());
                lh_accel_destroy(h->lh); free(h->geoms); free(h);
                return NULL;
            }
        }
    }
    if (lh_accel_commit(h->lh, 0) != 0) {
        ri_log(LOG_ERROR, "(HIPBVH) %s", lh_last_error());
        lh_accel_destroy(h->lh); free(h->geoms); free(h);
        return NULL;
    }
    if (getenv("RI_HIP_STATS_FILE")) {
        static int registered = 0;
        if (!registered) { registered = 1; atexit(ri_hipbvh_dump_stats); }
        g_ri_hip_last = h;
    }
    return h;
}

/* the C-ABI handle behind a RI_ACCEL_HIP accelerator: for callers that batch (integration/ri_render_hip.c) */
lh_accel_t *ri_hipbvh_handle(void *accel)
{
    ri_hipbvh_t *h = (ri_hipbvh_t *)accel;
    return (h && h->magic == RI_HIPBVH_MAGIC) ? h->lh : NULL;
}

/* accel_free_func */
void ri_hipbvh_free(void *accel)
{
    ri_hipbvh_t *h = (ri_hipbvh_t *)accel;
    if (!h) return;
    if (h == g_ri_hip_last) { ri_hipbvh_dump_stats(); g_ri_hip_last = NULL; }
    lh_accel_destroy(h->lh);
    free(h->geoms);
    free(h);
}

/* accel_intersect_func: one synchronous ray (the reference's calling
 * convention); transports that batch use lh_accel_intersect_host/_device */
int ri_hipbvh_intersect(void *accel, ri_ray_t *ray, ri_intersection_state_t *state, void *user)
{
    ri_hipbvh_t *h = (ri_hipbvh_t *)accel;
    uint32_t prim, mesh, index;
    double t, u, v;
    int hit;

    if (user) {          /* ri_bvh_diag_t (bvh.h:103-110): zeroed and filled as bvh.c:451-456 does, with this walk's numbers */
        ri_bvh_diag_t *dg = (ri_bvh_diag_t *)user; uint32_t d4[4] = {0, 0, 0, 0};
        memset(dg, 0, sizeof(*dg));
        if (lh_accel_intersect_diag_host(h->lh, 1, ray->org, ray->dir, &prim, &t, &u, &v, d4) != 0) return 0;
        dg->ninner_node_traversals = d4[0]; dg->nleaf_node_traversals = d4[1]; dg->ntriangle_isects = d4[2];
        hit = prim != 0xFFFFFFFFu;
    } else hit = lh_accel_intersect1(h->lh, ray->org, ray->dir, &prim, &t, &u, &v);
    if (hit <= 0) return 0;

    lh_accel_prim_lookup(h->lh, prim, &mesh, &index);
    state->t = t; state->u = u; state->v = v;
    state->geom = h->geoms[mesh];
    state->index = index;
    ri_intersection_state_build(state, ray->org, ray->dir);   /* as bvh.c:537-539 */
    return 1;
}

/* what ri_accel_bind's new case does */
int ri_accel_bind_hip(ri_accel_t *accel)
{
    ri_log(LOG_DEBUG, "(Accel ) Use HIP (MI355X) accelerator");
    accel->build     = ri_hipbvh_build;
    accel->free      = ri_hipbvh_free;
    accel->intersect = ri_hipbvh_intersect;
    return 0;
}

/* the members of the reference's own ri_beam_t (beam.h:45-84) that its two beam walks read, as ri_beam_set (beam.c:331-465) left
 * them: the device takes the beam as it is (lh_beam_set_t, lucille_hip.h) -- no member added to ri_beam_t, nothing recomputed */
static void hipbvh_beam_fields(lh_beam_set_t *o, const ri_beam_t *beam)
{
    int i, k;
    for (k = 0; k < 3; k++) { o->org[k] = beam->org[k]; o->dirsign[k] = beam->dirsign[k]; }
    for (i = 0; i < 4; i++) for (k = 0; k < 3; k++) { o->dir[i][k] = beam->dir[i][k]; o->normal[i][k] = beam->normal[i][k]; }
    o->dominant_axis = beam->dominant_axis;
}

/* ri_bvh_intersect_beam (bvh.h:203-206, bvh.c:544-609) on the device, the reference's signature: after the call raster_out->t
 * holds what the reference's own path leaves there for this beam (as right after ri_bvh_invalidate_cache: the leaf caches of
 * projected triangles are per beam here).  Returns 0 like the reference; `user` is ignored. */
int ri_hipbvh_intersect_beam(void *accel, ri_beam_t *beam, ri_raster_plane_t *raster_out, void *user)
{
    ri_hipbvh_t *h = (ri_hipbvh_t *)accel;
    lh_raster_plane_t lp; lh_beam_set_t b; double corner[3]; int32_t status = 0; int i, k;
    (void)user;
    if (!h || h->magic != RI_HIPBVH_MAGIC || !beam || !raster_out || !raster_out->t) return 0;
    hipbvh_beam_fields(&b, beam);
    lp.width = raster_out->width; lp.height = raster_out->height; lp.fov = raster_out->fov;
    for (i = 0; i < 3; i++) for (k = 0; k < 3; k++) lp.frame[3 * i + k] = raster_out->frame[i][k];
    for (k = 0; k < 3; k++) { lp.eye[k] = raster_out->org[k]; corner[k] = raster_out->corner[k]; }
    if (lh_accel_beam_raster_set_host(h->lh, 1, &b, corner, &lp, raster_out->t, &status, NULL) != 0)
        ri_log(LOG_ERROR, "(HIPBVH) %s", lh_last_error());
    return 0;
}

/* ri_bvh_intersect_beam_visibility (bvh.h:208-221, bvh.c:612-667) on the device, the reference's signature on the reference's
 * ri_beam_t: RI_BEAM_MISS_COMPLETELY / _HIT_COMPLETELY / _HIT_PARTIALLY (beam.h:27-29) exactly as the CPU BVH classifies the
 * beam (the walk runs on the bit-faithful rebuild of ITS tree); `user` is ignored as the reference ignores it. */
int ri_hipbvh_intersect_beam_visibility(void *accel, ri_beam_t *beam, void *user)
{
    ri_hipbvh_t *h = (ri_hipbvh_t *)accel;
    lh_beam_set_t b; int32_t cls = RI_BEAM_MISS_COMPLETELY;
    (void)user;
    if (!h || h->magic != RI_HIPBVH_MAGIC || !beam) return RI_BEAM_MISS_COMPLETELY;
    hipbvh_beam_fields(&b, beam);
    if (lh_accel_beam_visibility_set_host(h->lh, 1, &b, &cls) != 0) {
        ri_log(LOG_ERROR, "(HIPBVH) %s", lh_last_error());
        return RI_BEAM_MISS_COMPLETELY;
    }
    return (int)cls;
}

/* the same for n beams in one launch (a testbed-style caller that sets up a grid of beams first: simplerender.cpp:640-720) */
int ri_hipbvh_intersect_beam_visibility_n(void *accel, size_t n, ri_beam_t *beams, int *result)
{
    ri_hipbvh_t *h = (ri_hipbvh_t *)accel;
    lh_beam_set_t *b; int32_t *cls; size_t i; int rc = 0;
    if (!h || h->magic != RI_HIPBVH_MAGIC || !beams || !result) return -1;
    if (n == 0) return 0;
    b = (lh_beam_set_t *)malloc(n * sizeof(*b)); cls = (int32_t *)malloc(n * sizeof(*cls));
    if (!b || !cls) { free(b); free(cls); return -1; }
    for (i = 0; i < n; i++) hipbvh_beam_fields(&b[i], &beams[i]);
    if (lh_accel_beam_visibility_set_host(h->lh, n, b, cls) != 0) { ri_log(LOG_ERROR, "(HIPBVH) %s", lh_last_error()); rc = -1; }
    else for (i = 0; i < n; i++) result[i] = (int)cls[i];
    free(b); free(cls);
    return rc;
}
