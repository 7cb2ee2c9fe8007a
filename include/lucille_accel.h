/*
 * lucille_accel.h -- host-side (plain C) mirror of lucille's accelerator plugin
 * interface for the ray-query path, with the HIP accelerator as a new method.
 *
 * Same names, argument meaning and error behaviour as the reference:
 *
 *   ri_accel_t, accel_*_func, ri_accel_new/free/bind   src/render/accel.h:24-92, accel.c:28-109
 *   ri_raytrace                                        src/render/raytrace.h:46-49, raytrace.c:31-69
 *   ri_geom_new / _add_positions / _add_indices ...    src/render/geom.h:67-131
 *   ri_scene_new / _add_geom / _build_accel            src/render/scene.h:60-96, scene.c:84-167
 *   ri_intersection_state_build                        src/render/intersection_state.c:99-248
 *   ri_render_init / ri_render_get                     src/render/render.h:104-105
 *
 * so that code written against lucille's API for THIS path (build a scene of
 * ri_geom_t, bind an accelerator, call ri_raytrace per ray) compiles and runs
 * against liblucille_hip.so, and parity tests read like lucille programs.
 * Structs carry only the members this path reads or writes; they are this
 * library's own layout (inside lucille itself the glue in
 * integration/ri_accel_hip.c adapts lucille's real structs to the flat C ABI of
 * lucille_hip.h -- see INTEGRATION.md).
 *
 * New relative to the reference, because a one-ray-synchronous vtable cannot
 * feed a GPU: RI_ACCEL_HIP, ri_raytrace_batch(), ri_accel_intersect_batch().
 */
#ifndef LUCILLE_ACCEL_H
#define LUCILLE_ACCEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double ri_float_t;          /* src/base/common.h:24-27 */
typedef double ri_vector_t[4];      /* src/base/vector.h:61    */

#define RI_ACCEL_UGRID 0            /* accel.h:20 (lives in lucille; not provided here) */
#define RI_ACCEL_BVH   1            /* accel.h:21 (lives in lucille; not provided here) */
#define RI_ACCEL_HIP   2            /* new: MI355X accelerator, option string "hip"     */

/* ---- geometry: the BVH's input (geom.h:29-65) -------------------------- */
typedef struct _ri_geom_t {
    ri_vector_t  *positions;   unsigned int npositions;
    ri_vector_t  *normals;     unsigned int nnormals;
    unsigned int *indices;     unsigned int nindices;
    int           two_side;
    ri_vector_t  *tangents;    unsigned int ntangents;      /* geom.h:34-39: looked at only next to normals */
    ri_vector_t  *binormals;   unsigned int nbinormals;
    ri_vector_t  *colors;      unsigned int ncolors;        /* vertex colour (Cs) */
    ri_float_t   *texcoords;                                /* st per vertex (geom.h:42-48) */
    ri_float_t   *texcoords_unshared;                       /* st per index */
    unsigned int  ntexcoords;
} ri_geom_t;

ri_geom_t *ri_geom_new(void);
void       ri_geom_free(ri_geom_t *geom);
void       ri_geom_add_positions(ri_geom_t *geom, unsigned int npositions, const ri_vector_t *positions);
void       ri_geom_add_normals(ri_geom_t *geom, unsigned int nnormals, const ri_vector_t *normals);
void       ri_geom_add_indices(ri_geom_t *geom, unsigned int nindices, const unsigned int *indices);
void       ri_geom_add_tangents(ri_geom_t *geom, unsigned int ntangents, const ri_vector_t *tangents);
void       ri_geom_add_binormals(ri_geom_t *geom, unsigned int nbinormals, const ri_vector_t *binormals);
void       ri_geom_add_colors(ri_geom_t *geom, unsigned int ncolors, const ri_vector_t *colors);
void       ri_geom_add_texcoords(ri_geom_t *geom, unsigned int ntexcoords, const ri_float_t *texcoords);
void       ri_geom_add_texcoords_unshared(ri_geom_t *geom, unsigned int ntexcoords, const ri_float_t *texcoords);

/* ---- ray / hit state (ray.h:22-68, intersection_state.h:34-61) --------- */
typedef struct _ri_ray_t {
    ri_vector_t org;           /* [in]  position                              */
    ri_vector_t dir;           /* [in]  direction, need not be normalised     */
    float       t;             /* reset to 0 by ri_raytrace (raytrace.c:49)   */
    int         dir_sign[3];   /* [out] scratch the reference writes (bvh.c:473-497) */
    ri_vector_t invdir;        /* [out] idem                                  */
    int         thread_num;
} ri_ray_t;

typedef struct _ri_intersection_state_t {
    ri_vector_t  P, Ng, Ns, E, I;
    double       t;
    char         inside;
    ri_geom_t   *geom;         /* borrowed pointer to the hit geometry        */
    uint32_t     index;        /* 3*i offset into geom->indices (bvh.c:1813)  */
    ri_vector_t  color, tangent, binormal, stqr;
    ri_float_t   u, v;
} ri_intersection_state_t;

void ri_intersection_state_build(ri_intersection_state_t *state_inout,
                                 const ri_vector_t eye, const ri_vector_t dir);

/* ---- accelerator plugin (accel.h:24-75) --------------------------------- */
typedef void *(*accel_build_func)(const void *data /* const ri_scene_t* */);
typedef void  (*accel_free_func)(void *accel);
typedef int   (*accel_intersect_func)(void *accel, ri_ray_t *ray,
                                      ri_intersection_state_t *state, void *user);

typedef struct _ri_accel_t {
    accel_build_func     build;
    accel_free_func      free;
    accel_intersect_func intersect;
    void                *data;
} ri_accel_t;

ri_accel_t *ri_accel_new(void);
void        ri_accel_free(ri_accel_t *accel);
/* 0 on success, -1 on unknown/unavailable method (accel.c:102-106) */
int         ri_accel_bind(ri_accel_t *accel, int method);

/* the HIP implementation of the three vtable entries */
void *ri_hipbvh_build(const void *scene);
void  ri_hipbvh_free(void *accel);
int   ri_hipbvh_intersect(void *accel, ri_ray_t *ray, ri_intersection_state_t *state, void *user);

/* ---- scene / render (scene.h:31-96, render.h:104-105) ------------------- */
typedef struct _ri_scene_t {
    ri_geom_t  **geom_list;  unsigned int ngeoms;     /* geoms in list order */
    ri_accel_t  *accel;
} ri_scene_t;

ri_scene_t *ri_scene_new(void);
void        ri_scene_free(ri_scene_t *scene);
void        ri_scene_add_geom(ri_scene_t *scene, const ri_geom_t *geom);
int         ri_scene_build_accel(ri_scene_t *scene);     /* 0 / -1 (scene.c:153-167) */

typedef struct _ri_render_t {
    ri_scene_t *scene;
    struct { uint64_t nrays; } stat;                     /* raytrace.c:43 */
    int         device;                                  /* HIP device for RI_ACCEL_HIP */
} ri_render_t;

void         ri_render_init(void);
ri_render_t *ri_render_get(void);
void         ri_render_free(void);

/* ---- queries ------------------------------------------------------------- */
/* 1 hit / 0 miss; state_out written only on a hit (raytrace.c:56-66) */
int ri_raytrace(ri_render_t *render, ri_ray_t *ray, ri_intersection_state_t *state_out);

/* n rays in one device launch; hit[i] in {0,1}; states[i] written on hit exactly as
 * ri_raytrace would.  Returns number of hits, -1 on error. */
long ri_raytrace_batch(ri_render_t *render, size_t n, ri_ray_t *rays,
                       ri_intersection_state_t *states, int *hit);

/* SoA batch straight on the accelerator: mode 0 closest (prim,t,u,v), 1 any (occluded) */
int ri_accel_intersect_batch(void *accel, size_t n, const double *org_xyz, const double *dir_xyz,
                             uint32_t *prim, double *t, double *u, double *v,
                             uint8_t *occluded, int mode);
/* primitive id -> (geom, index), the pair state->geom/state->index carry */
int ri_accel_prim_lookup(void *accel, uint32_t prim, ri_geom_t **geom, uint32_t *index);

/* tile-level entry point: one w x h tile of the ambient-occlusion frame rendered on the device --
 * camera rays (ri_camera_get_pos_and_dir, src/ri/camera.c:248-318), closest hits, hit epilogue, gather_nsamples
 * cosine-stratified AO rays per hit (ambientocclusion.c:42-151), radiance (N - occluded) / N, box filter over
 * pixel_samples^2 sub-samples -- into HOST memory: rgb = h rows of w RGB float triples in image orientation
 * (bucket_write's y flip applied, render.c:962-964).  cam: the members of ri_camera_t the ray generator reads.
 * Returns 0 / -1.  Several GPUs from one process: lh_multi_* in lucille_hip.h. */
typedef struct _ri_tile_camera_t {
    int    width, height;        /* Format */
    int    rh;                   /* Orientation "rh": z flipped */
    int    ortho;                /* Projection "orthographic" */
    double flength;              /* 1 / tan(fov / 2), camera.c:219 */
    double cam2world[16];        /* row-vector convention, vector.h:182-210 */
} ri_tile_camera_t;
int ri_render_tile_ao(void *accel, const ri_tile_camera_t *camera, int x0, int y0, int w, int h,
                      int pixel_samples, int gather_nsamples, uint64_t seed, float *rgb);

/* ---- BVH extras of the boundary (bvh.h:194-227) ---------------------------- */

/* beam = frustum of 4 corner rays with a common origin: ri_beam_t member for member as beam.h:45-84 declares it.
 * ri_beam_set fills what the reference's fills (beam.c:331-465); the device queries read org, dir, normal,
 * dominant_axis and dirsign of the beam they are handed (lh_beam_set_t, lucille_hip.h) -- nothing is recomputed
 * from the caller's un-normalised directions. */
#define RI_BEAM_MISS_COMPLETELY 0     /* beam.h:27-29 */
#define RI_BEAM_HIT_COMPLETELY  1
#define RI_BEAM_HIT_PARTIALLY   2

typedef struct _ri_beam_t {
    ri_vector_t org;
    ri_vector_t dir[4];               /* P[i] - org, P[i] on the axis-aligned plane at distance d (beam.c:412-432) */
    ri_vector_t length;               /* side length (xyz); never set by ri_beam_set */
    ri_float_t  d;                    /* distance to the axis-aligned plane: 1024 */
    ri_float_t  t_max;                /* RI_INFINITY */
    int         is_tetrahedron;
    ri_vector_t invdir[4];
    int         dominant_axis;
    int         dirsign[3];
    ri_vector_t normal[4];
    struct _ri_beam_t *children;      /* beam.h:76-80: subdivided beams; unused by the compiled reference */
    int         nchildren;
} ri_beam_t;

/* 0, or -1 when the corner directions straddle an octant (beam.c:352-376) */
int ri_beam_set(ri_beam_t *beam, ri_vector_t org, ri_vector_t dir[4]);

/* ri_bvh_intersect_beam_visibility (bvh.c:612-667): RI_BEAM_* class, `user` ignored */
int  ri_hipbvh_intersect_beam_visibility(void *accel, ri_beam_t *beam, void *user);
/* n beams in one launch; result[i] = RI_BEAM_* or -1 where ri_beam_set refuses the beam */
int  ri_hipbvh_intersect_beam_visibility_batch(void *accel, size_t n, const double *org_xyz,
                                               const double *corner_dirs_xyz, int32_t *result);
/* ri_bvh_diag_t (bvh.h:103-110): what ri_bvh_intersect zeroes and fills through `user` (bvh.c:451-456).  ri_hipbvh_intersect
 * fills it with the numbers of ITS walk over ITS tree for that ray: 4-wide node visits, leaf visits, and -- where the reference
 * counts one "triangle isect" per leaf (bvh.c:827) -- the triangle records that went through the filter. */
typedef struct _ri_bvh_diag_t {
    uint32_t ninner_node_traversals;
    uint32_t nleaf_node_traversals;
    uint32_t ntriangle_isects;
} ri_bvh_diag_t;

/* ---- the beam-raster path (raster.h:24-84, bvh.h:203-206) ----
 * ri_raster_plane_t as raster.h:24-57 declares it; _new / _setup / _free as raster.c:24-160 (setup allocates and zeroes the
 * five arrays and computes `offset`, the lower-left corner in NDC).  ri_hipbvh_intersect_beam = ri_bvh_intersect_beam
 * (bvh.c:544-609): returns 0 like the reference; afterwards raster_out->t holds what the reference's path leaves there (its
 * quirks included: lucille_hip.h "the beam-raster path"); u, v, geom, index are never written (the reference does not
 * either); `user` is ignored.  Beams whose footprint leaves the window are cut to it (the reference writes out of bounds). */
typedef struct _ri_raster_plane_t {
    ri_float_t  *t, *u, *v;           /* [width * height] */
    ri_geom_t  **geom;
    uint32_t    *index;
    int          width, height;
    ri_vector_t  frame[3];            /* du dv dw */
    ri_vector_t  corner;              /* lower-left of the raster plane in 3-D */
    ri_vector_t  org;                 /* eye */
    ri_float_t   fov;                 /* degrees */
    ri_float_t   offset[2];
    ri_float_t   scale[2];
} ri_raster_plane_t;

ri_raster_plane_t *ri_raster_plane_new(void);
int  ri_raster_plane_setup(ri_raster_plane_t *plane, int width, int height, ri_vector_t frame[3], ri_vector_t corner,
                           ri_vector_t org, ri_float_t fov);
int  ri_raster_plane_free(ri_raster_plane_t *plane);
int  ri_hipbvh_intersect_beam(void *accel, ri_beam_t *beam, ri_raster_plane_t *raster_out, void *user);
/* n beams over windows of one size in one launch (plain arrays; lh_accel_beam_raster_host in lucille_hip.h has the details):
 * corner_dirs n x 4 x 3, corners n x 3, frame9 = du dv dw, t_out n x height x width, status n (0 traced, 1 nothing done,
 * -1 refused by ri_beam_set), flags n x 4 or NULL.  Returns 0 / -1. */
int  ri_hipbvh_intersect_beam_batch(void *accel, size_t n, const double *org_xyz, const double *corner_dirs_xyz,
                                    const double *corners_xyz, int width, int height, const double *frame9,
                                    const double *eye_xyz, double fov, double *t_out, int32_t *status, uint64_t *flags);

/* ri_bvh_invalidate_cache (bvh.c:389-428) frees the lazily built per-leaf 2-D triangle caches of
 * the beam-raster path; this accelerator keeps none (every beam projects with its own origin, i.e. it
 * always behaves as the reference does right after this call): a no-op kept for source compatibility. */
void ri_hipbvh_invalidate_cache(void *accel);

/* ri_bvh_clear_stat_traversal / ri_bvh_report_stat_traversal (bvh.c:669-706).  The reference
 * fills its globals only when compiled with -DRI_BVH_TRACE_STATISTICS; here the switch is a
 * run-time one (or the environment variable RI_BVH_TRACE_STATISTICS=1 at build time of the accel). */
void ri_hipbvh_trace_statistics(int enable);
void ri_hipbvh_clear_stat_traversal(void);
void ri_hipbvh_report_stat_traversal(void);
/* the same totals as numbers: nrays, node visits, filter tests, fp64 tests, hits */
void ri_hipbvh_get_stat_traversal(uint64_t out[5]);

#ifdef __cplusplus
}
#endif
#endif
