/*
 * lucille_oracle.h -- CPU restatement of lucille's BVH build / traversal /
 * ray-triangle hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (lucille_amd, the
 * C-ABI library liblucille_hip.so) never links, loads or calls anything here.
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit against the
 * compiled reference (oracle/_ref, built from /root/reference by
 * oracle/Makefile) in tests/test_oracle_vs_ref.py, and against the golden
 * fixtures the compiled reference produced (tests/golden/, generator
 * tests/golden/make_golden.py) in tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the lucille source tree).
 */
#ifndef LUCILLE_ORACLE_H
#define LUCILLE_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LO_MISS 0xFFFFFFFFu

typedef struct lo_scene lo_scene_t;

/* per-batch traversal counters, same definitions as ri_bvh_stat_traversal_t
 * (src/render/bvh.h:117-129, incremented at bvh.c:460,829-831,845,1129,1149) */
typedef struct lo_counters {
    uint64_t ninner_node_traversals;
    uint64_t nleaf_node_traversals;
    uint64_t ntested_triangles;
    uint64_t nactually_hit_triangles;
    uint64_t nrays;
} lo_counters_t;

typedef struct lo_tree_stats {
    uint64_t ninner;
    uint64_t nleaf;
    uint64_t max_depth;
    uint64_t max_leaf_tris;
    uint64_t ntriangles;
} lo_tree_stats_t;

lo_scene_t *lo_scene_new(void);
void        lo_scene_free(lo_scene_t *scene);

/* One call == one ri_geom_t appended to scene->geom_list
 * (src/render/scene.c:144-148).  positions are xyz triples (the reference
 * stores double[4]; w is never read on this path). */
int lo_scene_add_mesh(lo_scene_t *scene, uint32_t npositions,
                      const double *positions_xyz, uint32_t nindices,
                      const uint32_t *indices);

/* ri_bvh_build (src/render/bvh.c:276-379).  leaf_size<=0 -> BVH_NTRIS_LEAF=16 */
int lo_scene_build(lo_scene_t *scene);

uint64_t lo_scene_ntriangles(const lo_scene_t *scene);
void     lo_scene_tree_stats(const lo_scene_t *scene, lo_tree_stats_t *out);
/* scene bbox after margin (bvh->bmin/bmax, bvh.c:325-340) */
void     lo_scene_bbox(const lo_scene_t *scene, double bmin[3], double bmax[3]);

/* Flattened triangle list in reference primID order (create_triangle_list,
 * bvh.c:1736-1826): 9 doubles per triangle (v0 v1 v2), plus (geom ordinal,
 * index=3*i) per triangle. */
void lo_scene_get_triangles(const lo_scene_t *scene, double *v9,
                            uint32_t *geom_ord, uint32_t *index);

/*
 * ri_bvh_intersect minus the ri_intersection_state_build epilogue
 * (bvh.c:430-542): closest hit over a ray batch.  prim = global primitive id
 * in create_triangle_list order, LO_MISS on miss; t=1e38,u=v=0 on miss.
 * nthreads>1 slices the batch contiguously over pthreads.
 * counters may be NULL.
 */
void lo_intersect_batch(const lo_scene_t *scene, size_t n,
                        const double *org_xyz, const double *dir_xyz,
                        uint32_t *prim, double *t, double *u, double *v,
                        lo_counters_t *counters, int nthreads);

/* brute force arg-min over ALL triangles with the triangle_isect arithmetic
 * (bvh.c:730-791); ties: last equal-t triangle in primID order wins (same
 * rule as within a reference leaf, bvh.c:780). */
void lo_brute_force_batch(const lo_scene_t *scene, size_t n,
                          const double *org_xyz, const double *dir_xyz,
                          uint32_t *prim, double *t, double *u, double *v,
                          int nthreads);

void lo_scene_leaf_order(const lo_scene_t *scene, uint32_t *leaf_prims, uint32_t *prim_leaf_first);

/* brute force: how many triangles hit with exactly t == t_ref[i] (>= 2: an
 * exact-t tie, where the reference's winner is traversal-order dependent) */
void lo_count_equal_t_batch(const lo_scene_t *scene, size_t n, const double *org_xyz,
                            const double *dir_xyz, const double *t_ref, uint32_t *count);

/* the S-soup generator of SURVEY.md Appendix C (xorshift64 13/7/17).
 * state is in/out so the ray stream continues after the triangles. */
void lo_soup_triangles(uint64_t *state, uint32_t ntri, double half_extent,
                       double *positions_xyz /* 9*ntri */,
                       uint32_t *indices /* 3*ntri */);
void lo_soup_rays(uint64_t *state, size_t nrays, double *org_xyz,
                  double *dir_xyz);

/* optional per-vertex normals / two_side flag of mesh `mesh` (geom->normals,
 * geom->two_side; consumed by ri_intersection_state_build) */
int lo_scene_set_normals(lo_scene_t *scene, uint32_t mesh, const double *normals_xyz, int two_side);

/* ------------------------------------------------------------------ */
/* callers on either side of the ray query (lucille_oracle_ao.c)       */
/* ------------------------------------------------------------------ */

typedef struct lo_camera {
    int    width, height;
    int    rh;               /* Orientation "rh" => 1 (camera->is_rh)        */
    int    ortho;      /* camera_projection == RI_ORTHOGRAPHIC (camera.c:276,285-301) */
    double flength;          /* 1/tan(fov/2), camera.c:219                   */
    double cam2world[16];    /* camera_to_world, row-major, row-vector conv. */
} lo_camera_t;

void lo_camera_ray(const lo_camera_t *cam, double x, double y, double org[3], double dir[3]);
void lo_subpixel_jitter(int xs, int ys, int xsamples, int ysamples, double jitter[2]);
int  lo_bucket_order(int width, int height, int bucket_size, unsigned int *out_xy);
void lo_ortho_basis(double basis[3][3], const double n[3]);
void lo_state_build(const lo_scene_t *scene, uint32_t prim, double t, double u, double v,
                    const double org[3], const double dir[3],
                    double P[3], double Ng[3], double Ns[3], int *inside);
int  lo_scene_set_attribute(lo_scene_t *scene, uint32_t mesh, int kind, const double *data);
void lo_state_build_full(const lo_scene_t *scene, uint32_t prim, double t, double u, double v,
                         const double org[3], const double dir[3], double state24[24]);
void lo_state_batch(const lo_scene_t *scene, size_t n, const double *org_xyz, const double *dir_xyz, uint32_t *prim, double *state24);
void lo_ao_rays(const double P[3], const double Ns[3], uint32_t ntheta, uint32_t nphi,
                const double *rnd, double *org_xyz, double *dir_xyz);
void lo_mt_stream(unsigned long seed, size_t n, double *out);
size_t lo_render_ao(const lo_scene_t *scene, const lo_camera_t *cam, int xsamples, int ysamples,
                    int gather_nsamples, int bucket_size, float *image,
                    double *rec_org, double *rec_dir, uint32_t *rec_prim, double *rec_t,
                    double *rec_u, double *rec_v, size_t rec_cap);

/* the path-traced tile (lucille_oracle_pt.c; transport arithmetic UNPINNED, see that file's header): spp samples
 * s0 .. s0 + spp - 1 of spp_total per pixel, rgb[h][w][3] += their mean; materials10 = ten floats (kd, ks, kt, ior) per
 * mesh, or override10 for every mesh; returns the rays traced */
double   lo_pt_rnd(uint64_t key);
uint64_t lo_render_pt(const lo_scene_t *scene, const lo_camera_t *cam, int x0, int y0, int w, int h, int s0, int spp, int spp_total,
                      int max_vertices, const uint32_t *prim_mesh, const float *materials10, const float *override10,
                      const float env_rgb[3], const float *env_map, int env_w, int env_h, int ref_weights, uint64_t seed,
                      float *rgb, uint16_t *path_rays, uint64_t *max_rays_on_a_path);

/* src/transport/pathtrace.c AS WRITTEN (lucille_oracle_ptref.c: restated from the reference's text alone, its own MT19937 in
 * the text's call order, the connect step, BRDF values without cosine / pdf, no roulette compensation): the tile (x0, y0, w, h)
 * of the frame, nsamples paths per pixel, rgb[h][w][3] floats top row first.  Transport parity-UNPINNED (the file is dead code);
 * exists so that the product's departures from the text are numbers in tests/test_oracle_ptref.py.  Returns the rays traced. */
uint64_t lo_render_ptref(const lo_scene_t *scene, const lo_camera_t *cam, int x0, int y0, int w, int h, int nsamples,
                         int max_vertices, const uint32_t *prim_mesh, const float *materials10, const float *override10,
                         const float env_rgb[3], const float *env_map, int env_w, int env_h, unsigned long mt_seed, float *rgb);

/* beam (frustum) visibility: ri_beam_set + ri_bvh_intersect_beam_visibility
 * (lucille_oracle_beam.c).  dirs_xyz: n x 4 corner directions.  result: 0 miss, 1 hit
 * completely, 2 hit partially (beam.h:27-29), -1 where ri_beam_set returns -1 */
void lo_beam_visibility_batch(const lo_scene_t *scene, size_t n, const double *org_xyz,
                              const double *dirs_xyz, int32_t *result);

/* The beam-raster path: ri_beam_set + ri_raster_plane_setup + ri_bvh_intersect_beam (bvh.c:544-609) for one beam, restated as
 * the reference behaves (lucille_oracle_beam.c has the list of its quirks).  dirs: 4 x 3; frame9: du dv dw; t_out: width *
 * height doubles.  -1: ri_beam_set refuses the beam.  flags_out[4] (may be NULL): pixel tests outside the raster window (heap
 * corruption in the reference), assert(t >= 0) / assert(outer_len < 8) failures (aborts in the reference), triangles rasterised. */
int lo_beam_raster(const lo_scene_t *scene, const double *org, const double *dirs, int width, int height,
                   const double *frame9, const double *corner, const double *eye, double fov, double *t_out,
                   uint64_t *flags_out);

#ifdef __cplusplus
}
#endif

#endif
