/*
 * lucille_hip.h -- C ABI of liblucille_hip.so: the MI355X (gfx950) accelerator
 * for lucille's ray-query hot path.  Plain C, plain pointers and sizes; no
 * C++/torch types cross this boundary.  Every function returns 0 on success
 * and -1 on failure (lh_last_error() holds a message); nothing throws.
 *
 * Each entry point names the reference interface it stands in for (paths are
 * relative to the lucille source tree).  The reference-side glue a lucille
 * maintainer adds (a new RI_ACCEL_HIP case in ri_accel_bind that forwards to
 * these functions) is shown in INTEGRATION.md and lives, compilable against
 * the reference's own headers, in integration/ri_accel_hip.c.
 */
#ifndef LUCILLE_HIP_H
#define LUCILLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LH_MISS      0xFFFFFFFFu     /* prim id of a miss                     */
#define LH_INFINITY  1.0e38          /* t of a miss == RI_INFINITY (include/ri.h:47) */

/* query modes */
#define LH_MODE_CLOSEST 0            /* ri_bvh_intersect semantics            */
#define LH_MODE_ANY     1            /* boolean occlusion: the value calculate_occlusion
                                        actually consumes (ambientocclusion.c:123-129) */

/* kernel variants: LH_VARIANT_DEFAULT picks the tuned walk (= 4); 0 is the textbook walk kept as in-process reference */
#define LH_VARIANT_DEFAULT (-1)

typedef struct lh_accel lh_accel_t;  /* opaque: host BVH + device SoA copies  */
typedef struct lh_rib_scene lh_rib_scene_t;   /* a parsed RIB (below)         */

typedef struct lh_accel_info {
    uint32_t ntriangles;
    uint32_t nnodes;
    uint32_t nleaves;
    uint32_t max_depth;
    uint64_t device_bytes;           /* HBM footprint of the scene: the node format the default
                                        kernel walks + triangles + the reference-order tree;
                                        other node formats are uploaded when a variant first asks */
    double   build_seconds;          /* host build of the traversal tree      */
    double   upload_seconds;
    int      device;
    double   ref_build_seconds;      /* host build of the reference-order tree (lucille's own tree,
                                        kept for ties / fragile hits / beams): part of every commit */
    uint32_t nnodes_traversal;       /* 4-wide nodes the default kernel walks  */
    uint32_t ntriangles_in_tree;     /* ntriangles minus the zero-area triangles the reference can never report (two equal
                                        vertices: its determinant test, bvh.c:754, rejects them for every ray): those keep their
                                        primitive ids and stay in lucille's own tree, but no leaf of the traversal tree holds them */
} lh_accel_info_t;

/* ---- runtime ----------------------------------------------------------- */
int         lh_device_count(void);
const char *lh_last_error(void);

/* ---- accelerator lifetime: accel_build_func / accel_free_func ----------
 * reference: src/render/accel.h:24-28; ri_bvh_build src/render/bvh.c:276-379;
 * ri_bvh_free bvh.c:381-387.  The reference's build() walks scene->geom_list;
 * across the C ABI the same walk is three calls: create, add_mesh per
 * ri_geom_t in list order (this fixes primitive ids exactly as
 * create_triangle_list does, bvh.c:1736-1826), commit. */
int  lh_accel_create(lh_accel_t **out, int device);
/* positions: first double of vertex 0; stride_bytes between vertices
 * (32 for lucille's ri_vector_t = double[4], 24 for packed xyz).  The data is
 * copied (the reference copies too, bvh.c:320-321). */
int  lh_accel_add_mesh(lh_accel_t *accel, uint32_t npositions, const double *positions,
                       size_t stride_bytes, uint32_t nindices, const uint32_t *indices);
/* builds the BVH on the host (build_threads <= 0: all cores) and uploads the
 * SoA scene to the device.  An empty scene commits to an always-miss accel
 * (bvh.c:311-315,446-449).  Fails (-1, lh_last_error) on an out-of-range vertex index, on a NaN /
 * infinite / > 1e30 vertex coordinate (the fp32 filter cannot bound it) and on >= 2^29 triangles.
 * Every entry point taking an accelerator holds its lock: calls from several threads are safe and
 * serialised (lucille's render threads call accel->intersect concurrently, render.c:1043-1105). */
int  lh_accel_commit(lh_accel_t *accel, int build_threads);
/* build_threads == LH_BUILD_ON_DEVICE (or LH_BUILD=device in the environment): both trees are built on the GPU -- the
 * traversal tree (LBVH + SAH -> the same 4-wide nodes) and lucille's own tree (ri_bvh_build, bvh.c:276-379, restated level by
 * level, bit for bit: exact-t tie winners, fragile hits and beams depend on it): a fraction of a second instead of seconds;
 * lucille re-builds its accelerator in every ri_scene_setup (scene.c:84-98).  LH_REF_BUILD=host leaves lucille's own tree to
 * a background host thread as in round 2: queries are exact by default (the first one waits for it, lh_accel_wait_exact does so
 * explicitly); lh_accel_set_param(accel, "fast_start", 1) (or LH_FAST_START=1) lets them run before it is attached, exact-t
 * ties resolving to the larger primitive id until then.  Hit records are otherwise independent of the trees. */
#define LH_BUILD_ON_DEVICE (-2)
/* build_threads == LH_BUILD_ON_HOST (or LH_BUILD=host), or a thread count > 0: the host builders (binned SAH for traversal,
 * lucille's own tree beside it on the same thread pool): the better traversal tree (frames ~2 % faster), seconds for tens of
 * millions of triangles.  build_threads == 0 lets the library choose: the device builders from LH_AUTO_DEVICE_TRIANGLES
 * triangles on (what `lsh_hip --build auto` always did: lucille sets its scene up in front of every frame), the host builders
 * below; an automatic device build that cannot be done (memory, a degenerate tree) is redone on the host. */
#define LH_BUILD_ON_HOST (-3)
#define LH_AUTO_DEVICE_TRIANGLES 1000000ull
int  lh_accel_wait_exact(lh_accel_t *accel);
/* lucille's own tree as the kernels read it, for inspection (tests compare the device-built tree with the host-built one):
 * *nnodes nodes of 128 bytes (lh_refbvh.h: two child boxes of 6 doubles, child[2], axis, is_leaf, first, count, parent, depth;
 * root = 0) and the ntriangles primitive ids in leaf order.  Either pointer may be NULL.  Fails if the tree is not attached. */
int  lh_accel_ref_tree(lh_accel_t *accel, uint32_t *nnodes, void *nodes_out, uint32_t *leaf_prims_out);
void lh_accel_destroy(lh_accel_t *accel);
int  lh_accel_info(const lh_accel_t *accel, lh_accel_info_t *out);

/* primitive id -> (mesh ordinal in add order, index = 3*i into that mesh's
 * index list): what state->geom / state->index carry in the reference
 * (bvh.c:855-860), so the host can run ri_intersection_state_build. */
int  lh_accel_prim_lookup(const lh_accel_t *accel, uint32_t prim, uint32_t *mesh,
                          uint32_t *index);

/* ---- queries: accel_intersect_func (src/render/accel.h:30-34) ----------
 * Rays are fp64 xyz triples exactly as ri_ray_t.org/.dir (src/render/ray.h:
 * 24-25); dir need not be normalised, t is in units of |dir|.
 * Closest mode writes prim (LH_MISS on miss), t (1e38 on miss), u, v as the
 * reference does (bvh.c:850-861,1187).  Any mode writes occluded[i] in {0,1}.
 * Unused outputs may be NULL. */

/* one synchronous ray: the reference's calling convention, kept correct (a
 * batch of one through the same kernel). Returns 1 hit / 0 miss / -1 error. */
int  lh_accel_intersect1(lh_accel_t *accel, const double org[3], const double dir[3],
                         uint32_t *prim, double *t, double *u, double *v);
/* Per-ray traversal diagnostics: what ri_bvh_intersect reports through its `user` argument (a ri_bvh_diag_t, bvh.h:103-110,
 * bvh.c:451-456) for THIS build's tree.  diag: n x 4 u32 -- 4-wide node visits, leaf visits, triangle records through the fp32
 * filter, fp64 tests -- of the sequential walk (nearest child first, as tests/cpu_model walks); records as every other path's.
 * With lh_accel_trace_statistics on, the launch's totals are the sums of these rows. */
int  lh_accel_intersect_diag_host(lh_accel_t *accel, size_t n, const double *org_xyz, const double *dir_xyz,
                                  uint32_t *prim_id, double *t, double *u, double *v, uint32_t *diag);
/* ... for rays resident on the device, closest- or any-hit (mode: LH_MODE_*); d_diag: n x 4 u32 on the device, hit records dropped */
int  lh_accel_intersect_diag_device(lh_accel_t *accel, size_t n, const void *d_org_xyz, const void *d_dir_xyz, int mode,
                                    void *d_diag, void *stream);
/* Concurrent lh_accel_intersect1 callers (lucille's render threads) are coalesced into one launch per batch of callers
 * (lh_query.hip "flat combining"); set_param("combine", 0) / LH_COMBINE=0 restores one launch per call.
 * out[0] = launches, out[1] = rays they carried since the last clear. */
int  lh_accel_combine_statistics(lh_accel_t *accel, uint64_t out[2], int clear);

/* host-resident batch (the batch form of accel_intersect_func, accel.h:30-34: rays and records in the caller's arrays).  Below 2 M rays:
 * copy up, one launch, copy down on the accel's own stream.  From 2 M rays: chunks through a ring of pinned staging blocks -- the rays of
 * chunk k + 1 cross the link while chunk k is traced and chunk k - 1's records go back (lh_query.hip; 860 Mrays/s closest hit over PCIe,
 * profiles/r06_hostpath.txt).  Outputs the caller does not need may be NULL.  Synchronous: the arrays are complete on return */
int  lh_accel_intersect_host(lh_accel_t *accel, size_t n, const double *org_xyz,
                             const double *dir_xyz, uint32_t *prim, double *t, double *u,
                             double *v, uint8_t *occluded, int mode);

/* device-resident batch on a caller stream (hipStream_t as void*; NULL =
 * default stream).  Asynchronous: returns after enqueue. */
int  lh_accel_intersect_device(lh_accel_t *accel, size_t n, const void *d_org_xyz,
                               const void *d_dir_xyz, void *d_prim, void *d_t, void *d_u,
                               void *d_v, void *d_occluded, int mode, int variant,
                               void *stream);

/* same launch with traversal statistics: counters[4] (host) receives
 * {inner-node visits, triangle tests, fp64 resolves, rays}; synchronous. */
int  lh_accel_intersect_device_counted(lh_accel_t *accel, size_t n, const void *d_org_xyz,
                                       const void *d_dir_xyz, void *d_prim, void *d_t,
                                       void *d_u, void *d_v, void *d_occluded, int mode,
                                       int variant, uint64_t counters[4]);
/* rays of the last counted launch that were finished outside the main kernel: the reference-order walk (exact-t ties,
 * fragile hits) and the private-stack walk for rays whose LDS stack column would have overflowed */
uint64_t lh_accel_last_retraced(const lh_accel_t *accel);
/* bytes of the node record the ray-dump entry points (lh_accel_intersect_host / _device) walk on this scene: 64 (4-wide
 * 16-bit-grid node), or 128 (8-wide, one cache line) when the scene's hot set does not fit the 256 MiB Infinity Cache --
 * there every record fetched costs a 128-byte line of HBM traffic.  lh_accel_set_param("wide8", -1 auto / 0 / 1). */
int lh_accel_dump_node_bytes(const lh_accel_t *accel);

/* ---- traversal statistics: ri_bvh_clear_stat_traversal / ri_bvh_report_stat_traversal ----
 * reference: src/render/bvh.c:669-706 (globals filled under -DRI_BVH_TRACE_STATISTICS,
 * :681-706, 829-831, 1149-1151).  While enabled, every lh_accel_intersect1/_host call and every
 * batch of the tile pipelines (lh_render_ao_tile / _bands / _frame_host, lh_render_pt_tile*) runs
 * the counting kernel variant and accumulates: counters[0] node visits (one per 4-wide node
 * record fetched), [1] triangles put through the fp32 filter, [2] triangles re-tested in
 * fp64, [3] rays, [4] rays that hit (AO pipeline: camera-ray hits + occluded AO rays; not counted by
 * the path tracer).  Counts describe THIS build's tree, not lucille's. */
int  lh_accel_trace_statistics(lh_accel_t *accel, int enable);
int  lh_accel_statistics(lh_accel_t *accel, uint64_t counters[5], int clear);
/* lane slots of the tile pipelines' counted launches: 64 per wave iteration that made a node step / ran a triangle pass / regrouped.
 * counters[0] / slots[0] = lane use of the node steps, counters[1] / slots[1] = of the triangle passes */
int  lh_accel_slot_statistics(lh_accel_t *accel, uint64_t slots[3], int clear);

/* number of persistent workgroups the persistent variants launch */
int  lh_accel_set_grid(lh_accel_t *accel, int blocks);
/* knobs by name: "grid", "min_active", "tri_batch", "ray_chunk" (sweeps), "variant" (0: the textbook reference walk, 4: the
 * default), "ao_fused", "wide8" (-1 auto / 0 / 1), "fast_start", "combine", "top_nodes", "ao_group";
 * "stack_cap": LDS stack rows of the default walk -- 0 (default): a launch of 65 536 rays or more walks at most 34 CHECKED rows
 * (four workgroups per CU; a ray that would overrun them is finished by the cooperative walk), smaller launches up to 64
 * unchecked; 8 .. 64: that many at most (64 = rounds 1-3's unchecked rows; small values: tests of the overflow path);
 * "min_active" / "tri_batch": working lanes below which a wave regroups / lanes holding a parked leaf from which it runs a triangle
 * pass -- 24 / 12 for the tile pipelines; left alone, ray dumps use 24 / 8 over the 4-wide nodes and 40 / 20 over the 8-wide ones,
 * once either is set here (or by LH_MIN_ACTIVE / LH_TRI_BATCH) every launch uses the caller's pair;
 * "ray_budget" (wave iterations after which a ray leaves the persistent walk for the cooperative one; sets "dump_budget" and
 * "ao_budget" with it), "dump_budget" (ray dumps: 2048), "ao_budget" (the fused AO stage: 384; 0: "ray_budget") */
int  lh_accel_set_param(lh_accel_t *accel, const char *name, int value);

/* ---- tile rendering: the callers on either side of the query, on the device ----
 * reference: subsample / render_bucket / bucket_write (src/render/render.c:715-823,
 * 1107-1166, 919-983), ri_camera_get_pos_and_dir (src/ri/camera.c:248-318),
 * ri_intersection_state_build (src/render/intersection_state.c:99-248),
 * ri_transport_ambientocclusion + calculate_occlusion
 * (src/transport/ambientocclusion.c:42-151,332-415). */

typedef struct lh_camera {
    int    width, height;       /* camera->horizontal/vertical_resolution             */
    int    rh;                  /* Orientation "rh" (camera->is_rh)                   */
    int    ortho;               /* 1: Projection "orthographic" (the reference's default when no
                                 * "fov" is given, camera.c:100,428); 0: perspective        */
    double flength;             /* 1/tan(fov/2) (camera.c:219)                         */
    double cam2world[16];       /* camera->camera_to_world, row-major, row vectors     */
} lh_camera_t;

typedef struct lh_tile_stats {
    uint64_t primary_rays;      /* w*h*pixel_samples^2                                 */
    uint64_t primary_hits;
    uint64_t ao_rays;           /* primary_hits * floor(sqrt(gather_nsamples))^2       */
    uint64_t ao_occluded;
} lh_tile_stats_t;

/* optional per-vertex normals of mesh `mesh` (add order), before commit:
 * geom->normals / geom->two_side as ri_intersection_state_build reads them */
int  lh_accel_set_normals(lh_accel_t *accel, uint32_t mesh, const double *normals,
                          size_t stride_bytes, int two_side);

/* camera rays of the tile [x0,x0+w) x [y0,y0+h), pixel_samples^2 per pixel, written to
 * device arrays of w*h*pixel_samples^2 xyz triples (sample id = ((ly*w+lx)*ps+sy)*ps+sx) */
int  lh_render_primary_rays(lh_accel_t *accel, const lh_camera_t *cam, int x0, int y0, int w, int h,
                            int pixel_samples, void *d_org_xyz, void *d_dir_xyz, void *stream);

/* one AO tile entirely on the device: camera rays -> closest hit -> hit epilogue ->
 * gather_nsamples AO rays per hit -> any-hit -> radiance.  d_rgb: float[h][w][3] in image
 * orientation (row 0 = the tile's TOP row of the output image, as bucket_write flips y).
 * d_uniforms: NULL for the built-in counter-based RNG (seed), or 2 doubles per AO ray in
 * (hit slot, j, i) order -- how a caller replays the reference's MT19937 stream.
 * Synchronous on `stream` (one 8-byte read-back of the hit count). */
int  lh_render_ao_tile(lh_accel_t *accel, const lh_camera_t *cam, int x0, int y0, int w, int h,
                       int pixel_samples, int gather_nsamples, uint64_t seed,
                       const void *d_uniforms, void *d_rgb, lh_tile_stats_t *stats, void *stream);

/* nbands full-width bands of band_rows lines each (band b = frame lines band_y0[b] .. + band_rows, clipped at the frame)
 * as ONE device batch -- how a rank renders ALL of its interleaved shards of a frame (image space sharded
 * over the GPUs) with one set of kernel launches: every launch of the persistent traversal kernel ends with a drain
 * as long as its slowest ray (~1.7 ms on BASELINE config 5), paid once per batch instead of once per shard.
 * band_y0: nbands ints (HOST).  d_rgb: float[nbands][band_rows][width][3], every band in image orientation. */
int  lh_render_ao_bands(lh_accel_t *accel, const lh_camera_t *cam, int nbands, const int *band_y0, int band_rows,
                        int pixel_samples, int gather_nsamples, uint64_t seed, void *d_rgb, lh_tile_stats_t *stats,
                        void *stream);

/* the same tile for a plain-C host (the batched frame loop inside lucille, integration/ri_render_hip.c): rgb is
 * HOST memory (h rows of w RGB floats, image orientation); uniforms (HOST, may be NULL) as d_uniforms above --
 * nuniforms must cover the worst case 2 * floor(sqrt(gather_nsamples))^2 * w * h * pixel_samples^2; the tile
 * consumes the first 2 * N * stats->primary_hits of them. */
int  lh_render_ao_tile_host(lh_accel_t *accel, const lh_camera_t *cam, int x0, int y0, int w, int h,
                            int pixel_samples, int gather_nsamples, uint64_t seed, const double *uniforms,
                            size_t nuniforms, float *rgb, lh_tile_stats_t *stats);

/* one path-traced tile on the device (BASELINE config 4: the reference's pathtrace.c is dead
 * code; its documented structure -- camera sample, Russian roulette on the reflectance,
 * cosine-sampled diffuse bounces to a vertex limit, environment radiance on escape -- re-expressed
 * as closest-hit batches, every bounce through the same kernel as ri_raytrace).  Adds
 * spp_count samples (numbered spp_begin...) of spp_total to d_rgb (float[h][w][3], image
 * orientation; the caller zeroes it before the first pass).  kd: diffuse reflectance in (0,1];
 * env_rgb: constant environment radiance.  stats: rays traced / paths.
 * One pass holds at most 2^30 paths (w x h x spp_count) and at most 4096 samples of a pixel: a path's radiance goes into
 * its pixel's 64-bit fixed-point sums (32 fraction bits, a sample clamped to +-2^18, a NaN channel dropped), so a pixel
 * does not depend on the order its paths end in -- tiling, sharding and slot order leave the image bit for bit the same. */
typedef struct lh_pt_stats { uint64_t paths, rays, max_depth_reached; } lh_pt_stats_t;

int  lh_render_pt_tile(lh_accel_t *accel, const lh_camera_t *cam, int x0, int y0, int w, int h,
                       int spp_begin, int spp_count, int spp_total, int max_path_vertices,
                       float kd, const float env_rgb[3], uint64_t seed, void *d_rgb,
                       lh_pt_stats_t *stats, void *stream);

/* ---- path tracer with the reference's three reflection types (src/transport/pathtrace.c:189-314,407-537) ----
 * Per-mesh material = ri_material_t (src/render/material.h:21-30; defaults of ri_material_new: kd 1, ks 0, kt 0,
 * ior 1): diffuse / specular / transmission reflectances (their averages d, s, t drive the Russian roulette and the
 * choice of reflection type: d + s + t <= 1) and the index of refraction.  Environment = the light source of
 * light_sample / ri_texture_ibl_fetch (src/render/texture.c:238-276): an angular-map light probe (RGBA float rows,
 * bilinear), scaled by rgb; map NULL: the constant radiance rgb. */
typedef struct lh_material { float kd[3], ks[3], kt[3]; float ior; } lh_material_t;
typedef struct lh_environment { float rgb[3]; const float *map_rgba; int width, height; } lh_environment_t;
#define LH_ALL_MESHES 0xFFFFFFFFu
#define LH_PT_REFERENCE_WEIGHTS 1     /* throughput *= the reference's brdf() value (kd / pi, ks, kt) instead of the
                                         unbiased weight of the same sampling scheme */
int  lh_accel_set_material(lh_accel_t *accel, uint32_t mesh /* or LH_ALL_MESHES */, const lh_material_t *material);
int  lh_accel_set_environment(lh_accel_t *accel, const lh_environment_t *environment);   /* after commit; the map is copied; NULL: back to the default (constant white) */
/* as lh_render_pt_tile, with the accelerator's materials and environment */
int  lh_render_pt_tile2(lh_accel_t *accel, const lh_camera_t *cam, int x0, int y0, int w, int h,
                        int spp_begin, int spp_count, int spp_total, int max_path_vertices, int flags,
                        uint64_t seed, void *d_rgb, lh_pt_stats_t *stats, void *stream);

/* a rank's interleaved full-width bands of an image-space sharded frame (BASELINE config 4 on N GPUs) as ONE pass: nbands bands
 * of band_rows lines, band k starting at frame line y0_first + k * band_stride, all inside the frame.  d_rgb:
 * [nbands][band_rows][width][3] float32, every band in image orientation (lh_render_ao_bands' layout).  Uses the accelerator's
 * environment; override NULL: its per-mesh materials, else that material for every mesh.  Paths are keyed by (frame pixel,
 * sample, bounce): the bands of all ranks together are the frame lh_render_pt_tile2 renders, bit for bit. */
int  lh_render_pt_bands(lh_accel_t *accel, const lh_camera_t *cam, int y0_first, int band_rows, int band_stride, int nbands,
                        int spp_begin, int spp_count, int spp_total, int max_path_vertices, int flags,
                        const lh_material_t *override_material, uint64_t seed, void *d_rgb, lh_pt_stats_t *stats, void *stream);

/* ---- the whole hit epilogue: ri_intersection_state_build (src/render/intersection_state.c:99-248) ----
 * Optional per-vertex attributes of mesh `mesh`, before commit (geom.h:34-48): colours, tangents, binormals
 * (xyz, `count` = vertices), texture coordinates (s, t per vertex) or texcoords_unshared (s, t per INDEX: `count` =
 * indices).  lh_accel_state_build_*: for n rays and their hit records, LH_STATE_DOUBLES per ray --
 * P[3] Ng[3] Ns[3] tangent[3] binormal[3] color[3] st[2] I[3] inside -- in the reference's operation order (fp64,
 * no contraction); records of misses are left as they are (device) / zero (host). */
#define LH_ATTR_COLOR 0
#define LH_ATTR_TANGENT 1
#define LH_ATTR_BINORMAL 2
#define LH_ATTR_TEXCOORD 3
#define LH_ATTR_TEXCOORD_UNSHARED 4
#define LH_STATE_DOUBLES 24
int  lh_accel_set_attribute(lh_accel_t *accel, uint32_t mesh, int kind, const double *data, size_t stride_bytes, uint32_t count);
int  lh_accel_state_build_device(lh_accel_t *accel, size_t n, const void *d_org_xyz, const void *d_dir_xyz, const void *d_prim,
                                 const void *d_t, const void *d_u, const void *d_v, void *d_state, void *stream);
int  lh_accel_state_build_host(lh_accel_t *accel, size_t n, const double *org_xyz, const double *dir_xyz, const uint32_t *prim,
                               const double *t, const double *u, const double *v, double *state);

/* ---- the G GPUs of one node from ONE process ----
 * reference role: the bucket queue drained by render threads (src/render/render.c:1043-1207) and the compiled-out MPI
 * design "every rank renders, rank 0 owns the display" (render.c:468-514, src/base/parallel.c:62-232).  One host build,
 * replicated to every device; frames: a dynamic queue of tiles, one host thread per device, finished tile slabs copied
 * device-to-device (hipMemcpyPeerAsync, one xGMI link per peer) to device 0 = the display owner, which places them
 * (bucket_write's row order) and delivers the frame; ray dumps: contiguous slices.  `devices` may list a device more
 * than once (replicas on one GPU: how the sharded path is tested on a one-GPU box); NULL: devices 0 .. n-1
 * (ndevices <= 0: all).  device_seconds (ndevices doubles, may be NULL): busy time of each replica's tile loop. */
typedef struct lh_multi lh_multi_t;
int  lh_multi_create(lh_multi_t **out, int ndevices, const int *devices);
void lh_multi_destroy(lh_multi_t *multi);
int  lh_multi_ndevices(const lh_multi_t *multi);
lh_accel_t *lh_multi_accel(lh_multi_t *multi, int replica);          /* borrowed: queries on one replica */
int  lh_multi_add_mesh(lh_multi_t *multi, uint32_t npositions, const double *positions, size_t stride_bytes,
                       uint32_t nindices, const uint32_t *indices);
int  lh_multi_set_normals(lh_multi_t *multi, uint32_t mesh, const double *normals, size_t stride_bytes, int two_side);
int  lh_multi_add_rib_scene(lh_multi_t *multi, const lh_rib_scene_t *scene);
int  lh_multi_commit(lh_multi_t *multi, int build_threads);
int  lh_multi_set_material(lh_multi_t *multi, uint32_t mesh, const lh_material_t *material);
int  lh_multi_set_environment(lh_multi_t *multi, const lh_environment_t *environment);
int  lh_multi_intersect_host(lh_multi_t *multi, size_t n, const double *org_xyz, const double *dir_xyz, uint32_t *prim,
                             double *t, double *u, double *v, uint8_t *occluded, int mode);
int  lh_multi_render_ao_frame_host(lh_multi_t *multi, const lh_camera_t *cam, int pixel_samples, int gather_nsamples,
                                   uint64_t seed, int tile, float *rgb, lh_tile_stats_t *stats, double *device_seconds);
int  lh_multi_render_pt_frame_host(lh_multi_t *multi, const lh_camera_t *cam, int spp, int spp_chunk, int max_path_vertices,
                                   int flags, uint64_t seed, int tile, float *rgb, lh_pt_stats_t *stats, double *device_seconds);

/* ---- one process per GPU: lucille's compiled-out MPI layer over RCCL ----
 * reference: ri_parallel_init / _barrier / _bcast / _gather / _send / _recv (src/base/parallel.c:62-232) and the frame
 * protocol on top of it, "every rank renders, rank 0 owns the display" (src/render/render.c:468-514).  One lh_dist_t per
 * process = one rank = one GPU.  Scene load: ONE host build on rank 0, then ncclBroadcast of the flattened arrays into
 * every rank's HBM (lh_dist_broadcast_scene).  Frames: image space in interleaved bands, a rank's bands one device batch,
 * ONE exchange step -- ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd of the ranks' slabs to rank 0.  RCCL is loaded at
 * run time (librccl.so.1).  LH_DIST_SHM: the same interface over a POSIX shared-memory segment, for ranks that share a
 * device (RCCL refuses those): how the N > 1 path is tested on a one-GPU box.
 * Rendezvous: the caller distributes the 128-byte id of rank 0 (lh_dist_unique_id) by its own means (torch.distributed, MPI),
 * or names a file on a shared file system (lh_dist_init_file: a fresh path per job). */
typedef struct lh_dist lh_dist_t;
#define LH_DIST_ID_BYTES 128
#define LH_DIST_RCCL 0
#define LH_DIST_SHM  1
int  lh_dist_unique_id(void *id128);                                        /* rank 0: ncclGetUniqueId */
int  lh_dist_init(lh_dist_t **out, const void *id128, int rank, int world, int device, int transport);   /* ncclCommInitRank */
int  lh_dist_init_file(lh_dist_t **out, const char *rendezvous_path, int rank, int world, int device);
void lh_dist_destroy(lh_dist_t *dist);
int  lh_dist_rank(const lh_dist_t *dist);
int  lh_dist_world(const lh_dist_t *dist);
int  lh_dist_transport(const lh_dist_t *dist);
int  lh_dist_barrier(lh_dist_t *dist);
/* the ranks of ONE node meet in shared memory (microseconds; host only -- synchronise the device first): what brackets a timed
 * frame.  lh_dist_barrier goes through the transport (RCCL: a one-byte gather and broadcast) and works across nodes. */
int  lh_dist_host_barrier(lh_dist_t *dist);
/* device buffers; stream NULL: the communicator's own.  gather: d_recv (rank 0 only) holds world * bytes */
int  lh_dist_broadcast(lh_dist_t *dist, void *d_buf, size_t bytes, void *stream);
int  lh_dist_gather(lh_dist_t *dist, const void *d_send, size_t bytes, void *d_recv, void *stream);
/* hit records for the wire: n records (prim u32 | t | u | v f64, the arrays lh_accel_intersect_device wrote) -> n 16-byte records
 * {prim u32, t, u, v f32 rounded to nearest from the fp64 bits} in d_rec16 (device, 16-byte aligned).  A dump service that returns
 * every record to ONE host is bound by the xGMI links at 28 bytes a ray (8 ranks: 350 MB per peer against 5.4 ms of tracing);
 * 16 bytes keep the exchange behind the tracing.  The fp64 records stay with the rank that traced them; fp32 t / u / v are within
 * 6e-8 relative of them (north_star: 1e-5); a miss keeps prim 0xFFFFFFFF and t = 1e38.  No communicator needed. */
int  lh_dist_pack_records16(size_t n, const void *d_prim, const void *d_t, const void *d_u, const void *d_v, void *d_rec16, void *stream);
/* rank 0: a committed accelerator; the others: a fresh one (lh_accel_create), committed on return -- no build, no host
 * copy of the tree on those ranks (lh_accel_export and replicas of it are refused) */
int  lh_dist_broadcast_scene(lh_dist_t *dist, lh_accel_t *accel);
/* the AO frame sharded over the ranks: bands of band_rows lines (<= 0: 16), dealt out in serpentine order (groups of `world`
 * bands, even groups in rank order, odd groups reversed: a steady change of cost down the image cancels); rgb (rank 0 only): height
 * rows of width RGB floats, top row first; stats: the frame's totals on rank 0, the rank's own elsewhere */
int  lh_dist_render_ao_frame_host(lh_dist_t *dist, lh_accel_t *accel, const lh_camera_t *cam, int pixel_samples,
                                  int gather_nsamples, uint64_t seed, int band_rows, float *rgb, lh_tile_stats_t *stats);

/* device scratch of the last lh_render_ao_tile call (for tests / pipelines):
 * which: 0 primary org, 1 primary dir, 2 prim, 3 t, 4 u, 5 v, 6 slot_of_sample,
 *        7 hit records (12 doubles: AO origin, tangent, binormal, Ns), 8 AO org, 9 AO dir,
 *        10 AO occluded */
int  lh_render_scratch(lh_accel_t *accel, int which, void **d_ptr, size_t *count);

/* ---- beam (frustum) visibility: ri_beam_set + ri_bvh_intersect_beam_visibility ----
 * reference: src/render/beam.c:331-465, src/render/bvh.c:612-667 (+ :1997-2281, :2435-2542,
 * :2648-2746), constants src/render/beam.h:27-29.  A beam is a common origin and 4 corner
 * directions (n x 4 x 3 doubles).  result[i]: LH_BEAM_MISS_COMPLETELY / _HIT_COMPLETELY /
 * _HIT_PARTIALLY exactly as the reference returns them (the answer depends on ITS tree, so the
 * query runs on the reference-order tree kept beside the traversal tree), or LH_BEAM_INVALID
 * where ri_beam_set returns -1 (corner directions straddle an octant, beam.c:352-376). */
#define LH_BEAM_MISS_COMPLETELY 0
#define LH_BEAM_HIT_COMPLETELY  1
#define LH_BEAM_HIT_PARTIALLY   2
#define LH_BEAM_INVALID        (-1)

int  lh_accel_beam_visibility_host(lh_accel_t *accel, size_t n, const double *org_xyz,
                                   const double *corner_dirs_xyz, int32_t *result);
int  lh_accel_beam_visibility_device(lh_accel_t *accel, size_t n, const void *d_org_xyz,
                                     const void *d_corner_dirs_xyz, void *d_result, void *stream);

/* The same queries for beams the CALLER has already set up with lucille's own ri_beam_set: what
 * ri_bvh_intersect_beam_visibility(accel, ri_beam_t *, user) / ri_bvh_intersect_beam(accel, ri_beam_t *, plane, user) are
 * handed (src/render/bvh.h:203-221).  lh_beam_set_t = the members of ri_beam_t (src/render/beam.h:45-84) those two walks read:
 * org, the four directions SCALED onto the plane at d = 1024 (beam.c:412-432), their cross products (beam.c:446-452), the
 * dominant axis and the direction signs (beam.c:381-403) -- taken as they are, nothing recomputed, so the answer is the
 * reference's for the beam it holds (integration/ri_accel_hip.c copies them out of the ri_beam_t).  invdir, d, t_max,
 * is_tetrahedron and the child beams are not read by either walk (t_max stays RI_INFINITY, beam.c:344). */
typedef struct lh_beam_set {
    double  org[3];
    double  dir[4][3];
    double  normal[4][3];
    int32_t dominant_axis;
    int32_t dirsign[3];
} lh_beam_set_t;

int  lh_accel_beam_visibility_set_host(lh_accel_t *accel, size_t n, const lh_beam_set_t *beams, int32_t *result);

/* ---- the beam-raster path: ri_beam_set + ri_raster_plane_setup + ri_bvh_intersect_beam ----
 * reference: src/render/bvh.c:544-609 (-> :2547-2643, :2315-2426, :2751-2820), src/render/beam.c:469-730,
 * src/render/raster.c:42-147,166-435, src/render/triangle.c:8-68.  The reference never calls this path and left it
 * unfinished; what it COMPUTES is reproduced bit for bit, quirks included (lh_beam.hip lists them): after the call the raster
 * plane's t array holds, per pixel of the window, the ray parameter of the LAST rasterised triangle that covered the pixel
 * in the reference's traversal order, 0.0 elsewhere (u, v, geom, index of ri_raster_plane_t are never written by the
 * reference either).  Like beam visibility it runs on the reference-order tree.
 *   plane:   the raster window shared by the batch -- width x height pixels, frame = du dv dw (raster.h:38), eye =
 *            plane->org, fov in degrees;
 *   corner:  n x 3, the lower-left corner of every beam's window (plane->corner; the testbed shifts it per beam,
 *            simplerender.cpp:703-720);
 *   t_out:   n x height x width doubles; written for status 0 only;
 *   status:  0 traced; 1 nothing done (empty scene or the beam misses the scene box: the reference returns before it clears
 *            the plane, bvh.c:560-563,586-593); LH_BEAM_INVALID where ri_beam_set refuses the beam;
 *   flags:   n x 4 u64 or NULL -- what is UNDEFINED in the reference, reported instead of reproduced: [0] pixel tests outside
 *            the window (the reference writes plane->t[t * width + s] unchecked, raster.c:300-316; here the box is cut to the
 *            window), [1] / [2] the asserts of beam.c:142 / :626 would have fired, [3] triangles rasterised. */
typedef struct lh_raster_plane {
    int32_t width, height;
    double  frame[9];
    double  eye[3];
    double  fov;
} lh_raster_plane_t;

int  lh_accel_beam_raster_host(lh_accel_t *accel, size_t n, const double *org_xyz, const double *corner_dirs_xyz,
                               const double *corner_xyz, const lh_raster_plane_t *plane, double *t_out,
                               int32_t *status, uint64_t *flags);
int  lh_accel_beam_raster_device(lh_accel_t *accel, size_t n, const void *d_org_xyz, const void *d_corner_dirs_xyz,
                                 const void *d_corner_xyz, const lh_raster_plane_t *plane, void *d_t_out,
                                 void *d_status, void *d_flags, void *stream);
/* ... for beams already set up by the caller's ri_beam_set (lh_beam_set_t above); status is never LH_BEAM_INVALID */
int  lh_accel_beam_raster_set_host(lh_accel_t *accel, size_t n, const lh_beam_set_t *beams, const double *corner_xyz,
                                   const lh_raster_plane_t *plane, double *t_out, int32_t *status, uint64_t *flags);

/* whole AO frame into HOST memory: the tile loop of render_frame_controller + bucket_write
 * (src/render/render.c:1168-1207, 919-983) over lh_render_ao_tile.  rgb: height rows of width RGB
 * float triples, top row first (bucket_write's y flip applied) -- what the reference hands its
 * display driver.  tile = tile edge in pixels (0: the largest power of two <= 4096 whose per-tile scratch stays under ~6 GB).
 * stats may be NULL. */
int  lh_render_ao_frame_host(lh_accel_t *accel, const lh_camera_t *cam, int pixel_samples,
                             int gather_nsamples, uint64_t seed, int tile, float *rgb,
                             lh_tile_stats_t *stats);

/* ---- RIB-subset reader + Radiance .hdr writer: `lsh scene.rib` without flex/bison ----
 * reference: the RenderMan front end for the verbs its example scenes use --
 * src/ri/transform.c, context.c, attribute.c, camera.c:209-240,360-438, display.c:70-200,
 * option.c:430-560, src/render/polygon.c:39-262,495-640, src/base/matrix.c, quaternion.c; numbers
 * are C floats as in src/lsh/lexrib.l:213 / parserib.y:119.  The result is what lucille's renderer
 * is handed: the ri_geom_t list (world-space double[4] positions / normals, triangle indices,
 * two_side) in RIB order -- same global primitive ids -- and the camera.  Verbs outside the
 * ray-query path (shaders, lights, colours, quadrics ...) are skipped and counted. */
typedef struct lh_rib_info {
    uint32_t    nmeshes;
    uint64_t    ntriangles;
    lh_camera_t camera;           /* ri_camera_setup (camera.c:209-240) of the parsed state       */
    int         perspective;      /* 0: no Projection "perspective" "fov" seen (reference: ortho)  */
    float       fov;
    int         pixel_samples[2]; /* PixelSamples (context.c:210-222)                              */
    int         gather_nsamples;  /* Option "gather" "nsamples" (option.c:545-549), default 64     */
    int         accel_method;     /* Option "raytrace" "accel_method": 0 grid, 1 bvh, 2 hip        */
    int         nthreads;
    int         world_complete;   /* WorldBegin ... WorldEnd seen                                  */
    uint32_t    nskipped, nrequests;  /* requests outside the ray-query path / all requests      */
    uint32_t    nunknown;             /* of the skipped: not RenderMan requests at all           */
    char        display_name[1024];   /* after display.c's extension rule (".hdr")                */
    char        display_type[64];
} lh_rib_info_t;

int  lh_rib_load(const char *path, lh_rib_scene_t **scene_out);   /* 0 / -1 (lh_rib_last_error) */
void lh_rib_free(lh_rib_scene_t *scene);
const char *lh_rib_last_error(void);
int  lh_rib_info(const lh_rib_scene_t *scene, lh_rib_info_t *info);
/* parse diagnostics, one per line, as `lsh` prints them to stdout ("Unknown RIB command: X") */
const char *lh_rib_messages(const lh_rib_scene_t *scene);
/* borrowed pointers into the scene: positions/normals are npositions x double[4] (normals may be
 * NULL), indices are triangle corners (3 per primitive) */
int  lh_rib_mesh(const lh_rib_scene_t *scene, uint32_t mesh, uint32_t *npositions,
                 const double **positions, uint32_t *nindices, const uint32_t **indices,
                 const double **normals, int *two_side);
/* meshes (+ normals) of a parsed scene into an accelerator, in order; the caller commits */
int  lh_accel_add_rib_scene(lh_accel_t *accel, const lh_rib_scene_t *scene);
/* Radiance RGBE, run-length coded, byte-compatible with the reference's "file" display driver
 * (src/display/hdrdrv.c:38-121, src/imageio/rgbe.c:78-96,118-140,241-345); rgb as above */
int  lh_hdr_write(const char *path, int width, int height, const float *rgb);

/* ---- synthetic workloads (BASELINE configs 3 and 5): input generation only ----
 * One xorshift64 stream (*state; shifts 13/7/17; seed 88172645463325252): triangles first, the ray
 * dump continues it.  positions_xyz: 9 doubles per triangle (packed xyz), indices: identity.
 * lh_synth_tessellate: midpoint subdivision, triangle i -> 4i..4i+3, tri_out holds
 * ntriangles * 4^levels * 9 doubles. */
#define LH_SYNTH_SEED 88172645463325252ULL
void lh_synth_soup_triangles(uint64_t *state, uint32_t ntriangles, double half_extent,
                             double *positions_xyz, uint32_t *indices);
void lh_synth_soup_rays(uint64_t *state, size_t n, double *org_xyz, double *dir_xyz);
/* advance the stream by ndraws uniforms without producing them (a triangle is 12 draws, a ray 5):
 * how a rank that owns a slice of a ray dump reaches its first ray */
void lh_synth_skip(uint64_t *state, uint64_t ndraws);
void lh_synth_tessellate(const double *tri_in, size_t ntriangles, int levels, double *tri_out);

/* copy of the flattened BVH for cross-checks (tests): sizes via lh_accel_info.
 * nodes: nnodes*64 bytes, tri32: ntriangles*48 bytes; either may be NULL. */
int  lh_accel_export(const lh_accel_t *accel, void *nodes, void *tri32);

#ifdef __cplusplus
}
#endif
#endif
