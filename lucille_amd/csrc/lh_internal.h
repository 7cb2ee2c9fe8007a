/*
 * lh_internal.h -- what the translation units of liblucille_hip.so share: the accelerator object, the refcounted host
 * scene, error reporting and the launch helpers.  Internal (the public C ABI is include/lucille_hip.h).
 *
 *   lh_commit.hip   lifetime: create / add meshes / commit (host or device build) / replicas / destroy / parameters
 *   lh_query.hip    ray queries: device, host and pipelined host batches, statistics, beam visibility
 *   lh_tile.hip     the callers on either side: AO tiles / bands / frames, hit epilogue, path-traced tiles
 */
#ifndef LH_INTERNAL_H
#define LH_INTERNAL_H

#include <hip/hip_runtime.h>

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

/* lh_last_error() of the calling thread; lh_fail formats it and returns -1 */
int lh_fail(const char *fmt, ...);
#define fail lh_fail
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return lh_fail("%s failed: %s", #x, hipGetErrorString(e_)); } while (0)

/* attribute kinds of lh_accel_set_attribute: per-vertex xyz (colour, tangent, binormal), per-vertex st,
 * per-index st (texcoords_unshared) -- the optional members of ri_geom_t that ri_intersection_state_build reads */
struct lh_mesh_copy { uint32_t npos, nidx; double *pos; uint32_t *idx; double *nrm; int two_side;
                      double *attr[5]; };

struct lh_buf { void *p; size_t cap; };
#define LH_AOQ_SLOTS 4
#define LH_PIPE_DEPTH_MAX 8   /* staging blocks of a pipelined host batch (lh_query.hip) */

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
    int received;             /* the scene arrived as an image from another rank (lh_dist.hip): device arrays only */
    int ref_on_device;        /* lucille's own tree of a device-built scene was built on the device too (lh_refbuild.hip): no host copy */
    int ref_state; pthread_t ref_thread; int ref_thread_live; int ref_threads;
    void **trash; uint32_t ntrash;   /* host blocks the device-side commit no longer needs: freed by the background thread (unmapping 0.5 GB takes 0.1 s) */
};
extern pthread_mutex_t g_scene_mu;

struct lh_accel {
    int device;
    int committed;
    int commit_failed;        /* a commit that failed half-way: device memory is released by destroy, a retry is refused */
    /* staged meshes (host copies, packed xyz) */
    lh_mesh_copy *meshes; uint32_t nmeshes;
    lh_host_scene *hs;        /* never NULL after create */
    void *d_ref_lca, *d_prim_leafpos, *d_ref_nodes, *d_ref_leaf_prims;
    void *d_danger;                    /* 8 + LH_DANGER_MAX x 6 doubles: the count, then the boxes of lh_dev_scene_t.danger (lh_commit.hip lh_danger_scan) */
    /* device */
    lh_dev_scene_t dev;
    void *d_nodes, *d_tri32, *d_tri64, *d_q4nodes, *d_q8nodes;
    int ncus;                          /* compute units of the device */
    int wide8;                         /* ray dumps walk the 8-wide nodes: -1 when the hot set exceeds the Infinity Cache (default), 0 never, 1 always */
    unsigned long long *d_cursor, *d_counters;   /* d_cursor: LH_NCURSOR blocks of LH_CURSOR_WORDS words (a line per cursor + the drained mask), one block per launch in flight */
    unsigned cursor_next;
    pthread_mutex_t mu;                /* serialises the entry points of ONE accelerator (recursive) */
    int stat_on;                       /* lh_accel_trace_statistics */
    unsigned long long stat[5];        /* nodes, filter tests, fp64 tests, rays, hits */
    unsigned long long stat_slots[3];  /* lane slots (64 per wave iteration) of node steps, triangle passes, regroups (tile pipelines, lh_accel_slot_statistics) */
    hipStream_t stream;
    uint64_t device_bytes;
    double upload_seconds;
    int grid_blocks;
    int grid_user;                     /* the persistent grid was set by the caller (set_param "grid", LH_GRID_BLOCKS): launches do not resize it */
    int min_active;
    uint32_t ray_chunk;                /* rays reserved per cursor atomic (LH_RAY_CHUNK) */
    int tri_batch;
    int knobs_user;                    /* min_active / tri_batch were set by the caller (set_param, LH_MIN_ACTIVE, LH_TRI_BATCH): ray dumps do not pick their own */
    int default_variant;
    /* per-stream fix-up queues of the persistent launches (rays out of visit budget / stack rows, fragile AO hits) */
    struct { hipStream_t stream; int used; lh_fixq_t q; } aoq[LH_AOQ_SLOTS];
    /* staging for host batches */
    void *d_stage; size_t stage_bytes;
    /* pipelined host batches: a ring of pinned in/out staging blocks and their device twins; rays up on s[0]; trace + records down alternate between s[1] and s[2] */
    struct { void *h_in[LH_PIPE_DEPTH_MAX], *h_out[LH_PIPE_DEPTH_MAX], *d_in[LH_PIPE_DEPTH_MAX], *d_out[LH_PIPE_DEPTH_MAX]; hipStream_t s[3];
             hipEvent_t in_done[LH_PIPE_DEPTH_MAX], done[LH_PIPE_DEPTH_MAX]; size_t cap; int depth, ready; } pipe;
    void *d_nrm9;                      /* hs->nrm9 on the device */
    void *d_attr9[3], *d_st6, *d_inside;            /* colour / tangent / binormal, st, inside flags (uploaded at commit if present) */
    void *d_prim_mesh;                 /* mesh ordinal per primitive (materials; uploaded on first use) */
    lh_material_t *materials; uint32_t nmaterials; void *d_materials; int materials_dirty;
    lh_environment_t env; void *d_env_map; int env_set;      /* env_set: lh_accel_set_environment was called (else the path tracer's environment is constant white) */
    lh_buf r_state;                    /* lh_accel_state_build_host staging */
    lh_buf r_uni;                      /* lh_render_ao_tile_host: caller uniforms on the device */
    lh_buf r_diag;                     /* LH_STAGE_TIMING: wave start / exit clocks */
    lh_buf r_bands;                    /* lh_render_ao_bands: first line of every band */
    void *h_read;                      /* 1 KiB of pinned host memory: the read-backs at the end of an AO batch (occlusion totals, hit count, queue flags) */
    /* tile-render scratch (lh_render_ao_tile) */
    lh_buf r_org, r_dir, r_prim, r_t, r_u, r_v, r_slot, r_hitrec, r_aorg, r_adir, r_occ, r_blocks, r_key, r_frame, r_occcount;
    uint64_t last_retraced;            /* rays the last counted launch finished outside the main kernel */
    int ao_fused;                      /* AO rays generated inside the any-hit kernel (default); 0: materialised in HBM */
    uint32_t ao_budget;                /* visit budget of the fused AO stage (0: dev.ray_budget) */
    int ao_budget_user;                /* set by the caller (set_param / LH_AO_BUDGET / "ray_budget"): taken as it is, whatever the launch's size */
    uint32_t dump_budget;              /* visit budget of ray-dump launches (the tile pipelines': dev.ray_budget) */
    int build_auto;                    /* the commit chose the builders by the size of the scene: a failing device build falls back to the host */
    int poison_outputs;                /* LH_POISON_OUTPUTS=1 (tests, tools/fuzz_*): a ray dump's output arrays are filled with 0x77 before the launch -- an answer slot that nobody
                                          writes shows up as a wrong record instead of as whatever the buffer held (the lost any-hit rays of r05 hid behind recycled buffers) */
    int fast_start;                    /* device-built scenes: launch before lucille's own tree is attached (ties by primitive id until then) */
    lh_buf p_org2, p_dir2, p_path, p_path2, p_thr, p_thr2, p_rad, p_counts;   /* path tracer */
    unsigned long long *d_total;
    size_t r_nsamples, r_nslots, r_nao;
    struct lh_combiner *comb;          /* lh_accel_intersect1: concurrent single-ray callers coalesced into one launch (lh_query.hip) */
    int combine;                       /* 1 (default): coalesce; 0: one launch per call, as in rounds 1-3 */
    /* lh_accel_intersect1 on the calling thread, over the host copy of the trees (lh_hostwalk.c): host_walk 1 (default) / 0;
     * hw_ns: what a host walk costs on this scene (ns, running mean of timed samples); hw_calls: calls answered there; hw_gpu_left:
     * calls still to send to the device before the host is probed again (a scene whose walks are long: S-soup-1M's incoherent rays
     * cost 5-10 us of cache misses each on the host, the coalesced device path amortises to about that) */
    int host_walk; unsigned long long hw_calls; double hw_ns; int hw_gpu_left;
};

#define LH_NCURSOR 64

/* lucille calls accel->intersect from up to 16 render threads at once (render.c:1043-1105): every
 * entry point that touches the accelerator's buffers holds its lock */
struct lh_guard {
    pthread_mutex_t *m;
    explicit lh_guard(const lh_accel_t *a) : m(a ? (pthread_mutex_t *)&a->mu : NULL) { if (m) pthread_mutex_lock(m); }
    ~lh_guard() { if (m) pthread_mutex_unlock(m); }
};


/* a spin-wait hint that is not x86-only */
#if defined(__x86_64__) || defined(__i386__)
#define LH_CPU_RELAX() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define LH_CPU_RELAX() __asm__ __volatile__("yield")
#else
#define LH_CPU_RELAX() ((void)0)
#endif

static inline double lh_now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define now_s lh_now_s

/* node formats a walk can read; only the one the default kernel uses is uploaded at commit, the
 * others (the textbook variant's 2-wide fp32 nodes, the 8-wide nodes of large ray dumps) on first use */
enum { LH_FMT_F32 = 1, LH_FMT_Q16X4 = 4, LH_FMT_Q8 = 32 };

/* lh_commit.hip */
int  lh_ensure_formats(lh_accel_t *a, int mask);
int  lh_sync_ref(lh_accel_t *a, bool wait);          /* attach the background-built reference-order tree (wait: block for it) */
int  lh_ensure_buf(lh_buf *b, size_t bytes);
void lh_free_buf(lh_buf *b);
/* the scene image (lh_commit.hip): one rank's committed scene handed to the others (lh_dist.hip) */
typedef struct lh_scene_image {
    uint32_t magic, ntris, nnodes, max_depth, nleaves, nq4, q4_depth, q4_stack, nq8, q8_depth, ref_nnodes, nmeshes;
    int      have_ref, ref_empty, has_nrm, has_attr[3], has_st, has_inside;
    float    bmin[3], bmax[3], grid_lo[3], grid_step[3];
    double   ref_bmin[3], ref_bmax[3], build_seconds, ref_build_seconds;
    double   deg_dcap; uint32_t nlive, pad_;        /* lh_bvh_t: the triangles outside the traversal tree and the rays that need the reference walk for them */
} lh_scene_image_t;
int  lh_scene_image_header(lh_accel_t *a, lh_scene_image_t *h);
int  lh_scene_image_arrays(lh_accel_t *a, const lh_scene_image_t *h, void **ptr, size_t *bytes, int cap);
uint32_t *lh_scene_image_prim_geom(lh_accel_t *a);
uint32_t *lh_scene_image_prim_index(lh_accel_t *a);
int  lh_scene_image_alloc(lh_accel_t *a, const lh_scene_image_t *h);
int  lh_scene_image_finish(lh_accel_t *a);
/* lh_hostwalk.c */
extern "C" int lh_host_walk_closest(const lh_bvh_t *b, const lh_refbvh_t *ref, const double o[3], const double d[3], uint32_t *prim, double *t, double *u, double *v);
/* lh_query.hip */
void lh_comb_destroy(lh_accel_t *a);                 /* the single-ray combiner's pinned block and stream (lh_accel_destroy) */
int  lh_launch(lh_accel_t *a, size_t n, const void *d_org, const void *d_dir, void *d_prim, void *d_t, void *d_u, void *d_v,
               void *d_occ, int mode, int variant, unsigned long long *d_counters, hipStream_t s, bool dump);
int  lh_aoq_slot(lh_accel_t *a, hipStream_t s);
int  lh_ensure_stage(lh_accel_t *a, size_t bytes);

#endif
