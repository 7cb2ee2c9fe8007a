/*
 * lh_device.h -- shared host/device declarations for the gfx950 kernels.
 * Internal to liblucille_hip.so (the public C-ABI is include/lucille_hip.h).
 */
#ifndef LH_DEVICE_H
#define LH_DEVICE_H

#include <stdint.h>
#include <stddef.h>
#include "../../include/lucille_hip.h"     /* lh_camera_t, lh_material_t in the launcher signatures */

#define LH_BLOCK      256          /* 4 wavefronts of 64 lanes              */
#define LH_MISS_PRIM  0xFFFFFFFFu
#define LH_T_INF      1.0e38       /* RI_INFINITY, include/ri.h:47          */

/* device-resident scene (all pointers are HBM addresses) */
typedef struct lh_dev_scene {
    const void *nodes;     /* lh_node_t[nnodes] (64 B each, 2-wide fp32): LH_VARIANT_DIRECT only; uploaded on first use, NULL otherwise */
    const void *tri32;     /* lh_tri32_t[ntris]  (48 B each, leaf order)     */
    const void *tri64;     /* lh_tri64_t[ntris]  (72 B each, prim-id order)  */
    float       grid_lo[3], grid_step[3];   /* the scene's 16-bit grid (lh_q4node_t, lh_q8node_t)  */
    const void *q4nodes;      /* lh_q4node_t[nq4nodes] (64 B each): what the default walk reads   */
    uint32_t    nq4nodes, q4_depth;
    uint32_t    q4_stack;  /* rows the 4-wide walk needs (lh_bvh_t); 0: 3 x q4_depth + 5 */
    const void *q8nodes;      /* lh_q8node_t[nq8nodes] (128 B each), or NULL                       */
    uint32_t    nq8nodes, q8_depth;
    int         prefer_q8;    /* this launch walks the 8-wide nodes (ray dumps over scenes larger than the Infinity Cache) */
    uint32_t    stack_rows;   /* LDS stack rows of this launch (set by the launchers)              */
    /* reference-order tree (lh_refbvh.c), for exact-t tie winners and beam queries; may be NULL */
    const void *ref_lca;      /* int4[ref_nnodes]: parent, depth, axis0, child[0]            */
    const void *prim_leafpos; /* uint2[ntris]: leaf node of the primitive, position in leaf  */
    const void *ref_nodes;    /* lh_refnode_t[ref_nnodes] (128 B each)                       */
    const void *ref_leaf_prims; /* uint32[ntris]: primitive ids in the reference's leaf order */
    uint32_t    ref_nnodes;
    int         ref_empty;
    double      ref_bmin[3], ref_bmax[3];
    uint32_t    ntris;
    uint32_t    nnodes;
    uint32_t    max_depth;
    float       scene_r;   /* max |coordinate| of the scene box              */
    float       deg_dcap;  /* a ray with a direction component beyond this is decided by the reference's own walk: the traversal tree leaves out
                              zero-area triangles whose fp64 determinant is provably below the reference's 1e-14 only up to there (lh_bvh.c
                              tri_dead_class); INFINITY: no such triangle in this scene */
    uint32_t    cap_srcs;  /* bit s: rays of source s (0 arrays, 1 AO rays of the refill, 2 camera rays of a path-traced pass) can exceed deg_dcap at all -- bit 0: it is
                              finite; bits 1, 2: it is below 1 (those rays are unit vectors).  0 for every scene without such triangles: the persistent walk's
                              refill then skips the test whole (one scalar branch) */
    uint32_t    ndanger;   /* deg_dcap < LH_DEG_DCAP_ALL: LH_DANGER_ALL (0) = every ray beyond deg_dcap takes the reference walk; else only rays that hit ... */
    uint32_t    danger[3]; /* ... this box, on the scene's 16-bit grid in the nodes' packing (lo | hi << 16 per axis, rounded outward): the union of the boxes of the
                              leaves of lucille's own tree that hold a zero-area triangle (lh_bvh.h), asked through slab_w like any node's box */
    uint32_t    ray_chunk; /* rays a persistent wave reserves per atomic on the global cursor */
    uint32_t    ray_budget;/* wave iterations after which a ray leaves the persistent walk for the cooperative one */
    int         stack_guard;    /* set by the launchers when the LDS rows do not cover the tree's worst case: the walks check before they push */
    uint32_t    stack_cap; /* 0: 64 LDS stack rows at most; 8..62: a lower cap (tests of the overflow path) */
    uint32_t    coop_patience; /* 0 (half a second), or: wall-clock ticks of 10 ns the pass beside a producer waits without progress before it leaves the queue to the sweep (tests) */
    uint32_t    ao_group;  /* fused AO stage: work items in groups of this many hit slots, a group's slots side by side per sample; 0 (the default): a slot's samples side by side */
    uint32_t    top_nodes; /* the first top_nodes 4-wide nodes (level order: the top of the tree) are walked from a copy in the workgroup's LDS; 0: none */
    const void *cam_src;       /* NULL, or: ray source 2 -- the launch's rays are the camera rays of a path-traced pass (PtCamSrc, lh_pt.h), org / dir unused */
    const uint32_t *n_dev;     /* NULL, or: the launch's ray count lives on the device (the path tracer's bounce chain; the host passes an upper bound) */
    uint32_t   *diag_out;      /* NULL, or: four counts per ray of this launch (4-wide node visits, leaf visits, triangle records through the fp32
                                  filter, fp64 tests): the per-ray diagnostics of ri_bvh_intersect's `user` argument (bvh.h:103-110, bvh.c:451-456) */
    unsigned long long *diag_clock;   /* diagnostics (LH_STAGE_TIMING): [2][waves] start / exit wall clock of every persistent wave, or NULL */
} lh_dev_scene_t;

/* traversal statistics accumulated by the COUNT variants (u64 each) */
enum { LH_CNT_NODES = 0, LH_CNT_TRIS = 1, LH_CNT_EXACT = 2, LH_CNT_RAYS = 3, LH_CNT_N = 4,
       /* diagnostics of the COUNT build (LH_DEBUG_COUNTERS=1 prints them): lane slots offered by wave
        * iterations that ran a node step / a triangle step (64 per iteration), regroup iterations x 64 */
       LH_CNT_NODE_SLOTS = 4, LH_CNT_TRI_SLOTS = 5, LH_CNT_REGROUP_SLOTS = 6,
       LH_CNT_RETRACED = 7,      /* rays sent through the reference's own walk (lh_reftrace.h) */
       LH_CNT_HIST = 8,          /* 24 buckets: rays by node visits, bucket b = visits in [2^(b-1), 2^b) (b = 0: none) */
       LH_CNT_DEV = 32 };

/* kernel variants (the `variant` argument of the query entry points; numbering kept from rounds 1-2) */
enum {
    LH_VARIANT_DIRECT = 0,   /* the textbook walk: one ray per lane, while-while, 2-wide fp32 nodes -- the in-process reference */
    LH_VARIANT_SPEC   = 4    /* the default: persistent waves, branch-free 4-wide (or 8-wide) node step, parked leaves */
};

#define LH_ROWS_UNCHECKED 64u          /* LDS stack rows (1 KiB each per 256-thread workgroup) up to which the walk runs unchecked (3 x depth + 5 rows, or the
                                         builder's count of the deepest path).  A workgroup with more than 64 KiB of LDS halves what a CU holds -- a
                                         21-level device-built tree at 66 rows, with the attribute set to 80 or to 78 KiB alike: 129-132 ms against
                                         94 ms at 40 checked rows and 87 ms for a 20-level tree under 64 (r03) -- so deeper trees take the checked walk */
#define LH_TOP_AUTO       0xFFFFFFFFu  /* lh_dev_scene_t.top_nodes: as many as the CU's LDS leaves over beside the stack rows (the default) */
#define LH_TOP_NODES_MAX  512u         /* 4-wide nodes a workgroup may keep in LDS behind its stack rows (32 KiB) */
#define LH_ROWS_CHECKED   34u          /* ... of the checked walk: four workgroups per CU (4 x 34 KiB) and the cooperative walk's 17 KiB beside them; what every
                                         launch of 65536 rays or more walks when its tree asks for more (rows4 in lh_kernels.hip) */
#define LH_NPART          8            /* cursor partitions of a persistent launch: one per XCD (each with its own L2) */
#define LH_CURSOR_STRIDE  32           /* 32-bit words between two cursors: a 128-byte line each (device-scope atomics on one line serialise like atomics on
                                         one address) */
#define LH_CURSOR_WORDS   ((LH_NPART + 1) * LH_CURSOR_STRIDE)   /* the cursors + the drained-partition mask, per launch */
#define LH_AO_QCAP        (1u << 22)   /* rays of one launch that may wait in the fix-up queue (8 B each): fragile AO hits, rays out of visit budget */
#define LH_DUMP_BUDGET    2048u        /* ... of ray-dump launches (incoherent rays: ages run to several times the steps) */
#define LH_DUMP_MIN_ACTIVE 24           /* ray dumps over the 4-wide nodes: regroup below this many working lanes ... */
#define LH_DUMP_TRI_BATCH  8            /* ... and pass over the parked leaves once this many lanes hold one (lh_query.hip lh_launch) */
#define LH_DUMP8_MIN_ACTIVE 40          /* ray dumps over the 8-wide nodes (scenes beyond the Infinity Cache): S-soup-10M, 50 M rays, 1 787 -> 1 821 Mrays/s closest hit, 2 796 -> 2 875 any hit
                                         * against 32 / 12 (tools/experiments/knob_sweep7.py, r05); 24 / 8, the 4-wide dumps' pair, loses 2 % here */
#define LH_DUMP8_TRI_BATCH  20
#define LH_TILE_CHUNK     1024u        /* rays per cursor atomic in the tile pipelines (camera rays, AO rays of a slot, path-tracing bounces: neighbours in the
                                         batch are neighbours in space; ray dumps keep "ray_chunk" = 256): config 4 frame 148 -> 134 ms, config 5 87.0 -> 85.5 */
#define LH_AO_BUDGET      384u         /* ... of the fused AO stage: one ray in 800 leaves the surface it starts on at so low an angle that it threads the boxes of
                                         hundreds of triangles; past 128 iterations 0.12 % of a config-5 frame's AO rays are still walking, past 384 0.02 %.
                                         Each of them restarts in the cooperative walk (16 lanes a ray), so the lower budget trades the launch's tail for a
                                         queue: whole frame 61.9 ms at 128, 59.4 at 384, 58.3 at 512; an eighth of it 10.4 / 10.4 / 11.2 (tools/ao_budget_probe.py).  A launch of 2^27 rays
                                         or more takes twice this (lh_tile.hip) unless the caller set a budget */
#define LH_RAY_BUDGET     128u         /* default visit budget of the persistent walk (set_param "ray_budget") */
#define LH_PRIM_OVERFLOW  0xFFFFFFFDu  /* the LDS stack column was too short for this ray: k_overflow_fix */
#define LH_OCC_OVERFLOW   4u

#ifdef __cplusplus
extern "C" {
#endif

/* the fix-up queue of the persistent launches of one stream (rays out of visit budget, fragile AO hits) and the second
 * stream its consumer runs on, concurrently with the launch that fills it */
typedef struct lh_fixq {
    void     *queue;          /* qcap x u64 (device), zero = empty slot */
    uint32_t *qcount;         /* [0] appends, [1] overflow flag, [2] producer waves that have left, [4 ..] consumer heads (device) */
    uint32_t  qcap;
    void     *aux_stream;     /* hipStream_t */
    void     *ev_ready, *ev_done;     /* hipEvent_t: queue reset on the launch stream; consumer finished */
} lh_fixq_t;

/* launchers implemented in lh_kernels.hip; stream is a hipStream_t */
int lh_launch_trace(const lh_dev_scene_t *sc, size_t n, const double *d_org,
                    const double *d_dir, uint32_t *d_prim, double *d_t,
                    double *d_u, double *d_v, int anyhit, uint8_t *d_occluded,
                    unsigned long long *d_counters /* LH_CNT_DEV or NULL */,
                    unsigned long long *d_workq /* LH_NPART persistent cursors */,
                    int variant, int grid_blocks, int min_active, int tri_batch,
                    const lh_fixq_t *q, int ncus, void *stream);
int lh_launch_trace_ao(const lh_dev_scene_t *sc, size_t nslots, int ntheta, int nphi, unsigned long long seed,
                       const double *d_hitrec, const unsigned long long *d_slot_key, unsigned int *d_occ_count,
                       unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks, int min_active,
                       int tri_batch, const lh_fixq_t *q, int ncus, const unsigned long long *d_nslots, uint32_t budget_big, void *stream);
int lh_trace_formats_needed(const lh_dev_scene_t *sc, int variant);
int lh_trace_rows(const lh_dev_scene_t *sc);        /* LDS stack rows the default walk launches with on this scene */

#ifdef __cplusplus
}
#endif
#endif
