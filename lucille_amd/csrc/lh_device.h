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
    const void *nodes;     /* lh_node_t[nnodes]  (64 B each)                 */
    const void *tri32;     /* lh_tri32_t[ntris]  (48 B each, leaf order)     */
    const void *tri64;     /* lh_tri64_t[ntris]  (72 B each, prim-id order)  */
    const void *qnodes;    /* lh_qnode_t[nnodes] (32 B each, 16-bit grid)     */
    float       grid_lo[3], grid_step[3];
    int         use_qnodes;   /* 0: fp32 2-wide, 1: 16-bit grid 2-wide, 2: 16-bit grid 4-wide, 3: 8-wide compressed */
    const void *q4nodes;      /* lh_q4node_t[nq4nodes] (64 B each)                          */
    const void *q8nodes;      /* lh_q8node_t[nq8nodes] (128 B each), or NULL                       */
    uint32_t    nq8nodes, q8_depth;
    int         prefer_q8;    /* this launch walks the 8-wide nodes (ray dumps over scenes larger than the Infinity Cache) */
    const void *q4tnodes;     /* the same nodes child-major (lh_quad.hip), or NULL                */
    uint32_t    nq4nodes, q4_depth;
    const void *c8nodes;      /* lh_c8node_t[nc8nodes] (80 B each): use_qnodes == 3               */
    const void *tri32_c8;     /* lh_tri32_t[ntris] in the 8-wide tree's leaf order                 */
    uint32_t    nc8nodes, c8_depth, stack_rows;
    uint32_t    c8_stride;    /* bytes between 8-wide records: 80 packed, 128 one record per cache line (LH_C8_STRIDE) */
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
    uint32_t    ray_chunk; /* rays a persistent wave reserves per atomic on the global cursor */
    int         stack_guard;    /* set by the launchers when the LDS rows do not cover the tree's worst case: the walks check before they push */
    uint32_t    stack_cap; /* 0: 64 LDS stack rows at most; 8..62: a lower cap (tests of the overflow path) */
    int         nodes_2wide_available;   /* host-built scenes: the 2-wide formats can be uploaded on demand (deep-tree fallback) */
} lh_dev_scene_t;

/* traversal statistics accumulated by the COUNT variants (u64 each) */
enum { LH_CNT_NODES = 0, LH_CNT_TRIS = 1, LH_CNT_EXACT = 2, LH_CNT_RAYS = 3, LH_CNT_N = 4,
       /* diagnostics of the COUNT build (LH_DEBUG_COUNTERS=1 prints them): lane slots offered by wave
        * iterations that ran a node step / a triangle step (64 per iteration), regroup iterations x 64 */
       LH_CNT_NODE_SLOTS = 4, LH_CNT_TRI_SLOTS = 5, LH_CNT_REGROUP_SLOTS = 6,
       LH_CNT_RETRACED = 7,      /* rays sent through the reference's own walk (lh_reftrace.h) */
       LH_CNT_DEV = 8 };

/* kernel variants (A/B-testable in one process) */
enum {
    LH_VARIANT_DIRECT      = 0,   /* one ray per lane, grid covers the batch  */
    LH_VARIANT_PERSIST_WAVE = 1,  /* persistent waves, 64-ray chunks          */
    LH_VARIANT_PERSIST_LANE = 2,  /* persistent waves, ballot-compacted refill */
    LH_VARIANT_UNIFIED      = 3,  /* + single-loop walk: one record per lane per iteration */
    LH_VARIANT_SPEC         = 4,  /* + speculative walk, leaves parked and tested in batches */
    LH_VARIANT_UNIFIED4     = 5,  /* single-loop walk over the 4-wide nodes: one record per lane per iteration */
    LH_VARIANT_QUAD         = 7,  /* lh_quad.hip: one ray per quad of lanes, lane k tests child k / triangle k (A/B) */
    LH_VARIANT_LEAN         = 6   /* lh_trace2.hip: the speculative 4-wide walk without fp64 state, 16-row LDS ring
                                     stack, candidates resolved by a separate fp64 pass */
};

#define LH_T2_ROWS        16           /* LDS ring rows per lane of the lean walk */
#define LH_PRIM_PENDING   0xFFFFFFF0u  /* | candidate count: the slot holds unresolved candidates (not a valid id: ids < 2^29) */
#define LH_OCC_PENDING    3u           /* any-hit: the ray waits in the pending queue */
#define LH_PRIM_OVERFLOW  0xFFFFFFFDu  /* the LDS stack column was too short for this ray: k_overflow_fix */
#define LH_OCC_OVERFLOW   4u

#ifdef __cplusplus
extern "C" {
#endif

/* launchers implemented in lh_kernels.hip; stream is a hipStream_t */
int lh_launch_trace(const lh_dev_scene_t *sc, size_t n, const double *d_org,
                    const double *d_dir, uint32_t *d_prim, double *d_t,
                    double *d_u, double *d_v, int anyhit, uint8_t *d_occluded,
                    unsigned long long *d_counters /* LH_CNT_N or NULL */,
                    unsigned long long *d_workq /* persistent cursor */,
                    int variant, int grid_blocks, int min_active, int tri_batch, void *stream);

int lh_launch_trace_ao(const lh_dev_scene_t *sc, size_t nslots, int ntheta, int nphi, unsigned long long seed,
                       const double *d_hitrec, const unsigned long long *d_slot_key, unsigned int *d_occ_count,
                       unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks, int min_active,
                       int tri_batch, uint32_t *d_queue, uint32_t *d_qcount, uint32_t qcap, void *stream);
int lh_launch_ao_queue(const lh_dev_scene_t *sc, int ntheta, int nphi, unsigned long long seed,
                       const double *d_hitrec, const unsigned long long *d_slot_key, unsigned int *d_occ_count,
                       unsigned long long *d_counters, uint32_t *d_queue, uint32_t *d_qcount, uint32_t qcap, void *stream);

/* lh_quad.hip: the quad-per-ray walk (variant LH_VARIANT_QUAD) */
#define LH_QUAD_WAVES_PER_SIMD 4         /* what its register allocation allows (128 VGPRs) */
int lh_launch_trace_pt(const lh_dev_scene_t *sc, size_t npaths, const lh_camera_t *cam, int x0, int y0, int w, int spp, int s0,
                       int full_width, int max_depth, unsigned long long seed, const double *d_nrm9, const double *d_col9,
                       const uint32_t *d_prim_mesh, const void *d_materials, const lh_material_t *override_mat,
                       const float env_rgb[3], const void *d_env_map, int env_w, int env_h, int ref_weights,
                       float *d_radiance, unsigned long long *d_nrays, unsigned int *d_maxdepth,
                       unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks, int min_active,
                       int tri_batch, void *stream);
int lh_quad_make_nodes(uint32_t nq4, const void *d_q4nodes, void *d_q4tnodes, void *stream);
int lh_quad_blocks_per_cu(uint32_t stack_rows);
int lh_launch_trace_quad(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dir, uint32_t *d_prim,
                         double *d_t, double *d_u, double *d_v, int anyhit, uint8_t *d_occ,
                         unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks, int min_active,
                         int tri_batch, int *over_fix_out, void *stream);

/* launchers implemented in lh_trace2.hip */
int lh_trace2_blocks_per_cu(void);
int lh_launch_trace2(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dir,
                     uint32_t *d_prim, double *d_t, double *d_u, double *d_v, int anyhit, uint8_t *d_occluded,
                     unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks,
                     int min_active, int tri_batch, int *d_spill, uint32_t *d_queue, uint32_t *d_qcount,
                     uint32_t qcap, void *stream);
int lh_launch_trace2_ao(const lh_dev_scene_t *sc, size_t nslots, int ntheta, int nphi, unsigned long long seed,
                        const double *d_hitrec, const unsigned long long *d_slot_key, unsigned int *d_occ_count,
                        unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks,
                        int min_active, int tri_batch, int *d_spill, uint32_t *d_queue, uint32_t *d_qcount,
                        uint32_t qcap, void *stream);

#ifdef __cplusplus
}
#endif
#endif
