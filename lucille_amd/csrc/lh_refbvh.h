/*
 * lh_refbvh.h -- reference-order tree (see lh_refbvh.c): lucille's own binned-SAH BVH,
 * bit-faithful, kept beside the fast traversal tree for the two tree-dependent results
 * (beam visibility, exact-t tie winners).
 */
#ifndef LH_REFBVH_H
#define LH_REFBVH_H

#include <stdint.h>
#include "lh_bvh.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 128-byte node: both children's fp64 boxes as the reference stores them in the parent
 * (bvh.c:1511-1548), children, split axis, leaf range, parent/depth for LCA walks */
typedef struct lh_refnode {
    double   box[2][6];      /* child k: bmin xyz, bmax xyz (with margin)     */
    int32_t  child[2];       /* node indices; -1 in leaves                    */
    int32_t  axis;           /* axis0 (bvh.c:1497)                            */
    int32_t  is_leaf;
    uint32_t first, count;   /* leaf: range in leaf_prims                     */
    int32_t  parent, depth;
} lh_refnode_t;

typedef struct lh_refbvh {
    uint32_t      ntris, nnodes, max_depth;
    int           empty;
    double        bmin[3], bmax[3];     /* scene box with margin (bvh.c:325-340) */
    lh_refnode_t *nodes;                /* root = 0                              */
    uint32_t     *leaf_prims;           /* primitive ids in the reference's leaf order */
    uint32_t     *prim_leaf;            /* per primitive: its leaf node           */
    uint32_t     *prim_pos;             /* per primitive: position inside the leaf */
} lh_refbvh_t;

int      lh_refbvh_build(lh_refbvh_t *out, const lh_tri64_t *tri64, uint32_t ntris, int nthreads);
void     lh_refbvh_release(lh_refbvh_t *bvh);
uint32_t lh_refbvh_tie_winner(const lh_refbvh_t *bvh, uint32_t a, uint32_t b, const int dir_sign[3]);

#ifdef __cplusplus
}
#endif
#endif
