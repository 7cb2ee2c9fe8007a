/*
 * lh_bvh.h -- host-side builder that turns lucille geometry into the
 * structure-of-arrays BVH the gfx950 kernels traverse.
 *
 * Replaces, for the HIP accelerator, what ri_bvh_build does for the CPU one
 * (reference: src/render/bvh.c:276-379 ri_bvh_build, :1736-1826
 * create_triangle_list, :1328-1564 bvh_construct).  The primitive numbering
 * is the reference's (running index of create_triangle_list: geom-list order,
 * then triangle order); the tree itself is this project's own, because hit
 * records are tree-independent while GPU traversal wants
 * small leaves, 64-byte nodes and fp32 boxes.
 */
#ifndef LH_BVH_H
#define LH_BVH_H

#include <math.h>
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LH_MAX_LEAF_TRIS 4
#define LH_MAX_DEPTH     60      /* builder falls back to median splits past 48 */

/* 64-byte inner node: boxes of BOTH children + their references.
 *   ref >= 0 : index of an inner node
 *   ref <  0 : leaf, x = ~ref, first = x >> 2, count = (x & 3) + 1
 *   LH_REF_EMPTY : no child (box is inverted, never hit)                   */
typedef struct lh_node {
    float   lo0[3], hi0[3];
    float   lo1[3], hi1[3];
    int32_t ref0, ref1;
    int32_t axis;      /* split axis (diagnostics only) */
    int32_t pad;
} lh_node_t;

#define LH_REF_EMPTY ((int32_t)0x80000000)

/* 48-byte leaf triangle record (fp32 filter form): v0, e1=v1-v0, e2=v2-v0
 * rounded from the fp64 differences, the primitive id and two precomputed
 * norms used by the conservative-filter tolerances.                        */
typedef struct lh_tri32 {
    float    v0[3];  float e1x;
    float    e1y, e1z, e2x, e2y;
    float    e2z;    uint32_t prim;
    float    ne1, ne2;                            /* |e1|_2, |e2|_2 rounded up */
} lh_tri32_t;

/* 72-byte exact triangle (fp64), indexed by primitive id */
typedef struct lh_tri64 { double v[3][3]; } lh_tri64_t;

/* 64-byte 4-wide node on the same 16-bit grid: the binary tree collapsed (largest-area
 * inner child opened first) so that one 64-byte record -- one L2 request -- decides four
 * children.  The traversal kernel runs at the L2 request-rate ceiling of its footprint
 * (profiles/README.md), so halving the records per ray is what raises rays/s.
 *   w[c][k] = lo | hi << 16 of child c on axis k: one dword holds both planes of an axis, so
 *   the kernel picks (near, far) for the ray's direction sign with ONE rotate
 *   (v_alignbit_b32 by 0 or 16) instead of two selects, then converts the halves (SDWA);
 *   ref[c] as in lh_node_t (LH_REF_EMPTY for unused slots, whose box is inverted)       */
typedef struct lh_q4node {
    uint32_t w[4][3];
    int32_t  ref[4];
} lh_q4node_t;

/* 8-wide node on the scene's 16-bit grid: 128 bytes = one cache line decides eight children.  Slot s holds the child the
 * walk visits with priority s ^ (ray octant); w[s][axis] = lo | hi << 16 as in lh_q4node_t; ref as in lh_q4node_t (the
 * same leaves, the same tri32 order).  Empty slots: inverted box, LH_REF_EMPTY.  Used for ray dumps over scenes that do
 * not fit the Infinity Cache, where every record fetched costs a whole 128-byte line of HBM traffic. */
typedef struct { uint32_t w[8][3]; int32_t ref[8]; } lh_q8node_t;

typedef struct lh_bvh {
    uint32_t    ntris;
    uint32_t    nnodes;
    uint32_t    max_depth;
    uint32_t    nleaves;
    lh_node_t  *nodes;     /* nnodes (>=1 when ntris>0)                     */
    lh_tri32_t *tri32;     /* ntris, leaf order                             */
    lh_tri64_t *tri64;     /* ntris, primitive-id order                     */
    uint32_t   *prim_geom; /* ntris: mesh ordinal of primitive              */
    uint32_t   *prim_index;/* ntris: 3*i offset into that mesh's indices    */
    float       bmin[3], bmax[3];  /* scene box, fp32 outward               */
    lh_q4node_t *q4nodes;          /* nq4nodes: 4-wide collapse of the same tree */
    uint32_t    nq4nodes, q4_depth;
    uint32_t    q4_stack;      /* LDS stack rows the 4-wide walk needs on this tree (the device builder measures its deepest path); 0: 3 x q4_depth + 5 */
    lh_q8node_t *q8nodes;          /* nq8nodes: 8-wide collapse on the 16-bit grid, or NULL (lh_bvh_ensure_q8) */
    uint32_t    nq8nodes, q8_depth;
    float       grid_lo[3], grid_step[3];   /* the 16-bit grid of q4nodes / q8nodes */
    double      build_seconds;
    uint32_t    nlive;         /* primitives in the leaves of the traversal tree: ntris minus the triangles the reference can never report (lh_bvh.c tri_dead_class) */
    double      deg_dcap;      /* rays with a direction component beyond this are decided by the reference's own walk (INFINITY: no such limit) */
    /* ... unless (round 6) the cap comes from zero-area triangles that STAY in the tree (deg_dcap < LH_DEG_DCAP_ALL) and the ray misses
     * the box of every leaf of lucille's own tree that holds one: the reference can only "hit" such a triangle -- by the noise of its
     * determinant -- on rays that reach its leaf.  ndanger >= 1 boxes (bmin xyz, bmax xyz: the child box in the leaf's parent), or
     * LH_DANGER_ALL (0): more than LH_DANGER_MAX such leaves, or not computed (yet): every ray beyond deg_dcap takes the reference walk */
    uint32_t    ndanger;
    double      danger[16][6];
} lh_bvh_t;
#define LH_DANGER_MAX   16u
#define LH_DANGER_ALL   0u             /* zero-initialised state = the safe one */
#define LH_DEG_DCAP_ALL 1024.0         /* = lh_bvh.c LH_DEG_DCAP: beyond it the triangles LEFT OUT of the tree (v1 == v2, short) are no longer provably missed */

/* ---- triangles whose determinant is the reference's rounding noise (lh_bvh.c prep_part, lh_build.hip k_prim_boxes, lh_commit.hip k_danger_scan) ----
 * The reference's a = e1 . (dir x e2) = dir . n is computed in fp64 with ~15 u D s2 of evaluation noise (u = 2^-53, D = max |dir_k|,
 * s2 = |e1|_1 |e2|_1).  For a triangle whose normal is no larger than that noise by many orders -- three points on a line, exactly or up to
 * the rounding of their coordinates (|n| ~ u s2 |P| / |e|: a collinear triangle far from the origin), a sliver of aspect below 1e-9 -- u, v
 * and t are then noise as well: the reference reports "hits" on rays that merely reach the triangle's leaf in ITS tree, at a t that need
 * not lie inside the triangle's box.  Such a triangle is vouched for by the traversal tree only for rays the reference cannot accept on it:
 * |a| <= D (|n_computed|_1 + 6 u s2) + 15 u D s2 < D (|n|_1 + 3e-15 s2) stays below the reference's 1e-14 (bvh.c:754) while
 * D <= 0.9e-14 / (|n|_1 + 3e-15 s2).  Returns the reciprocal of that bound -- never less than s2, rounds 5-6's bound for max |n_k| <= 8.9e-16 s2,
 * which the formula would relax to 1.6 / s2 -- or 0: the triangle's determinant carries at least nine digits.
 * Round 5: max |n_k| <= 8.9e-16 s2 only.  Round 6's fourth fuzz campaign (tools/fuzz_parity.py seeds 661 / 662, kind 9): collinear triangles
 * of |n| = 3e-15 s2 (coordinates of 8 and 9 600 units, edges of 0.1 and 100) were hit by the reference 5e-6 in t BEFORE their box, behind
 * a nearer triangle's hit: 3 of 60 000 and 1 of 60 000 rays.  1e-9: the sliver scenes of the same fuzzer (|n| >= 2e-9 s2) have always been equal. */
#define LH_ZERO_AREA_REL 1.0e-9
#if defined(__HIPCC__)
__host__ __device__ __forceinline__
#else
static inline
#endif
double lh_zero_area_weight(const double *a, const double *b, const double *c)
{
    const double e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2], e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    const double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    const double s2 = (fabs(e1x) + fabs(e1y) + fabs(e1z)) * (fabs(e2x) + fabs(e2y) + fabs(e2z));
    const double m = fmax(fabs(nx), fmax(fabs(ny), fabs(nz)));
    if (!(m <= LH_ZERO_AREA_REL * s2)) return 0.0;
    {
        const double w = ((fabs(nx) + fabs(ny) + fabs(nz)) + 3.0e-15 * s2) / 0.9e-14;
        return w > s2 ? w : s2;
    }
}

typedef struct lh_mesh_view {
    uint32_t        npositions;
    const double   *positions;       /* first component of position 0       */
    size_t          stride_bytes;    /* 24 for xyz, 32 for lucille's double[4] */
    uint32_t        nindices;
    const uint32_t *indices;
} lh_mesh_view_t;

/* returns 0 on success, -1 on bad input / out of memory, -2 on a NaN / infinite / > 1e30 vertex coordinate */
int  lh_bvh_build(lh_bvh_t *out, const lh_mesh_view_t *meshes, uint32_t nmeshes,
                  int nthreads);
int  lh_bvh_build_hook(lh_bvh_t *out, const lh_mesh_view_t *meshes, uint32_t nmeshes, int nthreads,
                       void (*after_flatten)(void *), void *hook_arg);
int  lh_bvh_flatten(lh_bvh_t *out, const lh_mesh_view_t *meshes, uint32_t nmeshes, int nthreads);   /* primitives only, no tree */
int  lh_bvh_ensure_q8(lh_bvh_t *bvh);      /* builds q8nodes on first use; 0 / -1 */
void lh_bvh_release(lh_bvh_t *bvh);

#ifdef __cplusplus
}
#endif
#endif
