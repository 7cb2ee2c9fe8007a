/*
 * lh_reftrace.h -- the reference's own ray query, step for step, on the reference-order tree
 * (lh_refbvh.c), and the test that says when it is needed.  Shared host/device source, fp64,
 * no FMA contraction.
 *
 * Why it exists.  The fast path returns the closest triangle among ALL triangles that pass
 * the reference's triangle_isect.  The reference returns the closest among the triangles its
 * traversal REACHES, and its box test (test_ray_aabb, bvh.c:869-936: exact fp64, margins of
 * 1e-14 relative) can fail, by rounding, for a hit point that lies on an edge or corner of a
 * box of ITS tree, or enter a box "after" the hit it contains (tmin >= t).  Such a hit is then
 * missed by the reference (a ray aimed exactly at a vertex of an isolated triangle is the
 * typical case) although its own triangle test accepts it.  Boxes are nested, so a hit point
 * on a face of an ancestor box is also on a face of its own triangle's bounding box:
 * lh_hit_fragile() flags exactly those hits (plus t ~ 0 and two different triangles at
 * almost-but-not-equal t), and flagged rays are re-traced by lh_ref_trace(), which IS the
 * reference algorithm: ri_bvh_intersect (bvh.c:430-542), bvh_traverse (:1092-1188),
 * test_ray_node (:938-1083), test_ray_aabb (:869-936), bvh_intersect_leaf_node (:793-864),
 * triangle_isect (:730-791), on lucille's own tree.  Everything else stays on the fast path:
 * a hit strictly inside every box on its path is reached by the reference too.
 */
#ifndef LH_REFTRACE_H
#define LH_REFTRACE_H

#include <math.h>
#include <stdint.h>

#include "lh_filter.h"
#include "lh_refbvh.h"

#define LH_PRIM_RETRACE 0xFFFFFFFEu   /* output marker: this ray goes through lh_ref_trace */
#define LH_OCC_RETRACE  2u
#define LH_FRAGILE_REL  1.0e-10       /* ~1e6 x the rounding it has to cover */

/* is the hit (t along org + t*dir, triangle tv = 9 doubles) within rounding reach of a face of the
 * triangle's own box on an axis where the box has thickness, or at t ~ 0?  (axes where the triangle
 * is flat: both planes of one axis go through the same monotone rounding and cannot cross) */
LH_HD int lh_hit_fragile(const double *tv, double ox, double oy, double oz,
                         double dx, double dy, double dz, double t)
{
    LH_NO_CONTRACT
    const double o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
    int k, fragile = 0;
    for (k = 0; k < 3; k++) {
        const double a = tv[k], b = tv[3 + k], c = tv[6 + k];
        const double lo = fmin(a, fmin(b, c)), hi = fmax(a, fmax(b, c));
        const double step = t * d[k], x = o[k] + step;
        const double scale = fabs(o[k]) + fabs(step) + fabs(lo) + fabs(hi);
        const double delta = LH_FRAGILE_REL * scale;
        if (hi - lo > delta && (x - lo <= delta || hi - x <= delta)) fragile = 1;
        if (fabs(step) > delta) fragile |= 2;          /* bit 1: the hit is away from the origin on some axis */
    }
    if ((fragile & 1) || !(fragile & 2)) return 1;
    {   /* a determinant that is rounding noise (zero-area or edge-on triangle): the accepted t, u, v are
         * noise too, and what the reference ends up with depends on its visiting order */
        const double e1[3] = {tv[3] - tv[0], tv[4] - tv[1], tv[5] - tv[2]}, e2[3] = {tv[6] - tv[0], tv[7] - tv[1], tv[8] - tv[2]};
        const double px = d[1] * e2[2] - d[2] * e2[1], py = d[2] * e2[0] - d[0] * e2[2], pz = d[0] * e2[1] - d[1] * e2[0];
        const double a = e1[0] * px + e1[1] * py + e1[2] * pz;
        const double n1 = fabs(e1[0]) + fabs(e1[1]) + fabs(e1[2]), n2 = fabs(e2[0]) + fabs(e2[1]) + fabs(e2[2]);
        const double nd = fabs(d[0]) + fabs(d[1]) + fabs(d[2]);
        if (!(fabs(a) > 1.0e-9 * n1 * n2 * nd)) return 1;
    }
    return 0;
}

typedef struct lh_refray {
    double org[3], dir[3], inv[3];
    int    sg[3];
} lh_refray_t;

/* bvh.c:473-497.  (Its |dir.y| <= 1e-14 branch leaves invdir[1] unset; such rays are outside the
 * contract; here they get the value the code evidently meant.) */
LH_HD void lh_ref_ray_setup(lh_refray_t *r, double ox, double oy, double oz, double dx, double dy, double dz)
{
    LH_NO_CONTRACT
    int k;
    r->org[0] = ox; r->org[1] = oy; r->org[2] = oz;
    r->dir[0] = dx; r->dir[1] = dy; r->dir[2] = dz;
    for (k = 0; k < 3; k++) {
        r->sg[k] = (r->dir[k] < 0.0) ? 1 : 0;
        r->inv[k] = (fabs(r->dir[k]) > 1.0e-14) ? 1.0 / r->dir[k]
                                                 : ((r->dir[k] < 0.0) ? -1.7976931348623157e308 : 1.7976931348623157e308);
    }
}

/* test_ray_aabb, bvh.c:869-936: box = bmin xyz, bmax xyz */
LH_HD int lh_ref_ray_aabb(const double *box, const lh_refray_t *r, double *tmin_out)
{
    LH_NO_CONTRACT
    const double nx = r->sg[0] ? box[3] : box[0], fx = r->sg[0] ? box[0] : box[3];
    const double ny = r->sg[1] ? box[4] : box[1], fy = r->sg[1] ? box[1] : box[4];
    const double nz = r->sg[2] ? box[5] : box[2], fz = r->sg[2] ? box[2] : box[5];
    const double tnx = (nx - r->org[0]) * r->inv[0], tfx = (fx - r->org[0]) * r->inv[0];
    const double tny = (ny - r->org[1]) * r->inv[1], tfy = (fy - r->org[1]) * r->inv[1];
    const double tnz = (nz - r->org[2]) * r->inv[2], tfz = (fz - r->org[2]) * r->inv[2];
    double tmin = (tnx > tny) ? tnx : tny, tmax = (tfx < tfy) ? tfx : tfy;
    tmin = (tmin > tnz) ? tmin : tnz;
    tmax = (tmax < tfz) ? tmax : tfz;
    *tmin_out = tmin;
    return (tmax > 0.0) && (tmin <= tmax);
}

/* the whole query; tri64 = 9 doubles per primitive id.  Returns 1 on a hit. */
LH_HD int lh_ref_trace(const lh_refnode_t *nodes, const uint32_t *leaf_prims, const double *tri64, int empty,
                       const double bmin[3], const double bmax[3],
                       double ox, double oy, double oz, double dx, double dy, double dz,
                       uint32_t *prim_out, double *t_out, double *u_out, double *v_out)
{
    LH_NO_CONTRACT
    lh_refray_t r; double scene[6], tmin; int stack[104], depth = 0, node = 0, k;
    double bt = 1.0e38, bu = 0.0, bv = 0.0; uint32_t bprim = 0xFFFFFFFFu;      /* bvh.c:1111-1115 */
    *prim_out = bprim; *t_out = bt; *u_out = 0.0; *v_out = 0.0;
    if (empty) return 0;                                                        /* bvh.c:446-449 */
    lh_ref_ray_setup(&r, ox, oy, oz, dx, dy, dz);
    for (k = 0; k < 3; k++) { scene[k] = bmin[k]; scene[3 + k] = bmax[k]; }
    if (!lh_ref_ray_aabb(scene, &r, &tmin)) return 0;                           /* bvh.c:521-526 */
    for (;;) {
        const lh_refnode_t *nd = &nodes[node];
        if (nd->is_leaf) {                                                      /* bvh.c:793-864 */
            double lt = 1.0e38, lu = 0.0, lv = 0.0; uint32_t lprim = 0, q; int any = 0;
            for (q = 0; q < nd->count; q++) {
                const uint32_t p = leaf_prims[nd->first + q]; double t, u, v;
                if (lh_exact_isect(tri64 + 9 * (size_t)p, ox, oy, oz, dx, dy, dz, &t, &u, &v) && !(t > lt)) {
                    lt = t; lu = u; lv = v; lprim = p; any = 1;                 /* bvh.c:780-789: last of equal t wins */
                }
            }
            if (any && lt < bt) { bt = lt; bu = lu; bv = lv; bprim = lprim; }   /* bvh.c:850: strict */
            if (depth < 1) break;
            node = stack[--depth];
        } else {                                                                /* bvh.c:1038-1083 */
            double t0, t1;
            const int h0 = lh_ref_ray_aabb(nd->box[0], &r, &t0) && (t0 < bt);
            const int h1 = lh_ref_ray_aabb(nd->box[1], &r, &t1) && (t1 < bt);
            if (!h0 && !h1) { if (depth < 1) break; node = stack[--depth]; }
            else if (h0 && !h1) node = nd->child[0];
            else if (!h0) node = nd->child[1];
            else {
                const int order = r.sg[nd->axis];                               /* bvh.c:1080 */
                if (depth < 103) stack[depth++] = nd->child[1 - order];
                node = nd->child[order];
            }
        }
    }
    *prim_out = bprim; *t_out = bt; *u_out = bu; *v_out = bv;
    return bt < 1.0e38;                                                         /* bvh.c:1187 */
}

#endif
