/*
 * lh_filter.h -- the arithmetic of the hot path, written once.
 *
 * Included by lh_kernels.hip (device code, gfx950) and by the CPU model under
 * tests/cpu_model/ (plain C, used only by the not-gpu tests to check on the
 * host that the fp32 filter is conservative with respect to the fp64 oracle).
 * Valid C99 and C++.
 *
 *   lh_ray_setup     per-ray constants for the conservative slab test
 *   lh_slab          entry/exit of one fp32 box, widened by the ray's slack
 *   lh_tri_filter    fp32 Moeller-Trumbore with tolerances: reject / candidate
 *                    / certain hit
 *   lh_exact_isect   fp64 Moeller-Trumbore in the reference's operation order
 *                    (src/render/bvh.c:730-791), no FMA contraction
 *
 * Why the filter is conservative (DESIGN.md section 4 has the derivation):
 * rounding the ray (org, dir) and the triangle origin v0 to fp32 moves every
 * point of the ray/triangle configuration by at most
 *     pos_err <= c * 2^-24 * (|org|_inf + R),      R = max |scene coordinate|
 * and every fp32 operation adds a relative 2^-24.  Both are folded into
 *   - slab slack    s_k  = KBOX * 2^-24 * (|org|_inf + R) * |1/dir_k|
 *   - MT tolerances tolu = g*|d|*|e2|, tolv = g*|d|*|e1|, tolt = g*|e1|*|e2|,
 *                   g    = KTRI * 2^-24 * (|org|_inf + R) * |1/det|
 * with KBOX = 16 and KTRI = 32, several times the worst-case constants.
 */
#ifndef LH_FILTER_H
#define LH_FILTER_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LH_HD __host__ __device__ __forceinline__
#else
#define LH_HD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define LH_RCP(a) __builtin_amdgcn_rcpf(a)   /* v_rcp_f32, 1 ulp */
#else
#define LH_RCP(a) (1.0f / (a))
#endif

#define LH_EPS24 5.9604645e-8f
#define LH_KBOX  16.0f
#define LH_KTRI  32.0f

typedef struct lh_ray32 {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;                 /* 1/dir (|dir_k| clamped to 1e-30)   */
    float cnx, cny, cnz;              /* -org*idir - slack                   */
    float cfx, cfy, cfz;              /* -org*idir + slack                   */
    /* 16-bit grid nodes: t = q*qa + qbn/qbf  (qa = step*idir, qb = (grid_lo-org)*idir -/+ slack) */
    float qax, qay, qaz, qbnx, qbny, qbnz, qbfx, qbfy, qbfz;
    float keps;                       /* KTRI*2^-24*(|org|_inf+R)            */
    float dn;                         /* |dir|_2 rounded up                  */
    int   ngx, ngy, ngz;              /* dir_k < 0                           */
} lh_ray32_t;

LH_HD float lh_safe_dir(float d)
{
    return (fabsf(d) < 1e-30f) ? copysignf(1e-30f, d) : d;
}

typedef struct lh_grid { float lo[3], step[3]; } lh_grid_t;

/* grid constants; scene_r as in lh_ray_setup.  The grid adds one rounding of (grid_lo-org)
 * and of step*idir per axis: both are inside the KBOX margin. */
LH_HD void lh_ray_setup_grid(lh_ray32_t *r, const float glo[3], const float gstep[3], float scene_r)
{
    const float omax = fmaxf(fabsf(r->ox), fmaxf(fabsf(r->oy), fabsf(r->oz)));
    const float pe = LH_KBOX * LH_EPS24 * (omax + scene_r);
    const float sx = pe * fabsf(r->ix), sy = pe * fabsf(r->iy), sz = pe * fabsf(r->iz);
    r->qax = gstep[0] * r->ix; r->qay = gstep[1] * r->iy; r->qaz = gstep[2] * r->iz;
    {
        const float bx = (glo[0] - r->ox) * r->ix, by = (glo[1] - r->oy) * r->iy, bz = (glo[2] - r->oz) * r->iz;
        r->qbnx = bx - sx; r->qbfx = bx + sx;
        r->qbny = by - sy; r->qbfy = by + sy;
        r->qbnz = bz - sz; r->qbfz = bz + sz;
    }
}

LH_HD void lh_ray_setup(lh_ray32_t *r, double ox, double oy, double oz,
                        double dx, double dy, double dz, float scene_r)
{
    float omax, scale, pe, sx, sy, sz;
    r->ox = (float)ox; r->oy = (float)oy; r->oz = (float)oz;
    r->dx = (float)dx; r->dy = (float)dy; r->dz = (float)dz;
    r->ix = 1.0f / lh_safe_dir(r->dx);
    r->iy = 1.0f / lh_safe_dir(r->dy);
    r->iz = 1.0f / lh_safe_dir(r->dz);
    omax = fmaxf(fabsf(r->ox), fmaxf(fabsf(r->oy), fabsf(r->oz)));
    scale = omax + scene_r;
    pe = LH_KBOX * LH_EPS24 * scale;
    sx = pe * fabsf(r->ix); sy = pe * fabsf(r->iy); sz = pe * fabsf(r->iz);
    r->cnx = fmaf(-r->ox, r->ix, -sx); r->cfx = fmaf(-r->ox, r->ix, sx);
    r->cny = fmaf(-r->oy, r->iy, -sy); r->cfy = fmaf(-r->oy, r->iy, sy);
    r->cnz = fmaf(-r->oz, r->iz, -sz); r->cfz = fmaf(-r->oz, r->iz, sz);
    r->ngx = r->dx < 0.0f; r->ngy = r->dy < 0.0f; r->ngz = r->dz < 0.0f;
    r->keps = LH_KTRI * LH_EPS24 * scale;
    r->qax = r->qay = r->qaz = 0.0f; r->qbnx = r->qbny = r->qbnz = 0.0f; r->qbfx = r->qbfy = r->qbfz = 0.0f;
    r->dn = sqrtf(fmaf(r->dx, r->dx, fmaf(r->dy, r->dy, r->dz * r->dz))) * 1.000001f;
}

/* conservative slab test of the box [lo,hi] against the ray, clipped to
 * [0, tb].  Hit iff *tn <= *tf.  Conservative form of test_ray_aabb's
 * (tmax > 0) && (tmin <= tmax) plus test_ray_node's tmin < t_best
 * (bvh.c:926,1038-1044). */
LH_HD int lh_slab(const lh_ray32_t *r, float lox, float loy, float loz,
                  float hix, float hiy, float hiz, float tb, float *tn_out)
{
    const float ax = r->ngx ? hix : lox, bx = r->ngx ? lox : hix;
    const float ay = r->ngy ? hiy : loy, by = r->ngy ? loy : hiy;
    const float az = r->ngz ? hiz : loz, bz = r->ngz ? loz : hiz;
    const float tn = fmaxf(fmaxf(fmaf(ax, r->ix, r->cnx), fmaf(ay, r->iy, r->cny)),
                           fmaxf(fmaf(az, r->iz, r->cnz), 0.0f));
    const float tf = fminf(fminf(fmaf(bx, r->ix, r->cfx), fmaf(by, r->iy, r->cfy)),
                           fminf(fmaf(bz, r->iz, r->cfz), tb));
    *tn_out = tn;
    return tn <= tf;
}

/* the same test on a 16-bit grid box in lh_q4node_t's / lh_q8node_t's packing: w = lo | hi << 16 per axis */
LH_HD int lh_slab_w(const lh_ray32_t *r, uint32_t wx, uint32_t wy, uint32_t wz, float tb, float *tn_out)
{
    const uint32_t sx = r->ngx ? ((wx >> 16) | (wx << 16)) : wx;
    const uint32_t sy = r->ngy ? ((wy >> 16) | (wy << 16)) : wy;
    const uint32_t sz = r->ngz ? ((wz >> 16) | (wz << 16)) : wz;
    const float tn = fmaxf(fmaxf(fmaf((float)(sx & 0xffffu), r->qax, r->qbnx), fmaf((float)(sy & 0xffffu), r->qay, r->qbny)),
                           fmaxf(fmaf((float)(sz & 0xffffu), r->qaz, r->qbnz), 0.0f));
    const float tf = fminf(fminf(fmaf((float)(sx >> 16), r->qax, r->qbfx), fmaf((float)(sy >> 16), r->qay, r->qbfy)),
                           fminf(fmaf((float)(sz >> 16), r->qaz, r->qbfz), tb));
    *tn_out = tn;
    return tn <= tf;
}

#define LH_TRI_REJECT    0
#define LH_TRI_CANDIDATE 1   /* cannot be decided in fp32: resolve in fp64   */
#define LH_TRI_CERTAIN   2   /* inside by more than the tolerance: a hit in
                                fp64 as well; *t_hi bounds its t from above */

/* fp32 Moeller-Trumbore on (v0, e1, e2) with tolerances.  tb = current culling
 * bound. */
LH_HD int lh_tri_filter(const lh_ray32_t *r, float v0x, float v0y, float v0z,
                        float e1x, float e1y, float e1z, float e2x, float e2y, float e2z,
                        float ne1, float ne2, float tb, float *t_hi)
{
    const float px = fmaf(r->dy, e2z, -(r->dz * e2y));
    const float py = fmaf(r->dz, e2x, -(r->dx * e2z));
    const float pz = fmaf(r->dx, e2y, -(r->dy * e2x));
    const float a = fmaf(e1x, px, fmaf(e1y, py, e1z * pz));
    const float inva = LH_RCP(a);
    const float sx = r->ox - v0x, sy = r->oy - v0y, sz = r->oz - v0z;
    const float qx = fmaf(sy, e1z, -(sz * e1y));
    const float qy = fmaf(sz, e1x, -(sx * e1z));
    const float qz = fmaf(sx, e1y, -(sy * e1x));
    const float u = fmaf(sx, px, fmaf(sy, py, sz * pz)) * inva;
    const float v = fmaf(qx, r->dx, fmaf(qy, r->dy, qz * r->dz)) * inva;
    const float t = fmaf(e2x, qx, fmaf(e2y, qy, e2z * qz)) * inva;
    const float g = r->keps * fabsf(inva);
    const float tolu = g * r->dn * ne2, tolv = g * r->dn * ne1, tolt = g * ne1 * ne2;
    /* relative uncertainty of the determinant itself: when the fp32 determinant is noise (zero-area or
     * edge-on triangle), so is the reference's -- and ITS noise can clear the |det| > 1e-14 test and
     * yield an accepted "hit" (t = u = v = -0.0 on a triangle with two equal vertices was seen).
     * Nothing can be concluded in fp32: fp64 decides.  (NaN-safe: !(x < c) is true for NaN.) */
    const float rela = (LH_KTRI * LH_EPS24) * ne1 * ne2 * r->dn * fabsf(inva);
    if (!(rela < 0.5f)) { *t_hi = 3.0e38f; return LH_TRI_CANDIDATE; }
    /* NaN-safe: a comparison with NaN is false => not rejected */
    const int reject = (u < -tolu) | (u > 1.0f + tolu) | (v < -tolv) |
                       (u + v > 1.0f + tolu + tolv) | (t < -tolt) | (t - tolt > tb);
    if (reject) return LH_TRI_REJECT;
    {
        /* The reference also drops
         * |det| <= 1e-14 as an ABSOLUTE threshold (bvh.c:754): a hit is only certain when
         * the fp64 determinant (within 25 % of a) clears it, else the fp64 resolve decides. */
        const int sure = (u >= tolu) & (u <= 1.0f - tolu) & (v >= tolv) &
                         (u + v <= 1.0f - tolu - tolv) & (t >= tolt) & (rela < 0.25f) &
                         (fabsf(a) > 2.0e-14f) & (t + tolt < 1.0e37f);
        *t_hi = (t + tolt) * 1.000001f;
        return sure ? LH_TRI_CERTAIN : LH_TRI_CANDIDATE;
    }
}

/* triangle_isect (bvh.c:730-791): identical operation order, IEEE double, no
 * contraction.  tv = 9 doubles v0 v1 v2.  The `t > *t_inout` leg of the
 * reference's last test is applied by the caller (tie rule lives there). */
#if defined(__clang__)
#define LH_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define LH_NO_CONTRACT
#endif

LH_HD int lh_exact_isect(const double *tv, double ox, double oy, double oz,
                         double dx, double dy, double dz,
                         double *t_out, double *u_out, double *v_out)
{
    LH_NO_CONTRACT
    const double v0x = tv[0], v0y = tv[1], v0z = tv[2];
    const double e1x = tv[3] - v0x, e1y = tv[4] - v0y, e1z = tv[5] - v0z;
    const double e2x = tv[6] - v0x, e2y = tv[7] - v0y, e2z = tv[8] - v0z;
    const double px = dy * e2z - dz * e2y;
    const double py = dz * e2x - dx * e2z;
    const double pz = dx * e2y - dy * e2x;
    const double a = e1x * px + e1y * py + e1z * pz;
    double inva, sx, sy, sz, qx, qy, qz, u, v, t;
    if (!(fabs(a) > 1.0e-14)) return 0;
    inva = 1.0 / a;
    sx = ox - v0x; sy = oy - v0y; sz = oz - v0z;
    qx = sy * e1z - sz * e1y;
    qy = sz * e1x - sx * e1z;
    qz = sx * e1y - sy * e1x;
    u = (sx * px + sy * py + sz * pz) * inva;
    v = (qx * dx + qy * dy + qz * dz) * inva;
    t = (e2x * qx + e2y * qy + e2z * qz) * inva;
    if ((u < 0.0) || (u > 1.0)) return 0;
    if ((v < 0.0) || ((u + v) > 1.0)) return 0;
    if (t < 0.0) return 0;
    *t_out = t; *u_out = u; *v_out = v;
    return 1;
}

#endif
