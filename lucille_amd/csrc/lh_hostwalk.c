/*
 * lh_hostwalk.c -- ONE synchronous ray answered on the calling thread, over the product's own host tree.
 *
 * Why it exists: lucille calls accel->intersect for one ray at a time from its
 * shaders (src/render/raytrace.c:31-69; 18 compiled call sites in shader.c, ibl.c, whitted.c).  A GPU answers one ray in
 * ~20 us -- launch, walk, synchronise -- whatever the kernel does: 50 k rays/s per thread, 200 k coalesced over sixteen
 * (DESIGN 11), where the reference's CPU BVH gives 5 M per thread on the same scene.  The batched entry points are where
 * the GPU is; a caller that insists on one ray gets the SAME answer faster from the host copy of the tree the commit built
 * anyway (lh_bvh.c: the 4-wide 16-bit-grid nodes the kernels walk, the fp32 triangle records, the fp64 triangles) --
 * nothing under oracle/, nothing the device path does not also do:
 *
 *   sequential walk over lh_q4node_t, nearest child first (the order of k_trace_small's overflow_walk);
 *   lh_slab_w / lh_tri_filter (lh_filter.h): the fp32 conservative filter, shared source with the kernels;
 *   lh_exact_isect in fp64, the reference's operation order (bvh.c:730-791), for every candidate;
 *   exact-t ties by lucille's own tree (lh_refbvh_tie_winner), fragile hits and rays beyond deg_dcap by the reference's
 *   own walk on that tree (lh_ref_trace) -- as k_fixups / k_coop_walk do on the device.
 *
 * Used by lh_accel_intersect1 (lh_query.hip) for scenes whose trees live on the host (host-built commits).  The records
 * equal the device path's bit for bit (tests/test_gpu_single_ray.py compares both with the oracle).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#include "lh_bvh.h"
#include "lh_refbvh.h"

/* fmaxf / fminf with their IEEE meaning (a NaN operand is ignored), inline: without -ffinite-math-only gcc calls libm for each of
 * the eight in a slab test -- 200 calls per ray, two thirds of this file's time -- and that switch would let the compiler drop the
 * NaN-safe comparisons lh_filter.h relies on */
static inline float hw_maxf(float a, float b) { return (a >= b || b != b) ? a : b; }
static inline float hw_minf(float a, float b) { return (a <= b || b != b) ? a : b; }
#define fmaxf(a, b) hw_maxf((a), (b))
#define fminf(a, b) hw_minf((a), (b))
#include "lh_filter.h"
#include "lh_reftrace.h"

#if defined(__SSE2__) && defined(__FMA__)
#include <immintrin.h>
/* the four slab tests of a 4-wide node at once: lh_slab_w's arithmetic, operation for operation (the same converts, the same FMAs,
 * the same nesting of max / min), one child per SSE lane.  _mm_max_ps / _mm_min_ps treat a NaN operand differently from fmaxf /
 * fminf, so a node with an unordered lane (coordinates near 1e30 times a reciprocal of 1e30) is redone by the scalar test.
 * Returns the hit mask (bit c: child c), entry distances in tn[4]. */
static inline int hw_node4(const lh_ray32_t *r, const lh_q4node_t *n, float tb, float tn_out[4])
{
    const __m128i m16 = _mm_set1_epi32(0xffff);
    const __m128i wx = _mm_setr_epi32((int)n->w[0][0], (int)n->w[1][0], (int)n->w[2][0], (int)n->w[3][0]);
    const __m128i wy = _mm_setr_epi32((int)n->w[0][1], (int)n->w[1][1], (int)n->w[2][1], (int)n->w[3][1]);
    const __m128i wz = _mm_setr_epi32((int)n->w[0][2], (int)n->w[1][2], (int)n->w[2][2], (int)n->w[3][2]);
    const __m128i lx = _mm_and_si128(wx, m16), hx = _mm_srli_epi32(wx, 16);
    const __m128i ly = _mm_and_si128(wy, m16), hy = _mm_srli_epi32(wy, 16);
    const __m128i lz = _mm_and_si128(wz, m16), hz = _mm_srli_epi32(wz, 16);
    const __m128 nx = _mm_cvtepi32_ps(r->ngx ? hx : lx), fx = _mm_cvtepi32_ps(r->ngx ? lx : hx);
    const __m128 ny = _mm_cvtepi32_ps(r->ngy ? hy : ly), fy = _mm_cvtepi32_ps(r->ngy ? ly : hy);
    const __m128 nz = _mm_cvtepi32_ps(r->ngz ? hz : lz), fz = _mm_cvtepi32_ps(r->ngz ? lz : hz);
    const __m128 anx = _mm_fmadd_ps(nx, _mm_set1_ps(r->qax), _mm_set1_ps(r->qbnx)), any_ = _mm_fmadd_ps(ny, _mm_set1_ps(r->qay), _mm_set1_ps(r->qbny)),
                 anz = _mm_fmadd_ps(nz, _mm_set1_ps(r->qaz), _mm_set1_ps(r->qbnz));
    const __m128 afx = _mm_fmadd_ps(fx, _mm_set1_ps(r->qax), _mm_set1_ps(r->qbfx)), afy = _mm_fmadd_ps(fy, _mm_set1_ps(r->qay), _mm_set1_ps(r->qbfy)),
                 afz = _mm_fmadd_ps(fz, _mm_set1_ps(r->qaz), _mm_set1_ps(r->qbfz));
    const __m128 tn = _mm_max_ps(_mm_max_ps(anx, any_), _mm_max_ps(anz, _mm_setzero_ps()));
    const __m128 tf = _mm_min_ps(_mm_min_ps(afx, afy), _mm_min_ps(afz, _mm_set1_ps(tb)));
    /* an unordered plane distance in ANY of the six products: _mm_max_ps / _mm_min_ps return their second operand when one is a NaN,
     * so a NaN in a FIRST operand would be dropped before tn / tf are looked at */
    const __m128 unord = _mm_or_ps(_mm_or_ps(_mm_cmpunord_ps(anx, any_), _mm_cmpunord_ps(anz, afx)), _mm_cmpunord_ps(afy, afz));
    int c, mask;
    if (__builtin_expect(_mm_movemask_ps(unord) != 0 || tb != tb, 0)) {
        mask = 0;
        for (c = 0; c < 4; c++) if (lh_slab_w(r, n->w[c][0], n->w[c][1], n->w[c][2], tb, &tn_out[c])) mask |= 1 << c;
    } else {
        _mm_storeu_ps(tn_out, tn);
        mask = _mm_movemask_ps(_mm_cmple_ps(tn, tf));
    }
    for (c = 0; c < 4; c++) if (n->ref[c] == LH_REF_EMPTY) mask &= ~(1 << c);
    return mask;
}
#else
static inline int hw_node4(const lh_ray32_t *r, const lh_q4node_t *n, float tb, float tn_out[4])
{
    int c, mask = 0;
    for (c = 0; c < 4; c++) if (n->ref[c] != LH_REF_EMPTY && lh_slab_w(r, n->w[c][0], n->w[c][1], n->w[c][2], tb, &tn_out[c])) mask |= 1 << c;
    return mask;
}
#endif

#define HW_MISS   0xFFFFFFFFu
#define HW_T_INF  1.0e38
#define HW_STACK  288            /* 3 per 4-wide level + slack: the deepest tree the builders hand over (LH_COOP_ROWS_MAX) */

typedef struct { double t, u, v; uint32_t prim, frag; } hw_best_t;

static void hw_resolve(const lh_bvh_t *b, const lh_refbvh_t *ref, uint32_t prim, const double o[3], const double d[3], hw_best_t *best)
{
    double t, u, v;
    const double *tv = &b->tri64[prim].v[0][0];
    if (!lh_exact_isect(tv, o[0], o[1], o[2], d[0], d[1], d[2], &t, &u, &v)) return;
    int take = t < best->t;
    /* two different triangles at almost equal t: which one the reference keeps can hinge on one of its box tests */
    if (best->prim != HW_MISS && prim != best->prim && t != best->t && fabs(t - best->t) <= LH_FRAGILE_REL * fabs(t)) best->frag |= 2u;
    if (!take && t == best->t && best->prim != HW_MISS && prim != best->prim) {
        if (ref) { const int sg[3] = { d[0] < 0.0, d[1] < 0.0, d[2] < 0.0 }; take = lh_refbvh_tie_winner(ref, prim, best->prim, sg) == prim; }
        else take = prim > best->prim;                 /* no reference-order tree (LH_REFTREE=0): the documented fallback */
    }
    if (take && t < HW_T_INF) {
        best->t = t; best->u = u; best->v = v; best->prim = prim;
        best->frag = (best->frag & 2u) | (uint32_t)lh_hit_fragile(tv, o[0], o[1], o[2], d[0], d[1], d[2], t);
    }
}

/* lh_walk.h danger_hit on the host: test_ray_aabb (bvh.c:869-936) against the listed leaf boxes, the same fp64 expressions */
static int hw_danger_hit(const lh_bvh_t *b, uint32_t nd, const double o[3], const double d[3])
{
    uint32_t i; int k;
    for (i = 0; i < nd; i++) {
        const double *bx = b->danger[i];
        double tmin = -1.0e308, tmax = 1.0e308;
        for (k = 0; k < 3; k++) {
            if (!(fabs(d[k]) > 1.0e-14)) continue;
            const double inv = 1.0 / d[k];
            const double lo = ((d[k] < 0.0 ? bx[3 + k] : bx[k]) - o[k]) * inv, hi = ((d[k] < 0.0 ? bx[k] : bx[3 + k]) - o[k]) * inv;
            tmin = (tmin > lo) ? tmin : lo; tmax = (tmax < hi) ? tmax : hi;
        }
        if ((tmax > 0.0) && (tmin <= tmax)) return 1;
    }
    return 0;
}

/* closest hit of one ray.  ref: lucille's own tree (or NULL).  Returns 1 on a hit, 0 on a miss (outputs written), -2 when the walk could
 * not be finished here (stack rows) and there is no reference tree to ask. */
int lh_host_walk_closest(const lh_bvh_t *b, const lh_refbvh_t *ref, const double o[3], const double d[3],
                         uint32_t *prim, double *t, double *u, double *v)
{
    hw_best_t best = { HW_T_INF, 0.0, 0.0, HW_MISS, 0u };
    *prim = HW_MISS; *t = HW_T_INF; *u = 0.0; *v = 0.0;
    if (b->ntris == 0) return 0;
    /* a direction component beyond deg_dcap: the traversal tree leaves out zero-area triangles it can vouch for only below it
     * (lh_bvh.c tri_dead_class) -- the reference's own walk decides; where the cap comes from zero-area triangles that stay in the
     * tree, only for rays that hit the box of one of their leaves in lucille's own tree (lh_bvh.h danger, lh_walk.h danger_hit) */
    int refw = 0;
    if (ref) {
        const double D = fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
        if (D > b->deg_dcap) {
            const uint32_t nd = __atomic_load_n(&b->ndanger, __ATOMIC_ACQUIRE);
            refw = (nd == LH_DANGER_ALL || D > LH_DEG_DCAP_ALL) ? 1 : hw_danger_hit(b, nd, o, d);
        }
    }
    if (!refw) {
        lh_ray32_t r; float tb = 1.0e38f, scene_r = 0.0f;
        int32_t stack[HW_STACK]; int sp = 0, k; int32_t cur = 0;
        uint32_t pend[4]; int np = 0, ovf = 0;
        for (k = 0; k < 3; k++) { scene_r = fmaxf(scene_r, fabsf(b->bmin[k])); scene_r = fmaxf(scene_r, fabsf(b->bmax[k])); }
        lh_ray_setup(&r, o[0], o[1], o[2], d[0], d[1], d[2], scene_r);
        lh_ray_setup_grid(&r, b->grid_lo, b->grid_step, scene_r);
        for (;;) {
            if (cur >= 0) {
                const lh_q4node_t *n = &b->q4nodes[cur];
                uint32_t key[4]; int slot[4], nh = 0, c, m; float tn4[4];
                const int hitmask = hw_node4(&r, n, tb, tn4);
                for (c = 0; c < 4; c++) {
                    union { float f; uint32_t w; } cv;
                    if (!(hitmask >> c & 1)) continue;
                    cv.f = tn4[c];                                     /* entry distances are >= 0: their bits order like integers */
                    const uint32_t kc = (cv.w & ~3u) | (uint32_t)c;    /* the kernels' key: distance bits, slot in the two low bits */
                    for (m = nh; m > 0 && key[m - 1] > kc; m--) { key[m] = key[m - 1]; slot[m] = slot[m - 1]; }
                    key[m] = kc; slot[m] = c; nh++;
                }
                for (m = nh - 1; m >= 1; m--) { if (sp < HW_STACK) stack[sp++] = n->ref[slot[m]]; else ovf = 1; }
                if (nh) cur = n->ref[slot[0]];
                else if (sp) cur = stack[--sp];
                else break;
            } else {
                const uint32_t x = ~(uint32_t)cur, first = x >> 2, cnt = (x & 3u) + 1u; uint32_t q;
                for (q = 0; q < cnt; q++) {
                    const lh_tri32_t *T = &b->tri32[first + q]; float t_hi;
                    const int cls = lh_tri_filter(&r, T->v0[0], T->v0[1], T->v0[2], T->e1x, T->e1y, T->e1z, T->e2x, T->e2y, T->e2z, T->ne1, T->ne2, tb, &t_hi);
                    if (cls == LH_TRI_REJECT) continue;
                    if (cls == LH_TRI_CERTAIN) tb = fminf(tb, t_hi);          /* a certain hit shrinks the culling bound */
                    if (np == 4) { for (k = 0; k < 4; k++) hw_resolve(b, ref, pend[k], o, d, &best); np = 0; }
                    pend[np++] = T->prim;
                }
                if (!sp) break;
                cur = stack[--sp];
            }
        }
        for (k = 0; k < np; k++) hw_resolve(b, ref, pend[k], o, d, &best);
        refw = ref && best.prim != HW_MISS && best.frag != 0u;           /* a hit the reference may not reach: its own walk decides */
        /* the private stack was full and children were left out (a tree deeper than HW_STACK covers: no builder of this library makes
         * one today): the walk's answer is not to be trusted -- the reference's own walk decides, or, without its tree, the caller
         * takes the device path (-2), whose cooperative walk has no such limit */
        if (ovf) { if (ref) refw = 1; else return -2; }
    }
    if (refw) {
        uint32_t p; double tt, uu, vv;
        const int hit = lh_ref_trace(ref->nodes, ref->leaf_prims, &b->tri64[0].v[0][0], ref->empty, ref->bmin, ref->bmax,
                                     o[0], o[1], o[2], d[0], d[1], d[2], &p, &tt, &uu, &vv);
        *prim = hit ? p : HW_MISS; *t = tt; *u = uu; *v = vv;
        return hit;
    }
    *prim = best.prim; *t = best.t; *u = best.u; *v = best.v;
    return best.prim != HW_MISS;
}
