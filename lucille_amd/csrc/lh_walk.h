/*
 * lh_walk.h -- device-side pieces of the traversal kernels (lh_kernels.hip): the exact-hit record, the
 * reference's tie rule, the fp64 resolve of one candidate, the per-lane walk state and the 16-bit-grid slab test.
 * Device code only (included inside the including file's anonymous namespace).
 */
#ifndef LH_WALK_H
#define LH_WALK_H

constexpr int   kDone  = (int)0x80000000; /* stack-bottom sentinel == LH_REF_EMPTY */
constexpr int   kPend  = 4;

struct Best {            /* exact (fp64) closest hit so far */
    double   t, u, v;
    uint32_t prim;
    uint32_t frag;       /* bit 0: this hit is within rounding reach of a box face of the reference's tree
                          * (lh_hit_fragile); bit 1 (sticky): two different triangles at almost equal t */
};

/* The reference's winner between two primitives hit at bit-equal t (lh_refbvh.c):
 * same leaf -> the later triangle (bvh.c:780 rejects only t > t_best); different leaves ->
 * the leaf the reference visits first (bvh.c:850 strict <), i.e. the one under
 * child[dir_sign[axis0]] of their lowest common ancestor (bvh.c:1080,1171-1178). */
__device__ __noinline__ bool tie_takes_new(const lh_dev_scene_t &sc, uint32_t pnew, uint32_t pold,
                                           double dx, double dy, double dz)
{
    if (!sc.ref_lca) return pnew > pold;          /* no reference-order tree: documented fallback */
    const uint2 *lp = (const uint2 *)sc.prim_leafpos;
    const int4 *nd = (const int4 *)sc.ref_lca;
    const uint2 a = lp[pnew], b = lp[pold];
    if (a.x == b.x) return a.y > b.y;
    int ca = (int)a.x, cb = (int)b.x;
    int4 na = nd[ca], nb = nd[cb];
    while (na.y > nb.y) { ca = na.x; na = nd[ca]; }
    while (nb.y > na.y) { cb = nb.x; nb = nd[cb]; }
    while (na.x != nb.x) { ca = na.x; na = nd[ca]; cb = nb.x; nb = nd[cb]; }
    const int4 l = nd[na.x];
    const int order = (l.z == 0 ? dx : (l.z == 1 ? dy : dz)) < 0.0 ? 1 : 0;
    const int first_child = order == 0 ? l.w : l.w + 1;     /* children are allocated adjacently */
    return first_child == ca;
}

/* resolve one queued candidate against the running exact best */
__device__ __forceinline__ void resolve(const lh_dev_scene_t &sc, uint32_t prim,
                                        double ox, double oy, double oz,
                                        double dx, double dy, double dz, Best &b)
{
    double t, u, v;
    const double *tv = (const double *)sc.tri64 + 9 * (size_t)prim;
    if (lh_exact_isect(tv, ox, oy, oz, dx, dy, dz, &t, &u, &v)) {
        bool take = t < b.t;
        if (!take && t == b.t && b.prim != LH_MISS_PRIM && prim != b.prim) take = tie_takes_new(sc, prim, b.prim, dx, dy, dz);
        /* almost-equal t of two triangles: which one the reference keeps can hinge on one box test */
        if (b.prim != LH_MISS_PRIM && prim != b.prim && t != b.t && fabs(t - b.t) <= LH_FRAGILE_REL * fabs(t)) b.frag |= 2u;
        if (take && t < LH_T_INF) {
            /* the fragility bit is evaluated HERE, for every hit that is taken, although only the last one's counts: evaluating it
             * once in finish() (70 of this function's 175 instructions, per candidate slot any lane of the wave had filled) needs the
             * kept triangle's corners again -- one more dependent load at the end of every regroup -- and lost on the same box:
             * S-soup-1M 2 245 -> 2 219 Mrays/s, the path-traced config-4 frame 116.9 -> 118.9 ms (r05, tools/experiments/ab_frames.py) */
            b.t = t; b.u = u; b.v = v; b.prim = prim;
            b.frag = (b.frag & 2u) | (uint32_t)lh_hit_fragile(tv, ox, oy, oz, dx, dy, dz, t);
        }
    }
}

struct Lane {
    lh_ray32_t r;          /* fp32 ray + slab/filter constants */
    uint32_t sh[3];        /* 16 where the direction is negative: rotate lo|hi<<16 into (near, far) */
    float tb;              /* culling bound (fp32, rounded up) */
    int   cur, sp;         /* traversal cursor, stack pointer  */
    uint32_t p0, p1, p2, p3;   /* pending fp64 candidates      */
    int   np;
    bool  certain;         /* any-hit: a certain fp32 hit was found */
    bool  over;            /* 8-wide walk: the LDS stack would overflow; the ray goes to the reference walk */
};

__device__ __forceinline__ void lane_init(Lane &L, const lh_dev_scene_t &sc,
                                          double ox, double oy, double oz,
                                          double dx, double dy, double dz)
{
    lh_ray_setup(&L.r, ox, oy, oz, dx, dy, dz, sc.scene_r);
    lh_ray_setup_grid(&L.r, sc.grid_lo, sc.grid_step, sc.scene_r);
    L.sh[0] = L.r.ngx ? 16u : 0u; L.sh[1] = L.r.ngy ? 16u : 0u; L.sh[2] = L.r.ngz ? 16u : 0u;
    L.tb = 1.0e38f;
    L.cur = 0; L.sp = 1;
    L.p0 = L.p1 = L.p2 = L.p3 = LH_MISS_PRIM; L.np = 0;
    L.certain = false; L.over = false;
}


/* A ray the traversal tree cannot answer by itself: the tree leaves out zero-area triangles of class 2 (v1 == v2; lh_bvh.c
 * tri_dead_class), whose fp64 determinant in the reference is provably below its 1e-14 only while every direction component
 * stays below sc.deg_dcap.  Beyond that (unnormalised directions are legal, ray.h:22-68) the reference's own walk on its own
 * tree -- which holds every triangle -- decides: the callers mark the ray as a fragile hit before it takes a step, and the
 * fix-up machinery (LH_Q_REF / LH_PRIM_RETRACE) does the rest.  deg_dcap is INFINITY for scenes without such triangles. */
/* Round 6: a cap that comes from zero-area triangles which STAY in the tree (deg_dcap = 1 / s2 < LH_DEG_DCAP_ALL: a numerically
 * collinear triangle with |e1|_1 |e2|_1 = s2; in a scene modelled in millimetres ANY such sliver pulls the cap below 1, and then
 * every unit-length ray would take the single-lane reference walk: ADVICE r05) concerns only rays that can REACH such a triangle in
 * the reference's walk, i.e. that hit the box of its leaf in lucille's own tree (the child box its parent holds: what
 * test_ray_node tests, bvh.c:938-1083).  The kernels ask ONE box -- the union of the listed leaves' boxes, rounded outward onto
 * the scene's 16-bit grid (lh_commit.hip lh_danger_scan) -- through slab_w, the conservative test every node's box goes through
 * (outward box, per-ray slack): a superset of the rays that reach one of those leaves (the leaves' ancestors are not asked).
 * `L` is the lane right after lane_init.  Why the grid form: it reads only what the walk keeps alive anyway (the lane's q*
 * constants).  The same test in fp64, or over the list behind a call, cost the persistent walk 16 bytes of scratch per lane in
 * every instantiation; through lh_slab (the fp32 box form) it kept nine of lane_init's values alive that the 4-wide walk
 * otherwise never computes: config 4 and 5 +1 % (profiles/r06_ab_variants.txt). */
__device__ __forceinline__ bool slab_w(const Lane &L, uint32_t wx, uint32_t wy, uint32_t wz, float &tn_out);
__device__ __forceinline__ bool ray_needs_ref_walk(const lh_dev_scene_t &sc, const Lane &L, double dx, double dy, double dz)
{
    const double D = fmax(fabs(dx), fmax(fabs(dy), fabs(dz)));
    if (!((D > (double)sc.deg_dcap) & (sc.ref_nodes != NULL))) return false;
    if (sc.ndanger == LH_DANGER_ALL || D > LH_DEG_DCAP_ALL) return true;
    float tn;
    return slab_w(L, sc.danger[0], sc.danger[1], sc.danger[2], tn);
}
/* a negative culling bound: the root's children are all missed (slab_w: 0 <= tn <= tf <= tb fails), the ray is finished after one
 * node step like any other -- no special case in the walks' control flow (a ray that is idle before its first step cost the
 * dump kernel 80 bytes of scratch per lane) -- and retires as a fragile hit */
#define LH_FORCE_REF_WALK(L, best) do { (L).tb = -1.0f; (best).prim = 0u; (best).frag = 1u; } while (0)

/* lh_slab_w (lh_filter.h) written for the VALU: per axis one rotate (v_alignbit_b32 by 0 or 16)
 * puts (near, far) into the (low, high) halves, two SDWA converts, two FMAs: 136 VALU ops per
 * 4-wide node step instead of 161 with per-plane selects.  Measured (A/B, 50 M rays): +1.5 %;
 * v_pk_fma_f32 for the two FMAs is 2 % SLOWER, reading the stack top before the slab tests
 * instead of after the pushes changes nothing -- the step is not VALU- or LDS-latency-bound
 * (profiles/README.md, r01d). */
__device__ __forceinline__ bool slab_w(const Lane &L, uint32_t wx, uint32_t wy, uint32_t wz, float &tn_out)
{
    const uint32_t sx = __builtin_amdgcn_alignbit(wx, wx, L.sh[0]);
    const uint32_t sy = __builtin_amdgcn_alignbit(wy, wy, L.sh[1]);
    const uint32_t sz = __builtin_amdgcn_alignbit(wz, wz, L.sh[2]);
    const float tn = fmaxf(fmaxf(fmaf((float)(sx & 0xffffu), L.r.qax, L.r.qbnx), fmaf((float)(sy & 0xffffu), L.r.qay, L.r.qbny)),
                           fmaxf(fmaf((float)(sz & 0xffffu), L.r.qaz, L.r.qbnz), 0.0f));
    const float tf = fminf(fminf(fmaf((float)(sx >> 16), L.r.qax, L.r.qbfx), fmaf((float)(sy >> 16), L.r.qay, L.r.qbfy)),
                           fminf(fmaf((float)(sz >> 16), L.r.qaz, L.r.qbfz), L.tb));
    tn_out = tn;
    return tn <= tf;
}

/* one triangle record through the fp32 filter (lh_tri_filter): the bookkeeping every walk shares.
 * Rejected: nothing.  Certain hit: ends an any-hit ray at once, shrinks the closest-hit culling bound.
 * Anything not rejected joins the pending list (resolved in fp64 now if the list is full).
 * Returns true when the ray is finished (any-hit only). */
template <bool ANYHIT, bool COUNT, class RayLoad>
__device__ __forceinline__ bool tri_step_g(Lane &L, const lh_dev_scene_t &sc, float v0x, float v0y, float v0z,
                                           float e1x, float e1y, float e1z, float e2x, float e2y, float e2z,
                                           float ne1, float ne2, uint32_t prim, RayLoad ray, Best &best, uint32_t &c_exact)
{
    float t_hi;
    const int cls = lh_tri_filter(&L.r, v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z, ne1, ne2, L.tb, &t_hi);
    if (cls == LH_TRI_REJECT) return false;
    const bool sure = (cls == LH_TRI_CERTAIN);
    if (ANYHIT && sure) { L.certain = true; return true; }
    if (sure) L.tb = fminf(L.tb, t_hi);
    if (L.np == kPend) {
        /* pending list full: resolve it now (rare) */
        if (COUNT) c_exact += kPend;
        double ox, oy, oz, dx, dy, dz;
        ray(ox, oy, oz, dx, dy, dz);
        resolve(sc, L.p0, ox, oy, oz, dx, dy, dz, best);
        resolve(sc, L.p1, ox, oy, oz, dx, dy, dz, best);
        resolve(sc, L.p2, ox, oy, oz, dx, dy, dz, best);
        resolve(sc, L.p3, ox, oy, oz, dx, dy, dz, best);
        L.np = 0;
        if (ANYHIT && best.prim != LH_MISS_PRIM) return true;
    }
    L.p3 = L.p2; L.p2 = L.p1; L.p1 = L.p0; L.p0 = prim; L.np++;
    return false;
}

/* the same with the fp64 ray held in registers (the lane walks) */
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ bool tri_step(Lane &L, const lh_dev_scene_t &sc, float v0x, float v0y, float v0z,
                                         float e1x, float e1y, float e1z, float e2x, float e2y, float e2z,
                                         float ne1, float ne2, uint32_t prim,
                                         double ox, double oy, double oz, double dx, double dy, double dz,
                                         Best &best, uint32_t &c_exact)
{
    return tri_step_g<ANYHIT, COUNT>(L, sc, v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z, ne1, ne2, prim,
                                     [=](double &a, double &b, double &c, double &d, double &e, double &f) { a = ox; b = oy; c = oz; d = dx; e = dy; f = dz; },
                                     best, c_exact);
}

/* (Round 5 also tried ONE candidate per regroup in the persistent walk -- a lane with more stays idle, unretired, and takes its next
 * one at the next regroup beside the other lanes' first, so the second to fourth test never run for a lane or two: the same
 * records, S-soup-1M and the config-5 AO frame unchanged, the path-traced config-4 frame 111.2 -> 113.1 ms: the idle lanes
 * cost more than the tests saved.  tools/experiments/ab_frames.py.) */
/* the end of a ray's walk: its unresolved candidates through the fp64 test (bvh.c:730-791 order).  skip: a primitive that
 * is known not to be hit (the triangle a flat-shaded AO ray starts on, lh_ao.h), or LH_MISS_PRIM */
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ void finish(Lane &L, const lh_dev_scene_t &sc,
                                       double ox, double oy, double oz,
                                       double dx, double dy, double dz, Best &best,
                                       uint32_t &c_exact, const uint32_t skip = LH_MISS_PRIM)
{
    if (ANYHIT && (L.certain || best.prim != LH_MISS_PRIM)) return;
    if (L.np > 0 && L.p0 != skip) { if (COUNT) c_exact++; resolve(sc, L.p0, ox, oy, oz, dx, dy, dz, best); }
    if (L.np > 1 && L.p1 != skip) { if (COUNT) c_exact++; resolve(sc, L.p1, ox, oy, oz, dx, dy, dz, best); }
    if (L.np > 2 && L.p2 != skip) { if (COUNT) c_exact++; resolve(sc, L.p2, ox, oy, oz, dx, dy, dz, best); }
    if (L.np > 3 && L.p3 != skip) { if (COUNT) c_exact++; resolve(sc, L.p3, ox, oy, oz, dx, dy, dz, best); }
    L.np = 0;
}

#endif
