/*
 * lh_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for lucille's
 * BVH traversal + ray/triangle intersection hot path.
 *
 * What this replaces (reference, CPU, one ray at a time, all fp64):
 *   ri_bvh_intersect          src/render/bvh.c:430-542
 *   bvh_traverse              src/render/bvh.c:1092-1188
 *   test_ray_node/aabb        src/render/bvh.c:869-1083
 *   bvh_intersect_leaf_node   src/render/bvh.c:793-864
 *   triangle_isect            src/render/bvh.c:730-791
 *
 * Design (see DESIGN.md for the full argument):
 *
 *  - one ray per lane, 64-lane wavefronts, 256-thread workgroups; each lane
 *    owns a column of a [rows][256] int stack in dynamic LDS (bank = lane % 32,
 *    so pushes and pops are conflict-free whatever the per-lane depth);
 *  - the default walk (traverse_spec4) reads 64-byte 4-wide nodes on the scene's
 *    16-bit grid (4 x dwordx4 decide four children) and 48-byte fp32 triangle
 *    records (3 x dwordx4 per test); ray dumps over scenes larger than the
 *    Infinity Cache read 128-byte 8-wide nodes (traverse_spec8: one cache line
 *    decides eight children); one textbook walk (one ray per lane, while-while,
 *    2-wide fp32 nodes: LH_VARIANT_DIRECT) is kept as the in-process reference
 *    the parity tests compare the tuned walks against.  The measured losers of
 *    rounds 1-2 (lean walk, quad-per-ray walk, compressed 8-wide nodes, the
 *    2-wide 16-bit walks) are gone from the product: tools/experiments/README.md;
 *  - traversal and the Moeller-Trumbore test run in fp32 as a CONSERVATIVE
 *    FILTER: boxes are rounded outward at build time, every slab interval is
 *    widened by a per-ray slack that bounds the fp32 perturbation of the ray,
 *    and every barycentric/t comparison carries a per-test tolerance.  A
 *    triangle the filter cannot reject is queued (a 4-deep per-lane pending
 *    list in registers);
 *  - the queued candidates are resolved in fp64 with the reference's exact
 *    operation order and no FMA contraction (exact_isect below == bvh.c:
 *    730-791), wave-coherently when the lane's traversal has finished, so
 *    (prim, t, u, v) are the reference's bits whenever no exact-t tie occurs;
 *  - "certain" fp32 hits (inside by more than the tolerance) shrink the
 *    culling bound for closest-hit and terminate any-hit rays at once, so AO
 *    rays almost never touch fp64;
 *  - the persistent kernel pulls work from a global cursor in wave-private
 *    ranges and refills, ballot/popcount-compacted, just the lanes that have
 *    finished.
 *
 * No MFMA: this is branchy gather work bounded by the memory system.
 * Compiled with -ffp-contract=off; fp32 code uses explicit fmaf().
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/lucille_hip.h"
#include "lh_device.h"
#include "lh_filter.h"
#include "lh_reftrace.h"
#include "lh_ao.h"

namespace {

#include "lh_walk.h"

/* the textbook walk (LH_VARIANT_DIRECT): while-while over the 2-wide fp32 nodes (the SURVEY 8d layout), one ray per lane,
 * no regrouping, no parked leaves.  Kept as the in-process reference the tuned walks are compared against. */
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ void traverse(Lane &L, const lh_dev_scene_t &sc,
                                         int (*stk)[LH_BLOCK], const int tid,
                                         double ox, double oy, double oz,
                                         double dx, double dy, double dz, Best &best,
                                         uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact)
{
    const float4 *__restrict__ tris  = (const float4 *)sc.tri32;

    while (L.cur != kDone) {
        /* ---- inner nodes ------------------------------------------------ */
        while (L.cur >= 0) {
            if (COUNT) c_nodes++;
            const float4 *p = (const float4 *)sc.nodes + 4 * (size_t)L.cur;
            const float4 n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3];
            /* child0: lo (n0.x n0.y n0.z) hi (n0.w n1.x n1.y); child1: lo (n1.z n1.w n2.x) hi (n2.y n2.z n2.w) */
            float tn0, tn1;
            const bool h0 = lh_slab(&L.r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, L.tb, &tn0);
            const bool h1 = lh_slab(&L.r, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, L.tb, &tn1);
            const int r0 = __float_as_int(n3.x), r1 = __float_as_int(n3.y);
            if (h0 | h1) {
                const bool second = h1 && (!h0 || tn1 < tn0);
                L.cur = second ? r1 : r0;
                if (h0 & h1) { stk[L.sp][tid] = second ? r0 : r1; L.sp++; }
            } else {
                L.sp--; L.cur = stk[L.sp][tid];
            }
        }
        if (L.cur == kDone) break;

        /* ---- leaf: fp32 conservative Moeller-Trumbore filter ------------ */
        {
            const uint32_t x = ~(uint32_t)L.cur;
            const uint32_t first = x >> 2, cnt = (x & 3u) + 1u;
            bool finished = false;
            for (uint32_t i = 0; i < cnt; i++) {
                const float4 *tp = tris + 3 * (size_t)(first + i);
                const float4 ta = tp[0], tb_ = tp[1], tc = tp[2];
                if (COUNT) c_tris++;
                if (tri_step<ANYHIT, COUNT>(L, sc, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w, __float_as_uint(tc.y), ox, oy, oz, dx, dy, dz, best, c_exact)) { finished = true; break; }
            }
            if (finished) { L.cur = kDone; break; }
            L.sp--; L.cur = stk[L.sp][tid];
        }
    }
}

/* Speculative walk over the 4-wide 16-bit grid nodes (lh_q4node_t): one 64-byte record --
 * one L2 request -- decides four children.  The node step is branch-free: the four slab
 * tests give (hit, entry distance); ranks by entry distance come from six key comparisons
 * (key = distance bits with the slot number in the two low bits, misses = max); every child
 * writes its reference to the LDS stack at a rank-derived slot (hits: farthest at the bottom,
 * nearest on top; misses: above the new top, i.e. into free space) and the next reference
 * is read back from the new top -- which is the nearest hit, or the previous top when
 * nothing was hit (a pop).  Leaves are parked and tested in batches as in traverse_spec. */
template <bool ANYHIT, bool COUNT, bool GUARD>
__device__ __forceinline__ void traverse_spec4(Lane &L, int &pend, const lh_dev_scene_t &sc,
                                               int (*stk)[LH_BLOCK], const int tid,
                                               double ox, double oy, double oz,
                                               double dx, double dy, double dz, Best &best,
                                               uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact,
                                               const int min_active, const int tri_batch,
                                               uint32_t &c_nslots, uint32_t &c_tslots)
{
    const float4 *__restrict__ tris  = (const float4 *)sc.tri32;
    constexpr int kNoLeaf = 0;

    const int rows = (int)sc.stack_rows;
    for (;;) {
        if (COUNT) { if (__ballot(L.cur >= 0) != 0ull) c_nslots++; }
        /* the step below writes up to slot sp + 3.  rows = 3 * depth + 5 covers every ray of a tree that deep; a deeper
         * tree (an LBVH built on the device over a degenerate distribution) gets 64 rows and a ray that would overrun them
         * is finished by k_overflow_fix with a private stack -- same arithmetic, same answer */
        if (GUARD && L.cur >= 0 && L.sp + 4 > rows) { L.over = true; L.cur = kDone; pend = kNoLeaf; }     /* a separate instantiation: the check costs the path-traced frame 4 % */
        if (L.cur >= 0) {
            const uint4 *p = (const uint4 *)sc.q4nodes + 4 * (size_t)L.cur;
            const uint4 a = p[0], b = p[1], c = p[2], r = p[3];
            if (COUNT) c_nodes++;
            float t0, t1, t2, t3;
            const bool h0 = slab_w(L, a.x, a.y, a.z, t0) & ((int)r.x != kDone);
            const bool h1 = slab_w(L, a.w, b.x, b.y, t1) & ((int)r.y != kDone);
            const bool h2 = slab_w(L, b.z, b.w, c.x, t2) & ((int)r.z != kDone);
            const bool h3 = slab_w(L, c.y, c.z, c.w, t3) & ((int)r.w != kDone);
            /* entry distances are >= 0, so their bit patterns order like unsigned integers */
            const uint32_t k0 = h0 ? ((__float_as_uint(t0) & ~3u) | 0u) : 0xFFFFFFFCu;
            const uint32_t k1 = h1 ? ((__float_as_uint(t1) & ~3u) | 1u) : 0xFFFFFFFDu;
            const uint32_t k2 = h2 ? ((__float_as_uint(t2) & ~3u) | 2u) : 0xFFFFFFFEu;
            const uint32_t k3 = h3 ? ((__float_as_uint(t3) & ~3u) | 3u) : 0xFFFFFFFFu;
            const int b10 = k1 < k0, b20 = k2 < k0, b30 = k3 < k0, b21 = k2 < k1, b31 = k3 < k1, b32 = k3 < k2;
            const int rk0 = b10 + b20 + b30, rk1 = (1 - b10) + b21 + b31;
            const int rk2 = (2 - b20 - b21) + b32, rk3 = 3 - b30 - b31 - b32;
            const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
            const int base = L.sp + nh - 1;
            stk[h0 ? base - rk0 : L.sp + rk0][tid] = (int)r.x;
            stk[h1 ? base - rk1 : L.sp + rk1][tid] = (int)r.y;
            stk[h2 ? base - rk2 : L.sp + rk2][tid] = (int)r.z;
            stk[h3 ? base - rk3 : L.sp + rk3][tid] = (int)r.w;
            L.sp = base;
            const int nxt = stk[base][tid];
            const int popped2 = stk[L.sp - 1][tid];
            const bool is_leaf = (nxt < 0) & (nxt != kDone);
            const bool park = is_leaf & (pend == kNoLeaf);
            pend = park ? nxt : pend;
            L.cur = park ? popped2 : nxt;
            L.sp -= park ? 1 : 0;
        }
        const unsigned long long m_node = __ballot(L.cur >= 0);
        const unsigned long long m_pend = __ballot(pend != kNoLeaf);
        if (m_pend != 0ull && (__popcll(m_pend) >= tri_batch || m_node == 0ull)) {
            if (COUNT) c_tslots++;
            if (pend != kNoLeaf) {
                const uint32_t x = ~(uint32_t)pend;
                const float4 *tp = tris + 3 * (size_t)(x >> 2);
                const float4 ta = tp[0], tb_ = tp[1], tc = tp[2];
                if (COUNT) c_tris++;
                const bool finished = tri_step<ANYHIT, COUNT>(L, sc, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w, __float_as_uint(tc.y), ox, oy, oz, dx, dy, dz, best, c_exact);
                if (finished) { L.cur = kDone; pend = kNoLeaf; }
                else if (x & 3u) pend = (int)~(((x >> 2) + 1u) << 2 | ((x & 3u) - 1u));
                else {
                    const bool waiting = (L.cur < 0) & (L.cur != kDone);
                    pend = waiting ? L.cur : kNoLeaf;
                    if (waiting) { L.sp--; L.cur = stk[L.sp][tid]; }
                }
            }
        }
        const unsigned long long m_work = __ballot((L.cur != kDone) | (pend != kNoLeaf));
        if (__popcll(m_work) < min_active) break;
    }
}

/* The same walk over the 8-wide 16-bit-grid nodes (lh_q8node_t): one 128-byte record -- one cache line -- decides eight
 * children.  No distance sort: slot s has priority s ^ oct (0 = nearest; the builder put the children into octant slots),
 * a hit child's stack slot is its rank among the hits in priority order (popcount of the nearer hits), nearest on top;
 * misses write to a scratch row.  For ray dumps over scenes that do not fit the Infinity Cache: there every record costs a
 * 128-byte line of HBM traffic, and this node uses all of it (S-soup-10M: 40 records per ray instead of 57). */
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ void traverse_spec8(Lane &L, int &pend, const lh_dev_scene_t &sc,
                                               int (*stk)[LH_BLOCK], const int tid,
                                               double ox, double oy, double oz,
                                               double dx, double dy, double dz, Best &best,
                                               uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact,
                                               const int min_active, const int tri_batch,
                                               uint32_t &c_nslots, uint32_t &c_tslots)
{
    const float4 *__restrict__ tris  = (const float4 *)sc.tri32;
    constexpr int kNoLeaf = 0;
    const uint32_t oct = (uint32_t)L.r.ngx | ((uint32_t)L.r.ngy << 1) | ((uint32_t)L.r.ngz << 2);
    const int rows = (int)sc.stack_rows - 1;        /* the last row takes the misses' writes */

    for (;;) {
        if (COUNT) { if (__ballot(L.cur >= 0) != 0ull) c_nslots++; }
        if (sc.stack_guard && L.cur >= 0 && L.sp + 8 > rows) { L.over = true; L.cur = kDone; pend = kNoLeaf; }
        if (L.cur >= 0) {
            const uint4 *p = (const uint4 *)sc.q8nodes + 8 * (size_t)L.cur;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], r0 = p[6], r1 = p[7];
            if (COUNT) c_nodes++;
            float t;
            const bool h0 = slab_w(L, a.x, a.y, a.z, t) & ((int)r0.x != kDone);
            const bool h1 = slab_w(L, a.w, b.x, b.y, t) & ((int)r0.y != kDone);
            const bool h2 = slab_w(L, b.z, b.w, c.x, t) & ((int)r0.z != kDone);
            const bool h3 = slab_w(L, c.y, c.z, c.w, t) & ((int)r0.w != kDone);
            const bool h4 = slab_w(L, d.x, d.y, d.z, t) & ((int)r1.x != kDone);
            const bool h5 = slab_w(L, d.w, e.x, e.y, t) & ((int)r1.y != kDone);
            const bool h6 = slab_w(L, e.z, e.w, f.x, t) & ((int)r1.z != kDone);
            const bool h7 = slab_w(L, f.y, f.z, f.w, t) & ((int)r1.w != kDone);
            /* hits as a bit mask in priority order (bit q: the child visited q-th) */
            const uint32_t pm = ((uint32_t)h0 << (0u ^ oct)) | ((uint32_t)h1 << (1u ^ oct)) | ((uint32_t)h2 << (2u ^ oct)) |
                                ((uint32_t)h3 << (3u ^ oct)) | ((uint32_t)h4 << (4u ^ oct)) | ((uint32_t)h5 << (5u ^ oct)) |
                                ((uint32_t)h6 << (6u ^ oct)) | ((uint32_t)h7 << (7u ^ oct));
            const int base = L.sp + __popc(pm) - 1;
#define LH_PUSH8(S, H, REF) stk[(H) ? base - __popc(pm & ((1u << ((S) ^ oct)) - 1u)) : rows][tid] = (int)(REF)
            LH_PUSH8(0u, h0, r0.x); LH_PUSH8(1u, h1, r0.y); LH_PUSH8(2u, h2, r0.z); LH_PUSH8(3u, h3, r0.w);
            LH_PUSH8(4u, h4, r1.x); LH_PUSH8(5u, h5, r1.y); LH_PUSH8(6u, h6, r1.z); LH_PUSH8(7u, h7, r1.w);
#undef LH_PUSH8
            L.sp = base;
            const int nxt = stk[base][tid];
            const int popped2 = stk[L.sp - 1][tid];
            const bool is_leaf = (nxt < 0) & (nxt != kDone);
            const bool park = is_leaf & (pend == kNoLeaf);
            pend = park ? nxt : pend;
            L.cur = park ? popped2 : nxt;
            L.sp -= park ? 1 : 0;
        }
        const unsigned long long m_node = __ballot(L.cur >= 0);
        const unsigned long long m_pend = __ballot(pend != kNoLeaf);
        if (m_pend != 0ull && (__popcll(m_pend) >= tri_batch || m_node == 0ull)) {
            if (COUNT) c_tslots++;
            if (pend != kNoLeaf) {
                const uint32_t x = ~(uint32_t)pend;
                const float4 *tp = tris + 3 * (size_t)(x >> 2);
                const float4 ta = tp[0], tb_ = tp[1], tc = tp[2];
                if (COUNT) c_tris++;
                const bool finished = tri_step<ANYHIT, COUNT>(L, sc, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w, __float_as_uint(tc.y), ox, oy, oz, dx, dy, dz, best, c_exact);
                if (finished) { L.cur = kDone; pend = kNoLeaf; }
                else if (x & 3u) pend = (int)~(((x >> 2) + 1u) << 2 | ((x & 3u) - 1u));
                else {
                    const bool waiting = (L.cur < 0) & (L.cur != kDone);
                    pend = waiting ? L.cur : kNoLeaf;
                    if (waiting) { L.sp--; L.cur = stk[L.sp][tid]; }
                }
            }
        }
        const unsigned long long m_work = __ballot((L.cur != kDone) | (pend != kNoLeaf));
        if (__popcll(m_work) < min_active) break;
    }
}

/* resolve whatever is still queued; afterwards `best` is the exact answer */

template <bool ANYHIT>
__device__ __forceinline__ void write_out(size_t i, const Lane &L, const Best &best,
                                          uint32_t *__restrict__ prim, double *__restrict__ t,
                                          double *__restrict__ u, double *__restrict__ v,
                                          uint8_t *__restrict__ occ, const bool retrace_on, const bool over_fix = false)
{
    if (over_fix && L.over) {            /* the LDS stack was too short for this ray: k_overflow_fix redoes it */
        if (ANYHIT) occ[i] = (uint8_t)LH_OCC_OVERFLOW; else prim[i] = LH_PRIM_OVERFLOW;
        return;
    }
    /* a hit the reference may not reach goes through the reference's own walk (k_ref_retrace);
     * a certain fp32 hit is strictly inside its triangle, hence inside every box: never fragile */
    const bool retrace = retrace_on && (L.over || (best.prim != LH_MISS_PRIM && best.frag != 0u && !(ANYHIT && L.certain)));
    if (ANYHIT) {
        occ[i] = retrace ? (uint8_t)LH_OCC_RETRACE : ((L.certain || best.prim != LH_MISS_PRIM) ? 1 : 0);
    } else {
        prim[i] = retrace ? LH_PRIM_RETRACE : best.prim; t[i] = best.t; u[i] = best.u; v[i] = best.v;
    }
}

__device__ __forceinline__ void add_counters(unsigned long long *c, uint32_t nodes, uint32_t tris,
                                             uint32_t exact, uint32_t rays)
{
    /* COUNT builds are diagnostic: plain atomics are fine */
    atomicAdd(&c[LH_CNT_NODES], (unsigned long long)nodes);
    atomicAdd(&c[LH_CNT_TRIS], (unsigned long long)tris);
    atomicAdd(&c[LH_CNT_EXACT], (unsigned long long)exact);
    atomicAdd(&c[LH_CNT_RAYS], (unsigned long long)rays);
}

/* ------------------------------------------------------------------------ */
/* LH_VARIANT_DIRECT: one ray per lane, the grid covers the batch           */
/* ------------------------------------------------------------------------ */
template <bool ANYHIT, bool COUNT>
__global__ __launch_bounds__(LH_BLOCK) void k_trace_direct(
    lh_dev_scene_t sc, size_t n, const double *__restrict__ org, const double *__restrict__ dir,
    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
    double *__restrict__ v, uint8_t *__restrict__ occ, unsigned long long *counters)
{
    extern __shared__ int lh_stack_lds[];          /* [stack entries][LH_BLOCK], sized at launch */
    int (*stk)[LH_BLOCK] = (int (*)[LH_BLOCK])lh_stack_lds;
    const int tid = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * LH_BLOCK + tid;
    if (i >= n) return;
    const double ox = org[3 * i], oy = org[3 * i + 1], oz = org[3 * i + 2];
    const double dx = dir[3 * i], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
    Lane L; Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
    uint32_t cn = 0, ct = 0, ce = 0;
    lane_init(L, sc, ox, oy, oz, dx, dy, dz);
    stk[0][tid] = kDone;
    traverse<ANYHIT, COUNT>(L, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce);
    finish<ANYHIT, COUNT>(L, sc, ox, oy, oz, dx, dy, dz, best, ce);
    write_out<ANYHIT>(i, L, best, prim, t, u, v, occ, sc.ref_nodes != NULL);
    if (COUNT) add_counters(counters, cn, ct, ce, 1);
}

/* ------------------------------------------------------------------------ */
/* the persistent kernel: wavefronts with ballot-compacted lane refill      */
/* ------------------------------------------------------------------------ */
/* SRC 0: rays from the fp64 org/dir arrays.  SRC 1 (any-hit): the work items are the ambient-occlusion
 * rays of a tile -- item i = (hit slot i / N, sample i % N) -- generated in the refill by lh_ao.h from the hit
 * record; an occluded ray adds one to its slot's counter, nothing per ray goes through HBM (round 1 wrote and
 * re-read 49 bytes per AO ray: 22 GB per 4096^2 x 64 frame).  A fragile hit (lh_reftrace.h; rare) is queued for
 * the reference's own walk.
 * WALK 3: traverse_spec4 (the default); 8: the same with the stack check (trees whose worst case the LDS rows do not
 * cover); 7: traverse_spec8 (8-wide nodes). */
struct AoSrc {
    const double *hitrec; const unsigned long long *slot_key; unsigned int *occ_count;
    unsigned long long seed; int ntheta, nphi;
    uint32_t *queue, *qcount; uint32_t qcap;
};

template <bool ANYHIT, bool COUNT, int WALK, int SRC>
__device__ __forceinline__ void trace_persist_lane(
    const lh_dev_scene_t &sc, size_t n, const double *__restrict__ org, const double *__restrict__ dir,
    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
    double *__restrict__ v, uint8_t *__restrict__ occ, unsigned long long *counters,
    unsigned long long *cursor, int min_active, int tri_batch, const AoSrc &ao, int *lds)
{
    int (*stk)[LH_BLOCK] = (int (*)[LH_BLOCK])lds;
    const int tid = threadIdx.x;
    uint32_t cn = 0, ct = 0, ce = 0, cr = 0, cns = 0, cts = 0, crs = 0, cn_ray0 = 0;
    Lane L; Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
    int pend = 0;                    /* parked leaf reference (0 = none) */
    uint32_t selfp = LH_MISS_PRIM;   /* SRC 1: the triangle this AO ray starts on, when it cannot occlude the ray (lh_ao.h) */
    size_t my = (size_t)-1;          /* ray this lane is working on */
    double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1;
    L.cur = kDone; L.sp = 1; L.np = 0; L.certain = false; L.over = false;
    bool exhausted = false;          /* wave-uniform: cursor ran past n */
    unsigned long long wbase = 0, wend = 0;   /* wave-uniform: this wave's reserved ray range */

    for (;;) {
        /* ---- regroup: retire finished lanes, refill them ----------------- */
        const bool idle = (L.cur == kDone) && (pend == 0);
        if (COUNT) crs++;
        if (idle && my != (size_t)-1) {
            finish<ANYHIT, COUNT>(L, sc, ox, oy, oz, dx, dy, dz, best, ce, SRC == 1 ? selfp : LH_MISS_PRIM);
            if (SRC == 0) write_out<ANYHIT>(my, L, best, prim, t, u, v, occ, sc.ref_nodes != NULL, WALK == 8 || WALK == 7);
            else {
                const bool hit = L.certain || best.prim != LH_MISS_PRIM;
                const bool retrace = sc.ref_nodes != NULL && (L.over || (best.prim != LH_MISS_PRIM && best.frag != 0u && !L.certain));
                if (retrace) {
                    const uint32_t k = atomicAdd(ao.qcount, 1u);
                    if (k < ao.qcap) { ao.queue[2 * (size_t)k] = (uint32_t)my; ao.queue[2 * (size_t)k + 1] = 5u; }   /* 5: the reference walk decides */
                    else atomicOr(ao.qcount + 1, 1u);
                } else if (hit) atomicAdd(&ao.occ_count[(uint32_t)my / (uint32_t)(ao.ntheta * ao.nphi)], 1u);
            }
            if (COUNT) {
                cr++;
                const uint32_t visits = cn - cn_ray0; cn_ray0 = cn;
                const int bkt = visits ? 32 - __clz((int)visits) : 0;
                atomicAdd(&counters[LH_CNT_HIST + (bkt < 23 ? bkt : 23)], 1ull);
            }
            my = (size_t)-1;
        }
        const unsigned long long idle_mask = __ballot(idle);
        /* refill from the wave's private range [wbase, wend); one atomic on the global cursor reserves
         * sc.ray_chunk rays (the cursor is ONE address: at a refill per ~33 rays it serialised the whole
         * grid -- 64 M same-address atomics/s for 2.1 Grays/s, profiles/README.md r01e) */
        if (idle_mask != 0ull && !exhausted) {
            if (wbase == wend) {
                unsigned long long b = 0;
                if ((tid & 63) == 0) b = atomicAdd(cursor, (unsigned long long)sc.ray_chunk);
                b = __shfl(b, 0);
                wbase = b < n ? b : n;
                wend = (b + sc.ray_chunk < n) ? b + sc.ray_chunk : n;
            }
            const int need = __popcll(idle_mask);
            const unsigned long long avail = wend - wbase;
            const int take = avail < (unsigned long long)need ? (int)avail : need;
            const int rank = __popcll(idle_mask & ((1ull << (tid & 63)) - 1ull));
            if (idle && rank < take) {
                const size_t i = wbase + rank;
                my = i;
                if (SRC == 0) {
                    ox = org[3 * i]; oy = org[3 * i + 1]; oz = org[3 * i + 2];
                    dx = dir[3 * i]; dy = dir[3 * i + 1]; dz = dir[3 * i + 2];
                } else {
                    const uint32_t N = (uint32_t)(ao.ntheta * ao.nphi), slot = (uint32_t)i / N;
                    const unsigned long long key = ao.slot_key[slot];
                    lh_ao_ray_builtin(ao.hitrec + LH_HITREC_DOUBLES * (size_t)slot, key, ao.seed, ao.ntheta, ao.nphi,
                                      (int)((uint32_t)i - slot * N), ox, oy, oz, dx, dy, dz);
                    selfp = lh_slot_selfprim(key);            /* LH_SLOT_NOSELF matches no primitive id (ids < 2^29) */
                }
                lane_init(L, sc, ox, oy, oz, dx, dy, dz);
                best.t = LH_T_INF; best.u = 0.0; best.v = 0.0; best.prim = LH_MISS_PRIM; best.frag = 0u;
                stk[0][tid] = kDone;
            }
            wbase += take;
            if (wbase >= n) exhausted = true;          /* the grid has handed out every ray */
        }
        const unsigned long long work = __ballot((L.cur != kDone) | (pend != 0));
        if (work == 0ull) break;
        /* ---- walk until too few lanes remain active ---------------------- */
        const int thresh = exhausted ? 1 : min_active;
        if (WALK == 3)
            traverse_spec4<ANYHIT, COUNT, false>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch, cns, cts);
        else if (WALK == 8)          /* the same with the stack check: trees whose worst case the LDS rows do not cover */
            traverse_spec4<ANYHIT, COUNT, true>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch, cns, cts);
        else
            traverse_spec8<ANYHIT, COUNT>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch, cns, cts);
    }
    if (COUNT) {
        add_counters(counters, cn, ct, ce, cr);
        atomicAdd(&counters[LH_CNT_NODE_SLOTS], (unsigned long long)cns);
        atomicAdd(&counters[LH_CNT_TRI_SLOTS], (unsigned long long)cts);
        atomicAdd(&counters[LH_CNT_REGROUP_SLOTS], (unsigned long long)crs);
    }
}

template <bool ANYHIT, bool COUNT, int WALK, int SRC>
__global__ __launch_bounds__(LH_BLOCK) void k_trace_persist_lane(
    lh_dev_scene_t sc, size_t n, const double *__restrict__ org, const double *__restrict__ dir,
    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
    double *__restrict__ v, uint8_t *__restrict__ occ, unsigned long long *counters,
    unsigned long long *cursor, int min_active, int tri_batch, const AoSrc ao)
{
    extern __shared__ int lh_stack_lds[];          /* [stack entries][LH_BLOCK], sized at launch */
    trace_persist_lane<ANYHIT, COUNT, WALK, SRC>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, ao, lh_stack_lds);
}

/* ------------------------------------------------------------------------ */
/* rays whose LDS stack column was too short (trees deeper than 19 4-wide   */
/* levels): the same walk, sequential, with a private stack                 */
/* ------------------------------------------------------------------------ */
#define LH_BIG_STACK 272        /* 3 * 88 + 8: the deepest 4-wide tree the device builder hands over */

template <bool ANYHIT>
__device__ void overflow_walk(const lh_dev_scene_t &sc, size_t i, const double *__restrict__ org, const double *__restrict__ dir,
                              uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u, double *__restrict__ v,
                              uint8_t *__restrict__ occ)
{
    const double ox = org[3 * i], oy = org[3 * i + 1], oz = org[3 * i + 2];
    const double dx = dir[3 * i], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
    const float4 *__restrict__ tris = (const float4 *)sc.tri32;
    Lane L; Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
    uint32_t ce = 0;
    int stack[LH_BIG_STACK]; int sp = 0;
    lane_init(L, sc, ox, oy, oz, dx, dy, dz);
    int cur = 0;
    for (;;) {
        if (cur >= 0) {
            const uint4 *p = (const uint4 *)sc.q4nodes + 4 * (size_t)cur;
            const uint4 a = p[0], b = p[1], c = p[2], r = p[3];
            float tn[4]; bool h[4]; const int ref[4] = {(int)r.x, (int)r.y, (int)r.z, (int)r.w};
            h[0] = slab_w(L, a.x, a.y, a.z, tn[0]) & (ref[0] != kDone);
            h[1] = slab_w(L, a.w, b.x, b.y, tn[1]) & (ref[1] != kDone);
            h[2] = slab_w(L, b.z, b.w, c.x, tn[2]) & (ref[2] != kDone);
            h[3] = slab_w(L, c.y, c.z, c.w, tn[3]) & (ref[3] != kDone);
            int order[4], nh = 0;
            for (int k = 0; k < 4; k++) if (h[k]) {
                int m = nh++;
                while (m > 0 && tn[order[m - 1]] > tn[k]) { order[m] = order[m - 1]; m--; }
                order[m] = k;
            }
            for (int k = nh - 1; k >= 1; k--) if (sp < LH_BIG_STACK) stack[sp++] = ref[order[k]];
            if (nh) cur = ref[order[0]];
            else if (sp) cur = stack[--sp];
            else break;
        } else {
            const uint32_t x = ~(uint32_t)cur, first = x >> 2, cnt = (x & 3u) + 1u;
            bool finished = false;
            for (uint32_t k = 0; k < cnt && !finished; k++) {
                const float4 *tp = tris + 3 * (size_t)(first + k);
                const float4 ta = tp[0], tb_ = tp[1], tc = tp[2];
                finished = tri_step<ANYHIT, false>(L, sc, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w,
                                                   __float_as_uint(tc.y), ox, oy, oz, dx, dy, dz, best, ce);
            }
            if (finished || sp == 0) break;
            cur = stack[--sp];
        }
    }
    L.over = false;
    finish<ANYHIT, false>(L, sc, ox, oy, oz, dx, dy, dz, best, ce);
    write_out<ANYHIT>(i, L, best, prim, t, u, v, occ, sc.ref_nodes != NULL);
}

__global__ __launch_bounds__(256) void k_overflow_fix(lh_dev_scene_t sc, size_t n, const double *__restrict__ org,
                                                      const double *__restrict__ dir, uint32_t *__restrict__ prim,
                                                      double *__restrict__ t, double *__restrict__ u, double *__restrict__ v,
                                                      uint8_t *__restrict__ occ, int anyhit, unsigned long long *counters)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (anyhit ? (occ[i] != LH_OCC_OVERFLOW) : (prim[i] != LH_PRIM_OVERFLOW)) return;
    if (anyhit) overflow_walk<true>(sc, i, org, dir, prim, t, u, v, occ);
    else overflow_walk<false>(sc, i, org, dir, prim, t, u, v, occ);
    if (counters) atomicAdd(&counters[LH_CNT_RETRACED], 1ull);
}

/* ------------------------------------------------------------------------ */
/* rays flagged by write_out: the reference's own walk on its own tree       */
/* ------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_ref_retrace(lh_dev_scene_t sc, size_t n, const double *__restrict__ org,
                                                     const double *__restrict__ dir, uint32_t *__restrict__ prim,
                                                     double *__restrict__ t, double *__restrict__ u, double *__restrict__ v,
                                                     uint8_t *__restrict__ occ, int anyhit, unsigned long long *counters)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (anyhit ? (occ[i] != LH_OCC_RETRACE) : (prim[i] != LH_PRIM_RETRACE)) return;
    uint32_t p; double tt, uu, vv;
    const int hit = lh_ref_trace((const lh_refnode_t *)sc.ref_nodes, (const uint32_t *)sc.ref_leaf_prims, (const double *)sc.tri64,
                                 sc.ref_empty, sc.ref_bmin, sc.ref_bmax, org[3 * i], org[3 * i + 1], org[3 * i + 2],
                                 dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], &p, &tt, &uu, &vv);
    if (anyhit) occ[i] = hit ? 1 : 0;
    else { prim[i] = p; t[i] = tt; u[i] = uu; v[i] = vv; }
    if (counters) atomicAdd(&counters[LH_CNT_RETRACED], 1ull);
}

/* the queued AO rays of the fused stage (SRC 1 above): regenerated from (slot, sample), decided by the reference's own
 * walk on its own tree, added to their slot's count.  Queue entries: (ray index, reason) pairs. */
__global__ __launch_bounds__(256) void k_ao_queue(const lh_dev_scene_t sc, const AoSrc ao, unsigned long long *counters)
{
    const uint32_t total = *ao.qcount < ao.qcap ? *ao.qcount : ao.qcap;
    const uint32_t N = (uint32_t)(ao.ntheta * ao.nphi);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const uint32_t i = ao.queue[2 * e], slot = i / N;
        double ox, oy, oz, dx, dy, dz;
        lh_ao_ray_builtin(ao.hitrec + LH_HITREC_DOUBLES * (size_t)slot, ao.slot_key[slot], ao.seed, ao.ntheta, ao.nphi,
                          (int)(i - slot * N), ox, oy, oz, dx, dy, dz);
        uint32_t p; double tt, uu, vv;
        const int hit = lh_ref_trace((const lh_refnode_t *)sc.ref_nodes, (const uint32_t *)sc.ref_leaf_prims, (const double *)sc.tri64,
                                     sc.ref_empty, sc.ref_bmin, sc.ref_bmax, ox, oy, oz, dx, dy, dz, &p, &tt, &uu, &vv);
        if (counters) atomicAdd(&counters[LH_CNT_RETRACED], 1ull);
        if (hit) atomicAdd(&ao.occ_count[slot], 1u);
    }
}

template <bool ANYHIT, bool COUNT>
int launch_one(const lh_dev_scene_t &sc, size_t n, const double *org, const double *dir,
               uint32_t *prim, double *t, double *u, double *v, uint8_t *occ,
               unsigned long long *counters, unsigned long long *cursor, int walk,
               int grid_blocks, int min_active, int tri_batch, size_t lds_bytes, hipStream_t s)
{
    if (walk == 0) {
        const size_t blocks = (n + LH_BLOCK - 1) / LH_BLOCK;
        if (blocks > 0x7fffffffull) return -1;
        hipLaunchKernelGGL((k_trace_direct<ANYHIT, COUNT>), dim3((unsigned)blocks), dim3(LH_BLOCK), lds_bytes, s,
                           sc, n, org, dir, prim, t, u, v, occ, counters);
    } else {
        if (hipMemsetAsync(cursor, 0, sizeof(unsigned long long), s) != hipSuccess) return -1;
        if (walk == 7)
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 7, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
        else if (walk == 8)
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 8, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
        else
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 3, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_walk(const lh_dev_scene_t &sc, size_t n, const double *org, const double *dir,
                uint32_t *prim, double *t, double *u, double *v, int anyhit, uint8_t *occ,
                unsigned long long *counters, unsigned long long *cursor, int walk,
                int grid_blocks, int min_active, int tri_batch, size_t lds_bytes, hipStream_t s)
{
    if (anyhit) {
        if (counters) return launch_one<true, true>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, walk, grid_blocks, min_active, tri_batch, lds_bytes, s);
        return launch_one<true, false>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, walk, grid_blocks, min_active, tri_batch, lds_bytes, s);
    }
    if (counters) return launch_one<false, true>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, walk, grid_blocks, min_active, tri_batch, lds_bytes, s);
    return launch_one<false, false>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, walk, grid_blocks, min_active, tri_batch, lds_bytes, s);
}

/* LDS stack rows of a 4-wide walk over this scene: 3 * depth + 5 covers every ray; beyond the cap (64 rows; tests lower it
 * through "stack_cap") the walk checks before it pushes and a ray that would overrun is finished elsewhere */
uint32_t rows4(const lh_dev_scene_t &sc, bool *guard)
{
    uint32_t need = 3 * sc.q4_depth + 5;
    const uint32_t cap = (sc.stack_cap >= 8 && sc.stack_cap < 64) ? sc.stack_cap : 64;
    *guard = need > cap;
    if (need > cap) need = cap;
    need = (need + 1u) & ~1u;
    if (need < 16 && !*guard) need = 16;
    return need;
}

/* rays per cursor atomic: the scene's setting, but never so large that a wave gets fewer than
 * ~4 ranges of a small batch (tail imbalance: late path-tracing bounces, small tiles) */
void clamp_chunk(lh_dev_scene_t &scl, size_t n, int grid_blocks)
{
    const size_t waves = (size_t)(grid_blocks > 0 ? grid_blocks : 1) * (LH_BLOCK / 64);
    size_t c = n / (waves * 4);
    if (c < 64) c = 64;
    if (c < scl.ray_chunk) scl.ray_chunk = (uint32_t)c;
    if (scl.ray_chunk == 0) scl.ray_chunk = 64;
}

} /* namespace */

/* the AO stage of a tile with the rays generated inside the any-hit kernel (SRC 1 above): nslots primary hits,
 * N = ntheta * nphi rays each, occluded rays counted per slot in d_occ_count (zeroed here).  Fragile hits go to
 * d_queue (2 words per entry, count + overflow flag in d_qcount[0..1]); lh_launch_ao_queue runs the reference walk
 * for them.  A tree deeper than the LDS rows needs the reference-order tree (overflowing rays are queued too). */
extern "C" int lh_launch_trace_ao(const lh_dev_scene_t *sc, size_t nslots, int ntheta, int nphi, unsigned long long seed,
                                  const double *d_hitrec, const unsigned long long *d_slot_key, unsigned int *d_occ_count,
                                  unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks, int min_active,
                                  int tri_batch, uint32_t *d_queue, uint32_t *d_qcount, uint32_t qcap, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const size_t n = nslots * (size_t)(ntheta * nphi);
    if (n == 0) return 0;
    if (n >= ((size_t)1 << 32)) return -1;
    lh_dev_scene_t scl = *sc;
    bool guard = false;
    scl.stack_rows = rows4(*sc, &guard);
    if (guard && !sc->ref_nodes) return -1;
    scl.stack_guard = guard ? 1 : 0;
    const size_t lds_bytes = (size_t)scl.stack_rows * LH_BLOCK * sizeof(int);
    if (scl.ray_chunk < 512) scl.ray_chunk = 512;      /* AO rays of a slot are coherent: longer ranges per wave (config 5: 92.9 -> 91.4 ms, tools/ao_sweep5.py) */
    clamp_chunk(scl, n, grid_blocks);
    AoSrc ao = {d_hitrec, d_slot_key, d_occ_count, seed, ntheta, nphi, d_queue, d_qcount, qcap};
    if (hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_qcount, 0, 2 * sizeof(uint32_t), s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_occ_count, 0, sizeof(unsigned int) * nslots, s) != hipSuccess) return -1;
#define LH_AO_LAUNCH(CNT, W) hipLaunchKernelGGL((k_trace_persist_lane<true, CNT, W, 1>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s, \
                           scl, n, (const double *)NULL, (const double *)NULL, (uint32_t *)NULL, (double *)NULL, (double *)NULL, \
                           (double *)NULL, (uint8_t *)NULL, d_counters, d_cursor, min_active, tri_batch, ao)
    if (guard) { if (d_counters) LH_AO_LAUNCH(true, 8); else LH_AO_LAUNCH(false, 8); }
    else { if (d_counters) LH_AO_LAUNCH(true, 3); else LH_AO_LAUNCH(false, 3); }
#undef LH_AO_LAUNCH
    if (hipGetLastError() != hipSuccess) return -1;
    hipLaunchKernelGGL(k_ao_queue, dim3(64), dim3(256), 0, s, scl, ao, d_counters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* the node formats a launch of `variant` reads on this scene (bit mask, LH_FMT_* in lh_internal.h): so that the commit
 * code can upload a format the first time a variant asks for it */
extern "C" int lh_trace_formats_needed(const lh_dev_scene_t *sc, int variant)
{
    (void)sc;
    return variant == LH_VARIANT_DIRECT ? 1 : 4;
}

/* one batch of rays through the hot path.  variant: LH_VARIANT_SPEC (the default: 4-wide nodes, or the 8-wide nodes when
 * sc->prefer_q8) or LH_VARIANT_DIRECT (the textbook walk over the 2-wide fp32 nodes; needs sc->nodes). */
extern "C" int lh_launch_trace(const lh_dev_scene_t *sc, size_t n, const double *d_org,
                               const double *d_dir, uint32_t *d_prim, double *d_t, double *d_u,
                               double *d_v, int anyhit, uint8_t *d_occluded,
                               unsigned long long *d_counters, unsigned long long *d_workq,
                               int variant, int grid_blocks, int min_active, int tri_batch, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    lh_dev_scene_t scl = *sc;
    uint32_t need; int walk; bool over_fix = false;
    if (variant == LH_VARIANT_DIRECT) {
        if (!sc->nodes) return -1;
        need = sc->max_depth + 2; walk = 0;          /* 2-wide: one push per level + the sentinel */
        need = (need + 1u) & ~1u;
        if (need < 16) need = 16;
    } else if (sc->prefer_q8 && sc->q8nodes) {
        /* the 8-wide walk pushes up to 7 per level and keeps a scratch row; beyond 48 rows (three workgroups per CU) the rare
         * ray that needs them is finished by k_overflow_fix over the 4-wide nodes (always resident) */
        need = 7 * sc->q8_depth + 10; walk = 7;
        const uint32_t cap = (sc->stack_cap >= 16 && sc->stack_cap < 64) ? sc->stack_cap : 48;
        if (need > cap) { need = cap; over_fix = true; scl.stack_guard = 1; }
        need = (need + 1u) & ~1u;
        if (need < 16 && !over_fix) need = 16;
    } else {
        /* a very deep tree (chains of nested geometry, an LBVH over a degenerate distribution): the 4-wide walk's worst case
         * does not fit the 64-row LDS stack; a ray that would overrun it (none in practice: the bound is three pushes on
         * every level) is finished by k_overflow_fix */
        scl.prefer_q8 = 0;
        need = rows4(*sc, &over_fix); walk = over_fix ? 8 : 3;
        scl.stack_guard = over_fix ? 1 : 0;
    }
    if (need > 64) return -1;
    scl.stack_rows = need;
    const size_t lds_bytes = (size_t)need * LH_BLOCK * sizeof(int);
    clamp_chunk(scl, n, grid_blocks);
    const int rc = launch_walk(scl, n, d_org, d_dir, d_prim, d_t, d_u, d_v, anyhit, d_occluded,
                               d_counters, d_workq, walk, grid_blocks, min_active, tri_batch, lds_bytes, s);
    if (rc != 0) return rc;
    if (over_fix)
        hipLaunchKernelGGL(k_overflow_fix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scl, n, d_org, d_dir,
                           d_prim, d_t, d_u, d_v, d_occluded, anyhit, d_counters);
    if (sc->ref_nodes)
        hipLaunchKernelGGL(k_ref_retrace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scl, n, d_org, d_dir,
                           d_prim, d_t, d_u, d_v, d_occluded, anyhit, d_counters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
