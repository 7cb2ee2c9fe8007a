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
#include <stdlib.h>
#include <pthread.h>

#include "../../include/lucille_hip.h"
#include "lh_device.h"
#include "lh_filter.h"
#include "lh_reftrace.h"
#include "lh_ao.h"

namespace {

#include "lh_walk.h"
#include "lh_pt.h"

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
 * nothing was hit (a pop).  A lane that reaches a leaf parks it in `pend` and keeps
 * descending from its stack; the ~75-instruction triangle path runs only when at least
 * `tri_batch` lanes hold a parked leaf (or nobody has an inner node left), one triangle per
 * parked lane per pass.
 *
 * The visit budget: `it` counts the wave's iterations (one node step and at most one triangle step per lane each); a ray
 * that has been in its lane for more than sc.ray_budget iterations when the wave regroups (at least every 64 iterations) --
 * a grazing ray skimming a tessellated floor: 4 of the 443 M AO rays of the config-5 frame visit more than 1024 nodes, and
 * the leaves they park cost a triangle step each, at 1-2 us per dependent step -- is marked `over` and handed to the
 * wave-cooperative walk (k_coop_walk below), so that the tail of a launch is bounded by
 * budget x latency instead of by its longest ray (profiles/README.md r03: 3-6 ms per launch on a
 * rank's share of the frame).  STRIDE: lanes per LDS stack row (LH_BLOCK, or 64 in k_coop_walk). */
constexpr int kNoLeaf = 0;       /* never a valid leaf reference (leaf refs are negative) */
constexpr uint32_t kRegroupMask = 63u;   /* the walk returns to the regroup point at least every 64 iterations */

/* RING: stack positions are taken modulo ring_mask + 1 rows (k_coop_walk: a lane that gives entries away from the bottom of
 * its stack drifts upwards without bound); the persistent kernel indexes rows directly */
/* SORTED = false (EVERY any-hit walk over the 4-wide nodes: the fused AO stage, SRC 1, and any-hit ray dumps, SRC 0 -- the macro's
 * name dates from the first of the two): the hit children go onto the stack in SLOT order, no ranking by entry distance -- any hit
 * ends the ray, the culling bound never moves, so the order only decides how soon an occluded ray meets its occluder, and the step
 * loses its six key comparisons and rank sums (~27 of 136 VALU operations).  Answers do not depend on it; visit counts do (r05,
 * profiles/README.md: config-5 AO frame 14.76 -> 14.07 node visits per ray, S-soup-1M any-hit dump 38.85 -> 38.45 and 2 677 -> 2 784
 * Mrays/s), and with them how many rays run past ray_budget into the cooperative walk.  LH_AO_UNSORTED=0 restores the ranking. */
#ifndef LH_AO_UNSORTED
#define LH_AO_UNSORTED 1
#endif
template <bool COUNT, int STRIDE, bool RING, bool SORTED = true>
__device__ __forceinline__ void node_step4(Lane &L, int &pend, const lh_dev_scene_t &sc, int (*stk)[STRIDE], const int tid, uint32_t &c_nodes,
                                           const int ring_mask = 0, const uint4 *top = NULL, const uint32_t ntop = 0u)
{
#define LH_ROW(x) (RING ? ((x) & ring_mask) : (x))
    /* the first ntop nodes (level order: the top of the tree, which every ray walks) are read from the workgroup's copy in LDS:
     * no request leaves the CU for them */
    uint4 a, b, c, r;
    if (!RING && (uint32_t)L.cur < ntop) {
        const uint4 *lp = top + 4 * (uint32_t)L.cur;
        a = lp[0]; b = lp[1]; c = lp[2]; r = lp[3];
    } else {
        const uint4 *p = (const uint4 *)sc.q4nodes + 4 * (size_t)L.cur;
        a = p[0]; b = p[1]; c = p[2]; r = p[3];
    }
    if (COUNT) c_nodes++;
    float t0, t1, t2, t3;
    const bool h0 = slab_w(L, a.x, a.y, a.z, t0) & ((int)r.x != kDone);
    const bool h1 = slab_w(L, a.w, b.x, b.y, t1) & ((int)r.y != kDone);
    const bool h2 = slab_w(L, b.z, b.w, c.x, t2) & ((int)r.z != kDone);
    const bool h3 = slab_w(L, c.y, c.z, c.w, t3) & ((int)r.w != kDone);
    const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    const int base = L.sp + nh - 1;
    if (SORTED) {
        /* entry distances are >= 0, so their bit patterns order like unsigned integers */
        const uint32_t k0 = h0 ? ((__float_as_uint(t0) & ~3u) | 0u) : 0xFFFFFFFCu;
        const uint32_t k1 = h1 ? ((__float_as_uint(t1) & ~3u) | 1u) : 0xFFFFFFFDu;
        const uint32_t k2 = h2 ? ((__float_as_uint(t2) & ~3u) | 2u) : 0xFFFFFFFEu;
        const uint32_t k3 = h3 ? ((__float_as_uint(t3) & ~3u) | 3u) : 0xFFFFFFFFu;
        const int b10 = k1 < k0, b20 = k2 < k0, b30 = k3 < k0, b21 = k2 < k1, b31 = k3 < k1, b32 = k3 < k2;
        const int rk0 = b10 + b20 + b30, rk1 = (1 - b10) + b21 + b31;
        const int rk2 = (2 - b20 - b21) + b32, rk3 = 3 - b30 - b31 - b32;
        stk[LH_ROW(h0 ? base - rk0 : L.sp + rk0)][tid] = (int)r.x;
        stk[LH_ROW(h1 ? base - rk1 : L.sp + rk1)][tid] = (int)r.y;
        stk[LH_ROW(h2 ? base - rk2 : L.sp + rk2)][tid] = (int)r.z;
        stk[LH_ROW(h3 ? base - rk3 : L.sp + rk3)][tid] = (int)r.w;
    } else {
        /* hit c sits below the hits before it (slot 0 ends on top); every miss goes to the row above the new top: free space */
        const int a1 = (int)h0, a2 = a1 + (int)h1, a3 = a2 + (int)h2;
        stk[LH_ROW(h0 ? base : base + 1)][tid] = (int)r.x;
        stk[LH_ROW(h1 ? base - a1 : base + 1)][tid] = (int)r.y;
        stk[LH_ROW(h2 ? base - a2 : base + 1)][tid] = (int)r.z;
        stk[LH_ROW(h3 ? base - a3 : base + 1)][tid] = (int)r.w;
    }
    L.sp = base;
    const int nxt = stk[LH_ROW(base)][tid];
    const int popped2 = stk[LH_ROW(L.sp - 1)][tid];
    const bool is_leaf = (nxt < 0) & (nxt != kDone);
    const bool park = is_leaf & (pend == kNoLeaf);
    pend = park ? nxt : pend;
    L.cur = park ? popped2 : nxt;
    L.sp -= park ? 1 : 0;
}

/* one triangle of the parked leaf of every lane that holds one */
template <bool ANYHIT, bool COUNT, int STRIDE, bool RING>
__device__ __forceinline__ void tri_pass(Lane &L, int &pend, const lh_dev_scene_t &sc, int (*stk)[STRIDE], const int tid,
                                         double ox, double oy, double oz, double dx, double dy, double dz, Best &best,
                                         uint32_t &c_tris, uint32_t &c_exact, const int ring_mask = 0)
{
    if (pend != kNoLeaf) {
        const float4 *__restrict__ tris = (const float4 *)sc.tri32;
        const uint32_t x = ~(uint32_t)pend;
        const float4 *tp = tris + 3 * (size_t)(x >> 2);
        const float4 ta = tp[0], tb_ = tp[1], tc = tp[2];
        if (COUNT) c_tris++;
        const bool finished = tri_step<ANYHIT, COUNT>(L, sc, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w, __float_as_uint(tc.y), ox, oy, oz, dx, dy, dz, best, c_exact);
        if (finished) { L.cur = kDone; pend = kNoLeaf; }
        else if (x & 3u) pend = (int)~(((x >> 2) + 1u) << 2 | ((x & 3u) - 1u));
        else {
            /* leaf finished: a lane that was waiting with a second leaf parks it now */
            const bool waiting = (L.cur < 0) & (L.cur != kDone);
            pend = waiting ? L.cur : kNoLeaf;
            if (waiting) { L.sp--; L.cur = stk[LH_ROW(L.sp)][tid]; }
        }
    }
#undef LH_ROW
}

template <bool ANYHIT, bool COUNT, bool GUARD, bool SORTED = true>
__device__ __forceinline__ void traverse_spec4(Lane &L, int &pend, const lh_dev_scene_t &sc,
                                               int (*stk)[LH_BLOCK], const int tid,
                                               double ox, double oy, double oz,
                                               double dx, double dy, double dz, Best &best,
                                               uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact,
                                               const int min_active, const int tri_batch,
                                               uint32_t &c_nslots, uint32_t &c_tslots, uint32_t &it)
{
    const int rows = (int)sc.stack_rows;
    const uint4 *top = (const uint4 *)(&stk[rows][0]);        /* the workgroup's copy of the first sc.top_nodes nodes, behind the stack rows */
    const uint32_t ntop = sc.top_nodes;
    for (;;) {
        if (COUNT) { if (__ballot(L.cur >= 0) != 0ull) c_nslots++; }
        /* the step below writes up to slot sp + 3.  rows = 3 * depth + 5 covers every ray of a tree that deep; a deeper
         * tree (an LBVH built on the device over a degenerate distribution) gets 64 rows and a ray that would overrun them
         * is finished by k_coop_walk -- same arithmetic, same answer */
        if (GUARD) {
            /* BEFORE the step, not after it: a lane that came out of a step holding a leaf (no check: it was not going to step) pops
             * an inner node in the triangle pass and steps with a stack pointer one below the one that was never checked -- the
             * write landed one row past the stack (dropped by the LDS until round 4 put the top of the tree there).  One compare
             * and a scalar branch per iteration */
            const bool ov = (L.cur >= 0) & (L.sp + 4 > rows);
            if (__builtin_expect(__ballot(ov) != 0ull, 0)) { if (ov) { L.over = true; L.cur = kDone; pend = kNoLeaf; } }
        }
        if (L.cur >= 0) node_step4<COUNT, LH_BLOCK, false, SORTED>(L, pend, sc, stk, tid, c_nodes, 0, top, ntop);
        const unsigned long long m_node = __ballot(L.cur >= 0);
        const unsigned long long m_pend = __ballot(pend != kNoLeaf);
        if (m_pend != 0ull && (__popcll(m_pend) >= tri_batch || m_node == 0ull)) {
            if (COUNT) c_tslots++;
            tri_pass<ANYHIT, COUNT, LH_BLOCK, false>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, c_tris, c_exact);
        }
        const unsigned long long m_work = __ballot((L.cur != kDone) | (pend != kNoLeaf));
        if (__popcll(m_work) < min_active || (++it & kRegroupMask) == 0u) break;      /* at least every 64 iterations: the visit budget is checked at regroup points */
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
                                               uint32_t &c_nslots, uint32_t &c_tslots, uint32_t &it)
{
    const float4 *__restrict__ tris  = (const float4 *)sc.tri32;
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
        if (__popcll(m_work) < min_active || (++it & kRegroupMask) == 0u) break;      /* at least every 64 iterations: the visit budget is checked at regroup points */
    }
}

/* resolve whatever is still queued; afterwards `best` is the exact answer */

template <bool ANYHIT>
__device__ __forceinline__ void write_out(size_t i, const Lane &L, const Best &best,
                                          uint32_t *__restrict__ prim, double *__restrict__ t,
                                          double *__restrict__ u, double *__restrict__ v,
                                          uint8_t *__restrict__ occ, const bool retrace_on)
{
    if (L.over) {            /* out of visit budget, or the LDS stack was too short for this ray: k_coop_walk redoes it */
        if (ANYHIT) occ[i] = (uint8_t)LH_OCC_OVERFLOW; else prim[i] = LH_PRIM_OVERFLOW;
        return;
    }
    /* a hit the reference may not reach goes through the reference's own walk (k_fixups);
     * a certain fp32 hit is strictly inside its triangle, hence inside every box: never fragile */
    const bool retrace = retrace_on && best.prim != LH_MISS_PRIM && best.frag != 0u && !(ANYHIT && L.certain);
    if (ANYHIT) {
        occ[i] = retrace ? (uint8_t)LH_OCC_RETRACE : ((L.certain || best.prim != LH_MISS_PRIM) ? 1 : 0);
    } else {
        prim[i] = retrace ? LH_PRIM_RETRACE : best.prim; t[i] = best.t; u[i] = best.u; v[i] = best.v;
    }
}

/* the fix-up queue of a launch: 64-bit entries (ray index | reason << 56) appended by the persistent kernel -- LH_Q_REF: a
 * fragile hit of the fused AO stage, the reference's own walk decides; LH_Q_COOP: out of visit budget / LDS stack rows,
 * k_coop_walk finishes it.  The consumer (k_coop_walk) runs CONCURRENTLY on a second stream: an entry is one agent-scope
 * atomic store into a slot that was zero, qcount[0] counts the appends (it may pass qcap: the entries beyond are not stored
 * and qcount[1] is set), qcount[2] counts the producer waves that have left (release: their entries are visible),
 * qcount[4 ..] holds the consumer groups' next entries. */
#define LH_Q_REF  5u
#define LH_Q_COOP 6u
struct FixQ { unsigned long long *queue; uint32_t *qcount; uint32_t qcap, nprod; uint32_t *heads; };
#define LH_Q_GROUPS 4096u           /* consumer groups at most (4 per wave, one wave per CU and a few) */

__device__ __forceinline__ bool fixq_push(const FixQ &q, size_t i, uint32_t reason)
{
    const uint32_t k = atomicAdd(q.qcount, 1u);
    if (k < q.qcap) {
        __hip_atomic_store(q.queue + k, (unsigned long long)i | ((unsigned long long)reason << 56), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    atomicOr(q.qcount + 1, 1u);
    return false;
}

/* before a launch: the entries of the previous launch on this queue back to zero, then its counters */
__global__ __launch_bounds__(256) void k_fixq_reset(unsigned long long *queue, uint32_t *qcount, uint32_t qcap)
{
    const uint32_t used = qcount[0] < qcap ? qcount[0] : qcap;
    for (uint32_t k = threadIdx.x; k < used; k += 256) queue[k] = 0ull;
    for (uint32_t k = threadIdx.x; k < LH_Q_GROUPS; k += 256) qcount[4 + k] = k;       /* heads: group g starts at entry g */
    __syncthreads();
    if (threadIdx.x < 3) qcount[threadIdx.x] = 0u;
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
    if (__builtin_expect(ray_needs_ref_walk(sc, L, dx, dy, dz), 0)) LH_FORCE_REF_WALK(L, best);
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
    uint32_t nslots; int group;           /* group: 64 = items ordered (64 slots) x (sample) x (slot in group), 0 = (slot) x (sample) */
    const unsigned long long *nslots_dev; /* NULL, or: the slot count lives on the device (the compaction's total: the host launches this stage without reading
                                             it back; nslots is then its upper bound) */
    uint32_t budget_big;                  /* 0, or: the visit budget of a launch that turns out to hold 2^27 rays or more (lh_tile.hip) */
};

/* Work item i of a fused AO stage -> (hit slot, sample).  The plain order -- a slot's N samples side by side -- puts the N
 * directions of ONE hemisphere into the lanes of a wave: one origin, sixty-four directions, and one address for their
 * sixty-four occlusion atomics.  The grouped order (round 4) takes the slots 64 at a time and runs the SAME sample index of
 * the group's 64 slots side by side: neighbouring pixels' hits, the same stratum of the hemisphere (calculate_occlusion's
 * (i, j), ambientocclusion.c:65-117) -- nearby origins, nearly parallel directions, the same nodes and leaves; the atomics of
 * a wave go to 64 different counters.  The rays are the same rays (keyed by absolute pixel and sample), so is the frame. */
__device__ __forceinline__ void ao_item(const AoSrc &ao, uint32_t i, uint32_t &slot, uint32_t &r)
{
    const uint32_t N = (uint32_t)(ao.ntheta * ao.nphi);
    if (ao.group == 0) { slot = i / N; r = i - slot * N; return; }
    const uint32_t G = (uint32_t)ao.group, per = G * N, blk = i / per, q = i - blk * per;
    const uint32_t left = ao.nslots - blk * G, m = left < G ? left : G;      /* the last group may be short */
    r = q / m; slot = blk * G + (q - r * m);
}

/* ray `i` of the launch.  SRC 0: from the arrays; 1: the AO ray (slot i / N, sample i % N) regenerated from the hit record
 * (selfp: the triangle it starts on, when that cannot occlude it); 2: the camera ray of path i of a path-traced pass */
template <int SRC>
__device__ __forceinline__ void src_ray(const lh_dev_scene_t &sc, uint32_t i, const double *__restrict__ org, const double *__restrict__ dir,
                                        const AoSrc &ao, double &ox, double &oy, double &oz, double &dx, double &dy, double &dz, uint32_t &selfp)
{
    if (SRC == 0) {
        ox = org[3 * (size_t)i]; oy = org[3 * (size_t)i + 1]; oz = org[3 * (size_t)i + 2];
        dx = dir[3 * (size_t)i]; dy = dir[3 * (size_t)i + 1]; dz = dir[3 * (size_t)i + 2];
    } else if (SRC == 1) {
        uint32_t slot, r;
        ao_item(ao, i, slot, r);
        const unsigned long long key = ao.slot_key[slot];
        lh_ao_ray_builtin(ao.hitrec + LH_HITREC_DOUBLES * (size_t)slot, key, ao.seed, ao.ntheta, ao.nphi, (int)r, ox, oy, oz, dx, dy, dz);
        selfp = lh_slot_selfprim(key);            /* LH_SLOT_NOSELF matches no primitive id (ids < 2^29) */
    } else {
        double o[3], d[3];
        pt_camera_ray((const PtCamSrc *)sc.cam_src, i, o, d);
        ox = o[0]; oy = o[1]; oz = o[2]; dx = d[0]; dy = d[1]; dz = d[2];
    }
}

constexpr uint32_t kNoRay = 0xFFFFFFFFu;
#ifndef LH_REFILL_PASSES
#define LH_REFILL_PASSES 2          /* ranges a refill may draw from (1: rounds 1-5: the lanes a range's end did not fill wait for the next regroup) */
#endif

template <bool ANYHIT, bool COUNT, int WALK, int SRC>
__device__ __forceinline__ void trace_persist_lane(
    const lh_dev_scene_t &sc, const uint32_t n, const double *__restrict__ org, const double *__restrict__ dir,
    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
    double *__restrict__ v, uint8_t *__restrict__ occ, unsigned long long *counters,
    uint32_t *cursor, int min_active, int tri_batch, const AoSrc &ao, const FixQ &fq, int *lds)
{
    int (*stk)[LH_BLOCK] = (int (*)[LH_BLOCK])lds;
    const int tid = threadIdx.x;
    uint32_t cn = 0, ct = 0, ce = 0, cr = 0, cns = 0, cts = 0, crs = 0, cn_ray0 = 0;
    Lane L; Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
    int pend = 0;                    /* parked leaf reference (0 = none) */
    uint32_t selfp = LH_MISS_PRIM;   /* SRC 1: the triangle this AO ray starts on, when it cannot occlude the ray (lh_ao.h) */
    uint32_t my = kNoRay;            /* ray this lane is working on (a launch holds fewer than 2^31 rays: 32-bit indices keep the walk under 128 VGPRs) */
    double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1;
    L.cur = kDone; L.sp = 1; L.np = 0; L.certain = false; L.over = false;
#ifdef LH_DIAG_CLOCK      /* wave start / exit clocks (LH_STAGE_TIMING); off in the product: two SGPRs live across the whole kernel */
    if (sc.diag_clock && (tid & 63) == 0) sc.diag_clock[(size_t)blockIdx.x * (LH_BLOCK / 64) + (tid >> 6)] = wall_clock64();
#endif
    const uint32_t budget = (SRC == 1 && ao.budget_big && n >= (1u << 27)) ? ao.budget_big : sc.ray_budget;
    bool exhausted = false;          /* wave-uniform: every partition's cursor ran past its end */
    uint32_t wbase = 0, wend = 0;    /* wave-uniform: this wave's reserved ray range */
    uint32_t part, drained = 0;      /* wave-uniform: the partition this wave draws from, bit mask of the partitions known to be handed out */
    uint32_t it = 0, it0 = 0;        /* wave-uniform iteration counter of the walk; its value when this lane's ray started */
    {
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));      /* for speed only: any placement is correct */
        part = ((uint32_t)xcc & 7u) % LH_NPART;
    }

    /* a batch smaller than the grid: the cursors hand out at most n / 64 + LH_NPART reservations, the waves beyond that count
     * could only find them spent */
    if ((uint64_t)(blockIdx.x * (LH_BLOCK / 64) + (tid >> 6)) * 64ull >= (uint64_t)n + 64ull * LH_NPART) exhausted = true;

    for (;;) {
        /* ---- regroup: retire finished lanes, refill them ----------------- */
        /* out of budget: the ray leaves the persistent walk here and is finished cooperatively (its partial results are dropped) */
        if (__builtin_expect(my != kNoRay && it - it0 > budget && ((L.cur != kDone) | (pend != 0)), 0)) { L.over = true; L.cur = kDone; pend = 0; }
        const bool idle = (L.cur == kDone) && (pend == 0);
        if (COUNT) crs++;
        if (idle && my != kNoRay) {
            if (!L.over) finish<ANYHIT, COUNT>(L, sc, ox, oy, oz, dx, dy, dz, best, ce, SRC == 1 ? selfp : LH_MISS_PRIM);
            /* rays the persistent walk does not finish go through the fix-up queue to the concurrent consumer: out of budget /
             * stack rows -> the cooperative walk; a hit the reference may not reach (fragile; a certain fp32 hit is strictly
             * inside its triangle, hence inside every box: never fragile) -> the reference's own walk.  A full queue (ray dumps:
             * the ray stays flagged in its output slot for k_fixups; AO stage: the caller redoes the stage materialised) */
            const bool fragile = sc.ref_nodes != NULL && !L.over && best.prim != LH_MISS_PRIM && best.frag != 0u && !(ANYHIT && L.certain);
            bool queued = false;
            if (__builtin_expect(L.over | fragile, 0)) queued = fixq_push(fq, my, L.over ? LH_Q_COOP : LH_Q_REF);
            if (SRC != 1) { if (__builtin_expect(!queued, 1)) write_out<ANYHIT>(my, L, best, prim, t, u, v, occ, sc.ref_nodes != NULL); }
            else if (!L.over && !fragile && (L.certain || best.prim != LH_MISS_PRIM)) { uint32_t sl, rr; ao_item(ao, my, sl, rr); atomicAdd(&ao.occ_count[sl], 1u); }
            if (COUNT) {
                cr++;
                /* rays by node visits: every ray of 64 visits or more, one in 256 of the shorter ones (weighted 256: the
                 * histogram's atomics land on a few addresses -- 1.7 G of them made a counted path-traced frame take 18 s) */
                const uint32_t visits = cn - cn_ray0; cn_ray0 = cn;
                const int bkt = visits ? 32 - __clz((int)visits) : 0;
                if (visits >= 64u) atomicAdd(&counters[LH_CNT_HIST + (bkt < 23 ? bkt : 23)], 1ull);
                else if ((my & 255u) == 0u) atomicAdd(&counters[LH_CNT_HIST + bkt], 256ull);
            }
            my = kNoRay;
        }
        const unsigned long long idle_mask = __ballot(idle);
        /* refill from the wave's private range [wbase, wend); one atomic on a cursor reserves sc.ray_chunk rays (one atomic
         * per regroup -- a refill every ~33 rays -- made the single cursor address the serialisation point of the whole grid:
         * 64 M same-address atomics/s at 2.1 Grays/s, profiles/README.md r01e).
         * XCD-aware: the batch is cut into LH_NPART contiguous partitions with a cursor each; a wave draws from the partition
         * of the XCD it runs on (each XCD has its own 4 MiB L2: the rays of one image region -- and the nodes and triangles
         * they touch -- stay in ONE L2 instead of all eight), and moves on to the next partition when its own is drained. */
        uint32_t newray = kNoRay;           /* the ray this lane starts now */
        if (idle_mask != 0ull && !exhausted) {
            /* Idle lanes take the next rays of the wave's range -- and, when the range ends short of them, go on into the NEXT range in the
             * same regroup (round 6).  Until then the lanes a range's last rays did not fill stayed idle until the following regroup: once
             * per range -- every sixteenth refill of a tile pipeline's 1 024-ray ranges, every sixth of a dump's 256.  Two passes cover every
             * case but a range clipped at its partition's end.  Config 4 108.1 -> 107.5 ms, config 5 48.2 -> 47.9 ms, S-soup-1M 2 263 -> 2 269
             * Mrays/s (profiles/r06_ab_refill.txt; LH_REFILL_PASSES=1: rounds 1-5).  It is NOT what makes short ranges at the end of a
             * launch lose: guided ranges lose as much with it (shares 7.23 / 7.60 / 7.67 -> 7.73 / 8.02 / 8.09 ms, and S-soup-1M -2 %:
             * the look at the cursor before the atomic is a second dependent trip to L2 per range). */
            const int need = __popcll(idle_mask);
            const int rank = __popcll(idle_mask & ((1ull << (tid & 63)) - 1ull));
            int filled = 0;
#pragma nounroll
            for (int pass = 0; pass < LH_REFILL_PASSES && filled < need && !exhausted; pass++) {
                if (__builtin_expect(wbase == wend, 0)) {
                    const uint32_t per = (n + LH_NPART - 1) / LH_NPART;
                    uint32_t chunk = sc.ray_chunk;
                    if ((SRC != 1 && sc.n_dev) || (SRC == 1 && ao.nslots_dev)) {                   /* the host sized the chunk for its upper bound of n */
                        const uint32_t c = n / (gridDim.x * (LH_BLOCK / 64) * 4u);
                        chunk = c < 64u ? 64u : (c < chunk ? c : chunk);
                    }
                    /* A partition found handed out is published in the word behind the cursors (a bit mask, zeroed with them), and a wave
                     * that runs dry reads that word before it probes further.  Without it every wave probed all LH_NPART cursors at
                     * the end of a launch: ~5000 waves x 8 device-scope atomics, serialised per address at ~95 ns each, were the
                     * ~0.5 ms "drain" of every launch -- an EMPTY launch of the path tracer's bounce chain took 0.49 ms
                     * (profiles/r03_pt_sky_timeline.csv) */
                    for (;;) {
                        if (drained == (1u << LH_NPART) - 1u) {       /* every partition has been handed out */
                            exhausted = true;
    #ifdef LH_DIAG_CLOCK
                            if (sc.diag_clock && (tid & 63) == 0) sc.diag_clock[(size_t)(2 * gridDim.x + blockIdx.x) * (LH_BLOCK / 64) + (tid >> 6)] = wall_clock64();
    #endif
                            break;
                        }
                        if (drained & (1u << part)) { part = (part + 1u) % LH_NPART; continue; }
                        const uint32_t p0 = per * part, p1 = (p0 + per < n) ? p0 + per : n;      /* per * LH_NPART < 2^31 + 8 */
                        const uint32_t plen = p1 > p0 ? p1 - p0 : 0u;                              /* a small batch leaves the last partitions empty */
                        /* (Round 6 tried GUIDED ranges here -- a look at the cursor, then min(chunk, what is left / (the partition's waves x k)) rays,
                         * never fewer than 64 -- to shrink the spread of the waves' exits: a rank's share of the config-5 frame 7.35 / 7.80 ms
                         * (rank 0 / 7) -> 7.96 / 8.16 (k = 1), 8.13 / 8.42 (k = 2), 8.57 / 8.60 (k = 4).  Not the atomics of the short ranges on ONE
                         * address: with 64 cursors, eight per XCD, the same rule costs the same (7.27 / 7.71 / 7.71 ms for ranks 0 / 3 / 7 ->
                         * 7.81 / 8.00 / 8.09 at k = 1; 64 cursors by themselves 7.54 / 7.73 / 7.93, the whole frame 48.3 -> 48.6 ms, S-soup-1M
                         * 2 266 -> 2 253 Mrays/s) -- short ranges at the end of a launch lose more in the walk than its exits gain.
                         * profiles/r06_share_probe.txt, r06_ab_cursors_guided.txt; tools/experiments/cursors_guided.patch.) */
                        uint32_t b = plen;
                        if (plen) {
                            if ((tid & 63) == 0) b = atomicAdd(cursor + part * LH_CURSOR_STRIDE, chunk);              /* < 2^31 + waves * chunk: no wrap */
                            b = (uint32_t)__shfl((int)b, 0);
                        }
                        if (b < plen) { b += p0; wbase = b; wend = (p1 - b > chunk) ? b + chunk : p1; break; }
                        drained |= 1u << part;
                        uint32_t seen = 0;
                        if ((tid & 63) == 0) {
                            seen = __hip_atomic_load(cursor + LH_NPART * LH_CURSOR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((seen & drained) != drained) atomicOr(cursor + LH_NPART * LH_CURSOR_STRIDE, drained);
                        }
                        drained |= (uint32_t)__shfl((int)seen, 0);
                        part = (part + 1u) % LH_NPART;
                    }
                }
                if (!exhausted) {
                    const uint32_t avail = wend - wbase;
                    const int take = avail < (uint32_t)(need - filled) ? (int)avail : need - filled;
                    if (idle && rank >= filled && rank < filled + take) newray = wbase + (uint32_t)(rank - filled);
                    wbase += (uint32_t)take; filled += take;
                }
            }
        }
        if (newray != kNoRay) {
            const uint32_t i = newray;
            my = i;
            src_ray<SRC>(sc, i, org, dir, ao, ox, oy, oz, dx, dy, dz, selfp);
            lane_init(L, sc, ox, oy, oz, dx, dy, dz);
            best.t = LH_T_INF; best.u = 0.0; best.v = 0.0; best.prim = LH_MISS_PRIM; best.frag = 0u;
            stk[0][tid] = kDone;
            /* rays beyond deg_dcap are the reference walk's (lh_walk.h).  cap_srcs (host-set, scalar): can a ray of THIS source exceed it at all --
             * dumps whenever it is finite; camera and AO rays (unit vectors) only where a zero-area triangle of |e1|_1 |e2|_1 > 1 stayed in the
             * tree.  Every other scene skips the test whole */
            if ((SRC == 0 ? (sc.cap_srcs & 1u) != 0u : sc.deg_dcap < 1.0f) && __builtin_expect(ray_needs_ref_walk(sc, L, dx, dy, dz), 0)) LH_FORCE_REF_WALK(L, best);
            it0 = it;
        }
        const unsigned long long work = __ballot((L.cur != kDone) | (pend != 0));
        if (work == 0ull) break;
        /* ---- walk until too few lanes remain active ---------------------- */
        const int thresh = exhausted ? 1 : min_active;
        if (WALK == 3)
            traverse_spec4<ANYHIT, COUNT, false, !(ANYHIT && LH_AO_UNSORTED)>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch, cns, cts, it);
        else if (WALK == 8)          /* the same with the stack check: trees whose worst case the LDS rows do not cover */
            traverse_spec4<ANYHIT, COUNT, true, !(ANYHIT && LH_AO_UNSORTED)>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch, cns, cts, it);
        else
            traverse_spec8<ANYHIT, COUNT>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch, cns, cts, it);
    }
#ifdef LH_DIAG_CLOCK
    if (sc.diag_clock && (tid & 63) == 0) sc.diag_clock[(size_t)(gridDim.x + blockIdx.x) * (LH_BLOCK / 64) + (tid >> 6)] = wall_clock64();
#endif
    /* this wave appends nothing more: its entries become visible to the concurrent consumer before the count does */
    if ((tid & 63) == 0) __hip_atomic_fetch_add(fq.qcount + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (COUNT) {
        add_counters(counters, cn, ct, ce, cr);
        atomicAdd(&counters[LH_CNT_NODE_SLOTS], (unsigned long long)cns);
        atomicAdd(&counters[LH_CNT_TRI_SLOTS], (unsigned long long)cts);
        atomicAdd(&counters[LH_CNT_REGROUP_SLOTS], (unsigned long long)crs);
    }
}

template <bool ANYHIT, bool COUNT, int WALK, int SRC>
__global__ __launch_bounds__(LH_BLOCK, WALK == 7 ? 3 : 4) void k_trace_persist_lane(
    lh_dev_scene_t sc, uint32_t n, const double *__restrict__ org, const double *__restrict__ dir,
    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
    double *__restrict__ v, uint8_t *__restrict__ occ, unsigned long long *counters,
    uint32_t *cursor, int min_active, int tri_batch, const AoSrc ao, const FixQ fq)
{
    extern __shared__ int lh_stack_lds[];          /* [stack entries][LH_BLOCK], sized at launch; then the top of the tree (sc.top_nodes x 64 bytes) */
    if (SRC != 1 && sc.n_dev) n = *sc.n_dev;       /* the path tracer's bounce chain: the count the previous shading pass left */
    if (SRC == 1 && ao.nslots_dev) n = (uint32_t)*ao.nslots_dev * (uint32_t)(ao.ntheta * ao.nphi);      /* the fused AO stage behind a compaction nobody read back (<= the host's bound < 2^31) */
    if (WALK != 7 && sc.top_nodes) {
        uint4 *dst = (uint4 *)(lh_stack_lds + (size_t)sc.stack_rows * LH_BLOCK);
        const uint4 *src = (const uint4 *)sc.q4nodes;
        for (uint32_t k = threadIdx.x; k < sc.top_nodes * 4u; k += LH_BLOCK) dst[k] = src[k];
        __syncthreads();
    }
    trace_persist_lane<ANYHIT, COUNT, WALK, SRC>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, ao, fq, lh_stack_lds);
}

/* ------------------------------------------------------------------------ */
/* the wave-cooperative walk: queued rays, 16 lanes each                      */
/* ------------------------------------------------------------------------ */
/* A ray the persistent kernel gave up on (visit budget, LDS rows) is traced again from the root by a group of lanes: its first lane
 * starts at the root; every iteration each busy lane takes one step of the default walk in its own LDS stack column, and
 * every idle lane takes over the BOTTOM entry of a busy lane's stack -- the largest unvisited subtree -- and walks it as
 * its own (same ray, own culling bound, own candidates).  Subtrees are disjoint, so every leaf is visited by exactly one
 * lane; the lanes' exact bests are merged at the end with resolve()'s rules (strictly smaller t wins, exact-t ties by the
 * reference's tree, nearly-equal t of two triangles marks the hit fragile).  Any-hit rays stop as soon as one lane holds a
 * certain hit.  Culling uses each lane's own bound (a donated subtree starts with the donor's), which is never tighter than
 * the sequential walk's: the same candidates or more reach the fp64 test, so the answer is the sequential walk's answer.
 * The chain of dependent node fetches of a long ray (thousands, ~1 us each) is cut by up to 16. */
#define LH_COOP_ROWS_MAX 272        /* 3 * 88 + 8: the deepest 4-wide tree the builders hand over */

__device__ __forceinline__ void merge_best(const lh_dev_scene_t &sc, Best &g, const Best &b, double dx, double dy, double dz)
{
    if (b.prim == LH_MISS_PRIM) return;
    if (g.prim == LH_MISS_PRIM) { g = b; return; }
    bool take = b.t < g.t;
    if (!take && b.t == g.t && b.prim != g.prim) take = tie_takes_new(sc, b.prim, g.prim, dx, dy, dz);
    const double tm = fabs(b.t) > fabs(g.t) ? fabs(b.t) : fabs(g.t);
    const uint32_t near2 = (b.prim != g.prim && b.t != g.t && fabs(b.t - g.t) <= LH_FRAGILE_REL * tm) ? 2u : 0u;
    const uint32_t sticky = ((g.frag | b.frag) & 2u) | near2;
    if (take) { g.t = b.t; g.u = b.u; g.v = b.v; g.prim = b.prim; g.frag = b.frag & 1u; }
    g.frag = (g.frag & 1u) | sticky;
}

/* the reference's own walk for one ray, out of line (its private stack stays out of the callers' frames) */
struct RefHit { double t, u, v; uint32_t prim; };
__device__ __noinline__ RefHit ref_trace_one(const lh_dev_scene_t &sc, double ox, double oy, double oz, double dx, double dy, double dz)
{
    RefHit h; uint32_t p = LH_MISS_PRIM; double tt = LH_T_INF, uu = 0.0, vv = 0.0;
    const int hit = lh_ref_trace((const lh_refnode_t *)sc.ref_nodes, (const uint32_t *)sc.ref_leaf_prims, (const double *)sc.tri64,
                                 sc.ref_empty, sc.ref_bmin, sc.ref_bmax, ox, oy, oz, dx, dy, dz, &p, &tt, &uu, &vv);
    h.prim = hit ? p : LH_MISS_PRIM; h.t = tt; h.u = uu; h.v = vv;
    return h;
}

/* One wave works on FOUR queued rays at a time, a group of 16 lanes each (most rays in the queue are only a little over
 * budget: a whole wave per ray would idle; donation stays inside the group); a group that has finished its ray takes the
 * next entry of its stride.
 * SRC 0: rays from org / dir, results into prim / t / u / v / occ.
 * SRC 1: AO rays regenerated from (slot, sample), occluded rays counted per slot; LH_Q_REF entries (and fragile results)
 * are decided by the reference's own walk, by the group's first lane. */
template <bool ANYHIT, int SRC>
__global__ __launch_bounds__(64) void k_coop_walk(lh_dev_scene_t sc, const double *__restrict__ org, const double *__restrict__ dir,
                                                  uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
                                                  double *__restrict__ v, uint8_t *__restrict__ occ, const AoSrc ao, const FixQ fq,
                                                  unsigned long long *counters, const uint32_t owner_groups)
{
    /* owner_groups == 0: the pass that runs NEXT TO the producer (second stream; few waves: it must not take the CUs from it).
     * Group g owns the queue entries g, g + groups, ...; where it stopped is left in fq.heads[g].  If the two streams turn out to
     * share a hardware queue this kernel only starts when the producer has finished: it then leaves at once, and the sweep --
     * owner_groups = that pass's group count, launched behind the producer on its own stream with a grid that fills the chip --
     * takes every entry no owner has taken (e >= heads[e mod owner_groups]).  (A serialised small pass over the 475 000 queued
     * rays of a BASELINE config-5 frame took 30 ms: bench.py's device-tree frame, 89 -> 122 ms, until r03.)
     * Round 5 tried ONE consumer instead -- the sweep's grid on the second stream, working the queue off to its end, nothing
     * launched afterwards (a sweep costs 65-73 us even with nothing to do) -- and lost: a rank's share of the config-5 frame
     * 8.43 -> 8.65 ms, the path-traced frame 128.8 -> 130.1 ms, S-soup-1M 2 247 -> 2 236 Mrays/s (profiles/README.md r05). */
    extern __shared__ int lh_stack_lds[];          /* [rows][64] stack + 2 x 64 exchange words */
    const int rows = (int)sc.stack_rows, rmask = rows - 1;          /* rows: a power of two (ring of stack positions) */
    int (*stk)[64] = (int (*)[64])lh_stack_lds;
    int *xref = lh_stack_lds + (size_t)rows * 64; float *xtb = (float *)(xref + 64);
    uint32_t *stash = (uint32_t *)(xref + 128) + 16 * (threadIdx.x >> 4);          /* this group's stashed LH_Q_REF entries (below) */
    int ns = 0;                                                                       /* group-uniform: how many */
    const int lane = threadIdx.x, g = lane >> 4;
    const unsigned long long gmask = 0xFFFFull << (16 * g), lt_mask = (1ull << lane) - 1ull;
    const uint32_t ngroups = gridDim.x * 4u;
    const uint32_t gid = blockIdx.x * 4u + (uint32_t)g;
    if (owner_groups == 0u) {
        uint32_t left = 0;
        if (lane == 0) left = __hip_atomic_load(fq.qcount + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)__shfl((int)left, 0) >= fq.nprod) return;          /* the producer is gone already: the sweep is faster */
    }
    uint32_t e = owner_groups ? gid : fq.heads[gid];  /* the group's next queue entry: gid, gid + ngroups, ... */
    if (owner_groups) while (e < fq.qcap && e < fq.heads[e % owner_groups]) e += ngroups;
    bool have = false;                                /* the group holds a ray */
    bool gdone = e >= fq.qcap;                        /* the group has seen the end of the queue */
    uint32_t known = 0, look = 0;                     /* wave-uniform: the append count at the wave's last look; iterations since */
    bool prod_done = false;                           /* wave-uniform: every producer wave was seen to have left */
    if (owner_groups) {                               /* the sweep runs behind the producer: the count is final */
        uint32_t cnt = 0;
        if (lane == 0) cnt = __hip_atomic_load(fq.qcount, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        cnt = (uint32_t)__shfl((int)cnt, 0);
        known = cnt < fq.qcap ? cnt : fq.qcap; prod_done = true;
        if (e >= known) gdone = true;
    }
    unsigned long long progress = wall_clock64();     /* when this wave last took an entry / saw the producer finish */
    size_t i = 0; double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1; uint32_t selfp = LH_MISS_PRIM;
    Lane L; Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
    int pend = kNoLeaf, floor_ = 1; uint32_t cn = 0, ct = 0, ce = 0;
    L.cur = kDone; L.sp = 1; L.np = 0; L.certain = false; L.over = false;
    stk[0][lane] = kDone;

    for (;;) {
        const unsigned long long m_busy0 = __ballot((L.cur != kDone) | (pend != kNoLeaf));
        if ((m_busy0 & gmask) == 0ull) {                 /* group-uniform: nothing left to walk */
            if (have) {
                /* every lane's candidates through the fp64 test, then the merge over the group */
                bool need_ref = false, hit = false;
                finish<ANYHIT, false>(L, sc, ox, oy, oz, dx, dy, dz, best, ce, selfp);
                if (ANYHIT) {
                    const bool exact = best.prim != LH_MISS_PRIM;
                    const bool sure = (__ballot(L.certain || (exact && best.frag == 0u)) & gmask) != 0ull;
                    hit = sure;
                    if (!sure && (__ballot(exact) & gmask) != 0ull) { need_ref = sc.ref_nodes != NULL; hit = true; }   /* only fragile hits: the reference walk decides */
                    if (SRC != 1 && (lane & 15) == 0) {
                        if (need_ref) hit = ref_trace_one(sc, ox, oy, oz, dx, dy, dz).prim != LH_MISS_PRIM;
                        occ[i] = (uint8_t)(hit ? 1 : 0);
                    }
                } else {
                    Best gb = best;
                    for (int off = 1; off < 16; off <<= 1) {
                        Best b;
                        b.t = __shfl_xor(gb.t, off); b.u = __shfl_xor(gb.u, off); b.v = __shfl_xor(gb.v, off);
                        b.prim = (uint32_t)__shfl_xor((int)gb.prim, off); b.frag = (uint32_t)__shfl_xor((int)gb.frag, off);
                        merge_best(sc, gb, b, dx, dy, dz);
                    }
                    if ((lane & 15) == 0) {
                        if (sc.ref_nodes != NULL && gb.prim != LH_MISS_PRIM && gb.frag != 0u) {        /* a fragile hit: the reference walk decides */
                            const RefHit rh = ref_trace_one(sc, ox, oy, oz, dx, dy, dz);
                            gb.prim = rh.prim; gb.t = rh.t; gb.u = rh.u; gb.v = rh.v;
                        }
                        prim[i] = gb.prim; t[i] = gb.t; u[i] = gb.u; v[i] = gb.v;
                    }
                }
                if (SRC == 1 && (lane & 15) == 0) {
                    if (need_ref) hit = ref_trace_one(sc, ox, oy, oz, dx, dy, dz).prim != LH_MISS_PRIM;
                    if (hit) { uint32_t sl, rr; ao_item(ao, (uint32_t)i, sl, rr); atomicAdd(&ao.occ_count[sl], 1u); }
                }
                if (counters && (lane & 15) == 0) atomicAdd(&counters[LH_CNT_RETRACED], 1ull);
                have = false;
            }
            /* the producer may still be running.  `known` is the wave's last look at the append count (refreshed rarely: every
             * look is a load of ONE word that 256 waves share); a slot below it is, or is about to be, non-zero */
            unsigned long long ent = 0ull;
            if (!gdone && e < known) {
                ent = __hip_atomic_load(fq.queue + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (ent != 0ull) {
                const uint32_t reason = (uint32_t)(ent >> 56);
                i = (size_t)(ent & 0x00FFFFFFFFFFFFFFull);
                e += ngroups;
                if (owner_groups) while (e < fq.qcap && e < fq.heads[e % owner_groups]) e += ngroups;
                if (e >= fq.qcap || (owner_groups && e >= known)) gdone = true;
                progress = wall_clock64();
                if (reason == LH_Q_REF) {
                    /* a fragile hit, or a ray beyond deg_dcap: the reference's own walk on its own tree decides, no cooperative walk.
                     * The entry is STASHED: the group walks its stashed rays sixteen at a time, one per lane (below) -- until round 6
                     * the group's first lane walked every such ray by itself while fifteen watched: 8 M rays/s for a launch whose
                     * rays all came this way (a zero-area triangle through the scene: profiles/r06_degenerate_cliff.txt) */
                    if ((lane & 15) == 0) stash[ns] = (uint32_t)i;
                    ns++;
                } else {
                    src_ray<SRC>(sc, (uint32_t)i, org, dir, ao, ox, oy, oz, dx, dy, dz, selfp);
                    lane_init(L, sc, ox, oy, oz, dx, dy, dz);
                    best.t = LH_T_INF; best.u = 0.0; best.v = 0.0; best.prim = LH_MISS_PRIM; best.frag = 0u;
                    pend = kNoLeaf; floor_ = 1; stk[0][lane] = kDone;
                    L.cur = (lane & 15) == 0 ? 0 : kDone;
                    have = true;
                }
            }
            /* the stashed reference walks: when sixteen have come together, when the queue has nothing for this group right now, or when it
             * has just ended for it (the wave leaves the loop once every group has seen the end: nothing may stay stashed) */
            if (ns == 16 || (ns > 0 && (ent == 0ull || gdone || have))) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if ((lane & 15) < ns) {
                    const uint32_t ri = stash[lane & 15];
                    double rox, roy, roz, rdx, rdy, rdz; uint32_t rself = LH_MISS_PRIM;
                    src_ray<SRC>(sc, ri, org, dir, ao, rox, roy, roz, rdx, rdy, rdz, rself);
                    const RefHit rh = ref_trace_one(sc, rox, roy, roz, rdx, rdy, rdz);
                    if (counters) atomicAdd(&counters[LH_CNT_RETRACED], 1ull);
                    if (SRC == 1) { if (rh.prim != LH_MISS_PRIM) { uint32_t sl, rr; ao_item(ao, ri, sl, rr); atomicAdd(&ao.occ_count[sl], 1u); } }
                    else if (ANYHIT) occ[ri] = rh.prim != LH_MISS_PRIM ? 1 : 0;
                    else { prim[ri] = rh.prim; t[ri] = rh.t; u[ri] = rh.u; v[ri] = rh.v; }
                }
                ns = 0;
                progress = wall_clock64();
            }
        }
        /* a look at the queue's counters: when the whole wave is idle (after a nap of ~25 us), else every 128 iterations */
        const bool wave_idle = __ballot(have) == 0ull;
        if (wave_idle || (++look & 127u) == 0u) {
            if (wave_idle) {
                if (__ballot(!gdone) == 0ull) break;                          /* the queue has ended for every group */
                /* never wait for ever: if the producer makes no progress for half a second (it may not be running at all: two
                 * streams can share a hardware queue, and then this kernel runs in front of it) leave -- the sweep launched
                 * behind the producer takes what is left */
                /* (only the pass beside the producer: the sweep works the queue off to its end however long an entry takes -- a
                 * reference walk through 300 000 triangles that share a vertex runs for longer than this in ONE lane, and a sweep that
                 * gave up after it left 110 000 of 157 000 queued any-hit rays of such a scene without an answer: tools/fuzz_parity.py,
                 * big scenes, seed 41 round 7) */
                if (owner_groups == 0u && __ballot(wall_clock64() - progress <= (sc.coop_patience ? (unsigned long long)sc.coop_patience : 50000000ull)) == 0ull) break;          /* wave-uniform: no group has seen anything for 0.5 s */
                if (__ballot(!gdone && e < known) != 0ull) { __builtin_amdgcn_s_sleep(16); continue; }     /* a slot below the known count: being written, or skipped above */
                for (int k = 0; k < 8; k++) __builtin_amdgcn_s_sleep(127);
            }
            uint32_t left = 0, cnt = 0;
            if (lane == 0) {
                left = __hip_atomic_load(fq.qcount + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                cnt = __hip_atomic_load(fq.qcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            left = (uint32_t)__shfl((int)left, 0); cnt = (uint32_t)__shfl((int)cnt, 0);
            known = cnt < fq.qcap ? cnt : fq.qcap;
            if (left >= fq.nprod) {                                /* every producer wave has left (acquire): the count is final */
                /* the pass next to the producer takes no new entry from here on: its few waves had the CUs' spare room while the
                 * producer ran; now the chip is empty, and the sweep behind the producer -- sixteen times the waves -- takes the
                 * backlog from heads[] on (a producer with three workgroups per CU outruns this pass by a third of the launch) */
                if (owner_groups == 0u) gdone = true;
                if (!prod_done) { prod_done = true; progress = wall_clock64(); continue; }      /* one more look at the count, now final */
                if (e >= known) gdone = true;
            }
            if (wave_idle) continue;
        }
        if (L.cur >= 0) node_step4<false, 64, true>(L, pend, sc, stk, lane, cn, rmask);
        const unsigned long long m_node = __ballot(L.cur >= 0);
        const unsigned long long m_pend = __ballot(pend != kNoLeaf);
        if (m_pend != 0ull && (__popcll(m_pend) >= 8 || m_node == 0ull))
            tri_pass<ANYHIT, false, 64, true>(L, pend, sc, stk, lane, ox, oy, oz, dx, dy, dz, best, ct, ce, rmask);
        if (ANYHIT) {                                    /* a certain hit ends the group's ray */
            if ((__ballot(L.certain) & gmask) != 0ull) { L.cur = kDone; pend = kNoLeaf; }
        }
        /* donation inside a group: its k-th idle lane takes the bottom stack entry of its k-th lane that has one to give */
        const bool busy = (L.cur != kDone) | (pend != kNoLeaf);
        const bool donor = busy && L.sp > floor_;
        const unsigned long long m_donor = __ballot(donor), m_idle = __ballot(!busy && have);
        const unsigned long long gd = m_donor & gmask, gi = m_idle & gmask;
        if (__ballot(gd != 0ull && gi != 0ull) != 0ull) {
            const int nd = __popcll(gd), ni = __popcll(gi);
            const int ngive = nd < ni ? nd : ni;
            const int drank = __popcll(gd & lt_mask), irank = __popcll(gi & lt_mask);
            if (donor && drank < ngive) {
                xref[16 * g + drank] = stk[floor_ & rmask][lane]; xtb[16 * g + drank] = L.tb;
                stk[floor_ & rmask][lane] = kDone; floor_++;    /* the slot becomes the stack-bottom sentinel */
            }
            __syncthreads();
            if (!busy && have && irank < ngive) {
                const int r = xref[16 * g + irank];
                L.tb = fminf(L.tb, xtb[16 * g + irank]);
                L.sp = 1; floor_ = 1; stk[0][lane] = kDone;
                const bool is_leaf = (r < 0) & (r != kDone);
                pend = is_leaf ? r : kNoLeaf;
                L.cur = is_leaf ? kDone : r;
            }
            __syncthreads();
        }
    }
    if (owner_groups == 0u && (lane & 15) == 0) fq.heads[gid] = e;          /* where the sweep goes on */
}

/* ------------------------------------------------------------------------ */
/* one scan over the outputs of a ray dump: what is still flagged            */
/* ------------------------------------------------------------------------ */
/* LH_PRIM_RETRACE / LH_OCC_RETRACE: a hit the reference may not reach -- the reference's own walk on its own tree decides.
 * LH_PRIM_OVERFLOW / LH_OCC_OVERFLOW: only when the fix-up queue was full (more than qcap rays out of budget in one launch) --
 * the same walk as the kernel's, sequential, with a private stack. */
/* ray i of a launch whose output slot was left flagged (the cold path: arrays, or the camera rays of a path-traced pass) */
__device__ __forceinline__ void flagged_ray(const lh_dev_scene_t &sc, size_t i, const double *__restrict__ org, const double *__restrict__ dir,
                                            double &ox, double &oy, double &oz, double &dx, double &dy, double &dz)
{
    if (sc.cam_src) {
        double o[3], d[3];
        pt_camera_ray((const PtCamSrc *)sc.cam_src, (uint32_t)i, o, d);
        ox = o[0]; oy = o[1]; oz = o[2]; dx = d[0]; dy = d[1]; dz = d[2];
    } else {
        ox = org[3 * i]; oy = org[3 * i + 1]; oz = org[3 * i + 2]; dx = dir[3 * i]; dy = dir[3 * i + 1]; dz = dir[3 * i + 2];
    }
}

/* cnt (or NULL): this ray's 4-wide node visits, leaf visits, triangle records through the fp32 filter, fp64 tests -- the walk
 * is sequential (nearest child first, a leaf's triangles in order), so the counts are the host model's (tests/cpu_model) */
template <bool ANYHIT>
__device__ void overflow_walk(const lh_dev_scene_t &sc, size_t i, const double *__restrict__ org, const double *__restrict__ dir,
                              uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u, double *__restrict__ v,
                              uint8_t *__restrict__ occ, uint32_t *cnt = NULL)
{
    double ox, oy, oz, dx, dy, dz;
    flagged_ray(sc, i, org, dir, ox, oy, oz, dx, dy, dz);
    const float4 *__restrict__ tris = (const float4 *)sc.tri32;
    Lane L; Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
    uint32_t ce = 0, cn = 0, cl = 0, ct = 0;
    int stack[LH_COOP_ROWS_MAX]; int sp = 0;
    lane_init(L, sc, ox, oy, oz, dx, dy, dz);
    int cur = 0;
    const bool refw = ray_needs_ref_walk(sc, L, dx, dy, dz);          /* the reference's own walk decides (lh_walk.h) */
    if (refw) { best.prim = 0u; best.frag = 1u; }
    else for (;;) {
        if (cur >= 0) {
            const uint4 *p = (const uint4 *)sc.q4nodes + 4 * (size_t)cur;
            const uint4 a = p[0], b = p[1], c = p[2], r = p[3];
            float tn[4]; bool h[4]; const int ref[4] = {(int)r.x, (int)r.y, (int)r.z, (int)r.w};
            cn++;
            h[0] = slab_w(L, a.x, a.y, a.z, tn[0]) & (ref[0] != kDone);
            h[1] = slab_w(L, a.w, b.x, b.y, tn[1]) & (ref[1] != kDone);
            h[2] = slab_w(L, b.z, b.w, c.x, tn[2]) & (ref[2] != kDone);
            h[3] = slab_w(L, c.y, c.z, c.w, tn[3]) & (ref[3] != kDone);
            /* nearest first, by the key of the default walk's node step (distance bits with the slot in the two low bits) */
            int order[4], nh = 0; uint32_t key[4];
            for (int k = 0; k < 4; k++) if (h[k]) {
                key[k] = (__float_as_uint(tn[k]) & ~3u) | (uint32_t)k;
                int m = nh++;
                while (m > 0 && key[order[m - 1]] > key[k]) { order[m] = order[m - 1]; m--; }
                order[m] = k;
            }
            for (int k = nh - 1; k >= 1; k--) if (sp < LH_COOP_ROWS_MAX) stack[sp++] = ref[order[k]];
            if (nh) cur = ref[order[0]];
            else if (sp) cur = stack[--sp];
            else break;
        } else {
            const uint32_t x = ~(uint32_t)cur, first = x >> 2, cnt_ = (x & 3u) + 1u;
            bool finished = false;
            cl++;
            for (uint32_t k = 0; k < cnt_ && !finished; k++) {
                const float4 *tp = tris + 3 * (size_t)(first + k);
                const float4 ta = tp[0], tb_ = tp[1], tc = tp[2];
                ct++;
                finished = tri_step<ANYHIT, true>(L, sc, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w,
                                                  __float_as_uint(tc.y), ox, oy, oz, dx, dy, dz, best, ce);
            }
            if (finished || sp == 0) break;
            cur = stack[--sp];
        }
    }
    L.over = false;
    finish<ANYHIT, true>(L, sc, ox, oy, oz, dx, dy, dz, best, ce);
    write_out<ANYHIT>(i, L, best, prim, t, u, v, occ, sc.ref_nodes != NULL);
    if (cnt) { cnt[0] = cn; cnt[1] = cl; cnt[2] = ct; cnt[3] = ce; }
}

__global__ __launch_bounds__(256) void k_fixups(lh_dev_scene_t sc, size_t n, const double *__restrict__ org,
                                                const double *__restrict__ dir, uint32_t *__restrict__ prim,
                                                double *__restrict__ t, double *__restrict__ u, double *__restrict__ v,
                                                uint8_t *__restrict__ occ, int anyhit, unsigned long long *counters,
                                                const uint32_t *qcount, int force)
{
    /* nothing was left flagged unless a push found the queue full (or the launch had no queue: force) */
    if (!force && qcount[1] == 0u) return;
    if (sc.n_dev && (size_t)*sc.n_dev < n) n = *sc.n_dev;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (anyhit ? (occ[i] == LH_OCC_OVERFLOW) : (prim[i] == LH_PRIM_OVERFLOW)) {
            if (anyhit) overflow_walk<true>(sc, i, org, dir, prim, t, u, v, occ);
            else overflow_walk<false>(sc, i, org, dir, prim, t, u, v, occ);
            if (counters) atomicAdd(&counters[LH_CNT_RETRACED], 1ull);
        }
        if (sc.ref_nodes == NULL) continue;
        if (anyhit ? (occ[i] != LH_OCC_RETRACE) : (prim[i] != LH_PRIM_RETRACE)) continue;
        double ox, oy, oz, dx, dy, dz;
        flagged_ray(sc, i, org, dir, ox, oy, oz, dx, dy, dz);
        const RefHit rh = ref_trace_one(sc, ox, oy, oz, dx, dy, dz);
        if (anyhit) occ[i] = rh.prim != LH_MISS_PRIM ? 1 : 0;
        else { prim[i] = rh.prim; t[i] = rh.t; u[i] = rh.u; v[i] = rh.v; }
        if (counters) atomicAdd(&counters[LH_CNT_RETRACED], 1ull);
    }
}

/* ------------------------------------------------------------------------ */
/* a handful of rays: one launch, one lane per ray                            */
/* ------------------------------------------------------------------------ */
/* lucille's synchronous accel->intersect (raytrace.c:31-69) reaches the device as batches of at most a few dozen rays (the
 * render threads' concurrent calls, coalesced: lh_query.hip).  The persistent kernel is the wrong tool there -- a cursor reset,
 * a fix-up queue reset, 768 workgroups that find nothing to do, a scan for flagged rays: four launches and ~100 us for
 * microseconds of work.  Here every ray gets a wave of ONE small launch and walks the 4-wide nodes sequentially with a private
 * stack (overflow_walk: the same filter, the same fp64 resolve, the same answer); a fragile hit goes through the
 * reference's own walk in the same lane. */
#define LH_SMALL_BATCH 64u
/* PER_WAVE: one ray per wave (lane 0), else one per lane.  counters (or NULL): the launch's totals (LH_CNT_*); sc.diag_out (or
 * NULL): four counts per ray -- ri_bvh_diag_t's numbers (bvh.h:103-110) for this build's tree */
template <bool ANYHIT, bool PER_WAVE>
__global__ __launch_bounds__(64) void k_trace_small(lh_dev_scene_t sc, uint32_t n, const double *__restrict__ org, const double *__restrict__ dir,
                                                    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
                                                    double *__restrict__ v, uint8_t *__restrict__ occ, unsigned long long *counters)
{
    /* ONE RAY PER WAVE: sixty-four unrelated rays in the lanes of one wave walk in lockstep through each other's branches
     * (16 rays: 243 us against 49 us for one, r04); a wave per ray runs them side by side on as many CUs -- the batch takes
     * as long as its longest ray.  63 idle lanes per wave are free here: the batch is tiny and the chip is empty. */
    const uint32_t i = PER_WAVE ? blockIdx.x : blockIdx.x * 64u + threadIdx.x;
    if (i >= n || (PER_WAVE && threadIdx.x != 0)) return;
    uint32_t cnt[4] = {0u, 0u, 0u, 0u};
    overflow_walk<ANYHIT>(sc, i, org, dir, prim, t, u, v, occ, cnt);
    if (sc.diag_out) { uint32_t *d = sc.diag_out + 4 * (size_t)i; d[0] = cnt[0]; d[1] = cnt[1]; d[2] = cnt[2]; d[3] = cnt[3]; }
    if (counters) add_counters(counters, cnt[0], cnt[2], cnt[3], 1);
    if (sc.ref_nodes == NULL) return;
    if (ANYHIT ? (occ[i] != LH_OCC_RETRACE) : (prim[i] != LH_PRIM_RETRACE)) return;
    double ox, oy, oz, dx, dy, dz;
    flagged_ray(sc, i, org, dir, ox, oy, oz, dx, dy, dz);
    const RefHit rh = ref_trace_one(sc, ox, oy, oz, dx, dy, dz);
    if (ANYHIT) occ[i] = rh.prim != LH_MISS_PRIM ? 1 : 0;
    else { prim[i] = rh.prim; t[i] = rh.t; u[i] = rh.u; v[i] = rh.v; }
    if (counters) atomicAdd(&counters[LH_CNT_RETRACED], 1ull);
}

/* a workgroup's LDS stack beyond 64 KiB (trees deeper than 19 four-wide levels: up to LH_ROWS_UNCHECKED rows) has to be
 * allowed per kernel function, once */
static int allow_lds(const void *func, size_t bytes)
{
    static const void *done[64]; static int ndone = 0; static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    if (bytes <= 64 * 1024) return 0;
    pthread_mutex_lock(&mu);
    bool have = false;
    for (int k = 0; k < ndone; k++) have = have || done[k] == func;
    int rc = 0;
    if (!have) {
        rc = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LH_ROWS_UNCHECKED * LH_BLOCK * sizeof(int) + LH_TOP_NODES_MAX * 64u)) == hipSuccess ? 0 : -1;
        if (rc == 0 && ndone < 64) done[ndone++] = func;
    }
    pthread_mutex_unlock(&mu);
    return rc;
}
#define LH_LAUNCH_PERSIST(KERNEL, ...) do { if (allow_lds((const void *)(KERNEL), lds_bytes) != 0) return -1; \
                                            hipLaunchKernelGGL((KERNEL), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s, __VA_ARGS__); } while (0)

template <bool ANYHIT, bool COUNT>
int launch_one(const lh_dev_scene_t &sc, size_t n, const double *org, const double *dir,
               uint32_t *prim, double *t, double *u, double *v, uint8_t *occ,
               unsigned long long *counters, unsigned long long *cursor, int walk,
               int grid_blocks, int min_active, int tri_batch, size_t lds_bytes, const FixQ &fq, hipStream_t s)
{
    if (walk == 0) {
        const size_t blocks = (n + LH_BLOCK - 1) / LH_BLOCK;
        if (blocks > 0x7fffffffull) return -1;
        hipLaunchKernelGGL((k_trace_direct<ANYHIT, COUNT>), dim3((unsigned)blocks), dim3(LH_BLOCK), lds_bytes, s,
                           sc, n, org, dir, prim, t, u, v, occ, counters);
    } else {
        if (hipMemsetAsync(cursor, 0, sizeof(uint32_t) * LH_CURSOR_WORDS, s) != hipSuccess) return -1;
        if (sc.cam_src) {                    /* ray source 2: closest hit over the 4-wide nodes only (lh_launch_trace checks) */
            if (walk == 8)
                LH_LAUNCH_PERSIST((k_trace_persist_lane<false, COUNT, 8, 2>), sc, (uint32_t)n, org, dir, prim, t, u, v, occ, counters, (uint32_t *)cursor, min_active, tri_batch, AoSrc{}, fq);
            else
                LH_LAUNCH_PERSIST((k_trace_persist_lane<false, COUNT, 3, 2>), sc, (uint32_t)n, org, dir, prim, t, u, v, occ, counters, (uint32_t *)cursor, min_active, tri_batch, AoSrc{}, fq);
        } else if (walk == 7)
            LH_LAUNCH_PERSIST((k_trace_persist_lane<ANYHIT, COUNT, 7, 0>), sc, (uint32_t)n, org, dir, prim, t, u, v, occ, counters, (uint32_t *)cursor, min_active, tri_batch, AoSrc{}, fq);
        else if (walk == 8)
            LH_LAUNCH_PERSIST((k_trace_persist_lane<ANYHIT, COUNT, 8, 0>), sc, (uint32_t)n, org, dir, prim, t, u, v, occ, counters, (uint32_t *)cursor, min_active, tri_batch, AoSrc{}, fq);
        else
            LH_LAUNCH_PERSIST((k_trace_persist_lane<ANYHIT, COUNT, 3, 0>), sc, (uint32_t)n, org, dir, prim, t, u, v, occ, counters, (uint32_t *)cursor, min_active, tri_batch, AoSrc{}, fq);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_walk(const lh_dev_scene_t &sc, size_t n, const double *org, const double *dir,
                uint32_t *prim, double *t, double *u, double *v, int anyhit, uint8_t *occ,
                unsigned long long *counters, unsigned long long *cursor, int walk,
                int grid_blocks, int min_active, int tri_batch, size_t lds_bytes, const FixQ &fq, hipStream_t s)
{
    if (anyhit) {
        if (counters) return launch_one<true, true>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, walk, grid_blocks, min_active, tri_batch, lds_bytes, fq, s);
        return launch_one<true, false>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, walk, grid_blocks, min_active, tri_batch, lds_bytes, fq, s);
    }
    if (counters) return launch_one<false, true>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, walk, grid_blocks, min_active, tri_batch, lds_bytes, fq, s);
    return launch_one<false, false>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, walk, grid_blocks, min_active, tri_batch, lds_bytes, fq, s);
}

/* LDS stack rows of a 4-wide walk over this scene.  3 * depth + 5 (or the builder's count of the deepest path) covers every
 * ray, but a launch that fills the chip is bound by rays in flight, and 128 VGPRs allow FOUR workgroups per CU: so a big launch
 * (`dense`) walks at most LH_ROWS_CHECKED rows -- four workgroups and the cooperative walk's in a CU's 160 KiB -- with a check
 * before every push, and the rare ray that would overrun its column goes to the cooperative walk.  Config-5 frame (the tree
 * asks for 59 rows: two workgroups per CU): 85.6 ms unchecked, 61.9 at 34 checked rows; S-soup-1M dump (44 rows: three) 2 234 ->
 * 2 278 Mrays/s.  (Rounds 2-3 measured the opposite for the frame -- 91 against 87 ms -- because the cooperative pass beside
 * the launch, which cannot be resident next to four 128-VGPR workgroups, kept its share of the queue after the launch had ended:
 * k_coop_walk.)  Small batches keep unchecked rows (up to 64): a handful of waves fills nothing, and without a queue an
 * overrunning ray would be left to k_fixups' sequential walk.  "stack_cap" forces a cap (8 .. 64). */
uint32_t rows4(const lh_dev_scene_t &sc, bool *guard, bool dense = true)
{
    uint32_t need = sc.q4_stack ? sc.q4_stack : 3 * sc.q4_depth + 5;      /* the builder's own count of the deepest path, or the bound of any tree that deep */
    const uint32_t cap = (sc.stack_cap >= 8 && sc.stack_cap <= LH_ROWS_UNCHECKED) ? sc.stack_cap
                       : ((dense || need > LH_ROWS_UNCHECKED) ? LH_ROWS_CHECKED : LH_ROWS_UNCHECKED);
    *guard = need > cap;
    if (need > cap) need = cap;
    need = (need + 1u) & ~1u;
    if (need < 16 && !*guard) need = 16;
    return need;
}

/* the cooperative walk's ring of stack rows: a power of two covering the tree's worst case (3 per level + sentinel + the
 * step's scratch slots); 0: the tree is deeper than any builder hands over */
uint32_t coop_rows(const lh_dev_scene_t &sc)
{
    const uint32_t need = 3 * sc.q4_depth + 6;
    uint32_t r = 16;
    while (r < need) r <<= 1;
    return r <= 512 ? r : 0;
}

/* nodes of the top of the tree a workgroup keeps in LDS behind its `rows` stack rows (node_step4): what the CU's 160 KiB leave
 * over once the workgroups its stack rows allow are resident -- never a workgroup fewer for it.  Measured (tools/top_probe.py,
 * r04): S-soup-1M 2 228 -> 2 249 Mrays/s on the host builder's tree, 2 212 -> 2 233 on the device builder's (144 nodes = 6.3 of a
 * ray's 42.5 node visits), config-5 AO frame 86.8 -> 86.2 ms: the requests that leave the CU for the hot top levels are not what
 * bounds the walk (profiles/README.md r04), so the gain is small; it is free.  sc.top_nodes: LH_TOP_AUTO, 0 (off), or a count. */
uint32_t top_nodes_for(const lh_dev_scene_t &sc, uint32_t rows)
{
    uint32_t want = sc.top_nodes;
    if (want == 0u) return 0u;
    /* the CU's 160 KiB also hold the workgroup of the cooperative walk that runs NEXT TO this launch (k_coop_walk: one wave,
     * a ring of coop_rows x 64 entries): its stream has the higher priority, so it is placed first -- a persistent workgroup
     * that no longer fits beside it is one workgroup per CU fewer for the whole launch (r04: 2 240 -> 1 680 Mrays/s, the
     * config-5 frame 85 -> 128 ms with every spare byte given to the top of the tree).  A count set by the caller is held to
     * the same room (512 nodes beside 34 dense rows made 4 x 43 KiB and silently dropped a workgroup per CU) */
    const uint32_t crows = coop_rows(sc);
    const uint32_t coop = crows ? (crows * 64u + 128u) * (uint32_t)sizeof(int) + 1024u : 0u;
    const uint32_t cu = 160u * 1024u - coop, stack = rows * LH_BLOCK * (uint32_t)sizeof(int);
    uint32_t wgs = (160u * 1024u) / stack;      /* what size_grid() launches per CU */
    if (wgs > 4u) wgs = 4u;                     /* 128 VGPRs: four workgroups per CU at most */
    if (wgs == 0u || cu < wgs * stack) return 0u;
    uint32_t spare = (cu / wgs - stack) / 64u;
    /* a workgroup above 64 KiB of LDS halves what a CU holds (lh_device.h): the copy never takes a launch across that line (the
     * small launches' up to 64 unchecked rows are 64 KiB by themselves: no copy there) */
    if (stack <= 64u * 1024u) { const uint32_t room = (64u * 1024u - stack) / 64u; if (spare > room) spare = room; }
    if (spare > LH_TOP_NODES_MAX) spare = LH_TOP_NODES_MAX;
    if (want == LH_TOP_AUTO || want > spare) want = spare;
    want &= ~15u;
    return want < sc.nq4nodes ? want : sc.nq4nodes;
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

/* the cooperative walk over the fix-up queue of the persistent launch just enqueued on s.  It runs on the queue's second
 * stream, CONCURRENTLY with that launch (a few small workgroups per CU next to the persistent ones: its low efficiency --
 * one ray per 16 lanes, restarted from the root -- is hidden behind the bulk of the frame), submitted AFTER it: if the two
 * streams share a hardware queue it simply runs afterwards. */
template <bool ANYHIT, int SRC>
int launch_coop(const lh_dev_scene_t &sc, const double *org, const double *dir, uint32_t *prim, double *t, double *u, double *v,
                uint8_t *occ, const AoSrc &ao, const FixQ &fq, const lh_fixq_t *q, unsigned long long *counters, int ncus, hipStream_t s)
{
    lh_dev_scene_t scl = sc;
    scl.stack_rows = coop_rows(sc);
    if (scl.stack_rows == 0) return -1;
    const size_t lds = ((size_t)scl.stack_rows * 64 + 128 + 64) * sizeof(int);          /* the stack ring, 2 x 64 exchange words, 4 x 16 stashed reference-walk entries */
    static bool attr_set[2][3] = {{false, false, false}, {false, false, false}};
    if (lds > 64 * 1024 && !attr_set[ANYHIT][SRC]) {
        if (hipFuncSetAttribute((const void *)k_coop_walk<ANYHIT, SRC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        attr_set[ANYHIT][SRC] = true;
    }
    hipStream_t aux = (hipStream_t)q->aux_stream;
    int grid = ncus > 0 ? ncus : 256;                      /* one wave per workgroup and CU, 4 rays per wave */
    if (grid * 4 > (int)LH_Q_GROUPS) grid = (int)LH_Q_GROUPS / 4;
    static int concurrent = -1;
    if (concurrent < 0) { const char *e = getenv("LH_COOP_CONCURRENT"); concurrent = (e && atoi(e) == 0) ? 0 : 1; }
    if (concurrent) {
        if (hipStreamWaitEvent(aux, (hipEvent_t)q->ev_ready, 0) != hipSuccess) return -1;
        hipLaunchKernelGGL((k_coop_walk<ANYHIT, SRC>), dim3(grid), dim3(64), lds, aux, scl, org, dir, prim, t, u, v, occ, ao, fq, counters, 0u);
        if (hipGetLastError() != hipSuccess) return -1;
        if (hipEventRecord((hipEvent_t)q->ev_done, aux) != hipSuccess) return -1;
        if (hipStreamWaitEvent(s, (hipEvent_t)q->ev_done, 0) != hipSuccess) return -1;
    }
    /* the sweep: the same kernel behind the producer, on its stream, sixteen waves per CU -- whatever the concurrent pass did not
     * take (nothing, when it ran next to the producer: the waves read a few words and leave) */
    /* (a sweep of grid x 4 or x 2 workgroups instead of x 16: a rank's share of the config-5 frame 7.35 / 7.80 -> 7.35 / 7.79 and 7.31 / 7.73 ms: nothing.  r06_share_probe.txt) */
    hipLaunchKernelGGL((k_coop_walk<ANYHIT, SRC>), dim3(grid * 16), dim3(64), lds, s, scl, org, dir, prim, t, u, v, occ, ao, fq, counters, (uint32_t)grid * 4u);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* the queue back to empty (stream-ordered), and the point the consumer's stream waits for */
int fixq_begin(const lh_fixq_t *q, hipStream_t s)
{
    hipLaunchKernelGGL(k_fixq_reset, dim3(1), dim3(256), 0, s, (unsigned long long *)q->queue, q->qcount, q->qcap);
    if (hipGetLastError() != hipSuccess) return -1;
    return hipEventRecord((hipEvent_t)q->ev_ready, s) == hipSuccess ? 0 : -1;
}

} /* namespace */

/* the AO stage of a tile with the rays generated inside the any-hit kernel (SRC 1 above): nslots primary hits,
 * N = ntheta * nphi rays each, occluded rays counted per slot in d_occ_count (zeroed here).  Rays the persistent kernel
 * does not finish -- fragile hits (the reference's own walk decides), rays out of visit budget or LDS rows (the cooperative
 * walk) -- go through d_queue (2 words per entry, count + overflow flag in d_qcount[0..1]) to k_coop_walk. */
extern "C" int lh_launch_trace_ao(const lh_dev_scene_t *sc, size_t nslots, int ntheta, int nphi, unsigned long long seed,
                                  const double *d_hitrec, const unsigned long long *d_slot_key, unsigned int *d_occ_count,
                                  unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks, int min_active,
                                  int tri_batch, const lh_fixq_t *q, int ncus, const unsigned long long *d_nslots, uint32_t budget_big, void *stream)
{
    /* d_nslots (or NULL): the slot count on the device -- nslots is then its upper bound (the samples of the batch), the buffers
     * are sized for it and the kernels read the count themselves: no host round trip between the compaction and this stage */
    hipStream_t s = (hipStream_t)stream;
    const size_t n = nslots * (size_t)(ntheta * nphi);
    if (n == 0) return 0;
    if (n >= ((size_t)1 << 31)) return -1;
    lh_dev_scene_t scl = *sc;
    bool guard = false;
    scl.stack_rows = rows4(*sc, &guard, n >= 65536);
    scl.stack_guard = guard ? 1 : 0;
    scl.top_nodes = top_nodes_for(*sc, scl.stack_rows);
    const size_t lds_bytes = (size_t)scl.stack_rows * LH_BLOCK * sizeof(int) + (size_t)scl.top_nodes * 64u;
    static uint32_t ao_chunk = 0;
    if (!ao_chunk) { const char *e = getenv("LH_AO_CHUNK"); ao_chunk = (e && atoi(e) >= 64 && atoi(e) <= 65536) ? (uint32_t)atoi(e) : LH_TILE_CHUNK; }
    scl.ray_chunk = ao_chunk;                                               /* AO rays of a slot are coherent: longer ranges per wave (LH_AO_CHUNK) */
    clamp_chunk(scl, n, grid_blocks);
    AoSrc ao = {d_hitrec, d_slot_key, d_occ_count, seed, ntheta, nphi, (uint32_t)nslots, (int)sc->ao_group, d_nslots, budget_big};
    if (d_nslots && sc->ao_group) return -1;            /* the grouped order needs the exact count on the host */
    FixQ fq = {(unsigned long long *)q->queue, q->qcount, q->qcap, (uint32_t)grid_blocks * (LH_BLOCK / 64), q->qcount + 4};
    if (hipMemsetAsync(d_cursor, 0, sizeof(uint32_t) * LH_CURSOR_WORDS, s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_occ_count, 0, sizeof(unsigned int) * nslots, s) != hipSuccess) return -1;
    if (fixq_begin(q, s) != 0) return -1;
#define LH_AO_LAUNCH(CNT, W) LH_LAUNCH_PERSIST((k_trace_persist_lane<true, CNT, W, 1>), \
                           scl, (uint32_t)n, (const double *)NULL, (const double *)NULL, (uint32_t *)NULL, (double *)NULL, (double *)NULL, \
                           (double *)NULL, (uint8_t *)NULL, d_counters, (uint32_t *)d_cursor, min_active, tri_batch, ao, fq)
    if (guard) { if (d_counters) LH_AO_LAUNCH(true, 8); else LH_AO_LAUNCH(false, 8); }
    else { if (d_counters) LH_AO_LAUNCH(true, 3); else LH_AO_LAUNCH(false, 3); }
#undef LH_AO_LAUNCH
    if (hipGetLastError() != hipSuccess) return -1;
    return launch_coop<true, 1>(scl, NULL, NULL, NULL, NULL, NULL, NULL, NULL, ao, fq, q, d_counters, ncus, s);
}

/* the node formats a launch of `variant` reads on this scene (bit mask, LH_FMT_* in lh_internal.h): so that the commit
 * code can upload a format the first time a variant asks for it */
extern "C" int lh_trace_rows(const lh_dev_scene_t *sc)
{
    bool guard = false;
    return (int)rows4(*sc, &guard);
}

extern "C" int lh_trace_formats_needed(const lh_dev_scene_t *sc, int variant)
{
    (void)sc;
    return variant == LH_VARIANT_DIRECT ? 1 : 4;
}

/* one batch of rays through the hot path.  variant: LH_VARIANT_SPEC (the default: 4-wide nodes, or the 8-wide nodes when
 * sc->prefer_q8) or LH_VARIANT_DIRECT (the textbook walk over the 2-wide fp32 nodes; needs sc->nodes).  q: the launch's
 * fix-up queue with its second stream, private to the stream. */
extern "C" int lh_launch_trace(const lh_dev_scene_t *sc, size_t n, const double *d_org,
                               const double *d_dir, uint32_t *d_prim, double *d_t, double *d_u,
                               double *d_v, int anyhit, uint8_t *d_occluded,
                               unsigned long long *d_counters, unsigned long long *d_workq,
                               int variant, int grid_blocks, int min_active, int tri_batch,
                               const lh_fixq_t *q, int ncus, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    /* the persistent kernel indexes rays with 32 bits: a larger batch is a sequence of launches */
    const size_t kMaxLaunch = (size_t)1 << 30;
    if (n > kMaxLaunch) {
        for (size_t off = 0; off < n; off += kMaxLaunch) {
            const size_t m = (n - off < kMaxLaunch) ? n - off : kMaxLaunch;
            const int rc = lh_launch_trace(sc, m, d_org + 3 * off, d_dir + 3 * off, d_prim ? d_prim + off : NULL, d_t ? d_t + off : NULL,
                                           d_u ? d_u + off : NULL, d_v ? d_v + off : NULL, anyhit, d_occluded ? d_occluded + off : NULL,
                                           d_counters, d_workq, variant, grid_blocks, min_active, tri_batch, q, ncus, stream);
            if (rc != 0) return rc;
        }
        return 0;
    }
    lh_dev_scene_t scl = *sc;
    uint32_t need; int walk; bool guard = false;
    if ((sc->cam_src || sc->n_dev) && (variant == LH_VARIANT_DIRECT || anyhit || q == NULL)) return -1;      /* the path tracer's chain: default walk, closest hit */
    static const bool small_ok = !(getenv("LH_SMALL_BATCH") && atoi(getenv("LH_SMALL_BATCH")) == 0);      /* LH_SMALL_BATCH=0: rounds 1-3's path for a handful of rays (A/B) */
    if (((n <= LH_SMALL_BATCH && small_ok) || sc->diag_out) && variant != LH_VARIANT_DIRECT && !sc->cam_src && !sc->n_dev && (sc->diag_out || !sc->stack_cap) && sc->q4nodes) {      /* the sequential walk has a private stack: a capped LDS stack (tests) does not concern per-ray diagnostics */
        /* a handful of rays (the coalesced one-ray callers): one small launch, a wave per ray, no queue, no cursors.
         * Per-ray diagnostics (diag_out): the same walk for a batch of any size, a lane per ray */
        if (n <= LH_SMALL_BATCH) {
            if (anyhit) hipLaunchKernelGGL((k_trace_small<true, true>), dim3((unsigned)n), dim3(64), 0, s, scl, (uint32_t)n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occluded, d_counters);
            else hipLaunchKernelGGL((k_trace_small<false, true>), dim3((unsigned)n), dim3(64), 0, s, scl, (uint32_t)n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occluded, d_counters);
        } else {
            const unsigned blocks = (unsigned)((n + 63) / 64);
            if (anyhit) hipLaunchKernelGGL((k_trace_small<true, false>), dim3(blocks), dim3(64), 0, s, scl, (uint32_t)n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occluded, d_counters);
            else hipLaunchKernelGGL((k_trace_small<false, false>), dim3(blocks), dim3(64), 0, s, scl, (uint32_t)n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occluded, d_counters);
        }
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (sc->diag_out) return -1;                 /* per-ray diagnostics exist for the default walk's nodes only */
    if (variant == LH_VARIANT_DIRECT) {
        if (!sc->nodes) return -1;
        need = sc->max_depth + 2; walk = 0;          /* 2-wide: one push per level + the sentinel */
        need = (need + 1u) & ~1u;
        if (need < 16) need = 16;
    } else if (scl.prefer_q8 && sc->q8nodes) {
        /* the 8-wide walk pushes up to 7 per level and keeps a scratch row; beyond 48 rows (three workgroups per CU) the rare
         * ray that needs them is finished by the cooperative walk over the 4-wide nodes (always resident) */
        need = 7 * sc->q8_depth + 10; walk = 7;
        const uint32_t cap = (sc->stack_cap >= 16 && sc->stack_cap < 64) ? sc->stack_cap : 48;
        if (need > cap) { need = cap; guard = true; }
        need = (need + 1u) & ~1u;
        if (need < 16 && !guard) need = 16;
        scl.stack_guard = guard ? 1 : 0;
    } else {
        /* a launch that fills the chip walks at most LH_ROWS_CHECKED rows (four workgroups per CU), a ray that would overrun
         * them is finished by the cooperative walk; so does a very deep tree (chains of nested geometry, an LBVH over a
         * degenerate distribution) whose worst case does not fit 64 rows -- rows4 */
        scl.prefer_q8 = 0;
        need = rows4(*sc, &guard, n >= 65536 || sc->n_dev != NULL); walk = guard ? 8 : 3;
        scl.stack_guard = guard ? 1 : 0;
    }
    if (need > LH_ROWS_UNCHECKED) return -1;
    scl.stack_rows = need;
    scl.top_nodes = (walk == 0 || walk == 7) ? 0u : top_nodes_for(*sc, need);
    const size_t lds_bytes = (size_t)need * LH_BLOCK * sizeof(int) + (size_t)scl.top_nodes * 64u;
    clamp_chunk(scl, n, grid_blocks);
    /* small batches (one synchronous ray, a bucket of lucille's renderer): no visit budget, no second stream -- a ray the walk
     * cannot finish (LDS rows) stays flagged for k_fixups */
    const bool coop = walk != 0 && q != NULL && (n >= 65536 || sc->n_dev != NULL);
    FixQ fq = {coop ? (unsigned long long *)q->queue : NULL, q ? q->qcount : NULL, coop ? q->qcap : 0u, (uint32_t)grid_blocks * (LH_BLOCK / 64), q ? q->qcount + 4 : NULL};
    if (!coop) scl.ray_budget = 0xFFFFFFFFu;
    if (walk != 0 && q != NULL && fixq_begin(q, s) != 0) return -1;
    if (walk != 0 && q == NULL) return -1;
    int rc = launch_walk(scl, n, d_org, d_dir, d_prim, d_t, d_u, d_v, anyhit, d_occluded,
                         d_counters, d_workq, walk, grid_blocks, min_active, tri_batch, lds_bytes, fq, s);
    if (rc != 0) return rc;
    if (coop) {
        rc = anyhit ? launch_coop<true, 0>(scl, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occluded, AoSrc{}, fq, q, d_counters, ncus, s)
             : sc->cam_src ? launch_coop<false, 2>(scl, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occluded, AoSrc{}, fq, q, d_counters, ncus, s)
                    : launch_coop<false, 0>(scl, d_org, d_dir, d_prim, d_t, d_u, d_v, d_occluded, AoSrc{}, fq, q, d_counters, ncus, s);
        if (rc != 0) return rc;
    }
    /* what is still flagged: fragile hits (the reference's own walk), and out-of-budget rays the queue had no room for */
    {
        const size_t blocks = (n + 255) / 256;
        hipLaunchKernelGGL(k_fixups, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, scl, n, d_org, d_dir,
                           d_prim, d_t, d_u, d_v, d_occluded, anyhit, d_counters, q ? q->qcount : NULL, (walk == 0 || q == NULL) ? 1 : 0);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
