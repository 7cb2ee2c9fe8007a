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
 *    decides eight children); the 2-wide fp32 / 16-bit walks and the compressed
 *    8-wide walk are kept for in-process A/B (variants 0-3, 5, LH_NODE_FORMAT);
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
 *  - persistent variants pull work from a global cursor: whole 64-ray chunks
 *    (PERSIST_WAVE) or, ballot/popcount-compacted, just enough rays to refill
 *    the lanes that have finished (PERSIST_LANE).
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
#include "lh_pt.h"

/* fetch one inner node and test both child boxes: fp32 nodes (4 x dwordx4) or 16-bit grid
 * nodes (2 x dwordx4) */
template <bool QN>
__device__ __forceinline__ void node_test(const lh_dev_scene_t &sc, const lh_ray32_t &r, float tb, int cur,
                                          bool &h0, bool &h1, float &tn0, float &tn1, int &r0, int &r1)
{
    if (QN) {
        const uint4 *p = (const uint4 *)sc.qnodes + 2 * (size_t)cur;
        const uint4 a = p[0], b = p[1];
        h0 = lh_slab_q(&r, (float)(a.x & 0xffffu), (float)(a.x >> 16), (float)(a.y & 0xffffu),
                       (float)(a.y >> 16), (float)(a.z & 0xffffu), (float)(a.z >> 16), tb, &tn0);
        h1 = lh_slab_q(&r, (float)(a.w & 0xffffu), (float)(a.w >> 16), (float)(b.x & 0xffffu),
                       (float)(b.x >> 16), (float)(b.y & 0xffffu), (float)(b.y >> 16), tb, &tn1);
        r0 = (int)b.z; r1 = (int)b.w;
    } else {
        const float4 *p = (const float4 *)sc.nodes + 4 * (size_t)cur;
        const float4 n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3];
        /* child0: lo (n0.x n0.y n0.z) hi (n0.w n1.x n1.y); child1: lo (n1.z n1.w n2.x) hi (n2.y n2.z n2.w) */
        h0 = lh_slab(&r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, tb, &tn0);
        h1 = lh_slab(&r, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, tb, &tn1);
        r0 = __float_as_int(n3.x); r1 = __float_as_int(n3.y);
    }
}

/* the per-lane traversal body; runs while the lane has work, leaves when
 * `stop()` says the wave should regroup.  Returns with L.cur == kDone when the
 * ray is finished. */
template <bool ANYHIT, bool COUNT, bool BURST, bool QN>
__device__ __forceinline__ void traverse(Lane &L, const lh_dev_scene_t &sc,
                                         int (*stk)[LH_BLOCK], const int tid,
                                         double ox, double oy, double oz,
                                         double dx, double dy, double dz, Best &best,
                                         uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact,
                                         const int min_active)
{
    const float4 *__restrict__ tris  = (const float4 *)sc.tri32;

    while (L.cur != kDone) {
        /* ---- inner nodes ------------------------------------------------ */
        while (L.cur >= 0) {
            if (COUNT) c_nodes++;
            float tn0, tn1; bool h0, h1; int r0, r1;
            node_test<QN>(sc, L.r, L.tb, L.cur, h0, h1, tn0, tn1, r0, r1);
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

        if (BURST) {
            /* wave regroup point: leave when too few lanes are still walking */
            if (__popcll(__ballot(1)) < min_active) break;
        }
    }
}

/* Single-loop ("if-if") walk: every iteration every active lane consumes exactly ONE
 * record -- an inner node (64 B) or one leaf triangle (48 B) -- fetched with the same
 * four dwordx4 loads from a per-lane pointer, so lanes that reach a leaf do not idle
 * while their neighbours are still descending (the while-while walk above measured
 * 21 % VALU lane utilisation on incoherent rays: rocprofv3 SQ_THREAD_CYCLES_VALU /
 * (SQ_ACTIVE_INST_VALU*64), profiles/r01_pmc_diag.md).  A leaf with k triangles is k
 * iterations: the leaf reference carries (first, count-1) and is advanced in place. */
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ void traverse_unified(Lane &L, const lh_dev_scene_t &sc,
                                                 int (*stk)[LH_BLOCK], const int tid,
                                                 double ox, double oy, double oz,
                                                 double dx, double dy, double dz, Best &best,
                                                 uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact,
                                                 const int min_active)
{
    const float4 *__restrict__ nodes = (const float4 *)sc.nodes;
    const float4 *__restrict__ tris  = (const float4 *)sc.tri32;

    while (L.cur != kDone) {
        const bool is_node = L.cur >= 0;
        const uint32_t x = ~(uint32_t)L.cur;
        const float4 *p = is_node ? nodes + 4 * (size_t)L.cur : tris + 3 * (size_t)(x >> 2);
        const float4 n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3];   /* tri32 is padded by 16 B */
        if (is_node) {
            if (COUNT) c_nodes++;
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
        } else {
            if (COUNT) c_tris++;
            const bool finished = tri_step<ANYHIT, COUNT>(L, sc, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.z, n2.w, __float_as_uint(n2.y), ox, oy, oz, dx, dy, dz, best, c_exact);
            if (finished) L.cur = kDone;
            else if (x & 3u) L.cur = (int)~(((x >> 2) + 1u) << 2 | ((x & 3u) - 1u));   /* next triangle of this leaf */
            else { L.sp--; L.cur = stk[L.sp][tid]; }
        }
        /* wave regroup point: leave when too few lanes are still walking */
        if (__popcll(__ballot(L.cur != kDone)) < min_active) break;
    }
}

/* Speculative walk with one postponed leaf per lane (variant 4).
 *
 * The unified walk still executes the ~75-instruction triangle path on every iteration
 * for the ~6 % of lanes that hold a leaf (rocprofv3: SIMDs ~90 % issue-busy, VALU lane
 * utilisation 29 %).  Here a lane that reaches a leaf parks it in `pend` and keeps
 * descending from its stack; the triangle path runs only when at least `tri_batch` lanes
 * hold a parked leaf (or nobody has an inner node left), one triangle per parked lane per
 * pass.  A lane that meets a second leaf while one is parked waits for the next pass.
 * The node step is branch-free: unconditional LDS push (slot sp is free space), pop read
 * of slot sp-1 (slot 0 holds the sentinel), selects for everything else. */
template <bool ANYHIT, bool COUNT, bool QN>
__device__ __forceinline__ void traverse_spec(Lane &L, int &pend, const lh_dev_scene_t &sc,
                                              int (*stk)[LH_BLOCK], const int tid,
                                              double ox, double oy, double oz,
                                              double dx, double dy, double dz, Best &best,
                                              uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact,
                                              const int min_active, const int tri_batch)
{
    const float4 *__restrict__ tris  = (const float4 *)sc.tri32;
    constexpr int kNoLeaf = 0;   /* never a valid leaf reference (leaf refs are negative) */

    for (;;) {
        /* ---- node step for every lane that holds an inner node ------------------ */
        if (L.cur >= 0) {
            if (COUNT) c_nodes++;
            float tn0, tn1; bool h0, h1; int r0, r1;
            node_test<QN>(sc, L.r, L.tb, L.cur, h0, h1, tn0, tn1, r0, r1);
            const bool any = h0 | h1, both = h0 & h1;
            const bool second = h1 && (!h0 || tn1 < tn0);
            stk[L.sp][tid] = second ? r0 : r1;                 /* far child; kept only if `both` */
            L.sp += both ? 1 : 0;
            const int popped = stk[L.sp - 1][tid];
            int nxt = any ? (second ? r1 : r0) : popped;
            L.sp -= any ? 0 : 1;
            /* park the leaf and keep walking if the parking slot is free */
            const bool is_leaf = (nxt < 0) & (nxt != kDone);
            const bool park = is_leaf & (pend == kNoLeaf);
            pend = park ? nxt : pend;
            const int popped2 = stk[L.sp - 1][tid];
            L.cur = park ? popped2 : nxt;
            L.sp -= park ? 1 : 0;
        }
        /* ---- triangle pass when enough leaves are parked ------------------------- */
        const unsigned long long m_node = __ballot(L.cur >= 0);
        const unsigned long long m_pend = __ballot(pend != kNoLeaf);
        if (m_pend != 0ull && (__popcll(m_pend) >= tri_batch || m_node == 0ull)) {
            if (pend != kNoLeaf) {
                const uint32_t x = ~(uint32_t)pend;
                const float4 *tp = tris + 3 * (size_t)(x >> 2);
                const float4 ta = tp[0], tb_ = tp[1], tc = tp[2];
                if (COUNT) c_tris++;
                const bool finished = tri_step<ANYHIT, COUNT>(L, sc, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w, __float_as_uint(tc.y), ox, oy, oz, dx, dy, dz, best, c_exact);
                if (finished) { L.cur = kDone; pend = kNoLeaf; }
                else if (x & 3u) pend = (int)~(((x >> 2) + 1u) << 2 | ((x & 3u) - 1u));   /* next triangle */
                else {
                    /* leaf finished: a lane that was waiting with a second leaf parks it now */
                    const bool waiting = (L.cur < 0) & (L.cur != kDone);
                    pend = waiting ? L.cur : kNoLeaf;
                    if (waiting) { L.sp--; L.cur = stk[L.sp][tid]; }
                }
            }
        }
        /* ---- regroup when too few lanes still have work --------------------------- */
        const unsigned long long m_work = __ballot((L.cur != kDone) | (pend != kNoLeaf));
        if (__popcll(m_work) < min_active) break;
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

/* Speculative walk over the 8-wide compressed nodes (lh_c8node_t, use_qnodes == 3): 80-byte records,
 * five dwordx4 loads, eight children per visit in octant order -- no distance sort: child s has
 * priority s ^ oct (0 = nearest) and hit children are written to the stack by rank among the hits
 * (popcount of nearer hits), nearest on top.  Leaves are parked and tested in batches exactly as in
 * traverse_spec4; their triangles come from tri32_c8.  If the stack would overflow (rows are capped
 * so that three workgroups fit a CU) the ray is handed to the reference walk (write_out). */
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ void traverse_c8(Lane &L, int &pend, const lh_dev_scene_t &sc,
                                            int (*stk)[LH_BLOCK], const int tid,
                                            double ox, double oy, double oz,
                                            double dx, double dy, double dz, Best &best,
                                            uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact,
                                            const int min_active, const int tri_batch)
{
    const float4 *__restrict__ tris = (const float4 *)sc.tri32_c8;
    constexpr int kNoLeaf = 0;
    const uint32_t oct = (uint32_t)L.r.ngx | ((uint32_t)L.r.ngy << 1) | ((uint32_t)L.r.ngz << 2);
    const int rows = (int)sc.stack_rows;

    for (;;) {
        if (L.cur >= 0) {
            const uint4 *p = (const uint4 *)((const char *)sc.c8nodes + (size_t)sc.c8_stride * (size_t)L.cur);
            const uint4 n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3], n4 = p[4];
            if (COUNT) c_nodes++;
            if (L.sp + 9 > rows) { L.over = true; L.cur = kDone; pend = kNoLeaf; }
            else {
                lh_c8frame_t f;
                lh_c8_frame(&L.r, __uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z),
                            n0.w & 255u, (n0.w >> 8) & 255u, (n0.w >> 16) & 255u, &f);
                const uint32_t imask = n0.w >> 24;
                /* near / far plane bytes for the ray's direction signs: 4 children per dword */
                const uint32_t nxA = L.r.ngx ? n3.z : n2.x, nxB = L.r.ngx ? n3.w : n2.y, fxA = L.r.ngx ? n2.x : n3.z, fxB = L.r.ngx ? n2.y : n3.w;
                const uint32_t nyA = L.r.ngy ? n4.x : n2.z, nyB = L.r.ngy ? n4.y : n2.w, fyA = L.r.ngy ? n2.z : n4.x, fyB = L.r.ngy ? n2.w : n4.y;
                const uint32_t nzA = L.r.ngz ? n4.z : n3.x, nzB = L.r.ngz ? n4.w : n3.y, fzA = L.r.ngz ? n3.x : n4.z, fzB = L.r.ngz ? n3.y : n4.w;
                uint32_t hp = 0u;                 /* hits, bit position = priority (0 nearest) */
                int ref[8];
#define LH_C8_CHILD(S, NX, NY, NZ, FX, FY, FZ, META) { \
                    const uint32_t m_ = ((META) >> (8 * ((S) & 3))) & 255u; \
                    const bool inner_ = (imask >> (S)) & 1u; \
                    float tn_; \
                    const bool hit_ = lh_slab_c8(&f, (float)(((NX) >> (8 * ((S) & 3))) & 255u), (float)(((NY) >> (8 * ((S) & 3))) & 255u), \
                                                 (float)(((NZ) >> (8 * ((S) & 3))) & 255u), (float)(((FX) >> (8 * ((S) & 3))) & 255u), \
                                                 (float)(((FY) >> (8 * ((S) & 3))) & 255u), (float)(((FZ) >> (8 * ((S) & 3))) & 255u), L.tb, &tn_) \
                                      & (inner_ | (m_ != 0u)); \
                    hp |= (hit_ ? 1u : 0u) << ((uint32_t)(S) ^ oct); \
                    ref[S] = inner_ ? (int)(n1.x + (uint32_t)__popc(imask & ((1u << (S)) - 1u))) \
                                    : (int)~(((n1.y + (m_ & 31u)) << 2) | ((m_ >> 5) & 3u)); }
                LH_C8_CHILD(0, nxA, nyA, nzA, fxA, fyA, fzA, n1.z)
                LH_C8_CHILD(1, nxA, nyA, nzA, fxA, fyA, fzA, n1.z)
                LH_C8_CHILD(2, nxA, nyA, nzA, fxA, fyA, fzA, n1.z)
                LH_C8_CHILD(3, nxA, nyA, nzA, fxA, fyA, fzA, n1.z)
                LH_C8_CHILD(4, nxB, nyB, nzB, fxB, fyB, fzB, n1.w)
                LH_C8_CHILD(5, nxB, nyB, nzB, fxB, fyB, fzB, n1.w)
                LH_C8_CHILD(6, nxB, nyB, nzB, fxB, fyB, fzB, n1.w)
                LH_C8_CHILD(7, nxB, nyB, nzB, fxB, fyB, fzB, n1.w)
#undef LH_C8_CHILD
                const int nh = __popc(hp);
                const int base = L.sp + nh - 1;
#pragma unroll
                for (int sl = 0; sl < 8; sl++) {
                    const uint32_t pr = (uint32_t)sl ^ oct;
                    if ((hp >> pr) & 1u) stk[base - __popc(hp & ((1u << pr) - 1u))][tid] = ref[sl];
                }
                L.sp = base;
                const int nxt = stk[base][tid];
                const bool is_leaf = (nxt < 0) & (nxt != kDone);
                const bool park = is_leaf & (pend == kNoLeaf);
                pend = park ? nxt : pend;
                const int popped2 = stk[L.sp - 1][tid];
                L.cur = park ? popped2 : nxt;
                L.sp -= park ? 1 : 0;
            }
        }
        const unsigned long long m_node = __ballot(L.cur >= 0);
        const unsigned long long m_pend = __ballot(pend != kNoLeaf);
        if (m_pend != 0ull && (__popcll(m_pend) >= tri_batch || m_node == 0ull)) {
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

/* Single-loop walk over the 4-wide nodes (variant 5): every iteration every active lane consumes ONE
 * record -- a 4-wide node (64 B) or one leaf triangle (48 B, padded) -- fetched by the same four
 * dwordx4 loads, so a wave pays one memory round trip per iteration for both kinds of work.  The node
 * step is traverse_spec4's (rank-derived stack writes); nothing is parked. */
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ void traverse_unified4(Lane &L, const lh_dev_scene_t &sc,
                                                  int (*stk)[LH_BLOCK], const int tid,
                                                  double ox, double oy, double oz,
                                                  double dx, double dy, double dz, Best &best,
                                                  uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact,
                                                  const int min_active)
{
    const uint4 *__restrict__ nodes = (const uint4 *)sc.q4nodes;
    const uint4 *__restrict__ tris  = (const uint4 *)sc.tri32;

    while (L.cur != kDone) {
        const bool is_node = L.cur >= 0;
        const uint32_t x = ~(uint32_t)L.cur;
        const uint4 *p = is_node ? nodes + 4 * (size_t)L.cur : tris + 3 * (size_t)(x >> 2);
        const uint4 a = p[0], b = p[1], c = p[2], r = p[3];          /* tri32 is padded by 16 B */
        if (is_node) {
            if (COUNT) c_nodes++;
            float t0, t1, t2, t3;
            const bool h0 = slab_w(L, a.x, a.y, a.z, t0) & ((int)r.x != kDone);
            const bool h1 = slab_w(L, a.w, b.x, b.y, t1) & ((int)r.y != kDone);
            const bool h2 = slab_w(L, b.z, b.w, c.x, t2) & ((int)r.z != kDone);
            const bool h3 = slab_w(L, c.y, c.z, c.w, t3) & ((int)r.w != kDone);
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
            L.cur = stk[base][tid];
        } else {
            if (COUNT) c_tris++;
            const bool finished = tri_step<ANYHIT, COUNT>(L, sc, __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x), __uint_as_float(c.z), __uint_as_float(c.w), c.y, ox, oy, oz, dx, dy, dz, best, c_exact);
            if (finished) L.cur = kDone;
            else if (x & 3u) L.cur = (int)~(((x >> 2) + 1u) << 2 | ((x & 3u) - 1u));
            else { L.sp--; L.cur = stk[L.sp][tid]; }
        }
        if (__popcll(__ballot(L.cur != kDone)) < min_active) break;
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
/* variant 0: one ray per lane                                              */
/* ------------------------------------------------------------------------ */
template <bool ANYHIT, bool COUNT, bool QN>
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
    traverse<ANYHIT, COUNT, false, QN>(L, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, 0);
    finish<ANYHIT, COUNT>(L, sc, ox, oy, oz, dx, dy, dz, best, ce);
    write_out<ANYHIT>(i, L, best, prim, t, u, v, occ, sc.ref_nodes != NULL);
    if (COUNT) add_counters(counters, cn, ct, ce, 1);
}

/* ------------------------------------------------------------------------ */
/* variant 1: persistent wavefronts, 64-ray chunks from a global cursor     */
/* ------------------------------------------------------------------------ */
template <bool ANYHIT, bool COUNT, bool QN>
__global__ __launch_bounds__(LH_BLOCK) void k_trace_persist_wave(
    lh_dev_scene_t sc, size_t n, const double *__restrict__ org, const double *__restrict__ dir,
    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
    double *__restrict__ v, uint8_t *__restrict__ occ, unsigned long long *counters,
    unsigned long long *cursor)
{
    extern __shared__ int lh_stack_lds[];          /* [stack entries][LH_BLOCK], sized at launch */
    int (*stk)[LH_BLOCK] = (int (*)[LH_BLOCK])lh_stack_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t cn = 0, ct = 0, ce = 0, cr = 0;
    for (;;) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(cursor, 64ull);
        base = __shfl(base, 0);
        if (base >= n) break;
        const size_t i = base + lane;
        if (i < n) {
            const double ox = org[3 * i], oy = org[3 * i + 1], oz = org[3 * i + 2];
            const double dx = dir[3 * i], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
            Lane L; Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
            lane_init(L, sc, ox, oy, oz, dx, dy, dz);
            stk[0][tid] = kDone;
            traverse<ANYHIT, COUNT, false, QN>(L, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, 0);
            finish<ANYHIT, COUNT>(L, sc, ox, oy, oz, dx, dy, dz, best, ce);
            write_out<ANYHIT>(i, L, best, prim, t, u, v, occ, sc.ref_nodes != NULL);
            if (COUNT) cr++;
        }
    }
    if (COUNT) add_counters(counters, cn, ct, ce, cr);
}

/* ------------------------------------------------------------------------ */
/* variant 2: persistent wavefronts with ballot-compacted lane refill       */
/* ------------------------------------------------------------------------ */
/* SRC 0: rays from the fp64 org/dir arrays.  SRC 1 (any-hit, WALK 3): the work items are the ambient-occlusion
 * rays of a tile -- item i = (hit slot i / N, sample i % N) -- generated in the refill by lh_ao.h from the hit
 * record; an occluded ray adds one to its slot's counter, nothing per ray goes through HBM (round 1 wrote and
 * re-read 49 bytes per AO ray: 22 GB per 4096^2 x 64 frame).  A fragile hit (lh_reftrace.h; rare) is queued for
 * the reference's own walk. */
struct AoSrc {
    const double *hitrec; const unsigned long long *slot_key; unsigned int *occ_count;
    unsigned long long seed; int ntheta, nphi;
    uint32_t *queue, *qcount; uint32_t qcap;
};

/* the reference's own walk for one ray, out of line: its private stack stays out of the persistent kernel's frame */
struct RefHit { double t, u, v; uint32_t prim; };
__device__ __noinline__ RefHit ref_trace_one(const lh_dev_scene_t &sc, double ox, double oy, double oz, double dx, double dy, double dz)
{
    RefHit h; uint32_t p = LH_MISS_PRIM; double tt = LH_T_INF, uu = 0.0, vv = 0.0;
    (void)lh_ref_trace((const lh_refnode_t *)sc.ref_nodes, (const uint32_t *)sc.ref_leaf_prims, (const double *)sc.tri64,
                       sc.ref_empty, sc.ref_bmin, sc.ref_bmax, ox, oy, oz, dx, dy, dz, &p, &tt, &uu, &vv);
    h.prim = p; h.t = tt; h.u = uu; h.v = vv;
    return h;
}

/* ray source 2: the paths of one path-tracing pass (lh_pt.h).  A lane takes path i, generates its camera ray, and every time
 * its ray is finished it shades the hit and goes on with the path's next ray -- until the path leaves the scene (radiance =
 * throughput x environment), loses the roulette or reaches the vertex limit.  Only radiance[path] goes through HBM. */
struct PtSrc {
    DevCamera cam; int x0, y0, w, spp, s0, full_width, max_depth, use_override, ref_weights;
    unsigned long long seed;
    const double *nrm9, *col9; const uint32_t *prim_mesh; const DevMaterial *materials; DevMaterial override_mat; DevEnv env;
    float *radiance;                 /* [n][3], zeroed by the caller */
    unsigned long long *nrays;       /* += rays traced */
    unsigned int *maxdepth;          /* max= deepest vertex index reached */
};

/* one path vertex: the hit record of the finished ray -> radiance written (miss), path ended (roulette / vertex limit), or the
 * next ray.  Inlined: out of line (to keep its fp64 temporaries out of the walk's register allocation) it was slower at every
 * occupancy -- 228 / 234 / 291 ms at 2 / 3 / 4 waves per SIMD against 192 ms inline at 2 (profiles/README.md) */
struct PtStep { double ox, oy, oz, dx, dy, dz; float g0, g1, g2; uint32_t pword; int go; };
__device__ __forceinline__ PtStep pt_vertex(const lh_dev_scene_t &sc, const PtSrc &pt, uint32_t hp, double ht, double hu, double hv,
                                         double ox, double oy, double oz, double dx, double dy, double dz,
                                         float g0, float g1, float g2, uint32_t pword, int pdepth)
{
    PtStep o;
    const uint32_t path = pword & ~LH_PT_INTERIOR;
    o.ox = ox; o.oy = oy; o.oz = oz; o.dx = dx; o.dy = dy; o.dz = dz; o.g0 = g0; o.g1 = g1; o.g2 = g2; o.pword = pword; o.go = 0;
    if (hp == LH_MISS_PRIM) {
        float e[3];
        env_fetch(pt.env, dx, dy, dz, e);
        float *rad = pt.radiance + 3 * (size_t)path;
        rad[0] = g0 * e[0]; rad[1] = g1 * e[1]; rad[2] = g2 * e[2];
        return o;
    }
    const DevMaterial M = pt.use_override ? pt.override_mat : pt.materials[pt.prim_mesh[hp]];
    const uint64_t key = pt_key(pt.seed, path, pt.spp, pt.s0, pt.x0, pt.y0, pt.w, pt.full_width, pdepth);
    if (!pt_survives(M, key, pdepth, pt.max_depth)) return o;
    {
        const double Or[3] = {ox, oy, oz}, D[3] = {dx, dy, dz}; const float G[3] = {g0, g1, g2};
        double o2[3], O[3]; float G2[3]; uint32_t pw2;
        pt_scatter(sc, pt.nrm9, pt.col9, M, pt.ref_weights, key, hp, pword, Or, D, ht, hu, hv, G, o2, O, G2, pw2);
        o.ox = o2[0]; o.oy = o2[1]; o.oz = o2[2]; o.dx = O[0]; o.dy = O[1]; o.dz = O[2];
        o.g0 = G2[0]; o.g1 = G2[1]; o.g2 = G2[2]; o.pword = pw2; o.go = 1;
    }
    return o;
}

template <bool ANYHIT, bool COUNT, int WALK, bool QN, int SRC>
__device__ __forceinline__ void trace_persist_lane(
    const lh_dev_scene_t &sc, size_t n, const double *__restrict__ org, const double *__restrict__ dir,
    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
    double *__restrict__ v, uint8_t *__restrict__ occ, unsigned long long *counters,
    unsigned long long *cursor, int min_active, int tri_batch, const AoSrc &ao, const PtSrc &pt, int *lds)
{
    int (*stk)[LH_BLOCK] = (int (*)[LH_BLOCK])lds;
    const int tid = threadIdx.x;
    float g0 = 1.0f, g1 = 1.0f, g2 = 1.0f;      /* SRC 2: path throughput */
    int pdepth = 0; uint32_t pword = 0u, lrays = 0u, ldeep = 0u;
    uint32_t cn = 0, ct = 0, ce = 0, cr = 0, cns = 0, cts = 0, crs = 0;
    Lane L; Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
    int pend = 0;                    /* WALK 2: parked leaf reference (0 = none) */
    uint32_t selfp = LH_MISS_PRIM;   /* SRC 1: the triangle this AO ray starts on, when it cannot occlude the ray (lh_ao.h) */
    size_t my = (size_t)-1;          /* ray this lane is working on */
    double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1;
    L.cur = kDone; L.sp = 1; L.np = 0; L.certain = false; L.over = false;
    bool exhausted = false;          /* wave-uniform: cursor ran past n */
    unsigned long long wbase = 0, wend = 0;   /* wave-uniform: this wave's reserved ray range */

    for (;;) {
        /* ---- regroup: retire finished lanes, refill them ----------------- */
        bool idle = (L.cur == kDone) && (pend == 0);
        if (COUNT) crs++;
        if (idle) {
            if (my != (size_t)-1) {
                finish<ANYHIT, COUNT>(L, sc, ox, oy, oz, dx, dy, dz, best, ce, SRC == 1 ? selfp : LH_MISS_PRIM);
                if (SRC == 0) write_out<ANYHIT>(my, L, best, prim, t, u, v, occ, sc.ref_nodes != NULL, WALK == 8 || WALK == 7);
                else if (SRC == 1) {
                    const bool hit = L.certain || best.prim != LH_MISS_PRIM;
                    const bool retrace = sc.ref_nodes != NULL && (L.over || (best.prim != LH_MISS_PRIM && best.frag != 0u && !L.certain));
                    if (retrace) {
                        const uint32_t k = atomicAdd(ao.qcount, 1u);
                        if (k < ao.qcap) { ao.queue[6 * (size_t)k] = (uint32_t)my; ao.queue[6 * (size_t)k + 1] = 5u; }   /* 5: the reference walk decides */
                        else atomicOr(ao.qcount + 1, 1u);
                    } else if (hit) atomicAdd(&ao.occ_count[(uint32_t)my / (uint32_t)(ao.ntheta * ao.nphi)], 1u);
                } else {
                    /* the path's vertex: the hit record as write_out + k_ref_retrace would leave it, then lh_pt.h */
                    uint32_t hp = best.prim; double ht = best.t, hu = best.u, hv = best.v;
                    if (sc.ref_nodes != NULL && (L.over || (best.prim != LH_MISS_PRIM && best.frag != 0u))) {
                        const RefHit rh = ref_trace_one(sc, ox, oy, oz, dx, dy, dz);
                        hp = rh.prim; ht = rh.t; hu = rh.u; hv = rh.v;
                    }
                    lrays++;
                    const PtStep st = pt_vertex(sc, pt, hp, ht, hu, hv, ox, oy, oz, dx, dy, dz, g0, g1, g2, pword, pdepth);
                    ox = st.ox; oy = st.oy; oz = st.oz; dx = st.dx; dy = st.dy; dz = st.dz;
                    g0 = st.g0; g1 = st.g1; g2 = st.g2; pword = st.pword;
                    if (st.go) {
                        lane_init(L, sc, ox, oy, oz, dx, dy, dz);
                        best.t = LH_T_INF; best.u = 0.0; best.v = 0.0; best.prim = LH_MISS_PRIM; best.frag = 0u;
                        stk[0][tid] = kDone;
                        pdepth++;
                        if ((uint32_t)pdepth > ldeep) ldeep = (uint32_t)pdepth;
                        idle = false;                             /* the lane goes on with the same path */
                    }
                }
                if (COUNT) cr++;
                if (idle) my = (size_t)-1;
            }
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
                } else if (SRC == 2) {
                    double po[3], pd[3];
                    pt_primary_ray(pt.cam, pt.x0, pt.y0, pt.w, pt.spp, pt.s0, pt.seed, i, po, pd);
                    ox = po[0]; oy = po[1]; oz = po[2]; dx = pd[0]; dy = pd[1]; dz = pd[2];
                    g0 = g1 = g2 = 1.0f; pdepth = 0; pword = (uint32_t)i;
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
        if (WALK == 5) {
            traverse_c8<ANYHIT, COUNT>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch);
        } else if (WALK == 4) {
            if (L.cur != kDone) traverse_unified4<ANYHIT, COUNT>(L, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh);
        } else if (WALK == 3) {
            traverse_spec4<ANYHIT, COUNT, false>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch, cns, cts);
        } else if (WALK == 8) {          /* the same with the stack check: trees whose worst case the LDS rows do not cover */
            traverse_spec4<ANYHIT, COUNT, true>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch, cns, cts);
        } else if (WALK == 7) {
            traverse_spec8<ANYHIT, COUNT>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch, cns, cts);
        } else if (WALK == 2) {
            /* every lane enters (idle lanes just vote in the ballots) */
            traverse_spec<ANYHIT, COUNT, QN>(L, pend, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh, tri_batch);
        } else if (L.cur != kDone) {
            if (WALK == 1) traverse_unified<ANYHIT, COUNT>(L, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh);
            else traverse<ANYHIT, COUNT, true, QN>(L, sc, stk, tid, ox, oy, oz, dx, dy, dz, best, cn, ct, ce, thresh);
        }
    }
    if (COUNT) {
        add_counters(counters, cn, ct, ce, cr);
        atomicAdd(&counters[LH_CNT_NODE_SLOTS], (unsigned long long)cns);
        atomicAdd(&counters[LH_CNT_TRI_SLOTS], (unsigned long long)cts);
        atomicAdd(&counters[LH_CNT_REGROUP_SLOTS], (unsigned long long)crs);
    }
    if (SRC == 2) {
        unsigned long long r = lrays; uint32_t dmax = ldeep;
        for (int off = 32; off >= 1; off >>= 1) { r += __shfl_down(r, off); const uint32_t o = __shfl_down(dmax, off); dmax = o > dmax ? o : dmax; }
        if ((tid & 63) == 0) { atomicAdd(pt.nrays, r); atomicMax(pt.maxdepth, dmax); }
    }
}

template <bool ANYHIT, bool COUNT, int WALK, bool QN, int SRC>
__global__ __launch_bounds__(LH_BLOCK) void k_trace_persist_lane(
    lh_dev_scene_t sc, size_t n, const double *__restrict__ org, const double *__restrict__ dir,
    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u,
    double *__restrict__ v, uint8_t *__restrict__ occ, unsigned long long *counters,
    unsigned long long *cursor, int min_active, int tri_batch, const AoSrc ao)
{
    extern __shared__ int lh_stack_lds[];          /* [stack entries][LH_BLOCK], sized at launch */
    trace_persist_lane<ANYHIT, COUNT, WALK, QN, SRC>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, ao, PtSrc{}, lh_stack_lds);
}

/* one path-tracing pass, fused: n paths, camera ray to last vertex inside the walk (ray source 2) */
template <bool COUNT, int WALK>
__global__ __launch_bounds__(LH_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_trace_pt(lh_dev_scene_t sc, size_t n, unsigned long long *counters,
                                                       unsigned long long *cursor, int min_active, int tri_batch, const PtSrc pt)
{
    extern __shared__ int lh_stack_lds[];
    trace_persist_lane<false, COUNT, WALK, true, 2>(sc, n, (const double *)NULL, (const double *)NULL, (uint32_t *)NULL, (double *)NULL, (double *)NULL,
                                                    (double *)NULL, (uint8_t *)NULL, counters, cursor, min_active, tri_batch, AoSrc{}, pt, lh_stack_lds);
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

template <bool ANYHIT, bool COUNT, bool QN>
int launch_one(const lh_dev_scene_t &sc, size_t n, const double *org, const double *dir,
               uint32_t *prim, double *t, double *u, double *v, uint8_t *occ,
               unsigned long long *counters, unsigned long long *cursor, int variant,
               int grid_blocks, int min_active, int tri_batch, size_t lds_bytes, hipStream_t s)
{
    if (variant == LH_VARIANT_DIRECT) {
        const size_t blocks = (n + LH_BLOCK - 1) / LH_BLOCK;
        if (blocks > 0x7fffffffull) return -1;
        hipLaunchKernelGGL((k_trace_direct<ANYHIT, COUNT, QN>), dim3((unsigned)blocks), dim3(LH_BLOCK), lds_bytes, s,
                           sc, n, org, dir, prim, t, u, v, occ, counters);
    } else {
        if (hipMemsetAsync(cursor, 0, sizeof(unsigned long long), s) != hipSuccess) return -1;
        if (variant == LH_VARIANT_PERSIST_WAVE)
            hipLaunchKernelGGL((k_trace_persist_wave<ANYHIT, COUNT, QN>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor);
        else if (variant == LH_VARIANT_PERSIST_LANE)
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 0, QN, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
        else if (variant == LH_VARIANT_UNIFIED)
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 1, false, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
        else if (variant == LH_VARIANT_SPEC && sc.use_qnodes == 3)
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 5, true, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
        else if (variant == LH_VARIANT_UNIFIED4 && sc.use_qnodes == 2)
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 4, true, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
        else if (variant == LH_VARIANT_SPEC && sc.use_qnodes == 2 && sc.prefer_q8 && sc.q8nodes)
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 7, true, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
        else if (variant == LH_VARIANT_SPEC && sc.use_qnodes == 2 && sc.stack_guard)
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 8, true, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
        else if (variant == LH_VARIANT_SPEC && sc.use_qnodes == 2)
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 3, true, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
        else
            hipLaunchKernelGGL((k_trace_persist_lane<ANYHIT, COUNT, 2, QN, 0>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s,
                               sc, n, org, dir, prim, t, u, v, occ, counters, cursor, min_active, tri_batch, AoSrc{});
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <bool QN>
int launch_fmt(const lh_dev_scene_t &sc, size_t n, const double *org, const double *dir,
               uint32_t *prim, double *t, double *u, double *v, int anyhit, uint8_t *occ,
               unsigned long long *counters, unsigned long long *cursor, int variant,
               int grid_blocks, int min_active, int tri_batch, size_t lds_bytes, hipStream_t s)
{
    if (anyhit) {
        if (counters) return launch_one<true, true, QN>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, variant, grid_blocks, min_active, tri_batch, lds_bytes, s);
        return launch_one<true, false, QN>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, variant, grid_blocks, min_active, tri_batch, lds_bytes, s);
    }
    if (counters) return launch_one<false, true, QN>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, variant, grid_blocks, min_active, tri_batch, lds_bytes, s);
    return launch_one<false, false, QN>(sc, n, org, dir, prim, t, u, v, occ, counters, cursor, variant, grid_blocks, min_active, tri_batch, lds_bytes, s);
}

int launch_stack(const lh_dev_scene_t &sc, size_t n, const double *org, const double *dir,
                 uint32_t *prim, double *t, double *u, double *v, int anyhit, uint8_t *occ,
                 unsigned long long *counters, unsigned long long *cursor, int variant,
                 int grid_blocks, int min_active, int tri_batch, size_t lds_bytes, hipStream_t s)
{
    /* the unified walk (variant 3) reads fp32 nodes only */
    if (sc.use_qnodes != 0 && variant != LH_VARIANT_UNIFIED)
        return launch_fmt<true>(sc, n, org, dir, prim, t, u, v, anyhit, occ, counters, cursor, variant, grid_blocks, min_active, tri_batch, lds_bytes, s);
    return launch_fmt<false>(sc, n, org, dir, prim, t, u, v, anyhit, occ, counters, cursor, variant, grid_blocks, min_active, tri_batch, lds_bytes, s);
}

} /* namespace */

/* the AO stage of a tile with the rays generated inside the any-hit kernel (SRC 1 above): nslots primary hits,
 * N = ntheta * nphi rays each, occluded rays counted per slot in d_occ_count (zeroed here).  Fragile hits go to
 * d_queue (6 words per entry, count + overflow flag in d_qcount[0..1]); lh_launch_ao_queue (lh_trace2.hip) runs
 * the reference walk for them.  Needs the 4-wide nodes (use_qnodes == 2) and a tree the LDS stack holds. */
extern "C" int lh_launch_trace_ao(const lh_dev_scene_t *sc, size_t nslots, int ntheta, int nphi, unsigned long long seed,
                                  const double *d_hitrec, const unsigned long long *d_slot_key, unsigned int *d_occ_count,
                                  unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks, int min_active,
                                  int tri_batch, uint32_t *d_queue, uint32_t *d_qcount, uint32_t qcap, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const size_t n = nslots * (size_t)(ntheta * nphi);
    if (n == 0) return 0;
    if (n >= ((size_t)1 << 32) || sc->use_qnodes != 2) return -1;
    lh_dev_scene_t scl = *sc;
    uint32_t need = 3 * sc->q4_depth + 5;
    const uint32_t cap = (sc->stack_cap >= 8 && sc->stack_cap < 64) ? sc->stack_cap : 64;
    if (need > cap) {                /* a deep (device-built) tree: 64 rows, the rare ray that needs more is queued for the reference walk */
        if (!sc->ref_nodes) return -1;
        need = cap; scl.stack_guard = 1;
    }
    need = (need + 1u) & ~1u;
    if (need < 16 && need != cap) need = 16;
    scl.stack_rows = need;
    const size_t lds_bytes = (size_t)need * LH_BLOCK * sizeof(int);
    {
        const size_t waves = (size_t)(grid_blocks > 0 ? grid_blocks : 1) * (LH_BLOCK / 64);
        if (scl.ray_chunk < 512) scl.ray_chunk = 512;      /* AO rays of a slot are coherent: longer ranges per wave (config 5: 92.9 -> 91.4 ms, tools/ao_sweep5.py) */
        size_t c = n / (waves * 4);
        if (c < 64) c = 64;
        if (c < scl.ray_chunk) scl.ray_chunk = (uint32_t)c;
        if (scl.ray_chunk == 0) scl.ray_chunk = 64;
    }
    AoSrc ao = {d_hitrec, d_slot_key, d_occ_count, seed, ntheta, nphi, d_queue, d_qcount, qcap};
    if (hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_qcount, 0, 2 * sizeof(uint32_t), s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_occ_count, 0, sizeof(unsigned int) * nslots, s) != hipSuccess) return -1;
#define LH_AO_LAUNCH(CNT, W) hipLaunchKernelGGL((k_trace_persist_lane<true, CNT, W, true, 1>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s, \
                           scl, n, (const double *)NULL, (const double *)NULL, (uint32_t *)NULL, (double *)NULL, (double *)NULL, \
                           (double *)NULL, (uint8_t *)NULL, d_counters, d_cursor, min_active, tri_batch, ao)
    if (scl.stack_guard) { if (d_counters) LH_AO_LAUNCH(true, 8); else LH_AO_LAUNCH(false, 8); }
    else { if (d_counters) LH_AO_LAUNCH(true, 3); else LH_AO_LAUNCH(false, 3); }
#undef LH_AO_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* one fused path-tracing pass (ray source 2): npaths paths of a w-wide tile at (x0, y0), spp samples per pixel starting at
 * sample s0.  d_radiance [npaths][3] is zeroed here; *d_nrays += rays traced, *d_maxdepth = max(., deepest vertex).  Needs the
 * 4-wide nodes; trees deeper than the LDS rows run the checked walk and lean on the reference walk for overflowing rays, so
 * they need the reference-order tree (returns -1 otherwise: the caller falls back to the wavefront passes). */
extern "C" int lh_launch_trace_pt(const lh_dev_scene_t *sc, size_t npaths, const lh_camera_t *cam, int x0, int y0, int w, int spp, int s0,
                                  int full_width, int max_depth, unsigned long long seed, const double *d_nrm9, const double *d_col9,
                                  const uint32_t *d_prim_mesh, const void *d_materials, const lh_material_t *override_mat,
                                  const float env_rgb[3], const void *d_env_map, int env_w, int env_h, int ref_weights,
                                  float *d_radiance, unsigned long long *d_nrays, unsigned int *d_maxdepth,
                                  unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks, int min_active,
                                  int tri_batch, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (npaths == 0) return 0;
    if (npaths >= ((size_t)1 << 31) || sc->use_qnodes != 2) return -1;
    lh_dev_scene_t scl = *sc;
    uint32_t need = 3 * sc->q4_depth + 5;
    const uint32_t cap = (sc->stack_cap >= 8 && sc->stack_cap < 64) ? sc->stack_cap : 64;
    if (need > cap) {
        if (!sc->ref_nodes) return -1;
        need = cap; scl.stack_guard = 1;
    }
    need = (need + 1u) & ~1u;
    if (need < 16 && need != cap) need = 16;
    scl.stack_rows = need;
    const size_t lds_bytes = (size_t)need * LH_BLOCK * sizeof(int);
    {
        const size_t waves = (size_t)(grid_blocks > 0 ? grid_blocks : 1) * (LH_BLOCK / 64);
        size_t c = npaths / (waves * 4);
        if (c < 64) c = 64;
        if (c < scl.ray_chunk) scl.ray_chunk = (uint32_t)c;
        if (scl.ray_chunk == 0) scl.ray_chunk = 64;
    }
    PtSrc pt;
    memset(&pt, 0, sizeof(pt));
    for (int k = 0; k < 16; k++) pt.cam.c2w[k] = cam->cam2world[k];
    pt.cam.flength = cam->flength; pt.cam.width = cam->width; pt.cam.height = cam->height; pt.cam.rh = cam->rh; pt.cam.ortho = cam->ortho;
    pt.x0 = x0; pt.y0 = y0; pt.w = w; pt.spp = spp; pt.s0 = s0; pt.full_width = full_width; pt.max_depth = max_depth;
    pt.use_override = override_mat != NULL; pt.ref_weights = ref_weights; pt.seed = seed;
    pt.nrm9 = d_nrm9; pt.col9 = d_col9; pt.prim_mesh = d_prim_mesh; pt.materials = (const DevMaterial *)d_materials;
    if (override_mat) { for (int k = 0; k < 3; k++) { pt.override_mat.kd[k] = override_mat->kd[k]; pt.override_mat.ks[k] = override_mat->ks[k]; pt.override_mat.kt[k] = override_mat->kt[k]; } pt.override_mat.ior = override_mat->ior; }
    pt.env.rgb[0] = env_rgb[0]; pt.env.rgb[1] = env_rgb[1]; pt.env.rgb[2] = env_rgb[2];
    pt.env.map = (const float4 *)d_env_map; pt.env.w = env_w; pt.env.h = env_h;
    pt.radiance = d_radiance; pt.nrays = d_nrays; pt.maxdepth = d_maxdepth;
    if (hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_radiance, 0, sizeof(float) * 3 * npaths, s) != hipSuccess) return -1;
#define LH_PT_LAUNCH(CNT, W) hipLaunchKernelGGL((k_trace_pt<CNT, W>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s, scl, npaths, d_counters, d_cursor, min_active, tri_batch, pt)
    if (scl.stack_guard) { if (d_counters) LH_PT_LAUNCH(true, 8); else LH_PT_LAUNCH(false, 8); }
    else { if (d_counters) LH_PT_LAUNCH(true, 3); else LH_PT_LAUNCH(false, 3); }
#undef LH_PT_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* the node formats (bit mask: 1 fp32 2-wide, 2 16-bit grid 2-wide, 4 16-bit grid 4-wide, 8 8-wide
 * compressed) the launch below reads for this scene and variant -- the same decisions, so that
 * lh_api.hip can upload a format the first time a variant asks for it */
extern "C" int lh_trace_formats_needed(const lh_dev_scene_t *sc, int variant)
{
    int uq = sc->use_qnodes;
    if (variant == LH_VARIANT_LEAN) variant = LH_VARIANT_SPEC;      /* same nodes; scenes it cannot take fall back to it */
    if (variant == LH_VARIANT_QUAD) return (sc->use_qnodes == 2) ? (4 | 16) : lh_trace_formats_needed(sc, LH_VARIANT_SPEC);
    if (variant == LH_VARIANT_UNIFIED) return 1;
    if (uq == 3) {
        if (variant == LH_VARIANT_SPEC) return 8;          /* rays whose stack would overflow go through the reference walk */
        uq = (variant == LH_VARIANT_UNIFIED4) ? 2 : 1;
    }
    if ((variant == LH_VARIANT_SPEC || variant == LH_VARIANT_UNIFIED4) && uq == 2) {
        if (3 * sc->q4_depth + 5 > 64 && sc->nodes_2wide_available) return 2;
        return 4;
    }
    if (uq == 0) return 1;
    return 2;
}

extern "C" int lh_launch_trace(const lh_dev_scene_t *sc, size_t n, const double *d_org,
                               const double *d_dir, uint32_t *d_prim, double *d_t, double *d_u,
                               double *d_v, int anyhit, uint8_t *d_occluded,
                               unsigned long long *d_counters, unsigned long long *d_workq,
                               int variant, int grid_blocks, int min_active, int tri_batch, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    /* stack entries needed: 2-wide walks <= tree depth + 1 (sentinel); the 4-wide walk pushes
     * up to 3 per level and writes up to 3 slots above its top.  The LDS stack is dynamic
     * shared memory of exactly that many rows (2-row granularity). */
    lh_dev_scene_t scl = *sc;
    uint32_t need = sc->max_depth + 1;
    if (sc->use_qnodes == 3) {
        if (variant == LH_VARIANT_SPEC) {
            /* worst case 7 per level + slack; capped at 48 rows (three workgroups per CU): a ray that would go
             * deeper is handed to the reference walk */
            need = 7 * sc->c8_depth + 10;
            if (need > 48) need = 48;
        } else if (variant == LH_VARIANT_UNIFIED4) scl.use_qnodes = 2;
        else scl.use_qnodes = 1;                 /* the 2-wide walks read the 16-bit grid nodes */
    }
    if (variant == LH_VARIANT_QUAD) {
        /* one ray per quad of lanes (lh_quad.hip); scenes without the child-major nodes take the default walk */
        if (sc->use_qnodes == 2 && sc->q4tnodes) {
            int of = 0;
            const int rc = lh_launch_trace_quad(sc, n, d_org, d_dir, d_prim, d_t, d_u, d_v, anyhit, d_occluded, d_counters,
                                                d_workq, grid_blocks, min_active, tri_batch, &of, stream);
            if (rc != 0) return rc;
            if (of)
                hipLaunchKernelGGL(k_overflow_fix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *sc, n, d_org, d_dir,
                                   d_prim, d_t, d_u, d_v, d_occluded, anyhit, d_counters);
            if (sc->ref_nodes)
                hipLaunchKernelGGL(k_ref_retrace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *sc, n, d_org, d_dir,
                                   d_prim, d_t, d_u, d_v, d_occluded, anyhit, d_counters);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
        variant = LH_VARIANT_SPEC;
    }
    bool over_fix = false;
    if (variant == LH_VARIANT_SPEC && scl.use_qnodes == 2 && sc->prefer_q8 && sc->q8nodes) {
        /* the 8-wide walk pushes up to 7 per level and keeps a scratch row; beyond 48 rows the rare ray that needs them is
         * finished by k_overflow_fix over the 4-wide nodes (always resident) */
        need = 7 * sc->q8_depth + 10;
        const uint32_t cap = (sc->stack_cap >= 16 && sc->stack_cap < 64) ? sc->stack_cap : 48;     /* 48 rows: three workgroups per CU */
        if (need > cap) { need = cap; over_fix = true; scl.stack_guard = 1; }
    } else if ((variant == LH_VARIANT_SPEC || variant == LH_VARIANT_UNIFIED4) && scl.use_qnodes == 2) {
        scl.prefer_q8 = 0;
        need = 3 * sc->q4_depth + 5;
        /* a very deep tree (chains of nested geometry): the 4-wide walk's worst case does not fit the
         * 64-row LDS stack.  Host-built trees come with the 2-wide nodes: that walk (<= LH_MAX_DEPTH + 1 rows) always fits.
         * Device-built trees have only the 4-wide nodes: 64 rows, and a ray that would overrun them (none in practice: the
         * bound is three pushes on every level) is finished by k_overflow_fix */
        const uint32_t cap = (sc->stack_cap >= 8 && sc->stack_cap < 64) ? sc->stack_cap : 64;     /* < 64: the tests' way to reach the overflow path */
        if (need > cap) {
            if (cap == 64 && sc->nodes_2wide_available) { scl.use_qnodes = 1; need = sc->max_depth + 1; }
            else if (variant == LH_VARIANT_SPEC) { need = cap; over_fix = true; scl.stack_guard = 1; }
            else return -1;
        }
    }
    need = (need + 1u) & ~1u;
    if (need < 16 && !over_fix) need = 16;
    if (need > 64) return -1;
    scl.stack_rows = need;
    const size_t lds_bytes = (size_t)need * LH_BLOCK * sizeof(int);
    /* rays per cursor atomic: the scene's setting, but never so large that a wave gets fewer than
     * ~4 ranges of a small batch (tail imbalance: late path-tracing bounces, small tiles) */
    {
        const size_t waves = (size_t)(grid_blocks > 0 ? grid_blocks : 1) * (LH_BLOCK / 64);
        size_t c = n / (waves * 4);
        if (c < 64) c = 64;
        if (c < scl.ray_chunk) scl.ray_chunk = (uint32_t)c;
        if (scl.ray_chunk == 0) scl.ray_chunk = 64;
    }
    sc = &scl;
    if (over_fix) {
        const int rc = launch_stack(*sc, n, d_org, d_dir, d_prim, d_t, d_u, d_v, anyhit, d_occluded,
                                    d_counters, d_workq, variant, grid_blocks, min_active, tri_batch, lds_bytes, s);
        if (rc != 0) return rc;
        hipLaunchKernelGGL(k_overflow_fix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *sc, n, d_org, d_dir,
                           d_prim, d_t, d_u, d_v, d_occluded, anyhit, d_counters);
        if (sc->ref_nodes)
            hipLaunchKernelGGL(k_ref_retrace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *sc, n, d_org, d_dir,
                               d_prim, d_t, d_u, d_v, d_occluded, anyhit, d_counters);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (sc->ref_nodes) {
        const int rc = launch_stack(*sc, n, d_org, d_dir, d_prim, d_t, d_u, d_v, anyhit, d_occluded,
                                    d_counters, d_workq, variant, grid_blocks, min_active, tri_batch, lds_bytes, s);
        if (rc != 0) return rc;
        hipLaunchKernelGGL(k_ref_retrace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *sc, n, d_org, d_dir,
                           d_prim, d_t, d_u, d_v, d_occluded, anyhit, d_counters);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    return launch_stack(*sc, n, d_org, d_dir, d_prim, d_t, d_u, d_v, anyhit, d_occluded,
                        d_counters, d_workq, variant, grid_blocks, min_active, tri_batch, lds_bytes, s);
}
