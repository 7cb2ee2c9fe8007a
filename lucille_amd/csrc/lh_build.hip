/*
 * lh_build.hip -- the traversal tree built ON THE DEVICE (LBVH), for scenes that are re-committed every frame.
 *
 * lucille rebuilds its accelerator in every ri_scene_setup (src/render/scene.c:84-98 -> ri_bvh_build,
 * src/render/bvh.c:276-379): with the host builder of lh_bvh.c that is 5.9 s in front of a 93 ms frame on the
 * BASELINE config 5 scene.  Hit records do not depend on the tree (any conservative BVH + the bit-exact
 * fp64 triangle test reproduces the reference), so the traversal tree may be built by whatever is fastest:
 *
 *   1. per-primitive fp32-outward boxes from the fp64 triangles; scene box (six atomics per workgroup);
 *   2. 63-bit Morton codes of the centroids (one scale for the three axes: cubic cells), radix-sorted with the primitive
 *      ids (hipcub);
 *   3. the binary radix tree over the sorted codes (Karras, "Maximizing Parallelism in the Construction of BVHs,
 *      Octrees, and k-d Trees", HPG 2012 -- the published algorithm, not code), ties between equal codes broken by
 *      position;
 *   3b. binned SAH inside every radix subtree of <= 512 primitives, one wave each, in LDS (k_sah_subtree);
 *   4. node boxes as range queries over block tables of the sorted primitives' boxes (no bottom-up hand-over, no fences), and
 *      in the same kernel the bottom-up SAH decision which subtrees of <= 4 primitives become ONE leaf;
 *   5. binned SAH, on the host, over the roots of the subtrees of <= `cut` primitives (HLBVH: Pantaleoni & Luebke 2010,
 *      Garanzha et al. 2011 -- the published idea): the top of a radix tree cuts through objects, the subtrees are fine;
 *   6. the same 64-byte 4-wide 16-bit-grid nodes the host builder emits (lh_q4node_t), collapsed level by level -- a 4-wide
 *      node takes its binary node's two children and opens the one with the largest area until it has four -- children of
 *      one node allocated adjacently; boxes quantised outward on the scene grid exactly as lh_bvh.c does (lo down, hi up);
 *   6b. for scenes whose ray dumps want them (lh_commit.hip), the same binary tree also as 128-byte 8-wide nodes (lh_q8node_t);
 *   7. the LDS stack rows the walk needs on this tree, from its deepest path (instead of the 3 x depth + 5 of any tree);
 *   8. the 48-byte triangle records (lh_tri32_t) in sorted = leaf order.
 *
 * Primitive numbering is the caller's (create_triangle_list order, bvh.c:1736-1826): ids ride along as payload.
 * BASELINE config 5 (21.1 M triangles): 0.06 s for the tree, 0.2 s for the whole commit, frame within 3 % of the frame on
 * the binned-SAH host tree (HISTORY.md 15); lh_commit.hip chooses (LH_BUILD=device, lh_accel_commit's build_threads ==
 * LH_BUILD_ON_DEVICE).  lucille's OWN tree, which exact-t ties, fragile hits and beams depend on, is built next to this one by
 * lh_refbuild.hip.
 */
#include <hip/hip_runtime.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <hipcub/hipcub.hpp>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <system_error>
#include <thread>
#include <vector>

#include "lh_bvh.h"
#include "lh_device.h"

namespace {

struct BNode {              /* inner node of the binary radix tree */
    int left, right;        /* >= 0: inner node; < 0: ~position of a leaf in the sorted order */
    int parent;
    uint32_t first, last;   /* range of sorted positions it covers */
    float lo[3], hi[3];
    float cost;             /* SAH cost of the cheapest way to finish this subtree (k_node_boxes) */
    int leaf;               /* 1: cheapest as ONE leaf of its <= 4 primitives */
};

/* SAH constants of the collapse decision: a triangle step of the walk against a node step (both one record fetch and ~100
 * instructions; the triangle pass runs at lower lane use) */
#define LH_SAH_CT 1.2f
#define LH_SAH_CI 1.0f

__device__ __forceinline__ float f_down(double d) { return __double2float_rd(d); }
__device__ __forceinline__ float f_up(double d) { return __double2float_ru(d); }

/* order-preserving float <-> uint for atomicMin / atomicMax */
__device__ __forceinline__ uint32_t f2o(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

/* bad[0]: a coordinate that is NaN / infinite / beyond 1e30; bad[1]: primitives the reference can never report (lh_bvh.c
 * tri_dead_class: two equal vertices -- they stay out of the traversal tree, marked by a NaN in plo[3 p]); bad[2]: some of them are
 * of class 2 (v1 == v2: rejected for rays whose direction components stay below LH_DEG_DCAP); bad[3]: the float bits of the largest
 * s2 of a zero-area triangle that STAYS in the tree (lh_bvh.c tri_zero_area_s2: rays beyond 1 / s2 go through the reference's own
 * walk).  drop = 0: nothing is marked */
#define LH_DEG_DCAP  1024.0
#define LH_DEG_S2CAP (1.0e-14 / (1.0e-15 * LH_DEG_DCAP))
__global__ __launch_bounds__(256) void k_prim_boxes(uint32_t n, const double *__restrict__ tri64, float *__restrict__ plo, float *__restrict__ phi,
                                                    uint32_t *__restrict__ scene /* 6 ordered uints: min xyz, max xyz */, int *__restrict__ bad, int drop)
{
    __shared__ uint32_t smin[4][3], smax[4][3];
    uint32_t omin[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, omax[3] = {0u, 0u, 0u};
    uint32_t ndead = 0, noise = 0;
    float s2keep = 0.0f;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const double *t = tri64 + 9 * (size_t)p;
        for (int k = 0; k < 3; k++) {
            const double a = t[k], b = t[3 + k], c = t[6 + k];
            if (!(fabs(a) <= 1.0e30) || !(fabs(b) <= 1.0e30) || !(fabs(c) <= 1.0e30)) atomicExch(bad, 1);
            const float lo = f_down(fmin(a, fmin(b, c))), hi = f_up(fmax(a, fmax(b, c)));
            plo[3 * (size_t)p + k] = lo; phi[3 * (size_t)p + k] = hi;
            const uint32_t ol = f2o(lo), oh = f2o(hi);
            omin[k] = ol < omin[k] ? ol : omin[k]; omax[k] = oh > omax[k] ? oh : omax[k];
        }
        int cls = 0;
        if (drop) {
            if ((t[0] == t[3] && t[1] == t[4] && t[2] == t[5]) || (t[0] == t[6] && t[1] == t[7] && t[2] == t[8])) cls = 1;
            else if (t[3] == t[6] && t[4] == t[7] && t[5] == t[8]) {
                const double sN = fabs(t[3] - t[0]) + fabs(t[4] - t[1]) + fabs(t[5] - t[2]);
                if (sN * sN * (1.0 + 1e-9) <= LH_DEG_S2CAP) cls = 2;
            }
            if (cls) { plo[3 * (size_t)p] = __uint_as_float(0x7fc00000u); ndead++; noise |= (uint32_t)(cls == 2); }
        }
        if (!cls) {                                  /* lh_bvh.h lh_zero_area_weight: what lh_bvh.c tri_zero_area_s2 calls */
            const double w = lh_zero_area_weight(t, t + 3, t + 6);
            if (w > 0.0) s2keep = fmaxf(s2keep, f_up(w));
        }
    }
    for (int off = 32; off >= 1; off >>= 1) s2keep = fmaxf(s2keep, __shfl_xor(s2keep, off));
    if ((threadIdx.x & 63) == 0 && s2keep > 0.0f) atomicMax((unsigned int *)&bad[3], __float_as_uint(s2keep));      /* positive floats order like their bits */
    if (drop) {
        for (int off = 32; off >= 1; off >>= 1) { ndead += (uint32_t)__shfl_xor((int)ndead, off); noise |= (uint32_t)__shfl_xor((int)noise, off); }
        if ((threadIdx.x & 63) == 0 && ndead) { atomicAdd(&bad[1], (int)ndead); if (noise) atomicExch(&bad[2], 1); }
    }
    /* scene bounds: six atomics per WORKGROUP of a grid that is a few thousand workgroups whatever n is.  Same-address
     * device-scope atomics serialise at ~95 ns each: one per thread was 126 M of them on a 21 M-triangle scene, one per wave
     * still 22 ms of a 60 ms build */
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t a = (uint32_t)__shfl_xor((int)omin[k], off), b = (uint32_t)__shfl_xor((int)omax[k], off);
            omin[k] = a < omin[k] ? a : omin[k]; omax[k] = b > omax[k] ? b : omax[k];
        }
        if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6][k] = omin[k]; smax[threadIdx.x >> 6][k] = omax[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        uint32_t lo = smin[0][k], hi = smax[0][k];
        for (int w = 1; w < 4; w++) { lo = smin[w][k] < lo ? smin[w][k] : lo; hi = smax[w][k] > hi ? smax[w][k] : hi; }
        atomicMin(&scene[k], lo); atomicMax(&scene[3 + k], hi);
    }
}

__device__ __forceinline__ uint64_t spread21(uint64_t x)
{
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(uint32_t n, const float *__restrict__ plo, const float *__restrict__ phi, const uint32_t *__restrict__ scene,
                         uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (plo[3 * (size_t)p] != plo[3 * (size_t)p]) { key[p] = ~0ull; val[p] = p; return; }      /* marked by k_prim_boxes: behind every 63-bit code, outside the tree */
    uint64_t code = 0;
    /* one scale for the three axes (the largest extent): cubic cells.  A scale per axis makes the cells of a flat scene -- a
     * floor with objects on it -- slabs, and every third split of the radix tree a cut across the thin direction */
    double ext = 0.0;
    for (int k = 0; k < 3; k++) { const double e = (double)o2f(scene[3 + k]) - (double)o2f(scene[k]); ext = e > ext ? e : ext; }
    for (int k = 0; k < 3; k++) {
        const float smin = o2f(scene[k]);
        const double c = 0.5 * ((double)plo[3 * (size_t)p + k] + (double)phi[3 * (size_t)p + k]);
        double q = ext > 0.0 ? (c - (double)smin) / ext * 2097152.0 : 0.0;
        if (q < 0.0) q = 0.0;
        if (q > 2097151.0) q = 2097151.0;
        code |= spread21((uint64_t)q) << k;
    }
    key[p] = code; val[p] = p;
}

/* common-prefix length of the keys at sorted positions i and j (-1 outside the array); equal codes: position decides */
__device__ __forceinline__ int delta(const uint64_t *__restrict__ key, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint64_t a = key[i], b = key[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

__global__ void k_radix_tree(int n, const uint64_t *__restrict__ key, BNode *__restrict__ nodes, int *__restrict__ leaf_parent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(key, n, i, i + 1) - delta(key, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(key, n, i, i - d);
    int lmax = 2;
    while (delta(key, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) if (delta(key, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(key, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2; ; t = (t + 1) / 2) {
        if (delta(key, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    BNode &nd = nodes[i];
    nd.first = (uint32_t)lo; nd.last = (uint32_t)hi;
    if (lo == gamma) { nd.left = ~gamma; leaf_parent[gamma] = i; } else { nd.left = gamma; nodes[gamma].parent = i; }
    if (hi == gamma + 1) { nd.right = ~(gamma + 1); leaf_parent[gamma + 1] = i; } else { nd.right = gamma + 1; nodes[gamma + 1].parent = i; }
    if (i == 0) nd.parent = -1;
}

/* ---- node boxes and the SAH leaf decision, one thread per inner node, no hand-over between threads ----
 * The textbook bottom-up refit (every leaf walks up, the second arrival at a node merges) needs two device-scope fences per
 * node -- on eight XCDs with an L2 each that is a cache write-back and invalidate: 137 ms for 21 M triangles.  A radix-tree
 * node covers a contiguous RANGE of the sorted primitives, so its box is a range query: boxes of the sorted primitives, then
 * of every 64 of them, every 4096, every 262144 (k_box_blocks); a node merges at most 63 + 63 entries per level. */
#define LH_BOX_RADIX 64u
#define LH_BOX_LEVELS 4

struct BoxTable { const float *lo[LH_BOX_LEVELS], *hi[LH_BOX_LEVELS]; };       /* level 0: the sorted primitives */

__global__ void k_sorted_boxes(uint32_t n, const uint32_t *__restrict__ sorted, const float *__restrict__ plo, const float *__restrict__ phi,
                               float *__restrict__ slo, float *__restrict__ shi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = sorted[i];
    for (int k = 0; k < 3; k++) { slo[3 * (size_t)i + k] = plo[3 * (size_t)p + k]; shi[3 * (size_t)i + k] = phi[3 * (size_t)p + k]; }
}

/* out[b] = union of in[64 b .. 64 b + 63]: a wave per output box */
__global__ void k_box_blocks(uint32_t n_in, const float *__restrict__ ilo, const float *__restrict__ ihi, float *__restrict__ olo, float *__restrict__ ohi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n_in) for (int k = 0; k < 3; k++) { lo[k] = ilo[3 * (size_t)i + k]; hi[k] = ihi[3 * (size_t)i + k]; }
    for (int k = 0; k < 3; k++)
        for (int off = 32; off >= 1; off >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
    if ((threadIdx.x & 63) == 0 && i < n_in) for (int k = 0; k < 3; k++) { olo[3 * (size_t)(i / 64u) + k] = lo[k]; ohi[3 * (size_t)(i / 64u) + k] = hi[k]; }
}

__device__ __forceinline__ void range_box(const BoxTable &T, uint32_t first, uint32_t last, float lo[3], float hi[3])
{
    for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    uint32_t i = first;
    while (i <= last) {
        uint32_t step = 1; int lev = 0;
        while (lev + 1 < LH_BOX_LEVELS && (i % (step * LH_BOX_RADIX)) == 0u && (uint64_t)i + (uint64_t)step * LH_BOX_RADIX - 1u <= (uint64_t)last) { step *= LH_BOX_RADIX; lev++; }
        const size_t e = (size_t)(i / step);
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], T.lo[lev][3 * e + k]); hi[k] = fmaxf(hi[k], T.hi[lev][3 * e + k]); }
        if ((uint64_t)i + step > 0xffffffffull) break;
        i += step;
    }
}

__device__ __forceinline__ float half_area(const float lo[3], const float hi[3])
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

/* bottom-up SAH of a subtree of at most LH_MAX_LEAF_TRIS primitives: the cost of the cheapest way to finish it -- as inner
 * node + its children's best, or as ONE leaf (same rule, same arithmetic as the round-2 refit).  A soup of unrelated triangles
 * keeps one triangle per leaf (its boxes barely shrink towards the leaves: the leaf's area is the node's), a tessellated
 * surface merges neighbours into leaves of up to four.  D bounds the recursion (a subtree of <= 4 leaves is <= 3 nodes deep) */
template <int D>
__device__ float subtree_cost(const BNode *__restrict__ nodes, const BoxTable &T, int ref, int leaf_max, bool *leaf_out)
{
    float lo[3], hi[3];
    if (ref < 0) {
        const size_t e = (size_t)~ref;
        for (int k = 0; k < 3; k++) { lo[k] = T.lo[0][3 * e + k]; hi[k] = T.hi[0][3 * e + k]; }
        if (leaf_out) *leaf_out = true;
        return LH_SAH_CT * half_area(lo, hi);
    }
    const BNode &nd = nodes[ref];
    range_box(T, nd.first, nd.last, lo, hi);
    const float area = half_area(lo, hi);
    float csum = 0.0f;
    if (D > 0) { csum += subtree_cost<(D > 0 ? D - 1 : 0)>(nodes, T, nd.left, leaf_max, NULL); csum += subtree_cost<(D > 0 ? D - 1 : 0)>(nodes, T, nd.right, leaf_max, NULL); }
    const uint32_t cnt = nd.last - nd.first + 1u;
    const float as_node = LH_SAH_CI * area + csum, as_leaf = LH_SAH_CT * area * (float)cnt;
    const bool leaf = cnt <= (uint32_t)leaf_max && as_leaf <= as_node;
    if (leaf_out) *leaf_out = leaf;
    return leaf ? as_leaf : as_node;
}

__global__ void k_node_boxes(int n, BNode *__restrict__ nodes, const BoxTable T, int leaf_max)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    BNode &nd = nodes[i];
    float lo[3], hi[3];
    range_box(T, nd.first, nd.last, lo, hi);
    for (int k = 0; k < 3; k++) { nd.lo[k] = lo[k]; nd.hi[k] = hi[k]; }
    bool leaf = false; float cost = 0.0f;
    if (nd.last - nd.first + 1u <= (uint32_t)leaf_max) cost = subtree_cost<LH_MAX_LEAF_TRIS - 1>(nodes, T, i, leaf_max, &leaf);
    nd.cost = cost; nd.leaf = leaf ? 1 : 0;
}

/* ---- binned SAH inside the subtrees (the bottom of the HLBVH) -----------------------------------------------------
 * The radix tree's subtrees of up to LH_SUB_MAX primitives are spatially compact, but inside them the cuts still fall where the
 * Morton code says.  Each is rebuilt top-down by ONE wave with the binned SAH (16 bins, three axes) in LDS: the wave permutes
 * the subtree's stretch of the sorted order and writes its nodes over the radix nodes of that stretch.  (A radix subtree over
 * the sorted positions a .. b is rooted at inner node a or b and owns, besides its root, exactly the inner nodes a + 1 .. b - 1;
 * the root keeps its index, so everything above still points at it.)  Boxes and leaf decisions come afterwards, from the same
 * range queries as for every other node. */
#ifndef LH_SUB_MAX
#define LH_SUB_MAX 512
#endif
#ifndef LH_SUB_BINS
#define LH_SUB_BINS 16
#endif

__global__ void k_sub_roots(int n, const BNode *__restrict__ nodes, uint32_t sub_max, uint32_t *__restrict__ out, uint32_t *__restrict__ nout)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const uint32_t size = nodes[i].last - nodes[i].first + 1u;
    if (size > sub_max || size < 3u) return;
    const int p = nodes[i].parent;
    if (p >= 0 && nodes[p].last - nodes[p].first + 1u <= sub_max) return;        /* inside a subtree */
    out[atomicAdd(nout, 1u)] = (uint32_t)i;
}

__global__ __launch_bounds__(64) void k_sah_subtree(uint32_t nroots, const uint32_t *__restrict__ roots, BNode *__restrict__ nodes,
                                                    uint32_t *__restrict__ sorted, const float *__restrict__ plo, const float *__restrict__ phi)
{
    __shared__ uint32_t pid[LH_SUB_MAX];                 /* primitive id of local item k (its position when the wave started) */
    __shared__ float blo[LH_SUB_MAX][3], bhi[LH_SUB_MAX][3];
    __shared__ uint16_t perm[LH_SUB_MAX], tmp[LH_SUB_MAX];       /* the order being built, as local items */
    __shared__ uint32_t bcnt[3][LH_SUB_BINS], bmin[3][LH_SUB_BINS][3], bmax[3][LH_SUB_BINS][3];
    __shared__ uint32_t stk_rng[LH_SUB_MAX], stk_node[LH_SUB_MAX];   /* ranges still to split: lo | hi << 16, and their node */
    if (blockIdx.x >= nroots) return;
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t r = roots[blockIdx.x];
    const uint32_t a = nodes[r].first, b = nodes[r].last, m = b - a + 1u;
    for (uint32_t k = lane; k < m; k += 64) {
        const uint32_t p = sorted[a + k];
        pid[k] = p; perm[k] = (uint16_t)k;
        for (int c = 0; c < 3; c++) { blo[k][c] = plo[3 * (size_t)p + c]; bhi[k][c] = phi[3 * (size_t)p + c]; }
    }
    if (lane == 0) { stk_rng[0] = 0u | (m << 16); stk_node[0] = r; }
    uint32_t used = 0;                                   /* nodes handed out besides the root: a + 1, a + 2, ... */
    int sp = 1;
    __syncthreads();
    while (sp > 0) {
        sp--;
        const uint32_t lo = stk_rng[sp] & 0xFFFFu, hi = stk_rng[sp] >> 16, node = stk_node[sp], cnt = hi - lo;
        /* ranges of up to four items (three quarters of all splits) are halved as they stand: they mostly end up as ONE leaf anyway */
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY}, scale[3] = {0.0f, 0.0f, 0.0f};
        int best = 0x7fffffff, owner = 0; float bc = INFINITY; uint32_t nleft = 0;
        if (cnt > 4u) {
        /* centroid bounds of the range */
        for (uint32_t k = lo + lane; k < hi; k += 64) {
            const uint32_t e = perm[k];
            for (int c = 0; c < 3; c++) { const float cc = 0.5f * (blo[e][c] + bhi[e][c]); clo[c] = fminf(clo[c], cc); chi[c] = fmaxf(chi[c], cc); }
        }
        for (int c = 0; c < 3; c++)
            for (int off = 32; off >= 1; off >>= 1) { clo[c] = fminf(clo[c], __shfl_xor(clo[c], off)); chi[c] = fmaxf(chi[c], __shfl_xor(chi[c], off)); }
        for (int t = lane; t < 3 * LH_SUB_BINS; t += 64) {
            bcnt[t / LH_SUB_BINS][t % LH_SUB_BINS] = 0u;
            for (int c = 0; c < 3; c++) { bmin[t / LH_SUB_BINS][t % LH_SUB_BINS][c] = 0xffffffffu; bmax[t / LH_SUB_BINS][t % LH_SUB_BINS][c] = 0u; }
        }
        __syncthreads();
        for (int c = 0; c < 3; c++) scale[c] = chi[c] > clo[c] ? (float)LH_SUB_BINS / (chi[c] - clo[c]) : 0.0f;
        for (uint32_t k = lo + lane; k < hi; k += 64) {
            const uint32_t e = perm[k];
            for (int ax = 0; ax < 3; ax++) {
                int j = (int)((0.5f * (blo[e][ax] + bhi[e][ax]) - clo[ax]) * scale[ax]);
                j = j < 0 ? 0 : (j > LH_SUB_BINS - 1 ? LH_SUB_BINS - 1 : j);
                atomicAdd(&bcnt[ax][j], 1u);
                for (int c = 0; c < 3; c++) { atomicMin(&bmin[ax][j][c], f2o(blo[e][c])); atomicMax(&bmax[ax][j][c], f2o(bhi[e][c])); }
            }
        }
        __syncthreads();
        /* candidate (LH_SUB_BINS - 1) ax + j: the split "bins 0 .. j | j + 1 .." along ax; a lane takes candidates lane, lane + 64 */
        bc = INFINITY; best = 0x7fffffff;
        for (int cand = lane; cand < 3 * (LH_SUB_BINS - 1); cand += 64) {
            const int ax = cand / (LH_SUB_BINS - 1), j = cand % (LH_SUB_BINS - 1);
            if (scale[ax] > 0.0f) {
                float ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY}, rl[3] = {INFINITY, INFINITY, INFINITY}, rh[3] = {-INFINITY, -INFINITY, -INFINITY};
                uint32_t nl = 0, nr = 0;
                for (int q = 0; q < LH_SUB_BINS; q++) {
                    const uint32_t cq = bcnt[ax][q];
                    if (!cq) continue;
                    if (q <= j) { nl += cq; for (int c = 0; c < 3; c++) { ll[c] = fminf(ll[c], o2f(bmin[ax][q][c])); lh[c] = fmaxf(lh[c], o2f(bmax[ax][q][c])); } }
                    else { nr += cq; for (int c = 0; c < 3; c++) { rl[c] = fminf(rl[c], o2f(bmin[ax][q][c])); rh[c] = fmaxf(rh[c], o2f(bmax[ax][q][c])); } }
                }
                if (nl && nr) {
                    const float cost = half_area(ll, lh) * (float)nl + half_area(rl, rh) * (float)nr;
                    if (cost < bc) { bc = cost; best = cand; nleft = nl; }          /* (candidates ascend: ties keep the lower) */
                }
            }
        }
        owner = lane;                                     /* the cheapest split (ties: the lower candidate), and whose nleft it is */
        for (int off = 32; off >= 1; off >>= 1) {
            const float oc = __shfl_xor(bc, off); const int ob = __shfl_xor(best, off), oo = __shfl_xor(owner, off);
            if (oc < bc || (oc == bc && ob < best)) { bc = oc; best = ob; owner = oo; }
        }
        }
        uint32_t nl = (uint32_t)__shfl((int)nleft, owner);
        const int ax = best < 0x7fffffff ? best / (LH_SUB_BINS - 1) : 0, j = best < 0x7fffffff ? best % (LH_SUB_BINS - 1) : 0;
        const bool sah = bc < INFINITY;
        if (!sah) nl = cnt / 2;                          /* equal centroids: halve the list as it stands */
        /* stable partition of perm[lo .. hi) into tmp, 64 items a round */
        uint32_t lbase = lo, rbase = lo + nl;
        for (uint32_t k0 = lo; k0 < hi; k0 += 64) {
            const uint32_t k = k0 + (uint32_t)lane;
            const bool in = k < hi; bool left = false; uint32_t e = 0;
            if (in) {
                e = perm[k];
                if (sah) {
                    int q = (int)((0.5f * (blo[e][ax] + bhi[e][ax]) - clo[ax]) * scale[ax]);
                    q = q < 0 ? 0 : (q > LH_SUB_BINS - 1 ? LH_SUB_BINS - 1 : q);
                    left = q <= j;
                } else left = (k - lo) < nl;
            }
            const unsigned long long ml = __ballot(in && left), mr = __ballot(in && !left);
            if (in) tmp[left ? lbase + (uint32_t)__popcll(ml & lt) : rbase + (uint32_t)__popcll(mr & lt)] = (uint16_t)e;
            lbase += (uint32_t)__popcll(ml); rbase += (uint32_t)__popcll(mr);
        }
        __syncthreads();
        for (uint32_t k = lo + lane; k < hi; k += 64) perm[k] = tmp[k];
        /* children: a single item is a leaf reference, a longer range gets a node and waits on the stack */
        const uint32_t mid = lo + nl;
        const bool push_l = mid - lo > 1u, push_r = hi - mid > 1u;
        uint32_t slot_l = 0, slot_r = 0;
        if (push_l) { slot_l = a + 1u + used; used++; }
        if (push_r) { slot_r = a + 1u + used; used++; }
        if (lane == 0) {
            BNode &nd = nodes[node];
            nd.first = a + lo; nd.last = a + hi - 1u;
            nd.left = push_l ? (int)slot_l : ~(int)(a + lo);
            nd.right = push_r ? (int)slot_r : ~(int)(a + mid);
            if (push_l) { nodes[slot_l].parent = (int)node; stk_rng[sp] = lo | (mid << 16); stk_node[sp] = slot_l; }
            if (push_r) { nodes[slot_r].parent = (int)node; stk_rng[sp + (push_l ? 1 : 0)] = mid | (hi << 16); stk_node[sp + (push_l ? 1 : 0)] = slot_r; }
        }
        sp += (push_l ? 1 : 0) + (push_r ? 1 : 0);
        __syncthreads();
    }
    for (uint32_t k = lane; k < m; k += 64) sorted[a + k] = pid[perm[k]];
}

/* ---- SAH over the top of the tree (HLBVH) ----------------------------------------------------------------------
 * A radix tree cuts space where the Morton code says, whatever lies there: its upper levels slice through objects and leave
 * overlapping halves, and those are the nodes every ray visits.  The subtrees of at most `cut` primitives below them are
 * compact patches and fine as they are.  So: the roots of those subtrees (k_cut_roots; ~3 n / cut of them) go to the host,
 * which builds a binned-SAH tree over their boxes in a few milliseconds (top_build) and hands it back as ordinary BNodes
 * behind the radix nodes; the collapse then starts at the new root. */
#ifndef LH_TOP_BINS
#define LH_TOP_BINS 32
#endif
struct CutRoot { int ref; uint32_t count; float lo[3], hi[3]; };

__global__ void k_cut_roots(int n, const BNode *__restrict__ nodes, const BoxTable T, uint32_t cut, CutRoot *__restrict__ out, uint32_t cap,
                            uint32_t *__restrict__ nout)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const BNode &nd = nodes[i];
    const uint32_t size = nd.last - nd.first + 1u;
    if (size <= cut) {
        const int p = nd.parent;
        if (p >= 0 && nodes[p].last - nodes[p].first + 1u <= cut) return;        /* inside a subtree */
        const uint32_t k = atomicAdd(nout, 1u);
        if (k < cap) { CutRoot r; r.ref = i; r.count = size; for (int c = 0; c < 3; c++) { r.lo[c] = nd.lo[c]; r.hi[c] = nd.hi[c]; } out[k] = r; }
        return;
    }
    for (int side = 0; side < 2; side++) {                                        /* a single primitive hanging off a large node */
        const int c = side ? nd.right : nd.left;
        if (c >= 0) continue;
        const uint32_t k = atomicAdd(nout, 1u);
        if (k < cap) {
            CutRoot r; r.ref = c; r.count = 1; const size_t e = (size_t)~c;
            for (int a = 0; a < 3; a++) { r.lo[a] = T.lo[0][3 * e + a]; r.hi[a] = T.hi[0][3 * e + a]; }
            out[k] = r;
        }
    }
}

static float top_area(const float lo[3], const float hi[3])
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

/* binned SAH (LH_TOP_BINS = 32 bins -- 16 until the end of round 3: config 5 86.1 -> 85.7 ms --, the three axes, subtrees weighted by their primitive counts) over it[b .. e).  A subtree over m items
 * has m - 1 nodes: it gets the slots out[nb .. nb + m - 2] (root first, then the left subtree's, then the right's), so the
 * layout does not depend on who builds what and the large subtrees near the top are built by threads of their own.  A node's
 * global index = base + slot.  Returns the reference of the subtree's root */
static int g_top_bins = LH_TOP_BINS;        /* LH_DEVICE_TOP_BINS (2 .. LH_TOP_BINS) */
static int top_build(CutRoot *it, int b, int e, BNode *out, int nb, int base, int par_depth, int level = 0)
{
    if (e - b == 1) return it[b].ref;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = b; i < e; i++)
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(lo[k], it[i].lo[k]); hi[k] = fmaxf(hi[k], it[i].hi[k]);
            const float c = 0.5f * (it[i].lo[k] + it[i].hi[k]);
            clo[k] = fminf(clo[k], c); chi[k] = fmaxf(chi[k], c);
        }
    const int NB = g_top_bins;
    int best_axis = -1, best_bin = 0; float best_cost = INFINITY;
    for (int k = 0; k < 3; k++) {
        const float ext = chi[k] - clo[k];
        if (!(ext > 0.0f)) continue;
        float blo[LH_TOP_BINS][3], bhi[LH_TOP_BINS][3]; double bcnt[LH_TOP_BINS];
        for (int j = 0; j < NB; j++) { bcnt[j] = 0.0; for (int a = 0; a < 3; a++) { blo[j][a] = INFINITY; bhi[j][a] = -INFINITY; } }
        const float scale = (float)NB / ext;
        for (int i = b; i < e; i++) {
            int j = (int)((0.5f * (it[i].lo[k] + it[i].hi[k]) - clo[k]) * scale);
            j = j < 0 ? 0 : (j >= NB ? NB - 1 : j);
            bcnt[j] += it[i].count;
            for (int a = 0; a < 3; a++) { blo[j][a] = fminf(blo[j][a], it[i].lo[a]); bhi[j][a] = fmaxf(bhi[j][a], it[i].hi[a]); }
        }
        float rarea[LH_TOP_BINS]; double rcnt[LH_TOP_BINS];
        {
            float rl[3] = {INFINITY, INFINITY, INFINITY}, rh[3] = {-INFINITY, -INFINITY, -INFINITY}; double rc = 0.0;
            for (int j = NB - 1; j >= 1; j--) {
                for (int a = 0; a < 3; a++) { rl[a] = fminf(rl[a], blo[j][a]); rh[a] = fmaxf(rh[a], bhi[j][a]); }
                rc += bcnt[j]; rcnt[j] = rc; rarea[j] = rc > 0.0 ? top_area(rl, rh) : 0.0f;
            }
        }
        float ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY}; double lc = 0.0;
        for (int j = 0; j + 1 < NB; j++) {
            for (int a = 0; a < 3; a++) { ll[a] = fminf(ll[a], blo[j][a]); lh[a] = fmaxf(lh[a], bhi[j][a]); }
            lc += bcnt[j];
            if (lc <= 0.0 || rcnt[j + 1] <= 0.0) continue;
            const float cost = (float)(top_area(ll, lh) * lc + rarea[j + 1] * rcnt[j + 1]);
            if (cost < best_cost) { best_cost = cost; best_axis = k; best_bin = j; }
        }
    }
    int mid;
    /* (a distribution that lets the SAH peel a few items per level -- an exponentially spaced line -- would recurse as deep as
     * it is long: beyond 48 levels the list is simply halved) */
    if (best_axis >= 0 && level < 48) {
        const int k = best_axis; const float scale = (float)NB / (chi[k] - clo[k]);
        CutRoot *m = std::partition(it + b, it + e, [&](const CutRoot &r) {
            int j = (int)((0.5f * (r.lo[k] + r.hi[k]) - clo[k]) * scale);
            j = j < 0 ? 0 : (j >= NB ? NB - 1 : j);
            return j <= best_bin;
        });
        mid = (int)(m - it);
    } else mid = b;
    if (mid <= b || mid >= e) mid = b + (e - b) / 2;                 /* equal centroids: split the list */
    const int nl = nb + 1, nr = nb + 1 + (mid - b - 1);              /* first slots of the two subtrees */
    int l, r;
    bool forked = false;
    if (par_depth > 0 && e - b > 4096) {
        try {
            std::thread th([&]() { l = top_build(it, b, mid, out, nl, base, par_depth - 1, level + 1); });
            forked = true;
            r = top_build(it, mid, e, out, nr, base, par_depth - 1, level + 1);
            th.join();
        } catch (const std::system_error &) { forked = false; }         /* no thread to be had: this one does both halves */
    }
    if (!forked) {
        l = top_build(it, b, mid, out, nl, base, 0, level + 1);
        r = top_build(it, mid, e, out, nr, base, 0, level + 1);
    }
    BNode &nd = out[nb];
    memset(&nd, 0, sizeof(nd));
    nd.left = l; nd.right = r; nd.parent = -1;
    for (int k = 0; k < 3; k++) { nd.lo[k] = lo[k]; nd.hi[k] = hi[k]; }
    return base + nb;
}

struct Child { float lo[3], hi[3]; int node; uint32_t first, count; };   /* node >= 0: inner binary node with > 4 primitives */

__device__ __forceinline__ void child_of(const BNode *__restrict__ nodes, const uint32_t *__restrict__ sorted,
                                         const float *__restrict__ plo, const float *__restrict__ phi, int ref, Child &c, int leaf_max)
{
    if (ref < 0) {
        const uint32_t pos = (uint32_t)~ref, p = sorted[pos];
        for (int k = 0; k < 3; k++) { c.lo[k] = plo[3 * (size_t)p + k]; c.hi[k] = phi[3 * (size_t)p + k]; }
        c.node = -1; c.first = pos; c.count = 1;
    } else {
        const BNode &b = nodes[ref];
        for (int k = 0; k < 3; k++) { c.lo[k] = b.lo[k]; c.hi[k] = b.hi[k]; }
        c.first = b.first; c.count = b.last - b.first + 1;
        c.node = b.leaf ? -1 : ref;            /* k_node_boxes' SAH decision: one leaf of its <= leaf_max primitives, or an inner node */
        (void)leaf_max;
    }
}

__device__ __forceinline__ void quant_axis(double g, double st, float lo, float hi, uint32_t &w)
{
    double ql = floor(((double)lo - g) / st), qh = ceil(((double)hi - g) / st);
    if (ql < 0.0) ql = 0.0;
    if (ql > 65535.0) ql = 65535.0;
    if (qh < 0.0) qh = 0.0;
    if (qh > 65535.0) qh = 65535.0;
    while (ql > 0.0 && g + ql * st > (double)lo) ql -= 1.0;
    while (qh < 65535.0 && g + qh * st < (double)hi) qh += 1.0;
    w = (uint32_t)ql | ((uint32_t)qh << 16);
}

/* ---- which binary nodes become 4-wide nodes (the host builder's rule, lh_bvh.c dp4_fill) ------------------------------
 * A ray pays one record per 4-wide node whose box it enters, so the expected cost of a collapse is the sum of the areas of the
 * binary nodes kept as 4-wide nodes.  cost[k-1] of a binary node = the cheapest way to hang its subtree into k free child slots
 * of a 4-wide parent: as one 4-wide node of its own (its area + the best split of ITS four slots over its two children), or
 * dissolved into its two children with the k slots split i : k - i (split[k-1] = i; 0 = a node of its own; split[0] = how its
 * own four slots are shared).  Children before parents: the binary tree's levels (k_bfs_level) in reverse.  The greedy rule
 * of round 2 -- open the child of largest area until four slots are full -- is LH_DEVICE_COLLAPSE=greedy. */
/* node slots of a sibling group as a function of the parity of the index it starts at: a group of two or more inner children
 * starts at an EVEN index (two 64-byte nodes share the 128-byte line the L2 fetches: a group of two costs one line instead of
 * possibly two, a group of four two instead of possibly three), i.e. it takes one empty node first if it would start at an odd
 * one; single children fill whatever comes.  The starts follow from a prefix "sum" over these pairs -- composition of the
 * parity functions, associative -- so the numbering stays a scan (what the host builder does with a running index). */
struct PL { uint32_t l0, l1; };        /* length if the start is even / odd */
struct PLOp {
    __host__ __device__ __forceinline__ PL operator()(const PL &a, const PL &b) const
    {
        PL r;
        r.l0 = a.l0 + ((a.l0 & 1u) ? b.l1 : b.l0);
        r.l1 = a.l1 + (((1u + a.l1) & 1u) ? b.l1 : b.l0);
        return r;
    }
};

template <int W> struct DPW { double cost[W]; uint8_t split[W]; };
typedef DPW<4> DP4;
typedef DPW<8> DP8;

__device__ __forceinline__ bool is_inner(const BNode *__restrict__ nodes, int ref) { return ref >= 0 && !nodes[ref].leaf; }

__global__ void k_bfs_level(uint32_t nin, const uint32_t *__restrict__ in, const BNode *__restrict__ nodes, uint32_t *__restrict__ out, uint32_t *__restrict__ cursor)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nin) return;
    const BNode &b = nodes[in[i]];
    if (is_inner(nodes, b.left)) out[atomicAdd(cursor, 1u)] = (uint32_t)b.left;
    if (is_inner(nodes, b.right)) out[atomicAdd(cursor, 1u)] = (uint32_t)b.right;
}

template <int W>
__global__ void k_dp_level(uint32_t cnt, const uint32_t *__restrict__ list, const BNode *__restrict__ nodes, DPW<W> *__restrict__ dp)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt) return;
    const uint32_t b = list[t];
    const BNode &nd = nodes[b];
    double cl[W], cr[W];                                  /* [i]: the child's subtree into i slots */
    for (int i = 0; i < W; i++) { cl[i] = 0.0; cr[i] = 0.0; }
    if (is_inner(nodes, nd.left)) for (int i = 1; i < W; i++) cl[i] = dp[nd.left].cost[i - 1];
    if (is_inner(nodes, nd.right)) for (int i = 1; i < W; i++) cr[i] = dp[nd.right].cost[i - 1];
    double g[W + 1]; uint8_t gi[W + 1];
    for (int k = 2; k <= W; k++) {
        g[k] = 1e300; gi[k] = 1;
        for (int i = 1; i < k; i++) { const double c = cl[i] + cr[k - i]; if (c < g[k]) { g[k] = c; gi[k] = (uint8_t)i; } }
    }
    DPW<W> o;
    o.cost[0] = (double)half_area(nd.lo, nd.hi) + g[W]; o.split[0] = gi[W];
    for (int k = 2; k <= W; k++) {
        if (o.cost[0] <= g[k]) { o.cost[k - 1] = o.cost[0]; o.split[k - 1] = 0; }
        else { o.cost[k - 1] = g[k]; o.split[k - 1] = gi[k]; }
    }
    dp[b] = o;
}

/* the W slots of wide node b as the table says: (subtree, slots) pairs, a subtree dissolving into its children while its
 * entry says so; children come out left to right */
template <int W>
__device__ __forceinline__ int slots_of(const BNode *__restrict__ nodes, const uint32_t *__restrict__ sorted, const float *__restrict__ plo,
                                        const float *__restrict__ phi, const DPW<W> *__restrict__ dp, int b, Child *ch, int leaf_max)
{
    int sr[W + 2], sk[W + 2], sp = 0, n = 0;
    const int i0 = dp[b].split[0];
    sr[sp] = nodes[b].right; sk[sp] = W - i0; sp++;
    sr[sp] = nodes[b].left; sk[sp] = i0; sp++;
    while (sp > 0) {
        sp--;
        const int ref = sr[sp], k = sk[sp];
        const int i = (is_inner(nodes, ref) && k >= 2) ? (int)dp[ref].split[k - 1] : 0;
        if (i != 0) {
            sr[sp] = nodes[ref].right; sk[sp] = k - i; sp++;
            sr[sp] = nodes[ref].left; sk[sp] = i; sp++;
        } else child_of(nodes, sorted, plo, phi, ref, ch[n++], leaf_max);
    }
    return n;
}

/* one level of the 4-wide collapse: work item = (binary node, index of its 4-wide node) */
__global__ void k_collapse_level(uint32_t nwork, const uint2 *__restrict__ work_in, uint2 *__restrict__ work_out,
                                 uint32_t *__restrict__ counters /* [0] next 4-wide index, [1] work_out count */,
                                 const BNode *__restrict__ nodes, const uint32_t *__restrict__ sorted,
                                 const float *__restrict__ plo, const float *__restrict__ phi,
                                 const float3 glo, const float3 gstep, lh_q4node_t *__restrict__ q4, int leaf_max, const DP4 *__restrict__ dp,
                                 PL *__restrict__ cnt_out /* counting pass: node slots per work item (by start parity), [nwork] = 0 */,
                                 const PL *__restrict__ offs /* emitting pass: their exclusive prefix composition */, uint32_t node_base, int pair_align)
{
    const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (cnt_out && wi == nwork) { cnt_out[wi].l0 = 0u; cnt_out[wi].l1 = 0u; }
    if (wi >= nwork) return;
    if (work_in[wi].x == 0xffffffffu) { if (cnt_out) { cnt_out[wi].l0 = 0u; cnt_out[wi].l1 = 0u; } return; }        /* a padding slot of the level above */
    const int b = (int)work_in[wi].x; const uint32_t k4 = work_in[wi].y;
    Child ch[4]; int n = 2;
    if (dp) {
        n = slots_of<4>(nodes, sorted, plo, phi, dp, b, ch, leaf_max);
    } else {
    child_of(nodes, sorted, plo, phi, nodes[b].left, ch[0], leaf_max);
    child_of(nodes, sorted, plo, phi, nodes[b].right, ch[1], leaf_max);
    }
    while (!dp && n < 4) {
        int best = -1; float ba = -1.0f;
        for (int c = 0; c < n; c++)
            if (ch[c].node >= 0) {
                const float dx = ch[c].hi[0] - ch[c].lo[0], dy = ch[c].hi[1] - ch[c].lo[1], dz = ch[c].hi[2] - ch[c].lo[2];
                const float a = dx * dy + dy * dz + dz * dx;
                if (a > ba) { ba = a; best = c; }
            }
        if (best < 0) break;
        const int g = ch[best].node;
        child_of(nodes, sorted, plo, phi, nodes[g].left, ch[best], leaf_max);
        child_of(nodes, sorted, plo, phi, nodes[g].right, ch[n], leaf_max);
        n++;
    }
    int ninner = 0;
    for (int c = 0; c < n; c++) ninner += ch[c].node >= 0;
    if (cnt_out) { cnt_out[wi].l0 = (uint32_t)ninner; cnt_out[wi].l1 = (uint32_t)ninner + ((pair_align && ninner >= 2) ? 1u : 0u); return; }
    /* children are numbered in the order of their parents (a prefix scan over the level's work items, not an atomic counter whose
     * order is whoever arrives first): neighbours in space stay neighbours in the array at every level, as in the host builder's
     * level order */
    uint32_t base4 = 0, basew = 0;
    if (offs) {
        const uint32_t off = (node_base & 1u) ? offs[wi].l1 : offs[wi].l0;
        base4 = node_base + off; basew = off;
        if (pair_align && ninner >= 2 && (base4 & 1u)) {          /* the group moves up to the next line: an empty node first */
            lh_q4node_t pad;
            for (int c = 0; c < 4; c++) { for (int k = 0; k < 3; k++) pad.w[c][k] = 65535u; pad.ref[c] = LH_REF_EMPTY; }
            q4[base4] = pad;
            work_out[basew] = make_uint2(0xffffffffu, base4);
            base4++; basew++;
        }
    } else if (ninner) { base4 = atomicAdd(&counters[0], (uint32_t)ninner); basew = atomicAdd(&counters[1], (uint32_t)ninner); }
    lh_q4node_t out;
    const double g[3] = {glo.x, glo.y, glo.z}, st[3] = {gstep.x, gstep.y, gstep.z};
    int slot = 0;
    for (int c = 0; c < 4; c++) {
        if (c < n) {
            for (int k = 0; k < 3; k++) quant_axis(g[k], st[k], ch[c].lo[k], ch[c].hi[k], out.w[c][k]);
            if (ch[c].node >= 0) {
                out.ref[c] = (int32_t)(base4 + (uint32_t)slot);
                work_out[basew + (uint32_t)slot] = make_uint2((uint32_t)ch[c].node, base4 + (uint32_t)slot);
                slot++;
            } else out.ref[c] = ~(int32_t)((ch[c].first << 2) | (ch[c].count - 1u));
        } else {
            for (int k = 0; k < 3; k++) out.w[c][k] = 65535u;          /* lo = 65535, hi = 0: inverted */
            out.ref[c] = LH_REF_EMPTY;
        }
    }
    q4[k4] = out;
}

/* one level of the 8-wide collapse (lh_q8node_t: one 128-byte record = one cache line, for ray dumps over scenes that do not
 * fit the Infinity Cache): as k_collapse_level, opening the larger-area inner child until there are eight; children go to octant
 * slots -- the walk visits slot s with priority s ^ (ray octant), so a child takes the free slot whose diagonal its centroid
 * offset points along most (the host builder's rule, lh_bvh.c build8q) */
__global__ void k_collapse8_level(uint32_t nwork, const uint2 *__restrict__ work_in, uint2 *__restrict__ work_out,
                                  uint32_t *__restrict__ counters, const BNode *__restrict__ nodes, const uint32_t *__restrict__ sorted,
                                  const float *__restrict__ plo, const float *__restrict__ phi, const float3 glo, const float3 gstep,
                                  lh_q8node_t *__restrict__ q8, int leaf_max, const DP8 *__restrict__ dp,
                                  uint32_t *__restrict__ cnt_out, const uint32_t *__restrict__ offs, uint32_t node_base)
{
    const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (cnt_out && wi == nwork) cnt_out[wi] = 0u;
    if (wi >= nwork) return;
    const int b = (int)work_in[wi].x; const uint32_t k8 = work_in[wi].y;
    Child ch[8]; int n = 2;
    if (dp) n = slots_of<8>(nodes, sorted, plo, phi, dp, b, ch, leaf_max);
    else {
    child_of(nodes, sorted, plo, phi, nodes[b].left, ch[0], leaf_max);
    child_of(nodes, sorted, plo, phi, nodes[b].right, ch[1], leaf_max);
    }
    while (!dp && n < 8) {
        int best = -1; float ba = -1.0f;
        for (int c = 0; c < n; c++)
            if (ch[c].node >= 0) {
                const float dx = ch[c].hi[0] - ch[c].lo[0], dy = ch[c].hi[1] - ch[c].lo[1], dz = ch[c].hi[2] - ch[c].lo[2];
                const float a = dx * dy + dy * dz + dz * dx;
                if (a > ba) { ba = a; best = c; }
            }
        if (best < 0) break;
        const int g = ch[best].node;
        child_of(nodes, sorted, plo, phi, nodes[g].left, ch[best], leaf_max);
        child_of(nodes, sorted, plo, phi, nodes[g].right, ch[n], leaf_max);
        n++;
    }
    int ninner = 0;
    for (int c = 0; c < n; c++) ninner += ch[c].node >= 0;
    if (cnt_out) { cnt_out[wi] = (uint32_t)ninner; return; }
    uint32_t base8 = 0, basew = 0;
    if (offs) { base8 = node_base + offs[wi]; basew = offs[wi]; }
    else if (ninner) { base8 = atomicAdd(&counters[0], (uint32_t)ninner); basew = atomicAdd(&counters[1], (uint32_t)ninner); }
    if (!q8) {                                            /* counting pass: only the work list of the next level */
        int slot = 0;
        for (int c = 0; c < n; c++)
            if (ch[c].node >= 0) { work_out[basew + (uint32_t)slot] = make_uint2((uint32_t)ch[c].node, base8 + (uint32_t)slot); slot++; }
        return;
    }
    double cen[3];
    for (int k = 0; k < 3; k++) {
        float lo = ch[0].lo[k], hi = ch[0].hi[k];
        for (int c = 1; c < n; c++) { lo = fminf(lo, ch[c].lo[k]); hi = fmaxf(hi, ch[c].hi[k]); }
        cen[k] = 0.5 * ((double)lo + (double)hi);
    }
    int slot_of[8]; bool used[8];
    for (int c = 0; c < 8; c++) { slot_of[c] = -1; used[c] = false; }
    for (int k = 0; k < n; k++) {
        int bc = -1, bs = -1; double bv = -1.0e300;
        for (int c = 0; c < n; c++) {
            if (slot_of[c] >= 0) continue;
            double d[3];
            for (int a = 0; a < 3; a++) d[a] = 0.5 * ((double)ch[c].lo[a] + (double)ch[c].hi[a]) - cen[a];
            for (int sl = 0; sl < 8; sl++) {
                if (used[sl]) continue;
                const double v = ((sl & 1) ? d[0] : -d[0]) + ((sl & 2) ? d[1] : -d[1]) + ((sl & 4) ? d[2] : -d[2]);
                if (v > bv) { bv = v; bc = c; bs = sl; }
            }
        }
        slot_of[bc] = bs; used[bs] = true;
    }
    lh_q8node_t out;
    const double g[3] = {glo.x, glo.y, glo.z}, st[3] = {gstep.x, gstep.y, gstep.z};
    for (int sl = 0; sl < 8; sl++) { for (int k = 0; k < 3; k++) out.w[sl][k] = 65535u; out.ref[sl] = LH_REF_EMPTY; }
    int slot = 0;
    for (int sl = 0; sl < 8; sl++)                        /* inner children adjacent, in slot order */
        for (int c = 0; c < n; c++) {
            if (slot_of[c] != sl) continue;
            for (int k = 0; k < 3; k++) quant_axis(g[k], st[k], ch[c].lo[k], ch[c].hi[k], out.w[sl][k]);
            if (ch[c].node >= 0) {
                out.ref[sl] = (int32_t)(base8 + (uint32_t)slot);
                work_out[basew + (uint32_t)slot] = make_uint2((uint32_t)ch[c].node, base8 + (uint32_t)slot);
                slot++;
            } else out.ref[sl] = ~(int32_t)((ch[c].first << 2) | (ch[c].count - 1u));
        }
    q8[k8] = out;
}

/* ---- how many LDS stack rows the walk really needs ------------------------------------------------------------------
 * A step at node X leaves at most (children of X) - 1 entries on the stack, so a ray that is about to step at X holds at most
 * the sum of that over X's ancestors.  The walk sizes its stack for the largest such sum (+ 5: sentinel and the step's own
 * writes) instead of 3 x depth: a 20-level tree of this builder needs fewer than the 64 rows up to which the walk runs without
 * a stack check.  (Re-laying the nodes depth first instead of level by level was tried here as well: 92.4 -> 95.3 ms on
 * BASELINE config 5 -- rays of a wave share the level-ordered lines -- and removed.) */
__global__ void k_stack_need(uint32_t begin, uint32_t end, const lh_q4node_t *__restrict__ q4, uint32_t *__restrict__ acc, uint32_t *__restrict__ need_max)
{
    const uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= end) return;
    uint32_t nch = 0;
    for (int c = 0; c < 4; c++) nch += q4[k].ref[c] != LH_REF_EMPTY;
    if (nch == 0) return;                                   /* a padding node: nobody refers to it, nobody wrote acc[k] */
    const uint32_t mine = k == 0 ? 0u : acc[k];
    atomicMax(need_max, mine);
    for (int c = 0; c < 4; c++) { const int32_t r = q4[k].ref[c]; if (r >= 0) acc[r] = mine + (nch ? nch - 1u : 0u); }
}

/* a scene of <= 4 primitives: one node, one leaf */
__global__ void k_single_leaf(uint32_t n, const uint32_t *__restrict__ scene, const float3 glo, const float3 gstep, lh_q4node_t *__restrict__ q4)
{
    if (threadIdx.x || blockIdx.x) return;
    lh_q4node_t out;
    const double g[3] = {glo.x, glo.y, glo.z}, st[3] = {gstep.x, gstep.y, gstep.z};
    for (int c = 0; c < 4; c++) { for (int k = 0; k < 3; k++) out.w[c][k] = 65535u; out.ref[c] = LH_REF_EMPTY; }
    for (int k = 0; k < 3; k++) quant_axis(g[k], st[k], o2f(scene[k]), o2f(scene[3 + k]), out.w[0][k]);
    out.ref[0] = ~(int32_t)((0u << 2) | (n - 1u));
    q4[0] = out;
}

__global__ void k_tri32(uint32_t n, const uint32_t *__restrict__ sorted, const double *__restrict__ tri64, lh_tri32_t *__restrict__ out)
{
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = sorted[i];
    const double *t = tri64 + 9 * (size_t)p;
    double e1[3], e2[3];
    lh_tri32_t o;
    for (int k = 0; k < 3; k++) { e1[k] = t[3 + k] - t[k]; e2[k] = t[6 + k] - t[k]; o.v0[k] = (float)t[k]; }
    o.e1x = (float)e1[0]; o.e1y = (float)e1[1]; o.e1z = (float)e1[2];
    o.e2x = (float)e2[0]; o.e2y = (float)e2[1]; o.e2z = (float)e2[2];
    const double n1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    const double n2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
    o.prim = p;
    o.ne1 = f_up(n1 * (1.0 + 1e-6)); o.ne2 = f_up(n2 * (1.0 + 1e-6));
    out[i] = o;
}

static inline void dfree(void *p) { if (p) (void)hipFree(p); }

#define BCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(err, errlen, "%s failed: %s", #x, hipGetErrorString(e_)); goto fail; } } while (0)

} /* namespace */

/* d_tri64: ntris x 9 doubles (primitive-id order) on the current device.  On success *d_q4nodes (*nq4 records), *d_tri32 (ntris + 2 records) and -- if want_q8 and the tree has more than one node -- *d_q8nodes (*nq8 records)
 * are hipMalloc'ed here and owned by the caller; bmin / bmax / grid as lh_bvh_t.
 * Returns 0, -1 (err filled), or -2 for a NaN / infinite / > 1e30 coordinate. */
extern "C" int lh_device_build(uint32_t ntris, const double *d_tri64, void **d_q4nodes, uint32_t *nq4, uint32_t *q4_depth, uint32_t *q4_stack,
                               int want_q8, void **d_q8nodes, uint32_t *nq8, uint32_t *q8_depth,
                               void **d_tri32, float bmin[3], float bmax[3], float grid_lo[3], float grid_step[3],
                               uint32_t *nlive, double *deg_dcap, void *stream, char *err, size_t errlen)
{
    hipStream_t s = (hipStream_t)stream;
    const uint32_t ntot = ntris;         /* every primitive: tri32 records, the sort */
    uint32_t n = ntris;                  /* the primitives in the tree: ntot minus what the reference can never report (k_prim_boxes) */
    float *plo = NULL, *phi = NULL; uint32_t *scene = NULL, *val_in = NULL, *sorted = NULL, *counters = NULL, *ncut = NULL; float *boxes = NULL; CutRoot *cuts = NULL;
    uint64_t *key_in = NULL, *key = NULL; int *leaf_parent = NULL, *bad = NULL; BNode *nodes = NULL; void *tmp = NULL; size_t tmp_bytes = 0;
    uint2 *work[2] = {NULL, NULL}; DP4 *dp = NULL; uint32_t *bfs = NULL; lh_q4node_t *q4 = NULL; lh_q8node_t *q8 = NULL; lh_tri32_t *t32 = NULL; uint32_t *lay = NULL, *offs = NULL, need_rows = 0; PL *pl_in = NULL, *pl_out = NULL; std::vector<uint32_t> lvl_begin, lb2; DP8 *dp8 = NULL;
    const unsigned nb = (ntot + 255) / 256;
    uint32_t h_scene[6], level = 0, nwork = 0, nq = 1;
    const uint32_t init_scene[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    int h_bad[4] = {0, 0, 0, 0}, leaf_max = LH_MAX_LEAF_TRIS;     /* leaves of up to 4 triangles WHERE THE SAH SAYS SO (k_node_boxes): forced 4-triangle leaves cost S-soup-1M 37 % (tools/experiments/leaf_probe.py), the SAH keeps that soup at one per leaf */
    { const char *e = getenv("LH_DEVICE_LEAF"); if (e && atoi(e) >= 1 && atoi(e) <= LH_MAX_LEAF_TRIS) leaf_max = atoi(e); }
    uint32_t cut = 512;                             /* primitives per subtree below the SAH-built top (LH_DEVICE_CUT; 0: plain radix tree).  config 5, frame /
                                                       tree time: 64 -> 86.9 ms / 0.125 s, 128 -> 87.3 / 0.058, 512 -> 87.6 / 0.029, 2048 -> 87.8 / 0.023; 256 makes
                                                       that tree one level too deep for the unchecked walk (97.6 ms) and is retried coarser (below) */
    { const char *e = getenv("LH_DEVICE_CUT"); if (e && atoi(e) >= 0) cut = (uint32_t)atoi(e); }
    const uint32_t cut_cap = cut ? (uint32_t)std::min<uint64_t>((uint64_t)ntot, std::max<uint64_t>(65536u, 16ull * ntot / cut)) : 0u;
    int root_ref = 0;
    { const char *e = getenv("LH_DEVICE_TOP_BINS"); g_top_bins = (e && atoi(e) >= 2 && atoi(e) <= LH_TOP_BINS) ? atoi(e) : LH_TOP_BINS; }
    const int pair_align = !(getenv("LH_Q4_PAIRS") && atoi(getenv("LH_Q4_PAIRS")) == 0);
    const bool use_dp = !(getenv("LH_DEVICE_COLLAPSE") && strcmp(getenv("LH_DEVICE_COLLAPSE"), "greedy") == 0);
    *d_q4nodes = NULL; *d_tri32 = NULL; *nq4 = 0; *q4_depth = 0; *q4_stack = 0; *d_q8nodes = NULL; *nq8 = 0; *q8_depth = 0;
    if (nlive) *nlive = ntot;
    if (deg_dcap) *deg_dcap = INFINITY;
    if (ntot == 0) return 0;
    /* LH_BUILD_TIMING=1: phase times on stderr (each mark synchronises the stream: diagnostics only) */
    const bool timing = getenv("LH_BUILD_TIMING") != NULL;
    double tmark = 0.0;
    auto mark = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        const double now = ts.tv_sec + 1e-9 * ts.tv_nsec;
        if (what) fprintf(stderr, "[lucille_hip] device build: %-28s %8.2f ms\n", what, (now - tmark) * 1e3);
        tmark = now;
    };
    mark(NULL);

    BCHK(hipMalloc((void **)&plo, sizeof(float) * 3 * (size_t)ntot)); BCHK(hipMalloc((void **)&phi, sizeof(float) * 3 * (size_t)ntot));
    BCHK(hipMalloc((void **)&scene, sizeof(uint32_t) * 8)); BCHK(hipMalloc((void **)&bad, sizeof(int) * 4));
    /* triangles the reference can never report stay out of the tree (lh_bvh.c tri_dead_class; LH_DROP_DEGENERATE=0: all stay);
     * a scene of nothing else, or of a single leaf, is built as it was handed over */
    for (int drop = (ntot > LH_MAX_LEAF_TRIS && !(getenv("LH_DROP_DEGENERATE") && atoi(getenv("LH_DROP_DEGENERATE")) == 0)) ? 1 : 0; ; drop = 0) {
        BCHK(hipMemcpyAsync(scene, init_scene, sizeof(init_scene), hipMemcpyHostToDevice, s));
        BCHK(hipMemsetAsync(bad, 0, sizeof(int) * 4, s));
        hipLaunchKernelGGL(k_prim_boxes, dim3(nb < 2048u ? nb : 2048u), dim3(256), 0, s, ntot, d_tri64, plo, phi, scene, bad, drop);
        BCHK(hipMemcpyAsync(h_scene, scene, sizeof(h_scene), hipMemcpyDeviceToHost, s));
        BCHK(hipMemcpyAsync(h_bad, bad, sizeof(h_bad), hipMemcpyDeviceToHost, s));
        BCHK(hipStreamSynchronize(s));
        if (h_bad[0]) { dfree(plo); dfree(phi); dfree(scene); dfree(bad); return -2; }
        if (!drop || (uint32_t)h_bad[1] < ntot) break;
    }
    n = ntot - (uint32_t)h_bad[1];
    if (nlive) *nlive = n;
    if (deg_dcap && h_bad[2]) *deg_dcap = LH_DEG_DCAP;
    if (deg_dcap && h_bad[3]) {                  /* zero-area triangles in the tree: rays beyond 1 / s2 are the reference's own walk's */
        float s2; memcpy(&s2, &h_bad[3], sizeof(s2));
        if (s2 > 0.0f && 1.0 / (double)s2 < *deg_dcap) *deg_dcap = 1.0 / (double)s2;
    }
    mark("boxes + scene bounds");
    for (int k = 0; k < 3; k++) {
        uint32_t lo = h_scene[k], hi = h_scene[3 + k]; float fl, fh;
        lo = (lo & 0x80000000u) ? (lo & 0x7fffffffu) : ~lo; hi = (hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi;
        memcpy(&fl, &lo, 4); memcpy(&fh, &hi, 4);
        bmin[k] = fl; bmax[k] = fh;
        const double ext = (double)fh - (double)fl;
        double st = ext > 0.0 ? ext / 65535.0 * (1.0 + 1e-6) : 1e-30;
        float fs = (float)st; if ((double)fs < st) fs = nextafterf(fs, INFINITY);
        grid_lo[k] = fl; grid_step[k] = fs;
    }
    BCHK(hipMalloc((void **)&q4, sizeof(lh_q4node_t) * (2 * (size_t)n + 2)));      /* <= n - 1 nodes + at most one padding node per sibling group; cut to size at the end */
    BCHK(hipMalloc((void **)&t32, sizeof(lh_tri32_t) * ((size_t)ntot + 2)));
    BCHK(hipMalloc((void **)&sorted, sizeof(uint32_t) * (size_t)ntot));
    {
        const float3 glo = make_float3(grid_lo[0], grid_lo[1], grid_lo[2]), gst = make_float3(grid_step[0], grid_step[1], grid_step[2]);
        if (ntot <= LH_MAX_LEAF_TRIS) {
            /* identity order, one leaf (the only place a device-built leaf holds more than leaf_max triangles) */
            uint32_t ids[LH_MAX_LEAF_TRIS]; for (uint32_t i = 0; i < n; i++) ids[i] = i;
            BCHK(hipMemcpyAsync(sorted, ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_single_leaf, dim3(1), dim3(64), 0, s, n, scene, glo, gst, q4);
            nq = 1; level = 1;
        } else {
            /* every primitive is sorted (the dropped ones carry the key ~0: behind every 63-bit Morton code -- hence all 64 bits);
             * the tree is built over the first n sorted positions */
            BCHK(hipMalloc((void **)&key_in, sizeof(uint64_t) * (size_t)ntot)); BCHK(hipMalloc((void **)&key, sizeof(uint64_t) * (size_t)ntot));
            BCHK(hipMalloc((void **)&val_in, sizeof(uint32_t) * (size_t)ntot));
            hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, s, ntot, plo, phi, scene, key_in, val_in);
            BCHK(hipcub::DeviceRadixSort::SortPairs(NULL, tmp_bytes, key_in, key, val_in, sorted, (int)ntot, 0, 64, s));
            BCHK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
            BCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_in, key, val_in, sorted, (int)ntot, 0, 64, s));
            mark("morton + radix sort");
          if (n <= LH_MAX_LEAF_TRIS) {          /* a handful of live triangles among the dropped ones: one leaf over sorted[0 .. n) */
            hipLaunchKernelGGL(k_single_leaf, dim3(1), dim3(64), 0, s, n, scene, glo, gst, q4);
            nq = 1; level = 1;
          } else {
            BCHK(hipMalloc((void **)&nodes, sizeof(BNode) * ((size_t)(n - 1) + cut_cap)));
            BCHK(hipMalloc((void **)&leaf_parent, sizeof(int) * (size_t)n));
            hipLaunchKernelGGL(k_radix_tree, dim3((n - 1 + 255) / 256), dim3(256), 0, s, (int)n, key, nodes, leaf_parent);
            mark("radix tree");
            if (n > LH_SUB_MAX && !(getenv("LH_DEVICE_SUBSAH") && atoi(getenv("LH_DEVICE_SUBSAH")) == 0)) {
                /* the roots of the subtrees of <= LH_SUB_MAX primitives (into val_in: free since the sort), then a wave per subtree */
                uint32_t h_nsub = 0;
                BCHK(hipMemsetAsync(bad, 0, sizeof(int), s));
                hipLaunchKernelGGL(k_sub_roots, dim3((n - 1 + 255) / 256), dim3(256), 0, s, (int)n, (const BNode *)nodes, (uint32_t)LH_SUB_MAX, val_in, (uint32_t *)bad);
                BCHK(hipMemcpyAsync(&h_nsub, bad, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                BCHK(hipStreamSynchronize(s));
                if (h_nsub) hipLaunchKernelGGL(k_sah_subtree, dim3(h_nsub), dim3(64), 0, s, h_nsub, (const uint32_t *)val_in, nodes, sorted, (const float *)plo, (const float *)phi);
                mark("binned SAH inside the subtrees");
            }
            BoxTable T;
            {
                /* boxes of the sorted primitives and of their blocks of 64 / 4096 / 262144, all in one allocation */
                size_t cnt[LH_BOX_LEVELS], total = 0;
                cnt[0] = n;
                for (int l = 1; l < LH_BOX_LEVELS; l++) cnt[l] = (cnt[l - 1] + LH_BOX_RADIX - 1) / LH_BOX_RADIX;
                for (int l = 0; l < LH_BOX_LEVELS; l++) total += cnt[l];
                BCHK(hipMalloc((void **)&boxes, sizeof(float) * 6 * total));
                size_t off = 0;
                for (int l = 0; l < LH_BOX_LEVELS; l++) { T.lo[l] = boxes + 6 * off; T.hi[l] = boxes + 6 * off + 3 * cnt[l]; off += cnt[l]; }
                hipLaunchKernelGGL(k_sorted_boxes, dim3(nb), dim3(256), 0, s, n, (const uint32_t *)sorted, (const float *)plo, (const float *)phi,
                                   (float *)T.lo[0], (float *)T.hi[0]);
                for (int l = 1; l < LH_BOX_LEVELS; l++)
                    hipLaunchKernelGGL(k_box_blocks, dim3((unsigned)((cnt[l - 1] + 255) / 256)), dim3(256), 0, s, (uint32_t)cnt[l - 1], T.lo[l - 1], T.hi[l - 1],
                                       (float *)T.lo[l], (float *)T.hi[l]);
                hipLaunchKernelGGL(k_node_boxes, dim3((n - 1 + 255) / 256), dim3(256), 0, s, (int)n, nodes, T, leaf_max);
                mark("node boxes + SAH leaves");
            }
            BCHK(hipMalloc((void **)&work[0], sizeof(uint2) * (2 * (size_t)n + 2))); BCHK(hipMalloc((void **)&work[1], sizeof(uint2) * (2 * (size_t)n + 2)));
            BCHK(hipMalloc((void **)&counters, sizeof(uint32_t) * 2));
            BCHK(hipMalloc((void **)&lay, sizeof(uint32_t) * ((size_t)n + 1)));
            BCHK(hipMalloc((void **)&offs, sizeof(uint32_t) * ((size_t)n + 1)));
            BCHK(hipMalloc((void **)&pl_in, sizeof(PL) * (2 * (size_t)n + 2))); BCHK(hipMalloc((void **)&pl_out, sizeof(PL) * (2 * (size_t)n + 2)));
            {
                size_t sb = 0, sb2 = 0; const PL zero = {0u, 0u};
                BCHK(hipcub::DeviceScan::ExclusiveSum(NULL, sb, lay, offs, (int)n, s));
                BCHK(hipcub::DeviceScan::ExclusiveScan(NULL, sb2, pl_in, pl_out, PLOp(), zero, (int)(2 * (size_t)n + 2), s));
                if (sb2 > sb) sb = sb2;
                if (sb > tmp_bytes) { dfree(tmp); tmp = NULL; tmp_bytes = sb; BCHK(hipMalloc(&tmp, tmp_bytes)); }
            }
            if (cut_cap) { BCHK(hipMalloc((void **)&cuts, sizeof(CutRoot) * (size_t)cut_cap)); BCHK(hipMalloc((void **)&ncut, sizeof(uint32_t))); }
            if (use_dp) {
                BCHK(hipMalloc((void **)&dp, sizeof(DP4) * ((size_t)(n - 1) + cut_cap)));
                BCHK(hipMalloc((void **)&bfs, sizeof(uint32_t) * ((size_t)(n - 1) + cut_cap + 1)));
            }
            /* top + collapse + the rows its deepest path needs.  A finer cut gives the better tree (config 5: 92.7 ms at 256 against
             * 93.9 at 1024) unless it makes the tree one level too deep for the unchecked walk's 64 LDS rows (97.6 ms): then the next
             * coarser cut is tried -- a second attempt costs ~25 ms of a 0.2 s commit */
            for (int attempt = 0; ; attempt++) {
                const uint32_t cc = attempt == 0 ? cut : (attempt == 1 ? cut * 4u : cut * 16u);
                root_ref = 0;
                if (cc > 0 && n > 4 * cc) {
                    uint32_t h_ncut = 0;
                    BCHK(hipMemsetAsync(ncut, 0, sizeof(uint32_t), s));
                    hipLaunchKernelGGL(k_cut_roots, dim3((n - 1 + 255) / 256), dim3(256), 0, s, (int)n, (const BNode *)nodes, T, cc, cuts, cut_cap, ncut);
                    BCHK(hipMemcpyAsync(&h_ncut, ncut, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    BCHK(hipStreamSynchronize(s));
                    if (h_ncut >= 2 && h_ncut <= cut_cap) {                    /* (more roots than room: a degenerate tree, left as it is) */
                        std::vector<CutRoot> items(h_ncut); std::vector<BNode> top(h_ncut - 1);
                        BCHK(hipMemcpy(items.data(), cuts, sizeof(CutRoot) * (size_t)h_ncut, hipMemcpyDeviceToHost));
                        std::sort(items.begin(), items.end(), [](const CutRoot &a, const CutRoot &b) { return a.ref < b.ref; });   /* the append order is not reproducible */
                        root_ref = top_build(items.data(), 0, (int)h_ncut, top.data(), 0, (int)(n - 1), 5);      /* up to 32 threads near the top */
                        BCHK(hipMemcpyAsync(nodes + (n - 1), top.data(), sizeof(BNode) * top.size(), hipMemcpyHostToDevice, s));
                        BCHK(hipStreamSynchronize(s));
                    }
                    mark("SAH over the subtree roots");
                }
                if (use_dp) {
                    /* the binary tree's levels from the root, then the table from the deepest level up */
                    uint32_t tot = 1, h_c = 0; lb2.clear();
                    const uint32_t rr = (uint32_t)root_ref;
                    BCHK(hipMemcpyAsync(bfs, &rr, sizeof(rr), hipMemcpyHostToDevice, s));
                    lb2.push_back(0); lb2.push_back(1);
                    while (lb2[lb2.size() - 1] > lb2[lb2.size() - 2]) {
                        const uint32_t b0 = lb2[lb2.size() - 2], b1 = lb2[lb2.size() - 1];
                        BCHK(hipMemsetAsync(counters, 0, sizeof(uint32_t), s));
                        hipLaunchKernelGGL(k_bfs_level, dim3((b1 - b0 + 255) / 256), dim3(256), 0, s, b1 - b0, (const uint32_t *)(bfs + b0), (const BNode *)nodes, bfs + b1, counters);
                        BCHK(hipMemcpyAsync(&h_c, counters, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                        BCHK(hipStreamSynchronize(s));
                        tot += h_c; lb2.push_back(tot);
                        if (lb2.size() > 4096) { snprintf(err, errlen, "device build: the binary tree is more than 4096 levels deep"); goto fail; }
                    }
                    for (size_t l = lb2.size() - 2; l-- > 0;) {
                        const uint32_t b0 = lb2[l], b1 = lb2[l + 1];
                        if (b1 > b0) hipLaunchKernelGGL(k_dp_level<4>, dim3((b1 - b0 + 255) / 256), dim3(256), 0, s, b1 - b0, (const uint32_t *)(bfs + b0), (const BNode *)nodes, dp);
                    }
                    mark("slots table of the collapse");
                }
                /* level-by-level collapse; every level's children are allocated adjacently */
                {
                    const uint2 root = make_uint2((uint32_t)root_ref, 0u);
                    BCHK(hipMemcpyAsync(work[0], &root, sizeof(root), hipMemcpyHostToDevice, s));
                }
                nwork = 1; nq = 1; level = 0; lvl_begin.clear(); need_rows = 0;
                while (nwork > 0) {
                    /* count the level's inner children, prefix-sum them (children numbered in the order of their parents), emit */
                    PL tot = {0u, 0u}; uint32_t total = 0;
                    hipLaunchKernelGGL(k_collapse_level, dim3((nwork + 1 + 127) / 128), dim3(128), 0, s, nwork, (const uint2 *)work[level & 1], work[(level + 1) & 1],
                                       counters, (const BNode *)nodes, (const uint32_t *)sorted, (const float *)plo, (const float *)phi, glo, gst, q4, leaf_max, (const DP4 *)dp,
                                       pl_in, (const PL *)NULL, nq, pair_align);
                    { size_t tb = tmp_bytes; const PL zero = {0u, 0u}; BCHK(hipcub::DeviceScan::ExclusiveScan(tmp, tb, pl_in, pl_out, PLOp(), zero, (int)(nwork + 1), s)); }
                    BCHK(hipMemcpyAsync(&tot, pl_out + nwork, sizeof(PL), hipMemcpyDeviceToHost, s));
                    hipLaunchKernelGGL(k_collapse_level, dim3((nwork + 127) / 128), dim3(128), 0, s, nwork, (const uint2 *)work[level & 1], work[(level + 1) & 1],
                                       counters, (const BNode *)nodes, (const uint32_t *)sorted, (const float *)plo, (const float *)phi, glo, gst, q4, leaf_max, (const DP4 *)dp,
                                       (PL *)NULL, (const PL *)pl_out, nq, pair_align);
                    BCHK(hipStreamSynchronize(s));
                    lvl_begin.push_back(nq);                                  /* first index of the level the kernel just filled */
                    total = (nq & 1u) ? tot.l1 : tot.l0;
                    nq += total; nwork = total; level++;
                    if (level > 200) { snprintf(err, errlen, "device build: runaway collapse"); goto fail; }
                }
                mark("collapse to 4-wide nodes");
                if (nq > 1) {
                    /* level l = indices [lb[l], lb[l + 1]): lb = 0, 1, then what the iterations recorded */
                    std::vector<uint32_t> lb; lb.push_back(0);
                    for (size_t k = 0; k < lvl_begin.size(); k++) lb.push_back(lvl_begin[k]);
                    lb.push_back(nq);
                    while (lb.size() >= 2 && lb[lb.size() - 1] == lb[lb.size() - 2]) lb.pop_back();       /* the last iteration adds nothing */
                    const int nl = (int)lb.size() - 1;
                    BCHK(hipMemsetAsync(lay + nq, 0, sizeof(uint32_t), s));
                    for (int l = 0; l < nl; l++)
                        hipLaunchKernelGGL(k_stack_need, dim3((lb[l + 1] - lb[l] + 255) / 256), dim3(256), 0, s, lb[l], lb[l + 1], (const lh_q4node_t *)q4, lay, lay + nq);
                    BCHK(hipMemcpyAsync(&need_rows, lay + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    BCHK(hipStreamSynchronize(s));
                    mark("stack rows of the deepest path");
                }
                if (need_rows + 5u <= LH_ROWS_UNCHECKED || cut == 0 || attempt == 2 || n <= 4 * cc) break;
            }
            if (want_q8 && nq > 1) {
                /* the same binary tree once more as 8-wide nodes.  How many there will be is not bounded by the 4-wide count (a
                 * node that opens eight small subtrees leaves eight nodes where the 4-wide collapse leaves four): the first
                 * pass only counts, the second writes into an allocation of exactly that size */
                uint32_t nw8 = 0, n8 = 0, lev8 = 0, n8_counted = 0;
                if (use_dp && lb2.size() >= 2) {
                    /* the slot table for eight slots, over the same levels of the binary tree */
                    BCHK(hipMalloc((void **)&dp8, sizeof(DP8) * ((size_t)(n - 1) + cut_cap)));
                    for (size_t l = lb2.size() - 2; l-- > 0;) {
                        const uint32_t b0 = lb2[l], b1 = lb2[l + 1];
                        if (b1 > b0) hipLaunchKernelGGL(k_dp_level<8>, dim3((b1 - b0 + 255) / 256), dim3(256), 0, s, b1 - b0, (const uint32_t *)(bfs + b0), (const BNode *)nodes, dp8);
                    }
                }
                for (int pass = 0; pass < 2; pass++) {
                    if (pass == 1) { n8_counted = n8; BCHK(hipMalloc((void **)&q8, sizeof(lh_q8node_t) * (size_t)n8_counted)); }
                    {
                        const uint2 root = make_uint2((uint32_t)root_ref, 0u);
                        BCHK(hipMemcpyAsync(work[0], &root, sizeof(root), hipMemcpyHostToDevice, s));
                    }
                    nw8 = 1; n8 = 1; lev8 = 0;
                    while (nw8 > 0) {
                        uint32_t total = 0;
                        hipLaunchKernelGGL(k_collapse8_level, dim3((nw8 + 1 + 127) / 128), dim3(128), 0, s, nw8, (const uint2 *)work[lev8 & 1], work[(lev8 + 1) & 1],
                                           counters, (const BNode *)nodes, (const uint32_t *)sorted, (const float *)plo, (const float *)phi, glo, gst, q8, leaf_max, (const DP8 *)dp8,
                                           lay, (const uint32_t *)NULL, n8);
                        { size_t tb = tmp_bytes; BCHK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, lay, offs, (int)(nw8 + 1), s)); }
                        BCHK(hipMemcpyAsync(&total, offs + nw8, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                        hipLaunchKernelGGL(k_collapse8_level, dim3((nw8 + 127) / 128), dim3(128), 0, s, nw8, (const uint2 *)work[lev8 & 1], work[(lev8 + 1) & 1],
                                           counters, (const BNode *)nodes, (const uint32_t *)sorted, (const float *)plo, (const float *)phi, glo, gst, q8, leaf_max, (const DP8 *)dp8,
                                           (uint32_t *)NULL, (const uint32_t *)offs, n8);
                        BCHK(hipStreamSynchronize(s));
                        n8 += total; nw8 = total; lev8++;
                        if (lev8 > 200 || n8 > n) { snprintf(err, errlen, "device build: runaway 8-wide collapse"); goto fail; }
                    }
                }
                if (n8 != n8_counted) { snprintf(err, errlen, "device build: the 8-wide collapse counted %u nodes and wrote %u", n8_counted, n8); goto fail; }
                *nq8 = n8; *q8_depth = lev8;
                mark("collapse to 8-wide nodes");
            }
          }
        }
    }
    hipLaunchKernelGGL(k_tri32, dim3(nb), dim3(256), 0, s, ntot, (const uint32_t *)sorted, d_tri64, t32);
    BCHK(hipGetLastError());
    BCHK(hipStreamSynchronize(s));
    mark("tri32 records");
    {
        /* the node array was sized for the worst case: keep what is used */
        lh_q4node_t *fit = NULL;
        if (hipMalloc((void **)&fit, sizeof(lh_q4node_t) * (size_t)nq) == hipSuccess) {
            if (hipMemcpyAsync(fit, q4, sizeof(lh_q4node_t) * (size_t)nq, hipMemcpyDeviceToDevice, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) { dfree(q4); q4 = fit; }
            else dfree(fit);
        }
    }
    *d_q4nodes = q4; *d_q8nodes = q8; *d_tri32 = t32; *nq4 = nq; *q4_depth = level; *q4_stack = nq > 1 ? need_rows + 5u : 0u;     /* LDS stack rows the walk needs (0: unknown, 3 x depth + 5) */
    dfree(plo); dfree(phi); dfree(scene); dfree(bad); dfree(key_in); dfree(key); dfree(val_in); dfree(sorted);
    dfree(nodes); dfree(leaf_parent); dfree(boxes); dfree(cuts); dfree(ncut); dfree(lay); dfree(tmp); dfree(work[0]); dfree(work[1]); dfree(counters); dfree(dp); dfree(dp8); dfree(bfs); dfree(offs); dfree(pl_in); dfree(pl_out);
    mark("free temporaries");
    return 0;
fail:
    dfree(plo); dfree(phi); dfree(scene); dfree(bad); dfree(key_in); dfree(key); dfree(val_in); dfree(sorted);
    dfree(nodes); dfree(leaf_parent); dfree(boxes); dfree(cuts); dfree(ncut); dfree(lay); dfree(tmp); dfree(work[0]); dfree(work[1]); dfree(counters); dfree(dp); dfree(dp8); dfree(bfs); dfree(offs); dfree(pl_in); dfree(pl_out);
    dfree(q4); dfree(q8); dfree(t32);
    return -1;
}
