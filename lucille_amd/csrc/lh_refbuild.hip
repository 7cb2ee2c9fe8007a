/*
 * lh_refbuild.hip -- lucille's OWN tree built on the device, bit for bit.
 *
 * The reference-order tree (lh_refbvh.c: ri_bvh_build, src/render/bvh.c:276-379; bvh_construct :1328-1564;
 * bin_triangle_edge :1571-1692; find_cut_from_bin :1230-1326; SAH :1210-1228; bbox_add_margin :1697-1731) decides the two
 * results of the reference that depend on its tree -- exact-t tie winners and beam visibility -- and carries the reference walk
 * that settles fragile hits.  On the host it costs 2.9 s for the 21.1 M triangles of BASELINE config 5, in the background of a
 * 0.2 s device commit: a scene that is re-committed every frame (scene.c:84-98) either waits for it or renders with "larger
 * primitive id wins".  Every step of that builder has a result that does not depend on the order it is computed in:
 *
 *   bins        integer histograms of floor((box.min - node.min) * 64 / size) per axis            -> atomics
 *   cut         189 candidates per node, double arithmetic in the reference's operation order      -> one thread per node
 *   partition   "lefts keep their order, rights are written from the end backwards" (:1437-1478)   -> a prefix sum of the
 *               left flags gives every element its place
 *   bounds      min / max of doubles                                                               -> ordered-integer atomics
 *
 * so the tree is built LEVEL BY LEVEL over all nodes of a depth at once, every element pass a streaming pass over HBM.  No
 * fused multiply-add anywhere (fp contract off): the doubles are the host's doubles.  Node numbering is breadth-first here and
 * depth-first on the host; nothing depends on it (tests/test_gpu_refbuild.py compares the two trees node by node from the root).
 */
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lh_refbvh.h"

namespace {

#define RB_LEAF 16u             /* BVH_NTRIS_LEAF bvh.c:81 */
#define RB_BINS 64              /* BVH_BIN_SIZE   bvh.c:82 */
#define RB_EPS 1.0e-14          /* RI_EPS src/base/common.h:27 */
#define RB_INF 1.0e38           /* RI_INFINITY include/ri.h:47 */
#define RB_NONE 0xffffffffu
#define RB_CHUNK 2048u          /* elements per workgroup of the histogram and the bounds pass */
#define RB_HIST (2 * 3 * RB_BINS)

struct RBox { double lo[3], hi[3]; };
struct RSeg {                   /* a node of the current level with more than RB_LEAF elements */
    uint32_t il, ir;            /* its range of the element array */
    int32_t node; uint32_t nl;
    double bmin[3], bmax[3], inv[3];
    double pos; int32_t axis, pad;
};

__device__ __forceinline__ unsigned long long d2o(double d) { const unsigned long long u = (unsigned long long)__double_as_longlong(d); return (u >> 63) ? ~u : (u | 0x8000000000000000ull); }
__device__ __forceinline__ double o2d(unsigned long long o) { return __longlong_as_double((long long)((o >> 63) ? (o & 0x7fffffffffffffffull) : ~o)); }

/* get_bbox_of_triangle bvh.c:1852-1869; scene bounds (calc_bbox_of_triangles :1872-1895) */
__global__ __launch_bounds__(256) void k_rb_boxes(uint32_t n, const double *__restrict__ tri64, RBox *__restrict__ E, uint32_t *__restrict__ idx,
                                                  uint32_t *__restrict__ seg, unsigned long long *__restrict__ sb)
{
    __shared__ double smin[4][3], smax[4][3];
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double *t = tri64 + 9 * (size_t)i;
        RBox b;
        for (int k = 0; k < 3; k++) {
            double l = t[k], h = t[k];
            l = (l < t[3 + k]) ? l : t[3 + k]; l = (l < t[6 + k]) ? l : t[6 + k];
            h = (h > t[3 + k]) ? h : t[3 + k]; h = (h > t[6 + k]) ? h : t[6 + k];
            b.lo[k] = l; b.hi[k] = h;
            lo[k] = (lo[k] < l) ? lo[k] : l; hi[k] = (hi[k] > h) ? hi[k] : h;
        }
        E[i] = b; idx[i] = i; seg[i] = 0u;
    }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off >= 1; off >>= 1) { lo[k] = fmin(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmax(hi[k], __shfl_xor(hi[k], off)); }
        if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6][k] = lo[k]; smax[threadIdx.x >> 6][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        double l = smin[0][k], h = smax[0][k];
        for (int w = 1; w < 4; w++) { l = fmin(l, smin[w][k]); h = fmax(h, smax[w][k]); }
        if (l <= h) { atomicMin(&sb[k], d2o(l)); atomicMax(&sb[3 + k], d2o(h)); }
    }
}

/* bbox_add_margin bvh.c:1697-1731 */
__device__ __forceinline__ void add_margin(double bmin[3], double bmax[3])
{
    double sc[3];
    for (int i = 0; i < 3; i++) { const double scale = bmax[i] - bmin[i]; sc[i] = (scale < RB_EPS) ? RB_EPS : RB_EPS * scale; }
    for (int i = 0; i < 3; i++) { bmin[i] -= sc[i]; bmax[i] += sc[i]; }
}

__device__ __forceinline__ void seg_inv(RSeg &s)
{
    for (int k = 0; k < 3; k++) { const double size = s.bmax[k] - s.bmin[k]; s.inv[k] = (size > RB_EPS) ? (double)RB_BINS / size : 0.0; }
}

/* the root: node 0 over all elements, the scene box with its margin (bvh.c:325-340) */
__global__ void k_rb_root(uint32_t n, const unsigned long long *__restrict__ sb, RSeg *__restrict__ segs, lh_refnode_t *__restrict__ nodes,
                          double *__restrict__ scene6)
{
    if (threadIdx.x || blockIdx.x) return;
    RSeg s; memset(&s, 0, sizeof(s));
    for (int k = 0; k < 3; k++) { s.bmin[k] = o2d(sb[k]); s.bmax[k] = o2d(sb[3 + k]); }
    add_margin(s.bmin, s.bmax);
    for (int k = 0; k < 3; k++) { scene6[k] = s.bmin[k]; scene6[3 + k] = s.bmax[k]; }
    s.il = 0; s.ir = n; s.node = 0; seg_inv(s);
    segs[0] = s;
    lh_refnode_t r; memset(&r, 0, sizeof(r));
    r.parent = -1; r.depth = 0; r.child[0] = r.child[1] = -1;
    if (n <= RB_LEAF) { r.is_leaf = 1; r.first = 0; r.count = n; }
    nodes[0] = r;
}

/* bin_triangle_edge bvh.c:1571-1692: hist[s][0][k][bin of box.min], hist[s][1][k][bin of box.max] */
__device__ __forceinline__ void bins_of(const RBox &b, const RSeg &s, uint32_t out[6])
{
    for (int k = 0; k < 3; k++) {
        const double qmin = (b.lo[k] - s.bmin[k]) * s.inv[k], qmax = (b.hi[k] - s.bmin[k]) * s.inv[k];
        uint32_t imin = (uint32_t)qmin, imax = (uint32_t)qmax;
        if (imin >= RB_BINS) imin = RB_BINS - 1;
        if (imax >= RB_BINS) imax = RB_BINS - 1;
        out[k] = (uint32_t)(k * RB_BINS) + imin; out[3 + k] = (uint32_t)((3 + k) * RB_BINS) + imax;
    }
}

/* Nodes are contiguous ranges of the element array.  LONG nodes (RB_LONG elements or more) are counted here: a workgroup
 * walks the long runs of its chunk, each in LDS and then 384 atomics (same-address atomics serialise at ~95 ns: the root's
 * 21 M elements would queue on 384 addresses otherwise).  Short nodes: k_rb_small, a wave each, no atomics in memory at all. */
#define RB_LONG 1024u
__global__ __launch_bounds__(256) void k_rb_bin(uint32_t n, const RBox *__restrict__ E, const uint32_t *__restrict__ seg, const RSeg *__restrict__ segs,
                                                uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[RB_HIST];
    __shared__ RSeg ss;
    __shared__ uint32_t s_next, any_long;
    const uint32_t base = blockIdx.x * RB_CHUNK, end = (base + RB_CHUNK < n) ? base + RB_CHUNK : n;
    if (threadIdx.x == 0) any_long = 0u;
    __syncthreads();
    bool saw_long = false;
    for (uint32_t i = base + threadIdx.x; i < end; i += 256) {
        const uint32_t s = seg[i];
        if (s != RB_NONE && segs[s].ir - segs[s].il >= RB_LONG) saw_long = true;
    }
    if (saw_long) any_long = 1u;
    __syncthreads();
    if (!any_long) return;
    uint32_t pos = base;
    while (pos < end) {
        const uint32_t s = seg[pos];
        if (s == RB_NONE) {
            if (threadIdx.x == 0) s_next = 0xffffffffu;
            __syncthreads();
            const uint32_t i = pos + threadIdx.x;
            if (i < end && seg[i] != RB_NONE) atomicMin(&s_next, i);
            __syncthreads();
            const uint32_t nx = s_next;
            __syncthreads();
            pos = (nx == 0xffffffffu) ? pos + 256u : nx;
            continue;
        }
        if (threadIdx.x == 0) ss = segs[s];
        for (uint32_t t = threadIdx.x; t < RB_HIST; t += 256) h[t] = 0u;
        __syncthreads();
        const uint32_t re = ss.ir < end ? ss.ir : end;
        if (ss.ir - ss.il >= RB_LONG) {
            for (uint32_t i = pos + threadIdx.x; i < re; i += 256) {
                uint32_t b[6]; bins_of(E[i], ss, b);
                for (int q = 0; q < 6; q++) atomicAdd(&h[b[q]], 1u);
            }
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < RB_HIST; t += 256) if (h[t]) atomicAdd(&hist[(size_t)s * RB_HIST + t], h[t]);
        }
        __syncthreads();
        pos = re;
    }
}

__device__ __forceinline__ double area_of(const double lo[3], const double hi[3])
{   /* calc_surface_area bvh.c:1190-1208 */
    double sa = (hi[0] - lo[0]) * (hi[1] - lo[1]) + (hi[1] - lo[1]) * (hi[2] - lo[2]) + (hi[2] - lo[2]) * (hi[0] - lo[0]);
    sa *= 2.0;
    return sa;
}

__device__ __forceinline__ double sah_cost(int ns1, double a1, int ns2, double a2, double s)
{   /* SAH bvh.c:1210-1228: double sum narrowed to float */
    const float Taabb = 0.2f, Ttri = 0.8f;
    const float T = 2.0f * Taabb + (a1 / s) * (double)ns1 * Ttri + (a2 / s) * (double)ns2 * Ttri;
    return T;
}

/* find_cut_from_bin bvh.c:1230-1326.  The reference keeps the first candidate, in (axis, bin) order, whose cost is below
 * everything before it: three threads take an axis each, the first-found minima are then compared in axis order */
__global__ __launch_bounds__(192) void k_rb_cut(uint32_t S, RSeg *__restrict__ segs, const uint32_t *__restrict__ hist)
{
    __shared__ double scost[192], spos[192];
    const uint32_t s = blockIdx.x * 64u + threadIdx.x / 3u; const int j = (int)(threadIdx.x % 3u);
    double best = RB_INF, best_pos = 0.0;
    const bool mine = s < S && segs[s].ir - segs[s].il >= RB_LONG;        /* short nodes: k_rb_small */
    if (mine) {
        const RSeg sg = segs[s];
        const uint32_t *bin = hist + (size_t)s * RB_HIST;
        const uint32_t n = sg.ir - sg.il;
        const double sa_total = area_of(sg.bmin, sg.bmax);
        const double bstep = (sg.bmax[j] - sg.bmin[j]) / (double)RB_BINS;
        uint64_t left = 0, right = n; double lmin[3], lmax[3], rmin[3], rmax[3];
        for (int k = 0; k < 3; k++) { lmin[k] = rmin[k] = sg.bmin[k]; lmax[k] = rmax[k] = sg.bmax[k]; }
        for (int b = 0; b < RB_BINS - 1; b++) {
            left += bin[j * RB_BINS + b]; right -= bin[(3 + j) * RB_BINS + b];
            const double pos = sg.bmin[j] + (b + 1) * bstep;
            lmax[j] = pos; rmin[j] = pos;
            const double cost = sah_cost((int)left, area_of(lmin, lmax), (int)right, area_of(rmin, rmax), sa_total);
            if (cost < best) { best = cost; best_pos = pos; }
        }
    }
    scost[threadIdx.x] = best; spos[threadIdx.x] = best_pos;
    __syncthreads();
    if (mine && j == 0) {
        double min_cost = RB_INF, cut_pos = 0.0; int cut_axis = 0;
        for (int a = 0; a < 3; a++) if (scost[threadIdx.x + a] < min_cost) { min_cost = scost[threadIdx.x + a]; cut_axis = a; cut_pos = spos[threadIdx.x + a]; }
        segs[s].axis = cut_axis; segs[s].pos = cut_pos;
    }
}

/* bins and cut of a SHORT node (fewer than RB_LONG elements), one wave per node: the 384 counters live in LDS, the 189
 * candidates are spread over the lanes (lane b: the cut after bin b on each axis; left / right counts by a prefix sum across
 * the lanes), and the reference's "first candidate below everything before it" is the minimum with ties to the lowest
 * (axis, bin) */
__global__ __launch_bounds__(256) void k_rb_small(uint32_t S, RSeg *__restrict__ segs, const RBox *__restrict__ E)
{
    __shared__ uint32_t hh[4][RB_HIST];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t s = blockIdx.x * 4u + (uint32_t)wv;
    uint32_t *h = hh[wv];
    RSeg sg; bool mine = false;
    if (s < S) { sg = segs[s]; mine = sg.ir - sg.il < RB_LONG; }
    for (int q = 0; q < 6; q++) h[q * 64 + lane] = 0u;
    __syncthreads();
    if (mine)
        for (uint32_t i = sg.il + (uint32_t)lane; i < sg.ir; i += 64) {
            uint32_t b[6]; bins_of(E[i], sg, b);
            for (int q = 0; q < 6; q++) atomicAdd(&h[b[q]], 1u);
        }
    __syncthreads();
    if (!mine) return;
    const uint32_t n = sg.ir - sg.il;
    const double sa_total = area_of(sg.bmin, sg.bmax);
    double best = RB_INF, best_pos = 0.0; int best_key = 0;
    for (int j = 0; j < 3; j++) {
        uint32_t c0 = h[j * 64 + lane], c1 = h[(3 + j) * 64 + lane];
        for (int off = 1; off < 64; off <<= 1) {                /* inclusive prefix sums across the lanes */
            const uint32_t u0 = (uint32_t)__shfl_up((int)c0, off), u1 = (uint32_t)__shfl_up((int)c1, off);
            if (lane >= off) { c0 += u0; c1 += u1; }
        }
        if (lane < RB_BINS - 1) {
            const uint64_t left = c0, right = (uint64_t)n - (uint64_t)c1;
            const double bstep = (sg.bmax[j] - sg.bmin[j]) / (double)RB_BINS;
            double lmin[3], lmax[3], rmin[3], rmax[3];
            for (int k = 0; k < 3; k++) { lmin[k] = rmin[k] = sg.bmin[k]; lmax[k] = rmax[k] = sg.bmax[k]; }
            const double pos = sg.bmin[j] + (lane + 1) * bstep;
            lmax[j] = pos; rmin[j] = pos;
            const double cost = sah_cost((int)left, area_of(lmin, lmax), (int)right, area_of(rmin, rmax), sa_total);
            if (cost < best) { best = cost; best_pos = pos; best_key = j * 64 + lane; }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double oc = __shfl_xor(best, off), op = __shfl_xor(best_pos, off); const int ok = __shfl_xor(best_key, off);
        if (oc < best || (oc == best && ok < best_key)) { best = oc; best_pos = op; best_key = ok; }
    }
    if (lane == 0) {
        const bool found = best < RB_INF;
        segs[s].axis = found ? best_key / 64 : 0; segs[s].pos = found ? best_pos : 0.0;
    }
}

/* partition bvh.c:1437-1478, first half: who goes left */
__global__ void k_rb_flag(uint32_t n, const RBox *__restrict__ E, const uint32_t *__restrict__ seg, const RSeg *__restrict__ segs, uint32_t *__restrict__ F)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t f = 0;
    if (i < n) { const uint32_t s = seg[i]; if (s != RB_NONE) f = E[i].hi[segs[s].axis] < segs[s].pos; }
    F[i] = f;
}

/* per node: its left count (the reference's repair of an empty side, bvh.c:1480-1486: halve), its two child nodes */
__global__ void k_rb_split(uint32_t S, RSeg *__restrict__ segs, const uint32_t *__restrict__ L, lh_refnode_t *__restrict__ nodes, uint32_t node_base,
                           int depth, unsigned long long *__restrict__ cb)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const uint32_t il = segs[s].il, ir = segs[s].ir, n = ir - il;
    uint32_t nl = L[ir] - L[il];
    if (nl == 0 || nl == n) nl = n / 2;
    segs[s].nl = nl;
    const int32_t me = segs[s].node;
    nodes[me].child[0] = (int32_t)(node_base + 2 * s); nodes[me].child[1] = (int32_t)(node_base + 2 * s + 1);
    nodes[me].axis = segs[s].axis; nodes[me].is_leaf = 0;
    for (int k = 0; k < 2; k++) {
        lh_refnode_t c; memset(&c, 0, sizeof(c));
        c.parent = me; c.depth = depth + 1; c.child[0] = c.child[1] = -1;
        nodes[node_base + 2 * s + k] = c;
        unsigned long long *o = cb + (size_t)(2 * s + k) * 6;
        for (int q = 0; q < 3; q++) { o[q] = ~0ull; o[3 + q] = 0ull; }
    }
}

/* partition, second half: lefts keep their order from the front, rights are written from the end backwards.  cid: the child
 * (2 s + 0 / 1) an element now belongs to */
__global__ void k_rb_scatter(uint32_t n, const RBox *__restrict__ E, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ seg,
                             const RSeg *__restrict__ segs, const uint32_t *__restrict__ L, RBox *__restrict__ E2, uint32_t *__restrict__ idx2,
                             uint32_t *__restrict__ cid)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = seg[i];
    if (s == RB_NONE) { E2[i] = E[i]; idx2[i] = idx[i]; cid[i] = RB_NONE; return; }
    const uint32_t il = segs[s].il, ir = segs[s].ir, nl = segs[s].nl;
    const uint32_t lefts_before = L[i] - L[il];
    const bool left = L[i + 1] != L[i];
    const uint32_t pos = left ? il + lefts_before : ir - 1u - ((i - il) - lefts_before);
    E2[pos] = E[i]; idx2[pos] = idx[i];
    cid[pos] = 2 * s + (pos < il + nl ? 0u : 1u);
}

/* an ordered-integer min / max that only goes to the atomic unit when it would change the value: after the first few
 * elements of a node almost none does, and same-address atomics serialise at ~95 ns each.  (A stale read can only be LARGER
 * than the current minimum / smaller than the current maximum: the test errs towards the atomic.) */
__device__ __forceinline__ void bound_min(unsigned long long *p, unsigned long long v) { if (v < *(volatile unsigned long long *)p) atomicMin(p, v); }
__device__ __forceinline__ void bound_max(unsigned long long *p, unsigned long long v) { if (v > *(volatile unsigned long long *)p) atomicMax(p, v); }

/* bounds of the children (calc_bbox_of_triangles bvh.c:1872-1895).  LONG children (RB_LONG elements or more): a workgroup
 * walks the long runs of its chunk, reduces each and updates the child's box with at most six atomics. */
__global__ __launch_bounds__(256) void k_rb_child_bounds(uint32_t n, const RBox *__restrict__ E, const uint32_t *__restrict__ cid, const RSeg *__restrict__ segs,
                                                         unsigned long long *__restrict__ cb)
{
    __shared__ double smin[4][3], smax[4][3];
    __shared__ uint32_t s_next, any_long;
    const uint32_t base = blockIdx.x * RB_CHUNK, end = (base + RB_CHUNK < n) ? base + RB_CHUNK : n;
    if (threadIdx.x == 0) any_long = 0u;
    __syncthreads();
    bool saw_long = false;
    for (uint32_t i = base + threadIdx.x; i < end; i += 256) {
        const uint32_t c = cid[i];
        if (c == RB_NONE) continue;
        const RSeg &sg = segs[c >> 1];
        if (((c & 1u) ? (sg.ir - sg.il) - sg.nl : sg.nl) >= RB_LONG) saw_long = true;
    }
    if (saw_long) any_long = 1u;
    __syncthreads();
    if (!any_long) return;
    uint32_t pos = base;
    while (pos < end) {
        const uint32_t c = cid[pos];
        if (c == RB_NONE) {
            if (threadIdx.x == 0) s_next = 0xffffffffu;
            __syncthreads();
            const uint32_t i = pos + threadIdx.x;
            if (i < end && cid[i] != RB_NONE) atomicMin(&s_next, i);
            __syncthreads();
            const uint32_t nx = s_next;
            __syncthreads();
            pos = (nx == 0xffffffffu) ? pos + 256u : nx;
            continue;
        }
        const RSeg &sg = segs[c >> 1];
        const uint32_t cbeg = (c & 1u) ? sg.il + sg.nl : sg.il, cend = (c & 1u) ? sg.ir : sg.il + sg.nl;
        const uint32_t re = cend < end ? cend : end;
        if (cend - cbeg >= RB_LONG) {
            unsigned long long *o = cb + (size_t)c * 6;
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = pos + threadIdx.x; i < re; i += 256) {
                const RBox b = E[i];
                for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], b.lo[k]); hi[k] = fmax(hi[k], b.hi[k]); }
            }
            for (int k = 0; k < 3; k++) {
                for (int off = 32; off >= 1; off >>= 1) { lo[k] = fmin(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmax(hi[k], __shfl_xor(hi[k], off)); }
                if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6][k] = lo[k]; smax[threadIdx.x >> 6][k] = hi[k]; }
            }
            __syncthreads();
            if (threadIdx.x < 3) {
                const int k = threadIdx.x;
                double l = smin[0][k], h = smax[0][k];
                for (int w = 1; w < 4; w++) { l = fmin(l, smin[w][k]); h = fmax(h, smax[w][k]); }
                bound_min(&o[k], d2o(l)); bound_max(&o[3 + k], d2o(h));
            }
            __syncthreads();
        }
        pos = re;
    }
}

/* short children: a wave each, the box written once */
__global__ __launch_bounds__(256) void k_rb_child_small(uint32_t S, const RSeg *__restrict__ segs, const RBox *__restrict__ E, unsigned long long *__restrict__ cb)
{
    const int lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (c >= 2 * S) return;
    const RSeg &sg = segs[c >> 1];
    const uint32_t cbeg = (c & 1u) ? sg.il + sg.nl : sg.il, cend = (c & 1u) ? sg.ir : sg.il + sg.nl;
    if (cend - cbeg >= RB_LONG) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = cbeg + (uint32_t)lane; i < cend; i += 64) {
        const RBox b = E[i];
        for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], b.lo[k]); hi[k] = fmax(hi[k], b.hi[k]); }
    }
    for (int k = 0; k < 3; k++)
        for (int off = 32; off >= 1; off >>= 1) { lo[k] = fmin(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmax(hi[k], __shfl_xor(hi[k], off)); }
    if (lane == 0) {
        unsigned long long *o = cb + (size_t)c * 6;
        for (int k = 0; k < 3; k++) { o[k] = d2o(lo[k]); o[3 + k] = d2o(hi[k]); }
    }
}

/* per child: its box with margin into the parent (bvh.c:1511-1548); a leaf (bvh.c:1345-1403) or a node of the next level */
__global__ void k_rb_children(uint32_t S, const RSeg *__restrict__ segs, const unsigned long long *__restrict__ cb, lh_refnode_t *__restrict__ nodes,
                              uint32_t node_base, uint32_t *__restrict__ A)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > 2 * S) return;
    if (c == 2 * S) { A[c] = 0u; return; }
    const uint32_t s = c >> 1, k = c & 1u;
    const uint32_t il = segs[s].il, ir = segs[s].ir, nl = segs[s].nl;
    const uint32_t first = k ? il + nl : il, count = k ? (ir - il) - nl : nl;
    double bmin[3], bmax[3];
    for (int q = 0; q < 3; q++) { bmin[q] = o2d(cb[(size_t)c * 6 + q]); bmax[q] = o2d(cb[(size_t)c * 6 + 3 + q]); }
    add_margin(bmin, bmax);
    lh_refnode_t &p = nodes[segs[s].node];
    for (int q = 0; q < 3; q++) { p.box[k][q] = bmin[q]; p.box[k][3 + q] = bmax[q]; }
    lh_refnode_t &me = nodes[node_base + c];
    if (count <= RB_LEAF) { me.is_leaf = 1; me.first = first; me.count = count; A[c] = 0u; }
    else A[c] = 1u;
}

__global__ void k_rb_next(uint32_t S, const RSeg *__restrict__ segs, const lh_refnode_t *__restrict__ nodes, uint32_t node_base,
                          const uint32_t *__restrict__ A, const uint32_t *__restrict__ P, RSeg *__restrict__ next)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * S || !A[c]) return;
    const uint32_t s = c >> 1, k = c & 1u;
    const uint32_t il = segs[s].il, ir = segs[s].ir, nl = segs[s].nl;
    RSeg o; memset(&o, 0, sizeof(o));
    o.il = k ? il + nl : il; o.ir = k ? ir : il + nl; o.node = (int32_t)(node_base + c);
    const lh_refnode_t &p = nodes[segs[s].node];
    for (int q = 0; q < 3; q++) { o.bmin[q] = p.box[k][q]; o.bmax[q] = p.box[k][3 + q]; }
    seg_inv(o);
    next[P[c]] = o;
}

/* every element: the node of the next level it is in, or -- its child became a leaf -- that leaf, for good */
__global__ void k_rb_relabel(uint32_t n, uint32_t *__restrict__ cid_to_seg, const uint32_t *__restrict__ A, const uint32_t *__restrict__ P,
                             uint32_t node_base, uint32_t *__restrict__ leaf_of)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cid_to_seg[i];
    if (c == RB_NONE) return;
    if (A[c]) cid_to_seg[i] = P[c];
    else { cid_to_seg[i] = RB_NONE; leaf_of[i] = node_base + c; }
}

/* gather_triangles bvh.c:1897-1917 = keep the element order; per primitive its leaf and its place in it */
__global__ void k_rb_finish(uint32_t n, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ leaf_of, const lh_refnode_t *__restrict__ nodes,
                            uint32_t *__restrict__ leaf_prims, uint32_t *__restrict__ leafpos)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = idx[i], leaf = leaf_of[i];
    leaf_prims[i] = p; leafpos[2 * (size_t)p] = leaf; leafpos[2 * (size_t)p + 1] = i - nodes[leaf].first;
}

__global__ void k_rb_lca(uint32_t nn, const lh_refnode_t *__restrict__ nodes, int *__restrict__ lca)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    lca[4 * (size_t)i] = nodes[i].parent; lca[4 * (size_t)i + 1] = nodes[i].depth; lca[4 * (size_t)i + 2] = nodes[i].axis; lca[4 * (size_t)i + 3] = nodes[i].child[0];
}

struct DBuf { void *p; size_t cap; };

#define RCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(err, errlen, "%s failed: %s", #x, hipGetErrorString(e_)); goto fail; } } while (0)

} /* namespace */

/* d_tri64: ntris x 9 doubles in primitive-id order on the current device.  On success the four arrays the kernels read
 * (lh_refnode_t nodes; leaf_prims: primitive ids in the reference's leaf order; lca: parent, depth, axis, child[0] per node;
 * leafpos: leaf node and place in it per primitive) are hipMalloc'ed here and owned by the caller; scene6 = the scene box with
 * margin.  Returns 0, or -1 with err filled (the caller then builds the tree on the host). */
extern "C" int lh_device_ref_build(uint32_t ntris, const double *d_tri64, void **d_nodes, uint32_t *nnodes, uint32_t *max_depth,
                                   void **d_leaf_prims, void **d_lca, void **d_leafpos, double scene6[6], void *stream, char *err, size_t errlen)
{
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n = ntris;
    RBox *E[2] = {NULL, NULL}; uint32_t *idx[2] = {NULL, NULL}, *seg[2] = {NULL, NULL}, *F = NULL, *leaf_of = NULL, *A = NULL, *P = NULL;
    unsigned long long *sb = NULL, *cb = NULL; double *d_scene = NULL;
    RSeg *segs[2] = {NULL, NULL}; uint32_t *hist = NULL; lh_refnode_t *nodes = NULL; void *tmp = NULL;
    uint32_t *leaf_prims = NULL, *leafpos = NULL; int *lca = NULL;
    size_t seg_cap = 0, hist_cap = 0, node_cap = 0, child_cap = 0, tmp_cap = 0;
    uint32_t S = 0, node_count = 1, level = 0;
    int cur = 0;
    const unsigned nbe = (n + 255) / 256;
    *d_nodes = NULL; *d_leaf_prims = NULL; *d_lca = NULL; *d_leafpos = NULL; *nnodes = 0; *max_depth = 0;
    if (n == 0) { snprintf(err, errlen, "empty scene"); return -1; }
    const bool timing = getenv("LH_BUILD_TIMING") != NULL;
    struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);

    for (int b = 0; b < 2; b++) {
        RCHK(hipMalloc((void **)&E[b], sizeof(RBox) * (size_t)n)); RCHK(hipMalloc((void **)&idx[b], sizeof(uint32_t) * (size_t)n));
        RCHK(hipMalloc((void **)&seg[b], sizeof(uint32_t) * (size_t)n));
    }
    RCHK(hipMalloc((void **)&F, sizeof(uint32_t) * ((size_t)n + 1))); RCHK(hipMalloc((void **)&leaf_of, sizeof(uint32_t) * (size_t)n));
    RCHK(hipMalloc((void **)&sb, sizeof(unsigned long long) * 6)); RCHK(hipMalloc((void **)&d_scene, sizeof(double) * 6));
    {
        const unsigned long long init[6] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull};
        RCHK(hipMemcpyAsync(sb, init, sizeof(init), hipMemcpyHostToDevice, s));
    }
    node_cap = (size_t)n / 4 + 64; seg_cap = 1024; child_cap = 2 * seg_cap + 1;
    RCHK(hipMalloc((void **)&nodes, sizeof(lh_refnode_t) * node_cap)); RCHK(hipMemsetAsync(nodes, 0, sizeof(lh_refnode_t) * node_cap, s));
    for (int b = 0; b < 2; b++) RCHK(hipMalloc((void **)&segs[b], sizeof(RSeg) * seg_cap));
    RCHK(hipMalloc((void **)&cb, sizeof(unsigned long long) * 6 * child_cap));
    RCHK(hipMalloc((void **)&A, sizeof(uint32_t) * child_cap)); RCHK(hipMalloc((void **)&P, sizeof(uint32_t) * child_cap));
    hist_cap = seg_cap; RCHK(hipMalloc((void **)&hist, sizeof(uint32_t) * RB_HIST * hist_cap));
    {
        size_t b1 = 0, b2 = 0;
        RCHK(hipcub::DeviceScan::ExclusiveSum(NULL, b1, F, F, (int)((size_t)n + 1), s));
        RCHK(hipcub::DeviceScan::ExclusiveSum(NULL, b2, A, P, (int)((size_t)n / 8 + 1024), s));
        tmp_cap = b1 > b2 ? b1 : b2; if (tmp_cap < 256) tmp_cap = 256;
        RCHK(hipMalloc(&tmp, tmp_cap));
    }
    hipLaunchKernelGGL(k_rb_boxes, dim3(nbe < 2048u ? nbe : 2048u), dim3(256), 0, s, n, d_tri64, E[0], idx[0], seg[0], sb);
    hipLaunchKernelGGL(k_rb_root, dim3(1), dim3(64), 0, s, n, (const unsigned long long *)sb, segs[0], nodes, d_scene);
    if (n <= RB_LEAF) { RCHK(hipMemsetAsync(leaf_of, 0, sizeof(uint32_t) * (size_t)n, s)); S = 0; } else S = 1;

    while (S > 0) {
        /* room for this level: 2 S child nodes, 2 S + 1 child records, S histograms, up to 2 S nodes of the next level */
        if ((size_t)node_count + 2 * (size_t)S > node_cap) {
            size_t nc = node_cap * 2; if (nc < (size_t)node_count + 2 * (size_t)S) nc = (size_t)node_count + 2 * (size_t)S;
            lh_refnode_t *nn = NULL;
            RCHK(hipMalloc((void **)&nn, sizeof(lh_refnode_t) * nc));
            if (hipMemsetAsync(nn, 0, sizeof(lh_refnode_t) * nc, s) != hipSuccess ||
                hipMemcpyAsync(nn, nodes, sizeof(lh_refnode_t) * node_count, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                (void)hipFree(nn); snprintf(err, errlen, "growing the node array failed"); goto fail;
            }
            (void)hipFree(nodes); nodes = nn; node_cap = nc;
        }
        if (2 * (size_t)S + 1 > child_cap) {
            RCHK(hipStreamSynchronize(s));
            (void)hipFree(cb); (void)hipFree(A); (void)hipFree(P); cb = NULL; A = P = NULL;
            child_cap = 4 * (size_t)S + 1;
            RCHK(hipMalloc((void **)&cb, sizeof(unsigned long long) * 6 * child_cap));
            RCHK(hipMalloc((void **)&A, sizeof(uint32_t) * child_cap)); RCHK(hipMalloc((void **)&P, sizeof(uint32_t) * child_cap));
            size_t b2 = 0;
            RCHK(hipcub::DeviceScan::ExclusiveSum(NULL, b2, A, P, (int)child_cap, s));
            if (b2 > tmp_cap) { (void)hipFree(tmp); tmp = NULL; tmp_cap = b2; RCHK(hipMalloc(&tmp, tmp_cap)); }
        }
        if ((size_t)S > hist_cap) {
            RCHK(hipStreamSynchronize(s));
            (void)hipFree(hist); hist = NULL; hist_cap = 2 * (size_t)S;
            RCHK(hipMalloc((void **)&hist, sizeof(uint32_t) * RB_HIST * hist_cap));
        }
        if (2 * (size_t)S > seg_cap) {
            /* the other list (the next level's) may need 2 S entries; the current one is in use: grow the other only */
            RCHK(hipStreamSynchronize(s));
            RSeg *ns = NULL;
            const size_t nc = 4 * (size_t)S;
            RCHK(hipMalloc((void **)&ns, sizeof(RSeg) * nc));
            (void)hipFree(segs[cur ^ 1]); segs[cur ^ 1] = ns;
            /* and the current one when it becomes "the other" next time: remember the capacity of the smaller */
            RSeg *ns2 = NULL;
            RCHK(hipMalloc((void **)&ns2, sizeof(RSeg) * nc));
            if (hipMemcpyAsync(ns2, segs[cur], sizeof(RSeg) * S, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                (void)hipFree(ns2); snprintf(err, errlen, "growing the node lists failed"); goto fail;
            }
            (void)hipFree(segs[cur]); segs[cur] = ns2; seg_cap = nc;
        }
        const unsigned nbs = (S + 127) / 128, nbc = (2 * S + 1 + 127) / 128;
        RCHK(hipMemsetAsync(hist, 0, sizeof(uint32_t) * RB_HIST * (size_t)S, s));
        hipLaunchKernelGGL(k_rb_bin, dim3((n + RB_CHUNK - 1) / RB_CHUNK), dim3(256), 0, s, n, (const RBox *)E[cur], (const uint32_t *)seg[cur], (const RSeg *)segs[cur], hist);
        hipLaunchKernelGGL(k_rb_cut, dim3((S + 63) / 64), dim3(192), 0, s, S, segs[cur], (const uint32_t *)hist);
        hipLaunchKernelGGL(k_rb_small, dim3((S + 3) / 4), dim3(256), 0, s, S, segs[cur], (const RBox *)E[cur]);
        hipLaunchKernelGGL(k_rb_flag, dim3((n + 1 + 255) / 256), dim3(256), 0, s, n, (const RBox *)E[cur], (const uint32_t *)seg[cur], (const RSeg *)segs[cur], F);
        { size_t b = tmp_cap; RCHK(hipcub::DeviceScan::ExclusiveSum(tmp, b, F, F, (int)((size_t)n + 1), s)); }
        hipLaunchKernelGGL(k_rb_split, dim3(nbs), dim3(128), 0, s, S, segs[cur], (const uint32_t *)F, nodes, node_count, (int)level, cb);
        hipLaunchKernelGGL(k_rb_scatter, dim3(nbe), dim3(256), 0, s, n, (const RBox *)E[cur], (const uint32_t *)idx[cur], (const uint32_t *)seg[cur],
                           (const RSeg *)segs[cur], (const uint32_t *)F, E[cur ^ 1], idx[cur ^ 1], seg[cur ^ 1]);
        hipLaunchKernelGGL(k_rb_child_bounds, dim3((n + RB_CHUNK - 1) / RB_CHUNK), dim3(256), 0, s, n, (const RBox *)E[cur ^ 1], (const uint32_t *)seg[cur ^ 1], (const RSeg *)segs[cur], cb);
        hipLaunchKernelGGL(k_rb_child_small, dim3((2 * S + 3) / 4), dim3(256), 0, s, S, (const RSeg *)segs[cur], (const RBox *)E[cur ^ 1], cb);
        hipLaunchKernelGGL(k_rb_children, dim3(nbc), dim3(128), 0, s, S, (const RSeg *)segs[cur], (const unsigned long long *)cb, nodes, node_count, A);
        { size_t b = tmp_cap; RCHK(hipcub::DeviceScan::ExclusiveSum(tmp, b, A, P, (int)(2 * (size_t)S + 1), s)); }
        hipLaunchKernelGGL(k_rb_next, dim3(nbc), dim3(128), 0, s, S, (const RSeg *)segs[cur], (const lh_refnode_t *)nodes, node_count, (const uint32_t *)A, (const uint32_t *)P, segs[cur ^ 1]);
        hipLaunchKernelGGL(k_rb_relabel, dim3(nbe), dim3(256), 0, s, n, seg[cur ^ 1], (const uint32_t *)A, (const uint32_t *)P, node_count, leaf_of);
        uint32_t nextS = 0;
        RCHK(hipMemcpyAsync(&nextS, P + 2 * (size_t)S, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        RCHK(hipStreamSynchronize(s));
        node_count += 2 * S; S = nextS; cur ^= 1; level++;
        if (level > 4096) { snprintf(err, errlen, "the reference-order tree is more than 4096 levels deep"); goto fail; }
    }
    RCHK(hipMalloc((void **)&leaf_prims, sizeof(uint32_t) * (size_t)n)); RCHK(hipMalloc((void **)&leafpos, sizeof(uint32_t) * 2 * (size_t)n));
    RCHK(hipMalloc((void **)&lca, sizeof(int) * 4 * (size_t)node_count));
    hipLaunchKernelGGL(k_rb_finish, dim3(nbe), dim3(256), 0, s, n, (const uint32_t *)idx[cur], (const uint32_t *)leaf_of, (const lh_refnode_t *)nodes, leaf_prims, leafpos);
    hipLaunchKernelGGL(k_rb_lca, dim3((node_count + 255) / 256), dim3(256), 0, s, node_count, (const lh_refnode_t *)nodes, lca);
    RCHK(hipMemcpyAsync(scene6, d_scene, sizeof(double) * 6, hipMemcpyDeviceToHost, s));
    RCHK(hipGetLastError());
    RCHK(hipStreamSynchronize(s));
    if (node_cap > (size_t)node_count + (size_t)node_count / 8) {          /* give the slack back */
        lh_refnode_t *nn = NULL;
        if (hipMalloc((void **)&nn, sizeof(lh_refnode_t) * node_count) == hipSuccess) {
            if (hipMemcpy(nn, nodes, sizeof(lh_refnode_t) * node_count, hipMemcpyDeviceToDevice) == hipSuccess) { (void)hipFree(nodes); nodes = nn; }
            else (void)hipFree(nn);
        }
    }
    *d_nodes = nodes; *nnodes = node_count; *max_depth = level; *d_leaf_prims = leaf_prims; *d_lca = lca; *d_leafpos = leafpos;
    for (int b = 0; b < 2; b++) { (void)hipFree(E[b]); (void)hipFree(idx[b]); (void)hipFree(seg[b]); (void)hipFree(segs[b]); }
    (void)hipFree(F); (void)hipFree(leaf_of); (void)hipFree(sb); (void)hipFree(d_scene); (void)hipFree(cb); (void)hipFree(A); (void)hipFree(P); (void)hipFree(hist); (void)hipFree(tmp);
    if (timing) {
        struct timespec ts1; clock_gettime(CLOCK_MONOTONIC, &ts1);
        fprintf(stderr, "[lucille_hip] device build: lucille's own tree            %8.2f ms (%u nodes, %u levels)\n",
                ((ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec)) * 1e3, node_count, level);
    }
    return 0;
fail:
    (void)hipStreamSynchronize(s);
    for (int b = 0; b < 2; b++) { if (E[b]) (void)hipFree(E[b]); if (idx[b]) (void)hipFree(idx[b]); if (seg[b]) (void)hipFree(seg[b]); if (segs[b]) (void)hipFree(segs[b]); }
    if (F) (void)hipFree(F); if (leaf_of) (void)hipFree(leaf_of); if (sb) (void)hipFree(sb); if (d_scene) (void)hipFree(d_scene);
    if (cb) (void)hipFree(cb); if (A) (void)hipFree(A); if (P) (void)hipFree(P); if (hist) (void)hipFree(hist); if (tmp) (void)hipFree(tmp);
    if (nodes) (void)hipFree(nodes); if (leaf_prims) (void)hipFree(leaf_prims); if (leafpos) (void)hipFree(leafpos); if (lca) (void)hipFree(lca);
    return -1;
}
