/*
 * lh_build.hip -- the traversal tree built ON THE DEVICE (LBVH), for scenes that are re-committed every frame.
 *
 * lucille rebuilds its accelerator in every ri_scene_setup (src/render/scene.c:84-98 -> ri_bvh_build,
 * src/render/bvh.c:276-379): with the host builder of lh_bvh.c that is 5.9 s in front of a 93 ms frame on the
 * BASELINE config 5 scene.  Hit records do not depend on the tree (any conservative BVH + the bit-exact
 * fp64 triangle test reproduces the reference), so the traversal tree may be built by whatever is fastest:
 *
 *   1. per-primitive fp32-outward boxes and centroids from the fp64 triangles; scene box (atomic min / max);
 *   2. 63-bit Morton codes of the centroids, radix-sorted with the primitive ids (hipcub);
 *   3. the binary radix tree over the sorted codes (Karras, "Maximizing Parallelism in the Construction of BVHs,
 *      Octrees, and k-d Trees", HPG 2012 -- the published algorithm, not code), ties between equal codes broken by
 *      position; boxes bottom-up with one atomic counter per inner node;
 *   4. the same 64-byte 4-wide 16-bit-grid nodes the host builder emits (lh_q4node_t): subtrees of <= 4 primitives become
 *      leaves (a contiguous range of the sorted order), the rest is collapsed level by level -- a 4-wide node takes its
 *      binary node's two children and opens the one with the largest area until it has four -- children of one node
 *      allocated adjacently; boxes quantised outward on the scene grid exactly as lh_bvh.c does (lo down, hi up, verified);
 *   5. the 48-byte triangle records (lh_tri32_t) in sorted = leaf order.
 *
 * Primitive numbering is the caller's (create_triangle_list order, bvh.c:1736-1826): ids ride along as payload.
 * The tree is shallower in quality than the binned-SAH host tree (more node visits per ray) and about three orders of
 * magnitude quicker to build; lh_api.hip chooses (LH_BUILD=device, lh_accel_commit's build_threads == LH_BUILD_ON_DEVICE).
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lh_bvh.h"
#include "lh_device.h"

namespace {

struct BNode {              /* inner node of the binary radix tree */
    int left, right;        /* >= 0: inner node; < 0: ~position of a leaf in the sorted order */
    int parent;
    uint32_t first, last;   /* range of sorted positions it covers */
    float lo[3], hi[3];
    float cost;             /* SAH cost of the cheapest way to finish this subtree (k_refit) */
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

__global__ void k_prim_boxes(uint32_t n, const double *__restrict__ tri64, float *__restrict__ plo, float *__restrict__ phi,
                             uint32_t *__restrict__ scene /* 6 ordered uints: min xyz, max xyz */, int *__restrict__ bad)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double *t = tri64 + 9 * (size_t)p;
    for (int k = 0; k < 3; k++) {
        const double a = t[k], b = t[3 + k], c = t[6 + k];
        if (!(fabs(a) <= 1.0e30) || !(fabs(b) <= 1.0e30) || !(fabs(c) <= 1.0e30)) atomicExch(bad, 1);
        const float lo = f_down(fmin(a, fmin(b, c))), hi = f_up(fmax(a, fmax(b, c)));
        plo[3 * (size_t)p + k] = lo; phi[3 * (size_t)p + k] = hi;
        atomicMin(&scene[k], f2o(lo)); atomicMax(&scene[3 + k], f2o(hi));
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
    uint64_t code = 0;
    for (int k = 0; k < 3; k++) {
        const float smin = o2f(scene[k]), smax = o2f(scene[3 + k]);
        const double ext = (double)smax - (double)smin;
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

__global__ void k_refit(int n, const uint32_t *__restrict__ sorted, const float *__restrict__ plo, const float *__restrict__ phi,
                        BNode *__restrict__ nodes, const int *__restrict__ leaf_parent, uint32_t *__restrict__ visits, int leaf_max)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cur = leaf_parent[i];
    while (cur >= 0) {
        if (atomicAdd(&visits[cur], 1u) == 0u) return;          /* the second arrival merges */
        __threadfence();
        BNode &nd = nodes[cur];
        float lo[3], hi[3];
        for (int side = 0; side < 2; side++) {
            const int c = side ? nd.right : nd.left;
            const float *cl, *ch;
            if (c < 0) { const uint32_t p = sorted[~c]; cl = plo + 3 * (size_t)p; ch = phi + 3 * (size_t)p; }
            else { cl = nodes[c].lo; ch = nodes[c].hi; }
            for (int k = 0; k < 3; k++) {
                const float a = ((volatile const float *)cl)[k], b = ((volatile const float *)ch)[k];
                if (side == 0) { lo[k] = a; hi[k] = b; } else { lo[k] = fminf(lo[k], a); hi[k] = fmaxf(hi[k], b); }
            }
        }
        for (int k = 0; k < 3; k++) { nd.lo[k] = lo[k]; nd.hi[k] = hi[k]; }
        /* bottom-up SAH: this subtree as inner node + its children's best, or -- up to leaf_max primitives -- as one leaf.  A soup
         * of unrelated triangles keeps one triangle per leaf (its boxes barely shrink towards the leaves: the leaf's area is the
         * node's), a tessellated surface merges neighbours into leaves of up to four */
        {
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            const float area = dx * dy + dy * dz + dz * dx;
            float csum = 0.0f;
            for (int side = 0; side < 2; side++) {
                const int c = side ? nd.right : nd.left;
                if (c < 0) {
                    const uint32_t p = sorted[~c];
                    const float ex = phi[3 * (size_t)p] - plo[3 * (size_t)p], ey = phi[3 * (size_t)p + 1] - plo[3 * (size_t)p + 1], ez = phi[3 * (size_t)p + 2] - plo[3 * (size_t)p + 2];
                    csum += LH_SAH_CT * (ex * ey + ey * ez + ez * ex);
                } else csum += ((volatile const float *)&nodes[c].cost)[0];
            }
            const uint32_t cnt = nd.last - nd.first + 1u;
            const float as_node = LH_SAH_CI * area + csum, as_leaf = LH_SAH_CT * area * (float)cnt;
            const bool leaf = cnt <= (uint32_t)leaf_max && as_leaf <= as_node;
            nd.cost = leaf ? as_leaf : as_node; nd.leaf = leaf ? 1 : 0;
        }
        __threadfence();
        cur = nd.parent;
    }
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
        c.node = b.leaf ? -1 : ref;            /* k_refit's SAH decision: one leaf of its <= leaf_max primitives, or an inner node */
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

/* one level of the 4-wide collapse: work item = (binary node, index of its 4-wide node) */
__global__ void k_collapse_level(uint32_t nwork, const uint2 *__restrict__ work_in, uint2 *__restrict__ work_out,
                                 uint32_t *__restrict__ counters /* [0] next 4-wide index, [1] work_out count */,
                                 const BNode *__restrict__ nodes, const uint32_t *__restrict__ sorted,
                                 const float *__restrict__ plo, const float *__restrict__ phi,
                                 const float3 glo, const float3 gstep, lh_q4node_t *__restrict__ q4, int leaf_max)
{
    const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= nwork) return;
    const int b = (int)work_in[wi].x; const uint32_t k4 = work_in[wi].y;
    Child ch[4]; int n = 2;
    child_of(nodes, sorted, plo, phi, nodes[b].left, ch[0], leaf_max);
    child_of(nodes, sorted, plo, phi, nodes[b].right, ch[1], leaf_max);
    while (n < 4) {
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
    uint32_t base4 = 0, basew = 0;
    if (ninner) { base4 = atomicAdd(&counters[0], (uint32_t)ninner); basew = atomicAdd(&counters[1], (uint32_t)ninner); }
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

/* d_tri64: ntris x 9 doubles (primitive-id order) on the current device.  On success *d_q4nodes (capacity ntris records,
 * *nq4 used) and *d_tri32 (ntris + 2 records) are hipMalloc'ed here and owned by the caller; bmin / bmax / grid as lh_bvh_t.
 * Returns 0, -1 (err filled), or -2 for a NaN / infinite / > 1e30 coordinate. */
extern "C" int lh_device_build(uint32_t ntris, const double *d_tri64, void **d_q4nodes, uint32_t *nq4, uint32_t *q4_depth,
                               void **d_tri32, float bmin[3], float bmax[3], float grid_lo[3], float grid_step[3],
                               void *stream, char *err, size_t errlen)
{
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n = ntris;
    float *plo = NULL, *phi = NULL; uint32_t *scene = NULL, *val_in = NULL, *sorted = NULL, *visits = NULL, *counters = NULL;
    uint64_t *key_in = NULL, *key = NULL; int *leaf_parent = NULL, *bad = NULL; BNode *nodes = NULL; void *tmp = NULL; size_t tmp_bytes = 0;
    uint2 *work[2] = {NULL, NULL}; lh_q4node_t *q4 = NULL; lh_tri32_t *t32 = NULL;
    const unsigned nb = (n + 255) / 256;
    uint32_t h_scene[6], h_cnt[2], level = 0, nwork = 0, nq = 1;
    const uint32_t init_scene[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    int h_bad = 0, leaf_max = LH_MAX_LEAF_TRIS;     /* leaves of up to 4 triangles WHERE THE SAH SAYS SO (k_refit): forced 4-triangle leaves cost S-soup-1M 37 % (tools/leaf_probe.py), the SAH keeps that soup at one per leaf */
    { const char *e = getenv("LH_DEVICE_LEAF"); if (e && atoi(e) >= 1 && atoi(e) <= LH_MAX_LEAF_TRIS) leaf_max = atoi(e); }
    *d_q4nodes = NULL; *d_tri32 = NULL; *nq4 = 0; *q4_depth = 0;
    if (n == 0) return 0;

    BCHK(hipMalloc((void **)&plo, sizeof(float) * 3 * (size_t)n)); BCHK(hipMalloc((void **)&phi, sizeof(float) * 3 * (size_t)n));
    BCHK(hipMalloc((void **)&scene, sizeof(uint32_t) * 8)); BCHK(hipMalloc((void **)&bad, sizeof(int)));
    BCHK(hipMemcpyAsync(scene, init_scene, sizeof(init_scene), hipMemcpyHostToDevice, s));
    BCHK(hipMemsetAsync(bad, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_prim_boxes, dim3(nb), dim3(256), 0, s, n, d_tri64, plo, phi, scene, bad);
    BCHK(hipMemcpyAsync(h_scene, scene, sizeof(h_scene), hipMemcpyDeviceToHost, s));
    BCHK(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, s));
    BCHK(hipStreamSynchronize(s));
    if (h_bad) { dfree(plo); dfree(phi); dfree(scene); dfree(bad); return -2; }
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
    BCHK(hipMalloc((void **)&q4, sizeof(lh_q4node_t) * (size_t)(n > 1 ? n : 1)));
    BCHK(hipMalloc((void **)&t32, sizeof(lh_tri32_t) * ((size_t)n + 2)));
    BCHK(hipMalloc((void **)&sorted, sizeof(uint32_t) * (size_t)n));
    {
        const float3 glo = make_float3(grid_lo[0], grid_lo[1], grid_lo[2]), gst = make_float3(grid_step[0], grid_step[1], grid_step[2]);
        if (n <= LH_MAX_LEAF_TRIS) {
            /* identity order, one leaf (the only place a device-built leaf holds more than leaf_max triangles) */
            uint32_t ids[LH_MAX_LEAF_TRIS]; for (uint32_t i = 0; i < n; i++) ids[i] = i;
            BCHK(hipMemcpyAsync(sorted, ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_single_leaf, dim3(1), dim3(64), 0, s, n, scene, glo, gst, q4);
            nq = 1; level = 1;
        } else {
            BCHK(hipMalloc((void **)&key_in, sizeof(uint64_t) * (size_t)n)); BCHK(hipMalloc((void **)&key, sizeof(uint64_t) * (size_t)n));
            BCHK(hipMalloc((void **)&val_in, sizeof(uint32_t) * (size_t)n));
            hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, s, n, plo, phi, scene, key_in, val_in);
            BCHK(hipcub::DeviceRadixSort::SortPairs(NULL, tmp_bytes, key_in, key, val_in, sorted, (int)n, 0, 63, s));
            BCHK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
            BCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_in, key, val_in, sorted, (int)n, 0, 63, s));
            BCHK(hipMalloc((void **)&nodes, sizeof(BNode) * (size_t)(n - 1)));
            BCHK(hipMalloc((void **)&leaf_parent, sizeof(int) * (size_t)n));
            BCHK(hipMalloc((void **)&visits, sizeof(uint32_t) * (size_t)(n - 1)));
            BCHK(hipMemsetAsync(visits, 0, sizeof(uint32_t) * (size_t)(n - 1), s));
            hipLaunchKernelGGL(k_radix_tree, dim3((n - 1 + 255) / 256), dim3(256), 0, s, (int)n, key, nodes, leaf_parent);
            hipLaunchKernelGGL(k_refit, dim3(nb), dim3(256), 0, s, (int)n, sorted, plo, phi, nodes, leaf_parent, visits, leaf_max);
            /* level-by-level collapse; every level's children are allocated adjacently */
            BCHK(hipMalloc((void **)&work[0], sizeof(uint2) * (size_t)n)); BCHK(hipMalloc((void **)&work[1], sizeof(uint2) * (size_t)n));
            BCHK(hipMalloc((void **)&counters, sizeof(uint32_t) * 2));
            {
                const uint2 root = make_uint2(0u, 0u);
                BCHK(hipMemcpyAsync(work[0], &root, sizeof(root), hipMemcpyHostToDevice, s));
            }
            nwork = 1; nq = 1;
            while (nwork > 0) {
                h_cnt[0] = nq; h_cnt[1] = 0;
                BCHK(hipMemcpyAsync(counters, h_cnt, sizeof(h_cnt), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_collapse_level, dim3((nwork + 127) / 128), dim3(128), 0, s, nwork, (const uint2 *)work[level & 1], work[(level + 1) & 1],
                                   counters, (const BNode *)nodes, (const uint32_t *)sorted, (const float *)plo, (const float *)phi, glo, gst, q4, leaf_max);
                BCHK(hipMemcpyAsync(h_cnt, counters, sizeof(h_cnt), hipMemcpyDeviceToHost, s));
                BCHK(hipStreamSynchronize(s));
                nq = h_cnt[0]; nwork = h_cnt[1]; level++;
                if (level > 200) { snprintf(err, errlen, "device build: runaway collapse"); goto fail; }
            }
        }
    }
    hipLaunchKernelGGL(k_tri32, dim3(nb), dim3(256), 0, s, n, (const uint32_t *)sorted, d_tri64, t32);
    BCHK(hipGetLastError());
    BCHK(hipStreamSynchronize(s));
    *d_q4nodes = q4; *d_tri32 = t32; *nq4 = nq; *q4_depth = level;
    dfree(plo); dfree(phi); dfree(scene); dfree(bad); dfree(key_in); dfree(key); dfree(val_in); dfree(sorted);
    dfree(nodes); dfree(leaf_parent); dfree(visits); dfree(tmp); dfree(work[0]); dfree(work[1]); dfree(counters);
    return 0;
fail:
    dfree(plo); dfree(phi); dfree(scene); dfree(bad); dfree(key_in); dfree(key); dfree(val_in); dfree(sorted);
    dfree(nodes); dfree(leaf_parent); dfree(visits); dfree(tmp); dfree(work[0]); dfree(work[1]); dfree(counters);
    dfree(q4); dfree(t32);
    return -1;
}
