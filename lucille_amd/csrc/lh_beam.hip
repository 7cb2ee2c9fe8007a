/*
 * lh_beam.hip -- beam (frustum) visibility on the GPU: SURVEY.md 8a row a14.
 *
 * Reference (CPU, fp64):
 *   ri_beam_set                              src/render/beam.c:331-465
 *   ri_bvh_intersect_beam_visibility         src/render/bvh.c:612-667
 *   test_beam_aabb / get_n_point             src/render/bvh.c:1997-2089
 *   test_beam_node                           src/render/bvh.c:2097-2126
 *   test_beam_triangle                       src/render/bvh.c:2139-2281
 *   leaf / traversal                         src/render/bvh.c:2435-2542, 2648-2746
 *
 * The answer (0 miss / 1 hit completely / 2 hit partially) is the class of the FIRST
 * non-missing triangle in the reference's own traversal order and depends on which leaves
 * its plane tests let through, so this kernel walks the reference-order tree (lh_refbvh.c)
 * with the reference's fp64 arithmetic, operation order and comparisons (no FMA
 * contraction) -- but not its schedule: a beam is walked by 16 lanes, which test a node's
 * eight planes or a leaf's up to 16 triangles at once and restore the reference's order with
 * a ballot (k_beam_visibility).  It is a low-volume query (the reference's only caller issues
 * (W/64)^2 root beams), bounded by dependent 128-byte gathers like the ray kernel; there is
 * nothing for MFMA here.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lh_device.h"
#include "lh_refbvh.h"

namespace {

#define LH_NC _Pragma("clang fp contract(off)")
constexpr double kEps = 1.0e-14;    /* RI_EPS */
constexpr double kTInf = 1.0e38;    /* RI_INFINITY */

struct Beam {
    double org[3], dir[4][3], normal[4][3];
    int dominant_axis, dirsign[3];
};

__device__ __forceinline__ void cross3(double d[3], const double a[3], const double b[3])
{
    LH_NC
    d[0] = a[1] * b[2] - a[2] * b[1]; d[1] = a[2] * b[0] - a[0] * b[2]; d[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double a[3], const double b[3])
{
    LH_NC
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* ri_beam_set (beam.c:331-465), including the maxval re-assignment at :387-390 */
__device__ int beam_set(Beam &b, const double *org, const double *dir /* 4x3 */)
{
    LH_NC
    for (int i = 0; i < 3; i++) {
        int zeros = 0, mask = 0;
        for (int j = 0; j < 4; j++) {
            if (fabs(dir[3 * j + i]) < kEps) zeros++;
            else mask += (dir[3 * j + i] < 0.0) ? 1 : -1;
        }
        if ((mask != -(4 - zeros)) && (mask != (4 - zeros))) return -1;
    }
    for (int i = 0; i < 3; i++) b.org[i] = org[i];
    double maxval = fabs(dir[0]); int dom = 0;
    if (maxval < fabs(dir[1])) { maxval = fabs(dir[0]); dom = 1; }
    if (maxval < fabs(dir[2])) { maxval = fabs(dir[2]); dom = 2; }
    b.dominant_axis = dom;
    for (int i = 0; i < 3; i++) b.dirsign[i] = (dir[i] < 0.0) ? 1 : 0;
    double normal[3] = {0.0, 0.0, 0.0};
    normal[dom] = 1.0;
    if (b.dirsign[dom]) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
    for (int i = 0; i < 4; i++) {
        const double d3[3] = {dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]};
        const double t = dot3(d3, normal);
        const double k = (fabs(t) > kEps) ? 1024.0 / t : 1.0;
        b.dir[i][0] = k * d3[0]; b.dir[i][1] = k * d3[1]; b.dir[i][2] = k * d3[2];
    }
    cross3(b.normal[0], b.dir[1], b.dir[0]);
    cross3(b.normal[1], b.dir[2], b.dir[1]);
    cross3(b.normal[2], b.dir[3], b.dir[2]);
    cross3(b.normal[3], b.dir[0], b.dir[3]);
    return 0;
}

/* test_beam_aabb (bvh.c:2053-2089): 1 = the box may be hit.  The "cull by t" block at the top of the reference function
 * (bvh.c:2065-2074, beam->t_max against the box's near side) sits inside `#if 0`: it is not part of the compiled reference, and
 * ri_beam_set leaves t_max = RI_INFINITY (beam.c:344), so there is nothing to restate -- the plane test below is the whole
 * function.  (t_max does bound the triangle test, bvh.c:2194: that is kTInf in beam_triangle.) */
__device__ __forceinline__ int beam_aabb(const double *box, const Beam &b)
{
    LH_NC
    for (int i = 0; i < 4; i++) {
        double no[3];
        for (int k = 0; k < 3; k++) {
            const double np = (b.normal[i][k] > 0.0) ? box[k] : box[3 + k];
            no[k] = np - b.org[k];
        }
        if (dot3(no, b.normal[i]) > 0.0) return 0;
    }
    return 1;
}

/* test_beam_triangle (bvh.c:2139-2281) */
__device__ int beam_triangle(const double *tv, const Beam &b)
{
    LH_NC
    double u[4], v[4], t[4], e1[3], e2[3];
    int mask = 0;
    for (int i = 0; i < 3; i++) { e1[i] = tv[3 + i] - tv[i]; e2[i] = tv[6 + i] - tv[i]; }
    for (int i = 0; i < 4; i++) {
        double p[3], q[3], s[3];
        cross3(p, b.dir[i], e2);
        const double a = dot3(e1, p);
        const double inva = (fabs(a) > kEps) ? 1.0 / a : 0.0;
        s[0] = b.org[0] - tv[0]; s[1] = b.org[1] - tv[1]; s[2] = b.org[2] - tv[2];
        cross3(q, s, e1);
        u[i] = dot3(s, p) * inva; v[i] = dot3(q, b.dir[i]) * inva; t[i] = dot3(e2, q) * inva;
        if ((u[i] < 0.0) || (u[i] > 1.0)) continue;
        if ((v[i] < 0.0) || ((u[i] + v[i]) > 1.0)) continue;
        if ((t[i] < 0.0) || (t[i] > kTInf)) continue;
        mask |= (1 << i);
    }
    if (mask == 0) {
        int cnt = 0;
        for (int i = 0; i < 4; i++) if (t[i] < 0.0) cnt++;
        if (cnt == 4) return 0;
        cnt = 0; for (int i = 0; i < 4; i++) if (u[i] < 0.0) cnt++;
        if ((cnt != 0) && (cnt != 4)) return 2;
        cnt = 0; for (int i = 0; i < 4; i++) if (u[i] > 1.0) cnt++;
        if ((cnt != 0) && (cnt != 4)) return 2;
        cnt = 0; for (int i = 0; i < 4; i++) if (v[i] < 0.0) cnt++;
        if ((cnt != 0) && (cnt != 4)) return 2;
        cnt = 0; for (int i = 0; i < 4; i++) if ((u[i] + v[i]) >= 1.0) cnt++;
        if ((cnt != 0) && (cnt != 4)) return 2;
        return 0;
    }
    return (mask == 0xf) ? 1 : 2;
}

/* One beam per group of 16 lanes (four beams per wave).  The reference walks one beam at a time and tests a leaf's triangles
 * one after the other until one is not missed (bvh.c:2435-2542); here the SAME tests -- beam_aabb's four plane tests per child
 * box, beam_triangle's four corner-ray tests per triangle, fp64, no contraction -- are spread over the group's lanes and the
 * reference's order is put back with a ballot:
 *   inner node: lane 4 c + i tests plane i of child c; a child is missed when one of its four lanes says so;
 *   leaf:       lane q tests triangle q (leaves hold <= 16; longer ones in rounds of 16); the answer is the class of the
 *               lowest lane that did not miss = the first non-missing triangle in the reference's order.
 * Control flow is uniform inside a group (every lane holds the beam and computes the same next node); the stack of node
 * indices -- BVH_MAXDEPTH + 1 entries (bvh.c:80,124-129) -- is one LDS column per group. */
__global__ __launch_bounds__(64) void k_beam_visibility(lh_dev_scene_t sc, size_t n, const double *__restrict__ org,
                                                        const double *__restrict__ dirs, int32_t *__restrict__ result)
{
    LH_NC
    __shared__ int stack[4][104];
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
    const size_t r = (size_t)blockIdx.x * 4 + (size_t)g;
    const bool live = r < n;
    Beam b;
    int ret = 0;
    bool walking = false;
    if (live) {
        if (beam_set(b, org + 3 * r, dirs + 12 * r) != 0) ret = -1;
        else if (!sc.ref_empty) {
            const double sb[6] = {sc.ref_bmin[0], sc.ref_bmin[1], sc.ref_bmin[2], sc.ref_bmax[0], sc.ref_bmax[1], sc.ref_bmax[2]};
            walking = beam_aabb(sb, b) != 0;
        }
    }
    const lh_refnode_t *nodes = (const lh_refnode_t *)sc.ref_nodes;
    const uint32_t *leaf_prims = (const uint32_t *)sc.ref_leaf_prims;
    const double *tri64 = (const double *)sc.tri64;
    int depth = 0, node = 0;
    while (__ballot(walking) != 0ull) {                       /* the wave goes on while one of its four beams does */
        const lh_refnode_t *nd = &nodes[walking ? node : 0];
        const bool leaf = walking && nd->is_leaf;
        /* ---- leaf: one triangle per lane ---- */
        int cls = 0;
        if (leaf) {
            for (uint32_t base = 0; base < nd->count && cls == 0; base += 16u) {
                int mine = 0;
                if (base + (uint32_t)l < nd->count) mine = beam_triangle(tri64 + 9 * (size_t)leaf_prims[nd->first + base + (uint32_t)l], b);
                /* the group's lanes in order: the first that did not miss */
                for (int q = 0; q < 16 && cls == 0; q++) { const int c = __shfl(mine, (g << 4) | q); if (c != 0) cls = c; }
            }
        }
        /* ---- inner node: one plane of one child per lane (lanes 0 .. 7 of the group) ---- */
        bool miss_plane = false;
        if (walking && !leaf && l < 8) {
            const double *box = nd->box[l >> 2]; const int i = l & 3;
            double no[3];
            for (int k = 0; k < 3; k++) no[k] = ((b.normal[i][k] > 0.0) ? box[k] : box[3 + k]) - b.org[k];
            miss_plane = dot3(no, b.normal[i]) > 0.0;
        }
        const uint32_t mp = (uint32_t)(__ballot(miss_plane) >> (16 * g)) & 0xFFu;
        if (!walking) continue;
        if (leaf) {
            if (cls != 0) { ret = cls; walking = false; }
            else if (depth < 1) { ret = 0; walking = false; }
            else node = stack[g][--depth];
        } else {
            const int hit = ((mp & 0x0Fu) == 0u ? 1 : 0) | ((mp & 0xF0u) == 0u ? 2 : 0);
            if (hit == 0) { if (depth < 1) { ret = 0; walking = false; } else node = stack[g][--depth]; }
            else if (hit == 1) node = nd->child[0];
            else if (hit == 2) node = nd->child[1];
            else {
                const int order = b.dirsign[b.dominant_axis];
                if (depth < 103) { stack[g][depth] = nd->child[1 - order]; depth++; }       /* every lane of the group writes the same value */
                node = nd->child[order];
            }
        }
    }
    if (live && l == 0) result[r] = ret;
}

} /* namespace */

extern "C" int lh_launch_beam_visibility(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dirs,
                                         int32_t *d_result, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_beam_visibility, dim3((unsigned)((n + 3) / 4)), dim3(64), 0, (hipStream_t)stream,
                       *sc, n, d_org, d_dirs, d_result);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
