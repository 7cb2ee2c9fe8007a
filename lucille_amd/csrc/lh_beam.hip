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

/* a beam the CALLER's ri_beam_set has set up (lh_beam_set_t, include/lucille_hip.h: the fields of lucille's own ri_beam_t the
 * beam queries read -- beam.h:45-84 org, dir[4], normal[4], dominant_axis, dirsign[3]): taken as it is, nothing recomputed */
__device__ __forceinline__ void beam_load(Beam &b, const lh_beam_set_t *s)
{
    for (int k = 0; k < 3; k++) { b.org[k] = s->org[k]; b.dirsign[k] = s->dirsign[k]; }
    for (int i = 0; i < 4; i++) for (int k = 0; k < 3; k++) { b.dir[i][k] = s->dir[i][k]; b.normal[i][k] = s->normal[i][k]; }
    b.dominant_axis = s->dominant_axis;
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
                                                        const double *__restrict__ dirs, int32_t *__restrict__ result,
                                                        const lh_beam_set_t *__restrict__ preset)
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
        int bad = 0;
        if (preset) beam_load(b, preset + r); else bad = beam_set(b, org + 3 * r, dirs + 12 * r);
        if (bad != 0) ret = -1;
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


/* ====================================================================================================================
 * The beam-raster path: ri_bvh_intersect_beam as the reference BEHAVES.
 *
 *   ri_bvh_intersect_beam                    src/render/bvh.c:544-609
 *   bvh_traverse_beam                        src/render/bvh.c:2547-2643
 *   bvh_intersect_leaf_node_beam             src/render/bvh.c:2315-2426
 *   project_triangles                        src/render/bvh.c:2751-2820
 *   ri_beam_clip_by_triangle2d               src/render/beam.c:469-730 (clip / intersect / inside / create_subbeam :101-311)
 *   ri_raster_plane_setup, ri_rasterize_triangle, ri_rasterize_beam, find_isect_pos_onto_the_triangle_plane
 *                                            src/render/raster.c:42-147,166-327,333-382,389-435
 *   ri_triangle_isect                        src/render/triangle.c:8-68
 *
 * The path is unfinished in the reference (no caller, debug printf()s): what it computes is kept, quirk for quirk, because the
 * raster plane it leaves behind is the only observable -- the parts of the beam OUTSIDE each projected triangle are what is
 * rasterised (the leaf swaps the clipper's outer / inner outputs, bvh.c:2387-2390), project_triangles scales the vertex and
 * not (vertex - org) (:2803-2804), the 2-D box of a rasterised triangle takes x from vertices 0, 1 and y from vertices 0, 2
 * (raster.c:268-276), there is no depth test (the LAST triangle in traversal order keeps a pixel, raster.c:316), only
 * plane->t is written, a 6- or 7-gon keeps its first three vertices (beam.c:654-662).  The per-leaf cache of projected
 * triangles (bvh.c:2343-2364) is filled by the first beam that visits a leaf; every beam here projects with its own origin,
 * which is the reference after ri_bvh_invalidate_cache (the testbed's "MUST CALL", simplerender.cpp:693).
 *
 * What is undefined in the reference is reported, not reproduced (flags, four u64 per beam): [0] pixel tests outside the
 * raster window (plane->t[t * width + s] is written unchecked, raster.c:300-316: heap corruption) -- the box is cut to the
 * window; [1] / [2] the asserts of beam.c:142 / :626 would have fired (abort); [3] triangles rasterised.
 *
 * Schedule: ONE WAVE PER BEAM.  The traversal, the projection and the 2-D clipping are wave-uniform (every lane holds the beam
 * and walks the reference-order tree in the reference's order; the polygons and the node stack live in LDS); the pixel loop of
 * a rasterised triangle is spread over the lanes by COLUMN (lane = s mod 64), so that a pixel is always written by the same
 * lane and "the last triangle wins" is that lane's program order -- no atomics, no depth buffer, the reference's result.
 * A low-volume fp64 query like beam visibility; no MFMA, nothing to tile. */
struct SubBeam { double org[3], dir[4][3]; int tetra, dom; };
struct RPlane { double *t; int width, height; double frame[3][3], corner[3], org[3], ktan, offset[2]; };
struct Plane2 { double p[2], n[2]; };

__device__ __forceinline__ double dot2(const double a[2], const double b[2]) { LH_NC return a[0] * b[0] + a[1] * b[1]; }

/* (int) of a double as the reference's compiler does it (cvttsd2si): truncation; out of range or NaN -> INT_MIN */
__device__ __forceinline__ int cast_int(double x)
{
    if (!(x > -2147483649.0 && x < 2147483648.0)) return (int)0x80000000;
    return (int)x;
}

__device__ __forceinline__ bool rb_inside(const double p[2], const Plane2 &b)              /* beam.c:156-173 */
{
    LH_NC
    double pb[2]; pb[0] = p[0] - b.p[0]; pb[1] = p[1] - b.p[1];
    return dot2(pb, b.n) >= 0;
}

__device__ __forceinline__ double rb_intersect(double i_out[2], const double s[2], const double p[2], const Plane2 &b, unsigned long long *flags)   /* beam.c:104-148 */
{
    LH_NC
    double v[2];
    v[0] = p[0] - s[0]; v[1] = p[1] - s[1];
    double vdotn = dot2(v, b.n);
    if (fabs(vdotn) < kEps) vdotn = 1.0;
    const double d = -(dot2(b.p, b.n));
    const double sdotn = dot2(s, b.n);
    const double t = -(sdotn + d) / vdotn;
    if (!(t >= 0.0)) flags[1]++;
    i_out[0] = s[0] + t * v[0]; i_out[1] = s[1] + t * v[1];
    return t;
}

/* clip (beam.c:197-277); the polygons are LDS arrays of (x, y) pairs, wave-uniform */
__device__ void rb_clip(double (*outer)[2], int &outer_len, double (*inner)[2], int &inner_len,
                        double (*vin)[2], int len_in, const Plane2 &pl, unsigned long long *flags)
{
    double s[2] = {vin[len_in - 1][0], vin[len_in - 1][1]};
    for (int j = 0; j < len_in; j++) {
        const double p[2] = {vin[j][0], vin[j][1]};
        double newv[2];
        if (rb_inside(p, pl)) {
            if (rb_inside(s, pl)) { inner[inner_len][0] = p[0]; inner[inner_len][1] = p[1]; inner_len++; }
            else {
                const double t = rb_intersect(newv, s, p, pl, flags);
                if (t < 1.0) { inner[inner_len][0] = newv[0]; inner[inner_len][1] = newv[1]; inner_len++; }
                inner[inner_len][0] = p[0]; inner[inner_len][1] = p[1]; inner_len++;
                outer[outer_len][0] = newv[0]; outer[outer_len][1] = newv[1]; outer_len++;
            }
        } else {
            if (rb_inside(s, pl)) {
                const double t = rb_intersect(newv, s, p, pl, flags);
                outer[outer_len][0] = newv[0]; outer[outer_len][1] = newv[1]; outer_len++;
                outer[outer_len][0] = p[0]; outer[outer_len][1] = p[1]; outer_len++;
                if (t > 0.0) { inner[inner_len][0] = newv[0]; inner[inner_len][1] = newv[1]; inner_len++; }
            } else { outer[outer_len][0] = p[0]; outer[outer_len][1] = p[1]; outer_len++; }
        }
        s[0] = p[0]; s[1] = p[1];
    }
}

/* ri_triangle_isect (triangle.c:8-68) with *t_inout = RI_INFINITY */
__device__ __forceinline__ bool rb_triangle_isect(double &t_out, const double tv[3][3], const double org[3], const double dir[3])
{
    LH_NC
    double e1[3], e2[3], p[3], s[3], q[3];
    for (int k = 0; k < 3; k++) { e1[k] = tv[1][k] - tv[0][k]; e2[k] = tv[2][k] - tv[0][k]; }
    cross3(p, dir, e2);
    const double a = dot3(e1, p);
    if (!(fabs(a) > kEps)) return false;
    const double inva = 1.0 / a;
    for (int k = 0; k < 3; k++) s[k] = org[k] - tv[0][k];
    cross3(q, s, e1);
    const double u = dot3(s, p) * inva, v = dot3(q, dir) * inva, t = dot3(e2, q) * inva;
    if ((u < 0.0) || (u > 1.0)) return false;
    if ((v < 0.0) || ((u + v) > 1.0)) return false;
    if ((t < kEps) || (t > kTInf)) return false;
    t_out = t;
    return true;
}

/* ri_rasterize_triangle (raster.c:166-327): the projection and the box are wave-uniform, the pixels go to the lanes by column */
__device__ void rb_rasterize_triangle(const RPlane &pl, const double tv[3][3], const int lane, unsigned long long *flags)
{
    LH_NC
    const int width = pl.width, height = pl.height;
    double p[3][2];
    for (int i = 0; i < 3; i++) {
        double vo[3], w[3];
        vo[0] = tv[i][0] - pl.org[0]; vo[1] = tv[i][1] - pl.org[1]; vo[2] = tv[i][2] - pl.org[2];
        w[0] =  pl.frame[0][0] * vo[0] + pl.frame[0][1] * vo[1] + pl.frame[0][2] * vo[2];
        w[1] =  pl.frame[1][0] * vo[0] + pl.frame[1][1] * vo[1] + pl.frame[1][2] * vo[2];
        w[2] = -pl.frame[2][0] * vo[0] - pl.frame[2][1] * vo[1] - pl.frame[2][2] * vo[2];
        p[i][0] = pl.ktan * w[0];
        p[i][1] = pl.ktan * w[1];
        p[i][0] /= -w[2]; p[i][1] /= -w[2];
        p[i][0] -= pl.offset[0]; p[i][1] -= pl.offset[1];
        p[i][0] *= 0.5 * width; p[i][1] *= 0.5 * height;
    }
    double bmin[2], bmax[2];
    bmin[0] = bmax[0] = p[0][0]; bmin[1] = bmax[1] = p[0][1];
    bmin[0] = (p[1][0] < bmin[0]) ? p[1][0] : bmin[0];
    bmax[0] = (p[1][0] > bmax[0]) ? p[1][0] : bmax[0];
    bmin[1] = (p[2][1] < bmin[1]) ? p[2][1] : bmin[1];
    bmax[1] = (p[2][1] > bmax[1]) ? p[2][1] : bmax[1];
    int s0 = cast_int(bmin[0]), s1 = cast_int(bmax[0]), t0 = cast_int(bmin[1]), t1 = cast_int(bmax[1]);
    flags[3]++;
    {
        const long long cs0 = s0 < 0 ? 0 : s0, cs1 = s1 > width ? width : s1, ct0 = t0 < 0 ? 0 : t0, ct1 = t1 > height ? height : t1;
        const long long all = (s1 > s0 && t1 > t0) ? ((long long)s1 - s0) * ((long long)t1 - t0) : 0;
        const long long in = (cs1 > cs0 && ct1 > ct0) ? (cs1 - cs0) * (ct1 - ct0) : 0;
        flags[0] += (unsigned long long)(all - in);
        s0 = (int)cs0; s1 = (int)cs1; t0 = (int)ct0; t1 = (int)ct1;
    }
    if (s1 <= s0 || t1 <= t0) return;
    /* this lane's columns: s = lane (mod 64) */
    int s = s0 + ((lane - s0) & 63);
    for (; s < s1; s += 64) {
        for (int t = t0; t < t1; t++) {
            double dir[3], tparam;
            dir[0] = pl.corner[0] + s * pl.frame[0][0] + t * pl.frame[1][0];
            dir[1] = pl.corner[1] + s * pl.frame[0][1] + t * pl.frame[1][1];
            dir[2] = pl.corner[2] + s * pl.frame[0][2] + t * pl.frame[1][2];
            if (rb_triangle_isect(tparam, tv, pl.org, dir)) pl.t[(size_t)t * width + s] = tparam;
        }
    }
}

/* ri_rasterize_beam (raster.c:333-382) + find_isect_pos_onto_the_triangle_plane (:389-435) for the sub-beam whose four 2-D
 * corners are poly[i0..i3] (create_subbeam, beam.c:279-311: the parent's directions with the two in-plane components replaced) */
__device__ void rb_rasterize_subbeam(const RPlane &pl, const Beam &b, double (*poly)[2], bool tetra, int i0, int i1, int i2, int i3,
                                     const double *tv, const int lane, unsigned long long *flags)
{
    LH_NC
    const int a0 = b.dominant_axis == 0 ? 1 : (b.dominant_axis == 1 ? 2 : 0), a1 = b.dominant_axis == 0 ? 2 : (b.dominant_axis == 1 ? 0 : 1);
    const int ix[4] = {i0, i1, i2, i3};
    double pts[4][3], e1[3], e2[3], s[3], tri[3][3];
    for (int k = 0; k < 3; k++) { e1[k] = tv[3 + k] - tv[k]; e2[k] = tv[6 + k] - tv[k]; s[k] = b.org[k] - tv[k]; }
    for (int i = 0; i < 4; i++) {
        double d[3] = {b.dir[i][0], b.dir[i][1], b.dir[i][2]}, p[3], q[3];
        d[a0] = poly[ix[i]][0]; d[a1] = poly[ix[i]][1];
        cross3(p, d, e2);
        const double a = dot3(e1, p);
        const double inva = (fabs(a) > kEps) ? 1.0 / a : 1.0;
        cross3(q, s, e1);
        const double t = dot3(e2, q) * inva;
        pts[i][0] = b.org[0] + t * d[0]; pts[i][1] = b.org[1] + t * d[1]; pts[i][2] = b.org[2] + t * d[2];
    }
    for (int k = 0; k < 3; k++) { tri[0][k] = pts[0][k]; tri[1][k] = pts[1][k]; tri[2][k] = pts[2][k]; }
    rb_rasterize_triangle(pl, tri, lane, flags);
    if (!tetra) {
        for (int k = 0; k < 3; k++) { tri[0][k] = pts[0][k]; tri[1][k] = pts[2][k]; tri[2][k] = pts[3][k]; }
        rb_rasterize_triangle(pl, tri, lane, flags);
    }
}

/* status: 0 the beam was traced (its plane cleared and rasterised), 1 nothing done (empty scene, or the beam misses the scene
 * box: the reference returns before it clears the plane, bvh.c:560-563,586-593), -1 ri_beam_set refuses the beam */
__global__ __launch_bounds__(64) void k_beam_raster(lh_dev_scene_t sc, size_t n, const double *__restrict__ org, const double *__restrict__ dirs,
                                                    const double *__restrict__ corner, lh_raster_plane_t rp, double ktan, double *__restrict__ t_out,
                                                    int32_t *__restrict__ status, unsigned long long *__restrict__ flags_out,
                                                    const lh_beam_set_t *__restrict__ preset)
{
    LH_NC
    __shared__ int stack[104];
    __shared__ double outer_polygon[3][10][2], inner_polygon[2][16][2];
    const int lane = threadIdx.x;
    const size_t r = blockIdx.x;
    if (r >= n) return;
    unsigned long long flags[4] = {0ull, 0ull, 0ull, 0ull};
    Beam b;
    int bad = 0;
    if (preset) beam_load(b, preset + r); else bad = beam_set(b, org + 3 * r, dirs + 12 * r);
    if (bad != 0) { if (lane == 0) { status[r] = -1; if (flags_out) for (int k = 0; k < 4; k++) flags_out[4 * r + k] = 0ull; } return; }
    RPlane pl;
    pl.width = rp.width; pl.height = rp.height; pl.ktan = ktan;
    pl.t = t_out + r * (size_t)rp.width * (size_t)rp.height;
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) pl.frame[i][k] = rp.frame[3 * i + k];
    for (int k = 0; k < 3; k++) { pl.corner[k] = corner[3 * r + k]; pl.org[k] = rp.eye[k]; }
    {   /* ri_raster_plane_setup raster.c:114-144: the lower-left corner in NDC */
        double w[3], p[2];
        w[0] =  pl.frame[0][0] * pl.corner[0] + pl.frame[0][1] * pl.corner[1] + pl.frame[0][2] * pl.corner[2];
        w[1] =  pl.frame[1][0] * pl.corner[0] + pl.frame[1][1] * pl.corner[1] + pl.frame[1][2] * pl.corner[2];
        w[2] = -pl.frame[2][0] * pl.corner[0] - pl.frame[2][1] * pl.corner[1] - pl.frame[2][2] * pl.corner[2];
        p[0] = ktan * w[0]; p[1] = ktan * w[1];
        p[0] /= -w[2]; p[1] /= -w[2];
        pl.offset[0] = p[0]; pl.offset[1] = p[1];
    }
    bool go = !sc.ref_empty;
    if (go) {
        const double sb[6] = {sc.ref_bmin[0], sc.ref_bmin[1], sc.ref_bmin[2], sc.ref_bmax[0], sc.ref_bmax[1], sc.ref_bmax[2]};
        go = beam_aabb(sb, b) != 0;
    }
    if (!go) { if (lane == 0) { status[r] = 1; if (flags_out) for (int k = 0; k < 4; k++) flags_out[4 * r + k] = 0ull; } return; }
    /* memset(raster->t, 0, ...) bvh.c:2570-2572: every lane clears the columns it will write */
    for (int t = 0; t < pl.height; t++) for (int s = lane; s < pl.width; s += 64) pl.t[(size_t)t * pl.width + s] = 0.0;

    const lh_refnode_t *nodes = (const lh_refnode_t *)sc.ref_nodes;
    const uint32_t *leaf_prims = (const uint32_t *)sc.ref_leaf_prims;
    const double *tri64 = (const double *)sc.tri64;
    const int a0 = b.dominant_axis == 0 ? 1 : (b.dominant_axis == 1 ? 2 : 0), a1 = b.dominant_axis == 0 ? 2 : (b.dominant_axis == 1 ? 0 : 1);
    int depth = 0, node = 0;
    for (;;) {
        const lh_refnode_t *nd = &nodes[node];
        if (nd->is_leaf) {
            for (uint32_t q = 0; q < nd->count; q++) {
                const double *tv = tri64 + 9 * (size_t)leaf_prims[nd->first + q];
                double tri2d[3][2];
                for (int j = 0; j < 3; j++) {             /* project_triangles bvh.c:2751-2820 (d = 1024) */
                    const double t = tv[3 * j + b.dominant_axis] - b.org[b.dominant_axis];
                    /* vdot(vo, n) with n the unit axis: the two products with 0.0 add +-0.0 -- the sum is vo[axis] up to the sign of zero, which fabs() ignores */
                    const double kk = (fabs(t) > kEps) ? 1024.0 / t : 0.0;
                    tri2d[j][0] = kk * tv[3 * j + a0];
                    tri2d[j][1] = kk * tv[3 * j + a1];
                }
                /* ri_beam_clip_by_triangle2d beam.c:469-730 on the root beam (is_tetrahedron = 0) */
                Plane2 plane[3];
                plane[0].n[0] =  (tri2d[1][1] - tri2d[0][1]); plane[0].n[1] = -(tri2d[1][0] - tri2d[0][0]);
                plane[0].p[0] = tri2d[0][0]; plane[0].p[1] = tri2d[0][1];
                plane[1].n[0] =  (tri2d[2][1] - tri2d[1][1]); plane[1].n[1] = -(tri2d[2][0] - tri2d[1][0]);
                plane[1].p[0] = tri2d[1][0]; plane[1].p[1] = tri2d[1][1];
                plane[2].n[0] =  (tri2d[0][1] - tri2d[2][1]); plane[2].n[1] = -(tri2d[0][0] - tri2d[2][0]);
                plane[2].p[0] = tri2d[2][0]; plane[2].p[1] = tri2d[2][1];
                int outer_len[3] = {0, 0, 0}, len = 4, idx = 1, cur = 0;
                for (int i = 0; i < 4; i++) { inner_polygon[0][i][0] = b.dir[i][a0]; inner_polygon[0][i][1] = b.dir[i][a1]; }
                for (int i = 0; i < 3; i++) {
                    int inner_len = 0;
                    rb_clip(outer_polygon[i], outer_len[i], inner_polygon[idx], inner_len, inner_polygon[cur], len, plane[i], flags);
                    cur = idx; len = inner_len;
                    if (inner_len == 0) break;
                    idx ^= 1;
                }
                for (int i = 0; i < 3; i++) {
                    if (outer_len[i] == 0) continue;
                    if (!(outer_len[i] < 8)) flags[2]++;
                    if (outer_len[i] == 5) {
                        rb_rasterize_subbeam(pl, b, outer_polygon[i], true, 0, 1, 2, 2, tv, lane, flags);
                        rb_rasterize_subbeam(pl, b, outer_polygon[i], false, 2, 3, 4, 0, tv, lane, flags);
                    } else if (outer_len[i] == 4) rb_rasterize_subbeam(pl, b, outer_polygon[i], false, 0, 1, 2, 3, tv, lane, flags);
                    else rb_rasterize_subbeam(pl, b, outer_polygon[i], true, 0, 1, 2, 2, tv, lane, flags);
                }
            }
            if (depth < 1) break;
            node = stack[--depth];
        } else {
            const int hit = beam_aabb(nd->box[0], b) | (beam_aabb(nd->box[1], b) << 1);
            if (hit == 0) { if (depth < 1) break; node = stack[--depth]; }
            else if (hit == 1) node = nd->child[0];
            else if (hit == 2) node = nd->child[1];
            else {
                const int order = b.dirsign[b.dominant_axis];
                if (depth < 103) { stack[depth] = nd->child[1 - order]; depth++; }
                node = nd->child[order];
            }
        }
    }
    if (lane == 0) { status[r] = 0; if (flags_out) for (int k = 0; k < 4; k++) flags_out[4 * r + k] = flags[k]; }
}

} /* namespace */

extern "C" int lh_launch_beam_visibility(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dirs,
                                         int32_t *d_result, void *stream, const lh_beam_set_t *d_preset)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_beam_visibility, dim3((unsigned)((n + 3) / 4)), dim3(64), 0, (hipStream_t)stream,
                       *sc, n, d_org, d_dirs, d_result, d_preset);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* ktan = 1.0 / tan(0.5 * fov * M_PI / 180.0) as the reference computes it on the host (raster.c:120,135-136): passed in, so that
 * the device never evaluates tan() */
extern "C" int lh_launch_beam_raster(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dirs, const double *d_corner,
                                     const lh_raster_plane_t *plane, double ktan, double *d_t, int32_t *d_status, unsigned long long *d_flags,
                                     void *stream, const lh_beam_set_t *d_preset)
{
    if (n == 0) return 0;
    if (n > 0x7fffffffull) return -1;
    hipLaunchKernelGGL(k_beam_raster, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, *sc, n, d_org, d_dirs, d_corner, *plane, ktan,
                       d_t, d_status, d_flags, d_preset);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
