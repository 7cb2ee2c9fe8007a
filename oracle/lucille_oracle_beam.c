/*
 * lucille_oracle_beam.c -- CPU restatement of the beam (frustum) visibility query.
 * TEST INFRASTRUCTURE ONLY (see lucille_oracle.h).
 *
 * Restated from (paths relative to the lucille tree):
 *   ri_beam_set                              src/render/beam.c:331-465  (incl. the latent
 *                                            maxval re-assignment at :387-390)
 *   ri_bvh_intersect_beam_visibility         src/render/bvh.c:612-667
 *   get_n_point / test_beam_aabb(_misses)    src/render/bvh.c:1997-2089
 *   test_beam_node                           src/render/bvh.c:2097-2126
 *   test_beam_triangle                       src/render/bvh.c:2139-2281
 *   bvh_intersect_leaf_node_beam_visibility  src/render/bvh.c:2435-2542
 *   bvh_traverse_beam_visibility             src/render/bvh.c:2648-2746
 *
 * The answer depends on the reference's tree (leaf contents and order, child visiting
 * order), so this runs on the oracle's restated tree (lucille_oracle.c).  Pinned against the
 * compiled reference in tests/test_oracle_vs_ref.py::test_beam_* and tests/golden/beams_*.npz.
 */
#include "lucille_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

#define EPS 1.0e-14
#define T_INF 1.0e38

typedef struct {
    double org[3], dir[4][3], normal[4][3], t_max;
    int dominant_axis, dirsign[3];
} beam_t;

int  lo_priv_node(const lo_scene_t *s, int32_t idx, const double **box0, const double **box1, int32_t child[2],
                  uint32_t *first, uint32_t *count);
void lo_priv_leaf_tri(const lo_scene_t *s, uint32_t sorted_index, const double **v0, const double **v1, const double **v2);
int  lo_priv_empty(const lo_scene_t *s);

static void cross(double d[3], const double a[3], const double b[3])
{
    d[0] = a[1] * b[2] - a[2] * b[1]; d[1] = a[2] * b[0] - a[0] * b[2]; d[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* ri_beam_set beam.c:331-465; returns -1 when the corner directions straddle an octant */
static int beam_set(beam_t *b, const double org[3], const double dir[4][3])
{
    int i, j, dominant_axis; double maxval;
    b->t_max = T_INF;
    for (i = 0; i < 3; i++) {
        int zeros = 0, mask = 0;
        for (j = 0; j < 4; j++) {
            if (fabs(dir[j][i]) < EPS) zeros++;
            else mask += (dir[j][i] < 0.0) ? 1 : -1;
        }
        if ((mask != -(4 - zeros)) && (mask != (4 - zeros))) return -1;
    }
    for (i = 0; i < 3; i++) b->org[i] = org[i];
    maxval = fabs(dir[0][0]); dominant_axis = 0;
    if (maxval < fabs(dir[0][1])) { maxval = fabs(dir[0][0]); dominant_axis = 1; }   /* sic: beam.c:387-390 */
    if (maxval < fabs(dir[0][2])) { maxval = fabs(dir[0][2]); dominant_axis = 2; }
    b->dominant_axis = dominant_axis;
    for (i = 0; i < 3; i++) b->dirsign[i] = (dir[0][i] < 0.0) ? 1 : 0;
    {
        double normal[3] = { 0.0, 0.0, 0.0 };
        normal[dominant_axis] = 1.0;
        if (b->dirsign[dominant_axis]) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
        for (i = 0; i < 4; i++) {
            double t = dot(dir[i], normal), k;
            if (fabs(t) > EPS) k = 1024.0 / t; else k = 1.0;
            b->dir[i][0] = k * dir[i][0]; b->dir[i][1] = k * dir[i][1]; b->dir[i][2] = k * dir[i][2];
        }
    }
    cross(b->normal[0], b->dir[1], b->dir[0]);
    cross(b->normal[1], b->dir[2], b->dir[1]);
    cross(b->normal[2], b->dir[3], b->dir[2]);
    cross(b->normal[3], b->dir[0], b->dir[3]);
    return 0;
}

/* test_beam_aabb bvh.c:2053-2089: 1 = may hit. box = bmin xyz, bmax xyz */
static int beam_aabb(const double *box, const beam_t *b)
{
    int i, k;
    for (i = 0; i < 4; i++) {
        double np[3], no[3];
        for (k = 0; k < 3; k++) np[k] = (b->normal[i][k] > 0.0) ? box[k] : box[3 + k];
        for (k = 0; k < 3; k++) no[k] = np[k] - b->org[k];
        if (dot(no, b->normal[i]) > 0.0) return 0;
    }
    return 1;
}

/* test_beam_triangle bvh.c:2139-2281 */
static int beam_triangle(const double *v0, const double *v1, const double *v2, const beam_t *b)
{
    double u[4], v[4], t[4], e1[3], e2[3]; int i, mask = 0, cnt;
    for (i = 0; i < 3; i++) { e1[i] = v1[i] - v0[i]; e2[i] = v2[i] - v0[i]; }
    for (i = 0; i < 4; i++) {
        double p[3], q[3], s[3], a, inva;
        cross(p, b->dir[i], e2);
        a = dot(e1, p);
        inva = (fabs(a) > EPS) ? 1.0 / a : 0.0;
        s[0] = b->org[0] - v0[0]; s[1] = b->org[1] - v0[1]; s[2] = b->org[2] - v0[2];
        cross(q, s, e1);
        u[i] = dot(s, p) * inva; v[i] = dot(q, b->dir[i]) * inva; t[i] = dot(e2, q) * inva;
        if ((u[i] < 0.0) || (u[i] > 1.0)) continue;
        if ((v[i] < 0.0) || ((u[i] + v[i]) > 1.0)) continue;
        if ((t[i] < 0.0) || (t[i] > b->t_max)) continue;
        mask |= (1 << i);
    }
    if (mask == 0) {
        cnt = 0; for (i = 0; i < 4; i++) if (t[i] < 0.0) cnt++;
        if (cnt == 4) return 0;
        cnt = 0; for (i = 0; i < 4; i++) if (u[i] < 0.0) cnt++;
        if ((cnt != 0) && (cnt != 4)) return 2;
        cnt = 0; for (i = 0; i < 4; i++) if (u[i] > 1.0) cnt++;
        if ((cnt != 0) && (cnt != 4)) return 2;
        cnt = 0; for (i = 0; i < 4; i++) if (v[i] < 0.0) cnt++;
        if ((cnt != 0) && (cnt != 4)) return 2;
        cnt = 0; for (i = 0; i < 4; i++) if ((u[i] + v[i]) >= 1.0) cnt++;
        if ((cnt != 0) && (cnt != 4)) return 2;
        return 0;
    } else if (mask == 0xf) return 1;
    return 2;
}

/* ri_beam_set + ri_bvh_intersect_beam_visibility for a batch; dirs: n x 4 x 3.
 * result: 0 miss, 1 hit completely, 2 hit partially, -1 ri_beam_set refused the beam */
void lo_beam_visibility_batch(const lo_scene_t *s, size_t n, const double *org, const double *dirs, int32_t *result)
{
    size_t r;
    for (r = 0; r < n; r++) {
        beam_t b; double d4[4][3]; int i, k, ret = 0;
        for (i = 0; i < 4; i++) for (k = 0; k < 3; k++) d4[i][k] = dirs[12 * r + 3 * i + k];
        if (beam_set(&b, &org[3 * r], d4) != 0) { result[r] = -1; continue; }
        if (lo_priv_empty(s)) { result[r] = 0; continue; }
        {
            double sb[6]; lo_scene_bbox(s, sb, sb + 3);
            if (!beam_aabb(sb, &b)) { result[r] = 0; continue; }
        }
        {
            int32_t stack[128]; int depth = 0; int32_t node = 0;
            for (;;) {
                const double *b0, *b1; int32_t child[2]; uint32_t first, count;
                if (lo_priv_node(s, node, &b0, &b1, child, &first, &count)) {      /* leaf */
                    uint32_t q; int cls = 0;
                    for (q = 0; q < count; q++) {
                        const double *v0, *v1, *v2;
                        lo_priv_leaf_tri(s, first + q, &v0, &v1, &v2);
                        cls = beam_triangle(v0, v1, v2, &b);
                        if (cls == 1 || cls == 2) break;
                    }
                    if (cls == 1 || cls == 2) { ret = cls; break; }
                    if (depth < 1) { ret = 0; break; }
                    node = stack[--depth];
                } else {
                    int hit = beam_aabb(b0, &b) | (beam_aabb(b1, &b) << 1);
                    if (hit == 0) { if (depth < 1) { ret = 0; break; } node = stack[--depth]; }
                    else if (hit == 1) node = child[0];
                    else if (hit == 2) node = child[1];
                    else { int order = b.dirsign[b.dominant_axis]; stack[depth++] = child[1 - order]; node = child[order]; }
                }
            }
        }
        result[r] = ret;
    }
}

/* ======================================================================================================================
 * The beam-raster path.  Restated from:
 *   ri_bvh_intersect_beam                    src/render/bvh.c:544-609
 *   bvh_traverse_beam                        src/render/bvh.c:2547-2643
 *   bvh_intersect_leaf_node_beam             src/render/bvh.c:2315-2426
 *   project_triangles                        src/render/bvh.c:2751-2820
 *   ri_beam_clip_by_triangle2d (+ clip, intersect, inside, create_subbeam)   src/render/beam.c:101-311,469-730
 *   ri_raster_plane_setup / ri_rasterize_triangle / ri_rasterize_beam / find_isect_pos_onto_the_triangle_plane
 *                                            src/render/raster.c:42-147,166-327,333-382,389-435
 *   ri_triangle_isect                        src/render/triangle.c:8-68
 *
 * This path is unfinished in the reference (no caller; debug printf()s) and is restated AS IT BEHAVES, not as it was meant:
 *   - the leaf passes (hit_beams, g_miss_beams) where ri_beam_clip_by_triangle2d expects (outer_out, inner_out)
 *     (bvh.c:2387-2390 vs beam.h:96-102): what is rasterised are the parts of the beam OUTSIDE the projected triangle;
 *   - project_triangles scales the vertex itself, not (vertex - org) (bvh.c:2803-2804; the subtraction is commented out);
 *   - the 2-D box of a rasterised triangle takes x from vertices 0, 1 and y from vertices 0, 2 only (raster.c:268-276);
 *   - no depth test: plane->t keeps the LAST triangle that covered a pixel, in traversal order (raster.c:316); u, v, geom and
 *     index of the raster plane are never written;
 *   - a polygon of 6 or 7 outer vertices is cut down to its first three (beam.c:654-662);
 *   - the projected-triangle cache of a leaf (node->child[1 + axis], bvh.c:2343-2364) is filled by the FIRST beam that visits
 *     the leaf and never refreshed: here every beam projects with its own origin, which is the reference's behaviour after
 *     ri_bvh_invalidate_cache (the testbed's "MUST CALL", simplerender.cpp:693) -- pinned that way (ref_harness.c).
 * Undefined in the reference, reported instead of reproduced (flags_out):
 *   [0] pixel tests outside the raster window -- the reference writes plane->t[t * width + s] unchecked (raster.c:300-316:
 *       heap corruption); here the 2-D box is cut to the window and the number of cut-away pixels is counted;
 *   [1] assert(t >= 0.0) of the clipper's intersect() would fire (beam.c:142: the reference aborts);
 *   [2] assert(outer_len < 8) would fire (beam.c:626);
 *   [3] triangles rasterised (diagnostic).
 * ====================================================================================================================== */
typedef struct { double p[2], n[2]; } plane2d_t;
typedef struct { double org[3], dir[4][3]; int is_tetrahedron, dominant_axis; } subbeam_t;
typedef struct {
    double *t; int width, height; double frame[3][3], corner[3], org[3], fov, offset[2];
    uint64_t *flags;
} rplane_t;

#define dot2(a, b) ((a)[0] * (b)[0] + (a)[1] * (b)[1])
int lo_priv_cast_int(double x);

static int rb_inside(const double p[2], const plane2d_t *b)            /* beam.c:156-173 */
{
    double pb[2]; pb[0] = p[0] - b->p[0]; pb[1] = p[1] - b->p[1];
    return dot2(pb, b->n) >= 0;
}

static double rb_intersect(double i_out[2], const double s[2], const double p[2], const plane2d_t *b, uint64_t *flags)   /* beam.c:104-148 */
{
    double v[2], vdotn, sdotn, d, t;
    v[0] = p[0] - s[0]; v[1] = p[1] - s[1];
    vdotn = dot2(v, b->n);
    if (fabs(vdotn) < EPS) vdotn = 1.0;
    d = -(dot2(b->p, b->n));
    sdotn = dot2(s, b->n);
    t = -(sdotn + d) / vdotn;
    if (!(t >= 0.0)) flags[1]++;                      /* assert(t >= 0.0), beam.c:142 */
    i_out[0] = s[0] + t * v[0]; i_out[1] = s[1] + t * v[1];
    return t;
}

static void rb_clip(double (*outer)[2], int *outer_len, double (*inner)[2], int *inner_len,
                    double (*const vin)[2], int len_in, const plane2d_t *pl, uint64_t *flags)       /* beam.c:197-277 */
{
    int j; const double *s = vin[len_in - 1];
    for (j = 0; j < len_in; j++) {
        const double *p = vin[j]; double newv[2], t;
        if (rb_inside(p, pl)) {
            if (rb_inside(s, pl)) { inner[*inner_len][0] = p[0]; inner[*inner_len][1] = p[1]; (*inner_len)++; }
            else {
                t = rb_intersect(newv, s, p, pl, flags);
                if (t < 1.0) { inner[*inner_len][0] = newv[0]; inner[*inner_len][1] = newv[1]; (*inner_len)++; }
                inner[*inner_len][0] = p[0]; inner[*inner_len][1] = p[1]; (*inner_len)++;
                outer[*outer_len][0] = newv[0]; outer[*outer_len][1] = newv[1]; (*outer_len)++;
            }
        } else {
            if (rb_inside(s, pl)) {
                t = rb_intersect(newv, s, p, pl, flags);
                outer[*outer_len][0] = newv[0]; outer[*outer_len][1] = newv[1]; (*outer_len)++;
                outer[*outer_len][0] = p[0]; outer[*outer_len][1] = p[1]; (*outer_len)++;
                if (t > 0.0) { inner[*inner_len][0] = newv[0]; inner[*inner_len][1] = newv[1]; (*inner_len)++; }
            } else { outer[*outer_len][0] = p[0]; outer[*outer_len][1] = p[1]; (*outer_len)++; }
        }
        s = p;
    }
}

static void rb_subbeam(subbeam_t *sb, double (*const p)[2], int tetra, int i0, int i1, int i2, int i3, const subbeam_t *parent)   /* beam.c:279-311 */
{
    static const int axis[3][2] = { {1, 2}, {2, 0}, {0, 1} };
    const int a0 = axis[parent->dominant_axis][0], a1 = axis[parent->dominant_axis][1];
    *sb = *parent;
    sb->dir[0][a0] = p[i0][0]; sb->dir[0][a1] = p[i0][1];
    sb->dir[1][a0] = p[i1][0]; sb->dir[1][a1] = p[i1][1];
    sb->dir[2][a0] = p[i2][0]; sb->dir[2][a1] = p[i2][1];
    sb->dir[3][a0] = p[i3][0]; sb->dir[3][a1] = p[i3][1];
    sb->is_tetrahedron = tetra;
}

/* ri_beam_clip_by_triangle2d, the OUTER beams only (the inner ones go to g_miss_beams and are never read, bvh.c:2387-2405);
 * returns their number (<= 6) */
static int rb_clip_by_triangle2d(subbeam_t *outer_out, const double tri2d[3][2], const subbeam_t *beam, uint64_t *flags)
{
    static const int axis[3][2] = { {1, 2}, {2, 0}, {0, 1} };
    double outer_polygon[3][10][2], inner_polygon[2][16][2];
    double (*input_p)[2], (*inner_p)[2];
    int i, idx, outer_len[3] = {0, 0, 0}, inner_len = 0, len, n = 0;
    plane2d_t plane[3];
    len = beam->is_tetrahedron ? 3 : 4;
    input_p = inner_polygon[0];
    for (i = 0; i < len; i++) {
        input_p[i][0] = beam->dir[i][axis[beam->dominant_axis][0]];
        input_p[i][1] = beam->dir[i][axis[beam->dominant_axis][1]];
    }
    plane[0].n[0] =  (tri2d[1][1] - tri2d[0][1]); plane[0].n[1] = -(tri2d[1][0] - tri2d[0][0]);
    plane[0].p[0] = tri2d[0][0]; plane[0].p[1] = tri2d[0][1];
    plane[1].n[0] =  (tri2d[2][1] - tri2d[1][1]); plane[1].n[1] = -(tri2d[2][0] - tri2d[1][0]);
    plane[1].p[0] = tri2d[1][0]; plane[1].p[1] = tri2d[1][1];
    plane[2].n[0] =  (tri2d[0][1] - tri2d[2][1]); plane[2].n[1] = -(tri2d[0][0] - tri2d[2][0]);
    plane[2].p[0] = tri2d[2][0]; plane[2].p[1] = tri2d[2][1];
    idx = 1;
    for (i = 0; i < 3; i++) {
        inner_p = inner_polygon[idx]; inner_len = 0; outer_len[i] = 0;
        rb_clip(outer_polygon[i], &outer_len[i], inner_p, &inner_len, input_p, len, &plane[i], flags);
        input_p = inner_p; len = inner_len;
        if (inner_len == 0) break;
        idx ^= 1;
    }
    for (i = 0; i < 3; i++) {
        if (outer_len[i] == 0) continue;
        if (!(outer_len[i] < 8)) flags[2]++;                 /* assert( outer_len[i] < 8 ), beam.c:626 */
        if (outer_len[i] == 5) {
            rb_subbeam(outer_out + n, outer_polygon[i], 1, 0, 1, 2, 2, beam); n++;
            rb_subbeam(outer_out + n, outer_polygon[i], 0, 2, 3, 4, 0, beam); n++;
        } else if (outer_len[i] == 4) {
            rb_subbeam(outer_out + n, outer_polygon[i], 0, 0, 1, 2, 3, beam); n++;
        } else {
            rb_subbeam(outer_out + n, outer_polygon[i], 1, 0, 1, 2, 2, beam); n++;
        }
    }
    return n;
}

/* ri_triangle_isect triangle.c:8-68 with *t_inout = RI_INFINITY */
static int rb_triangle_isect(double *t_out, const double tv[3][3], const double org[3], const double dir[3])
{
    double e1[3], e2[3], p[3], s[3], q[3], a, inva, t, u, v; int k;
    for (k = 0; k < 3; k++) { e1[k] = tv[1][k] - tv[0][k]; e2[k] = tv[2][k] - tv[0][k]; }
    cross(p, dir, e2);
    a = dot(e1, p);
    if (fabs(a) > EPS) inva = 1.0 / a; else return 0;
    for (k = 0; k < 3; k++) s[k] = org[k] - tv[0][k];
    cross(q, s, e1);
    u = dot(s, p) * inva; v = dot(q, dir) * inva; t = dot(e2, q) * inva;
    if ((u < 0.0) || (u > 1.0)) return 0;
    if ((v < 0.0) || ((u + v) > 1.0)) return 0;
    if ((t < EPS) || (t > T_INF)) return 0;
    *t_out = t;
    return 1;
}

/* ri_rasterize_triangle raster.c:166-327 */
static void rb_rasterize_triangle(rplane_t *pl, const double tv[3][3])
{
    const int width = pl->width, height = pl->height;
    const double fov_rad = pl->fov * M_PI / 180.0;
    double p[3][3], bmin[2], bmax[2]; int i, s, t, s0, s1, t0, t1;
    for (i = 0; i < 3; i++) {
        double vo[3], w[3];
        vo[0] = tv[i][0] - pl->org[0]; vo[1] = tv[i][1] - pl->org[1]; vo[2] = tv[i][2] - pl->org[2];
        w[0] =  pl->frame[0][0] * vo[0] + pl->frame[0][1] * vo[1] + pl->frame[0][2] * vo[2];
        w[1] =  pl->frame[1][0] * vo[0] + pl->frame[1][1] * vo[1] + pl->frame[1][2] * vo[2];
        w[2] = -pl->frame[2][0] * vo[0] - pl->frame[2][1] * vo[1] - pl->frame[2][2] * vo[2];
        p[i][0] = (1.0 / tan(0.5 * fov_rad)) * w[0];
        p[i][1] = (1.0 / tan(0.5 * fov_rad)) * w[1];
        p[i][2] = w[2];
        p[i][0] /= -w[2]; p[i][1] /= -w[2];
        p[i][0] -= pl->offset[0]; p[i][1] -= pl->offset[1];
        p[i][0] *= 0.5 * width; p[i][1] *= 0.5 * height;
    }
    bmin[0] = bmax[0] = p[0][0]; bmin[1] = bmax[1] = p[0][1];
    bmin[0] = (p[1][0] < bmin[0]) ? p[1][0] : bmin[0];
    bmax[0] = (p[1][0] > bmax[0]) ? p[1][0] : bmax[0];
    bmin[1] = (p[2][1] < bmin[1]) ? p[2][1] : bmin[1];
    bmax[1] = (p[2][1] > bmax[1]) ? p[2][1] : bmax[1];
    s0 = lo_priv_cast_int(bmin[0]); s1 = lo_priv_cast_int(bmax[0]); t0 = lo_priv_cast_int(bmin[1]); t1 = lo_priv_cast_int(bmax[1]);
    pl->flags[3]++;
    {
        /* pixels of the box outside the window: the reference would test (and possibly write) them */
        const int64_t cs0 = s0 < 0 ? 0 : s0, cs1 = s1 > width ? width : s1, ct0 = t0 < 0 ? 0 : t0, ct1 = t1 > height ? height : t1;
        const int64_t all = (s1 > s0 && t1 > t0) ? ((int64_t)s1 - s0) * ((int64_t)t1 - t0) : 0;
        const int64_t in = (cs1 > cs0 && ct1 > ct0) ? (cs1 - cs0) * (ct1 - ct0) : 0;
        pl->flags[0] += (uint64_t)(all - in);
        s0 = (int)cs0; s1 = (int)cs1; t0 = (int)ct0; t1 = (int)ct1;
    }
    for (t = t0; t < t1; t++) {
        for (s = s0; s < s1; s++) {
            double dir[3], tparam;
            dir[0] = pl->corner[0] + s * pl->frame[0][0] + t * pl->frame[1][0];
            dir[1] = pl->corner[1] + s * pl->frame[0][1] + t * pl->frame[1][1];
            dir[2] = pl->corner[2] + s * pl->frame[0][2] + t * pl->frame[1][2];
            if (rb_triangle_isect(&tparam, tv, pl->org, dir)) pl->t[t * width + s] = tparam;
        }
    }
}

/* ri_rasterize_beam raster.c:333-382 + find_isect_pos_onto_the_triangle_plane :389-435 */
static void rb_rasterize_beam(rplane_t *pl, const subbeam_t *b, const double *v0, const double *v1, const double *v2)
{
    double pts[4][3], e1[3], e2[3], s[3], tri[3][3]; int i, k;
    for (k = 0; k < 3; k++) { e1[k] = v1[k] - v0[k]; e2[k] = v2[k] - v0[k]; s[k] = b->org[k] - v0[k]; }
    for (i = 0; i < 4; i++) {
        double p[3], q[3], a, inva, t;
        cross(p, b->dir[i], e2);
        a = dot(e1, p);
        if (fabs(a) > EPS) inva = 1.0 / a; else inva = 1.0;
        cross(q, s, e1);
        t = dot(e2, q) * inva;
        pts[i][0] = b->org[0] + t * b->dir[i][0]; pts[i][1] = b->org[1] + t * b->dir[i][1]; pts[i][2] = b->org[2] + t * b->dir[i][2];
    }
    for (k = 0; k < 3; k++) { tri[0][k] = pts[0][k]; tri[1][k] = pts[1][k]; tri[2][k] = pts[2][k]; }
    rb_rasterize_triangle(pl, tri);
    if (!b->is_tetrahedron) {
        for (k = 0; k < 3; k++) { tri[0][k] = pts[0][k]; tri[1][k] = pts[2][k]; tri[2][k] = pts[3][k]; }
        rb_rasterize_triangle(pl, tri);
    }
}

/* (int) of a double as the reference's compiler does it (cvttsd2si): truncation; out of range or NaN -> INT_MIN */
int lo_priv_cast_int(double x)
{
    if (!(x > -2147483649.0 && x < 2147483648.0)) return (int)0x80000000;
    return (int)x;
}

/* ri_beam_set + ri_raster_plane_setup + ri_bvh_intersect_beam for one beam.  dirs: 4 x 3 corner directions; frame9: du dv dw;
 * t_out: width * height doubles = plane->t of a freshly set-up plane after the call.  Returns -1 when ri_beam_set refuses
 * the beam (t_out is left untouched), else 0.  flags_out[4]: see the header comment of this section. */
int lo_beam_raster(const lo_scene_t *s, const double *org, const double *dirs, int width, int height, const double *frame9,
                   const double *corner, const double *eye, double fov, double *t_out, uint64_t *flags_out)
{
    static const int uv[3][2] = { {1, 2}, {2, 0}, {0, 1} };
    beam_t b; subbeam_t root; rplane_t pl; double d4[4][3]; int i, k; uint64_t dummy[4];
    uint64_t *flags = flags_out ? flags_out : dummy;
    flags[0] = flags[1] = flags[2] = flags[3] = 0;
    for (i = 0; i < 4; i++) for (k = 0; k < 3; k++) d4[i][k] = dirs[3 * i + k];
    if (beam_set(&b, org, d4) != 0) return -1;
    /* ri_raster_plane_setup raster.c:42-147 */
    pl.t = t_out; pl.width = width; pl.height = height; pl.fov = fov; pl.flags = flags;
    for (i = 0; i < 3; i++) for (k = 0; k < 3; k++) pl.frame[i][k] = frame9[3 * i + k];
    for (k = 0; k < 3; k++) { pl.corner[k] = corner[k]; pl.org[k] = eye[k]; }
    {
        const double fov_rad = pl.fov * M_PI / 180.0; double w[3], p[3];
        w[0] =  pl.frame[0][0] * corner[0] + pl.frame[0][1] * corner[1] + pl.frame[0][2] * corner[2];
        w[1] =  pl.frame[1][0] * corner[0] + pl.frame[1][1] * corner[1] + pl.frame[1][2] * corner[2];
        w[2] = -pl.frame[2][0] * corner[0] - pl.frame[2][1] * corner[1] - pl.frame[2][2] * corner[2];
        p[0] = (1.0 / tan(0.5 * fov_rad)) * w[0]; p[1] = (1.0 / tan(0.5 * fov_rad)) * w[1]; p[2] = w[2];
        p[0] /= -w[2]; p[1] /= -w[2];
        pl.offset[0] = p[0]; pl.offset[1] = p[1];
    }
    memset(t_out, 0, sizeof(double) * (size_t)width * (size_t)height);          /* raster.c:72 (a fresh plane) */
    if (lo_priv_empty(s)) return 0;                                   /* bvh.c:560-563 */
    {
        double sb[6]; lo_scene_bbox(s, sb, sb + 3);
        if (!beam_aabb(sb, &b)) return 0;                             /* bvh.c:586-593 */
    }
    memset(t_out, 0, sizeof(double) * (size_t)width * (size_t)height);          /* bvh.c:2570-2572 */
    for (k = 0; k < 3; k++) root.org[k] = b.org[k];
    for (i = 0; i < 4; i++) for (k = 0; k < 3; k++) root.dir[i][k] = b.dir[i][k];
    root.is_tetrahedron = 0; root.dominant_axis = b.dominant_axis;
    {
        int32_t stack[128]; int depth = 0; int32_t node = 0;
        for (;;) {
            const double *b0, *b1; int32_t child[2]; uint32_t first, count;
            if (lo_priv_node(s, node, &b0, &b1, child, &first, &count)) {      /* leaf: bvh.c:2315-2426 */
                uint32_t q;
                for (q = 0; q < count; q++) {
                    const double *v[3]; double tri2d[3][2]; subbeam_t hit[8]; int j, nhit;
                    const int axis = b.dominant_axis;
                    lo_priv_leaf_tri(s, first + q, &v[0], &v[1], &v[2]);
                    for (j = 0; j < 3; j++) {                                   /* project_triangles bvh.c:2751-2820, d = 1024 */
                        double vo[3], t, kk, n[3] = { 0.0, 0.0, 0.0 };
                        n[axis] = 1.0;
                        vo[0] = v[j][0] - b.org[0]; vo[1] = v[j][1] - b.org[1]; vo[2] = v[j][2] - b.org[2];
                        t = dot(vo, n);
                        if (fabs(t) > EPS) kk = 1024.0 / t; else kk = 0.0;
                        tri2d[j][0] = kk * v[j][uv[axis][0]];
                        tri2d[j][1] = kk * v[j][uv[axis][1]];
                    }
                    nhit = rb_clip_by_triangle2d(hit, tri2d, &root, flags);
                    for (j = 0; j < nhit; j++) rb_rasterize_beam(&pl, &hit[j], v[0], v[1], v[2]);
                }
                if (depth < 1) break;
                node = stack[--depth];
            } else {                                                            /* bvh.c:2594-2634 */
                int hitm = beam_aabb(b0, &b) | (beam_aabb(b1, &b) << 1);
                if (hitm == 0) { if (depth < 1) break; node = stack[--depth]; }
                else if (hitm == 1) node = child[0];
                else if (hitm == 2) node = child[1];
                else { int order = b.dirsign[b.dominant_axis]; stack[depth++] = child[1 - order]; node = child[order]; }
            }
        }
    }
    return 0;
}
