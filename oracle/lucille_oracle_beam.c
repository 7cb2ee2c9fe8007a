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
