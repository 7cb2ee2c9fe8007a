/*
 * lucille_oracle.c -- CPU restatement (plain C, double precision, no FMA) of
 * lucille's BVH hot path.  TEST INFRASTRUCTURE ONLY: see lucille_oracle.h.
 *
 * What is restated, and from where (paths relative to the lucille tree):
 *
 *   create_triangle_list    src/render/bvh.c:1736-1826
 *   calc_scene_bbox         src/render/bvh.c:1829-1849
 *   bbox_add_margin         src/render/bvh.c:1697-1731
 *   bin_triangle_edge       src/render/bvh.c:1571-1692
 *   SAH / find_cut_from_bin src/render/bvh.c:1210-1326
 *   bvh_construct           src/render/bvh.c:1328-1564
 *   ri_bvh_intersect        src/render/bvh.c:430-542   (ray precompute)
 *   test_ray_aabb           src/render/bvh.c:869-936
 *   test_ray_node           src/render/bvh.c:938-1083  (scalar branch :1030-1044)
 *   bvh_traverse            src/render/bvh.c:1092-1188
 *   bvh_intersect_leaf_node src/render/bvh.c:793-864
 *   triangle_isect          src/render/bvh.c:730-791
 *
 * The data structures are this file's own (flat index-linked node array, flat
 * triangle array); only the arithmetic, its operation order, the comparison
 * operators and the traversal order are taken from the reference, because
 * those decide the bits of (prim, t, u, v) and the per-ray counters.
 *
 * Build with -O2 -msse2 -ffp-contract=off: the reference is built by gcc -O2
 * -msse2 on x86-64 (SConstruct:81-85,183-185), i.e. SSE2 scalar doubles, no
 * x87 excess precision and no fused multiply-add.
 *
 * Known undefined behaviour in the reference that is NOT reproduced:
 * bvh.c:483-487 leaves ray->invdir[1] unset when |dir.y| <= 1e-14 (it writes
 * invdir[2] instead).  Here invdir[1] = +-DBL_MAX, the evident intent; parity
 * batches exclude such rays.
 */
#include "lucille_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define LO_NTRIS_LEAF 16   /* BVH_NTRIS_LEAF  bvh.c:81 */
#define LO_BIN_SIZE   64   /* BVH_BIN_SIZE    bvh.c:82 */
#define LO_MAXDEPTH   100  /* BVH_MAXDEPTH    bvh.c:80 */
#define LO_EPS        1.0e-14 /* RI_EPS       src/base/common.h:27 */
#define LO_INFINITY   1.0e38  /* RI_INFINITY  include/ri.h:47 */

typedef struct {
    double v[3][3];
    uint32_t geom;   /* ordinal of the mesh in geom_list order */
    uint32_t index;  /* 3*i offset into that mesh's index list (bvh.c:1813) */
    uint32_t prim;   /* running idx of create_triangle_list (bvh.c:1792-1821) */
} lo_tri_t;

typedef struct {
    double   bmin[3], bmax[3];
    uint64_t index;
} lo_tribox_t;

typedef struct {
    /* inner: child boxes; [k][0..2]=bmin, [k][3..5]=bmax */
    double   box[2][6];
    int32_t  child[2];
    int32_t  axis;
    int32_t  is_leaf;
    uint32_t first;  /* leaf: first triangle in the sorted triangle array */
    uint32_t count;  /* leaf: triangle count */
} lo_node_t;

typedef struct {
    uint32_t npos, nidx;
    double  *pos;   /* xyz */
    uint32_t *idx;
    double  *nrm;   /* optional vertex normals xyz (geom->normals) */
    int      two_side;
    double  *attr[5]; /* optional: colors, tangents, binormals (xyz per vertex), texcoords (st per vertex),
                         texcoords_unshared (st per index) -- geom.h:34-48 */
} lo_mesh_t;

struct lo_scene {
    lo_mesh_t *meshes;
    uint32_t   nmeshes;

    int        built;
    int        empty;
    double     bmin[3], bmax[3];
    lo_tri_t  *tris;       /* leaf-sorted */
    lo_tri_t  *tris_orig;  /* primID order */
    uint64_t   ntris;
    lo_node_t *nodes;
    uint64_t   nnodes, cap_nodes;
    uint64_t   max_depth;
};

/* ---------------------------------------------------------------- scene */

lo_scene_t *lo_scene_new(void)
{
    return (lo_scene_t *)calloc(1, sizeof(lo_scene_t));
}

static void free_tree(lo_scene_t *s)
{
    free(s->tris); free(s->tris_orig); free(s->nodes);
    s->tris = s->tris_orig = NULL; s->nodes = NULL;
    s->nnodes = s->cap_nodes = 0; s->ntris = 0; s->built = 0;
}

void lo_scene_free(lo_scene_t *s)
{
    uint32_t i;
    if (!s) return;
    for (i = 0; i < s->nmeshes; i++) {
        int k;
        free(s->meshes[i].pos); free(s->meshes[i].idx); free(s->meshes[i].nrm);
        for (k = 0; k < 5; k++) free(s->meshes[i].attr[k]);
    }
    free(s->meshes);
    free_tree(s);
    free(s);
}

int lo_scene_add_mesh(lo_scene_t *s, uint32_t npos, const double *pos,
                      uint32_t nidx, const uint32_t *idx)
{
    lo_mesh_t *m;
    s->meshes = (lo_mesh_t *)realloc(s->meshes, sizeof(lo_mesh_t) * (s->nmeshes + 1));
    m = &s->meshes[s->nmeshes++];
    m->npos = npos; m->nidx = nidx; m->nrm = NULL; m->two_side = 0;
    memset(m->attr, 0, sizeof(m->attr));
    m->pos = (double *)malloc(sizeof(double) * 3 * (npos ? npos : 1));
    m->idx = (uint32_t *)malloc(sizeof(uint32_t) * (nidx ? nidx : 1));
    memcpy(m->pos, pos, sizeof(double) * 3 * npos);
    memcpy(m->idx, idx, sizeof(uint32_t) * nidx);
    return 0;
}

int lo_scene_set_normals(lo_scene_t *s, uint32_t mesh, const double *normals_xyz, int two_side)
{
    lo_mesh_t *m;
    if (mesh >= s->nmeshes) return -1;
    m = &s->meshes[mesh];
    free(m->nrm); m->nrm = NULL;
    if (normals_xyz) {
        m->nrm = (double *)malloc(sizeof(double) * 3 * (m->npos ? m->npos : 1));
        memcpy(m->nrm, normals_xyz, sizeof(double) * 3 * m->npos);
    }
    m->two_side = two_side;
    return 0;
}

/* optional attributes of a mesh (ri_geom_add_colors / _tangents / _binormals / _texcoords /
 * _texcoords_unshared, geom.c:123-290).  kind 0..2: xyz per vertex; 3: st per vertex; 4: st per index. */
int lo_scene_set_attribute(lo_scene_t *s, uint32_t mesh, int kind, const double *data)
{
    lo_mesh_t *m; size_t n;
    if (mesh >= s->nmeshes || kind < 0 || kind > 4) return -1;
    m = &s->meshes[mesh];
    free(m->attr[kind]); m->attr[kind] = NULL;
    if (!data) return 0;
    n = (kind <= 2) ? 3 * (size_t)m->npos : (kind == 3 ? 2 * (size_t)m->npos : 2 * (size_t)m->nidx);
    m->attr[kind] = (double *)malloc(sizeof(double) * (n ? n : 1));
    memcpy(m->attr[kind], data, sizeof(double) * n);
    return 0;
}

/* the corners' attribute values of primitive `prim` (NULL where the mesh has none) */
void lo_priv_prim_attributes(const lo_scene_t *s, uint32_t prim, const double *a[5][3])
{
    const lo_tri_t *t = &s->tris_orig[prim];
    const lo_mesh_t *m = &s->meshes[t->geom];
    uint32_t i[3]; int k, c;
    for (c = 0; c < 3; c++) i[c] = m->idx[t->index + c];
    for (k = 0; k < 5; k++)
        for (c = 0; c < 3; c++) {
            if (!m->attr[k]) a[k][c] = NULL;
            else if (k <= 2) a[k][c] = &m->attr[k][3 * (size_t)i[c]];
            else if (k == 3) a[k][c] = &m->attr[k][2 * (size_t)i[c]];
            else a[k][c] = &m->attr[k][2 * (size_t)(t->index + c)];
        }
}

/* ---------------------------------------------------------------- build */

/* bbox_add_margin  bvh.c:1697-1731 */
static void add_margin(double bmin[3], double bmax[3])
{
    int i; double sc[3];
    for (i = 0; i < 3; i++) {
        double scale = bmax[i] - bmin[i];
        sc[i] = (scale < LO_EPS) ? LO_EPS : LO_EPS * scale;
    }
    for (i = 0; i < 3; i++) { bmin[i] -= sc[i]; bmax[i] += sc[i]; }
}

/* calc_bbox_of_triangles bvh.c:1872-1895 / calc_scene_bbox :1829-1849
 * (vmin/vmax are plain < / > selects, src/base/vector.h) */
static void bbox_of(double bmin[3], double bmax[3], const lo_tribox_t *b, uint64_t n)
{
    uint64_t i; int k;
    for (k = 0; k < 3; k++) { bmin[k] = b[0].bmin[k]; bmax[k] = b[0].bmax[k]; }
    for (i = 1; i < n; i++)
        for (k = 0; k < 3; k++) {
            bmin[k] = (bmin[k] < b[i].bmin[k]) ? bmin[k] : b[i].bmin[k];
            bmax[k] = (bmax[k] > b[i].bmax[k]) ? bmax[k] : b[i].bmax[k];
        }
}

static double surface_area(const double bmin[3], const double bmax[3])
{   /* calc_surface_area bvh.c:1190-1208 */
    double sa = (bmax[0] - bmin[0]) * (bmax[1] - bmin[1]) +
                (bmax[1] - bmin[1]) * (bmax[2] - bmin[2]) +
                (bmax[2] - bmin[2]) * (bmax[0] - bmin[0]);
    sa *= 2.0;
    return sa;
}

/* SAH bvh.c:1210-1228: the sum is evaluated in double (float constants are
 * promoted) and then narrowed to float on assignment to T. */
static double sah(int ns1, double left_area, int ns2, double right_area, double s)
{
    const float Taabb = 0.2f, Ttri = 0.8f;
    float T;
    T = 2.0f * Taabb + (left_area / s) * (double)ns1 * Ttri +
        (right_area / s) * (double)ns2 * Ttri;
    return T;
}

typedef struct { uint32_t bin[2][3][LO_BIN_SIZE]; } lo_bins_t;

/* bin_triangle_edge bvh.c:1571-1692 */
static void bin_edges(lo_bins_t *bb, const double smin[3], const double smax[3],
                      const lo_tribox_t *b, uint64_t n)
{
    double size[3], inv[3]; int k; uint64_t i;
    const double binsize = (double)LO_BIN_SIZE;
    for (k = 0; k < 3; k++) {
        size[k] = smax[k] - smin[k];
        inv[k] = (size[k] > LO_EPS) ? binsize / size[k] : 0.0;
    }
    memset(bb, 0, sizeof(*bb));
    for (i = 0; i < n; i++)
        for (k = 0; k < 3; k++) {
            double qmin = (b[i].bmin[k] - smin[k]) * inv[k];
            double qmax = (b[i].bmax[k] - smin[k]) * inv[k];
            uint32_t imin = (uint32_t)qmin, imax = (uint32_t)qmax;
            if (imin >= LO_BIN_SIZE) imin = LO_BIN_SIZE - 1;
            if (imax >= LO_BIN_SIZE) imax = LO_BIN_SIZE - 1;
            bb->bin[0][k][imin]++;
            bb->bin[1][k][imax]++;
        }
}

/* find_cut_from_bin bvh.c:1230-1326 */
static void find_cut(double *cut_pos, int *cut_axis, const lo_bins_t *bb,
                     const double bmin[3], const double bmax[3], uint64_t ntris)
{
    int i, j, k; double bstep[3];
    double min_cost = LO_INFINITY, min_pos = 0.0; int min_axis = 0;
    double sa_total = surface_area(bmin, bmax);
    for (k = 0; k < 3; k++) bstep[k] = (bmax[k] - bmin[k]) / (double)LO_BIN_SIZE;
    for (j = 0; j < 3; j++) {
        uint64_t left = 0, right = ntris;
        double lmin[3], lmax[3], rmin[3], rmax[3];
        for (k = 0; k < 3; k++) { lmin[k] = rmin[k] = bmin[k]; lmax[k] = rmax[k] = bmax[k]; }
        for (i = 0; i < LO_BIN_SIZE - 1; i++) {
            double pos, cost;
            left  += bb->bin[0][j][i];
            right -= bb->bin[1][j][i];
            pos = bmin[j] + (i + 1) * bstep[j];
            lmax[j] = pos; rmin[j] = pos;
            cost = sah((int)left, surface_area(lmin, lmax), (int)right,
                       surface_area(rmin, rmax), sa_total);
            if (cost < min_cost) { min_cost = cost; min_axis = j; min_pos = pos; }
        }
    }
    *cut_axis = min_axis; *cut_pos = min_pos;
}

static int32_t new_node(lo_scene_t *s)
{
    if (s->nnodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 1024;
        s->nodes = (lo_node_t *)realloc(s->nodes, sizeof(lo_node_t) * s->cap_nodes);
    }
    memset(&s->nodes[s->nnodes], 0, sizeof(lo_node_t));
    return (int32_t)s->nnodes++;
}

/* bvh_construct bvh.c:1328-1564.  `boxes` is the in-place partitioned list,
 * `scratch` the copy it is partitioned from (tri_bboxes_buf, offset 0). */
static void construct(lo_scene_t *s, int32_t me, const double bmin[3],
                      const double bmax[3], lo_tribox_t *boxes,
                      lo_tribox_t *scratch, uint64_t il, uint64_t ir,
                      uint64_t depth)
{
    uint64_t n = ir - il, i, nl = 0, nr;
    double cut_pos; int cut_axis;
    lo_bins_t bins;
    double lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t cl, cr; int k;

    if (depth > s->max_depth) s->max_depth = depth;

    if (n <= LO_NTRIS_LEAF) {
        /* gather_triangles bvh.c:1897-1917 */
        for (i = 0; i < n; i++) s->tris[il + i] = s->tris_orig[boxes[il + i].index];
        s->nodes[me].is_leaf = 1;
        s->nodes[me].first = (uint32_t)il;
        s->nodes[me].count = (uint32_t)n;
        return;
    }

    bin_edges(&bins, bmin, bmax, boxes + il, n);
    find_cut(&cut_pos, &cut_axis, &bins, bmin, bmax, n);

    /* partition bvh.c:1437-1468: left fills forward, right fills backward */
    nr = n - 1;
    memcpy(scratch, boxes + il, sizeof(lo_tribox_t) * n);
    for (i = 0; i < n; i++) {
        if (scratch[i].bmax[cut_axis] < cut_pos) boxes[il + nl++] = scratch[i];
        else                                     boxes[il + nr--] = scratch[i];
    }
    if (nl == 0 || nl == n) nl = n / 2;   /* bvh.c:1471-1478 */

    cl = new_node(s); cr = new_node(s);
    s->nodes[me].child[0] = cl; s->nodes[me].child[1] = cr;
    s->nodes[me].axis = cut_axis;

    bbox_of(lmin, lmax, boxes + il, nl);
    add_margin(lmin, lmax);
    for (k = 0; k < 3; k++) { s->nodes[me].box[0][k] = lmin[k]; s->nodes[me].box[0][3 + k] = lmax[k]; }
    construct(s, cl, lmin, lmax, boxes, scratch, il, il + nl, depth + 1);

    bbox_of(rmin, rmax, boxes + il + nl, n - nl);
    add_margin(rmin, rmax);
    for (k = 0; k < 3; k++) { s->nodes[me].box[1][k] = rmin[k]; s->nodes[me].box[1][3 + k] = rmax[k]; }
    construct(s, cr, rmin, rmax, boxes, scratch, il + nl, ir, depth + 1);
}

int lo_scene_build(lo_scene_t *s)
{
    uint64_t n = 0, idx = 0; uint32_t g, i; int k, c;
    lo_tribox_t *boxes, *scratch;

    free_tree(s);
    for (g = 0; g < s->nmeshes; g++) n += s->meshes[g].nidx / 3;
    s->built = 1;
    s->ntris = n;
    if (n == 0) { s->empty = 1; return 0; }   /* bvh.c:311-315 */
    s->empty = 0;

    /* create_triangle_list bvh.c:1736-1826 */
    s->tris_orig = (lo_tri_t *)malloc(sizeof(lo_tri_t) * n);
    s->tris = (lo_tri_t *)malloc(sizeof(lo_tri_t) * n);
    boxes = (lo_tribox_t *)malloc(sizeof(lo_tribox_t) * n);
    scratch = (lo_tribox_t *)malloc(sizeof(lo_tribox_t) * n);
    for (g = 0; g < s->nmeshes; g++) {
        const lo_mesh_t *m = &s->meshes[g];
        for (i = 0; i < m->nidx / 3; i++) {
            lo_tri_t *t = &s->tris_orig[idx];
            for (c = 0; c < 3; c++)
                for (k = 0; k < 3; k++)
                    t->v[c][k] = m->pos[3 * (size_t)m->idx[3 * i + c] + k];
            t->geom = g; t->index = 3 * i; t->prim = (uint32_t)idx;
            for (k = 0; k < 3; k++) {   /* get_bbox_of_triangle bvh.c:1852-1869 */
                double lo = t->v[0][k], hi = t->v[0][k];
                lo = (lo < t->v[1][k]) ? lo : t->v[1][k];
                lo = (lo < t->v[2][k]) ? lo : t->v[2][k];
                hi = (hi > t->v[1][k]) ? hi : t->v[1][k];
                hi = (hi > t->v[2][k]) ? hi : t->v[2][k];
                boxes[idx].bmin[k] = lo; boxes[idx].bmax[k] = hi;
            }
            boxes[idx].index = idx;
            idx++;
        }
    }
    memcpy(s->tris, s->tris_orig, sizeof(lo_tri_t) * n);

    bbox_of(s->bmin, s->bmax, boxes, n);
    add_margin(s->bmin, s->bmax);

    s->max_depth = 0;
    new_node(s);   /* root = 0 */
    construct(s, 0, s->bmin, s->bmax, boxes, scratch, 0, n, 0);

    free(boxes); free(scratch);
    return 0;
}

uint64_t lo_scene_ntriangles(const lo_scene_t *s) { return s->ntris; }

void lo_scene_bbox(const lo_scene_t *s, double bmin[3], double bmax[3])
{
    int k; for (k = 0; k < 3; k++) { bmin[k] = s->bmin[k]; bmax[k] = s->bmax[k]; }
}

void lo_scene_tree_stats(const lo_scene_t *s, lo_tree_stats_t *o)
{
    uint64_t i;
    memset(o, 0, sizeof(*o));
    o->ntriangles = s->ntris; o->max_depth = s->max_depth;
    for (i = 0; i < s->nnodes; i++) {
        if (s->nodes[i].is_leaf) {
            o->nleaf++;
            if (s->nodes[i].count > o->max_leaf_tris) o->max_leaf_tris = s->nodes[i].count;
        } else o->ninner++;
    }
}

void lo_scene_get_triangles(const lo_scene_t *s, double *v9, uint32_t *geom, uint32_t *index)
{
    uint64_t i; int c, k;
    for (i = 0; i < s->ntris; i++) {
        for (c = 0; c < 3; c++) for (k = 0; k < 3; k++) v9[9 * i + 3 * c + k] = s->tris_orig[i].v[c][k];
        if (geom) geom[i] = s->tris_orig[i].geom;
        if (index) index[i] = s->tris_orig[i].index;
    }
}

/* ------------------------------------------------------------- traverse */

typedef struct {
    double org[3], dir[3], invdir[3];
    int sign[3];
} lo_ray_t;

/* ray precompute bvh.c:473-497 (see header note about invdir[1]) */
static void ray_setup(lo_ray_t *r, const double *o, const double *d)
{
    int k;
    for (k = 0; k < 3; k++) {
        r->org[k] = o[k]; r->dir[k] = d[k];
        r->sign[k] = (d[k] < 0.0) ? 1 : 0;
        if (fabs(d[k]) > LO_EPS) r->invdir[k] = 1.0 / d[k];
        else                     r->invdir[k] = (d[k] < 0.0) ? -DBL_MAX : DBL_MAX;
    }
}

/* test_ray_aabb bvh.c:869-936; b[0..2]=bmin b[3..5]=bmax */
static int ray_aabb(double *tmin_out, double *tmax_out, const double *b, const lo_ray_t *r)
{
    double tmin, tmax;
    const double min_x = r->sign[0] ? b[3] : b[0];
    const double min_y = r->sign[1] ? b[4] : b[1];
    const double min_z = r->sign[2] ? b[5] : b[2];
    const double max_x = r->sign[0] ? b[0] : b[3];
    const double max_y = r->sign[1] ? b[1] : b[4];
    const double max_z = r->sign[2] ? b[2] : b[5];
    const double tmin_x = (min_x - r->org[0]) * r->invdir[0];
    const double tmax_x = (max_x - r->org[0]) * r->invdir[0];
    const double tmin_y = (min_y - r->org[1]) * r->invdir[1];
    const double tmax_y = (max_y - r->org[1]) * r->invdir[1];
    const double tmin_z = (min_z - r->org[2]) * r->invdir[2];
    const double tmax_z = (max_z - r->org[2]) * r->invdir[2];
    tmin = (tmin_x > tmin_y) ? tmin_x : tmin_y;
    tmax = (tmax_x < tmax_y) ? tmax_x : tmax_y;
    tmin = (tmin > tmin_z) ? tmin : tmin_z;
    tmax = (tmax < tmax_z) ? tmax : tmax_z;
    if ((tmax > 0.0) && (tmin <= tmax)) { *tmin_out = tmin; *tmax_out = tmax; return 1; }
    return 0;
}

/* triangle_isect bvh.c:730-791 */
static inline int tri_isect(uint32_t *tid, double *t_io, double *u_io, double *v_io,
                            const lo_tri_t *tri, const double *org, const double *dir,
                            uint32_t id)
{
    double e1[3], e2[3], p[3], s[3], q[3], a, inva, t, u, v;
    const double *v0 = tri->v[0], *v1 = tri->v[1], *v2 = tri->v[2];
    e1[0] = v1[0] - v0[0]; e1[1] = v1[1] - v0[1]; e1[2] = v1[2] - v0[2];
    e2[0] = v2[0] - v0[0]; e2[1] = v2[1] - v0[1]; e2[2] = v2[2] - v0[2];
    p[0] = dir[1] * e2[2] - dir[2] * e2[1];
    p[1] = dir[2] * e2[0] - dir[0] * e2[2];
    p[2] = dir[0] * e2[1] - dir[1] * e2[0];
    a = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (fabs(a) > LO_EPS) inva = 1.0 / a; else return 0;
    s[0] = org[0] - v0[0]; s[1] = org[1] - v0[1]; s[2] = org[2] - v0[2];
    q[0] = s[1] * e1[2] - s[2] * e1[1];
    q[1] = s[2] * e1[0] - s[0] * e1[2];
    q[2] = s[0] * e1[1] - s[1] * e1[0];
    u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inva;
    v = (q[0] * dir[0] + q[1] * dir[1] + q[2] * dir[2]) * inva;
    t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inva;
    if ((u < 0.0) || (u > 1.0)) return 0;
    if ((v < 0.0) || ((u + v) > 1.0)) return 0;
    if ((t < 0.0) || (t > *t_io)) return 0;
    *t_io = t; *u_io = u; *v_io = v; *tid = id;
    return 1;
}

typedef struct { double t, u, v; uint32_t prim; } lo_hit_t;

/* bvh_traverse bvh.c:1092-1188 + leaf :793-864 + node test :1030-1044,1080 */
static int traverse(const lo_scene_t *s, const lo_ray_t *r, lo_hit_t *h, lo_counters_t *c)
{
    int32_t stack[LO_MAXDEPTH + 1]; int depth = 0;
    const lo_node_t *node = &s->nodes[0];
    h->t = LO_INFINITY; h->u = 0.0; h->v = 0.0; h->prim = LO_MISS;

    for (;;) {
        if (node->is_leaf) {
            double t = LO_INFINITY, u = 0.0, v = 0.0; uint32_t tid = 0, i; int hitsum = 0;
            const lo_tri_t *tris = &s->tris[node->first];
            if (c) { c->nleaf_node_traversals++; c->ntested_triangles += node->count; }
            for (i = 0; i < node->count; i++) {
                int hit = tri_isect(&tid, &t, &u, &v, &tris[i], r->org, r->dir, i);
                if (hit && c) c->nactually_hit_triangles++;
                hitsum |= hit;
            }
            if (hitsum && (t < h->t)) { h->t = t; h->u = u; h->v = v; h->prim = tris[tid].prim; }
            if (depth < 1) break;
            node = &s->nodes[stack[--depth]];
        } else {
            double tmin_l, tmax_l, tmin_r, tmax_r; int ret = 0, order;
            int hl, hr;
            if (c) c->ninner_node_traversals++;
            hl = ray_aabb(&tmin_l, &tmax_l, node->box[0], r);
            hr = ray_aabb(&tmin_r, &tmax_r, node->box[1], r);
            if (hl && (tmin_l < h->t)) ret |= 1;
            if (hr && (tmin_r < h->t)) ret |= 2;
            order = r->sign[node->axis];
            if (ret == 0) {
                if (depth < 1) break;
                node = &s->nodes[stack[--depth]];
            } else if (ret == 1) node = &s->nodes[node->child[0]];
            else if (ret == 2)   node = &s->nodes[node->child[1]];
            else {
                stack[depth++] = node->child[1 - order];
                node = &s->nodes[node->child[order]];
            }
        }
    }
    return h->t < LO_INFINITY;
}

typedef struct {
    const lo_scene_t *s; size_t begin, end;
    const double *org, *dir; uint32_t *prim; double *t, *u, *v;
    lo_counters_t c; int want_counters; int brute;
} lo_job_t;

static void *job_run(void *arg)
{
    lo_job_t *j = (lo_job_t *)arg; size_t i;
    const lo_scene_t *s = j->s;
    for (i = j->begin; i < j->end; i++) {
        lo_hit_t h; lo_ray_t r;
        h.t = LO_INFINITY; h.u = 0.0; h.v = 0.0; h.prim = LO_MISS;
        if (j->brute) {
            uint64_t k; uint32_t tid = 0; int any = 0;
            for (k = 0; k < s->ntris; k++)
                any |= tri_isect(&tid, &h.t, &h.u, &h.v, &s->tris_orig[k],
                                 &j->org[3 * i], &j->dir[3 * i], (uint32_t)k);
            if (any) h.prim = tid; else { h.t = LO_INFINITY; }
        } else if (!s->empty) {               /* bvh.c:446-449 */
            double tmin, tmax;
            ray_setup(&r, &j->org[3 * i], &j->dir[3 * i]);
            if (j->want_counters) j->c.nrays++;
            {   /* scene bbox reject bvh.c:519-526 */
                double b[6]; int k;
                for (k = 0; k < 3; k++) { b[k] = s->bmin[k]; b[3 + k] = s->bmax[k]; }
                if (ray_aabb(&tmin, &tmax, b, &r))
                    traverse(s, &r, &h, j->want_counters ? &j->c : NULL);
            }
        }
        j->prim[i] = h.prim; j->t[i] = h.t; j->u[i] = h.u; j->v[i] = h.v;
    }
    return NULL;
}

static void run_batch(const lo_scene_t *s, size_t n, const double *org, const double *dir,
                      uint32_t *prim, double *t, double *u, double *v,
                      lo_counters_t *counters, int nthreads, int brute)
{
    int i; lo_job_t *jobs; pthread_t *th;
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n && n > 0) nthreads = (int)n;
    jobs = (lo_job_t *)calloc((size_t)nthreads, sizeof(lo_job_t));
    th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (i = 0; i < nthreads; i++) {
        jobs[i].s = s; jobs[i].begin = n * (size_t)i / (size_t)nthreads;
        jobs[i].end = n * (size_t)(i + 1) / (size_t)nthreads;
        jobs[i].org = org; jobs[i].dir = dir; jobs[i].prim = prim;
        jobs[i].t = t; jobs[i].u = u; jobs[i].v = v;
        jobs[i].want_counters = counters != NULL; jobs[i].brute = brute;
    }
    if (nthreads == 1) job_run(&jobs[0]);
    else {
        for (i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, job_run, &jobs[i]);
        for (i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    }
    if (counters) {
        memset(counters, 0, sizeof(*counters));
        for (i = 0; i < nthreads; i++) {
            counters->ninner_node_traversals += jobs[i].c.ninner_node_traversals;
            counters->nleaf_node_traversals += jobs[i].c.nleaf_node_traversals;
            counters->ntested_triangles += jobs[i].c.ntested_triangles;
            counters->nactually_hit_triangles += jobs[i].c.nactually_hit_triangles;
            counters->nrays += jobs[i].c.nrays;
        }
    }
    free(jobs); free(th);
}

void lo_intersect_batch(const lo_scene_t *s, size_t n, const double *org, const double *dir,
                        uint32_t *prim, double *t, double *u, double *v,
                        lo_counters_t *counters, int nthreads)
{
    run_batch(s, n, org, dir, prim, t, u, v, counters, nthreads, 0);
}

void lo_brute_force_batch(const lo_scene_t *s, size_t n, const double *org, const double *dir,
                          uint32_t *prim, double *t, double *u, double *v, int nthreads)
{
    run_batch(s, n, org, dir, prim, t, u, v, NULL, nthreads, 1);
}

/* primitive ids in the reference's leaf order (the order gather_triangles leaves them in,
 * bvh.c:1897-1917) and, per primitive, a leaf ordinal (first-triangle offset of its leaf) */
void lo_scene_leaf_order(const lo_scene_t *s, uint32_t *leaf_prims, uint32_t *prim_leaf_first)
{
    uint64_t i, k;
    for (i = 0; i < s->ntris; i++) leaf_prims[i] = s->tris[i].prim;
    for (i = 0; i < s->nnodes; i++)
        if (s->nodes[i].is_leaf)
            for (k = 0; k < s->nodes[i].count; k++)
                prim_leaf_first[s->tris[s->nodes[i].first + k].prim] = s->nodes[i].first;
}

/* ---- accessors for lucille_oracle_ao.c ---------------------------------- */

int lo_priv_intersect1(const lo_scene_t *s, const double *org, const double *dir,
                       uint32_t *prim, double *t, double *u, double *v)
{
    lo_hit_t h; lo_ray_t r; double tmin, tmax, b[6]; int k;
    h.t = LO_INFINITY; h.u = 0.0; h.v = 0.0; h.prim = LO_MISS;
    if (!s->empty) {
        ray_setup(&r, org, dir);
        for (k = 0; k < 3; k++) { b[k] = s->bmin[k]; b[3 + k] = s->bmax[k]; }
        if (ray_aabb(&tmin, &tmax, b, &r)) traverse(s, &r, &h, NULL);
    }
    *prim = h.prim; *t = h.t; *u = h.u; *v = h.v;
    return h.prim != LO_MISS;
}

/* vertex positions / normals of primitive `prim`: v[3][3]; n may come back NULL */
void lo_priv_prim_vertices(const lo_scene_t *s, uint32_t prim, const double **v0, const double **v1,
                           const double **v2, const double **n0, const double **n1, const double **n2,
                           int *inside_flag_two_side, uint32_t *index, uint32_t *nindices)
{
    const lo_tri_t *t = &s->tris_orig[prim];
    const lo_mesh_t *m = &s->meshes[t->geom];
    uint32_t i0 = m->idx[t->index], i1 = m->idx[t->index + 1], i2 = m->idx[t->index + 2];
    *v0 = &m->pos[3 * (size_t)i0]; *v1 = &m->pos[3 * (size_t)i1]; *v2 = &m->pos[3 * (size_t)i2];
    if (m->nrm) { *n0 = &m->nrm[3 * (size_t)i0]; *n1 = &m->nrm[3 * (size_t)i1]; *n2 = &m->nrm[3 * (size_t)i2]; }
    else { *n0 = *n1 = *n2 = NULL; }
    *inside_flag_two_side = m->two_side; *index = t->index; *nindices = m->nidx;
}

/* ---- accessors for lucille_oracle_beam.c --------------------------------- */
int lo_priv_empty(const lo_scene_t *s) { return s->empty; }

/* returns is_leaf */
int lo_priv_node(const lo_scene_t *s, int32_t idx, const double **box0, const double **box1, int32_t child[2],
                 uint32_t *first, uint32_t *count)
{
    const lo_node_t *n = &s->nodes[idx];
    *box0 = n->box[0]; *box1 = n->box[1]; child[0] = n->child[0]; child[1] = n->child[1];
    *first = n->first; *count = n->count;
    return n->is_leaf;
}

void lo_priv_leaf_tri(const lo_scene_t *s, uint32_t sorted_index, const double **v0, const double **v1, const double **v2)
{
    const lo_tri_t *t = &s->tris[sorted_index];
    *v0 = t->v[0]; *v1 = t->v[1]; *v2 = t->v[2];
}

/* number of triangles whose triangle_isect hit has EXACTLY t == t_ref[i]:
 * >= 2 means the reference's winner depends on its traversal order (leaf
 * order / first-visited-leaf rule, bvh.c:780,850), i.e. an exact-t tie. */
void lo_count_equal_t_batch(const lo_scene_t *s, size_t n, const double *org, const double *dir,
                            const double *t_ref, uint32_t *count)
{
    size_t i; uint64_t k;
    for (i = 0; i < n; i++) {
        uint32_t c = 0;
        for (k = 0; k < s->ntris; k++) {
            double t = LO_INFINITY, u = 0.0, v = 0.0; uint32_t tid = 0;
            if (tri_isect(&tid, &t, &u, &v, &s->tris_orig[k], &org[3 * i], &dir[3 * i], 0) && t == t_ref[i]) c++;
        }
        count[i] = c;
    }
}

/* ------------------------------------------------------ synthetic inputs */

/* SURVEY.md Appendix C generator (the build's own tooling, not reference code) */
static double xs_rnd(uint64_t *s)
{
    uint64_t x = *s;
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    *s = x;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

void lo_soup_triangles(uint64_t *state, uint32_t ntri, double sz, double *P, uint32_t *idx)
{
    uint32_t i; int k;
    for (i = 0; i < ntri; i++) {
        double cx = xs_rnd(state), cy = xs_rnd(state), cz = xs_rnd(state);
        for (k = 0; k < 3; k++) {
            size_t p = 3 * ((size_t)3 * i + k);
            P[p + 0] = cx + sz * (2 * xs_rnd(state) - 1);
            P[p + 1] = cy + sz * (2 * xs_rnd(state) - 1);
            P[p + 2] = cz + sz * (2 * xs_rnd(state) - 1);
            idx[3 * (size_t)i + k] = 3 * i + k;
        }
    }
}

void lo_soup_rays(uint64_t *state, size_t n, double *org, double *dir)
{
    size_t i;
    for (i = 0; i < n; i++) {
        double z, ph, r;
        org[3 * i + 0] = xs_rnd(state);
        org[3 * i + 1] = xs_rnd(state);
        org[3 * i + 2] = xs_rnd(state);
        z = 2 * xs_rnd(state) - 1;
        ph = 6.283185307179586 * xs_rnd(state);
        r = sqrt(1 - z * z);
        dir[3 * i + 0] = r * cos(ph);
        dir[3 * i + 1] = r * sin(ph);
        dir[3 * i + 2] = z;
    }
}
