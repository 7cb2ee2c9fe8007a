/*
 * lh_rib.c -- RIB-subset reader and Radiance .hdr writer: what a
 * `lsh scene.rib` run needs on either side of the ray-query path, without flex/bison.
 *
 * The reader produces exactly what lucille's RenderMan front end hands its renderer for the
 * verbs its example scenes use -- the ri_geom_t list (world-space double[4] positions and
 * normals, triangle indices, two_side) in RIB order and the camera -- so that the accelerator
 * is built over the same primitives (same global primitive ids) and the camera rays are the
 * same.  Restated from the reference (file:line = /root/reference/src/...):
 *
 *   numbers are C floats           lsh/lexrib.l:213 (atof) -> parserib.y:119 (float num)
 *   transform stack                ri/transform.c:43-110, ri/context.c:83-113,
 *                                  ri/attribute.c:74-160 (AttributeBegin pushes the CTM too)
 *   matrix mul / translate / scale / rotate / inverse
 *                                  base/matrix.c:47-395, base/quaternion.c:21-80
 *   WorldBegin                     ri/context.c:136-158 (CTM -> world_to_camera, push identity)
 *   Format / Projection / Orientation / PixelSamples / Sides / Display / Option
 *                                  ri/camera.c:360-438, ri/context.c:183-222,
 *                                  ri/attribute.c:350-357, ri/display.c:70-200, ri/option.c:430-560
 *   PointsPolygons                 render/polygon.c:495-640 (tri / quad -> 0 1 2, 0 2 3; Sides 2
 *                                  duplicates reversed faces; om = CTM x orientation)
 *   Polygon                        render/polygon.c:39-262 (fan; reversed when "rh")
 *   camera setup                   ri/camera.c:209-240
 *   .hdr display driver            display/hdrdrv.c:38-121, imageio/rgbe.c:78-96,118-140,241-345
 *
 * Everything else (shaders, lights, textures, colours, quadrics, subdivision) is outside the
 * ray-query path: such verbs are skipped and counted in `nskipped`.
 */
#include "lucille_hip.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

/* ------------------------------------------------------------------ errors -- */

static char g_rib_err[512];
const char *lh_rib_last_error(void) { return g_rib_err; }

#define RIB_FAIL(...) do { snprintf(g_rib_err, sizeof(g_rib_err), __VA_ARGS__); return -1; } while (0)

/* ---------------------------------------------------------------- matrices -- */

typedef struct { double f[4][4]; } mat4;

static void m_identity(mat4 *m)
{
    int i, j;
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) m->f[j][i] = (i == j) ? 1.0 : 0.0;
}

/* dst = a x b, each element accumulated left to right from zero (matrix.c:47-64) */
static void m_mul(mat4 *dst, const mat4 *a, const mat4 *b)
{
    int i, j, k; mat4 r;
    for (j = 0; j < 4; j++)
        for (i = 0; i < 4; i++) {
            double s = 0.0;
            for (k = 0; k < 4; k++) s += a->f[j][k] * b->f[k][i];
            r.f[j][i] = s;
        }
    *dst = r;
}

/* dst = op x dst : how Translate/Scale/Rotate/ConcatTransform compose (matrix.c:66-160) */
static void m_premul(mat4 *dst, const mat4 *op)
{
    mat4 old = *dst;
    m_mul(dst, op, &old);
}

/* ri_matrix_inverse (matrix.c:254-371): cofactor expansion whose pair products, transposed
 * source and determinant are RtFloat (float) temporaries while the elements are doubles; the
 * term tables below list (pair, source) indices in the reference's evaluation order, including
 * its (11, 0) term in the last element. */
static const unsigned char kPairsLo[12][2] = {
    {10, 15}, {11, 14}, {9, 15}, {11, 13}, {9, 14}, {10, 13}, {8, 15}, {11, 12}, {8, 14}, {10, 12}, {8, 13}, {9, 12}};
static const unsigned char kPairsHi[12][2] = {
    {2, 7}, {3, 6}, {1, 7}, {3, 5}, {1, 6}, {2, 5}, {0, 7}, {3, 4}, {0, 6}, {2, 4}, {0, 5}, {1, 4}};
/* per output element: 3 plus terms then 3 minus terms, each (pair index, source index) */
static const unsigned char kCof[16][12] = {
    {0, 5, 3, 6, 4, 7,      1, 5, 2, 6, 5, 7},
    {1, 4, 6, 6, 9, 7,      0, 4, 7, 6, 8, 7},
    {2, 4, 7, 5, 10, 7,     3, 4, 6, 5, 11, 7},
    {5, 4, 8, 5, 11, 6,     4, 4, 9, 5, 10, 6},
    {1, 1, 2, 2, 5, 3,      0, 1, 3, 2, 4, 3},
    {0, 0, 7, 2, 8, 3,      1, 0, 6, 2, 9, 3},
    {3, 0, 6, 1, 11, 3,     2, 0, 7, 1, 10, 3},
    {4, 0, 9, 1, 10, 2,     5, 0, 8, 1, 11, 2},
    {0, 13, 3, 14, 4, 15,   1, 13, 2, 14, 5, 15},
    {1, 12, 6, 14, 9, 15,   0, 12, 7, 14, 8, 15},
    {2, 12, 7, 13, 10, 15,  3, 12, 6, 13, 11, 15},
    {5, 12, 8, 13, 11, 14,  4, 12, 9, 13, 10, 14},
    {2, 10, 5, 11, 1, 9,    4, 11, 0, 9, 3, 10},
    {8, 11, 0, 8, 7, 10,    6, 10, 9, 11, 1, 8},
    {6, 9, 11, 11, 3, 8,    10, 11, 2, 8, 7, 9},
    {10, 10, 4, 8, 9, 9,    8, 9, 11, 0, 5, 8}};

static void m_inverse(mat4 *m)
{
    float pair[12], src[16], det; int i, j, e, k;
    for (i = 0; i < 4; i++)
        for (j = 0; j < 4; j++) src[i + 4 * j] = (float)m->f[i][j];
    for (e = 0; e < 16; e++) {
        float plus, minus; const unsigned char *c = kCof[e];
        if (e == 0 || e == 8) {
            const unsigned char (*pp)[2] = (e == 0) ? kPairsLo : kPairsHi;
            for (k = 0; k < 12; k++) pair[k] = src[pp[k][0]] * src[pp[k][1]];
        }
        plus  = pair[c[0]] * src[c[1]] + pair[c[2]] * src[c[3]] + pair[c[4]] * src[c[5]];
        minus = pair[c[6]] * src[c[7]] + pair[c[8]] * src[c[9]] + pair[c[10]] * src[c[11]];
        m->f[e >> 2][e & 3] = (double)plus;
        m->f[e >> 2][e & 3] -= (double)minus;
    }
    det = (float)(src[0] * m->f[0][0] + src[1] * m->f[0][1] + src[2] * m->f[0][2] + src[3] * m->f[0][3]);
    det = 1.0f / det;
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) m->f[j][i] *= det;
}

static void m_transpose(mat4 *m)
{
    int i, j; mat4 t = *m;
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) m->f[i][j] = t.f[j][i];
}

/* row vector x matrix with w = 1, accumulated from zero (vector.h:182-210) */
static void v_transform(double dst[4], const double src[3], const mat4 *m)
{
    const double v[4] = {src[0], src[1], src[2], 1.0}; int i, j;
    for (j = 0; j < 4; j++) {
        double s = 0.0;
        for (i = 0; i < 4; i++) s += v[i] * m->f[i][j];
        dst[j] = s;
    }
}

static void v_normalize(double d[4])
{   /* ri_vector_normalize (vector.h:75-86): the threshold is the float literal 1.0e-17f */
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (n2 > 1.0e-17f) { const double r = 1.0 / sqrt(n2); d[0] *= r; d[1] *= r; d[2] *= r; }
}

/* Rotate: quaternion of (-angle) about the normalised axis, then its matrix
 * (matrix.c:83-101, quaternion.c:21-80) */
static void m_rotation(mat4 *r, float angle, float ax, float ay, float az)
{
    const double deg2rad = (double)M_PI / 180.0;
    double n[4] = {ax, ay, az, 0.0};
    const double s = sin(-deg2rad * angle / 2.0), c = cos(-deg2rad * angle / 2.0);
    double qx, qy, qz, qw, norm, k, xs, ys, zs, wx, wy, wz, xx, xy, xz, yy, yz, zz;
    v_normalize(n);
    qx = n[0] * s; qy = n[1] * s; qz = n[2] * s; qw = c;
    norm = qx * qx + qy * qy + qz * qz + qw * qw;
    k = (norm > 0.0) ? 2.0 / norm : 0.0;
    xs = qx * k; ys = qy * k; zs = qz * k;
    wx = qw * xs; wy = qw * ys; wz = qw * zs;
    xx = qx * xs; xy = qx * ys; xz = qx * zs;
    yy = qy * ys; yz = qy * zs; zz = qz * zs;
    m_identity(r);
    r->f[0][0] = 1.0 - (yy + zz); r->f[0][1] = xy - wz;         r->f[0][2] = xz + wy;
    r->f[1][0] = xy + wz;         r->f[1][1] = 1.0 - (xx + zz); r->f[1][2] = yz - wx;
    r->f[2][0] = xz - wy;         r->f[2][1] = yz + wx;         r->f[2][2] = 1.0 - (xx + yy);
}

/* --------------------------------------------------------------- tokenizer -- */

enum { T_END = 0, T_WORD, T_STR, T_NUM, T_LBR, T_RBR };

typedef struct {
    const char *p, *end;
    int   kind;
    float num;
    char *text;      /* word / string payload (grows) */
    size_t cap;
    int   line;
} lexer_t;

static int lex_reserve(lexer_t *lx, size_t n)
{
    if (n + 1 > lx->cap) {
        size_t nc = lx->cap ? lx->cap * 2 : 256; char *t;
        while (nc < n + 1) nc *= 2;
        t = (char *)realloc(lx->text, nc);
        if (!t) return -1;
        lx->text = t; lx->cap = nc;
    }
    return 0;
}

static int is_number_start(const char *p, const char *end)
{
    if (p < end && (*p == '-' || *p == '+')) p++;
    if (p < end && *p == '.') p++;
    return p < end && isdigit((unsigned char)*p);
}

static int lex_next(lexer_t *lx)
{
    const char *p = lx->p, *end = lx->end;
    for (;;) {
        while (p < end && isspace((unsigned char)*p)) { if (*p == '\n') lx->line++; p++; }
        if (p < end && *p == '#') { while (p < end && *p != '\n') p++; continue; }
        break;
    }
    if (p >= end) { lx->p = p; lx->kind = T_END; return T_END; }
    if (*p == '[') { lx->p = p + 1; return lx->kind = T_LBR; }
    if (*p == ']') { lx->p = p + 1; return lx->kind = T_RBR; }
    if (*p == '"') {
        const char *q = ++p; size_t n;
        while (q < end && *q != '"') q++;
        n = (size_t)(q - p);
        if (lex_reserve(lx, n) != 0) return lx->kind = T_END;
        memcpy(lx->text, p, n); lx->text[n] = 0;
        lx->p = (q < end) ? q + 1 : q;
        return lx->kind = T_STR;
    }
    if (is_number_start(p, end)) {
        char *stop = NULL;
        const double d = strtod(p, &stop);        /* atof (lexrib.l:213) ... */
        lx->num = (float)d;                        /* ... stored in a float (parserib.y:119) */
        lx->p = (stop && stop > p) ? stop : p + 1;
        return lx->kind = T_NUM;
    }
    {
        const char *q = p; size_t n;
        while (q < end && !isspace((unsigned char)*q) && *q != '[' && *q != ']' && *q != '"' && *q != '#') q++;
        n = (size_t)(q - p);
        if (n == 0) { lx->p = p + 1; return lex_next(lx); }
        if (lex_reserve(lx, n) != 0) return lx->kind = T_END;
        memcpy(lx->text, p, n); lx->text[n] = 0;
        lx->p = q;
        return lx->kind = T_WORD;
    }
}

/* one argument of a request: a number, a string, or an array of either */
typedef struct {
    int     is_array, is_string;
    size_t  n;
    float  *num;      /* n floats (n == 1 for a scalar)  */
    char  **str;      /* n strings                       */
} arg_t;

typedef struct { arg_t *a; size_t n, cap; } arglist_t;

static void args_clear(arglist_t *L)
{
    size_t i, k;
    for (i = 0; i < L->n; i++) {
        free(L->a[i].num);
        if (L->a[i].str) { for (k = 0; k < L->a[i].n; k++) free(L->a[i].str[k]); free(L->a[i].str); }
    }
    L->n = 0;
}

static arg_t *args_push(arglist_t *L)
{
    if (L->n == L->cap) {
        size_t nc = L->cap ? 2 * L->cap : 16; arg_t *t = (arg_t *)realloc(L->a, nc * sizeof(arg_t));
        if (!t) return NULL;
        L->a = t; L->cap = nc;
    }
    memset(&L->a[L->n], 0, sizeof(arg_t));
    return &L->a[L->n++];
}

static int arg_add_num(arg_t *a, float v, size_t *cap)
{
    if (a->n == *cap) {
        size_t nc = *cap ? 2 * *cap : 16; float *t = (float *)realloc(a->num, nc * sizeof(float));
        if (!t) return -1;
        a->num = t; *cap = nc;
    }
    a->num[a->n++] = v;
    return 0;
}

static int arg_add_str(arg_t *a, const char *s, size_t *cap)
{
    if (a->n == *cap) {
        size_t nc = *cap ? 2 * *cap : 4; char **t = (char **)realloc(a->str, nc * sizeof(char *));
        if (!t) return -1;
        a->str = t; *cap = nc;
    }
    a->str[a->n] = strdup(s);
    if (!a->str[a->n]) return -1;
    a->n++; a->is_string = 1;
    return 0;
}

/* reads the arguments that follow a request name; leaves the lexer ON the next request */
static int read_args(lexer_t *lx, arglist_t *L)
{
    args_clear(L);
    for (;;) {
        const int k = lex_next(lx);
        arg_t *a; size_t cap = 0;
        if (k == T_END || k == T_WORD) return 0;
        if (k == T_RBR) continue;                  /* stray bracket: ignore */
        a = args_push(L);
        if (!a) return -1;
        if (k == T_NUM) { if (arg_add_num(a, lx->num, &cap) != 0) return -1; }
        else if (k == T_STR) { if (arg_add_str(a, lx->text, &cap) != 0) return -1; }
        else {                                     /* T_LBR */
            a->is_array = 1;
            for (;;) {
                const int e = lex_next(lx);
                if (e == T_RBR) break;
                if (e == T_END || e == T_WORD || e == T_LBR) return -2;     /* unterminated array */
                if (e == T_NUM) { if (a->is_string || arg_add_num(a, lx->num, &cap) != 0) return -2; }
                else { if ((a->n && !a->is_string) || arg_add_str(a, lx->text, &cap) != 0) return -2; }
            }
        }
    }
}

/* ------------------------------------------------------------------- scene -- */

typedef struct {
    uint32_t  npos, nidx;
    double   *pos;        /* npos x double[4] */
    double   *nrm;        /* npos x double[4] or NULL */
    uint32_t *idx;
    int       two_side;
} rib_mesh_t;

struct lh_rib_scene {
    rib_mesh_t *mesh; uint32_t nmesh, capmesh;
    /* graphics state */
    mat4   stack[64]; int depth;                   /* CTM stack (top = stack[depth]) */
    int    sides[64]; int adepth;                  /* attribute stack: Sides */
    int    is_attr[128]; int nblocks;              /* open Attribute/Transform blocks, for matching Ends */
    mat4   world_to_camera;
    int    world_begun, world_ended;
    int    rh;
    /* options */
    int    xres, yres; float fov; int perspective;
    float  xsamples, ysamples;
    int    gather_nsamples, accel_method, nthreads;
    char   display_name[1024], display_type[64];
    char   searchpath[2048];
    char   top_dir[1024];
    uint32_t nskipped, nrequests, nunknown;
    char  *log; size_t loglen, logcap;         /* what lsh prints to stdout while parsing */
};

static mat4 *ctm(lh_rib_scene_t *s) { return &s->stack[s->depth]; }

static void rib_log(lh_rib_scene_t *s, const char *fmt, const char *arg)
{
    char line[600]; size_t n;
    snprintf(line, sizeof(line), fmt, arg);
    n = strlen(line);
    if (s->loglen + n + 2 > s->logcap) {
        size_t nc = s->logcap ? 2 * s->logcap : 1024; char *t;
        while (nc < s->loglen + n + 2) nc *= 2;
        t = (char *)realloc(s->log, nc);
        if (!t) return;
        s->log = t; s->logcap = nc;
    }
    memcpy(s->log + s->loglen, line, n); s->loglen += n;
    s->log[s->loglen++] = '\n'; s->log[s->loglen] = 0;
}

static rib_mesh_t *new_mesh(lh_rib_scene_t *s)
{
    if (s->nmesh == s->capmesh) {
        uint32_t nc = s->capmesh ? 2 * s->capmesh : 16;
        rib_mesh_t *t = (rib_mesh_t *)realloc(s->mesh, nc * sizeof(rib_mesh_t));
        if (!t) return NULL;
        s->mesh = t; s->capmesh = nc;
    }
    memset(&s->mesh[s->nmesh], 0, sizeof(rib_mesh_t));
    return &s->mesh[s->nmesh++];
}

/* om = CTM x orientation (polygon.c:533-545) */
static void object_matrix(lh_rib_scene_t *s, mat4 *om)
{
    mat4 o; m_identity(&o);
    if (s->rh) o.f[2][2] = -o.f[2][2];
    m_mul(om, ctm(s), &o);
}

/* inverse transpose of the upper 3x3 of om (polygon.c:618-634) */
static void normal_matrix(const mat4 *om, mat4 *itm)
{
    *itm = *om;
    itm->f[0][3] = itm->f[1][3] = itm->f[2][3] = 0.0;
    itm->f[3][0] = itm->f[3][1] = itm->f[3][2] = 0.0; itm->f[3][3] = 1.0;
    m_inverse(itm);
    m_transpose(itm);
}

static const arg_t *find_param(const arglist_t *L, size_t first, const char *name, const char *alt)
{
    size_t i;
    for (i = first; i + 1 < L->n; i += 2) {
        const arg_t *k = &L->a[i];
        if (!k->is_string || k->n != 1) continue;
        if (strcmp(k->str[0], name) == 0 || (alt && strcmp(k->str[0], alt) == 0)) return &L->a[i + 1];
    }
    return NULL;
}

static int fill_vertices(lh_rib_scene_t *s, rib_mesh_t *m, uint32_t nv, const arg_t *P, const arg_t *N, int two)
{
    mat4 om, itm; uint32_t j; const uint32_t total = two ? 2 * nv : nv;
    object_matrix(s, &om);
    if (!P || P->is_string || P->n < (size_t)3 * nv) RIB_FAIL("polygon: \"P\" missing or shorter than the vertex count");
    m->pos = (double *)calloc((size_t)total * 4, sizeof(double));
    if (!m->pos) RIB_FAIL("out of memory");
    for (j = 0; j < nv; j++) {
        const double v[3] = {P->num[3 * j], P->num[3 * j + 1], P->num[3 * j + 2]};
        v_transform(&m->pos[4 * j], v, &om);
        if (two) memcpy(&m->pos[4 * (nv + j)], &m->pos[4 * j], 4 * sizeof(double));
    }
    m->npos = total; m->two_side = two;
    if (N && !N->is_string && N->n >= (size_t)3 * nv) {
        normal_matrix(&om, &itm);
        m->nrm = (double *)calloc((size_t)total * 4, sizeof(double));
        if (!m->nrm) RIB_FAIL("out of memory");
        for (j = 0; j < nv; j++) {
            const double v[3] = {N->num[3 * j], N->num[3 * j + 1], N->num[3 * j + 2]};
            double *d = &m->nrm[4 * j];
            v_transform(d, v, &itm);
            v_normalize(d);
            if (two) { double *e = &m->nrm[4 * (nv + j)]; e[0] = -d[0]; e[1] = -d[1]; e[2] = -d[2]; e[3] = -d[3]; }
        }
    }
    return 0;
}

/* PointsPolygons nverts[] verts[] params (polygon.c:495-640) */
static int do_points_polygons(lh_rib_scene_t *s, const arglist_t *L)
{
    const arg_t *nv, *vs; size_t i, j = 0, k, npolys; uint32_t nvertices = 0, nidx = 0, cap; uint32_t *idx;
    static const int order[6] = {0, 1, 2, 0, 2, 3};
    int warned = 0; rib_mesh_t *m; const int two = s->sides[s->adepth] == 2;
    if (L->n < 2 || L->a[0].is_string || L->a[1].is_string) RIB_FAIL("PointsPolygons: expected two integer arrays");
    nv = &L->a[0]; vs = &L->a[1]; npolys = nv->n;
    if (npolys == 0) return 0;                    /* polygon.c:526: no geometry */
    cap = 0;
    for (i = 0; i < npolys; i++) cap += 6;
    idx = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cap * 2 + 8);
    if (!idx) RIB_FAIL("out of memory");
    for (i = 0; i < npolys; i++) {
        const int n = (int)nv->num[i];
        if (n > 4 && !warned) { warned = 1; continue; }          /* polygon.c:553-560, quirk kept: j not advanced */
        if (j + (size_t)(n > 0 ? n : 0) > vs->n) {  /* tests/ribparse/indices_shortage: not fatal */
            free(idx); rib_log(s, "warning: PointsPolygons: %s, primitive ignored", "vertex index list shorter than the face list"); return 0;
        }
        for (k = 0; k < (size_t)(n > 0 ? n : 0); k++) {
            const float fv = vs->num[j + k];
            if (!(fv >= 0.0f) || fv > 1.0e9f) {          /* negative / absurd / NaN vertex index */
                free(idx); rib_log(s, "warning: PointsPolygons: %s, primitive ignored", "vertex index out of range"); return 0;
            }
            if (nvertices < (uint32_t)fv) nvertices = (uint32_t)fv;
        }
        if (n == 3) { for (k = 0; k < 3; k++) idx[nidx + k] = (uint32_t)(int)vs->num[j + order[k]]; nidx += 3; }
        else {
            if (j + 4 > vs->n) {
                free(idx); rib_log(s, "warning: PointsPolygons: %s, primitive ignored", "vertex index list shorter than the face list"); return 0;
            }
            for (k = 0; k < 6; k++) idx[nidx + k] = (uint32_t)(int)vs->num[j + order[k]];
            nidx += 6;
        }
        j += (size_t)(n > 0 ? n : 0);
    }
    nvertices++;
    if (two) {                                    /* reversed copies offset by nvertices (polygon.c:583-604) */
        for (i = 0; i < nidx / 3; i++)
            for (k = 0; k < 3; k++) idx[nidx + 3 * i + k] = idx[3 * i + 2 - k] + nvertices;
        nidx *= 2;
    }
    {
        const arg_t *P = find_param(L, 2, "P", NULL);
        if (nidx == 0 || !P || P->is_string || P->n < (size_t)3 * nvertices) {
            free(idx); rib_log(s, "warning: PointsPolygons: %s, primitive ignored", "\"P\" missing or shorter than the vertex count"); return 0;
        }
        m = new_mesh(s);
        if (!m) { free(idx); RIB_FAIL("out of memory"); }
        m->idx = idx; m->nidx = nidx;
        return fill_vertices(s, m, nvertices, P, find_param(L, 2, "N", "vertex normal N"), two);
    }
}

/* Polygon params: one convex polygon as a fan (polygon.c:39-262) */
static int do_polygon(lh_rib_scene_t *s, const arglist_t *L)
{
    const arg_t *P = find_param(L, 0, "P", NULL); uint32_t nverts, nidx, j, *idx; rib_mesh_t *m;
    const int two = s->sides[s->adepth] == 2;
    if (!P || P->is_string) RIB_FAIL("Polygon: \"P\" missing");
    nverts = (uint32_t)(P->n / 3);
    if (nverts == 0) return 0;
    if (nverts < 3) RIB_FAIL("Polygon: fewer than 3 vertices");
    nidx = (two ? 6 : 3) * (nverts - 2);
    idx = (uint32_t *)malloc(sizeof(uint32_t) * nidx);
    if (!idx) RIB_FAIL("out of memory");
    for (j = 0; j < nverts - 2; j++) {
        if (s->rh) { idx[3 * j] = j + 2; idx[3 * j + 1] = j + 1; idx[3 * j + 2] = 0; }
        else       { idx[3 * j] = 0;     idx[3 * j + 1] = j + 1; idx[3 * j + 2] = j + 2; }
    }
    if (two) {
        uint32_t *b = idx + nidx / 2; const uint32_t nv = nverts;
        for (j = 0; j < nverts - 2; j++) {
            if (s->rh) { b[3 * j] = nv;         b[3 * j + 1] = nv + j + 1; b[3 * j + 2] = nv + j + 2; }
            else       { b[3 * j] = nv + j + 2; b[3 * j + 1] = nv + j + 1; b[3 * j + 2] = nv + j; }
        }
    }
    m = new_mesh(s);
    if (!m) { free(idx); RIB_FAIL("out of memory"); }
    m->idx = idx; m->nidx = nidx;
    return fill_vertices(s, m, nverts, P, find_param(L, 0, "N", NULL), two);
}

static int matrix_arg(const arglist_t *L, mat4 *m)
{
    const arg_t *a = L->n ? &L->a[0] : NULL; int i, j;
    if (!a || a->is_string || a->n < 16) return -1;
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) m->f[j][i] = a->num[4 * j + i];
    return 0;
}

/* scalars may arrive as N separate numbers or one array */
static int scalars(const arglist_t *L, float *out, size_t want)
{
    size_t i, got = 0;
    for (i = 0; i < L->n && got < want; i++) {
        const arg_t *a = &L->a[i]; size_t k;
        if (a->is_string) break;
        for (k = 0; k < a->n && got < want; k++) out[got++] = a->num[k];
    }
    return got == want ? 0 : -1;
}

static int casecmp_(const char *a, const char *b)
{
    for (; *a && *b; a++, b++) if (tolower((unsigned char)*a) != tolower((unsigned char)*b)) return 1;
    return *a || *b;
}

/* Display name type mode (display.c:70-200): "file" always ends in .hdr here (no libtiff) */
static void do_display(lh_rib_scene_t *s, const arglist_t *L)
{
    const char *name, *type; char *ext;
    if (L->n < 2 || !L->a[0].is_string || !L->a[1].is_string) return;
    name = L->a[0].str[0]; type = L->a[1].str[0];
    if (name[0] == '+') return;                   /* additional displays are not rendered */
    snprintf(s->display_name, sizeof(s->display_name) - 8, "%s", name);
    snprintf(s->display_type, sizeof(s->display_type), "%s", type);
    if (casecmp_(type, "file") == 0) {
        ext = strrchr(s->display_name, '.');
        if (!ext) strcat(s->display_name, ".hdr");
        else if (casecmp_(ext, ".hdr") != 0) { strcpy(ext, ".hdr"); snprintf(s->display_type, sizeof(s->display_type), "hdr"); }
    }
}

static void do_option(lh_rib_scene_t *s, const arglist_t *L)
{
    const char *name; size_t i;
    if (L->n < 1 || !L->a[0].is_string) return;
    name = L->a[0].str[0];
    for (i = 1; i + 1 < L->n; i += 2) {
        const arg_t *k = &L->a[i], *v = &L->a[i + 1]; const char *key;
        if (!k->is_string || k->n != 1) continue;
        key = k->str[0];
        /* inline declarations ("string accel_method"): the name is the last word */
        if (strrchr(key, ' ')) key = strrchr(key, ' ') + 1;
        if (strcmp(name, "raytrace") == 0) {
            if (strcmp(key, "accel_method") == 0 && v->is_string && v->n) {      /* option.c:453-462 */
                if (strcmp(v->str[0], "grid") == 0) s->accel_method = 0;
                else if (strcmp(v->str[0], "bvh") == 0) s->accel_method = 1;
                else if (strcmp(v->str[0], "hip") == 0) s->accel_method = 2;      /* INTEGRATION.md */
            } else if (strcmp(key, "nthreads") == 0 && !v->is_string && v->n) s->nthreads = (int)v->num[0];
        } else if (strcmp(name, "gather") == 0) {
            if (strcmp(key, "nsamples") == 0 && !v->is_string && v->n) s->gather_nsamples = (int)v->num[0];   /* option.c:545-549 */
        } else if (strcmp(name, "searchpath") == 0) {
            if (strcmp(key, "archive") == 0 && v->is_string && v->n) snprintf(s->searchpath, sizeof(s->searchpath), "%s", v->str[0]);
        }
    }
}

static int parse_file(lh_rib_scene_t *s, const char *path, int depth);

static int file_exists(const char *p)
{
    struct stat st;
    return access(p, R_OK) == 0 && stat(p, &st) == 0 && S_ISREG(st.st_mode);
}

static void dir_of(const char *path, char *out, size_t n)
{
    const char *sl = strrchr(path, '/');
    if (!sl) snprintf(out, n, ".");
    else { size_t k = (size_t)(sl - path); if (k >= n) k = n - 1; memcpy(out, path, k); out[k] = 0; if (k == 0) snprintf(out, n, "/"); }
}

/* ReadArchive: next to the including file, then each Option "searchpath" "archive" entry
 * (':'-separated) taken relative to the working directory and to the top-level RIB */
static int do_read_archive(lh_rib_scene_t *s, const arglist_t *L, const char *from, int depth)
{
    char dir[1024], cand[4096], sp[2048]; const char *name; char *tok, *save = NULL;
    if (L->n < 1 || !L->a[0].is_string) RIB_FAIL("ReadArchive: missing file name");
    name = L->a[0].str[0];
    if (depth > 16) RIB_FAIL("ReadArchive: nesting deeper than 16 (%s)", name);
    if (name[0] == '/' && file_exists(name)) return parse_file(s, name, depth + 1);
    dir_of(from, dir, sizeof(dir));
    snprintf(cand, sizeof(cand), "%s/%s", dir, name);
    if (file_exists(cand)) return parse_file(s, cand, depth + 1);
    snprintf(cand, sizeof(cand), "%s/%s", s->top_dir, name);
    if (file_exists(cand)) return parse_file(s, cand, depth + 1);
    snprintf(sp, sizeof(sp), "%s", s->searchpath);
    for (tok = strtok_r(sp, ":", &save); tok; tok = strtok_r(NULL, ":", &save)) {
        if (tok[0] == '@' || tok[0] == '&') continue;
        snprintf(cand, sizeof(cand), "%s/%s", tok, name);
        if (file_exists(cand)) return parse_file(s, cand, depth + 1);
        snprintf(cand, sizeof(cand), "%s/%s/%s", s->top_dir, tok, name);
        if (file_exists(cand)) return parse_file(s, cand, depth + 1);
    }
    RIB_FAIL("ReadArchive: cannot find \"%s\"", name);
}

static int push_ctm(lh_rib_scene_t *s)
{
    if (s->depth + 1 >= 64) RIB_FAIL("transform stack overflow");
    s->stack[s->depth + 1] = s->stack[s->depth]; s->depth++;
    return 0;
}

static int dispatch(lh_rib_scene_t *s, const char *verb, const arglist_t *L, const char *path, int depth)
{
    float f[4]; mat4 m;
    s->nrequests++;
    if (strcmp(verb, "PointsPolygons") == 0) return do_points_polygons(s, L);
    if (strcmp(verb, "Polygon") == 0) return do_polygon(s, L);
    if (strcmp(verb, "Transform") == 0) { if (matrix_arg(L, &m) != 0) RIB_FAIL("Transform: expected 16 numbers"); *ctm(s) = m; return 0; }
    if (strcmp(verb, "ConcatTransform") == 0) { if (matrix_arg(L, &m) != 0) RIB_FAIL("ConcatTransform: expected 16 numbers"); m_premul(ctm(s), &m); return 0; }
    if (strcmp(verb, "Identity") == 0) { m_identity(ctm(s)); return 0; }
    if (strcmp(verb, "Translate") == 0) {
        if (scalars(L, f, 3) != 0) RIB_FAIL("Translate: expected 3 numbers");
        m_identity(&m); m.f[3][0] = f[0]; m.f[3][1] = f[1]; m.f[3][2] = f[2]; m_premul(ctm(s), &m); return 0;
    }
    if (strcmp(verb, "Scale") == 0) {
        if (scalars(L, f, 3) != 0) RIB_FAIL("Scale: expected 3 numbers");
        m_identity(&m); m.f[0][0] = f[0]; m.f[1][1] = f[1]; m.f[2][2] = f[2]; m_premul(ctm(s), &m); return 0;
    }
    if (strcmp(verb, "Rotate") == 0) {
        if (scalars(L, f, 4) != 0) RIB_FAIL("Rotate: expected 4 numbers");
        m_rotation(&m, f[0], f[1], f[2], f[3]); m_premul(ctm(s), &m); return 0;
    }
    if (strcmp(verb, "AttributeBegin") == 0 || strcmp(verb, "TransformBegin") == 0) {
        const int attr = verb[0] == 'A';
        if (s->nblocks >= 128) RIB_FAIL("block nesting deeper than 128");
        if (push_ctm(s) != 0) return -1;
        if (attr) { if (s->adepth + 1 >= 64) RIB_FAIL("attribute stack overflow"); s->sides[s->adepth + 1] = s->sides[s->adepth]; s->adepth++; }
        s->is_attr[s->nblocks++] = attr;
        return 0;
    }
    if (strcmp(verb, "AttributeEnd") == 0) {
        if (s->adepth < 1) return 0;              /* attribute.c:133-137: warns, ignores */
        s->adepth--; if (s->depth > 0) s->depth--; if (s->nblocks > 0) s->nblocks--;
        return 0;
    }
    if (strcmp(verb, "TransformEnd") == 0) { if (s->depth > 0) s->depth--; if (s->nblocks > 0) s->nblocks--; return 0; }
    if (strcmp(verb, "Sides") == 0) { if (scalars(L, f, 1) != 0) RIB_FAIL("Sides: expected a number"); s->sides[s->adepth] = (int)f[0]; return 0; }
    if (strcmp(verb, "WorldBegin") == 0) {        /* context.c:136-158 */
        s->world_to_camera = *ctm(s); s->world_begun = 1;
        if (push_ctm(s) != 0) return -1;
        m_identity(ctm(s));
        return 0;
    }
    if (strcmp(verb, "WorldEnd") == 0) { s->world_ended = 1; return 0; }
    if (strcmp(verb, "Format") == 0) {            /* camera.c:360-380 (incl. its yres<0 -> xres slip) */
        int xr, yr;
        if (scalars(L, f, 3) != 0) RIB_FAIL("Format: expected 3 numbers");
        xr = (int)f[0]; yr = (int)f[1];
        if (xr < 0) xr = 640;
        if (yr < 0) xr = 480;
        s->xres = xr; s->yres = yr;
        return 0;
    }
    if (strcmp(verb, "Projection") == 0) {        /* camera.c:413-432 */
        const arg_t *fov;
        if (L->n < 1 || !L->a[0].is_string) RIB_FAIL("Projection: expected a name");
        if (strcmp(L->a[0].str[0], "perspective") != 0 && strcmp(L->a[0].str[0], "orthographic") != 0) return 0;
        fov = find_param(L, 1, "fov", "float fov");
        if (fov && !fov->is_string && fov->n) { s->fov = fov->num[0]; s->perspective = 1; }
        return 0;
    }
    if (strcmp(verb, "Orientation") == 0) {       /* context.c:183-194 */
        if (L->n >= 1 && L->a[0].is_string) {
            if (strcmp(L->a[0].str[0], "rh") == 0) s->rh = 1;
            else if (strcmp(L->a[0].str[0], "lh") == 0) s->rh = 0;
        }
        return 0;
    }
    if (strcmp(verb, "PixelSamples") == 0) {      /* context.c:210-222 */
        if (scalars(L, f, 2) != 0) RIB_FAIL("PixelSamples: expected 2 numbers");
        s->xsamples = f[0] < 1.0f ? 1.0f : f[0]; s->ysamples = f[1] < 1.0f ? 1.0f : f[1];
        return 0;
    }
    if (strcmp(verb, "Display") == 0) { do_display(s, L); return 0; }
    if (strcmp(verb, "Option") == 0) { do_option(s, L); return 0; }
    if (strcmp(verb, "ReadArchive") == 0) return do_read_archive(s, L, path, depth);
    if (strcmp(verb, "FrameBegin") == 0 || strcmp(verb, "FrameEnd") == 0 || strcmp(verb, "version") == 0) return 0;
    {   /* requests of the RenderMan interface that do not touch the ray-query path */
        static const char *const known[] = {
            "Surface", "Displacement", "Atmosphere", "Interior", "Exterior", "Imager", "LightSource", "AreaLightSource",
            "Illuminate", "Color", "Opacity", "ShadingRate", "ShadingInterpolation", "Matte", "Attribute", "Declare",
            "Shutter", "Exposure", "Quantize", "PixelFilter", "PixelVariance", "Hider", "Clipping", "ClippingPlane",
            "CropWindow", "ScreenWindow", "FrameAspectRatio", "DepthOfField", "ColorSamples", "RelativeDetail", "Bound",
            "Detail", "DetailRange", "GeometricApproximation", "ReverseOrientation", "TextureCoordinates", "Basis",
            "Perspective", "Skew", "CoordinateSystem", "CoordSysTransform", "MotionBegin", "MotionEnd", "SolidBegin",
            "SolidEnd", "ObjectBegin", "ObjectEnd", "ObjectInstance", "MakeTexture", "MakeLatLongEnvironment",
            "MakeCubeFaceEnvironment", "MakeShadow", "ErrorHandler", "ArchiveRecord", "Procedural", "Geometry",
            "Sphere", "Cone", "Cylinder", "Hyperboloid", "Paraboloid", "Disk", "Torus", "Points", "Curves", "Blobby",
            "Patch", "PatchMesh", "NuPatch", "TrimCurve", "SubdivisionMesh", "GeneralPolygon", "PointsGeneralPolygons",
            NULL};
        int k;
        for (k = 0; known[k]; k++) if (strcmp(verb, known[k]) == 0) break;
        if (!known[k]) { s->nunknown++; rib_log(s, "Unknown RIB command: %s", verb); }   /* tests/ribparse/unknown_protocol */
        s->nskipped++;
    }
    return 0;
}

static int parse_file(lh_rib_scene_t *s, const char *path, int depth)
{
    FILE *fp = fopen(path, "rb"); long size; char *buf; lexer_t lx; arglist_t L; int rc = 0; char *verb = NULL;
    if (!fp) RIB_FAIL("cannot open \"%s\"", path);
    if (fseek(fp, 0, SEEK_END) != 0 || (size = ftell(fp)) < 0 || size > (1L << 40) || fseek(fp, 0, SEEK_SET) != 0) {
        fclose(fp); RIB_FAIL("\"%s\" is not a readable regular file", path);       /* e.g. a directory */
    }
    buf = (char *)malloc((size_t)size + 1);
    if (!buf) { fclose(fp); RIB_FAIL("out of memory"); }
    if (size > 0 && fread(buf, 1, (size_t)size, fp) != (size_t)size) { fclose(fp); free(buf); RIB_FAIL("short read on \"%s\"", path); }
    fclose(fp); buf[size] = 0;
    memset(&lx, 0, sizeof(lx)); memset(&L, 0, sizeof(L));
    lx.p = buf; lx.end = buf + size; lx.line = 1;
    lex_next(&lx);
    while (lx.kind != T_END && !s->world_ended) {
        int line = lx.line;
        if (lx.kind != T_WORD) { lex_next(&lx); continue; }     /* stray token between requests */
        free(verb); verb = strdup(lx.text);
        rc = read_args(&lx, &L);
        if (rc == -2) { snprintf(g_rib_err, sizeof(g_rib_err), "%s:%d: malformed array in %s", path, line, verb); rc = -1; break; }
        if (rc != 0) { snprintf(g_rib_err, sizeof(g_rib_err), "%s:%d: out of memory", path, line); break; }
        rc = dispatch(s, verb, &L, path, depth);
        if (rc != 0) {
            char tmp[400]; snprintf(tmp, sizeof(tmp), "%.380s", g_rib_err);
            snprintf(g_rib_err, sizeof(g_rib_err), "%s:%d: %s", path, line, tmp);
            break;
        }
    }
    args_clear(&L); free(L.a); free(lx.text); free(verb); free(buf);
    return rc;
}

/* -------------------------------------------------------------- public ABI -- */

int lh_rib_load(const char *path, lh_rib_scene_t **out)
{
    lh_rib_scene_t *s;
    if (!path || !out) RIB_FAIL("lh_rib_load: NULL argument");
    s = (lh_rib_scene_t *)calloc(1, sizeof(*s));
    if (!s) RIB_FAIL("out of memory");
    m_identity(&s->stack[0]); m_identity(&s->world_to_camera);
    s->sides[0] = 1;                              /* attribute.c:51 */
    s->xres = 640; s->yres = 480; s->fov = 90.0f; /* camera.c:93-121 */
    s->xsamples = s->ysamples = 2.0f;             /* display.c defaults */
    s->gather_nsamples = 64; s->accel_method = 1; /* option.c:116,148 */
    snprintf(s->display_name, sizeof(s->display_name), "untitled.hdr");
    snprintf(s->display_type, sizeof(s->display_type), "file");
    dir_of(path, s->top_dir, sizeof(s->top_dir));
    if (parse_file(s, path, 0) != 0) { lh_rib_free(s); return -1; }
    *out = s;
    return 0;
}

void lh_rib_free(lh_rib_scene_t *s)
{
    uint32_t i;
    if (!s) return;
    for (i = 0; i < s->nmesh; i++) { free(s->mesh[i].pos); free(s->mesh[i].nrm); free(s->mesh[i].idx); }
    free(s->mesh); free(s->log); free(s);
}

const char *lh_rib_messages(const lh_rib_scene_t *s) { return (s && s->log) ? s->log : ""; }

int lh_rib_info(const lh_rib_scene_t *s, lh_rib_info_t *info)
{
    mat4 m, o; int i, j;
    if (!s || !info) RIB_FAIL("lh_rib_info: NULL argument");
    memset(info, 0, sizeof(*info));
    info->nmeshes = s->nmesh;
    for (i = 0; i < (int)s->nmesh; i++) info->ntriangles += s->mesh[i].nidx / 3;
    info->camera.width = s->xres; info->camera.height = s->yres; info->camera.rh = s->rh;
    /* ri_camera_setup (camera.c:209-240): pi is 3.141592 there */
    info->camera.flength = 1.0 / tan((s->fov * 3.141592 / 180.0) * 0.5);
    m = s->world_to_camera; m_inverse(&m);
    m_identity(&o); if (s->rh) o.f[2][2] = -o.f[2][2];
    m_mul(&m, &m, &o);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) info->camera.cam2world[4 * j + i] = m.f[j][i];
    info->perspective = s->perspective; info->camera.ortho = !s->perspective;
    info->fov = s->fov;
    info->pixel_samples[0] = (int)s->xsamples; info->pixel_samples[1] = (int)s->ysamples;
    info->gather_nsamples = s->gather_nsamples;
    info->accel_method = s->accel_method;
    info->nthreads = s->nthreads;
    info->world_complete = s->world_begun && s->world_ended;
    info->nskipped = s->nskipped; info->nrequests = s->nrequests; info->nunknown = s->nunknown;
    snprintf(info->display_name, sizeof(info->display_name), "%s", s->display_name);
    snprintf(info->display_type, sizeof(info->display_type), "%s", s->display_type);
    return 0;
}

int lh_rib_mesh(const lh_rib_scene_t *s, uint32_t mesh, uint32_t *npositions, const double **positions,
                uint32_t *nindices, const uint32_t **indices, const double **normals, int *two_side)
{
    const rib_mesh_t *m;
    if (!s || mesh >= s->nmesh) RIB_FAIL("lh_rib_mesh: mesh %u out of range", mesh);
    m = &s->mesh[mesh];
    if (npositions) *npositions = m->npos;
    if (positions) *positions = m->pos;
    if (nindices) *nindices = m->nidx;
    if (indices) *indices = m->idx;
    if (normals) *normals = m->nrm;
    if (two_side) *two_side = m->two_side;
    return 0;
}

/* ----------------------------------------------------------- .hdr writer ---- */

/* one colour -> shared-exponent bytes (rgbe.c:78-96): the scale is held in a float */
static void to_rgbe(unsigned char out[4], float r, float g, float b)
{
    float v = r; int e;
    if (g > v) v = g;
    if (b > v) v = b;
    if (v < 1e-32) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    v = (float)(frexp(v, &e) * 256.0 / v);
    out[0] = (unsigned char)(r * v); out[1] = (unsigned char)(g * v); out[2] = (unsigned char)(b * v);
    out[3] = (unsigned char)(e + 128);
}

/* Run-length code of one channel of one scanline.  The file has to come out byte for byte like the reference driver's
 * (tests/test_rib.py compares with the compiled reference), so the POLICY is the reference writer's (rgbe.c:241-291),
 * stated here as rules over spans and implemented as a span stream, not as that function's scan loop:
 *   - the channel is a sequence of spans: maximal stretches of one byte value, cut at 127;
 *   - a span of 4 or more is written as a run record (128 + length, value);
 *   - the short spans between two such runs (or the line's ends) form one literal stretch, written as literal records
 *     (count <= 128, bytes) -- unless the stretch is a single span of 2 or 3, which is written as a run record. */
typedef struct { FILE *fp; const unsigned char *d; int lit_at, lit_len, lit_spans, err; } rle_out_t;

static void rle_record(rle_out_t *o, int is_run, const unsigned char *bytes, int count)
{
    unsigned char head = (unsigned char)(is_run ? 128 + count : count);
    if (o->err) return;
    if (fwrite(&head, 1, 1, o->fp) != 1 || fwrite(bytes, is_run ? 1 : (size_t)count, 1, o->fp) != 1) o->err = 1;
}

static void rle_flush_literals(rle_out_t *o)
{
    if (o->lit_spans == 1 && o->lit_len >= 2) rle_record(o, 1, o->d + o->lit_at, o->lit_len);
    else {
        int off;
        for (off = 0; off < o->lit_len; off += 128)
            rle_record(o, 0, o->d + o->lit_at + off, o->lit_len - off < 128 ? o->lit_len - off : 128);
    }
    o->lit_len = 0; o->lit_spans = 0;
}

static int put_channel(FILE *fp, const unsigned char *d, int n)
{
    rle_out_t o; int pos = 0;
    o.fp = fp; o.d = d; o.lit_at = 0; o.lit_len = 0; o.lit_spans = 0; o.err = 0;
    while (pos < n) {
        int span = 1;
        while (span < 127 && pos + span < n && d[pos + span] == d[pos]) span++;
        if (span >= 4) {
            rle_flush_literals(&o);
            rle_record(&o, 1, d + pos, span);
        } else {
            if (o.lit_spans == 0) o.lit_at = pos;
            o.lit_len += span; o.lit_spans++;
        }
        pos += span;
    }
    rle_flush_literals(&o);
    return o.err ? -1 : 0;
}

/* rgb: height rows of width RGB float triples, top row first (what bucket_write hands the display
 * driver, render.c:962-975); negative components are clamped to 0 (hdrdrv.c:88-90) */
int lh_hdr_write(const char *path, int width, int height, const float *rgb)
{
    FILE *fp; int x, y, c, rc = 0; unsigned char *line = NULL, px[4];
    if (!path || !rgb || width <= 0 || height <= 0) RIB_FAIL("lh_hdr_write: bad argument");
    fp = fopen(path, "wb");
    if (!fp) RIB_FAIL("lh_hdr_write: cannot open \"%s\"", path);
    fprintf(fp, "#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", height, width);   /* rgbe.c:118-140 */
    if (width >= 8 && width <= 0x7fff) line = (unsigned char *)malloc((size_t)4 * width);
    for (y = 0; y < height && rc == 0; y++) {
        const float *row = rgb + (size_t)3 * width * y;
        if (!line) {                               /* too narrow / wide for RLE: flat pixels (rgbe.c:303-305) */
            for (x = 0; x < width && rc == 0; x++) {
                to_rgbe(px, row[3 * x] < 0.0f ? 0.0f : row[3 * x], row[3 * x + 1] < 0.0f ? 0.0f : row[3 * x + 1],
                        row[3 * x + 2] < 0.0f ? 0.0f : row[3 * x + 2]);
                if (fwrite(px, 4, 1, fp) != 1) rc = -1;
            }
            continue;
        }
        px[0] = 2; px[1] = 2; px[2] = (unsigned char)(width >> 8); px[3] = (unsigned char)(width & 0xFF);
        if (fwrite(px, 4, 1, fp) != 1) { rc = -1; break; }
        for (x = 0; x < width; x++) {
            to_rgbe(px, row[3 * x] < 0.0f ? 0.0f : row[3 * x], row[3 * x + 1] < 0.0f ? 0.0f : row[3 * x + 1],
                    row[3 * x + 2] < 0.0f ? 0.0f : row[3 * x + 2]);
            for (c = 0; c < 4; c++) line[c * width + x] = px[c];
        }
        for (c = 0; c < 4 && rc == 0; c++) rc = put_channel(fp, line + (size_t)c * width, width);
    }
    free(line);
    if (fclose(fp) != 0) rc = -1;
    if (rc != 0) RIB_FAIL("lh_hdr_write: write error on \"%s\"", path);
    return 0;
}
