/*
 * lucille_oracle_ptref.c -- src/transport/pathtrace.c AS WRITTEN, restated from its text alone.
 * TEST INFRASTRUCTURE ONLY (see lucille_oracle.h).
 *
 * Why a second path-tracing checker: lucille_oracle_pt.c restates the PRODUCT's wavefront transport (lh_pt.h) and so can only
 * tell whether the device and the host agree with each other.  This file is written from the reference's file, function by
 * function, without looking at the product, so that every place where the product departs from pathtrace.c is a number in a
 * test (tests/test_oracle_ptref.py) and a line in HISTORY.md 11 instead of a sentence:
 *
 *   ri_transport_pathtrace   pathtrace.c:128-186   pixel loop (x outer, y from the top), nsamples paths per pixel, mean
 *   trace_pixel              pathtrace.c:189-243   camera ray; miss -> ri_texture_ibl_fetch; else trace_path, then the CONNECT
 *                                                  step: one more reflection type and direction at the path's last vertex,
 *                                                  G *= brdf, light_sample (a visibility ray; blocked -> 0, free -> the probe)
 *   trace_path               pathtrace.c:245-314   recursive: vertex limit MAX_PATH_VERTICES (10, depth starts at 2), roulette,
 *                                                  type, direction, next hit (a miss ENDS the path: the connect step samples anew),
 *                                                  G *= brdf
 *   sample_pixel             pathtrace.c:316-352   randomMT() + x, randomMT() + y through the camera
 *   light_sample             pathtrace.c:354-382
 *   russian_roulette         pathtrace.c:384-410   reject when randomMT() > ave(kd) + ave(ks) + ave(kt); NO weight for surviving
 *   sample_reflection_type   pathtrace.c:412-440   'D' / 'S' / 'T' in the proportions of the three averages; NO 1 / P(type)
 *   sample_outdir            pathtrace.c:442-485   about state.Ng, NOT turned towards the incoming ray
 *   sample_cosweight         pathtrace.c:487-519   cos(theta) = sqrt(r0), float local direction, ri_ortho_basis(normal)
 *   brdf                     pathtrace.c:521-548   kd c / pi, ks c, kt c -- the BRDF's value, no cosine, no pdf
 *   ri_reflect / ri_refract  reflection.c:26-128
 *   randomMT                 random.c:116-153 (one global MT19937, seed 4357), in the call order of the text
 *
 * ONE liberty, without which the text cannot be run at all: it shoots the next ray from state.P itself (pathtrace.c:287,233),
 * and ri_bvh_intersect accepts t = 0 (bvh.c:759-782: no epsilon), so whether a bounce "hits" the triangle it starts on is
 * decided by the last bit of P.  Rays here start at P + 1e-6 n on the side they leave on -- the offset lucille's live AO
 * transport uses (ambientocclusion.c:65-70).  Everything else is the text, including what looks like mistakes (the interior
 * flag of trace_path is set when the ray was ALREADY inside, :270-283; cosine lobes about a normal that may face away).
 *
 * Parity status: UNPINNED and unpinnable -- pathtrace.c is not compiled by the reference's own build (src/transport/
 * SConscript:3-10) and does not compile against the tree's headers; there is nothing to run.  Its closest hits are
 * lo_priv_intersect1 = ri_bvh_intersect, pinned.
 */
#include "lucille_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int  lo_priv_intersect1(const lo_scene_t *s, const double *org, const double *dir, uint32_t *prim, double *t, double *u, double *v);
void lo_priv_prim_attributes(const lo_scene_t *s, uint32_t prim, const double *a[5][3]);
void *lo_priv_mt_new(unsigned long seed);
double lo_priv_mt_next(void *m);
void lo_priv_mt_free(void *m);

#define MAX_PATH_VERTICES 10            /* pathtrace.c:66 */
#define PI_ 3.14159265358979323846      /* M_PI */

typedef struct { float kd[3], ks[3], kt[3], ior; } mat_t;              /* ri_material_t's members the transport reads */
typedef struct { uint32_t prim; double P[3], Ng[3], color[3]; const mat_t *material; } state_t;
typedef struct { int depth, interior; double G[3], indir[3]; state_t state; } pathnode_t;
typedef struct {
    const lo_scene_t *s; const uint32_t *prim_mesh; const mat_t *materials, *override;
    const float *env_rgb, *env_map; int env_w, env_h; void *mt; uint64_t rays; int max_vertices;
} ctx_t;

static double rnd(ctx_t *c) { return lo_priv_mt_next(c->mt); }
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void normalize3(double d[3])
{
    const double n2 = dot3(d, d);
    if (n2 > (double)1.0e-17f) { const double r = 1.0 / sqrt(n2); d[0] *= r; d[1] *= r; d[2] *= r; }
}
static double ave(const float k[3]) { return ((double)k[0] + (double)k[1] + (double)k[2]) / 3.0; }      /* ri_vector_ave */

/* ri_texture_ibl_fetch (texture.c:238-276), angular map, bilinear (texture.c:86-180); NULL map: the constant colour */
static void ibl_fetch(const ctx_t *c, const double dir[3], double out[3])
{
    double d[3] = {dir[0], dir[1], dir[2]}, r, n2, u, v, px, py, fx, fy; int x, y, x1, y1, k;
    const int w = c->env_w, h = c->env_h; const float *map = c->env_map;
    if (!map) { for (k = 0; k < 3; k++) out[k] = c->env_rgb[k]; return; }
    normalize3(d);
    r = (d[2] >= -1.0 && d[2] < 1.0) ? (1.0 / 3.1415926535) * acos(d[2]) : 0.0;
    n2 = d[0] * d[0] + d[1] * d[1];
    if (n2 > 1.0e-6) r /= sqrt(n2);
    u = 0.5 * (d[0] * r) + 0.5; v = 0.5 - 0.5 * (d[1] * r);
    u -= floor(u); v -= floor(v);
    if (u < 0.0) u = 0.0;
    if (u >= 1.0) u = 1.0;
    if (v < 0.0) v = 0.0;
    if (v >= 1.0) v = 1.0;
    px = u * (w - 1); py = v * (h - 1); x = (int)px; y = (int)py; fx = px - x; fy = py - y;
    x1 = x < w - 1 ? x + 1 : x; y1 = y < h - 1 ? y + 1 : y;
    for (k = 0; k < 3; k++)
        out[k] = ((1.0 - fx) * (1.0 - fy) * map[4 * ((size_t)y * w + x) + k] + (1.0 - fx) * fy * map[4 * ((size_t)y1 * w + x) + k] +
                  fx * (1.0 - fy) * map[4 * ((size_t)y * w + x1) + k] + fx * fy * map[4 * ((size_t)y1 * w + x1) + k]) * c->env_rgb[k];
}

/* ri_raytrace + the members of ri_intersection_state_t the transport reads (P, Ng, color, geom->material) */
static int raytrace(ctx_t *c, const double org[3], const double dir[3], state_t *st)
{
    uint32_t prim; double t, u, v, P[3], Ng[3], Ns[3]; int inside, k; const double *attr[5][3];
    c->rays++;
    if (!lo_priv_intersect1(c->s, org, dir, &prim, &t, &u, &v)) return 0;
    lo_state_build(c->s, prim, t, u, v, org, dir, P, Ng, Ns, &inside);
    st->prim = prim;
    for (k = 0; k < 3; k++) { st->P[k] = P[k]; st->Ng[k] = Ng[k]; st->color[k] = 1.0; }
    lo_priv_prim_attributes(c->s, prim, attr);
    if (attr[0][0]) { const double w = 1.0 - u - v; for (k = 0; k < 3; k++) st->color[k] = (attr[0][0][k] * w + attr[0][1][k] * u) + attr[0][2][k] * v; }
    st->material = c->override ? c->override : &c->materials[c->prim_mesh[prim]];
    return 1;
}

static int russian_roulette(ctx_t *c, const mat_t *m)                       /* pathtrace.c:384-410 */
{
    const double r = rnd(c), d = ave(m->kd), s = ave(m->ks), t = ave(m->kt);
    return !(r > d + s + t);
}

static int sample_reflection_type(ctx_t *c, const mat_t *m)                 /* pathtrace.c:412-440 */
{
    const double d = ave(m->kd), s = ave(m->ks), t = ave(m->kt), r = rnd(c) * (d + s + t);
    return r < d ? 'D' : (r < d + s ? 'S' : 'T');
}

static void reflect(double out[3], const double in[3], const double n[3])  /* ri_reflect: the dot product is a float (reflection.c:31,45) */
{
    const float dt = (float)dot3(in, n); int k;
    for (k = 0; k < 3; k++) out[k] = in[k] - n[k] * (double)(2 * dt);
}

static int refract(double out[3], const double in[3], const double n[3], double eta)      /* ri_refract reflection.c:69-128 */
{
    double cos1 = dot3(in, n), coeff, N[3], e = 1.0 / eta; int k;
    if (cos1 < 0.0) { cos1 = -cos1; for (k = 0; k < 3; k++) N[k] = n[k]; }
    else { e = eta; for (k = 0; k < 3; k++) N[k] = -n[k]; }
    coeff = 1.0 - (e * e) * (1.0 - cos1 * cos1);
    if (coeff <= 0.0) { reflect(out, in, n); normalize3(out); return 1; }
    coeff = e * cos1 - sqrt(coeff);
    for (k = 0; k < 3; k++) out[k] = coeff * N[k] + e * in[k];
    normalize3(out);
    return 0;
}

static void sample_cosweight(ctx_t *c, double out[3], const double normal[3])            /* pathtrace.c:487-519 */
{
    double basis[3][3], r0, r1, cost, sint, phi; float v[3]; int i;
    lo_ortho_basis(basis, normal);
    r0 = rnd(c); r1 = rnd(c);
    cost = sqrt(r0); sint = sqrt(1.0 - r0); phi = 2.0 * PI_ * r1;
    v[0] = (float)(cos(phi) * sint); v[1] = (float)(sin(phi) * sint); v[2] = (float)cost;
    for (i = 0; i < 3; i++) out[i] = v[0] * basis[0][i] + v[1] * basis[1][i] + v[2] * basis[2][i];
}

static void sample_outdir(ctx_t *c, double out[3], int *type, int interior, const mat_t *m, const double in[3], const double normal[3])
{                                                                           /* pathtrace.c:442-485 */
    if (*type == 'D') sample_cosweight(c, out, normal);
    else if (*type == 'S') reflect(out, in, normal);
    else {
        const float eta = interior ? m->ior / 1.0f : 1.0f / m->ior;
        *type = refract(out, in, normal, (double)eta) ? 'S' : 'T';
    }
}

static void brdf(double f[3], int type, const state_t *st)                   /* pathtrace.c:521-548 */
{
    const mat_t *m = st->material; int k;
    for (k = 0; k < 3; k++)
        f[k] = type == 'D' ? (double)m->kd[k] * st->color[k] / PI_ : (type == 'S' ? (double)m->ks[k] * st->color[k] : (double)m->kt[k] * st->color[k]);
}

/* the one liberty (header): the ray leaves from P + 1e-6 n on the side `dir` points to */
static void leave(double org[3], const state_t *st, const double dir[3])
{
    const double sgn = dot3(dir, st->Ng) < 0.0 ? -1.0 : 1.0; int k;
    for (k = 0; k < 3; k++) org[k] = st->P[k] + sgn * 1.0e-6 * st->Ng[k];
}

static int trace_path(ctx_t *c, pathnode_t *path)                            /* pathtrace.c:245-314 */
{
    int type, hit, prev_interior, k; double bsdf[3], outdir[3], org[3]; state_t next; const mat_t *m;
    if (path->depth >= c->max_vertices) return 0;
    m = path->state.material;
    if (!russian_roulette(c, m)) return 0;
    type = sample_reflection_type(c, m);
    prev_interior = path->interior;
    if (path->interior && !(fabs((double)m->ior - (double)1.0f) < 1.0e-10)) path->interior = 0;      /* floateq, util.h:72 */
    sample_outdir(c, outdir, &type, path->interior, m, path->indir, path->state.Ng);
    if (type == 'T' && prev_interior) path->interior = 1;                    /* sic */
    leave(org, &path->state, outdir);
    hit = raytrace(c, org, outdir, &next);
    if (!hit) return 0;
    brdf(bsdf, type, &path->state);
    for (k = 0; k < 3; k++) { path->G[k] *= bsdf[k]; path->indir[k] = outdir[k]; }
    path->depth++;
    path->state = next;
    return trace_path(c, path);
}

static void trace_pixel(ctx_t *c, const lo_camera_t *cam, double radiance[3], int x, int y)      /* pathtrace.c:189-243 */
{
    double org[3], dir[3], bsdf[3], Le[3], p0, p1; state_t st; pathnode_t node; int type, k;
    p0 = rnd(c) + x; p1 = rnd(c) + y;                                       /* sample_pixel */
    lo_camera_ray(cam, p0, p1, org, dir);
    if (!raytrace(c, org, dir, &st)) { ibl_fetch(c, dir, radiance); return; }
    node.depth = 2; node.G[0] = node.G[1] = node.G[2] = 1.0; node.state = st; node.interior = 0;
    for (k = 0; k < 3; k++) node.indir[k] = dir[k];
    trace_path(c, &node);
    /* connect the path to the light */
    type = sample_reflection_type(c, node.state.material);
    sample_outdir(c, dir, &type, node.interior, node.state.material, node.indir, node.state.Ng);
    brdf(bsdf, type, &node.state);
    for (k = 0; k < 3; k++) node.G[k] *= bsdf[k];
    leave(org, &node.state, dir);
    {   /* light_sample */
        state_t blocker;
        if (raytrace(c, org, dir, &blocker)) Le[0] = Le[1] = Le[2] = 0.0;
        else ibl_fetch(c, dir, Le);
    }
    for (k = 0; k < 3; k++) radiance[k] = Le[k] * node.G[k];
}

/* ri_transport_pathtrace for the tile (x0, y0, w, h) of the frame: rgb[h][w][3] floats, top row first; nsamples paths per pixel
 * from ONE MT19937 stream in the text's loop order.  max_vertices <= 0: MAX_PATH_VERTICES.  Returns the rays traced. */
uint64_t lo_render_ptref(const lo_scene_t *s, const lo_camera_t *cam, int x0, int y0, int w, int h, int nsamples, int max_vertices,
                         const uint32_t *prim_mesh, const float *materials10, const float *override10, const float env_rgb[3],
                         const float *env_map, int env_w, int env_h, unsigned long mt_seed, float *rgb)
{
    ctx_t c; int x, y, i;
    memset(&c, 0, sizeof(c));
    c.s = s; c.prim_mesh = prim_mesh; c.materials = (const mat_t *)materials10; c.override = (const mat_t *)override10;
    c.env_rgb = env_rgb; c.env_map = env_map; c.env_w = env_w; c.env_h = env_h; c.max_vertices = max_vertices > 0 ? max_vertices : MAX_PATH_VERTICES;
    c.mt = lo_priv_mt_new(mt_seed ? mt_seed : 4357ul);
    if (!c.mt) return 0;
    for (x = 0; x < w; x++)
        for (y = h - 1; y >= 0; y--) {
            double dcol[3] = {0.0, 0.0, 0.0}; float *o = rgb + 3 * ((size_t)(h - 1 - y) * w + x);
            for (i = 0; i < nsamples; i++) {
                double rad[3]; float f[3];
                trace_pixel(&c, cam, rad, x0 + x, y0 + y);
                f[0] = (float)rad[0]; f[1] = (float)rad[1]; f[2] = (float)rad[2];          /* radiance is a float vector in the text */
                dcol[0] += (double)f[0]; dcol[1] += (double)f[1]; dcol[2] += (double)f[2];
            }
            o[0] = (float)(dcol[0] / (double)nsamples); o[1] = (float)(dcol[1] / (double)nsamples); o[2] = (float)(dcol[2] / (double)nsamples);
        }
    lo_priv_mt_free(c.mt);
    return c.rays;
}
