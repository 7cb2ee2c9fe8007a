/*
 * lucille_oracle_pt.c -- CPU restatement of the path-traced transport the product runs on the device
 * (lucille_amd/csrc/lh_pt.h + the wavefront kernels of lh_render.hip), one path at a time, plain fp64 C.
 *
 * THIS IS TEST INFRASTRUCTURE (see lucille_oracle.h): only tests/ load it.
 *
 * Parity status: the closest hit of every path vertex is lo_priv_intersect1 = ri_bvh_intersect, PINNED against the
 * compiled reference (tests/test_oracle_vs_ref.py).  The transport arithmetic around it is UNPINNED: the reference's
 * src/transport/pathtrace.c is dead code its own build leaves out (src/transport/SConscript:3-10; it does not compile
 * against the tree's current headers), so there is no reference output and no golden vector to pin it on.  What this
 * file restates is that file's documented algorithm, function by function:
 *
 *   trace_pixel / trace_path   pathtrace.c:189-314   vertex loop, MAX_PATH_VERTICES, throughput G
 *   sample_pixel               pathtrace.c:316-352   camera ray through a random sub-pixel position
 *   russian_roulette           pathtrace.c:407-430   survive with probability ave(kd) + ave(ks) + ave(kt)
 *   sample_reflection_type     pathtrace.c:432-459   'D' / 'S' / 'T' by the same three averages
 *   sample_outdir              pathtrace.c:461-498   cosine lobe / mirror / refraction, total internal reflection -> 'S'
 *   sample_cosweight           pathtrace.c:500-531   cos-weighted direction about ri_ortho_basis(normal)
 *   brdf                       pathtrace.c:533-565   kd c / pi, ks c, kt c
 *   ri_reflect / ri_refract    reflection.c:26-128
 *   ri_texture_ibl_fetch       texture.c:238-276 (+ the bilinear fetch :86-180)
 *
 * and the three places where the product (and therefore this checker) departs from it, each documented in DESIGN.md:
 *   (1) uniforms are counter-based -- lo_pt_rnd(key), key = (seed, frame pixel, sample, vertex, which draw) -- not
 *       MT19937 draws in scheduling order: a frame must not depend on tiling or sharding;
 *   (2) the path's last segment is the extension ray itself: a path that leaves the scene collects the environment
 *       in the direction it left (light_sample's extra visibility ray, :354-382, is that ray);
 *   (3) weights: flag 0 = the unbiased estimator of this sampling scheme (divide by P(type) x survival), flag 1 =
 *       the reference's own factors (1/pi on 'D', nothing else).
 * Every operation is a single IEEE operation on both sides (no contraction: oracle/Makefile builds with -ffp-contract=off;
 * the cosine lobe's sin / cos are the product's own polynomial, lobe_f32); only the light probe's acos comes from libm.
 */
#include "lucille_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int  lo_priv_intersect1(const lo_scene_t *s, const double *org, const double *dir,
                        uint32_t *prim, double *t, double *u, double *v);
void lo_priv_prim_vertices(const lo_scene_t *s, uint32_t prim, const double **v0, const double **v1,
                           const double **v2, const double **n0, const double **n1, const double **n2,
                           int *two_side, uint32_t *index, uint32_t *nindices);
void lo_priv_prim_attributes(const lo_scene_t *s, uint32_t prim, const double *a[5][3]);

/* ri_material_t's kd, ks, kt, ior (material.h:21-30) as ten floats */
typedef struct { float kd[3], ks[3], kt[3], ior; } pt_material_t;

static uint32_t mix32(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)(x >> 16);
}

double lo_pt_rnd(uint64_t key) { return (double)mix32(key) * 2.3283064365386963e-10; }       /* y * 2^-32, like randomMT2 */

/* key of (frame pixel, sample): draws 0, 1 = the sub-pixel position; vertex d uses 4 (d + 1) + {0: roulette, 1, 2: lobe, 3: type} */
static uint64_t path_key(uint64_t seed, int px, int py, int full_width, int sample)
{
    return (seed * 0x9E3779B97F4A7C15ULL) ^ ((((uint64_t)py * (uint64_t)full_width + (uint64_t)px) << 20) + (uint64_t)sample) * 64ull;
}

static void normalize3(double d[3])
{   /* ri_vector_normalize (vector.h:75-86): the threshold is a float literal */
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (n2 > (double)1.0e-17f) { const double r = 1.0 / sqrt(n2); d[0] *= r; d[1] *= r; d[2] *= r; }
}

static void cross3(double d[3], const double a[3], const double b[3])
{
    d[0] = a[1] * b[2] - a[2] * b[1]; d[1] = a[2] * b[0] - a[0] * b[2]; d[2] = a[0] * b[1] - a[1] * b[0];
}

/* the product's pt_lobe (lh_pt.h), operation for operation: single IEEE multiplies and adds only, so the bits agree */
static void lobe_f32(float z0, float z1, float *d0, float *d1, float *d2)
{
    const int k = (int)(4.0f * z1 + 0.5f);
    const float r = z1 - 0.25f * (float)k;
    const float x = 6.28318530717958647692f * r, x2 = x * x;
    float sp, cp, s, c, rad; int q;
    sp = -1.9515295891e-4f * x2; sp = sp + 8.3321608736e-3f; sp = sp * x2; sp = sp - 1.6666654611e-1f; sp = sp * x2; sp = sp * x; sp = sp + x;
    cp = 2.443315711809948e-5f * x2; cp = cp - 1.388731625493765e-3f; cp = cp * x2; cp = cp + 4.166664568298827e-2f; cp = cp * x2; cp = cp * x2;
    cp = cp - 0.5f * x2; cp = cp + 1.0f;
    q = k & 3;
    s = q == 0 ? sp : (q == 1 ? cp : (q == 2 ? -sp : -cp));
    c = q == 0 ? cp : (q == 1 ? -sp : (q == 2 ? -cp : sp));
    rad = sqrtf(z0);
    *d0 = c * rad; *d1 = s * rad; *d2 = sqrtf(1.0f - z0);
}

static double ave3(const float k[3]) { return ((double)k[0] + k[1] + k[2]) / 3.0; }

/* ri_texture_ibl_fetch (texture.c:238-276): angular map, bilinear; map = w x h RGBA floats or NULL (constant rgb) */
static void env_lookup(const float rgb[3], const float *map, int w, int h, const double dir[3], float out[3])
{
    double d[3] = {dir[0], dir[1], dir[2]}, r, n2, u, v, px, py, fx, fy, w0, w1, w2, w3;
    int x, y, x1, y1, c;
    const double pi = 3.1415926535;
    if (!map) { out[0] = rgb[0]; out[1] = rgb[1]; out[2] = rgb[2]; return; }
    normalize3(d);
    r = (d[2] >= -1.0 && d[2] < 1.0) ? (1.0 / pi) * acos(d[2]) : 0.0;
    n2 = d[0] * d[0] + d[1] * d[1];
    if (n2 > 1.0e-6) r /= sqrt(n2);
    u = 0.5 * (d[0] * r) + 0.5; v = 0.5 - 0.5 * (d[1] * r);
    u -= floor(u); v -= floor(v);
    if (u < 0.0) u = 0.0;
    if (u >= 1.0) u = 1.0;
    if (v < 0.0) v = 0.0;
    if (v >= 1.0) v = 1.0;
    px = u * (w - 1); py = v * (h - 1);
    x = (int)px; y = (int)py; fx = px - x; fy = py - y;
    x1 = x < w - 1 ? x + 1 : x; y1 = y < h - 1 ? y + 1 : y;
    w0 = (1.0 - fx) * (1.0 - fy); w1 = (1.0 - fx) * fy; w2 = fx * (1.0 - fy); w3 = fx * fy;
    for (c = 0; c < 3; c++) {
        const double t00 = map[4 * ((size_t)y * w + x) + c], t01 = map[4 * ((size_t)y1 * w + x) + c];
        const double t10 = map[4 * ((size_t)y * w + x1) + c], t11 = map[4 * ((size_t)y1 * w + x1) + c];
        out[c] = (float)(w0 * t00 + w1 * t01 + w2 * t10 + w3 * t11) * rgb[c];
    }
}

/*
 * One path: radiance[3] (float, as the device accumulates it), the number of rays it traced.
 * prim_mesh[prim] = ordinal of the mesh (ri_geom_t) the primitive belongs to; materials[mesh]; override != NULL: one
 * material for every mesh.
 */
static uint32_t one_path(const lo_scene_t *s, const lo_camera_t *cam, int px, int py, int sample, int max_vertices,
                         const uint32_t *prim_mesh, const pt_material_t *materials, const pt_material_t *override,
                         const float env_rgb[3], const float *env_map, int env_w, int env_h, int ref_weights, uint64_t seed,
                         float radiance[3])
{
    const uint64_t key0 = path_key(seed, px, py, cam->width, sample);
    double org[3], dir[3];
    float G[3] = {1.0f, 1.0f, 1.0f};
    int depth = 0, interior = 0, k;
    uint32_t nrays = 0;
    radiance[0] = radiance[1] = radiance[2] = 0.0f;
    lo_camera_ray(cam, (double)px + lo_pt_rnd(key0), (double)py + lo_pt_rnd(key0 + 1), org, dir);      /* sample_pixel */
    for (;; depth++) {
        uint32_t prim; double t, u, v;
        const uint64_t key = key0 + 4ull * (uint64_t)(depth + 1);
        const pt_material_t *M;
        double kd_, ks_, kt_, ksum, wgt, P[3], Ng[3], Ns[3], N[3], e01[3], e02[3], O[3], side = 1.0, rt, pk;
        const double *v0, *v1, *v2, *n0, *n1, *n2, *attr[5][3];
        const float *kk;
        float col[3] = {1.0f, 1.0f, 1.0f}, wsel;
        int two_side, type, back; uint32_t index, nindices;

        nrays++;
        if (!lo_priv_intersect1(s, org, dir, &prim, &t, &u, &v)) {              /* left the scene: the environment, x throughput */
            float e[3];
            env_lookup(env_rgb, env_map, env_w, env_h, dir, e);
            for (k = 0; k < 3; k++) radiance[k] = G[k] * e[k];
            break;
        }
        M = override ? override : &materials[prim_mesh[prim]];
        kd_ = ave3(M->kd); ks_ = ave3(M->ks); kt_ = ave3(M->kt);
        /* trace_path's vertex limit (the camera vertex and this one count) and russian_roulette */
        ksum = ((double)M->kd[0] + M->kd[1] + M->kd[2] + M->ks[0] + M->ks[1] + M->ks[2] + M->kt[0] + M->kt[1] + M->kt[2]) / 3.0;
        if (depth + 2 >= max_vertices || !(ksum > 0.0) || lo_pt_rnd(key) > ksum) break;      /* absorbed: contributes nothing */
        ksum = kd_ + ks_ + kt_;

        /* ri_intersection_state_build's P, Ng, Ns, colour (intersection_state.c:99-248) */
        lo_priv_prim_vertices(s, prim, &v0, &v1, &v2, &n0, &n1, &n2, &two_side, &index, &nindices);
        lo_priv_prim_attributes(s, prim, attr);
        wgt = 1.0 - u - v;
        for (k = 0; k < 3; k++) { P[k] = org[k] + dir[k] * t; e01[k] = v1[k] - v0[k]; e02[k] = v2[k] - v0[k]; }
        cross3(Ng, e01, e02); normalize3(Ng);
        if (n0) {
            for (k = 0; k < 3; k++) { const double a = n0[k] * wgt, b = n1[k] * u, c = n2[k] * v; Ns[k] = (a + b) + c; }
            normalize3(Ns);
        } else for (k = 0; k < 3; k++) Ns[k] = Ng[k];
        if (attr[0][0])
            for (k = 0; k < 3; k++) { const double a = attr[0][0][k] * wgt, b = attr[0][1][k] * u, c = attr[0][2][k] * v; col[k] = (float)((a + b) + c); }
        back = Ns[0] * dir[0] + Ns[1] * dir[1] + Ns[2] * dir[2] > 0.0;         /* the normal facing the incoming ray */
        for (k = 0; k < 3; k++) N[k] = back ? -Ns[k] : Ns[k];

        rt = lo_pt_rnd(key + 3) * ksum;                                         /* sample_reflection_type */
        type = rt < kd_ ? 'D' : (rt < kd_ + ks_ ? 'S' : 'T');
        if (type == 'T') {                                                      /* ri_refract, unit incident direction */
            double in[3] = {dir[0], dir[1], dir[2]}, e, cos1, coeff;
            normalize3(in);
            e = interior ? (double)M->ior : 1.0 / (double)M->ior;
            cos1 = -(in[0] * N[0] + in[1] * N[1] + in[2] * N[2]);
            coeff = 1.0 - (e * e) * (1.0 - cos1 * cos1);
            if (coeff <= 0.0) type = 'S';                                       /* total internal reflection */
            else {
                const double c2 = e * cos1 - sqrt(coeff);
                for (k = 0; k < 3; k++) O[k] = c2 * N[k] + e * in[k];
                normalize3(O);
                side = -1.0; interior = !interior;
            }
        }
        if (type == 'S') {                                                      /* ri_reflect: r = in - 2 n (in . n) */
            const double dn = dir[0] * N[0] + dir[1] * N[1] + dir[2] * N[2];
            for (k = 0; k < 3; k++) O[k] = dir[k] - 2.0 * dn * N[k];
        } else if (type == 'D') {                                               /* sample_cosweight about ri_ortho_basis(N) */
            double basis[3][3]; float d0, d1, d2;
            lo_ortho_basis(basis, N);
            lobe_f32((float)lo_pt_rnd(key + 1), (float)lo_pt_rnd(key + 2), &d0, &d1, &d2);     /* v.f[] = (float)(...), pathtrace.c:519-521 */
            for (k = 0; k < 3; k++) O[k] = (double)d0 * basis[0][k] + (double)d1 * basis[1][k] + (double)d2 * N[k];
        }
        kk = type == 'D' ? M->kd : (type == 'S' ? M->ks : M->kt);               /* brdf */
        pk = type == 'D' ? kd_ : (type == 'S' ? ks_ : kt_);
        wsel = ref_weights ? (type == 'D' ? 0.318309886f : 1.0f) : (float)(1.0 / pk);
        for (k = 0; k < 3; k++) {
            G[k] = G[k] * kk[k] * col[k] * wsel;
            org[k] = P[k] + side * N[k] * 1.0e-6;
            dir[k] = O[k];
        }
    }
    return nrays;
}

/*
 * spp samples (s0 .. s0 + spp - 1 of spp_total) of every pixel of the tile (x0, y0, w, h): rgb[h][w][3] (float, top row
 * first, as lh_render_pt_tile writes it) += the mean, accumulated per pixel in sample order in fp32.  path_rays (may be
 * NULL): rays of each path, [pixel][sample].  Returns the number of rays traced; *max_rays_on_a_path = the longest path.
 */
uint64_t lo_render_pt(const lo_scene_t *s, const lo_camera_t *cam, int x0, int y0, int w, int h, int s0, int spp, int spp_total,
                      int max_vertices, const uint32_t *prim_mesh, const float *materials10, const float *override10,
                      const float env_rgb[3], const float *env_map, int env_w, int env_h, int ref_weights, uint64_t seed,
                      float *rgb, uint16_t *path_rays, uint64_t *max_rays_on_a_path)
{
    uint64_t rays = 0, longest = 0;
    const float inv = 1.0f / (float)spp_total;
    int lx, ly, sm;
    for (ly = 0; ly < h; ly++)
        for (lx = 0; lx < w; lx++) {
            float sr = 0.0f, sg = 0.0f, sb = 0.0f, *o = rgb + 3 * ((size_t)(h - 1 - ly) * w + lx);
            for (sm = 0; sm < spp; sm++) {
                float rad[3];
                const uint32_t n = one_path(s, cam, x0 + lx, y0 + ly, s0 + sm, max_vertices, prim_mesh, (const pt_material_t *)materials10,
                                            (const pt_material_t *)override10, env_rgb, env_map, env_w, env_h, ref_weights, seed, rad);
                sr += rad[0]; sg += rad[1]; sb += rad[2];
                rays += n; if (n > longest) longest = n;
                if (path_rays) path_rays[((size_t)ly * w + lx) * spp + sm] = (uint16_t)(n > 65535 ? 65535 : n);
            }
            o[0] += sr * inv; o[1] += sg * inv; o[2] += sb * inv;
        }
    if (max_rays_on_a_path) *max_rays_on_a_path = longest;
    return rays;
}
