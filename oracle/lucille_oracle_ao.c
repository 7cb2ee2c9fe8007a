/*
 * lucille_oracle_ao.c -- CPU restatement of the callers on either side of the
 * ray query: camera rays, the tile loop, the hit epilogue and the ambient-
 * occlusion ray producer.  TEST INFRASTRUCTURE ONLY (see lucille_oracle.h).
 *
 * Restated from (paths relative to the lucille tree):
 *
 *   ri_camera_get_pos_and_dir      src/ri/camera.c:248-318   (perspective branch)
 *   ri_vector_transform            src/base/vector.h:182-210
 *   subsample / sample_subpixel /
 *     init_sigma                   src/render/render.c:715-917
 *   create_bucket_list + spiral    src/render/render.c:582-710, src/render/spiral.c:86-131
 *   render_bucket / bucket_write   src/render/render.c:1107-1166, 919-983
 *   ri_intersection_state_build    src/render/intersection_state.c:99-248
 *   ri_normal_of_triangle, ri_lerp_vector  src/base/geometric.c:31-72
 *   ri_ortho_basis                 src/render/reflection.c:311-333
 *   calculate_occlusion,
 *     ri_transport_ambientocclusion src/transport/ambientocclusion.c:42-151,332-415
 *   randomMT2 / seedMT2            src/base/random.c:98-111,211-250  (MT19937, 1998 seeding)
 *
 * Pinned bit-for-bit (ray stream: origins, directions, order; hit records; the
 * float image) against the compiled reference rendering
 * examples/ambient_occlusion/ambient_occlusion.rib single-threaded:
 * tests/test_oracle_vs_ref.py::test_ao_* and tests/golden/ao_c1.npz.
 * sin/cos/sqrt/tan come from the same libm the reference links.
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

/* ------------------------------------------------------------ MT19937 --- */
#define MT_N 624
#define MT_M 397

typedef struct { unsigned long mt[MT_N]; int mti; } lo_mt_t;

static void mt_seed(lo_mt_t *m, unsigned long seed)
{   /* seedMT2: Knuth LCG 69069 (random.c:98-111) */
    m->mt[0] = seed & 0xffffffffUL;
    for (m->mti = 1; m->mti < MT_N; m->mti++) m->mt[m->mti] = (69069 * m->mt[m->mti - 1]) & 0xffffffffUL;
}

static double mt_next(lo_mt_t *m)
{   /* randomMT2 (random.c:211-250) */
    static const unsigned long mag01[2] = { 0x0UL, 0x9908b0dfUL };
    unsigned long y;
    if (m->mti >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (m->mt[kk] & 0x80000000UL) | (m->mt[kk + 1] & 0x7fffffffUL);
            m->mt[kk] = m->mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 0x1];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (m->mt[kk] & 0x80000000UL) | (m->mt[kk + 1] & 0x7fffffffUL);
            m->mt[kk] = m->mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 0x1];
        }
        y = (m->mt[MT_N - 1] & 0x80000000UL) | (m->mt[0] & 0x7fffffffUL);
        m->mt[MT_N - 1] = m->mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 0x1];
        m->mti = 0;
    }
    y = m->mt[m->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680UL;
    y ^= (y << 15) & 0xefc60000UL;
    y ^= (y >> 18);
    return (double)y * 2.3283064365386963e-10;
}

/* exported for tests: the first n numbers of thread 0's stream */
/* randomMT / seedMT (random.c:80-90,116-153: the same generator as randomMT2 with one global state) for the other oracle files */
void *lo_priv_mt_new(unsigned long seed) { lo_mt_t *m = (lo_mt_t *)malloc(sizeof(*m)); if (m) { mt_seed(m, seed); m->mti = MT_N; } return m; }
double lo_priv_mt_next(void *m) { return mt_next((lo_mt_t *)m); }
void lo_priv_mt_free(void *m) { free(m); }

void lo_mt_stream(unsigned long seed, size_t n, double *out)
{
    lo_mt_t m; size_t i;
    mt_seed(&m, seed);
    m.mti = MT_N;   /* seedMT2 leaves mti == N: first call regenerates */
    for (i = 0; i < n; i++) out[i] = mt_next(&m);
}

/* ------------------------------------------------------------- vectors --- */
static void vnormalize(double d[3])
{   /* ri_vector_normalize, vector.h:75-86: threshold is the FLOAT literal 1.0e-17f */
    double norm2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (norm2 > 1.0e-17f) { double rsq = 1.0 / sqrt(norm2); d[0] *= rsq; d[1] *= rsq; d[2] *= rsq; }
}

static void vcross(double d[3], const double a[3], const double b[3])
{
    d[0] = a[1] * b[2] - a[2] * b[1]; d[1] = a[2] * b[0] - a[0] * b[2]; d[2] = a[0] * b[1] - a[1] * b[0];
}

/* ri_ortho_basis reflection.c:311-333; basis rows [0]=tangent [1]=binormal [2]=n */
void lo_ortho_basis(double basis[3][3], const double n[3])
{
    int i;
    basis[2][0] = n[0]; basis[2][1] = n[1]; basis[2][2] = n[2];
    basis[1][0] = basis[1][1] = basis[1][2] = 0.0;
    for (i = 0; i < 3; i++) if (basis[2][i] < 0.6 && basis[2][i] > -0.6) break;
    if (i >= 3) i = 0;
    basis[1][i] = 1.0;
    vcross(basis[0], basis[1], basis[2]); vnormalize(basis[0]);
    vcross(basis[1], basis[2], basis[0]); vnormalize(basis[1]);
}

/* -------------------------------------------------------------- camera --- */

/* ri_camera_get_pos_and_dir (camera.c:248-318), perspective, then the
 * normalisation subsample() applies (render.c:781).  c2w row-major, row-vector
 * convention (vector.h:182-210: dst[j] = sum_i v[i]*m[i][j], v[3] forced to 1). */
void lo_camera_ray(const lo_camera_t *c, double x, double y, double org[3], double dir[3])
{
    double v[4], o[4], pos[4], dp[4]; int i, j;
    const double w = c->width, h = c->height;
    const float sign = c->rh ? -1.0 : 1.0;
    v[0] = (2.0f * x - w) / w;
    v[1] = (2.0f * y - h) / h;
    v[2] = sign * c->flength;
    v[3] = 1.0;
    o[0] = o[1] = o[2] = 0.0; o[3] = 1.0;
    if (c->ortho) {                                /* camera.c:285-301 */
        o[0] = v[0]; o[1] = v[1];
        v[2] = sign * 1.0;
    }
    for (j = 0; j < 4; j++) {
        pos[j] = 0.0; dp[j] = 0.0;
        for (i = 0; i < 4; i++) { pos[j] += o[i] * c->cam2world[4 * i + j]; dp[j] += v[i] * c->cam2world[4 * i + j]; }
    }
    for (i = 0; i < 3; i++) { org[i] = pos[i]; dir[i] = dp[i] - pos[i]; }
    vnormalize(dir);
}

/* init_sigma / sample_subpixel (render.c:830-917) */
static void radical_perm(unsigned int *sigma, unsigned int period)
{
    unsigned int i, inverse, digit, bits;
    for (i = 0; i < period; i++) {
        digit = period; inverse = 0;
        for (bits = i; bits; bits >>= 1) { digit >>= 1; if (bits & 1) inverse += digit; }
        sigma[i] = inverse;
    }
}

void lo_subpixel_jitter(int xs, int ys, int xsamples, int ysamples, double jitter[2])
{
    unsigned int *sx = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)(xsamples > 0 ? xsamples : 1));
    unsigned int *sy = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)(ysamples > 0 ? ysamples : 1));
    unsigned int j, k;
    radical_perm(sx, (unsigned int)xsamples); radical_perm(sy, (unsigned int)ysamples);
    j = (unsigned int)xs & ((unsigned int)xsamples - 1);      /* periodx - 1 on BOTH (render.c:841-842) */
    k = (unsigned int)ys & ((unsigned int)xsamples - 1);
    jitter[0] = (double)xs + (double)sx[k] / (double)xsamples;
    jitter[1] = (double)ys + (double)sy[j] / (double)ysamples;
    jitter[0] /= (double)xsamples; jitter[1] /= (double)ysamples;
    jitter[0] += 0.5 / (xsamples * xsamples);
    jitter[1] += 0.5 / (ysamples * ysamples);
    free(sx); free(sy);
}

/* NthBucketSpiral (spiral.c:86-131) */
static void nth_bucket_spiral(int n, int nxb, int nyb, unsigned int *bx, unsigned int *by)
{
    int nx = nxb, ny = nyb, nxny, minnxny, x, y;
    int minnb = (nxb < nyb) ? nxb : nyb;
    int center = (minnb - 1) / 2;
    while (n < nx * ny) { nx = nx - 1; ny = ny - 1; }
    nxny = nx * ny; minnxny = (nx < ny) ? nx : ny;
    if (minnxny % 2 == 1) {
        if (n <= (nxny + ny)) { x = nx - minnxny / 2; y = -minnxny / 2 + n - nxny; }
        else { x = nx - minnxny / 2 - (n - (nxny + ny)); y = ny - minnxny / 2; }
    } else {
        if (n <= (nxny + ny)) { x = -minnxny / 2; y = ny - minnxny / 2 - (n - nxny); }
        else { x = -minnxny / 2 + (n - (nxny + ny)); y = -minnxny / 2; }
    }
    *bx = (unsigned int)(x + center); *by = (unsigned int)(y + center);
}

/* bucket visiting order of one frame: out_xy[2*k] = (bx, by) bucket coordinates */
int lo_bucket_order(int width, int height, int bucket_size, unsigned int *out_xy)
{
    int nx = (int)ceil(width / (double)bucket_size), ny = (int)ceil(height / (double)bucket_size), n;
    for (n = 0; n < nx * ny; n++) nth_bucket_spiral(n, nx, ny, &out_xy[2 * n], &out_xy[2 * n + 1]);
    return nx * ny;
}

/* -------------------------------------------------------- hit epilogue --- */

/* ri_intersection_state_build (intersection_state.c:99-248): P, Ng, Ns, inside */
void lo_state_build(const lo_scene_t *s, uint32_t prim, double t, double u, double v,
                    const double org[3], const double dir[3],
                    double P[3], double Ng[3], double Ns[3], int *inside)
{
    const double *v0, *v1, *v2, *n0, *n1, *n2; int two_side, k; uint32_t index, nindices;
    double v01[3], v02[3];
    lo_priv_prim_vertices(s, prim, &v0, &v1, &v2, &n0, &n1, &n2, &two_side, &index, &nindices);
    for (k = 0; k < 3; k++) P[k] = org[k] + dir[k] * t;
    for (k = 0; k < 3; k++) { v01[k] = v1[k] - v0[k]; v02[k] = v2[k] - v0[k]; }
    vcross(Ng, v01, v02); vnormalize(Ng);
    if (n0) {
        const double w = 1.0 - u - v;
        for (k = 0; k < 3; k++) { double a = n0[k] * w, b = n1[k] * u, c = n2[k] * v; Ns[k] = (a + b) + c; }
    } else {
        for (k = 0; k < 3; k++) Ns[k] = Ng[k];
    }
    if (inside) *inside = (two_side && index >= nindices / 2) ? 1 : 0;
}

void lo_priv_prim_attributes(const lo_scene_t *s, uint32_t prim, const double *a[5][3]);

/* the whole ri_intersection_state_build (intersection_state.c:99-248): state24 = P[3] Ng[3] Ns[3] tangent[3]
 * binormal[3] color[3] st[2] I[3] inside -- the layout of the product's LH_STATE_DOUBLES record */
void lo_state_build_full(const lo_scene_t *s, uint32_t prim, double t, double u, double v,
                         const double org[3], const double dir[3], double state24[24])
{
    const double *a[5][3]; double P[3], Ng[3], Ns[3], I[3], basis[3][3]; int inside, k;
    const double w = 1.0 - u - v;
    lo_state_build(s, prim, t, u, v, org, dir, P, Ng, Ns, &inside);
    lo_priv_prim_attributes(s, prim, a);
    for (k = 0; k < 3; k++) { state24[k] = P[k]; state24[3 + k] = Ng[k]; state24[6 + k] = Ns[k]; I[k] = dir[k]; }
    vnormalize(I);
    {
        const double *n0, *n1, *n2, *v0, *v1, *v2; int two; uint32_t index, nind;
        lo_priv_prim_vertices(s, prim, &v0, &v1, &v2, &n0, &n1, &n2, &two, &index, &nind);
        if (n0 && a[1][0] && a[2][0]) {          /* tangents AND binormals, only looked at when normals exist (:161-176) */
            for (k = 0; k < 3; k++) {
                double x = a[1][0][k] * w, y = a[1][1][k] * u, z = a[1][2][k] * v; state24[9 + k] = (x + y) + z;
                x = a[2][0][k] * w; y = a[2][1][k] * u; z = a[2][2][k] * v; state24[12 + k] = (x + y) + z;
            }
        } else {
            lo_ortho_basis(basis, Ng);
            for (k = 0; k < 3; k++) { state24[9 + k] = basis[0][k]; state24[12 + k] = basis[1][k]; }
        }
    }
    if (a[0][0]) for (k = 0; k < 3; k++) { double x = a[0][0][k] * w, y = a[0][1][k] * u, z = a[0][2][k] * v; state24[15 + k] = (x + y) + z; }
    else { state24[15] = 1.0; state24[16] = 1.0; state24[17] = 1.0; }
    if (a[3][0] || a[4][0]) {                    /* shared texcoords win over unshared (:210-224); lerp_uv :266-280 */
        const int kk = a[3][0] ? 3 : 4;
        state24[18] = (1 - u - v) * a[kk][0][0] + u * a[kk][1][0] + v * a[kk][2][0];
        state24[19] = (1 - u - v) * a[kk][0][1] + u * a[kk][1][1] + v * a[kk][2][1];
    } else { state24[18] = 0.0; state24[19] = 0.0; }
    for (k = 0; k < 3; k++) state24[20 + k] = I[k];
    state24[23] = (double)inside;
}

/* closest hit + full state of n rays; misses leave zeros */
void lo_state_batch(const lo_scene_t *s, size_t n, const double *org, const double *dir, uint32_t *prim, double *state24)
{
    size_t i;
    for (i = 0; i < n; i++) {
        uint32_t p; double t, u, v;
        memset(state24 + 24 * i, 0, 24 * sizeof(double));
        lo_intersect_batch(s, 1, org + 3 * i, dir + 3 * i, &p, &t, &u, &v, NULL, 1);
        prim[i] = p;
        if (p != 0xFFFFFFFFu) lo_state_build_full(s, p, t, u, v, org + 3 * i, dir + 3 * i, state24 + 24 * i);
    }
}

/* AO rays of one shading point (calculate_occlusion, ambientocclusion.c:42-151).
 * rnd = 2*ntheta*nphi numbers in consumption order (z0 then z1, i fastest). */
void lo_ao_rays(const double P[3], const double Ns[3], uint32_t ntheta, uint32_t nphi,
                const double *rnd, double *org_xyz, double *dir_xyz)
{
    double basis[3][3], org[3]; uint32_t i, j, k; size_t r = 0;
    const double eps = 1.0e-6;
    lo_ortho_basis(basis, Ns);
    for (k = 0; k < 3; k++) org[k] = P[k] + Ns[k] * eps;
    for (j = 0; j < nphi; j++)
        for (i = 0; i < ntheta; i++, r++) {
            double z0 = (i + rnd[2 * r]) / (double)ntheta;
            double z1 = (j + rnd[2 * r + 1]) / (double)nphi;
            double cos_theta = sqrt(z0), phi = 2.0 * M_PI * z1, d[3];
            d[0] = cos(phi) * cos_theta; d[1] = sin(phi) * cos_theta; d[2] = sqrt(1.0 - cos_theta * cos_theta);
            for (k = 0; k < 3; k++) {
                org_xyz[3 * r + k] = org[k];
                dir_xyz[3 * r + k] = d[0] * basis[0][k] + d[1] * basis[1][k] + d[2] * basis[2][k];
            }
        }
}

/* ----------------------------------------------------------- the frame --- */

/*
 * ri_render_frame with the hard-wired AO transport (render.c:800-804), one
 * thread (thread 0's MT stream, seed 4357).  image: [height][width][3] float,
 * y flipped exactly as bucket_write does (render.c:962-964).  If rec_* are
 * non-NULL they receive every ray in the order the reference issues them
 * (capacity rec_cap rays; returns the number of rays traced).
 */
size_t lo_render_ao(const lo_scene_t *s, const lo_camera_t *cam, int xsamples, int ysamples,
                    int gather_nsamples, int bucket_size, float *image,
                    double *rec_org, double *rec_dir, uint32_t *rec_prim, double *rec_t,
                    double *rec_u, double *rec_v, size_t rec_cap)
{
    const int W = cam->width, H = cam->height;
    const int nxb = (int)ceil(W / (double)bucket_size), nyb = (int)ceil(H / (double)bucket_size);
    unsigned int *order = (unsigned int *)malloc(sizeof(unsigned int) * 2 * (size_t)(nxb * nyb));
    const int nphi = (int)sqrt((double)gather_nsamples), ntheta = nphi;
    double *rnd = (double *)malloc(sizeof(double) * 2 * (size_t)(nphi * ntheta + 1));
    double *ao_o = (double *)malloc(sizeof(double) * 3 * (size_t)(nphi * ntheta + 1));
    double *ao_d = (double *)malloc(sizeof(double) * 3 * (size_t)(nphi * ntheta + 1));
    lo_mt_t mt; size_t nrays = 0; int b, nb;
    mt_seed(&mt, 4357); mt.mti = MT_N;
    memset(image, 0, sizeof(float) * 3 * (size_t)W * (size_t)H);
    nb = lo_bucket_order(W, H, bucket_size, order);

#define LO_REC(o_, d_, p_, t_, u_, v_) do { if (rec_org && nrays < rec_cap) { int q_; \
        for (q_ = 0; q_ < 3; q_++) { rec_org[3 * nrays + q_] = (o_)[q_]; rec_dir[3 * nrays + q_] = (d_)[q_]; } \
        rec_prim[nrays] = (p_); rec_t[nrays] = (t_); rec_u[nrays] = (u_); rec_v[nrays] = (v_); } nrays++; } while (0)

    for (b = 0; b < nb; b++) {
        const int bx = (int)order[2 * b] * bucket_size, by = (int)order[2 * b + 1] * bucket_size;
        const int bw = (bx + bucket_size > W) ? W - bx : bucket_size;
        const int bh = (by + bucket_size > H) ? H - by : bucket_size;
        int px, py;
        for (py = by; py < by + bh; py++)
            for (px = bx; px < bx + bw; px++) {
                double accum[3] = { 0.0, 0.0, 0.0 }; int xs, ys, k;
                for (ys = 0; ys < ysamples; ys++)
                    for (xs = 0; xs < xsamples; xs++) {
                        double jit[2], org[3], dir[3], t, u, v, rad[3] = { 0.0, 0.0, 0.0 }; uint32_t prim; int hit;
                        lo_subpixel_jitter(xs, ys, xsamples, ysamples, jit);
                        lo_camera_ray(cam, (double)(px + jit[0]), (double)(py + jit[1]), org, dir);
                        hit = lo_priv_intersect1(s, org, dir, &prim, &t, &u, &v);
                        LO_REC(org, dir, prim, t, u, v);
                        if (hit) {
                            double P[3], Ng[3], Ns[3], occlusion = 0.0, ns; int r, n = nphi * ntheta;
                            lo_state_build(s, prim, t, u, v, org, dir, P, Ng, Ns, NULL);
                            for (r = 0; r < 2 * n; r++) rnd[r] = mt_next(&mt);
                            lo_ao_rays(P, Ns, (uint32_t)ntheta, (uint32_t)nphi, rnd, ao_o, ao_d);
                            for (r = 0; r < n; r++) {
                                uint32_t p2; double t2, u2, v2;
                                int h2 = lo_priv_intersect1(s, &ao_o[3 * r], &ao_d[3 * r], &p2, &t2, &u2, &v2);
                                LO_REC(&ao_o[3 * r], &ao_d[3 * r], p2, t2, u2, v2);
                                if (h2) occlusion += 1.0;
                            }
                            ns = (double)(uint32_t)(ntheta * nphi);
                            rad[0] = rad[1] = rad[2] = 1.0 * (ns - occlusion) / ns;
                        }
                        for (k = 0; k < 3; k++) accum[k] = accum[k] + rad[k];
                    }
                for (k = 0; k < 3; k++) {
                    double val = accum[k] * ((double)1.0 / (xsamples * ysamples));
                    float f = (float)val;
                    if (f < 0.0) f = 0.0;                             /* hdrdrv.c:88-90 */
                    image[3 * ((size_t)px + (size_t)(H - py - 1) * (size_t)W) + k] += f;
                }
            }
    }
#undef LO_REC
    free(order); free(rnd); free(ao_o); free(ao_d);
    return nrays;
}
