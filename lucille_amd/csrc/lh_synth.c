/*
 * lh_synth.c -- the synthetic workloads BASELINE.json's configs 3 and 5 are quoted on
 * (SURVEY.md section 8d / Appendix C): "S-soup" random triangle soups and incoherent ray
 * dumps from one xorshift64 stream (shifts 13/7/17, u = (x >> 11) * 2^-53).  Input
 * generation only -- no ray arithmetic; bench.py, lsh_hip --soup and the tests feed the
 * accelerator from here, so the product's benchmark does not borrow the checker's library.
 *
 *   triangle i : centre c ~ U[0,1)^3 (cx, cy, cz in this order), then for each vertex
 *                c + half_extent * (2u - 1) per component (x, y, z); identity index list,
 *                so primitive i <-> (mesh 0, index 3 i)
 *   ray        : org ~ U[0,1)^3, z = 2u - 1, phi = 2 pi u, r = sqrt(1 - z^2),
 *                dir = (r cos phi, r sin phi, z); the stream continues after the triangles
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/lucille_hip.h"

static inline double xs_next(uint64_t *s)
{
    uint64_t x = *s;
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    *s = x;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/* jump-ahead: xorshift64 is linear over GF(2), x_{k+1} = T x_k with T a 64x64 bit matrix, so
 * `ndraws` steps are one product with T^ndraws (square-and-multiply, columns as uint64 words).  A rank
 * that owns rays [b, e) of a dump skips 5 b draws instead of generating b rays (ray dumps
 * are cut into contiguous slices). */
static uint64_t mat_apply(const uint64_t m[64], uint64_t x)
{
    uint64_t y = 0; int j;
    for (j = 0; j < 64; j++) if ((x >> j) & 1u) y ^= m[j];        /* m[j] = image of basis vector j */
    return y;
}

static void mat_mul(uint64_t out[64], const uint64_t a[64], const uint64_t b[64])
{
    uint64_t t[64]; int j;
    for (j = 0; j < 64; j++) t[j] = mat_apply(a, b[j]);           /* (a b) e_j = a (b e_j) */
    for (j = 0; j < 64; j++) out[j] = t[j];
}

void lh_synth_skip(uint64_t *state, uint64_t ndraws)
{
    uint64_t T[64], R[64]; int j;
    for (j = 0; j < 64; j++) {
        uint64_t x = (uint64_t)1 << j;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        T[j] = x; R[j] = (uint64_t)1 << j;
    }
    while (ndraws) {
        if (ndraws & 1u) mat_mul(R, T, R);
        mat_mul(T, T, T);
        ndraws >>= 1;
    }
    *state = mat_apply(R, *state);
}

void lh_synth_soup_triangles(uint64_t *state, uint32_t ntriangles, double half_extent, double *positions_xyz,
                             uint32_t *indices)
{
    uint32_t i; int k;
    for (i = 0; i < ntriangles; i++) {
        const double cx = xs_next(state), cy = xs_next(state), cz = xs_next(state);
        for (k = 0; k < 3; k++) {
            double *p = positions_xyz + 3 * ((size_t)3 * i + (size_t)k);
            p[0] = cx + half_extent * (2 * xs_next(state) - 1);
            p[1] = cy + half_extent * (2 * xs_next(state) - 1);
            p[2] = cz + half_extent * (2 * xs_next(state) - 1);
            indices[3 * (size_t)i + k] = 3 * i + (uint32_t)k;
        }
    }
}

void lh_synth_soup_rays(uint64_t *state, size_t n, double *org_xyz, double *dir_xyz)
{
    size_t i;
    for (i = 0; i < n; i++) {
        double z, ph, r;
        org_xyz[3 * i + 0] = xs_next(state);
        org_xyz[3 * i + 1] = xs_next(state);
        org_xyz[3 * i + 2] = xs_next(state);
        z = 2 * xs_next(state) - 1;
        ph = 6.283185307179586 * xs_next(state);
        r = sqrt(1 - z * z);
        dir_xyz[3 * i + 0] = r * cos(ph);
        dir_xyz[3 * i + 1] = r * sin(ph);
        dir_xyz[3 * i + 2] = z;
    }
}

/* midpoint subdivision, `levels` times: triangle i -> 4i .. 4i+3 = (a, ab, ca) (ab, b, bc) (ca, bc, c)
 * (ab, bc, ca); no vertex sharing (BASELINE config 5's "tessellated RIB": deterministic).
 * in: ntriangles x 9 doubles; out: ntriangles * 4^levels x 9 doubles (caller-allocated). */
void lh_synth_tessellate(const double *tri_in, size_t ntriangles, int levels, double *tri_out)
{
    /* in place from the back: level l turns the first n triangles of the buffer into 4n */
    size_t n = ntriangles, i; int l, k;
    for (i = 0; i < 9 * ntriangles; i++) tri_out[i] = tri_in[i];
    for (l = 0; l < levels; l++) {
        for (i = n; i-- > 0;) {
            double a[3], b[3], c[3], ab[3], bc[3], ca[3];
            double *o = tri_out + 36 * i;
            for (k = 0; k < 3; k++) { a[k] = tri_out[9 * i + k]; b[k] = tri_out[9 * i + 3 + k]; c[k] = tri_out[9 * i + 6 + k]; }
            for (k = 0; k < 3; k++) { ab[k] = 0.5 * (a[k] + b[k]); bc[k] = 0.5 * (b[k] + c[k]); ca[k] = 0.5 * (c[k] + a[k]); }
            for (k = 0; k < 3; k++) {
                o[k] = a[k];       o[3 + k] = ab[k];  o[6 + k] = ca[k];
                o[9 + k] = ab[k];  o[12 + k] = b[k];  o[15 + k] = bc[k];
                o[18 + k] = ca[k]; o[21 + k] = bc[k]; o[24 + k] = c[k];
                o[27 + k] = ab[k]; o[30 + k] = bc[k]; o[33 + k] = ca[k];
            }
        }
        n *= 4;
    }
}
