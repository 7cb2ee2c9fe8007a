/*
 * lh_ao.h -- the ambient-occlusion ray producer, written once for the two places that run it on the
 * device: k_ao_rays (lh_render.hip: rays materialised in HBM -- the parity-replay path, and what
 * lh_render_scratch shows) and the refill of the any-hit traversal kernel (lh_trace2.hip: the ray is a
 * function of (hit slot, sample index) and never touches HBM).
 *
 * Reference: calculate_occlusion, src/transport/ambientocclusion.c:42-151 -- origin P + 1e-6 Ns (in the
 * hit record), for j < nphi, i < ntheta: z0 = (i + xi) / ntheta, z1 = (j + xi') / nphi, cos(theta) =
 * sqrt(z0), phi = 2 pi z1, local direction (cos phi cos theta, sin phi cos theta, sqrt(1 - cos^2 theta))
 * taken to world space through the basis rows (tangent, binormal, Ns).
 *
 *   replay   (rnd != NULL): xi, xi' supplied by the caller (the reference's MT19937 stream), everything
 *            in fp64 with the reference's operation order -- the parity path;
 *   built-in (rnd == NULL): xi, xi' from a counter-based generator keyed by the ABSOLUTE sample position
 *            (frame pixel, sub-sample, AO index), so a frame does not depend on tiling or sharding; the
 *            local direction is computed in fp32 (sincospif: phi = 2 pi z1 needs no argument reduction), then
 *            combined with the fp64 basis.  The traced ray IS that fp64 ray; its occlusion answer has the
 *            reference's semantics for it.
 */
#ifndef LH_AO_H
#define LH_AO_H

#include <stdint.h>

/* counter-based uniforms with 32-bit resolution (like randomMT2's y * 2^-32) */
__device__ __forceinline__ uint32_t lh_mix32(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)(x >> 16);
}

/* hit record of one primary hit: AO origin (P + 1e-6 Ns), tangent, binormal, Ns -- 12 doubles */
#define LH_HITREC_DOUBLES 12

/* slot key: low 34 bits = absolute sample position (frame pixel x sub-sample; keys the built-in generator), high 30 bits =
 * the primitive the AO rays of this slot start on WHEN that primitive cannot occlude them -- a flat-shaded hit: Ns = Ng, the
 * origin lies 1e-6 off the triangle's plane on the side every direction of the hemisphere points to (d . Ns = sqrt(1 - z0) >=
 * 2.4e-4), so the reference's own test of that triangle (bvh.c:730-791) ends at t < 0, ten orders of magnitude away from fp64
 * rounding -- or LH_SLOT_NOSELF when it can (interpolated normals may tilt the hemisphere below the plane).  The any-hit
 * kernel skips that one triangle instead of sending it through the fp64 test (1.0 fp64 re-tests per AO ray on BASELINE
 * config 5 without this). */
#define LH_SLOTKEY_BITS 34
#define LH_SLOTKEY_MASK ((1ull << LH_SLOTKEY_BITS) - 1ull)
#define LH_SLOT_NOSELF  0x3FFFFFFFu
__device__ __forceinline__ uint32_t lh_slot_selfprim(unsigned long long key) { return (uint32_t)(key >> LH_SLOTKEY_BITS); }

/* built-in generator: AO ray r (= j * ntheta + i) of the hit whose absolute sample key is `key` */
__device__ __forceinline__ void lh_ao_ray_builtin(const double *__restrict__ h, unsigned long long key, unsigned long long seed,
                                                  int ntheta, int nphi, int r,
                                                  double &ox, double &oy, double &oz, double &dx, double &dy, double &dz)
{
#pragma clang fp contract(off)
    const int N = ntheta * nphi;
    const int i = r % ntheta, j = r / ntheta;
    key &= LH_SLOTKEY_MASK;
    const uint64_t k = (seed * 0x9E3779B97F4A7C15ULL) ^ ((key * (uint64_t)N + (uint64_t)r) * 2ull);
    const float r0 = (float)(lh_mix32(k) >> 8) * 5.9604645e-8f;            /* 24 bits: stays below 1 in fp32 */
    const float r1 = (float)(lh_mix32(k + 1ull) >> 8) * 5.9604645e-8f;
    const float z0 = ((float)i + r0) / (float)ntheta;
    const float z1 = ((float)j + r1) / (float)nphi;
    const float ct = __builtin_sqrtf(z0), st = __builtin_sqrtf(fmaxf(1.0f - z0, 0.0f));
    float sphi, cphi;
    sincospif(2.0f * z1, &sphi, &cphi);                /* phi = 2 pi z1: no argument reduction beyond the exact 2 z1 */
    const double d0 = (double)(cphi * ct), d1 = (double)(sphi * ct), d2 = (double)st;
    ox = h[0]; oy = h[1]; oz = h[2];
    dx = d0 * h[3] + d1 * h[6] + d2 * h[9];
    dy = d0 * h[4] + d1 * h[7] + d2 * h[10];
    dz = d0 * h[5] + d1 * h[8] + d2 * h[11];
}

#endif
