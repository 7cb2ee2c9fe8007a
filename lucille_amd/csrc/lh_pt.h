/*
 * lh_pt.h -- the path tracer's per-vertex arithmetic, for the two places that run it on the device: the shading pass of
 * lh_render.hip (k_pt_decide + k_pt_scatter per bounce: decides and compacts the live paths, scatters the survivors) and the closest-hit walk
 * of lh_kernels.hip, whose ray source 2 generates the camera rays of a pass in its refill (pt_camera_ray) instead of reading
 * them from HBM.  Counter-based keys (pixel, sample, bounce): a frame does not depend on tiling, sharding or slot order.
 *
 * Reference (dead code in the tree, so this is its documented algorithm): src/transport/pathtrace.c:189-314 (trace loop),
 * 316-352 (sample_pixel), 407-430 (russian_roulette), 432-459 (sample_reflection_type), 500-531 (sample_cosweight),
 * 533-565 (brdf); ri_reflect / ri_refract reflection.c:26-128; ri_texture_ibl_fetch texture.c:238-276.
 * Device code only (included inside the including file's anonymous namespace).
 */
#ifndef LH_PT_H
#define LH_PT_H

#define LH_NC _Pragma("clang fp contract(off)")
#define LH_PT_INTERIOR 0x80000000u

/* path id -> (pixel, sample) -> (line, band): three divisions by the pass's spp, width and band height per key, ~25 instructions each
 * as run-time 32-bit divisions.  n / d for n < 2^30 as a multiplication: with l = ceil(log2 d), k = max(32, 30 + l) and
 * m = floor(2^k / d) + 1 (below 2^32 for d >= 2), m d = 2^k + e with 0 < e <= d <= 2^l, so n m / 2^k = n / d + n e / (d 2^k) and
 * n e < 2^(30 + l) <= 2^k keeps the excess below 1 / d: the quotient is floor(n / d) exactly.  (Packed on the host, lh_div_make.) */
struct LhDiv { uint32_t m, sh, one, pad; };
struct PtDivs { LhDiv spp, w, rows; };
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t lh_div(uint32_t n, const LhDiv d) { return d.one ? n : (__umulhi(n, d.m) >> d.sh); }
#endif
static inline LhDiv lh_div_make(uint32_t d)
{
    LhDiv r; r.pad = 0; r.one = d <= 1u; r.m = 0; r.sh = 0;
    if (d > 1u) {
        uint32_t l = 0; while ((1ull << l) < (unsigned long long)d) l++;
        const uint32_t k = 30u + l > 32u ? 30u + l : 32u;
        r.m = (uint32_t)(((unsigned long long)1 << k) / d + 1ull); r.sh = k - 32u;
    }
    return r;
}

struct DevCamera {
    double c2w[16];
    double flength;
    int width, height, rh, ortho;
};
/* lh_material_t + what every vertex would otherwise recompute from it with fp64 divisions: the channel averages the
 * roulette and the lobe choice run on, and 1 / P(lobe).  Packed on the host (lh_pt_material_pack: the same IEEE operations,
 * so the same bits as computing them here). */
struct DevMaterial { float kd[3], ks[3], kt[3], ior; double ad, as, at, asum9; float wd, ws, wt, pad; };
struct DevEnv { float rgb[3]; const float4 *map; int w, h; };

__device__ __forceinline__ void vnormalize(double d[3])
{   /* ri_vector_normalize (vector.h:75-86): FLOAT threshold literal */
    LH_NC
    const double norm2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (norm2 > (double)1.0e-17f) { const double rsq = 1.0 / sqrt(norm2); d[0] *= rsq; d[1] *= rsq; d[2] *= rsq; }
}

__device__ __forceinline__ void vcross(double d[3], const double a[3], const double b[3])
{
    LH_NC
    d[0] = a[1] * b[2] - a[2] * b[1]; d[1] = a[2] * b[0] - a[0] * b[2]; d[2] = a[0] * b[1] - a[1] * b[0];
}

/* cosine-weighted direction about +z from two uniforms, single precision: radius sqrt(z0) in the plane at angle 2 pi z1,
 * sqrt(1 - z0) along the axis.  sin / cos of 2 pi z1 by quadrant reduction (exact) and the Cephes sinf / cosf kernels on
 * |x| <= pi / 4, every operation a single IEEE multiply or add (no contraction): bit-identical on the host and the device. */
__device__ __forceinline__ void pt_lobe(float z0, float z1, float &d0, float &d1, float &d2)
{
    LH_NC
    const int k = (int)(4.0f * z1 + 0.5f);                 /* nearest quarter turn, 0 .. 4 */
    const float r = z1 - 0.25f * (float)k;                  /* exact; |r| <= 1 / 8 */
    const float x = 6.28318530717958647692f * r, x2 = x * x;
    float sp = -1.9515295891e-4f * x2; sp = sp + 8.3321608736e-3f; sp = sp * x2; sp = sp - 1.6666654611e-1f; sp = sp * x2; sp = sp * x; sp = sp + x;
    float cp = 2.443315711809948e-5f * x2; cp = cp - 1.388731625493765e-3f; cp = cp * x2; cp = cp + 4.166664568298827e-2f; cp = cp * x2; cp = cp * x2;
    cp = cp - 0.5f * x2; cp = cp + 1.0f;
    const int q = k & 3;
    const float s = q == 0 ? sp : (q == 1 ? cp : (q == 2 ? -sp : -cp));
    const float c = q == 0 ? cp : (q == 1 ? -sp : (q == 2 ? -cp : sp));
    const float rad = sqrtf(z0);
    d0 = c * rad; d1 = s * rad; d2 = sqrtf(1.0f - z0);
}

__device__ __forceinline__ double rnd01(uint64_t key) { return (double)lh_mix32(key) * 2.3283064365386963e-10; }

/* ri_texture_ibl_fetch (texture.c:238-276) + ri_texture_fetch's bilinear filter (:86-180) */
__device__ __forceinline__ void env_fetch(const DevEnv &e, double dx, double dy, double dz, float out[3])
{
    if (!e.map) { out[0] = e.rgb[0]; out[1] = e.rgb[1]; out[2] = e.rgb[2]; return; }
    double d[3] = {dx, dy, dz};
    vnormalize(d);
    const double pi = 3.1415926535;
    double r = (d[2] >= -1.0 && d[2] < 1.0) ? (1.0 / pi) * acos(d[2]) : 0.0;
    const double n2 = d[0] * d[0] + d[1] * d[1];
    if (n2 > 1.0e-6) r /= sqrt(n2);
    double u = 0.5 * (d[0] * r) + 0.5, v = 0.5 - 0.5 * (d[1] * r);
    u -= floor(u); v -= floor(v);
    if (u < 0.0) u = 0.0; if (u >= 1.0) u = 1.0;
    if (v < 0.0) v = 0.0; if (v >= 1.0) v = 1.0;
    const double px = u * (e.w - 1), py = v * (e.h - 1);
    int x = (int)px, y = (int)py;
    const double fx = px - x, fy = py - y;
    const int x1 = x < e.w - 1 ? x + 1 : x, y1 = y < e.h - 1 ? y + 1 : y;
    const float4 t00 = e.map[(size_t)y * e.w + x], t01 = e.map[(size_t)y1 * e.w + x];
    const float4 t10 = e.map[(size_t)y * e.w + x1], t11 = e.map[(size_t)y1 * e.w + x1];
    const double w0 = (1.0 - fx) * (1.0 - fy), w1 = (1.0 - fx) * fy, w2 = fx * (1.0 - fy), w3 = fx * fy;
    out[0] = (float)(w0 * t00.x + w1 * t01.x + w2 * t10.x + w3 * t11.x) * e.rgb[0];
    out[1] = (float)(w0 * t00.y + w1 * t01.y + w2 * t10.y + w3 * t11.y) * e.rgb[1];
    out[2] = (float)(w0 * t00.z + w1 * t01.z + w2 * t10.z + w3 * t11.z) * e.rgb[2];
}

/* the camera ray of path `id` = (pixel * spp + s) of a w-wide tile at (x0, y0): through a random sub-pixel position
 * (sample_pixel, pathtrace.c:316-352; ri_camera_get_pos_and_dir, camera.c:248-318) */
/* path id -> frame pixel.  The pass covers a w-wide region whose lines are those of full bands: line `row` of the pass is
 * line y0 + (row / band_rows) * band_stride + row % band_rows of the frame (an ordinary tile: band_rows = its height; a rank's
 * interleaved bands of a sharded frame: band_stride = world x band_rows) */
__device__ __forceinline__ void pt_pixel(uint32_t pix, int x0, int y0, int w, int band_rows, int band_stride, const PtDivs dv, int &px, int &py)
{
    const uint32_t row = lh_div(pix, dv.w), band = lh_div(row, dv.rows);
    px = x0 + (int)(pix - row * (uint32_t)w);
    py = y0 + (int)band * band_stride + (int)(row - band * (uint32_t)band_rows);
}

__device__ __forceinline__ void pt_primary_ray(const DevCamera &cam, int x0, int y0, int w, int band_rows, int band_stride, int spp, int s0,
                                               const PtDivs dv, unsigned long long seed, size_t id, double pos3[3], double d[3])
{
    LH_NC
    /* a pass holds at most 2^30 paths: 32-bit ids, divisions by multiplication (LhDiv) */
    const uint32_t id32 = (uint32_t)id, pix = lh_div(id32, dv.spp);
    const int s = (int)(id32 - pix * (uint32_t)spp);
    int px, py;
    pt_pixel(pix, x0, y0, w, band_rows, band_stride, dv, px, py);
    const uint64_t key = (seed * 0x9E3779B97F4A7C15ULL) ^ ((((uint64_t)py * (uint64_t)cam.width + (uint64_t)px) << 20) + (uint64_t)(s0 + s)) * 64ull;
    const double x = (double)px + rnd01(key), y = (double)py + rnd01(key + 1);
    const double W = cam.width, H = cam.height;
    const float sign = cam.rh ? -1.0f : 1.0f;
    double v[4], o[4] = {0.0, 0.0, 0.0, 1.0}, pos[4], dp[4];
    v[0] = (2.0f * x - W) / W; v[1] = (2.0f * y - H) / H; v[2] = sign * cam.flength; v[3] = 1.0;
    if (cam.ortho) { o[0] = v[0]; o[1] = v[1]; v[2] = sign * 1.0; }      /* camera.c:285-301 */
    for (int c = 0; c < 4; c++) {
        pos[c] = 0.0; dp[c] = 0.0;
        for (int r = 0; r < 4; r++) { pos[c] += o[r] * cam.c2w[4 * r + c]; dp[c] += v[r] * cam.c2w[4 * r + c]; }
    }
    d[0] = dp[0] - pos[0]; d[1] = dp[1] - pos[1]; d[2] = dp[2] - pos[2];
    vnormalize(d);
    pos3[0] = pos[0]; pos3[1] = pos[1]; pos3[2] = pos[2];
}

/* the camera rays of one pass as a ray SOURCE (lh_kernels.hip, ray source 2; the first bounce's shading pass): path id ->
 * ray, nothing materialised.  Lives in device memory (written by k_pt_begin), read with scalar loads. */
struct PtCamSrc { DevCamera cam; unsigned long long seed; int x0, y0, w, spp, s0, band_rows, band_stride, pad; PtDivs dv; };

__device__ __forceinline__ void pt_camera_ray(const PtCamSrc *__restrict__ c, uint32_t id, double pos3[3], double d[3])
{
    pt_primary_ray(c->cam, c->x0, c->y0, c->w, c->band_rows, c->band_stride, c->spp, c->s0, c->dv, c->seed, (size_t)id, pos3, d);
}

__device__ __forceinline__ uint64_t pt_key(unsigned long long seed, uint32_t path, int spp, int s0, int x0, int y0, int w, int band_rows, int band_stride,
                                           const PtDivs dv, int full_width, int depth)
{
    const uint32_t pix = lh_div(path, dv.spp);
    int px, py;
    pt_pixel(pix, x0, y0, w, band_rows, band_stride, dv, px, py);
    const uint64_t gx = (uint64_t)px, gy = (uint64_t)py;
    return ((seed * 0x9E3779B97F4A7C15ULL) ^ (((gy * (uint64_t)full_width + gx) << 20) + (uint64_t)(s0 + (int)(path - pix * (uint32_t)spp))) * 64ull)
           + 4ull * (uint64_t)(depth + 1);
}

/* a path vertex that hit something: does the path go on?  (vertex limit; Russian roulette on d + s + t, pathtrace.c:407-430) */
__device__ __forceinline__ bool pt_survives(const double ksum /* DevMaterial.asum9: (kd0 + kd1 + kd2 + ks0 + ... + kt2) / 3 */, uint64_t key, int depth, int max_depth)
{
    return !(depth + 2 >= max_depth || !(ksum > 0.0) || rnd01(key) > ksum);
}

/* a surviving vertex: hit epilogue (P, Ng, Ns, colour), reflection type D / S / T, the next ray and the new throughput */
__device__ __forceinline__ void pt_scatter(const lh_dev_scene_t &sc, const double *__restrict__ nrm9, const double *__restrict__ col9,
                                           const DevMaterial &M, int ref_weights, uint64_t key, uint32_t p, uint32_t pword,
                                           const double org[3], const double D[3], double tt, double uu, double vv, const float G[3],
                                           double org2[3], double O[3], float G2[3], uint32_t &pword2)
{
    LH_NC
    const double kd_ = M.ad, ks_ = M.as, kt_ = M.at;            /* ri_vector_ave of kd, ks, kt */
    const double ksum = kd_ + ks_ + kt_;
    /* ri_intersection_state_build subset: P, Ng, Ns, colour */
    const double *tv = (const double *)sc.tri64 + 9 * (size_t)p;
    const double wgt = 1.0 - uu - vv;
    double P[3], Ng[3], Ns[3];
    for (int k = 0; k < 3; k++) P[k] = org[k] + D[k] * tt;
    bool has_n = false;
    if (nrm9) { const double n0x = nrm9[9 * (size_t)p]; has_n = (n0x == n0x); }
    if (has_n) {
        const double *nn = nrm9 + 9 * (size_t)p;
        for (int k = 0; k < 3; k++) { const double a = nn[k] * wgt, b = nn[3 + k] * uu, c = nn[6 + k] * vv; Ns[k] = (a + b) + c; }
        vnormalize(Ns);
    } else {                                    /* flat shaded: the geometric normal (only then is it needed at all -- and the triangle's corners) */
        double v01[3], v02[3];
        for (int k = 0; k < 3; k++) { v01[k] = tv[3 + k] - tv[k]; v02[k] = tv[6 + k] - tv[k]; }
        vcross(Ng, v01, v02); vnormalize(Ng);
        Ns[0] = Ng[0]; Ns[1] = Ng[1]; Ns[2] = Ng[2];
    }
    float col[3] = {1.0f, 1.0f, 1.0f};
    if (col9) {
        const double *cc = col9 + 9 * (size_t)p;
        if (cc[0] == cc[0]) for (int k = 0; k < 3; k++) { const double a = cc[k] * wgt, b = cc[3 + k] * uu, c = cc[6 + k] * vv; col[k] = (float)((a + b) + c); }
    }
    /* the normal facing the incoming ray */
    const bool back = Ns[0] * D[0] + Ns[1] * D[1] + Ns[2] * D[2] > 0.0;
    double N[3] = {back ? -Ns[0] : Ns[0], back ? -Ns[1] : Ns[1], back ? -Ns[2] : Ns[2]};
    /* reflection type (sample_reflection_type, pathtrace.c:432-459) */
    const double rt = rnd01(key + 3) * ksum;
    int type = rt < kd_ ? 0 : (rt < kd_ + ks_ ? 1 : 2);
    uint32_t interior = pword & LH_PT_INTERIOR;
    double side = 1.0;                          /* which side of the surface the next ray starts on */
    if (type == 2) {
        /* ri_refract (reflection.c:69-128) with the unit direction: relative index ior when leaving, 1 / ior when entering */
        double In[3] = {D[0], D[1], D[2]};
        vnormalize(In);
        const double e = interior ? (double)M.ior : 1.0 / (double)M.ior;
        const double cos1 = -(In[0] * N[0] + In[1] * N[1] + In[2] * N[2]);
        const double coeff = 1.0 - (e * e) * (1.0 - cos1 * cos1);
        if (coeff <= 0.0) type = 1;             /* total internal reflection */
        else {
            const double c2 = e * cos1 - sqrt(coeff);
            for (int k = 0; k < 3; k++) O[k] = c2 * N[k] + e * In[k];
            vnormalize(O);
            side = -1.0;
            interior ^= LH_PT_INTERIOR;
        }
    }
    if (type == 1) {                            /* ri_reflect (reflection.c:26-50): r = in - 2 n (in . n) */
        const double dn = D[0] * N[0] + D[1] * N[1] + D[2] * N[2];
        for (int k = 0; k < 3; k++) O[k] = D[k] - 2.0 * dn * N[k];
    } else if (type == 0) {                     /* sample_cosweight (pathtrace.c:500-531) about the facing normal */
        /* the first axis the normal is not close to (selects, not an indexed store: an array indexed at run time lives in scratch) */
        const int ax = (N[0] < 0.6 && N[0] > -0.6) ? 0 : ((N[1] < 0.6 && N[1] > -0.6) ? 1 : ((N[2] < 0.6 && N[2] > -0.6) ? 2 : 0));
        double b0[3], b1[3] = {ax == 0 ? 1.0 : 0.0, ax == 1 ? 1.0 : 0.0, ax == 2 ? 1.0 : 0.0};
        vcross(b0, b1, N); vnormalize(b0);
        vcross(b1, N, b0); vnormalize(b1);
        /* the local direction in single precision, as the reference holds it (v.f[] = (float)(...), pathtrace.c:519-521),
         * by arithmetic that gives the same bits on any IEEE machine (pt_lobe): the oracle repeats it exactly */
        float d0, d1, d2;
        pt_lobe((float)rnd01(key + 1), (float)rnd01(key + 2), d0, d1, d2);
        for (int k = 0; k < 3; k++) O[k] = (double)d0 * b0[k] + (double)d1 * b1[k] + (double)d2 * N[k];
    }
    /* throughput (brdf, pathtrace.c:533-565) */
    /* the chosen lobe's reflectance and weight by bit masks: written as selects of the record's fields the compiler turns them into
     * loads at a run-time offset, which pins the record in scratch */
    const uint32_t m0 = type == 0 ? 0xFFFFFFFFu : 0u, m1 = type == 1 ? 0xFFFFFFFFu : 0u, m2 = type == 2 ? 0xFFFFFFFFu : 0u;
#define LH_PICK(a, b, c) __uint_as_float((__float_as_uint(a) & m0) | (__float_as_uint(b) & m1) | (__float_as_uint(c) & m2))
    const float kk[3] = {LH_PICK(M.kd[0], M.ks[0], M.kt[0]), LH_PICK(M.kd[1], M.ks[1], M.kt[1]), LH_PICK(M.kd[2], M.ks[2], M.kt[2])};
    const float wsel = ref_weights ? (type == 0 ? 0.318309886f : 1.0f) : LH_PICK(M.wd, M.ws, M.wt);     /* unbiased: 1 / (P(type) x survival) = (float)(1 / ave) */
#undef LH_PICK
    for (int k = 0; k < 3; k++) {
        G2[k] = G[k] * kk[k] * col[k] * wsel;
        org2[k] = P[k] + side * N[k] * 1.0e-6;
    }
    pword2 = (pword & ~LH_PT_INTERIOR) | interior;
}

#endif
