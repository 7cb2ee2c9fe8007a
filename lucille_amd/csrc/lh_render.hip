/*
 * lh_render.hip -- the callers on either side of the ray query, batched per tile
 * and kept on the device: camera rays, hit epilogue, ambient-occlusion ray
 * producer and the radiance reduction.
 *
 * Reference (CPU, one pixel / one ray at a time):
 *   subsample + sample_subpixel        src/render/render.c:715-861
 *   ri_camera_get_pos_and_dir          src/ri/camera.c:248-318
 *   ri_intersection_state_build        src/render/intersection_state.c:99-248
 *   ri_ortho_basis                     src/render/reflection.c:311-333
 *   calculate_occlusion,
 *     ri_transport_ambientocclusion    src/transport/ambientocclusion.c:42-151,332-415
 *   render_bucket / bucket_write       src/render/render.c:1107-1166,919-983
 *
 * Wavefront pipeline of one tile (all buffers resident in HBM, no host round trip
 * except one 8-byte hit count):
 *
 *   k_primary_rays  -> [trace closest] -> k_hit_count / k_scan_blocks / k_ao_setup
 *                   -> k_ao_rays -> [trace any] -> k_ao_resolve
 *
 * All geometry arithmetic is fp64 with the reference's operation order and no FMA
 * contraction, so primary rays, P, Ng, Ns and the basis equal the oracle's bits;
 * AO directions use a counter-based RNG by default (the reference's MT19937 stream
 * is order-dependent) or caller-supplied uniforms (parity replay), and the device
 * libm's sin/cos.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/lucille_hip.h"
#include "lh_device.h"
#include "lh_ao.h"

namespace {

#include "lh_pt.h"

/* the pixels of one device batch: nbands full-rows-of-w bands of band_rows lines each, band b starting at frame line
 * band_y0[b] (or y0 when there is a single band: a plain rectangle).  Pixel index p of the batch: band = p / (w * band_rows),
 * line = (p / w) % band_rows, column = p % w.  A rank's interleaved shards of a frame are ONE such batch: one set of kernel
 * launches -- and one kernel drain, ~1.7 ms on BASELINE config 5 where a few grazing AO rays walk thousands of floor boxes
 * (profiles/README.md r02c) -- per frame instead of one per shard.  Lines >= height (a ragged last band) produce rays that
 * start far outside the scene and miss. */
struct Region {
    int x0, w, band_rows, nbands, height, y0;
    const int *band_y0;
};

/* 32-bit index arithmetic: a batch holds fewer than 2^32 samples (checked by the callers); the 64-bit divisions by run-time values
 * that stood here cost ~100 instructions each, five per thread of k_primary_rays and k_ao_setup */
__device__ __forceinline__ void region_pixel(const Region &rg, uint32_t pix, int &px, int &py)
{
    const uint32_t per = (uint32_t)rg.w * (uint32_t)rg.band_rows;
    const uint32_t band = pix / per, within = pix - band * per;
    const uint32_t line = within / (uint32_t)rg.w;
    px = rg.x0 + (int)(within - line * (uint32_t)rg.w);
    py = (rg.band_y0 ? rg.band_y0[band] : rg.y0) + (int)line;
}

/* radical inverse permutation of init_sigma (render.c:870-917) */
__device__ __forceinline__ unsigned sigma_of(unsigned i, unsigned period)
{
    unsigned digit = period, inverse = 0;
    for (unsigned bits = i; bits; bits >>= 1) { digit >>= 1; if (bits & 1) inverse += digit; }
    return inverse;
}

/* one thread per pixel sub-sample: sample id = ((ly*w + lx)*ys + sy)*xs + sx */
__global__ void k_primary_rays(DevCamera cam, Region rg, int xs, int ys,
                               double *__restrict__ org, double *__restrict__ dir)
{
    LH_NC
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)rg.w * rg.band_rows * rg.nbands * xs * ys;
    if (id >= total) return;
    const uint32_t id32 = (uint32_t)id, pix = id32 / (uint32_t)(xs * ys), sub = id32 - pix * (uint32_t)(xs * ys);
    const int sy = (int)(sub / (uint32_t)xs), sx = (int)(sub - (uint32_t)sy * (uint32_t)xs);
    int px, py;
    region_pixel(rg, pix, px, py);
    if (py >= rg.height) {          /* below the frame (ragged last band): a ray that misses everything */
        org[3 * id] = 1.0e30; org[3 * id + 1] = 1.0e30; org[3 * id + 2] = 1.0e30;
        dir[3 * id] = 1.0; dir[3 * id + 1] = 1.0; dir[3 * id + 2] = 1.0;
        return;
    }
    /* sample_subpixel (render.c:830-861) */
    const unsigned j = (unsigned)sx & ((unsigned)xs - 1), k = (unsigned)sy & ((unsigned)xs - 1);
    double jx = (double)sx + (double)sigma_of(k, (unsigned)xs) / (double)xs;
    double jy = (double)sy + (double)sigma_of(j, (unsigned)ys) / (double)ys;
    jx /= (double)xs; jy /= (double)ys;
    jx += 0.5 / (xs * xs); jy += 0.5 / (ys * ys);
    /* ri_camera_get_pos_and_dir (camera.c:248-318) */
    const double x = (double)(px + jx), y = (double)(py + jy);
    const double W = cam.width, H = cam.height;
    const float sign = cam.rh ? -1.0f : 1.0f;
    double v[4], o[4] = {0.0, 0.0, 0.0, 1.0}, pos[4], dp[4];
    v[0] = (2.0f * x - W) / W; v[1] = (2.0f * y - H) / H; v[2] = sign * cam.flength; v[3] = 1.0;
    if (cam.ortho) { o[0] = v[0]; o[1] = v[1]; v[2] = sign * 1.0; }      /* camera.c:285-301 */
    for (int c = 0; c < 4; c++) {
        pos[c] = 0.0; dp[c] = 0.0;
        for (int r = 0; r < 4; r++) { pos[c] += o[r] * cam.c2w[4 * r + c]; dp[c] += v[r] * cam.c2w[4 * r + c]; }
    }
    double d[3] = {dp[0] - pos[0], dp[1] - pos[1], dp[2] - pos[2]};
    vnormalize(d);
    org[3 * id] = pos[0]; org[3 * id + 1] = pos[1]; org[3 * id + 2] = pos[2];
    dir[3 * id] = d[0]; dir[3 * id + 1] = d[1]; dir[3 * id + 2] = d[2];
}

/* ---- deterministic compaction of the primary hits (sample order) ---------- */
__global__ void k_hit_count(size_t n, const uint32_t *__restrict__ prim, uint32_t *__restrict__ block_counts)
{
    __shared__ uint32_t wsum[4];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool hit = (i < n) && (prim[i] != LH_MISS_PRIM);
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

/* exclusive scan of block_counts in place (one workgroup); total -> *total_out */
__global__ void k_scan_blocks(uint32_t nblocks, uint32_t *__restrict__ block_counts, unsigned long long *total_out)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < nblocks) ? block_counts[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            uint32_t t = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) block_counts[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

struct DevNormals { const double *nrm; };   /* 9 doubles per prim (n0 n1 n2), NaN n0.x => none */

/* one thread per primary sample: slot = exclusive scan of the hit flags; writes the
 * per-hit record {org(3), basis(9)} (12 doubles) and slot_of_sample */
__global__ void k_ao_setup(size_t n, const lh_dev_scene_t sc, const double *__restrict__ nrm9,
                           const double *__restrict__ org, const double *__restrict__ dir,
                           const uint32_t *__restrict__ prim, const double *__restrict__ t,
                           const double *__restrict__ u, const double *__restrict__ v,
                           const uint32_t *__restrict__ block_offsets, uint32_t *__restrict__ slot_of_sample,
                           double *__restrict__ hitrec, unsigned long long *__restrict__ slot_key,
                           const Region rg, int spp, int full_width)
{
    LH_NC
    __shared__ uint32_t wsum[4];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool hit = (i < n) && (prim[i] != LH_MISS_PRIM);
    const unsigned long long m = __ballot(hit);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < wv; k++) woff += wsum[k];
    if (i < n) slot_of_sample[i] = LH_MISS_PRIM;
    if (!hit) return;
    const uint32_t slot = block_offsets[blockIdx.x] + woff + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    slot_of_sample[i] = slot;
    {   /* absolute sample key (frame position, not tile position): keeps the built-in
         * RNG independent of how the frame is tiled or sharded */
        int ipx, ipy;
        const uint32_t i32 = (uint32_t)i, ipix = i32 / (uint32_t)spp;
        region_pixel(rg, ipix, ipx, ipy);
        const unsigned long long px = (unsigned long long)ipx, py = (unsigned long long)ipy;
        slot_key[slot] = (py * (unsigned long long)full_width + px) * (unsigned long long)spp + (i32 - ipix * (uint32_t)spp);      /* < 2^34: checked by the caller */
    }

    /* ri_intersection_state_build (intersection_state.c:99-248): P, Ng, Ns */
    const uint32_t p = prim[i];
    const double *tv = (const double *)sc.tri64 + 9 * (size_t)p;
    const double tt = t[i], uu = u[i], vv = v[i];
    double P[3], Ng[3], Ns[3], v01[3], v02[3];
    for (int k = 0; k < 3; k++) P[k] = org[3 * i + k] + dir[3 * i + k] * tt;
    for (int k = 0; k < 3; k++) { v01[k] = tv[3 + k] - tv[k]; v02[k] = tv[6 + k] - tv[k]; }
    vcross(Ng, v01, v02); vnormalize(Ng);
    bool has_n = false;
    if (nrm9) { const double n0x = nrm9[9 * (size_t)p]; has_n = (n0x == n0x); }
    if (has_n) {
        const double *nn = nrm9 + 9 * (size_t)p; const double w = 1.0 - uu - vv;
        for (int k = 0; k < 3; k++) { const double a = nn[k] * w, b = nn[3 + k] * uu, c = nn[6 + k] * vv; Ns[k] = (a + b) + c; }
    } else {
        Ns[0] = Ng[0]; Ns[1] = Ng[1]; Ns[2] = Ng[2];
    }
    /* flat-shaded: the origin triangle cannot occlude its own AO rays (lh_ao.h) -- unless it is degenerate (Ng = 0) */
    const bool noself = !has_n && (Ng[0] != 0.0 || Ng[1] != 0.0 || Ng[2] != 0.0);
    slot_key[slot] |= (unsigned long long)(noself ? p : LH_SLOT_NOSELF) << LH_SLOTKEY_BITS;
    /* ri_ortho_basis(basis, Ns) (reflection.c:311-333) and the 1e-6 offset (ambientocclusion.c:65-73) */
    double b0[3], b1[3] = {0.0, 0.0, 0.0};
    int ax = 3;
    for (int k = 0; k < 3; k++) if (Ns[k] < 0.6 && Ns[k] > -0.6) { ax = k; break; }
    if (ax >= 3) ax = 0;
    b1[ax] = 1.0;
    vcross(b0, b1, Ns); vnormalize(b0);
    vcross(b1, Ns, b0); vnormalize(b1);
    double *r = hitrec + 12 * (size_t)slot;
    const double eps = 1.0e-6;
    for (int k = 0; k < 3; k++) { r[k] = P[k] + Ns[k] * eps; r[3 + k] = b0[k]; r[6 + k] = b1[k]; r[9 + k] = Ns[k]; }
}

/* counter-based uniforms in [0,1) with 32-bit resolution (like randomMT2's y*2^-32): lh_ao.h */
__device__ __forceinline__ uint32_t mix32(uint64_t x) { return lh_mix32(x); }

/* one thread per AO ray: ray id = slot*N + (j*ntheta + i) (calculate_occlusion's loop order).
 * rnd != NULL: the caller's uniforms (2 per ray), fp64 throughout -- the parity replay of the reference's
 * MT19937 stream.  rnd == NULL: the built-in generator of lh_ao.h -- bit for bit the rays the any-hit kernel
 * generates in its refill when the tile runs fused (this kernel is then only used to show them). */
__global__ void k_ao_rays(size_t nslots, int ntheta, int nphi, unsigned long long seed,
                          const double *__restrict__ hitrec, const double *__restrict__ rnd /* 2 per ray or NULL */,
                          const unsigned long long *__restrict__ slot_key, double *__restrict__ org, double *__restrict__ dir)
{
    LH_NC
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int N = ntheta * nphi;
    if (id >= nslots * (size_t)N) return;
    const size_t slot = id / N; const int r = (int)(id % N);
    const double *h = hitrec + 12 * slot;
    if (!rnd) {
        double ox, oy, oz, dx, dy, dz;
        lh_ao_ray_builtin(h, slot_key[slot], seed, ntheta, nphi, r, ox, oy, oz, dx, dy, dz);
        org[3 * id] = ox; org[3 * id + 1] = oy; org[3 * id + 2] = oz;
        dir[3 * id] = dx; dir[3 * id + 1] = dy; dir[3 * id + 2] = dz;
        return;
    }
    const int i = r % ntheta, j = r / ntheta;
    const double r0 = rnd[2 * id], r1 = rnd[2 * id + 1];
    const double z0 = ((double)(uint32_t)i + r0) / (double)(uint32_t)ntheta;
    const double z1 = ((double)(uint32_t)j + r1) / (double)(uint32_t)nphi;
    const double cos_theta = sqrt(z0), phi = 2.0 * 3.14159265358979323846 * z1;
    double sp, cp;
    sincos(phi, &sp, &cp);
    const double d0 = cp * cos_theta, d1 = sp * cos_theta, d2 = sqrt(1.0 - cos_theta * cos_theta);
    for (int k = 0; k < 3; k++) {
        org[3 * id + k] = h[k];
        dir[3 * id + k] = d0 * h[3 + k] + d1 * h[6 + k] + d2 * h[9 + k];
    }
}

/* one thread per pixel: accumulates its sub-samples exactly like subsample()
 * (render.c:749-822) and bucket_write (:962-975): rgb[(h-1-ly)*w + lx] */
__global__ void k_ao_resolve(int w, int h, int band_rows, int xs, int ys, int N, const uint32_t *__restrict__ slot_of_sample,
                             const uint8_t *__restrict__ occ, const unsigned int *__restrict__ occ_count,
                             float *__restrict__ rgb, unsigned long long *__restrict__ occ_total)
{
    LH_NC
    const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inside = pix < (size_t)w * h;
    /* h = all lines of the batch (nbands * band_rows); every band is flipped within itself, as a tile is */
    const int lx = (int)(pix % w), line = (int)(pix / w), band = line / band_rows, ly = line % band_rows;
    double accum = 0.0;
    unsigned int nocc = 0;
    const int S = inside ? xs * ys : 0;
    for (int s = 0; s < S; s++) {
        const uint32_t slot = slot_of_sample[pix * S + s];
        double rad = 0.0;
        if (slot != LH_MISS_PRIM) {
            double occlusion = 0.0;
            if (occ_count) {                       /* fused AO stage: occluded rays were counted per slot */
                const unsigned int c = occ_count[slot];
                occlusion = (double)c; nocc += c;
            } else {
                const uint8_t *o = occ + (size_t)slot * N;
                for (int r = 0; r < N; r++) if (o[r]) { occlusion += 1.0; nocc++; }
            }
            const double ns = (double)(uint32_t)N;
            rad = 1.0 * (ns - occlusion) / ns;
        }
        accum = accum + rad;
    }
    if (inside) {
        const double val = accum * ((double)1.0 / (xs * ys));
        float f = (float)val;
        if (f < 0.0f) f = 0.0f;
        float *o = rgb + 3 * ((size_t)(band * band_rows + (band_rows - 1 - ly)) * w + lx);
        o[0] = f; o[1] = f; o[2] = f;
    }
    if (occ_total) {
        /* one atomic per WORKGROUP on one of 64 counters: atomics on one address are served one after the other (~5 ns each) --
         * one per wave, 262 144 of them on a 4096^2 frame, were 1.2 of the 1.35 ms this kernel took */
        __shared__ unsigned int wsum[4];
        for (int off = 32; off > 0; off >>= 1) nocc += (unsigned int)__shfl_xor((int)nocc, off);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = nocc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (tot) atomicAdd(occ_total + (blockIdx.x & 63u), (unsigned long long)tot);
        }
    }
}


/* ---- the whole hit epilogue for a batch of hit records --------------------------------------
 * ri_intersection_state_build (intersection_state.c:99-248), every member a shader reads: P, Ng, Ns, tangent,
 * binormal, colour, st, I (normalised direction), inside.  Attributes are per-primitive SoA arrays in
 * primitive-id order (9 doubles = 3 corners x xyz; st: 6 doubles; first value NaN = the mesh has none):
 * one thread per ray, LH_STATE_DOUBLES per record, misses are left untouched. */
__global__ void k_state_build(size_t n, const lh_dev_scene_t sc, const double *__restrict__ nrm9, const double *__restrict__ col9,
                              const double *__restrict__ tan9, const double *__restrict__ bin9, const double *__restrict__ st6,
                              const uint8_t *__restrict__ inside_of_prim,
                              const double *__restrict__ org, const double *__restrict__ dir, const uint32_t *__restrict__ prim,
                              const double *__restrict__ t, const double *__restrict__ u, const double *__restrict__ v,
                              double *__restrict__ state)
{
    LH_NC
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = prim[i];
    if (p == LH_MISS_PRIM) return;
    const double *tv = (const double *)sc.tri64 + 9 * (size_t)p;
    const double tt = t[i], uu = u[i], vv = v[i], w = 1.0 - uu - vv;
    double *o = state + LH_STATE_DOUBLES * i;
    double Ng[3], v01[3], v02[3], I[3];
    for (int k = 0; k < 3; k++) { o[k] = org[3 * i + k] + dir[3 * i + k] * tt; I[k] = dir[3 * i + k]; }
    vnormalize(I);
    for (int k = 0; k < 3; k++) { v01[k] = tv[3 + k] - tv[k]; v02[k] = tv[6 + k] - tv[k]; }
    vcross(Ng, v01, v02); vnormalize(Ng);
    bool has_n = false, has_tb = false;
    if (nrm9) { const double x = nrm9[9 * (size_t)p]; has_n = (x == x); }
    if (has_n && tan9 && bin9) { const double x = tan9[9 * (size_t)p], y = bin9[9 * (size_t)p]; has_tb = (x == x) && (y == y); }
#define LH_LERP9(A, dst) { const double *q_ = (A) + 9 * (size_t)p; \
        for (int k = 0; k < 3; k++) { const double a_ = q_[k] * w, b_ = q_[3 + k] * uu, c_ = q_[6 + k] * vv; (dst)[k] = (a_ + b_) + c_; } }
    for (int k = 0; k < 3; k++) o[3 + k] = Ng[k];
    if (has_n) LH_LERP9(nrm9, o + 6) else for (int k = 0; k < 3; k++) o[6 + k] = Ng[k];
    if (has_tb) { LH_LERP9(tan9, o + 9) LH_LERP9(bin9, o + 12) }
    else {
        /* ri_ortho_basis(tmpbasis, Ng) (reflection.c:311-333) */
        double b0[3], b1[3] = {0.0, 0.0, 0.0};
        int ax = 3;
        for (int k = 0; k < 3; k++) if (Ng[k] < 0.6 && Ng[k] > -0.6) { ax = k; break; }
        if (ax >= 3) ax = 0;
        b1[ax] = 1.0;
        vcross(b0, b1, Ng); vnormalize(b0);
        vcross(b1, Ng, b0); vnormalize(b1);
        for (int k = 0; k < 3; k++) { o[9 + k] = b0[k]; o[12 + k] = b1[k]; }
    }
    bool has_c = false, has_st = false;
    if (col9) { const double x = col9[9 * (size_t)p]; has_c = (x == x); }
    if (has_c) LH_LERP9(col9, o + 15) else { o[15] = 1.0; o[16] = 1.0; o[17] = 1.0; }
    if (st6) { const double x = st6[6 * (size_t)p]; has_st = (x == x); }
    if (has_st) {
        /* lerp_uv (intersection_state.c:58-76): (1-u-v) st0 + u st1 + v st2 */
        const double *q = st6 + 6 * (size_t)p;
        o[18] = (w * q[0] + uu * q[2]) + vv * q[4];
        o[19] = (w * q[1] + uu * q[3]) + vv * q[5];
    } else { o[18] = 0.0; o[19] = 0.0; }
    for (int k = 0; k < 3; k++) o[20 + k] = I[k];
    o[23] = (inside_of_prim && inside_of_prim[p]) ? 1.0 : 0.0;
#undef LH_LERP9
}

/* ------------------------------------------------------------------------------------ */
/* wavefront path tracer (BASELINE config 4)                                */
/*                                                                                       */
/* The reference's Kajiya path tracer (src/transport/pathtrace.c) is dead code that no     */
/* longer compiles; what is kept is its documented algorithm (pathtrace.c:189-314,407-537) */
/* re-expressed as closest-hit batches:                                                    */
/*   camera sample through a random sub-pixel position (sample_pixel :316-352);            */
/*   miss -> the IBL radiance in the ray's direction (ri_texture_ibl_fetch, texture.c:     */
/*   238-276: angular map, bilinear) or a constant environment;                            */
/*   hit  -> Russian roulette on d + s + t = the averages of the material's kd, ks, kt     */
/*   (russian_roulette :407-430), reflection type D / S / T drawn in those proportions     */
/*   (sample_reflection_type :432-459), next direction: cosine-weighted about the normal   */
/*   (sample_cosweight :500-531), mirror reflection (ri_reflect) or refraction with the    */
/*   material's ior, entering / leaving tracked per path, total internal reflection turns  */
/*   T into S (sample_outdir :461-498, ri_refract reflection.c:69-128); throughput *=      */
/*   reflectance x the interpolated vertex colour (brdf :533-565), to a vertex limit.      */
/* Two departures, both stated in DESIGN.md: (1) the reference multiplies the BRDF value    */
/* (kd / pi ...) without the cosine / pdf factor and never compensates the roulette: the    */
/* default here is the unbiased estimator of the same sampling scheme (a furnace renders    */
/* white), LH_PT_REFERENCE_WEIGHTS reproduces the reference's factors; (2) its final        */
/* "connect" step (one more sampled direction + visibility ray, light_sample :354-382) is   */
/* what the extension ray already is in wavefront form: a path that leaves the scene        */
/* collects the environment in the direction it left.  Parity for this row is at the        */
/* ri_raytrace level (every bounce goes through the closest-hit kernel and fp64 resolve).   */
/* ------------------------------------------------------------------------------------ */

/* After the closest-hit launch of bounce `depth`: the live paths are DECIDED (k_pt_decide) and the survivors SCATTERED
 * (k_pt_scatter).  Round 2 ran decide -> flag count -> scan -> emit with the survivor count read back by the host before the
 * next launch (110 dependent launches and 28 host round trips per 64-sample pass); rounds 3-4 ran ONE kernel per bounce, a
 * workgroup deciding 2048 consecutive paths, listing its survivors in LDS and scattering them -- at the scatter's 136-146
 * VGPRs and 96 bytes of scratch: three waves per SIMD, every one of them waiting on its loads 61-74 % of the time, the decision
 * loop one dependent load chain per item (profiles/r05_pmc_pt_summary.txt: 31 ms for a 2^30-path pass's first bounce with the
 * vector pipes 37 % busy).  Round 5 splits the two halves at the place where their needs differ:
 *   k_pt_decide  -- a streaming pass at eight waves per SIMD: a workgroup takes LH_PT_ITEMS x 256 consecutive paths, loads their
 *     hit words (and, for the misses, throughputs) in one batch, decides them (miss -> radiance = throughput x environment, the
 *     path ends; hit -> vertex limit and Russian roulette on d + s + t, pathtrace.c:407-430 -> ends with radiance 0, or goes
 *     on), reserves the survivors' slots with one atomic on counts[depth + 1] and writes each survivor's SOURCE INDEX into its
 *     slot of the next bounce's path-word array, in path order within the workgroup;
 *   k_pt_scatter -- one thread per survivor slot j: source index i = path_of2[j], hit epilogue, lobe, next ray, throughput,
 *     written once into slot j (the path word over the source index it has just read).
 * The launch's own path count is counts[depth], left there by the previous bounce: nothing comes back to the host inside a
 * pass.  Every path writes its radiance exactly once (where it ends), so the buffer needs no clearing.  Slot order across
 * workgroups depends on scheduling; a path's arithmetic does not (keys are (pixel, sample, bounce)): the frame is the same. */
#define LH_PT_ITEMS 8
struct PtPass {
    DevMaterial override_mat; DevEnv env;
    const PtCamSrc *cam;                 /* bounce 0: the pass's rays are the camera rays, regenerated from the path id */
    unsigned long long seed;
    int use_override, ref_weights, depth, max_depth, s0, spp, x0, y0, w, band_rows, band_stride, full_width;
    PtDivs dv;                           /* divisions by spp, w, band_rows as multiplications (lh_pt.h) */
};

/* PROBE: the environment is a light probe (a miss evaluates it in the path's direction: fp64 acos / sqrt / divisions, kept out of the
 * constant-environment instantiation, whose decision loop is unrolled over its batch).
 * A workgroup's span is LH_PT_ROUNDS batches of LH_PT_ITEMS x 256 paths behind ONE atomic on the survivor count: every workgroup of
 * every launch adds to the same word, and same-address atomics retire one every ~20 ns -- at 2048 paths an atomic a 2^30-path
 * pass's first decision took 6.0 ms for its 13 GB; at 8192 the atomics are a quarter of that. */
#define LH_PT_ROUNDS 4

/* A path's radiance goes into its pixel's accumulator where the path ends -- three 64-bit FIXED-POINT sums per pixel of the pass
 * (32 fraction bits; a sample clamped to +-2^18: the 2^12 samples of a pixel that lh_tile.hip allows in one pass stay below 2^62) -- instead of into a 12-byte record per path that a resolve pass sums afterwards (rounds 2-4): those records
 * were written where the paths ended, a few of every 128-byte line per bounce (6.7 + 5.6 + ... GB of partial-line writes and
 * 12.9 GB read back per 2^30-path pass, 12.9 GB of HBM held).  Integer sums do not depend on the order of their terms: a pixel
 * is the same whatever the slot order, tiling or sharding, bit for bit, as before.  Lanes of a wave hold consecutive slots,
 * i.e. runs of paths of the same pixel: the runs are summed across the wave first (a segmented scan over run numbers), one
 * atomic per run and channel. */
#define LH_PT_FIX_CLAMP 262144.0f
__device__ __forceinline__ unsigned long long pt_fix(float r)
{
    r = fminf(fmaxf(r, -LH_PT_FIX_CLAMP), LH_PT_FIX_CLAMP);          /* (a NaN never gets here: pt_accumulate drops it) */
    float fl = floorf(r), fr = r - fl;          /* fr is exact in fp32 -- except for a tiny negative r (-1e-10: floor -1, r + 1 rounds to 1.0f) */
    if (fr >= 1.0f) { fr = 0.0f; fl += 1.0f; }  /* ... which is the carry: without it (uint32_t)(1.0f * 2^32) is out of range */
    const uint32_t lo = (uint32_t)(fr * 4294967296.0f);
    return ((unsigned long long)(long long)(int)fl << 32) | (unsigned long long)lo;
}

__device__ __forceinline__ void pt_accumulate(unsigned long long *__restrict__ accum, uint32_t pix, bool ends, float r0, float r1, float r2, int lane)
{
    const bool v0 = ends && r0 == r0 && r0 != 0.0f, v1 = ends && r1 == r1 && r1 != 0.0f, v2 = ends && r2 == r2 && r2 != 0.0f;      /* a NaN channel adds nothing */
    if (__ballot(v0 | v1 | v2) == 0ull) return;                        /* wave-uniform: nothing ends here with light */
    unsigned long long a = v0 ? pt_fix(r0) : 0ull, b = v1 ? pt_fix(r1) : 0ull, c = v2 ? pt_fix(r2) : 0ull;
    /* run number: the pixel may come back later in the wave (two workgroups' survivors of one pixel with a third's between) */
    const uint32_t before = (uint32_t)__shfl_up((int)pix, 1);
    const unsigned long long heads = __ballot(lane == 0 || before != pix);
    const uint32_t run = (uint32_t)__popcll(heads & ((2ull << lane) - 1ull));
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t ru = (uint32_t)__shfl_up((int)run, off);
        const unsigned long long au = __shfl_up(a, off), bu = __shfl_up(b, off), cu = __shfl_up(c, off);
        if (lane >= off && ru == run) { a += au; b += bu; c += cu; }
    }
    const bool last = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (last && (a | b | c) != 0ull) {
        unsigned long long *o = accum + 3 * (size_t)pix;
        if (a) atomicAdd(o, a);
        if (b) atomicAdd(o + 1, b);
        if (c) atomicAdd(o + 2, c);
    }
}

/* the same when every path that ends here with light carries the SAME radiance (camera rays that leave the scene under a constant
 * environment -- half of a frame's paths): a run's sum is its count x that value, counted from the wave's ballots, no scan */
__device__ __forceinline__ void pt_accumulate_uniform(unsigned long long *__restrict__ accum, uint32_t pix, bool ends, const float e[3], int lane)
{
    const unsigned long long mm = __ballot(ends);
    if (mm == 0ull) return;
    const uint32_t before = (uint32_t)__shfl_up((int)pix, 1);
    const unsigned long long heads = __ballot(lane == 0 || before != pix);
    const bool last = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (last) {
        const unsigned long long upto = (2ull << lane) - 1ull;
        const int start = 63 - __clzll((long long)(heads & upto));                    /* bit `lane`'s run begins at its nearest head below */
        const unsigned long long cnt = (unsigned long long)__popcll(mm & upto & ~((1ull << start) - 1ull));
        if (cnt) {
            unsigned long long *o = accum + 3 * (size_t)pix;
            for (int c = 0; c < 3; c++) if (e[c] == e[c] && e[c] != 0.0f) atomicAdd(o + c, cnt * pt_fix(e[c]));
        }
    }
}

template <bool FIRST, bool PROBE>
__global__ __launch_bounds__(256) void k_pt_decide(const PtPass ps, const uint32_t *__restrict__ prim_mesh, const DevMaterial *__restrict__ materials,
                                                   uint32_t *__restrict__ counts, const double *__restrict__ dir, const uint32_t *__restrict__ prim,
                                                   const uint32_t *__restrict__ path_of, const float *__restrict__ thr, unsigned long long *__restrict__ accum,
                                                   uint32_t *__restrict__ src_of)
{
    LH_NC
    constexpr int NB = LH_PT_ROUNDS * LH_PT_ITEMS * 4;             /* ballots of a span: (round, item, wave) */
    __shared__ unsigned long long sbal[NB];
    __shared__ uint32_t soff[NB];
    __shared__ uint32_t gbase;
    const uint32_t n = counts[ps.depth];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t batch = 256u * LH_PT_ITEMS, span = batch * LH_PT_ROUNDS;
    for (uint32_t base = blockIdx.x * span; base < n; base += gridDim.x * span) {      /* n < 2^31: no wrap */
#pragma unroll 1
        for (int r = 0; r < LH_PT_ROUNDS; r++) {
            const uint32_t b0 = base + (uint32_t)r * batch;
            /* the batch's loads first, all of them in flight together: hit words, path words, the misses' throughputs */
            uint32_t pr[LH_PT_ITEMS], pw[LH_PT_ITEMS];
            float g0[LH_PT_ITEMS], g1[LH_PT_ITEMS], g2[LH_PT_ITEMS];
#pragma unroll
            for (int k = 0; k < LH_PT_ITEMS; k++) {
                const uint32_t i = b0 + (uint32_t)k * 256u + threadIdx.x;
                pr[k] = i < n ? prim[i] : 0u;
                pw[k] = FIRST ? i : (i < n ? path_of[i] : 0u);
            }
#pragma unroll
            for (int k = 0; k < LH_PT_ITEMS; k++) {
                const uint32_t i = b0 + (uint32_t)k * 256u + threadIdx.x;
                g0[k] = 1.0f; g1[k] = 1.0f; g2[k] = 1.0f;
                if (!FIRST && i < n && pr[k] == LH_MISS_PRIM) { g0[k] = thr[3 * (size_t)i]; g1[k] = thr[3 * (size_t)i + 1]; g2[k] = thr[3 * (size_t)i + 2]; }
            }
#pragma unroll
            for (int k = 0; k < LH_PT_ITEMS; k++) {
                const uint32_t i = b0 + (uint32_t)k * 256u + threadIdx.x;
                bool go = false;
                const uint32_t path = pw[k] & ~LH_PT_INTERIOR;
                float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
                if (i < n) {
                    const uint32_t p = pr[k];
                    if (p == LH_MISS_PRIM) {
                        float e[3];
                        if (PROBE) {
                            double o[3], d[3];
                            if (FIRST) pt_camera_ray(ps.cam, i, o, d);
                            else { d[0] = dir[3 * (size_t)i]; d[1] = dir[3 * (size_t)i + 1]; d[2] = dir[3 * (size_t)i + 2]; }
                            env_fetch(ps.env, d[0], d[1], d[2], e);
                        } else { e[0] = ps.env.rgb[0]; e[1] = ps.env.rgb[1]; e[2] = ps.env.rgb[2]; }
                        if (FIRST) { r0 = e[0]; r1 = e[1]; r2 = e[2]; }
                        else { r0 = g0[k] * e[0]; r1 = g1[k] * e[1]; r2 = g2[k] * e[2]; }
                    } else {
                        const double ksum = ps.use_override ? ps.override_mat.asum9 : materials[prim_mesh[p]].asum9;
                        go = pt_survives(ksum, pt_key(ps.seed, path, ps.spp, ps.s0, ps.x0, ps.y0, ps.w, ps.band_rows, ps.band_stride, ps.dv, ps.full_width, ps.depth), ps.depth, ps.max_depth);
                    }
                }
                /* a path the roulette ended adds nothing */
                if (FIRST && !PROBE) pt_accumulate_uniform(accum, lh_div(path, ps.dv.spp), i < n && pr[k] == LH_MISS_PRIM, ps.env.rgb, lane);
                else pt_accumulate(accum, lh_div(path, ps.dv.spp), i < n && !go, r0, r1, r2, lane);
                const unsigned long long m = __ballot(go);
                if (lane == 0) sbal[(r * LH_PT_ITEMS + k) * 4 + wv] = m;
            }
        }
        __syncthreads();
        /* NB counts -> exclusive offsets in path order (round, item, wave), by the first wave: two entries a lane */
        if (wv == 0) {
            static_assert(NB == 128, "two ballots a lane");
            const uint32_t c0 = (uint32_t)__popcll(sbal[2 * lane]), c1 = (uint32_t)__popcll(sbal[2 * lane + 1]);
            uint32_t incl = c0 + c1;
            for (int off = 1; off < 64; off <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, off); if (lane >= off) incl += up; }
            soff[2 * lane] = incl - c0 - c1; soff[2 * lane + 1] = incl - c1;
            if (lane == 63) gbase = incl ? atomicAdd(&counts[ps.depth + 1], incl) : 0u;
        }
        __syncthreads();
        const uint32_t g = gbase;
#pragma unroll 4
        for (int q = 0; q < LH_PT_ROUNDS * LH_PT_ITEMS; q++) {
            const unsigned long long m = sbal[q * 4 + wv];
            if ((m >> lane) & 1ull) src_of[g + soff[q * 4 + wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = base + (uint32_t)q * 256u + threadIdx.x;
        }
        __syncthreads();
    }
}

template <bool FIRST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_pt_scatter(const PtPass ps, const lh_dev_scene_t sc, const double *__restrict__ nrm9,
                                                    const double *__restrict__ col9, const uint32_t *__restrict__ prim_mesh,
                                                    const DevMaterial *__restrict__ materials, const uint32_t *__restrict__ counts,
                                                    const double *__restrict__ org, const double *__restrict__ dir,
                                                    const uint32_t *__restrict__ prim, const double *__restrict__ t,
                                                    const double *__restrict__ u, const double *__restrict__ v,
                                                    const uint32_t *__restrict__ path_of, const float *__restrict__ thr,
                                                    double *__restrict__ org2, double *__restrict__ dir2,
                                                    uint32_t *path_of2 /* in: the survivor's source index; out: its path word */, float *__restrict__ thr2)
{
    LH_NC
    const uint32_t n2 = counts[ps.depth + 1];
#pragma unroll 1
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n2; j += gridDim.x * 256u) {
        const uint32_t i = path_of2[j];
        const uint32_t pword = FIRST ? i : path_of[i];
        const uint32_t path = pword & ~LH_PT_INTERIOR;
        const uint32_t p = prim[i];
        const double tt = t[i], uu = u[i], vv = v[i];
        float G[3] = {1.0f, 1.0f, 1.0f};
        if (!FIRST) { G[0] = thr[3 * (size_t)i]; G[1] = thr[3 * (size_t)i + 1]; G[2] = thr[3 * (size_t)i + 2]; }
        double Or[3], D[3];
        if (FIRST) pt_camera_ray(ps.cam, i, Or, D);
        else {
            Or[0] = org[3 * (size_t)i]; Or[1] = org[3 * (size_t)i + 1]; Or[2] = org[3 * (size_t)i + 2];
            D[0] = dir[3 * (size_t)i]; D[1] = dir[3 * (size_t)i + 1]; D[2] = dir[3 * (size_t)i + 2];
        }
        const uint64_t key = pt_key(ps.seed, path, ps.spp, ps.s0, ps.x0, ps.y0, ps.w, ps.band_rows, ps.band_stride, ps.dv, ps.full_width, ps.depth);
        /* field by field: a whole-struct copy from one of two places keeps the record in scratch (96 bytes a lane, and the lobe's
         * reflectance then read from it by a run-time offset) */
        DevMaterial M;
#define LH_MFIELDS LH_MF(kd[0]) LH_MF(kd[1]) LH_MF(kd[2]) LH_MF(ks[0]) LH_MF(ks[1]) LH_MF(ks[2]) LH_MF(kt[0]) LH_MF(kt[1]) LH_MF(kt[2]) LH_MF(ior) \
                   LH_MF(ad) LH_MF(as) LH_MF(at) LH_MF(asum9) LH_MF(wd) LH_MF(ws) LH_MF(wt)
        M.pad = 0.0f;
        if (ps.use_override) {
#define LH_MF(x) M.x = ps.override_mat.x;
            LH_MFIELDS
#undef LH_MF
        } else {
            const DevMaterial *mp = materials + prim_mesh[p];
#define LH_MF(x) M.x = mp->x;
            LH_MFIELDS
#undef LH_MF
        }
#undef LH_MFIELDS
        double o2[3], O[3]; float G2[3]; uint32_t pw2;
        pt_scatter(sc, nrm9, col9, M, ps.ref_weights, key, p, pword, Or, D, tt, uu, vv, G, o2, O, G2, pw2);
        for (int c = 0; c < 3; c++) { thr2[3 * (size_t)j + c] = G2[c]; org2[3 * (size_t)j + c] = o2[c]; dir2[3 * (size_t)j + c] = O[c]; }
        path_of2[j] = pw2;
    }
}

/* counts[0] = the pass's paths, counts[1 ..] = 0; the pass's camera-ray source into device memory */
__global__ void k_pt_begin(uint32_t *counts, uint32_t paths, int nentries, const PtCamSrc src, PtCamSrc *dst)
{
    for (int i = threadIdx.x; i < nentries; i += blockDim.x) counts[i] = i == 0 ? paths : 0u;
    if (threadIdx.x == 0) *dst = src;
}

/* per pixel: add the mean of this pass's samples -- the pixel's three fixed-point sums (pt_accumulate) as floats x 1 / spp_total --
 * and leave the sums zero for the next pass.  (Rounds 2-4 summed 12-byte per-path records here, staged through LDS: 3.3 ms of a
 * 2^30-path pass.) */
__global__ __launch_bounds__(256) void k_pt_resolve(int w, int h, int band_rows, float inv_total_spp, unsigned long long *__restrict__ accum, float *__restrict__ rgb)
{
    const size_t npix = (size_t)w * h, pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    float sum[3];
    for (int c = 0; c < 3; c++) {
        const long long v = (long long)accum[3 * pix + c];
        accum[3 * pix + c] = 0ull;
        sum[c] = (float)((double)v * 2.3283064365386963e-10);
    }
    /* every band is written in image orientation (its first frame line last); a tile is one band */
    const int lx = (int)(pix % w), ly = (int)(pix / w), band = ly / band_rows;
    float *o = rgb + 3 * ((size_t)(band * band_rows + (band_rows - 1 - (ly - band * band_rows))) * w + lx);
    o[0] += sum[0] * inv_total_spp; o[1] += sum[1] * inv_total_spp; o[2] += sum[2] * inv_total_spp;
}

} /* namespace */

/* ---- host side -------------------------------------------------------------- */

static Region make_region(int x0, int w, int nbands, int band_rows, const int *d_band_y0, int y0, int height)
{
    Region rg; rg.x0 = x0; rg.w = w; rg.band_rows = band_rows; rg.nbands = nbands; rg.height = height; rg.y0 = y0; rg.band_y0 = d_band_y0;
    return rg;
}

/* d_band_y0 NULL: one band starting at y0 (a rectangle); height_limit: lines at or beyond it produce missing rays
 * (pass INT_MAX-like for "no limit": the plain rectangle entry points never clip) */
extern "C" int lh_render_launch_primary_region(const lh_camera_t *cam, int x0, int w, int nbands, int band_rows, const int *d_band_y0,
                                               int y0, int height_limit, int xs, int ys, double *d_org, double *d_dir, void *stream)
{
    DevCamera c;
    for (int i = 0; i < 16; i++) c.c2w[i] = cam->cam2world[i];
    c.flength = cam->flength; c.width = cam->width; c.height = cam->height; c.rh = cam->rh; c.ortho = cam->ortho;
    const size_t total = (size_t)w * band_rows * nbands * xs * ys;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_primary_rays, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       c, make_region(x0, w, nbands, band_rows, d_band_y0, y0, height_limit), xs, ys, d_org, d_dir);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int lh_render_launch_primary(const lh_camera_t *cam, int x0, int y0, int w, int h, int xs, int ys,
                                        double *d_org, double *d_dir, void *stream)
{
    return lh_render_launch_primary_region(cam, x0, w, 1, h, NULL, y0, 0x7fffffff, xs, ys, d_org, d_dir, stream);
}

extern "C" int lh_render_launch_compact(const lh_dev_scene_t *sc, const double *d_nrm9, size_t n, const double *d_org,
                                        const double *d_dir, const uint32_t *d_prim, const double *d_t,
                                        const double *d_u, const double *d_v, uint32_t *d_block_counts,
                                        uint32_t *d_slot_of_sample, double *d_hitrec,
                                        unsigned long long *d_slot_key, int x0, int w, int nbands, int band_rows,
                                        const int *d_band_y0, int y0, int spp, int full_width,
                                        unsigned long long *d_total, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_hit_count, dim3(nb), dim3(256), 0, s, n, d_prim, d_block_counts);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, nb, d_block_counts, d_total);
    hipLaunchKernelGGL(k_ao_setup, dim3(nb), dim3(256), 0, s, n, *sc, d_nrm9, d_org, d_dir, d_prim, d_t, d_u, d_v,
                       d_block_counts, d_slot_of_sample, d_hitrec, d_slot_key,
                       make_region(x0, w, nbands, band_rows, d_band_y0, y0, 0x7fffffff), spp, full_width);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int lh_render_launch_ao_rays(size_t nslots, int ntheta, int nphi, unsigned long long seed,
                                        const double *d_hitrec, const double *d_rnd,
                                        const unsigned long long *d_slot_key, double *d_org, double *d_dir, void *stream)
{
    const size_t total = nslots * (size_t)(ntheta * nphi);
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_ao_rays, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       nslots, ntheta, nphi, seed, d_hitrec, d_rnd, d_slot_key, d_org, d_dir);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int lh_render_launch_resolve(int w, int h, int band_rows, int xs, int ys, int N, const uint32_t *d_slot_of_sample,
                                        const uint8_t *d_occ, const unsigned int *d_occ_count, float *d_rgb,
                                        unsigned long long *d_occ_total, void *stream)
{
    const size_t total = (size_t)w * h;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_ao_resolve, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w, h, band_rows, xs, ys, N, d_slot_of_sample, d_occ, d_occ_count, d_rgb, d_occ_total);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* lh_material_t -> the device record (lh_pt.h: + channel averages, 1 / P(lobe)) */
static void pack_material(const lh_material_t *m, DevMaterial *d)
{
    memset(d, 0, sizeof(*d));
    for (int k = 0; k < 3; k++) { d->kd[k] = m->kd[k]; d->ks[k] = m->ks[k]; d->kt[k] = m->kt[k]; }
    d->ior = m->ior;
    d->ad = ((double)m->kd[0] + m->kd[1] + m->kd[2]) / 3.0; d->as = ((double)m->ks[0] + m->ks[1] + m->ks[2]) / 3.0;
    d->at = ((double)m->kt[0] + m->kt[1] + m->kt[2]) / 3.0;
    d->asum9 = ((double)m->kd[0] + m->kd[1] + m->kd[2] + m->ks[0] + m->ks[1] + m->ks[2] + m->kt[0] + m->kt[1] + m->kt[2]) / 3.0;
    d->wd = (float)(1.0 / d->ad); d->ws = (float)(1.0 / d->as); d->wt = (float)(1.0 / d->at);
}
extern "C" size_t lh_pt_material_bytes(void) { return sizeof(DevMaterial); }
extern "C" void lh_pt_material_pack(const lh_material_t *m, void *out) { pack_material(m, (DevMaterial *)out); }

/* start of a pass: counts[0] = its paths, the camera-ray source (PtCamSrc) into d_cam */
extern "C" int lh_pt_launch_begin(const lh_camera_t *cam, int x0, int y0, int w, int h, int band_rows, int band_stride, int spp, int s0,
                                  unsigned long long seed, void *d_cam, uint32_t *d_counts, int ncounts, void *stream)
{
    PtCamSrc c;
    memset(&c, 0, sizeof(c));
    for (int i = 0; i < 16; i++) c.cam.c2w[i] = cam->cam2world[i];
    c.cam.flength = cam->flength; c.cam.width = cam->width; c.cam.height = cam->height; c.cam.rh = cam->rh; c.cam.ortho = cam->ortho;
    c.seed = seed; c.x0 = x0; c.y0 = y0; c.w = w; c.spp = spp; c.s0 = s0; c.band_rows = band_rows; c.band_stride = band_stride;
    c.dv.spp = lh_div_make((uint32_t)spp); c.dv.w = lh_div_make((uint32_t)w); c.dv.rows = lh_div_make((uint32_t)band_rows);
    const size_t total = (size_t)w * h * spp;
    hipLaunchKernelGGL(k_pt_begin, dim3(1), dim3(256), 0, (hipStream_t)stream, d_counts, (uint32_t)total, ncounts, c, (PtCamSrc *)d_cam);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" size_t lh_pt_cam_bytes(void) { return sizeof(PtCamSrc); }

/* the shading pass of bounce `depth` over counts[depth] paths (at most n_max); survivors are counted into counts[depth + 1] */
extern "C" int lh_pt_launch_shade(size_t n_max, const lh_dev_scene_t *sc, const double *d_nrm9, const double *d_col9,
                                  const uint32_t *d_prim_mesh, const void *d_materials, const lh_material_t *override_mat,
                                  const float env_rgb[3], const void *d_env_map, int env_w, int env_h, int ref_weights,
                                  int depth, int max_depth, unsigned long long seed, int s0, int spp, int x0, int y0, int w,
                                  int band_rows, int band_stride, int full_width, const void *d_cam, uint32_t *d_counts, const double *d_org, const double *d_dir, const uint32_t *d_prim,
                                  const double *d_t, const double *d_u, const double *d_v, const uint32_t *d_path_of,
                                  const float *d_thr, unsigned long long *d_accum, double *d_org2, double *d_dir2, uint32_t *d_path_of2,
                                  float *d_thr2, int ncus, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_max == 0) return 0;
    PtPass ps;
    memset(&ps, 0, sizeof(ps));
    if (override_mat) pack_material(override_mat, &ps.override_mat);
    ps.env.rgb[0] = env_rgb[0]; ps.env.rgb[1] = env_rgb[1]; ps.env.rgb[2] = env_rgb[2];
    ps.env.map = (const float4 *)d_env_map; ps.env.w = env_w; ps.env.h = env_h;
    ps.cam = (const PtCamSrc *)d_cam;
    ps.seed = seed; ps.use_override = override_mat != NULL; ps.ref_weights = ref_weights; ps.depth = depth; ps.max_depth = max_depth;
    ps.s0 = s0; ps.spp = spp; ps.x0 = x0; ps.y0 = y0; ps.w = w; ps.band_rows = band_rows; ps.band_stride = band_stride; ps.full_width = full_width;
    ps.dv.spp = lh_div_make((uint32_t)spp); ps.dv.w = lh_div_make((uint32_t)w); ps.dv.rows = lh_div_make((uint32_t)band_rows);
    const size_t spans = (n_max + 256 * LH_PT_ITEMS * LH_PT_ROUNDS - 1) / (256 * LH_PT_ITEMS * LH_PT_ROUNDS);
    const size_t cus = (size_t)(ncus > 0 ? ncus : 256);
    const unsigned nb = (unsigned)(spans < cus * 16 ? spans : cus * 16);                    /* decide: 2 x the eight resident workgroups of a CU */
    const size_t blocks = (n_max + 255) / 256;
    const unsigned ns = (unsigned)(blocks < cus * 16 ? blocks : cus * 16);                  /* scatter: grid-stride over the survivors counted by then */
    if (depth == 0) {
        if (d_env_map)
            hipLaunchKernelGGL((k_pt_decide<true, true>), dim3(nb), dim3(256), 0, s, ps, d_prim_mesh, (const DevMaterial *)d_materials, d_counts, d_dir, d_prim,
                               d_path_of, d_thr, d_accum, d_path_of2);
        else
            hipLaunchKernelGGL((k_pt_decide<true, false>), dim3(nb), dim3(256), 0, s, ps, d_prim_mesh, (const DevMaterial *)d_materials, d_counts, d_dir, d_prim,
                               d_path_of, d_thr, d_accum, d_path_of2);
        hipLaunchKernelGGL((k_pt_scatter<true>), dim3(ns), dim3(256), 0, s, ps, *sc, d_nrm9, d_col9, d_prim_mesh, (const DevMaterial *)d_materials,
                           (const uint32_t *)d_counts, d_org, d_dir, d_prim, d_t, d_u, d_v, d_path_of, d_thr, d_org2, d_dir2, d_path_of2, d_thr2);
    } else {
        if (d_env_map)
            hipLaunchKernelGGL((k_pt_decide<false, true>), dim3(nb), dim3(256), 0, s, ps, d_prim_mesh, (const DevMaterial *)d_materials, d_counts, d_dir, d_prim,
                               d_path_of, d_thr, d_accum, d_path_of2);
        else
            hipLaunchKernelGGL((k_pt_decide<false, false>), dim3(nb), dim3(256), 0, s, ps, d_prim_mesh, (const DevMaterial *)d_materials, d_counts, d_dir, d_prim,
                               d_path_of, d_thr, d_accum, d_path_of2);
        hipLaunchKernelGGL((k_pt_scatter<false>), dim3(ns), dim3(256), 0, s, ps, *sc, d_nrm9, d_col9, d_prim_mesh, (const DevMaterial *)d_materials,
                           (const uint32_t *)d_counts, d_org, d_dir, d_prim, d_t, d_u, d_v, d_path_of, d_thr, d_org2, d_dir2, d_path_of2, d_thr2);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int lh_pt_launch_resolve(int w, int h, int band_rows, float inv_total_spp, unsigned long long *d_accum, float *d_rgb, void *stream)
{
    const size_t total = (size_t)w * h;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_pt_resolve, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w, h, band_rows, inv_total_spp, d_accum, d_rgb);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int lh_render_launch_state_build(size_t n, const lh_dev_scene_t *sc, const double *d_nrm9, const double *d_col9,
                                            const double *d_tan9, const double *d_bin9, const double *d_st6, const uint8_t *d_inside,
                                            const double *d_org, const double *d_dir, const uint32_t *d_prim, const double *d_t,
                                            const double *d_u, const double *d_v, double *d_state, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_state_build, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, *sc, d_nrm9, d_col9,
                       d_tan9, d_bin9, d_st6, d_inside, d_org, d_dir, d_prim, d_t, d_u, d_v, d_state);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
