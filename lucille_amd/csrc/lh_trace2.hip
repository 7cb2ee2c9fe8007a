/*
 * lh_trace2.hip -- the lean traversal kernel + the fp64 resolve pass (variant 6: an opt-in A/B walk; measured 4 % slower
 * than the default walk of lh_kernels.hip, whose AO source it also lost to -- DESIGN.md 3.1).
 *
 * Same algorithm and the same arithmetic as lh_kernels.hip's speculative 4-wide walk (fp32 conservative
 * filter over 64-byte 16-bit-grid nodes and 48-byte triangle records, parked leaves, ballot-compacted
 * refill from wave-private ray ranges), re-cut for occupancy -- the r01 kernel sat at 3 waves per SIMD
 * (128 VGPRs because the fp64 resolve was inlined, 44 KB of LDS stack per workgroup) with 66 % of its
 * wave-cycles in s_waitcnt (profiles/README.md r01d):
 *
 *   * the walk keeps NO fp64 state.  A finished ray leaves its <= 4 unresolved candidates (primitive ids)
 *     in its own output slot -- the 8-byte t and u cells hold two ids each until the resolve pass
 *     overwrites them -- and k_resolve2 re-tests them in fp64 with the reference's operation order
 *     (lh_exact_isect == triangle_isect, bvh.c:730-791), applies the tie rule and the fragile-hit test,
 *     and writes (prim, t, u, v).  Any-hit rays that end on a certain fp32 hit never reach fp64; the
 *     rare uncertain ones go through a small device queue (k_resolve_queue);
 *   * the per-lane stack is a 16-row ring in LDS (16 KB per 256-thread workgroup instead of 44): 98.8 % of
 *     the rays of S-soup-1M never hold more than 12 entries; a lane that needs more spills the oldest 8
 *     rows to its own 64-entry strip in global memory and reloads them when it pops back down;
 *   * rays come from a SOURCE: fp64 org/dir arrays (ri_raytrace batches, ray dumps), or the ambient-
 *     occlusion producer itself (lh_ao.h): the AO ray of (hit slot, sample) is generated in the refill,
 *     its occlusion is counted per slot with one atomic -- no 48-byte ray and 1-byte result per AO ray
 *     through HBM (22 GB per 4096^2 x 64 frame in round 1);
 *   * a ray whose pending list overflows (a fifth unresolved candidate) is handed to the reference's
 *     own walk (k_ref_retrace), which is exact by construction.
 *
 * Reference replaced: ri_bvh_intersect / bvh_traverse / test_ray_node / bvh_intersect_leaf_node /
 * triangle_isect (src/render/bvh.c:430-542, 1092-1188, 938-1083, 793-864, 730-791); AO producer
 * calculate_occlusion (src/transport/ambientocclusion.c:42-151).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "lh_device.h"
#include "lh_filter.h"
#include "lh_reftrace.h"
#include "lh_ao.h"

namespace {

#include "lh_walk.h"

constexpr int kRows   = LH_T2_ROWS;        /* LDS ring rows per lane (power of two) */
constexpr int kSpill  = 8;                 /* rows moved per spill / reload          */
constexpr int kNoLeaf = 0;                 /* never a valid leaf reference            */

#define RING(x) ((x) & (kRows - 1))

/* ray sources */
enum { SRC_ARRAYS = 0, SRC_AO = 1 };

struct Walk {               /* the lean per-lane state: fp32 ray + cursor + candidates */
    lh_ray32_t r;
    uint32_t sh[3];
    float tb;
    int   cur, sp, floor_;  /* floor_: logical index of the oldest LDS-resident stack entry */
    int   pend;
    uint32_t p0, p1, p2, p3;
    int   np;               /* 0..4 candidates; 5: overflow -> the reference walk */
    bool  certain;
};

__device__ __forceinline__ void walk_init(Walk &W, const lh_dev_scene_t &sc, double ox, double oy, double oz,
                                          double dx, double dy, double dz)
{
    lh_ray_setup(&W.r, ox, oy, oz, dx, dy, dz, sc.scene_r);
    lh_ray_setup_grid(&W.r, sc.grid_lo, sc.grid_step, sc.scene_r);
    W.sh[0] = W.r.ngx ? 16u : 0u; W.sh[1] = W.r.ngy ? 16u : 0u; W.sh[2] = W.r.ngz ? 16u : 0u;
    W.tb = 1.0e38f;
    W.cur = 0; W.sp = 1; W.floor_ = 0; W.pend = kNoLeaf;
    W.p0 = W.p1 = W.p2 = W.p3 = LH_MISS_PRIM; W.np = 0;
    W.certain = false;
}

__device__ __forceinline__ bool slab2(const Walk &W, uint32_t wx, uint32_t wy, uint32_t wz, float &tn_out)
{
    const uint32_t sx = __builtin_amdgcn_alignbit(wx, wx, W.sh[0]);
    const uint32_t sy = __builtin_amdgcn_alignbit(wy, wy, W.sh[1]);
    const uint32_t sz = __builtin_amdgcn_alignbit(wz, wz, W.sh[2]);
    const float tn = fmaxf(fmaxf(fmaf((float)(sx & 0xffffu), W.r.qax, W.r.qbnx), fmaf((float)(sy & 0xffffu), W.r.qay, W.r.qbny)),
                           fmaxf(fmaf((float)(sz & 0xffffu), W.r.qaz, W.r.qbnz), 0.0f));
    const float tf = fminf(fminf(fmaf((float)(sx >> 16), W.r.qax, W.r.qbfx), fmaf((float)(sy >> 16), W.r.qay, W.r.qbfy)),
                           fminf(fmaf((float)(sz >> 16), W.r.qaz, W.r.qbfz), W.tb));
    tn_out = tn;
    return tn <= tf;
}

struct T2Args {
    size_t n;
    const double *org, *dir;                  /* SRC_ARRAYS */
    uint32_t *prim; double *t, *u;            /* closest: output slots (candidates parked in t, u) */
    uint8_t *occ;                             /* any-hit, SRC_ARRAYS */
    unsigned long long *counters, *cursor;
    int min_active, tri_batch;
    int *spill;                               /* [grid lanes][64] */
    uint32_t *queue; uint32_t *qcount; uint32_t qcap;     /* pending any-hit rays: 6 words each */
    /* SRC_AO */
    const double *hitrec; const unsigned long long *slot_key; unsigned int *occ_count;
    unsigned long long seed; int ntheta, nphi;
};

/* the ray of work item i */
template <int SRC>
__device__ __forceinline__ void fetch_ray(const T2Args &a, size_t i, double &ox, double &oy, double &oz,
                                          double &dx, double &dy, double &dz)
{
    if (SRC == SRC_ARRAYS) {
        ox = a.org[3 * i]; oy = a.org[3 * i + 1]; oz = a.org[3 * i + 2];
        dx = a.dir[3 * i]; dy = a.dir[3 * i + 1]; dz = a.dir[3 * i + 2];
    } else {
        const uint32_t N = (uint32_t)(a.ntheta * a.nphi);
        const uint32_t slot = (uint32_t)(i / N), r = (uint32_t)(i % N);
        lh_ao_ray_builtin(a.hitrec + LH_HITREC_DOUBLES * (size_t)slot, a.slot_key[slot], a.seed, a.ntheta, a.nphi, (int)r,
                          ox, oy, oz, dx, dy, dz);
    }
}

/* one pending any-hit ray into the device queue; false when it is full */
__device__ __forceinline__ bool queue_push(const T2Args &a, size_t i, const Walk &W)
{
    const uint32_t k = atomicAdd(a.qcount, 1u);
    if (k >= a.qcap) return false;
    uint32_t *q = a.queue + 6 * (size_t)k;
    q[0] = (uint32_t)i; q[1] = (uint32_t)W.np | ((uint32_t)(i >> 32) << 8);
    q[2] = W.p0; q[3] = W.p1; q[4] = W.p2; q[5] = W.p3;
    return true;
}

/* a finished ray leaves the walk */
template <bool ANYHIT, int SRC>
__device__ __forceinline__ void retire(const T2Args &a, size_t i, const Walk &W)
{
    if (!ANYHIT) {
        if (W.np > kPend) { a.prim[i] = LH_PRIM_RETRACE; return; }
        if (W.np == 0) { a.prim[i] = LH_MISS_PRIM; a.t[i] = LH_T_INF; a.u[i] = 0.0; return; }   /* v is written by the resolve pass */
        a.prim[i] = LH_PRIM_PENDING | (uint32_t)W.np;
        ((uint2 *)a.t)[i] = make_uint2(W.p0, W.p1);
        ((uint2 *)a.u)[i] = make_uint2(W.p2, W.p3);
        return;
    }
    if (SRC == SRC_ARRAYS) {
        if (W.certain) { a.occ[i] = 1; return; }
        if (W.np == 0) { a.occ[i] = 0; return; }
        if (W.np > kPend || !queue_push(a, i, W)) { a.occ[i] = (uint8_t)LH_OCC_RETRACE; return; }
        a.occ[i] = (uint8_t)LH_OCC_PENDING;
    } else {
        if (W.certain) { atomicAdd(&a.occ_count[i / (uint32_t)(a.ntheta * a.nphi)], 1u); return; }
        if (W.np == 0) return;
        if (!queue_push(a, i, W)) atomicOr(a.qcount + 1, 1u);          /* queue full: the host re-renders the tile on the unfused path */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* the walk                                                                                    */
/* ------------------------------------------------------------------------------------------ */
template <bool ANYHIT, bool COUNT, int SRC>
__global__ __launch_bounds__(LH_BLOCK) void k_trace2(const lh_dev_scene_t sc, const T2Args a)
{
    __shared__ int stk[kRows][LH_BLOCK];
    const int tid = threadIdx.x;
    int *const my_spill = a.spill + 64 * ((size_t)blockIdx.x * LH_BLOCK + tid);
    const float4 *__restrict__ tris = (const float4 *)sc.tri32;
    uint32_t cn = 0, ct = 0, cr = 0;
    Walk W;
    W.cur = kDone; W.sp = 1; W.floor_ = 0; W.pend = kNoLeaf; W.np = 0; W.certain = false;
    size_t my = (size_t)-1;
    bool exhausted = false;
    unsigned long long wbase = 0, wend = 0;
    const size_t n = a.n;

    for (;;) {
        /* ---- regroup: retire finished lanes, refill them ---------------------------------- */
        const bool idle = (W.cur == kDone) && (W.pend == kNoLeaf);
        const unsigned long long idle_mask = __ballot(idle);
        if (idle && my != (size_t)-1) {
            retire<ANYHIT, SRC>(a, my, W);
            if (COUNT) cr++;
            my = (size_t)-1;
        }
        if (idle_mask != 0ull && !exhausted) {
            if (wbase == wend) {
                unsigned long long b = 0;
                if ((tid & 63) == 0) b = atomicAdd(a.cursor, (unsigned long long)sc.ray_chunk);
                b = __shfl(b, 0);
                wbase = b < n ? b : n;
                wend = (b + sc.ray_chunk < n) ? b + sc.ray_chunk : n;
            }
            const int need = __popcll(idle_mask);
            const unsigned long long avail = wend - wbase;
            const int take = avail < (unsigned long long)need ? (int)avail : need;
            const int rank = __popcll(idle_mask & ((1ull << (tid & 63)) - 1ull));
            if (idle && rank < take) {
                double ox, oy, oz, dx, dy, dz;
                my = wbase + rank;
                fetch_ray<SRC>(a, my, ox, oy, oz, dx, dy, dz);
                walk_init(W, sc, ox, oy, oz, dx, dy, dz);
                stk[0][tid] = kDone;
            }
            wbase += take;
            if (wbase >= n) exhausted = true;
        }
        if (__ballot((W.cur != kDone) | (W.pend != kNoLeaf)) == 0ull) break;
        const int thresh = exhausted ? 1 : a.min_active;

        /* ---- walk until too few lanes remain active --------------------------------------- */
        for (;;) {
            /* ring maintenance (rare): an iteration pops at most 3 entries (node step 2, triangle pass 1) and
             * writes at most 4 rows above the top: keep 3 entries resident below it, 4 rows free above it */
            if (W.floor_ > 0 && W.sp - W.floor_ < 3) {
                W.floor_ -= kSpill;
                for (int p = 0; p < kSpill; p++) stk[RING(W.floor_ + p)][tid] = my_spill[W.floor_ + p];
            }
            if (W.cur >= 0) {
                if (W.sp + 4 - W.floor_ > kRows) {
                    for (int p = 0; p < kSpill; p++) my_spill[W.floor_ + p] = stk[RING(W.floor_ + p)][tid];
                    W.floor_ += kSpill;
                }
                const uint4 *p = (const uint4 *)sc.q4nodes + 4 * (size_t)W.cur;
                const uint4 na = p[0], nb = p[1], nc = p[2], nr = p[3];
                if (COUNT) cn++;
                float t0, t1, t2, t3;
                const bool h0 = slab2(W, na.x, na.y, na.z, t0) & ((int)nr.x != kDone);
                const bool h1 = slab2(W, na.w, nb.x, nb.y, t1) & ((int)nr.y != kDone);
                const bool h2 = slab2(W, nb.z, nb.w, nc.x, t2) & ((int)nr.z != kDone);
                const bool h3 = slab2(W, nc.y, nc.z, nc.w, t3) & ((int)nr.w != kDone);
                const uint32_t k0 = h0 ? ((__float_as_uint(t0) & ~3u) | 0u) : 0xFFFFFFFCu;
                const uint32_t k1 = h1 ? ((__float_as_uint(t1) & ~3u) | 1u) : 0xFFFFFFFDu;
                const uint32_t k2 = h2 ? ((__float_as_uint(t2) & ~3u) | 2u) : 0xFFFFFFFEu;
                const uint32_t k3 = h3 ? ((__float_as_uint(t3) & ~3u) | 3u) : 0xFFFFFFFFu;
                const int b10 = k1 < k0, b20 = k2 < k0, b30 = k3 < k0, b21 = k2 < k1, b31 = k3 < k1, b32 = k3 < k2;
                const int rk0 = b10 + b20 + b30, rk1 = (1 - b10) + b21 + b31;
                const int rk2 = (2 - b20 - b21) + b32, rk3 = 3 - b30 - b31 - b32;
                const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
                const int base = W.sp + nh - 1;
                stk[RING(h0 ? base - rk0 : W.sp + rk0)][tid] = (int)nr.x;
                stk[RING(h1 ? base - rk1 : W.sp + rk1)][tid] = (int)nr.y;
                stk[RING(h2 ? base - rk2 : W.sp + rk2)][tid] = (int)nr.z;
                stk[RING(h3 ? base - rk3 : W.sp + rk3)][tid] = (int)nr.w;
                W.sp = base;
                const int nxt = stk[RING(base)][tid];
                const int popped2 = stk[RING(W.sp - 1)][tid];
                const bool is_leaf = (nxt < 0) & (nxt != kDone);
                const bool park = is_leaf & (W.pend == kNoLeaf);
                W.pend = park ? nxt : W.pend;
                W.cur = park ? popped2 : nxt;
                W.sp -= park ? 1 : 0;
            }
            const unsigned long long m_node = __ballot(W.cur >= 0);
            const unsigned long long m_pend = __ballot(W.pend != kNoLeaf);
            if (m_pend != 0ull && (__popcll(m_pend) >= a.tri_batch || m_node == 0ull)) {
                if (W.pend != kNoLeaf) {
                    const uint32_t x = ~(uint32_t)W.pend;
                    const float4 *tp = tris + 3 * (size_t)(x >> 2);
                    const float4 ta = tp[0], tb_ = tp[1], tc = tp[2];
                    if (COUNT) ct++;
                    float t_hi;
                    bool finished = false;
                    const int cls = lh_tri_filter(&W.r, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w, W.tb, &t_hi);
                    if (cls != LH_TRI_REJECT) {
                        const bool sure = (cls == LH_TRI_CERTAIN);
                        if (ANYHIT && sure) { W.certain = true; finished = true; }
                        else {
                            if (sure) W.tb = fminf(W.tb, t_hi);
                            if (W.np >= kPend) { W.np = kPend + 1; finished = true; }      /* a fifth candidate: the reference walk decides */
                            else { W.p3 = W.p2; W.p2 = W.p1; W.p1 = W.p0; W.p0 = __float_as_uint(tc.y); W.np++; }
                        }
                    }
                    if (finished) { W.cur = kDone; W.pend = kNoLeaf; W.floor_ = 0; }
                    else if (x & 3u) W.pend = (int)~(((x >> 2) + 1u) << 2 | ((x & 3u) - 1u));
                    else {
                        const bool waiting = (W.cur < 0) & (W.cur != kDone);
                        W.pend = waiting ? W.cur : kNoLeaf;
                        if (waiting) { W.sp--; W.cur = stk[RING(W.sp)][tid]; }
                    }
                }
            }
            const unsigned long long m_work = __ballot((W.cur != kDone) | (W.pend != kNoLeaf));
            if (__popcll(m_work) < thresh) break;
        }
    }
    if (COUNT) {
        atomicAdd(&a.counters[LH_CNT_NODES], (unsigned long long)cn);
        atomicAdd(&a.counters[LH_CNT_TRIS], (unsigned long long)ct);
        atomicAdd(&a.counters[LH_CNT_RAYS], (unsigned long long)cr);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* fp64 resolve of the parked candidates: closest-hit, one thread per ray                      */
/* ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_resolve2(const lh_dev_scene_t sc, size_t n, const double *__restrict__ org,
                                                  const double *__restrict__ dir, uint32_t *__restrict__ prim,
                                                  double *__restrict__ t, double *__restrict__ u, double *__restrict__ v,
                                                  unsigned long long *counters)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = prim[i];
    if (h == LH_MISS_PRIM) { v[i] = 0.0; return; }
    if ((h & 0xFFFFFFF8u) != LH_PRIM_PENDING) return;          /* LH_PRIM_RETRACE: k_ref_retrace's */
    const int np = (int)(h & 7u);
    const uint2 c01 = ((const uint2 *)t)[i], c23 = ((const uint2 *)u)[i];
    const double ox = org[3 * i], oy = org[3 * i + 1], oz = org[3 * i + 2];
    const double dx = dir[3 * i], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
    Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
    resolve(sc, c01.x, ox, oy, oz, dx, dy, dz, best);
    if (np > 1) resolve(sc, c01.y, ox, oy, oz, dx, dy, dz, best);
    if (np > 2) resolve(sc, c23.x, ox, oy, oz, dx, dy, dz, best);
    if (np > 3) resolve(sc, c23.y, ox, oy, oz, dx, dy, dz, best);
    /* a hit the reference may not reach goes through the reference's own walk (k_ref_retrace) */
    const bool retrace = (sc.ref_nodes != NULL) && best.prim != LH_MISS_PRIM && best.frag != 0u;
    prim[i] = retrace ? LH_PRIM_RETRACE : best.prim; t[i] = best.t; u[i] = best.u; v[i] = best.v;
    if (counters) atomicAdd(&counters[LH_CNT_EXACT], (unsigned long long)np);
}

/* pending any-hit rays (rare: a certain fp32 hit ends an any-hit ray at once) */
template <int SRC>
__global__ __launch_bounds__(256) void k_resolve_queue(const lh_dev_scene_t sc, const T2Args a)
{
    const uint32_t total = *a.qcount < a.qcap ? *a.qcount : a.qcap;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const uint32_t *q = a.queue + 6 * e;
        const size_t i = (size_t)q[0] | ((size_t)(q[1] >> 8) << 32);
        const int np = (int)(q[1] & 0xffu);
        double ox, oy, oz, dx, dy, dz;
        fetch_ray<SRC>(a, i, ox, oy, oz, dx, dy, dz);
        int hit = -1;                                   /* -1: the reference walk decides */
        if (np <= kPend) {
            Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
            for (int k = 0; k < np && best.prim == LH_MISS_PRIM; k++) resolve(sc, q[2 + k], ox, oy, oz, dx, dy, dz, best);
            if (best.prim == LH_MISS_PRIM) hit = 0;
            else if (best.frag == 0u || sc.ref_nodes == NULL) hit = 1;
            if (a.counters) atomicAdd(&a.counters[LH_CNT_EXACT], (unsigned long long)np);
        }
        if (SRC == SRC_ARRAYS) {
            a.occ[i] = hit < 0 ? (uint8_t)LH_OCC_RETRACE : (uint8_t)hit;
        } else {
            if (hit < 0) {
                uint32_t p; double tt, uu, vv;
                hit = lh_ref_trace((const lh_refnode_t *)sc.ref_nodes, (const uint32_t *)sc.ref_leaf_prims, (const double *)sc.tri64,
                                   sc.ref_empty, sc.ref_bmin, sc.ref_bmax, ox, oy, oz, dx, dy, dz, &p, &tt, &uu, &vv);
                if (a.counters) atomicAdd(&a.counters[LH_CNT_RETRACED], 1ull);
            }
            if (hit) atomicAdd(&a.occ_count[i / (uint32_t)(a.ntheta * a.nphi)], 1u);
        }
    }
}

/* SRC_AO rays whose pending list overflowed are queued with np = 5 and handled above; with ray arrays
 * the marker goes to the output slot and this scan (shared with lh_kernels.hip's) re-traces it */
__global__ __launch_bounds__(256) void k_ref_retrace2(lh_dev_scene_t sc, size_t n, const double *__restrict__ org,
                                                      const double *__restrict__ dir, uint32_t *__restrict__ prim,
                                                      double *__restrict__ t, double *__restrict__ u, double *__restrict__ v,
                                                      uint8_t *__restrict__ occ, int anyhit, unsigned long long *counters)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (anyhit ? (occ[i] != LH_OCC_RETRACE) : (prim[i] != LH_PRIM_RETRACE)) return;
    uint32_t p; double tt, uu, vv;
    const int hit = lh_ref_trace((const lh_refnode_t *)sc.ref_nodes, (const uint32_t *)sc.ref_leaf_prims, (const double *)sc.tri64,
                                 sc.ref_empty, sc.ref_bmin, sc.ref_bmax, org[3 * i], org[3 * i + 1], org[3 * i + 2],
                                 dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], &p, &tt, &uu, &vv);
    if (anyhit) occ[i] = hit ? 1 : 0;
    else { prim[i] = p; t[i] = tt; u[i] = uu; v[i] = vv; }
    if (counters) atomicAdd(&counters[LH_CNT_RETRACED], 1ull);
}

template <bool ANYHIT, int SRC>
void launch_t2(const lh_dev_scene_t &sc, const T2Args &a, bool count, int grid, hipStream_t s)
{
    if (count) hipLaunchKernelGGL((k_trace2<ANYHIT, true, SRC>), dim3(grid), dim3(LH_BLOCK), 0, s, sc, a);
    else       hipLaunchKernelGGL((k_trace2<ANYHIT, false, SRC>), dim3(grid), dim3(LH_BLOCK), 0, s, sc, a);
}

} /* namespace */

/* workgroups per CU the lean kernel can keep resident (LDS ring 16 KB, registers): the persistent grid */
extern "C" int lh_trace2_blocks_per_cu(void)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_trace2<false, false, SRC_ARRAYS>, LH_BLOCK, 0) != hipSuccess || nb < 1) nb = 4;
    int nb2 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, (const void *)k_trace2<true, false, SRC_AO>, LH_BLOCK, 0) == hipSuccess && nb2 >= 1 && nb2 < nb) nb = nb2;
    return nb;
}

/* ray arrays -> hit records / occlusion bytes.  The caller provides the work cursor, the spill strips
 * (grid_blocks * LH_BLOCK * 64 ints) and, for any-hit, the pending queue. */
extern "C" int lh_launch_trace2(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dir,
                                uint32_t *d_prim, double *d_t, double *d_u, double *d_v, int anyhit, uint8_t *d_occluded,
                                unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks,
                                int min_active, int tri_batch, int *d_spill, uint32_t *d_queue, uint32_t *d_qcount,
                                uint32_t qcap, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    lh_dev_scene_t scl = *sc;
    {
        const size_t waves = (size_t)(grid_blocks > 0 ? grid_blocks : 1) * (LH_BLOCK / 64);
        size_t c = n / (waves * 4);
        if (c < 64) c = 64;
        if (c < scl.ray_chunk) scl.ray_chunk = (uint32_t)c;
        if (scl.ray_chunk == 0) scl.ray_chunk = 64;
    }
    T2Args a;
    memset(&a, 0, sizeof(a));
    a.n = n; a.org = d_org; a.dir = d_dir; a.prim = d_prim; a.t = d_t; a.u = d_u; a.occ = d_occluded;
    a.counters = d_counters; a.cursor = d_cursor; a.min_active = min_active; a.tri_batch = tri_batch;
    a.spill = d_spill; a.queue = d_queue; a.qcount = d_qcount; a.qcap = qcap;
    if (hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), s) != hipSuccess) return -1;
    if (anyhit) {
        if (hipMemsetAsync(d_qcount, 0, 2 * sizeof(uint32_t), s) != hipSuccess) return -1;
        launch_t2<true, SRC_ARRAYS>(scl, a, d_counters != NULL, grid_blocks, s);
        hipLaunchKernelGGL((k_resolve_queue<SRC_ARRAYS>), dim3(256), dim3(256), 0, s, scl, a);
    } else {
        launch_t2<false, SRC_ARRAYS>(scl, a, d_counters != NULL, grid_blocks, s);
        hipLaunchKernelGGL(k_resolve2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scl, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_counters);
    }
    if (scl.ref_nodes)
        hipLaunchKernelGGL(k_ref_retrace2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scl, n, d_org, d_dir,
                           d_prim, d_t, d_u, d_v, d_occluded, anyhit, d_counters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* the queued AO rays of lh_launch_trace_ao (lh_kernels.hip): regenerated, decided by the reference walk,
 * added to their slot's count */
extern "C" int lh_launch_ao_queue(const lh_dev_scene_t *sc, int ntheta, int nphi, unsigned long long seed,
                                  const double *d_hitrec, const unsigned long long *d_slot_key, unsigned int *d_occ_count,
                                  unsigned long long *d_counters, uint32_t *d_queue, uint32_t *d_qcount, uint32_t qcap, void *stream)
{
    T2Args a;
    memset(&a, 0, sizeof(a));
    a.counters = d_counters; a.queue = d_queue; a.qcount = d_qcount; a.qcap = qcap;
    a.hitrec = d_hitrec; a.slot_key = d_slot_key; a.occ_count = d_occ_count; a.seed = seed; a.ntheta = ntheta; a.nphi = nphi;
    hipLaunchKernelGGL((k_resolve_queue<SRC_AO>), dim3(64), dim3(256), 0, (hipStream_t)stream, *sc, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* the AO stage of a tile with the rays generated inside the any-hit kernel: nslots primary hits (hit
 * records + absolute sample keys), N = ntheta * nphi rays each, occlusion counted per slot in
 * d_occ_count (zeroed here).  d_qcount[1] != 0 afterwards: the pending queue overflowed and the caller
 * must redo the stage on the materialised path. */
extern "C" int lh_launch_trace2_ao(const lh_dev_scene_t *sc, size_t nslots, int ntheta, int nphi, unsigned long long seed,
                                   const double *d_hitrec, const unsigned long long *d_slot_key, unsigned int *d_occ_count,
                                   unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks,
                                   int min_active, int tri_batch, int *d_spill, uint32_t *d_queue, uint32_t *d_qcount,
                                   uint32_t qcap, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const size_t n = nslots * (size_t)(ntheta * nphi);
    if (n == 0) return 0;
    if (n >= ((size_t)1 << 32)) return -1;                      /* the caller cuts the tile */
    lh_dev_scene_t scl = *sc;
    {
        const size_t waves = (size_t)(grid_blocks > 0 ? grid_blocks : 1) * (LH_BLOCK / 64);
        size_t c = n / (waves * 4);
        if (c < 64) c = 64;
        if (c < scl.ray_chunk) scl.ray_chunk = (uint32_t)c;
        if (scl.ray_chunk == 0) scl.ray_chunk = 64;
    }
    T2Args a;
    memset(&a, 0, sizeof(a));
    a.n = n; a.counters = d_counters; a.cursor = d_cursor; a.min_active = min_active; a.tri_batch = tri_batch;
    a.spill = d_spill; a.queue = d_queue; a.qcount = d_qcount; a.qcap = qcap;
    a.hitrec = d_hitrec; a.slot_key = d_slot_key; a.occ_count = d_occ_count; a.seed = seed; a.ntheta = ntheta; a.nphi = nphi;
    if (hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_qcount, 0, 2 * sizeof(uint32_t), s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_occ_count, 0, sizeof(unsigned int) * nslots, s) != hipSuccess) return -1;
    launch_t2<true, SRC_AO>(scl, a, d_counters != NULL, grid_blocks, s);
    hipLaunchKernelGGL((k_resolve_queue<SRC_AO>), dim3(256), dim3(256), 0, s, scl, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
