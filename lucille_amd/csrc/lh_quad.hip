/*
 * lh_quad.hip -- the quad-per-ray walk (variant 7, opt-in).
 *
 * One ray per QUAD of lanes over the same 4-wide 16-bit-grid tree: lane k of the quad loads and tests child k
 * (one 16-byte piece of the node: four lanes -> one 64-byte contiguous request), the four entry distances are
 * ranked with DPP quad permutes, every lane writes its child reference to the ray's LDS stack at its rank-derived
 * slot, and the next reference is read back from the new top -- the same branch-free step as traverse_spec4
 * (lh_kernels.hip), spread over four lanes.  A parked leaf (<= 4 triangles) is tested in ONE pass, lane k taking
 * triangle k.  Each lane keeps the unresolved candidates of its own triangles and resolves them in fp64 when the
 * ray ends; the four exact bests are merged over the quad with the reference's tie rule (lh_walk.h).
 *
 * Why it was tried: tools/ubench/gather.hip -- in round 1's run dependent random 64-byte gathers with one chain per quad
 * ran 1.45x the per-lane mode.  What it costs: a wave carries 16 rays instead of 64 -- 1.9x the VALU wave-instructions, and
 * 256 rays in flight per CU instead of 768.  Measured: half the lane walk's rate (profiles/README.md, r02 experiments); and
 * the microbenchmark's promise was an artefact of unperturbed chain links (corrected: 1.12x at the 4 waves per SIMD this
 * kernel's registers allow).  Kept as a tested opt-in.
 *
 * Node layout: lh_q4node_t re-cut child-major ("q4t"): piece k = { x: lo|hi<<16, y, z, child ref } -- made on the
 * device from the q4 nodes the first time the variant is used (lh_quad_make_nodes).
 *
 * Reference replaced: bvh_traverse / test_ray_node / bvh_intersect_leaf_node / triangle_isect
 * (src/render/bvh.c:1092-1188, 938-1083, 793-864, 730-791).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lh_device.h"
#include "lh_filter.h"
#include "lh_reftrace.h"

namespace {

#include "lh_walk.h"

constexpr int kNoLeaf = 0;
constexpr int kRaysPerBlock = LH_BLOCK / 4;

/* quad permutes (DPP): value of lane (quad base + P[sub]) */
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int CTRL> __device__ __forceinline__ uint32_t qperm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ float qpermf(float v) { return __uint_as_float(qperm<CTRL>(__float_as_uint(v))); }
template <int CTRL> __device__ __forceinline__ double qpermd(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = qperm<CTRL>((uint32_t)b), hi = qperm<CTRL>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int kRot1 = QP(1, 2, 3, 0), kRot2 = QP(2, 3, 0, 1), kRot3 = QP(3, 0, 1, 2), kSwap1 = QP(1, 0, 3, 2);

__device__ __forceinline__ bool quad_any(bool b)
{
    uint32_t x = b ? 1u : 0u;
    x |= qperm<kSwap1>(x);
    x |= qperm<kRot2>(x);
    return x != 0u;
}

/* the two exact bests of a lane pair -> the ray's: the same decisions as resolve() (lh_walk.h) takes when the higher
 * lane's hit arrives after the lower lane's.  Both lanes of the pair compute the same record. */
__device__ __forceinline__ void merge_pair(const lh_dev_scene_t &sc, Best &mine, const Best &theirs, const bool i_am_hi,
                                           double dx, double dy, double dz)
{
    const Best lo = i_am_hi ? theirs : mine, hi = i_am_hi ? mine : theirs;
    Best out = lo;
    if (lo.prim == LH_MISS_PRIM) out = hi;
    else if (hi.prim != LH_MISS_PRIM) {
        bool take = hi.t < lo.t;
        if (!take && hi.t == lo.t && hi.prim != lo.prim) take = tie_takes_new(sc, hi.prim, lo.prim, dx, dy, dz);
        if (take) out = hi;
        uint32_t sticky = (lo.frag | hi.frag) & 2u;
        if (hi.prim != lo.prim && hi.t != lo.t && fabs(hi.t - lo.t) <= LH_FRAGILE_REL * fmax(fabs(hi.t), fabs(lo.t))) sticky = 2u;
        out.frag = (out.frag & 1u) | sticky;
    }
    mine = out;
}

template <int CTRL> __device__ __forceinline__ Best best_from(const Best &b)
{
    Best o;
    o.t = qpermd<CTRL>(b.t); o.u = qpermd<CTRL>(b.u); o.v = qpermd<CTRL>(b.v);
    o.prim = qperm<CTRL>(b.prim); o.frag = qperm<CTRL>(b.frag);
    return o;
}

/* one walk segment: until fewer than min_rays rays of the wave still have work */
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ void traverse_quad(Lane &L, int &pend, const lh_dev_scene_t &sc, int *__restrict__ col,
                                              const int sub, const double *__restrict__ ro, const double *__restrict__ rd,
                                              Best &best, uint32_t &c_nodes, uint32_t &c_tris, uint32_t &c_exact,
                                              const int min_rays, const int tri_rays)
{
    const uint4 *__restrict__ nodes = (const uint4 *)sc.q4tnodes;
    const float4 *__restrict__ tris = (const float4 *)sc.tri32;
    const int rows = (int)sc.stack_rows;

    for (;;) {
        if (sc.stack_guard && L.cur >= 0 && L.sp + 4 > rows) { L.over = true; L.cur = kDone; pend = kNoLeaf; }
        if (L.cur >= 0) {                                   /* uniform over the quad */
            const uint4 w = nodes[4 * (size_t)L.cur + sub];
            if (COUNT) c_nodes += (sub == 0);
            float tn;
            const bool h = slab_w(L, w.x, w.y, w.z, tn) & ((int)w.w != kDone);
            const uint32_t key = h ? ((__float_as_uint(tn) & ~3u) | (uint32_t)sub) : (0xFFFFFFFCu | (uint32_t)sub);
            const uint32_t k1 = qperm<kRot1>(key), k2 = qperm<kRot2>(key), k3 = qperm<kRot3>(key);
            const int rank = (int)(k1 < key) + (int)(k2 < key) + (int)(k3 < key);
            uint32_t nh = h ? 1u : 0u;
            nh += qperm<kSwap1>(nh);
            nh += qperm<kRot2>(nh);
            const int base = L.sp + (int)nh - 1;
            col[h ? base - rank : L.sp + rank] = (int)w.w;  /* hits: farthest at the bottom; misses: above the new top */
            L.sp = base;
            const int nxt = col[base];
            const int popped2 = col[base - 1];
            const bool is_leaf = (nxt < 0) & (nxt != kDone);
            const bool park = is_leaf & (pend == kNoLeaf);
            pend = park ? nxt : pend;
            L.cur = park ? popped2 : nxt;
            L.sp -= park ? 1 : 0;
        }
        const unsigned long long m_node = __ballot(L.cur >= 0);
        const unsigned long long m_pend = __ballot(pend != kNoLeaf);
        if (m_pend != 0ull && (__popcll(m_pend) >= 4 * tri_rays || m_node == 0ull)) {
            if (pend != kNoLeaf) {                          /* uniform over the quad */
                const uint32_t x = ~(uint32_t)pend;
                bool fin = false;
                if ((uint32_t)sub <= (x & 3u)) {            /* lane k: triangle k of the leaf */
                    const float4 *tp = tris + 3 * (size_t)((x >> 2) + (uint32_t)sub);
                    const float4 ta = tp[0], tb_ = tp[1], tc = tp[2];
                    if (COUNT) c_tris++;
                    fin = tri_step_g<ANYHIT, COUNT>(L, sc, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w, __float_as_uint(tc.y),
                                                    [=](double &a, double &b, double &c, double &d, double &e, double &f) {
                                                        a = ro[0]; b = ro[1]; c = ro[2]; d = rd[0]; e = rd[1]; f = rd[2]; },
                                                    best, c_exact);
                }
                /* the ray's culling bound: the tightest certain hit of the four lanes */
                float tb = fminf(L.tb, qpermf<kSwap1>(L.tb));
                tb = fminf(tb, qpermf<kRot2>(tb));
                L.tb = tb;
                if (ANYHIT) fin = quad_any(fin);
                if (fin) { L.cur = kDone; pend = kNoLeaf; }
                else {
                    const bool waiting = (L.cur < 0) & (L.cur != kDone);
                    pend = waiting ? L.cur : kNoLeaf;
                    if (waiting) { L.sp--; L.cur = col[L.sp]; }
                }
            }
        }
        const unsigned long long m_work = __ballot((L.cur != kDone) | (pend != kNoLeaf));
        if (__popcll(m_work) < 4 * min_rays) break;
    }
}

template <bool ANYHIT, bool COUNT>
__global__ __launch_bounds__(LH_BLOCK, LH_QUAD_WAVES_PER_SIMD) void k_trace_quad(
    lh_dev_scene_t sc, size_t n, const double *__restrict__ org, const double *__restrict__ dir,
    uint32_t *__restrict__ prim, double *__restrict__ t, double *__restrict__ u, double *__restrict__ v,
    uint8_t *__restrict__ occ, unsigned long long *counters, unsigned long long *cursor, int min_rays, int tri_rays,
    int over_fix)
{
    extern __shared__ int lh_quad_lds[];            /* [rays per block][stride]: one stack column per RAY */
    const int tid = threadIdx.x, sub = tid & 3;
    const int stride = (int)sc.stack_rows + 1;      /* odd: the rays' equal-depth slots fall into different banks */
    int *col = lh_quad_lds + (tid >> 2) * stride;
    uint32_t cn = 0, ct = 0, ce = 0, cr = 0;
    Lane L; Best best = {LH_T_INF, 0.0, 0.0, LH_MISS_PRIM, 0u};
    int pend = kNoLeaf;
    size_t my = (size_t)-1;
    L.cur = kDone; L.sp = 1; L.np = 0; L.certain = false; L.over = false;
    bool exhausted = false;
    unsigned long long wbase = 0, wend = 0;
    const unsigned long long quad_leaders = 0x1111111111111111ull;

    for (;;) {
        const bool idle = (L.cur == kDone) && (pend == kNoLeaf);       /* uniform over the quad */
        const unsigned long long idle_q = __ballot(idle) & quad_leaders;
        if (idle && my != (size_t)-1) {
            /* the fp64 ray is not kept in registers over the walk: read again for the resolve */
            const double ox = org[3 * my], oy = org[3 * my + 1], oz = org[3 * my + 2];
            const double dx = dir[3 * my], dy = dir[3 * my + 1], dz = dir[3 * my + 2];
            finish<ANYHIT, COUNT>(L, sc, ox, oy, oz, dx, dy, dz, best, ce);
            const bool over = L.over;
            if (ANYHIT) {
                /* occluded for sure: a certain fp32 hit, or an exact hit the reference cannot miss; a fragile hit goes
                 * through the reference's own walk */
                const bool hit = best.prim != LH_MISS_PRIM;
                const bool sure = quad_any(L.certain | (hit & (best.frag == 0u)));
                const bool some = quad_any(hit);
                if (sub == 0) {
                    if (over) occ[my] = (uint8_t)(over_fix ? LH_OCC_OVERFLOW : (sc.ref_nodes ? LH_OCC_RETRACE : 0));
                    else occ[my] = sure ? 1 : ((some & (sc.ref_nodes != NULL)) ? (uint8_t)LH_OCC_RETRACE : (some ? 1 : 0));
                }
            } else {
                merge_pair(sc, best, best_from<kSwap1>(best), (sub & 1) != 0, dx, dy, dz);
                merge_pair(sc, best, best_from<kRot2>(best), (sub & 2) != 0, dx, dy, dz);
                if (sub == 0) {
                    if (over && over_fix) prim[my] = LH_PRIM_OVERFLOW;
                    else {
                        const bool retrace = sc.ref_nodes != NULL && (over || (best.prim != LH_MISS_PRIM && best.frag != 0u));
                        prim[my] = retrace ? LH_PRIM_RETRACE : best.prim; t[my] = best.t; u[my] = best.u; v[my] = best.v;
                    }
                }
            }
            if (COUNT) cr += (sub == 0);
            my = (size_t)-1;
        }
        if (idle_q != 0ull && !exhausted) {
            if (wbase == wend) {
                unsigned long long b = 0;
                if ((tid & 63) == 0) b = atomicAdd(cursor, (unsigned long long)sc.ray_chunk);
                b = __shfl(b, 0);
                wbase = b < n ? b : n;
                wend = (b + sc.ray_chunk < n) ? b + sc.ray_chunk : n;
            }
            const int need = __popcll(idle_q);
            const unsigned long long avail = wend - wbase;
            const int take = avail < (unsigned long long)need ? (int)avail : need;
            const int lead = (tid & 63) & ~3;
            const int rank = __popcll(idle_q & ((1ull << lead) - 1ull));
            if (idle && rank < take) {
                const size_t i = wbase + rank;
                my = i;
                const double ox = org[3 * i], oy = org[3 * i + 1], oz = org[3 * i + 2];
                const double dx = dir[3 * i], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
                lane_init(L, sc, ox, oy, oz, dx, dy, dz);
                best.t = LH_T_INF; best.u = 0.0; best.v = 0.0; best.prim = LH_MISS_PRIM; best.frag = 0u;
                col[0] = kDone;
            }
            wbase += take;
            if (wbase >= n) exhausted = true;
        }
        const unsigned long long work = __ballot((L.cur != kDone) | (pend != kNoLeaf));
        if (work == 0ull) break;
        traverse_quad<ANYHIT, COUNT>(L, pend, sc, col, sub, org + 3 * my, dir + 3 * my, best, cn, ct, ce, exhausted ? 1 : min_rays, tri_rays);
    }
    if (COUNT) {
        atomicAdd(&counters[LH_CNT_NODES], (unsigned long long)cn);
        atomicAdd(&counters[LH_CNT_TRIS], (unsigned long long)ct);
        atomicAdd(&counters[LH_CNT_EXACT], (unsigned long long)ce);
        atomicAdd(&counters[LH_CNT_RAYS], (unsigned long long)cr);
    }
}

__global__ void k_q4_to_q4t(uint32_t n, const uint4 *__restrict__ q4, uint4 *__restrict__ q4t)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 a = q4[4 * (size_t)i], b = q4[4 * (size_t)i + 1], c = q4[4 * (size_t)i + 2], r = q4[4 * (size_t)i + 3];
    q4t[4 * (size_t)i]     = make_uint4(a.x, a.y, a.z, r.x);
    q4t[4 * (size_t)i + 1] = make_uint4(a.w, b.x, b.y, r.y);
    q4t[4 * (size_t)i + 2] = make_uint4(b.z, b.w, c.x, r.z);
    q4t[4 * (size_t)i + 3] = make_uint4(c.y, c.z, c.w, r.w);
}

} /* namespace */

extern "C" int lh_quad_make_nodes(uint32_t nq4, const void *d_q4nodes, void *d_q4tnodes, void *stream)
{
    if (nq4 == 0) return 0;
    hipLaunchKernelGGL(k_q4_to_q4t, dim3((nq4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, nq4, (const uint4 *)d_q4nodes, (uint4 *)d_q4tnodes);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* workgroups per CU the quad kernel can hold (LDS: one stack column per ray; registers: see the build log) */
extern "C" int lh_quad_blocks_per_cu(uint32_t stack_rows)
{
    const size_t lds = (size_t)(stack_rows + 1) * kRaysPerBlock * sizeof(int);
    int by_lds = (int)((160u * 1024u) / lds);
    return by_lds < LH_QUAD_WAVES_PER_SIMD ? by_lds : LH_QUAD_WAVES_PER_SIMD;
}

/* the main launch; *over_fix_out tells the caller that rays may carry the overflow marker (lh_launch_trace then runs
 * k_overflow_fix and k_ref_retrace over the batch exactly as for the lane walk) */
extern "C" int lh_launch_trace_quad(const lh_dev_scene_t *sc, size_t n, const double *d_org, const double *d_dir, uint32_t *d_prim,
                                    double *d_t, double *d_u, double *d_v, int anyhit, uint8_t *d_occ,
                                    unsigned long long *d_counters, unsigned long long *d_cursor, int grid_blocks, int min_active,
                                    int tri_batch, int *over_fix_out, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!sc->q4tnodes) return -1;
    lh_dev_scene_t scl = *sc;
    uint32_t need = 3 * sc->q4_depth + 5;
    const uint32_t cap = (sc->stack_cap >= 8 && sc->stack_cap < 64) ? sc->stack_cap : 64;
    int over_fix = 0;
    if (need > cap) { need = cap; over_fix = 1; scl.stack_guard = 1; }
    if (need < 16 && !over_fix) need = 16;
    scl.stack_rows = need;
    *over_fix_out = over_fix;
    const size_t lds_bytes = (size_t)(need + 1) * kRaysPerBlock * sizeof(int);
    {
        const size_t waves = (size_t)(grid_blocks > 0 ? grid_blocks : 1) * (LH_BLOCK / 64);
        size_t c = n / (waves * 4);
        if (c < 16) c = 16;
        uint32_t chunk = scl.ray_chunk / 4; if (chunk < 16) chunk = 16;      /* a wave carries 16 rays, not 64 */
        if (c < chunk) chunk = (uint32_t)c;
        scl.ray_chunk = chunk;
    }
    int min_rays = min_active / 4; if (min_rays < 1) min_rays = 1;
    int tri_rays = tri_batch / 4; if (tri_rays < 1) tri_rays = 1;
    if (hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), s) != hipSuccess) return -1;
#define LQ(AH, CNT) hipLaunchKernelGGL((k_trace_quad<AH, CNT>), dim3(grid_blocks), dim3(LH_BLOCK), lds_bytes, s, scl, n, d_org, d_dir, \
                                       d_prim, d_t, d_u, d_v, d_occ, d_counters, d_cursor, min_rays, tri_rays, over_fix)
    if (anyhit) { if (d_counters) LQ(true, true); else LQ(true, false); }
    else { if (d_counters) LQ(false, true); else LQ(false, false); }
#undef LQ
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
