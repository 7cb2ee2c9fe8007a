// Microbenchmark: dependent random 64-byte "node" gathers, the access pattern of BVH traversal.
//   mode 0: one chain per LANE, 4 x dwordx4 per step (current kernel's pattern)
//   mode 1: one chain per QUAD of lanes, 1 x dwordx4 per lane per step (cooperative fetch)
//   mode 2: one chain per lane, 2 x dwordx4 (32-byte nodes)
//   mode 3: one chain per PAIR of lanes, 2 x dwordx4 per lane (64-byte nodes)
//   mode 4: one chain per lane, 1 x dwordx4 (16-byte nodes)
// Each step's next index comes from the loaded data (dependent chain, like traversal).
// Usage: gather <array MB> <steps> <blocks> <mode...>     (GATHER_LDS=<bytes>: dynamic LDS per 256-thread block, to cap
//        the resident blocks per CU at 160 KB / bytes -- the traversal kernel's LDS stack allows 3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define LAUNCH(M, ...) do { if (perturb) hipLaunchKernelGGL((k_gather<M, true>), __VA_ARGS__, alu); else hipLaunchKernelGGL((k_gather<M, false>), __VA_ARGS__, alu); } while (0)
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// PERTURB: modes 1, 3 and 6 originally followed the link unmodified -- a pure functional graph, on which the chains of a
// launch merge into common trunks and cycles (a random mapping's rho structure) and the L2 hit rate climbs: their r01 figures
// (181-193 G/s) overstate the quad-cooperative ceiling.  GATHER_PERTURB=1 (default now) xors the chain's own running sum into
// the link, as modes 0, 2, 4, 5, 7, 8 always did.
template <int MODE, bool PERTURB>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ nodes, uint32_t nnodes, int steps, uint32_t *out, int alu,
                                                const uint4 *__restrict__ tris = nullptr, uint32_t ntris = 0, int node_steps = 8)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    /* GATHER_ALU=N (round 5): N fp32 FMAs per step on the loaded data, eight independent chains, their result folded into the next
     * link -- what a node step's arithmetic does to the gather rate (the 4-wide step: ~136 VALU ops, an 8-wide step: ~250) */
#define ALU_PAD(V) do { if (alu > 0) { float c0 = __uint_as_float(((V).x & 0x7fffffu) | 0x3f800000u), c1 = c0 + 1.f, c2 = c0 + 2.f, c3 = c0 + 3.f, c4 = c0 + 4.f, c5 = c0 + 5.f, c6 = c0 + 6.f, c7 = c0 + 7.f; \
        for (int k = 0; k < alu; k += 8) { c0 = fmaf(c0, 0.9999f, 0.25f); c1 = fmaf(c1, 0.9999f, 0.25f); c2 = fmaf(c2, 0.9999f, 0.25f); c3 = fmaf(c3, 0.9999f, 0.25f); \
                                            c4 = fmaf(c4, 0.9999f, 0.25f); c5 = fmaf(c5, 0.9999f, 0.25f); c6 = fmaf(c6, 0.9999f, 0.25f); c7 = fmaf(c7, 0.9999f, 0.25f); } \
        acc += __float_as_uint(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7) >> 31; } } while (0)
    if (MODE == 0) {
        uint32_t cur = (gid * 2654435761u) % nnodes;
        for (int s = 0; s < steps; s++) {
            const uint4 *p = nodes + 4 * (size_t)cur;
            uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.y + b.y + c.y + d.y;
            ALU_PAD(b);
            cur = (a.x ^ (acc & 1)) % nnodes;
        }
    } else if (MODE == 1) {
        const uint32_t chain = gid >> 2, sub = gid & 3;
        uint32_t cur = (chain * 2654435761u) % nnodes;
        for (int s = 0; s < steps; s++) {
            uint4 a = nodes[4 * (size_t)cur + sub];
            acc += a.y;
            uint32_t nx = __shfl(a.x, (threadIdx.x & 63) & ~3);       // lane 0 of the quad holds the link
            cur = (nx ^ (PERTURB ? (__shfl(acc, (threadIdx.x & 63) & ~3) & 1) : 0)) % nnodes;
        }
    } else if (MODE == 2) {
        uint32_t cur = (gid * 2654435761u) % (nnodes * 2);
        for (int s = 0; s < steps; s++) {
            const uint4 *p = nodes + 2 * (size_t)cur;
            uint4 a = p[0], b = p[1];
            acc += a.y + b.y;
            cur = (a.x ^ (acc & 1)) % (nnodes * 2);
        }
    } else if (MODE == 3) {
        const uint32_t chain = gid >> 1, sub = gid & 1;
        uint32_t cur = (chain * 2654435761u) % nnodes;
        for (int s = 0; s < steps; s++) {
            const uint4 *p = nodes + 4 * (size_t)cur + 2 * sub;
            uint4 a = p[0], b = p[1];
            acc += a.y + b.y;
            uint32_t nx = __shfl(a.x, (threadIdx.x & 63) & ~1);
            cur = (nx ^ (PERTURB ? (__shfl(acc, (threadIdx.x & 63) & ~1) & 1) : 0)) % nnodes;
        }
    } else if (MODE == 5) {          /* 128-byte nodes, one chain per lane, 8 x dwordx4 */
        uint32_t cur = (gid * 2654435761u) % (nnodes / 2);
        for (int s = 0; s < steps; s++) {
            const uint4 *p = nodes + 8 * (size_t)cur;
            uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
            acc += a.y + b.y + c.y + d.y + e.y + f.y + g.y + h.y;
            ALU_PAD(b);
            cur = (a.x ^ (acc & 1)) % (nnodes / 2);
        }
    } else if (MODE == 6) {          /* 128-byte nodes, one chain per QUAD, 2 x dwordx4 per lane */
        const uint32_t chain = gid >> 2, sub = gid & 3;
        uint32_t cur = (chain * 2654435761u) % (nnodes / 2);
        for (int s = 0; s < steps; s++) {
            const uint4 *p = nodes + 8 * (size_t)cur + 2 * sub;
            uint4 a = p[0], b = p[1];
            acc += a.y + b.y;
            uint32_t nx = __shfl(a.x, (threadIdx.x & 63) & ~3);
            cur = (nx ^ (PERTURB ? (__shfl(acc, (threadIdx.x & 63) & ~3) & 1) : 0)) % (nnodes / 2);
        }
    } else if (MODE == 10 || MODE == 11) {
        /* TWO arrays (round 6): `node_steps` dependent node records (64 B: mode 10, 128 B: mode 11) out of
         * `nodes`, then one 48-byte triangle record out of `tris`, and again -- the 8-wide walk's mix on S-soup-10M
         * (39.7 node + 5.1 triangle records per ray).  nnodes counts records of the mode's own size. */
        uint32_t cur = (gid * 2654435761u) % nnodes;
        int k = (int)(gid % (uint32_t)(node_steps + 1));              /* chains out of phase: the mix is the same at every instant */
        for (int s = 0; s < steps; s++) {
            uint32_t link;
            if (k == node_steps) {
                const uint4 *p = tris + 3 * (size_t)((cur * 40503u + acc) % ntris);
                uint4 a = p[0], b = p[1], c = p[2];
                acc += a.y + b.y + c.y; link = a.x; k = 0;
                ALU_PAD(b);
            } else if (MODE == 10) {
                const uint4 *p = nodes + 4 * (size_t)cur;
                uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.y + b.y + c.y + d.y; link = a.x; k++;
                ALU_PAD(b);
            } else {
                const uint4 *p = nodes + 8 * (size_t)cur;
                uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
                acc += a.y + b.y + c.y + d.y + e.y + f.y + g.y + h.y; link = a.x; k++;
                ALU_PAD(b);
            }
            cur = (link ^ (acc & 1)) % nnodes;
        }
    } else if (MODE == 7) {          /* 256-byte nodes (2 lines), one chain per lane, first 5 x dwordx4 (80 B used) */
        uint32_t cur = (gid * 2654435761u) % (nnodes / 4);
        for (int s = 0; s < steps; s++) {
            const uint4 *p = nodes + 16 * (size_t)cur;
            uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
            acc += a.y + b.y + c.y + d.y + e.y;
            cur = (a.x ^ (acc & 1)) % (nnodes / 4);
        }
    } else if (MODE == 8) {          /* 64-byte nodes, one chain per LANE, quad-transposed fetch: in phase p every lane
                                      * of a quad loads one 16-byte piece of quad-lane p's node (4 lanes -> one 64-byte
                                      * contiguous request), then the quad transposes the pieces with DPP */
        const int j = threadIdx.x & 3;
        uint32_t cur = (gid * 2654435761u) % nnodes;
        for (int s = 0; s < steps; s++) {
            uint4 R[4];
#define QB(v, P) (uint32_t)__builtin_amdgcn_mov_dpp((int)(v), (P) * 0x55, 0xf, 0xf, true)
            R[0] = nodes[4 * (size_t)QB(cur, 0) + j];
            R[1] = nodes[4 * (size_t)QB(cur, 1) + j];
            R[2] = nodes[4 * (size_t)QB(cur, 2) + j];
            R[3] = nodes[4 * (size_t)QB(cur, 3) + j];
            /* butterfly transpose of the 4x4 (lane x register) matrix of 16-byte pieces */
#define XSWAP(A, B, CTRL, BIT) { \
                const bool hi = (j & (BIT)) != 0; \
                uint4 snd = hi ? A : B, rcv; \
                rcv.x = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd.x, CTRL, 0xf, 0xf, true); \
                rcv.y = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd.y, CTRL, 0xf, 0xf, true); \
                rcv.z = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd.z, CTRL, 0xf, 0xf, true); \
                rcv.w = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd.w, CTRL, 0xf, 0xf, true); \
                if (hi) A = rcv; else B = rcv; }
            XSWAP(R[0], R[1], 0xB1, 1)       /* quad_perm [1,0,3,2] */
            XSWAP(R[2], R[3], 0xB1, 1)
            XSWAP(R[0], R[2], 0x4E, 2)       /* quad_perm [2,3,0,1] */
            XSWAP(R[1], R[3], 0x4E, 2)
            acc += R[0].y + R[1].y + R[2].y + R[3].y;
            cur = (R[0].x ^ (acc & 1)) % nnodes;
        }
    } else {
        uint32_t cur = (gid * 2654435761u) % (nnodes * 4);
        for (int s = 0; s < steps; s++) {
            uint4 a = nodes[cur];
            acc += a.y;
            cur = (a.x ^ (acc & 1)) % (nnodes * 4);
        }
    }
    out[gid] = acc;
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? atol(argv[1]) : 64;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    const int blocks = argc > 3 ? atoi(argv[3]) : 1024;
    const uint32_t nnodes = (uint32_t)(mb * 1024 * 1024 / 64);
    std::vector<uint32_t> h((size_t)nnodes * 16);
    uint64_t x = 88172645463325252ULL;
    for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 20); }
    uint4 *d; uint32_t *out;
    CHK(hipMalloc(&d, h.size() * 4)); CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const bool perturb = !(getenv("GATHER_PERTURB") && atoi(getenv("GATHER_PERTURB")) == 0);
    const size_t lds = getenv("GATHER_LDS") ? (size_t)atol(getenv("GATHER_LDS")) : 0;
    const int alu = getenv("GATHER_ALU") ? atoi(getenv("GATHER_ALU")) : 0;
    if (alu) printf("%d FMAs per step between the load and the next link\n", alu);
    if (lds) printf("dynamic LDS %zu bytes per block: at most %zu blocks (%zu waves per SIMD) per CU\n", lds, (size_t)(160 * 1024) / lds, (size_t)(160 * 1024) / lds);
    /* GATHER_NODE_MB / GATHER_TRI_MB (modes 10, 11): node records and triangle records in arrays of their own */
    const size_t node_mb = getenv("GATHER_NODE_MB") ? (size_t)atol(getenv("GATHER_NODE_MB")) : 0, tri_mb = getenv("GATHER_TRI_MB") ? (size_t)atol(getenv("GATHER_TRI_MB")) : 0;
    const int node_steps = getenv("GATHER_NODE_STEPS") ? atoi(getenv("GATHER_NODE_STEPS")) : 8;
    uint4 *dn = nullptr, *dt = nullptr;
    if (node_mb && tri_mb) {
        std::vector<uint32_t> hh((node_mb > tri_mb ? node_mb : tri_mb) * 1024 * 1024 / 4);
        for (size_t i = 0; i < hh.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hh[i] = (uint32_t)(x >> 20); }
        CHK(hipMalloc(&dn, node_mb << 20)); CHK(hipMalloc(&dt, tri_mb << 20));
        CHK(hipMemcpy(dn, hh.data(), node_mb << 20, hipMemcpyHostToDevice)); CHK(hipMemcpy(dt, hh.data(), tri_mb << 20, hipMemcpyHostToDevice));
    }
    for (int a = 4; a < argc; a++) {
        const int mode = atoi(argv[a]);
        float best = 1e30f;
        if (mode == 10 || mode == 11) {
            if (!dn) { printf("modes 10 / 11 need GATHER_NODE_MB and GATHER_TRI_MB\n"); return 1; }
            const uint32_t nrec = (uint32_t)((node_mb << 20) / (mode == 10 ? 64 : 128)), ntri = (uint32_t)((tri_mb << 20) / 48);
            for (int rep = 0; rep < 4; rep++) {
                CHK(hipEventRecord(e0));
                if (mode == 10) hipLaunchKernelGGL((k_gather<10, true>), dim3(blocks), dim3(256), lds, 0, dn, nrec, steps, out, alu, dt, ntri, node_steps);
                else            hipLaunchKernelGGL((k_gather<11, true>), dim3(blocks), dim3(256), lds, 0, dn, nrec, steps, out, alu, dt, ntri, node_steps);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms;
            }
            const double g = (double)blocks * 256 * steps / (best * 1e-3) / 1e9;
            printf("nodes %4zu MB of %3d-byte records, triangles %4zu MB of 48-byte records, %d node steps per triangle step, blocks %5d mode %d: %.3f ms  %.2f G records/s\n",
                   node_mb, mode == 10 ? 64 : 128, tri_mb, node_steps, blocks, mode, best, g);
            continue;
        }
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0));
            switch (mode) {
            case 0: LAUNCH(0, dim3(blocks), dim3(256), lds, 0, d, nnodes, steps, out); break;
            case 1: LAUNCH(1, dim3(blocks), dim3(256), lds, 0, d, nnodes, steps, out); break;
            case 2: LAUNCH(2, dim3(blocks), dim3(256), lds, 0, d, nnodes, steps, out); break;
            case 3: LAUNCH(3, dim3(blocks), dim3(256), lds, 0, d, nnodes, steps, out); break;
            case 5: LAUNCH(5, dim3(blocks), dim3(256), lds, 0, d, nnodes, steps, out); break;
            case 6: LAUNCH(6, dim3(blocks), dim3(256), lds, 0, d, nnodes, steps, out); break;
            case 8: LAUNCH(8, dim3(blocks), dim3(256), lds, 0, d, nnodes, steps, out); break;
            case 7: LAUNCH(7, dim3(blocks), dim3(256), lds, 0, d, nnodes, steps, out); break;
            default: LAUNCH(4, dim3(blocks), dim3(256), lds, 0, d, nnodes, steps, out); break;
            }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms;
        }
        const double chains = (double)blocks * 256 / ((mode == 1 || mode == 6) ? 4 : (mode == 3 ? 2 : 1));
        const double bytes_per_step = (mode == 2) ? 32 : (mode == 4 ? 16 : ((mode == 5 || mode == 6) ? 128 : (mode == 7 ? 80 : 64)));
        const double gsteps = chains * steps / (best * 1e-3) / 1e9;
        printf("array %4zu MB blocks %5d mode %d: %.3f ms  %.2f G chain-steps/s  %.2f TB/s useful\n", mb, blocks, mode, best, gsteps,
               gsteps * bytes_per_step / 1e3);
    }
    return 0;
}
