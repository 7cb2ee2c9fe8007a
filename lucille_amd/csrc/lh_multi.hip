/*
 * lh_multi.hip -- the hot path over the G GPUs of one node, from ONE C process.
 *
 * What it stands in for: lucille's bucket queue drained by render threads (render_frame_controller /
 * render_bucket, src/render/render.c:1043-1207) and its compiled-out MPI design "every rank renders,
 * rank 0 owns the display" (src/render/render.c:468-514, src/base/parallel.c:62-232).  Here the workers are
 * GPUs:
 *
 *   * ONE host build of the scene, uploaded to every device (replicated BVH: lh_accel_commit_replica shares
 *     the refcounted host scene -- no G identical builds, no G copies in host memory);
 *   * frames: a queue of tiles drained by one host thread per device (dynamic, so a device that meets empty
 *     sky simply takes more tiles); each finished tile slab goes device-to-device (hipMemcpyPeerAsync: one
 *     xGMI link per peer, transfers from different devices land in parallel) into device 0's slab array --
 *     the display owner -- where one placement kernel applies bucket_write's row order and one copy delivers
 *     the frame to the host;
 *   * ray dumps: contiguous slices (n r / G .. n (r+1) / G), one per device, traced concurrently through the
 *     pipelined host path of every replica.
 *
 * There is no per-ray communication and no collective: the only exchange step is the tile gather.
 * The same device may be listed more than once (two replicas on one GPU): that is how the sharded path is
 * tested on a one-GPU box, bit for bit against the unsharded frame.
 */
#include <hip/hip_runtime.h>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/lucille_hip.h"

extern "C" int lh_accel_commit_replica(lh_accel_t *dst, lh_accel_t *src);      /* lh_api.hip */
extern "C" void lh_set_error(const char *msg);                                   /* lh_api.hip: lh_last_error of this thread */

struct lh_multi {
    int n;
    int *dev;                 /* device ordinal of replica k */
    lh_accel_t **acc;
    int committed;
    /* frame assembly on replica 0's device */
    float *d_slabs; size_t slabs_cap;          /* [ntiles][tile*tile*3] */
    float *d_frame; size_t frame_cap;          /* [H][W][3] */
    int   *d_tiles; size_t tiles_cap;          /* x0,y0,w,h per tile */
    std::vector<float *> d_tile;               /* per replica: one tile slab on its own device */
    std::vector<size_t> tile_cap;
    std::vector<hipStream_t> stream;           /* per replica */
};

static int mfail(const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    lh_set_error(buf);
    return -1;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

extern "C" int lh_multi_create(lh_multi_t **out, int ndevices, const int *devices)
{
    if (!out) return mfail("lh_multi_create: out is NULL");
    const int avail = lh_device_count();
    if (avail <= 0) return mfail("lh_multi_create: no HIP device visible (this library has no CPU fallback)");
    if (ndevices <= 0) ndevices = avail;
    if (ndevices > 64) return mfail("lh_multi_create: %d devices requested", ndevices);
    lh_multi_t *m = new lh_multi();
    m->n = ndevices; m->committed = 0;
    m->dev = (int *)calloc((size_t)ndevices, sizeof(int));
    m->acc = (lh_accel_t **)calloc((size_t)ndevices, sizeof(lh_accel_t *));
    m->d_slabs = NULL; m->slabs_cap = 0; m->d_frame = NULL; m->frame_cap = 0; m->d_tiles = NULL; m->tiles_cap = 0;
    m->d_tile.assign((size_t)ndevices, NULL); m->tile_cap.assign((size_t)ndevices, 0); m->stream.assign((size_t)ndevices, NULL);
    for (int k = 0; k < ndevices; k++) {
        m->dev[k] = devices ? devices[k] : k % avail;
        if (m->dev[k] < 0 || m->dev[k] >= avail) { const int d = m->dev[k]; lh_multi_destroy(m); return mfail("lh_multi_create: device %d out of range [0,%d)", d, avail); }
        if (lh_accel_create(&m->acc[k], m->dev[k]) != 0) { lh_multi_destroy(m); return -1; }
    }
    /* peer access towards replica 0's device, where the tile slabs are gathered (ignored if already on / same device) */
    for (int k = 1; k < ndevices; k++) {
        if (m->dev[k] == m->dev[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->dev[k], m->dev[0]) == hipSuccess && can) {
            (void)hipSetDevice(m->dev[k]); (void)hipDeviceEnablePeerAccess(m->dev[0], 0); (void)hipGetLastError();
        }
    }
    *out = m;
    return 0;
}

extern "C" void lh_multi_destroy(lh_multi_t *m)
{
    if (!m) return;
    for (int k = 0; k < m->n; k++) {
        if (m->d_tile.size() > (size_t)k && m->d_tile[k]) { (void)hipSetDevice(m->dev[k]); (void)hipFree(m->d_tile[k]); }
        if (m->stream.size() > (size_t)k && m->stream[k]) { (void)hipSetDevice(m->dev[k]); (void)hipStreamDestroy(m->stream[k]); }
    }
    if (m->n > 0) {
        (void)hipSetDevice(m->dev[0]);
        if (m->d_slabs) (void)hipFree(m->d_slabs);
        if (m->d_frame) (void)hipFree(m->d_frame);
        if (m->d_tiles) (void)hipFree(m->d_tiles);
    }
    for (int k = 0; k < m->n; k++) if (m->acc[k]) lh_accel_destroy(m->acc[k]);
    free(m->acc); free(m->dev);
    delete m;
}

extern "C" int lh_multi_ndevices(const lh_multi_t *m) { return m ? m->n : 0; }

extern "C" lh_accel_t *lh_multi_accel(lh_multi_t *m, int k) { return (m && k >= 0 && k < m->n) ? m->acc[k] : NULL; }

extern "C" int lh_multi_add_mesh(lh_multi_t *m, uint32_t npositions, const double *positions, size_t stride_bytes,
                                 uint32_t nindices, const uint32_t *indices)
{
    if (!m) return mfail("lh_multi_add_mesh: NULL");
    return lh_accel_add_mesh(m->acc[0], npositions, positions, stride_bytes, nindices, indices);
}

extern "C" int lh_multi_set_normals(lh_multi_t *m, uint32_t mesh, const double *normals, size_t stride_bytes, int two_side)
{
    if (!m) return mfail("lh_multi_set_normals: NULL");
    return lh_accel_set_normals(m->acc[0], mesh, normals, stride_bytes, two_side);
}

extern "C" int lh_multi_add_rib_scene(lh_multi_t *m, const lh_rib_scene_t *scene)
{
    if (!m) return mfail("lh_multi_add_rib_scene: NULL");
    return lh_accel_add_rib_scene(m->acc[0], scene);
}

/* one host build (replica 0), then the upload to the other devices, all at once */
extern "C" int lh_multi_commit(lh_multi_t *m, int build_threads)
{
    if (!m) return mfail("lh_multi_commit: NULL");
    if (m->committed) return mfail("lh_multi_commit: already committed");
    if (lh_accel_commit(m->acc[0], build_threads) != 0) return -1;
    std::vector<int> rc((size_t)m->n, 0);
    std::vector<std::string> err((size_t)m->n);
    std::vector<std::thread> th;
    for (int k = 1; k < m->n; k++)
        th.emplace_back([m, k, &rc, &err] {
            rc[k] = lh_accel_commit_replica(m->acc[k], m->acc[0]);
            if (rc[k] != 0) err[k] = lh_last_error();
        });
    for (auto &t : th) t.join();
    for (int k = 1; k < m->n; k++) if (rc[k] != 0) return mfail("lh_multi_commit: replica %d (device %d): %s", k, m->dev[k], err[k].c_str());
    for (int k = 0; k < m->n; k++) {
        if (hipSetDevice(m->dev[k]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[k], hipStreamNonBlocking) != hipSuccess)
            return mfail("lh_multi_commit: stream creation on device %d failed", m->dev[k]);
    }
    m->committed = 1;
    return 0;
}

/* ---- ray dumps: contiguous slices, one per replica ------------------------------------------------- */
extern "C" int lh_multi_intersect_host(lh_multi_t *m, size_t n, const double *org_xyz, const double *dir_xyz, uint32_t *prim,
                                       double *t, double *u, double *v, uint8_t *occluded, int mode)
{
    if (!m || !m->committed) return mfail("lh_multi_intersect_host: not committed");
    if (n == 0) return 0;
    if (!org_xyz || !dir_xyz) return mfail("lh_multi_intersect_host: NULL ray arrays");
    std::vector<int> rc((size_t)m->n, 0);
    std::vector<std::string> err((size_t)m->n);
    std::vector<std::thread> th;
    for (int k = 0; k < m->n; k++)
        th.emplace_back([=, &rc, &err] {
            const size_t b = n * (size_t)k / (size_t)m->n, e = n * (size_t)(k + 1) / (size_t)m->n;
            if (e == b) return;
            rc[k] = lh_accel_intersect_host(m->acc[k], e - b, org_xyz + 3 * b, dir_xyz + 3 * b, prim ? prim + b : NULL, t ? t + b : NULL,
                                            u ? u + b : NULL, v ? v + b : NULL, occluded ? occluded + b : NULL, mode);
            if (rc[k] != 0) err[k] = lh_last_error();
        });
    for (auto &t_ : th) t_.join();
    for (int k = 0; k < m->n; k++) if (rc[k] != 0) return mfail("lh_multi_intersect_host: replica %d: %s", k, err[k].c_str());
    return 0;
}

/* ---- frames ---------------------------------------------------------------------------------------- */
namespace {

/* slab k holds tile k as lh_render_*_tile delivers it (rows already in image orientation inside the tile):
 * bucket_write's placement (render.c:962-975) into the frame, top row first */
__global__ void k_place_tiles(int ntiles, const int *__restrict__ tiles, const float *__restrict__ slabs, size_t slab_stride,
                              int W, int H, float *__restrict__ frame)
{
    const int tk = blockIdx.y;
    if (tk >= ntiles) return;
    const int x0 = tiles[4 * tk], y0 = tiles[4 * tk + 1], w = tiles[4 * tk + 2], h = tiles[4 * tk + 3];
    const float *s = slabs + slab_stride * (size_t)tk;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)w * h * 3; i += (size_t)gridDim.x * blockDim.x) {
        const size_t px = i / 3; const int c = (int)(i % 3);
        const int r = (int)(px / w), x = (int)(px % w);
        frame[((size_t)(H - (y0 + h) + r) * W + (x0 + x)) * 3 + c] = s[i];
    }
}

struct Tile { int x0, y0, w, h; };

std::vector<Tile> tile_grid(int W, int H, int tile)
{
    std::vector<Tile> out;
    for (int y0 = 0; y0 < H; y0 += tile)
        for (int x0 = 0; x0 < W; x0 += tile)
            out.push_back({x0, y0, (x0 + tile <= W) ? tile : W - x0, (y0 + tile <= H) ? tile : H - y0});
    return out;
}

int ensure_dev(float **p, size_t *cap, size_t bytes)
{
    if (*cap >= bytes && *p) return 0;
    if (*p) { (void)hipFree(*p); *p = NULL; *cap = 0; }
    if (hipMalloc((void **)p, bytes ? bytes : 16) != hipSuccess) return -1;
    *cap = bytes;
    return 0;
}

} /* namespace */

/* The frame loop shared by the AO and the path-traced frame: `render_tile(replica, tile, d_slab)` renders one
 * tile on its replica into that replica's slab; this function owns the queue, the gather and the delivery. */
template <typename F>
static int frame_loop(lh_multi_t *m, int W, int H, int tile, float *rgb, double *device_seconds, F render_tile)
{
    if (W <= 0 || H <= 0) return mfail("frame: bad resolution");
    if (tile <= 0) tile = 512;
    const std::vector<Tile> tiles = tile_grid(W, H, tile);
    const int nt = (int)tiles.size();
    const size_t slab_floats = (size_t)tile * tile * 3;
    if (hipSetDevice(m->dev[0]) != hipSuccess) return mfail("frame: hipSetDevice failed");
    if (ensure_dev(&m->d_slabs, &m->slabs_cap, slab_floats * sizeof(float) * (size_t)nt) != 0 ||
        ensure_dev(&m->d_frame, &m->frame_cap, (size_t)W * H * 3 * sizeof(float)) != 0 ||
        ensure_dev((float **)&m->d_tiles, &m->tiles_cap, sizeof(int) * 4 * (size_t)nt) != 0) return mfail("frame: out of device memory on device %d", m->dev[0]);
    for (int k = 0; k < m->n; k++) {
        if (hipSetDevice(m->dev[k]) != hipSuccess || ensure_dev(&m->d_tile[k], &m->tile_cap[k], slab_floats * sizeof(float)) != 0)
            return mfail("frame: out of device memory on device %d", m->dev[k]);
    }
    std::atomic<int> next(0);
    std::vector<int> rc((size_t)m->n, 0);
    std::vector<std::string> err((size_t)m->n);
    std::vector<double> secs((size_t)m->n, 0.0);
    std::vector<std::thread> th;
    for (int k = 0; k < m->n; k++)
        th.emplace_back([&, k] {
            const double t0 = now_s();
            if (hipSetDevice(m->dev[k]) != hipSuccess) { rc[k] = -1; err[k] = "hipSetDevice failed"; return; }
            for (;;) {
                const int tk = next.fetch_add(1);
                if (tk >= nt) break;
                const Tile &T = tiles[tk];
                if (render_tile(k, T, m->d_tile[k], m->stream[k]) != 0) { rc[k] = -1; err[k] = lh_last_error(); return; }
                /* the exchange step: this tile's slab to the display owner (replica 0's device) */
                const size_t bytes = (size_t)T.w * T.h * 3 * sizeof(float);
                hipError_t e = hipMemcpyPeerAsync(m->d_slabs + slab_floats * (size_t)tk, m->dev[0], m->d_tile[k], m->dev[k], bytes, m->stream[k]);
                if (e == hipSuccess) e = hipStreamSynchronize(m->stream[k]);       /* the slab buffer is reused by the next tile */
                if (e != hipSuccess) { rc[k] = -1; err[k] = hipGetErrorString(e); return; }
            }
            secs[k] = now_s() - t0;
        });
    for (auto &t : th) t.join();
    for (int k = 0; k < m->n; k++) if (rc[k] != 0) return mfail("frame: replica %d (device %d): %s", k, m->dev[k], err[k].c_str());
    if (device_seconds) for (int k = 0; k < m->n; k++) device_seconds[k] = secs[k];
    /* placement on the display owner, then one copy to the host */
    if (hipSetDevice(m->dev[0]) != hipSuccess) return mfail("frame: hipSetDevice failed");
    std::vector<int> flat((size_t)nt * 4);
    for (int i = 0; i < nt; i++) { flat[4 * i] = tiles[i].x0; flat[4 * i + 1] = tiles[i].y0; flat[4 * i + 2] = tiles[i].w; flat[4 * i + 3] = tiles[i].h; }
    hipStream_t s0 = m->stream[0];
    if (hipMemcpyAsync(m->d_tiles, flat.data(), sizeof(int) * 4 * (size_t)nt, hipMemcpyHostToDevice, s0) != hipSuccess) return mfail("frame: tile table upload failed");
    hipLaunchKernelGGL(k_place_tiles, dim3(64, (unsigned)nt), dim3(256), 0, s0, nt, (const int *)m->d_tiles, (const float *)m->d_slabs, slab_floats, W, H, m->d_frame);
    if (hipGetLastError() != hipSuccess) return mfail("frame: placement kernel launch failed");
    if (hipMemcpyAsync(rgb, m->d_frame, (size_t)W * H * 3 * sizeof(float), hipMemcpyDeviceToHost, s0) != hipSuccess ||
        hipStreamSynchronize(s0) != hipSuccess) return mfail("frame: copy to the host failed");
    return 0;
}

extern "C" int lh_multi_render_ao_frame_host(lh_multi_t *m, const lh_camera_t *cam, int pixel_samples, int gather_nsamples,
                                             uint64_t seed, int tile, float *rgb, lh_tile_stats_t *stats, double *device_seconds)
{
    if (!m || !m->committed) return mfail("lh_multi_render_ao_frame_host: not committed");
    if (!cam || !rgb) return mfail("lh_multi_render_ao_frame_host: NULL argument");
    std::vector<lh_tile_stats_t> part((size_t)m->n, lh_tile_stats_t{0, 0, 0, 0});
    const int rc = frame_loop(m, cam->width, cam->height, tile, rgb, device_seconds,
        [&](int k, const Tile &T, float *d_slab, hipStream_t s) {
            lh_tile_stats_t st;
            if (lh_render_ao_tile(m->acc[k], cam, T.x0, T.y0, T.w, T.h, pixel_samples, gather_nsamples, seed, NULL, d_slab, &st, (void *)s) != 0) return -1;
            part[k].primary_rays += st.primary_rays; part[k].primary_hits += st.primary_hits;
            part[k].ao_rays += st.ao_rays; part[k].ao_occluded += st.ao_occluded;
            return 0;
        });
    if (rc != 0) return rc;
    if (stats) {
        *stats = lh_tile_stats_t{0, 0, 0, 0};
        for (int k = 0; k < m->n; k++) {
            stats->primary_rays += part[k].primary_rays; stats->primary_hits += part[k].primary_hits;
            stats->ao_rays += part[k].ao_rays; stats->ao_occluded += part[k].ao_occluded;
        }
    }
    return 0;
}

extern "C" int lh_multi_set_material(lh_multi_t *m, uint32_t mesh, const lh_material_t *material)
{
    /* materials live on the replicas, and replicas 1..n-1 get their meshes in lh_multi_commit */
    if (!m || !m->committed) return mfail("lh_multi_set_material: not committed (set materials after lh_multi_commit)");
    for (int k = 0; k < m->n; k++) if (lh_accel_set_material(m->acc[k], mesh, material) != 0) return -1;
    return 0;
}

extern "C" int lh_multi_set_environment(lh_multi_t *m, const lh_environment_t *environment)
{
    if (!m || !m->committed) return mfail("lh_multi_set_environment: not committed");
    for (int k = 0; k < m->n; k++) if (lh_accel_set_environment(m->acc[k], environment) != 0) return -1;
    return 0;
}

extern "C" int lh_multi_render_pt_frame_host(lh_multi_t *m, const lh_camera_t *cam, int spp, int spp_chunk, int max_path_vertices,
                                             int flags, uint64_t seed, int tile, float *rgb, lh_pt_stats_t *stats, double *device_seconds)
{
    if (!m || !m->committed) return mfail("lh_multi_render_pt_frame_host: not committed");
    if (!cam || !rgb) return mfail("lh_multi_render_pt_frame_host: NULL argument");
    if (spp < 1) return mfail("lh_multi_render_pt_frame_host: bad sample count");
    if (spp_chunk < 1 || spp_chunk > spp) spp_chunk = spp;
    if (spp_chunk > 4096) spp_chunk = 4096;          /* samples of a pixel in one pass (lh_render_pt_tile) */
    std::vector<lh_pt_stats_t> part((size_t)m->n, lh_pt_stats_t{0, 0, 0});
    const int rc = frame_loop(m, cam->width, cam->height, tile, rgb, device_seconds,
        [&](int k, const Tile &T, float *d_slab, hipStream_t s) {
            if (hipMemsetAsync(d_slab, 0, (size_t)T.w * T.h * 3 * sizeof(float), s) != hipSuccess) return -1;
            for (int s0 = 0; s0 < spp; s0 += spp_chunk) {
                lh_pt_stats_t st;
                const int cnt = (s0 + spp_chunk <= spp) ? spp_chunk : spp - s0;
                if (lh_render_pt_tile2(m->acc[k], cam, T.x0, T.y0, T.w, T.h, s0, cnt, spp, max_path_vertices, flags, seed, d_slab, &st, (void *)s) != 0) return -1;
                part[k].paths += st.paths; part[k].rays += st.rays;
                if (st.max_depth_reached > part[k].max_depth_reached) part[k].max_depth_reached = st.max_depth_reached;
            }
            return 0;
        });
    if (rc != 0) return rc;
    if (stats) {
        *stats = lh_pt_stats_t{0, 0, 0};
        for (int k = 0; k < m->n; k++) {
            stats->paths += part[k].paths; stats->rays += part[k].rays;
            if (part[k].max_depth_reached > stats->max_depth_reached) stats->max_depth_reached = part[k].max_depth_reached;
        }
    }
    return 0;
}
