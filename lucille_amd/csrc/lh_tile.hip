/*
 * lh_tile.hip -- the callers on either side of the query, kept on the device: AO tiles / bands / frames (subsample,
 * render_bucket, bucket_write: src/render/render.c:715-823,1107-1166,919-983; ri_transport_ambientocclusion:
 * src/transport/ambientocclusion.c:42-151,332-415), the hit epilogue (ri_intersection_state_build,
 * src/render/intersection_state.c:99-248) and the wavefront path tracer's tile loop (src/transport/pathtrace.c:189-314,
 * 407-537).  The kernels are in lh_render.hip / lh_kernels.hip.
 */
#include <algorithm>
#include <thread>
#include <vector>

#include "lh_internal.h"

#define ensure_buf lh_ensure_buf

/* launchers in lh_render.hip */
extern "C" int lh_render_launch_primary(const lh_camera_t *cam, int x0, int y0, int w, int h, int xs, int ys,
                                        double *d_org, double *d_dir, void *stream);
extern "C" int lh_render_launch_compact(const lh_dev_scene_t *sc, const double *d_nrm9, size_t n, const double *d_org,
                                        const double *d_dir, const uint32_t *d_prim, const double *d_t,
                                        const double *d_u, const double *d_v, uint32_t *d_block_counts,
                                        uint32_t *d_slot_of_sample, double *d_hitrec,
                                        unsigned long long *d_slot_key, int x0, int w, int nbands, int band_rows,
                                        const int *d_band_y0, int y0, int spp, int full_width,
                                        unsigned long long *d_total, void *stream);
extern "C" int lh_render_launch_primary_region(const lh_camera_t *cam, int x0, int w, int nbands, int band_rows, const int *d_band_y0,
                                               int y0, int height_limit, int xs, int ys, double *d_org, double *d_dir, void *stream);
extern "C" int lh_render_launch_ao_rays(size_t nslots, int ntheta, int nphi, unsigned long long seed,
                                        const double *d_hitrec, const double *d_rnd,
                                        const unsigned long long *d_slot_key, double *d_org, double *d_dir, void *stream);
extern "C" int lh_render_launch_resolve(int w, int h, int band_rows, int xs, int ys, int N, const uint32_t *d_slot_of_sample,
                                        const uint8_t *d_occ, const unsigned int *d_occ_count, float *d_rgb,
                                        unsigned long long *d_occ_total, void *stream);

extern "C" int lh_render_primary_rays(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h,
                                      int ps, void *d_org, void *d_dir, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_primary_rays: accel not committed");
    if (!cam || !d_org || !d_dir) return fail("lh_render_primary_rays: NULL argument");
    if (w < 0 || h < 0 || ps < 1) return fail("lh_render_primary_rays: bad tile");
    HIPCHK(hipSetDevice(a->device));
    if (lh_render_launch_primary(cam, x0, y0, w, h, ps, ps, (double *)d_org, (double *)d_dir, stream) != 0)
        return fail("primary ray kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

/* one device batch of the AO pipeline over a Region (lh_render.hip): a rectangle, or nbands full-width bands */
static int ao_region(lh_accel_t *a, const lh_camera_t *cam, int x0, int w, int nbands, int band_rows, const int *d_band_y0, int y0,
                     uint64_t valid_pixels, int ps, int gather_nsamples, uint64_t seed, const void *d_uniforms, void *d_rgb,
                     lh_tile_stats_t *stats, void *stream)
{
    const int h = nbands * band_rows;              /* lines of the batch */
    HIPCHK(hipSetDevice(a->device));
    hipStream_t s = (hipStream_t)stream;
    const int nphi = (int)sqrt((double)gather_nsamples), ntheta = nphi, N = nphi * ntheta;   /* ambientocclusion.c:378-380 */
    const size_t S = (size_t)w * h * ps * ps;
    if ((unsigned long long)cam->width * (unsigned long long)cam->height * (unsigned long long)(ps * ps) >= (1ull << 34))
        return fail("AO pipeline: more than 2^34 samples in the frame (slot keys carry 34 bits)");
    if (S >= ((size_t)1 << 31)) return fail("AO pipeline: more than 2^31 samples in one batch; render the frame in tiles");     /* 32-bit sample indices on the device */
    const unsigned nb = (unsigned)((S + 255) / 256);
    if (ensure_buf(&a->r_org, S * 24) || ensure_buf(&a->r_dir, S * 24) || ensure_buf(&a->r_prim, S * 4) ||
        ensure_buf(&a->r_t, S * 8) || ensure_buf(&a->r_u, S * 8) || ensure_buf(&a->r_v, S * 8) ||
        ensure_buf(&a->r_slot, S * 4) || ensure_buf(&a->r_blocks, (size_t)nb * 4)) return -1;
    /* lh_accel_trace_statistics: the counting instantiations of the same kernels, accumulated over the batch */
    unsigned long long *cnt = (a->stat_on && a->hs->bvh.ntris) ? a->d_counters : NULL;
    if (cnt) HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * LH_CNT_DEV, s));
    /* LH_STAGE_TIMING=1: HIP events between the stages of the batch, printed to stderr (tools/experiments/rank_breakdown.py) */
    const bool stage_timing = getenv("LH_STAGE_TIMING") != NULL;
    hipEvent_t ev[6] = {NULL, NULL, NULL, NULL, NULL, NULL};
    if (stage_timing) { for (int k = 0; k < 6; k++) HIPCHK(hipEventCreate(&ev[k])); HIPCHK(hipEventRecord(ev[0], s)); }
    /* ... and the wall clock at which every persistent wave starts and leaves (two launches: closest, AO) */
    const size_t nwaves = (size_t)a->grid_blocks * (LH_BLOCK / 64);
    if (stage_timing) { if (ensure_buf(&a->r_diag, sizeof(unsigned long long) * 6 * nwaves)) return -1; HIPCHK(hipMemsetAsync(a->r_diag.p, 0, sizeof(unsigned long long) * 6 * nwaves, s)); }
    a->dev.diag_clock = stage_timing ? (unsigned long long *)a->r_diag.p : NULL;
    /* 1. camera rays */
    if (lh_render_launch_primary_region(cam, x0, w, nbands, band_rows, d_band_y0, y0, cam->height, ps, ps,
                                        (double *)a->r_org.p, (double *)a->r_dir.p, s) != 0)
        return fail("primary ray kernel launch failed");
    if (stage_timing) HIPCHK(hipEventRecord(ev[1], s));
    /* 2. closest hit */
    if (lh_launch(a, S, a->r_org.p, a->r_dir.p, a->r_prim.p, a->r_t.p, a->r_u.p, a->r_v.p, NULL, LH_MODE_CLOSEST,
               LH_VARIANT_DEFAULT, cnt, s, false) != 0) return -1;
    if (stage_timing) { HIPCHK(hipEventRecord(ev[2], s)); a->dev.diag_clock = (unsigned long long *)a->r_diag.p + 3 * nwaves; }
    /* 3. compaction (deterministic: hits in sample order).  The fused AO stage does not need the total on the host: its buffers are
     * sized for the worst case (every sample hits) and its kernels read the count where the compaction left it -- a batch costs ONE
     * host round trip, at its end (round 5: the two in the middle were ~0.25 ms of a rank's 8.9 ms share of the config-5 frame).
     * The materialised stage (caller uniforms: the parity replay; LH_AO_FUSED=0; a fix-up queue that overflowed) sizes its ray
     * arrays from the count and reads it first. */
    unsigned long long nhit = 0, nocc = 0;
    unsigned long long *d_nhit = a->d_total + 64;          /* the compaction's total, kept clear of k_ao_resolve's 64 counters */
    const size_t nao_max = S * (size_t)N;
    bool fused = a->ao_fused && !d_uniforms && a->hs->bvh.ntris;
    const bool late_count = fused && nao_max < ((size_t)1 << 31) && !a->dev.ao_group && !getenv("LH_AO_SYNC");      /* the persistent kernel's 32-bit ray index covers the worst case */
    if (a->hs->bvh.ntris) {
        if (ensure_buf(&a->r_hitrec, S * 96) || ensure_buf(&a->r_key, S * 8)) return -1;   /* worst case: every sample hits */
        if (lh_render_launch_compact(&a->dev, (const double *)a->d_nrm9, S, (const double *)a->r_org.p,
                                     (const double *)a->r_dir.p, (const uint32_t *)a->r_prim.p, (const double *)a->r_t.p,
                                     (const double *)a->r_u.p, (const double *)a->r_v.p, (uint32_t *)a->r_blocks.p,
                                     (uint32_t *)a->r_slot.p, (double *)a->r_hitrec.p, (unsigned long long *)a->r_key.p,
                                     x0, w, nbands, band_rows, d_band_y0, y0, ps * ps, cam->width, a->d_total, s) != 0)
            return fail("compaction kernels failed: %s", hipGetErrorString(hipGetLastError()));
        HIPCHK(hipMemcpyAsync(d_nhit, a->d_total, sizeof(nhit), hipMemcpyDeviceToDevice, s));
        if (!late_count) {
            HIPCHK(hipMemcpyAsync(&nhit, d_nhit, sizeof(nhit), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (nhit * (unsigned long long)N >= (1ull << 31)) fused = false;
        }
    } else {
        HIPCHK(hipMemsetAsync(a->r_slot.p, 0xFF, S * 4, s));
        HIPCHK(hipMemsetAsync(d_nhit, 0, sizeof(nhit), s));
    }
    if (stage_timing) HIPCHK(hipEventRecord(ev[3], s));
    /* AO stage.  Fused (default): the any-hit kernel generates ray (slot, r) in its refill (lh_ao.h) and counts
     * the occluded rays per slot -- nothing per AO ray goes through HBM.  Materialised: caller uniforms (the parity
     * replay), LH_AO_FUSED=0, or a fix-up queue overflow of the fused launch. */
    if (fused && !late_count && nhit == 0) fused = false;          /* nothing was hit: nothing to trace, the resolve sees misses only */
    const bool fused_tried = fused;
    int qslot = -1;
    /* the batch's read-backs land in pinned memory: three small copies behind the last kernel, one wait */
    if (!a->h_read) HIPCHK(hipHostMalloc(&a->h_read, 1024, hipHostMallocDefault));
    unsigned long long *h_nocc64 = (unsigned long long *)a->h_read, *h_nhit = h_nocc64 + 64; uint32_t *h_qc = (uint32_t *)(h_nocc64 + 65);
    h_qc[0] = h_qc[1] = 0u;
    if (fused && (late_count || nhit)) {
        const size_t nslots = late_count ? S : (size_t)nhit;
        if (ensure_buf(&a->r_occcount, nslots * sizeof(unsigned int))) return -1;
        qslot = lh_aoq_slot(a, s);
        if (qslot < 0) return -1;
        const uint32_t budget_keep = a->dev.ray_budget;
        if (a->ao_budget) a->dev.ray_budget = a->ao_budget;
        /* the tail a budget costs a launch is fixed, the queue a low budget sends to the sweep grows with the launch: a launch of
         * 2^27 rays or more doubles the default (config 5: whole frame 57.6 -> 56.7 ms, half of it 30.8 -> 29.9; a quarter and an
         * eighth are best at 384 -- tools/ao_budget_probe.py).  With the count on the device the kernel picks between the two */
        const uint32_t big = (a->ao_budget && !a->ao_budget_user) ? 2u * a->ao_budget : 0u;
        if (!late_count && big && nhit * (unsigned long long)N >= (1ull << 27)) a->dev.ray_budget = big;
        const int rc_ao = lh_launch_trace_ao(&a->dev, nslots, ntheta, nphi, seed, (const double *)a->r_hitrec.p, (const unsigned long long *)a->r_key.p,
                               (unsigned int *)a->r_occcount.p, cnt, (unsigned long long *)((uint32_t *)a->d_cursor + (size_t)LH_CURSOR_WORDS * (a->cursor_next++ % LH_NCURSOR)), a->grid_blocks,
                               a->min_active, a->tri_batch, &a->aoq[qslot].q, a->ncus, late_count ? d_nhit : NULL, late_count ? big : 0u, (void *)s);
        a->dev.ray_budget = budget_keep;
        if (rc_ao != 0) return fail("fused AO launch failed: %s", hipGetErrorString(hipGetLastError()));
        if (!late_count) {
            uint32_t qc[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(qc, a->aoq[qslot].q.qcount, sizeof(qc), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (qc[1] != 0) fused = false;             /* more than LH_AO_QCAP uncertain AO rays: redo the stage materialised */
            if (stage_timing) fprintf(stderr, "[lucille_hip]   fused AO stage: %u rays through the fix-up queue (budget %u)\n", qc[0], a->dev.ray_budget);
        }
    }
    /* stages 4-6 with the AO rays in HBM; also the second try of a batch whose fused launch overflowed its queue */
    auto materialised = [&]() -> int {
        const size_t nao_m = (size_t)nhit * N;
        if (nao_m) {
            if (ensure_buf(&a->r_aorg, nao_m * 24) || ensure_buf(&a->r_adir, nao_m * 24) || ensure_buf(&a->r_occ, nao_m)) return -1;
            /* 4. AO rays */
            if (lh_render_launch_ao_rays(nhit, ntheta, nphi, seed, (const double *)a->r_hitrec.p, (const double *)d_uniforms,
                                         (const unsigned long long *)a->r_key.p, (double *)a->r_aorg.p, (double *)a->r_adir.p, s) != 0)
                return fail("AO ray kernel launch failed");
            /* 5. any-hit */
            if (cnt && fused_tried) HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * LH_CNT_DEV, s));   /* the abandoned fused pass is not counted (nor are the camera rays then) */
            if (lh_launch(a, nao_m, a->r_aorg.p, a->r_adir.p, NULL, NULL, NULL, NULL, a->r_occ.p, LH_MODE_ANY,
                       LH_VARIANT_DEFAULT, cnt, s, false) != 0) return -1;
        }
        return 0;
    };
    auto resolve = [&](bool from_counts) -> int {
        /* 6. radiance */
        HIPCHK(hipMemsetAsync(a->d_total, 0, sizeof(unsigned long long) * 64, s));
        if (lh_render_launch_resolve(w, h, band_rows, ps, ps, N, (const uint32_t *)a->r_slot.p, (const uint8_t *)a->r_occ.p,
                                     from_counts ? (const unsigned int *)a->r_occcount.p : NULL, (float *)d_rgb, a->d_total, s) != 0)
            return fail("resolve kernel launch failed");
        HIPCHK(hipMemcpyAsync(h_nocc64, a->d_total, sizeof(unsigned long long) * 64, hipMemcpyDeviceToHost, s));
        return 0;
    };
    if (!fused && materialised() != 0) return -1;
    if (stage_timing) HIPCHK(hipEventRecord(ev[4], s));
    if (resolve(fused) != 0) return -1;
    if (stage_timing) HIPCHK(hipEventRecord(ev[5], s));
    if (late_count) {
        /* the batch's one round trip: hit count, occlusion totals, the queue's overflow flag */
        HIPCHK(hipMemcpyAsync(h_nhit, d_nhit, sizeof(nhit), hipMemcpyDeviceToHost, s));
        if (qslot >= 0) HIPCHK(hipMemcpyAsync(h_qc, a->aoq[qslot].q.qcount, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        nhit = *h_nhit;
        if (stage_timing) fprintf(stderr, "[lucille_hip]   fused AO stage: %u rays through the fix-up queue (budget %u)\n", h_qc[0], a->ao_budget ? a->ao_budget : a->dev.ray_budget);
        if (h_qc[1] != 0) {                           /* more than LH_AO_QCAP uncertain AO rays: the stage once more, materialised */
            fused = false;
            if (materialised() != 0 || resolve(false) != 0) return -1;
            HIPCHK(hipStreamSynchronize(s));
        }
    } else HIPCHK(hipStreamSynchronize(s));
    const size_t nao = (size_t)nhit * N;
    for (int k = 0; k < 64; k++) nocc += h_nocc64[k];
    if (stage_timing) {
        float ms[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < 5; k++) (void)hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]);
        fprintf(stderr, "[lucille_hip] AO batch stages (ms): primary %.3f closest %.3f compact %.3f ao %.3f resolve %.3f | samples %zu hits %llu ao rays %zu\n",
                ms[0], ms[1], ms[2], ms[3], ms[4], S, nhit, nao);
        for (int k = 0; k < 6; k++) (void)hipEventDestroy(ev[k]);
        std::vector<unsigned long long> clk(6 * nwaves);
        HIPCHK(hipMemcpy(clk.data(), a->r_diag.p, sizeof(unsigned long long) * 6 * nwaves, hipMemcpyDeviceToHost));
        for (int launch = 0; launch < 2; launch++) {
            const unsigned long long *st = clk.data() + 3 * nwaves * launch, *ex = st + nwaves, *dry = ex + nwaves;
            unsigned long long t0 = ~0ull; std::vector<double> e;
            for (size_t w = 0; w < nwaves; w++) if (st[w] && st[w] < t0) t0 = st[w];
            for (size_t w = 0; w < nwaves; w++) if (ex[w]) e.push_back((double)(ex[w] - t0) * 1e-5);        /* 100 MHz -> ms */
            if (e.empty()) continue;
            std::sort(e.begin(), e.end());
            auto q = [&](double f) { return e[(size_t)(f * (e.size() - 1))]; };
            double last_start = 0; for (size_t w = 0; w < nwaves; w++) if (st[w]) last_start = fmax(last_start, (double)(st[w] - t0) * 1e-5);
            fprintf(stderr, "[lucille_hip]   %s kernel: %zu waves, last start %.3f ms; exits (ms) min %.3f p10 %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f\n",
                    launch ? "AO" : "closest", e.size(), last_start, e.front(), q(0.10), q(0.50), q(0.90), q(0.99), e.back());
            /* when a wave found every cursor dry, and how long it went on after that (its last range and its slowest last rays) */
            std::vector<double> d, g;
            for (size_t w = 0; w < nwaves; w++) if (dry[w] && ex[w]) { d.push_back((double)(dry[w] - t0) * 1e-5); g.push_back((double)(ex[w] - dry[w]) * 1e-5); }
            if (!d.empty()) {
                std::sort(d.begin(), d.end()); std::sort(g.begin(), g.end());
                auto qq = [&](std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
                fprintf(stderr, "[lucille_hip]     cursors found dry at (ms) min %.3f p50 %.3f p90 %.3f max %.3f; exit - dry (ms) min %.3f p10 %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f (%zu waves)\n",
                        d.front(), qq(d, 0.5), qq(d, 0.9), d.back(), g.front(), qq(g, 0.1), qq(g, 0.5), qq(g, 0.9), qq(g, 0.99), g.back(), d.size());
            }
        }
        a->dev.diag_clock = NULL;
    }
    a->r_nsamples = S; a->r_nslots = (size_t)nhit; a->r_nao = fused ? 0 : nao;
    if (cnt) {
        unsigned long long hc[LH_CNT_DEV];
        HIPCHK(hipMemcpy(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost));
        a->stat[0] += hc[LH_CNT_NODES]; a->stat[1] += hc[LH_CNT_TRIS]; a->stat[2] += hc[LH_CNT_EXACT];
        a->stat[3] += hc[LH_CNT_RAYS]; a->stat[4] += nhit + nocc;
        a->stat_slots[0] += hc[LH_CNT_NODE_SLOTS]; a->stat_slots[1] += hc[LH_CNT_TRI_SLOTS]; a->stat_slots[2] += hc[LH_CNT_REGROUP_SLOTS];
        if (getenv("LH_DEBUG_COUNTERS")) {
            fprintf(stderr, "[lucille_hip] AO batch: rays by node visits (bucket b: [2^(b-1), 2^b)):");
            for (int b = 0; b < 24; b++) fprintf(stderr, " %llu", hc[LH_CNT_HIST + b]);
            fprintf(stderr, "\n");
        }
    }
    if (stats) {
        stats->primary_rays = valid_pixels * (uint64_t)(ps * ps); stats->primary_hits = nhit; stats->ao_rays = nao; stats->ao_occluded = nocc;
    }
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

extern "C" int lh_render_ao_tile(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int ps,
                                 int gather_nsamples, uint64_t seed, const void *d_uniforms, void *d_rgb,
                                 lh_tile_stats_t *stats, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_ao_tile: accel not committed");
    if (!cam || !d_rgb) return fail("lh_render_ao_tile: NULL argument");
    if (w <= 0 || h <= 0 || ps < 1 || gather_nsamples < 1) return fail("lh_render_ao_tile: bad tile/sample counts");
    return ao_region(a, cam, x0, w, 1, h, NULL, y0, (uint64_t)w * h, ps, gather_nsamples, seed, d_uniforms, d_rgb, stats, stream);
}

/* nbands full-width bands of band_rows lines (band b = frame lines band_y0[b] ...; a band that runs past the frame is
 * clipped) as ONE device batch: how a rank renders all of its interleaved shards of a frame with one set of launches.
 * d_rgb: float[nbands][band_rows][width][3], every band in image orientation (top line first) like a tile. */
extern "C" int lh_render_ao_bands(lh_accel_t *a, const lh_camera_t *cam, int nbands, const int *band_y0, int band_rows, int ps,
                                  int gather_nsamples, uint64_t seed, void *d_rgb, lh_tile_stats_t *stats, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_ao_bands: accel not committed");
    if (!cam || !d_rgb || (nbands > 0 && !band_y0)) return fail("lh_render_ao_bands: NULL argument");
    if (nbands < 0 || band_rows <= 0 || ps < 1 || gather_nsamples < 1) return fail("lh_render_ao_bands: bad band/sample counts");
    if (nbands == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return 0; }
    HIPCHK(hipSetDevice(a->device));
    uint64_t valid = 0;
    for (int b = 0; b < nbands; b++) {
        if (band_y0[b] < 0 || band_y0[b] >= cam->height) return fail("lh_render_ao_bands: band %d starts at line %d outside the frame", b, band_y0[b]);
        const int rows = (band_y0[b] + band_rows <= cam->height) ? band_rows : cam->height - band_y0[b];
        valid += (uint64_t)rows * cam->width;
    }
    if (ensure_buf(&a->r_bands, sizeof(int) * (size_t)nbands)) return -1;
    HIPCHK(hipMemcpyAsync(a->r_bands.p, band_y0, sizeof(int) * (size_t)nbands, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));          /* band_y0 is the caller's memory */
    return ao_region(a, cam, 0, cam->width, nbands, band_rows, (const int *)a->r_bands.p, 0, valid, ps, gather_nsamples, seed, NULL,
                     d_rgb, stats, stream);
}

/* the same tile for a plain-C host program: uniforms (optional) come from and the tile goes to HOST memory */
extern "C" int lh_render_ao_tile_host(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int ps,
                                      int gather_nsamples, uint64_t seed, const double *uniforms, size_t nuniforms,
                                      float *rgb, lh_tile_stats_t *stats)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_ao_tile_host: accel not committed");
    if (!cam || !rgb) return fail("lh_render_ao_tile_host: NULL argument");
    if (w <= 0 || h <= 0) return fail("lh_render_ao_tile_host: bad tile");
    HIPCHK(hipSetDevice(a->device));
    const size_t fb = (size_t)w * h * 3 * sizeof(float);
    if (ensure_buf(&a->r_frame, fb)) return -1;
    void *d_uni = NULL;
    if (uniforms) {
        const int nphi = (int)sqrt((double)gather_nsamples);
        const size_t need = (size_t)2 * nphi * nphi * w * h * ps * ps;       /* worst case: every sample hits */
        if (nuniforms < need) return fail("lh_render_ao_tile_host: %zu uniforms given, the tile may consume %zu", nuniforms, need);
        if (ensure_buf(&a->r_uni, need * sizeof(double))) return -1;
        HIPCHK(hipMemcpyAsync(a->r_uni.p, uniforms, need * sizeof(double), hipMemcpyHostToDevice, a->stream));
        d_uni = a->r_uni.p;
    }
    if (lh_render_ao_tile(a, cam, x0, y0, w, h, ps, gather_nsamples, seed, d_uni, a->r_frame.p, stats, a->stream) != 0) return -1;
    HIPCHK(hipMemcpyAsync(rgb, a->r_frame.p, fb, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}

extern "C" int lh_render_scratch(lh_accel_t *a, int which, void **d_ptr, size_t *count)
{
    lh_guard guard(a);
    if (!a || !a->committed || !d_ptr || !count) return fail("lh_render_scratch: bad argument");
    lh_buf *b[] = {&a->r_org, &a->r_dir, &a->r_prim, &a->r_t, &a->r_u, &a->r_v, &a->r_slot, &a->r_hitrec,
                   &a->r_aorg, &a->r_adir, &a->r_occ};
    if (which < 0 || which > 10) return fail("lh_render_scratch: unknown buffer %d", which);
    *d_ptr = b[which]->p;
    *count = which <= 6 ? a->r_nsamples : (which == 7 ? a->r_nslots : a->r_nao);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* hit epilogue for a batch (ri_intersection_state_build)                   */
/* ------------------------------------------------------------------------ */
extern "C" int lh_render_launch_state_build(size_t n, const lh_dev_scene_t *sc, const double *d_nrm9, const double *d_col9,
                                            const double *d_tan9, const double *d_bin9, const double *d_st6, const uint8_t *d_inside,
                                            const double *d_org, const double *d_dir, const uint32_t *d_prim, const double *d_t,
                                            const double *d_u, const double *d_v, double *d_state, void *stream);

extern "C" int lh_accel_state_build_device(lh_accel_t *a, size_t n, const void *d_org, const void *d_dir, const void *d_prim,
                                           const void *d_t, const void *d_u, const void *d_v, void *d_state, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_accel_state_build_device: accel not committed");
    if (n == 0 || a->hs->bvh.ntris == 0) return 0;
    if (!d_org || !d_dir || !d_prim || !d_t || !d_u || !d_v || !d_state) return fail("lh_accel_state_build_device: NULL argument");
    HIPCHK(hipSetDevice(a->device));
    if (lh_render_launch_state_build(n, &a->dev, (const double *)a->d_nrm9, (const double *)a->d_attr9[0], (const double *)a->d_attr9[1],
                                     (const double *)a->d_attr9[2], (const double *)a->d_st6, (const uint8_t *)a->d_inside,
                                     (const double *)d_org, (const double *)d_dir, (const uint32_t *)d_prim, (const double *)d_t,
                                     (const double *)d_u, (const double *)d_v, (double *)d_state, stream) != 0)
        return fail("state-build kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int lh_accel_state_build_host(lh_accel_t *a, size_t n, const double *org, const double *dir, const uint32_t *prim,
                                         const double *t, const double *u, const double *v, double *state)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_accel_state_build_host: accel not committed");
    if (n == 0) return 0;
    if (!org || !dir || !prim || !t || !u || !v || !state) return fail("lh_accel_state_build_host: NULL argument");
    HIPCHK(hipSetDevice(a->device));
    const size_t b_ray = sizeof(double) * 3 * n, b_d = sizeof(double) * n, b_state = sizeof(double) * LH_STATE_DOUBLES * n;
    if (ensure_buf(&a->r_state, 2 * b_ray + 3 * b_d + sizeof(uint32_t) * n + 8 + b_state)) return -1;
    char *base = (char *)a->r_state.p;
    double *d_state = (double *)base, *d_org = (double *)(base + b_state), *d_dir = d_org + 3 * n, *d_t = d_dir + 3 * n, *d_u = d_t + n, *d_v = d_u + n;
    uint32_t *d_prim = (uint32_t *)(d_v + n);
    HIPCHK(hipMemcpyAsync(d_org, org, b_ray, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_dir, dir, b_ray, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_t, t, b_d, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_u, u, b_d, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_v, v, b_d, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemcpyAsync(d_prim, prim, sizeof(uint32_t) * n, hipMemcpyHostToDevice, a->stream));
    HIPCHK(hipMemsetAsync(d_state, 0, b_state, a->stream));
    if (lh_accel_state_build_device(a, n, d_org, d_dir, d_prim, d_t, d_u, d_v, d_state, a->stream) != 0) return -1;
    HIPCHK(hipMemcpyAsync(state, d_state, b_state, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return 0;
}

/* ------------------------------------------------------------------------ */
/* materials / environment of the path tracer                               */
/* ------------------------------------------------------------------------ */
extern "C" int lh_accel_set_material(lh_accel_t *a, uint32_t mesh, const lh_material_t *mat)
{
    lh_guard guard(a);
    if (!a || !mat) return fail("lh_accel_set_material: NULL argument");
    const uint32_t nm = a->committed ? a->hs->nmeshes : a->nmeshes;
    if (mesh != LH_ALL_MESHES && mesh >= nm) return fail("lh_accel_set_material: mesh %u out of range", mesh);
    for (int k = 0; k < 3; k++) {
        if (!(mat->kd[k] >= 0.0f && mat->ks[k] >= 0.0f && mat->kt[k] >= 0.0f)) return fail("lh_accel_set_material: negative or NaN reflectance");
    }
    const double sum = (mat->kd[0] + mat->kd[1] + mat->kd[2] + mat->ks[0] + mat->ks[1] + mat->ks[2] + mat->kt[0] + mat->kt[1] + mat->kt[2]) / 3.0;
    if (sum > 1.0 + 1e-6) return fail("lh_accel_set_material: kd + ks + kt averages exceed 1 (pathtrace.c:419 asserts d + s + t <= 1)");
    if (!(mat->ior > 0.0f)) return fail("lh_accel_set_material: ior must be positive");
    if (a->nmaterials < nm) {
        lh_material_t *nmats = (lh_material_t *)realloc(a->materials, sizeof(lh_material_t) * (nm ? nm : 1));
        if (!nmats) return fail("out of memory");
        for (uint32_t k = a->nmaterials; k < nm; k++) {        /* ri_material_new (material.c:20-40): kd 1, ks 0, kt 0, ior 1 */
            memset(&nmats[k], 0, sizeof(lh_material_t));
            nmats[k].kd[0] = nmats[k].kd[1] = nmats[k].kd[2] = 1.0f; nmats[k].ior = 1.0f;
        }
        a->materials = nmats; a->nmaterials = nm;
    }
    for (uint32_t k = 0; k < a->nmaterials; k++) if (mesh == LH_ALL_MESHES || mesh == k) a->materials[k] = *mat;
    a->materials_dirty = 1;
    return 0;
}

extern "C" int lh_accel_set_environment(lh_accel_t *a, const lh_environment_t *env)
{
    lh_guard guard(a);
    if (!a) return fail("lh_accel_set_environment: NULL argument");
    if (!a->committed) return fail("lh_accel_set_environment: accel not committed");
    if (!env) {                  /* back to the default: constant white, no map */
        if (a->d_env_map) { (void)hipFree(a->d_env_map); a->d_env_map = NULL; }
        memset(&a->env, 0, sizeof(a->env)); a->env_set = 0;
        return 0;
    }
    if (env->map_rgba && (env->width < 1 || env->height < 1)) return fail("lh_accel_set_environment: bad map size");
    HIPCHK(hipSetDevice(a->device));
    if (a->d_env_map) { (void)hipFree(a->d_env_map); a->d_env_map = NULL; }
    a->env = *env; a->env.map_rgba = NULL; a->env_set = 1;
    if (env->map_rgba) {
        const size_t b = sizeof(float) * 4 * (size_t)env->width * env->height;
        HIPCHK(hipMalloc(&a->d_env_map, b));
        HIPCHK(hipMemcpy(a->d_env_map, env->map_rgba, b, hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" size_t lh_pt_material_bytes(void);
extern "C" void lh_pt_material_pack(const lh_material_t *m, void *out);

static int sync_materials(lh_accel_t *a)
{
    const uint32_t nm = a->hs->nmeshes ? a->hs->nmeshes : 1;
    if (a->nmaterials < nm) {
        lh_material_t def; memset(&def, 0, sizeof(def)); def.kd[0] = def.kd[1] = def.kd[2] = 1.0f; def.ior = 1.0f;
        lh_material_t *nmats = (lh_material_t *)realloc(a->materials, sizeof(lh_material_t) * nm);
        if (!nmats) return fail("out of memory");
        for (uint32_t k = a->nmaterials; k < nm; k++) nmats[k] = def;
        a->materials = nmats; a->nmaterials = nm; a->materials_dirty = 1;
    }
    if (a->materials_dirty || !a->d_materials) {
        if (a->d_materials) { (void)hipFree(a->d_materials); a->d_materials = NULL; }
        const size_t mb = lh_pt_material_bytes();
        std::vector<char> packed(mb * a->nmaterials);
        for (uint32_t k = 0; k < a->nmaterials; k++) lh_pt_material_pack(&a->materials[k], packed.data() + mb * k);
        HIPCHK(hipMalloc(&a->d_materials, packed.size()));
        HIPCHK(hipMemcpy(a->d_materials, packed.data(), packed.size(), hipMemcpyHostToDevice));
        a->materials_dirty = 0;
    }
    if (!a->d_prim_mesh && a->hs->bvh.ntris) {
        HIPCHK(hipMalloc(&a->d_prim_mesh, sizeof(uint32_t) * (size_t)a->hs->bvh.ntris));
        HIPCHK(hipMemcpy(a->d_prim_mesh, a->hs->bvh.prim_geom, sizeof(uint32_t) * (size_t)a->hs->bvh.ntris, hipMemcpyHostToDevice));
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* wavefront path tracer (kernels: lh_render.hip, arithmetic: lh_pt.h)       */
/* ------------------------------------------------------------------------ */
extern "C" int lh_pt_launch_begin(const lh_camera_t *cam, int x0, int y0, int w, int h, int band_rows, int band_stride, int spp, int s0,
                                  unsigned long long seed, void *d_cam, uint32_t *d_counts, int ncounts, void *stream);
extern "C" size_t lh_pt_cam_bytes(void);
extern "C" int lh_pt_launch_shade(size_t n_max, const lh_dev_scene_t *sc, const double *d_nrm9, const double *d_col9,
                                  const uint32_t *d_prim_mesh, const void *d_materials, const lh_material_t *override_mat,
                                  const float env_rgb[3], const void *d_env_map, int env_w, int env_h, int ref_weights,
                                  int depth, int max_depth, unsigned long long seed, int s0, int spp, int x0, int y0, int w,
                                  int band_rows, int band_stride, int full_width, const void *d_cam, uint32_t *d_counts, const double *d_org, const double *d_dir, const uint32_t *d_prim,
                                  const double *d_t, const double *d_u, const double *d_v, const uint32_t *d_path_of,
                                  const float *d_thr, unsigned long long *d_accum, double *d_org2, double *d_dir2, uint32_t *d_path_of2,
                                  float *d_thr2, int ncus, void *stream);
extern "C" int lh_pt_launch_resolve(int w, int h, int band_rows, float inv_total_spp, unsigned long long *d_accum, float *d_rgb, void *stream);

/* the pass over a w-wide region of h lines = full bands of band_rows lines, band k starting at frame line y0 + k * band_stride
 * (an ordinary tile: band_rows = h) */
static int pt_tile(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int band_rows, int band_stride, int s0, int spp, int spp_total, int max_vertices,
                   const lh_material_t *override_mat, const float env_rgb[3], const void *d_env_map, int env_w, int env_h, int flags,
                   uint64_t seed, void *d_rgb, lh_pt_stats_t *stats, void *stream)
{
    if (w <= 0 || h <= 0 || spp < 1 || spp_total < spp || max_vertices < 2 || max_vertices > 65536 || band_rows < 1 || h % band_rows != 0)
        return fail("lh_render_pt_tile: bad arguments");
    HIPCHK(hipSetDevice(a->device));
    if (sync_materials(a) != 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    const size_t S = (size_t)w * h * spp;
    if (S > ((size_t)1 << 30)) return fail("lh_render_pt_tile: more than 2^30 paths in one pass; lower spp_count or the tile size");
    if (spp > 4096) return fail("lh_render_pt_tile: more than 4096 samples of a pixel in one pass (the pixels' fixed-point sums); lower spp_count");
    unsigned long long *cnt = (a->stat_on && a->hs->bvh.ntris) ? a->d_counters : NULL;      /* lh_accel_trace_statistics */
    if (cnt) HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * LH_CNT_DEV, s));
    const int nbounce = max_vertices - 1;                  /* bounce d traces the ray to path vertex d + 2; the last vertex scatters nothing */
    if (ensure_buf(&a->r_org, S * 24) || ensure_buf(&a->r_dir, S * 24) || ensure_buf(&a->p_org2, S * 24) ||
        ensure_buf(&a->p_dir2, S * 24) || ensure_buf(&a->r_prim, S * 4) || ensure_buf(&a->r_t, S * 8) ||
        ensure_buf(&a->r_u, S * 8) || ensure_buf(&a->r_v, S * 8) || ensure_buf(&a->p_path, S * 4) ||
        ensure_buf(&a->p_path2, S * 4) || ensure_buf(&a->p_thr, S * 12) || ensure_buf(&a->p_thr2, S * 12) ||
        ensure_buf(&a->p_rad, (size_t)w * h * 24) || ensure_buf(&a->p_counts, ((size_t)nbounce + 2) * 4 + 64 + lh_pt_cam_bytes())) return -1;
    /* the chain's ray / path-word / throughput records alternate between two sets; bounce 0 reads none (its rays are the camera
     * rays, generated inside the closest-hit kernel and again by the shading pass for the paths that go on) */
    double *org = NULL, *dir = NULL, *org2 = (double *)a->r_org.p, *dir2 = (double *)a->r_dir.p;
    uint32_t *path = NULL, *path2 = (uint32_t *)a->p_path.p, *counts = (uint32_t *)a->p_counts.p;
    float *thr = NULL, *thr2 = (float *)a->p_thr.p;
    void *d_cam = (char *)a->p_counts.p + (((size_t)nbounce + 2) * 4 + 63) / 64 * 64;
    /* counts[d] = rays of bounce d: [0] = S here, [d + 1] accumulated by bounce d's shading pass.  The host never reads a count
     * inside the chain (every launch gets S as its upper bound and the count's address) -- except every 8th bounce of a long
     * chain (a furnace test's 400 vertices), to stop once every path has ended */
    /* the pixels' radiance sums (three 64-bit fixed-point words each, lh_render.hip pt_accumulate): the resolve leaves them zero, but
     * the buffer may be new, grown, or left by a pass that failed */
    HIPCHK(hipMemsetAsync(a->p_rad.p, 0, (size_t)w * h * 24, s));
    if (lh_pt_launch_begin(cam, x0, y0, w, h, band_rows, band_stride, spp, s0, seed, d_cam, counts, nbounce + 2, s) != 0) return fail("pt begin launch failed");
    int rc = 0;
    for (int depth = 0; depth < nbounce && rc == 0; depth++) {
        a->dev.n_dev = counts + depth; a->dev.cam_src = depth == 0 ? d_cam : NULL;
        rc = lh_launch(a, S, org, dir, a->r_prim.p, a->r_t.p, a->r_u.p, a->r_v.p, NULL, LH_MODE_CLOSEST, LH_VARIANT_SPEC, cnt, s, false);
        a->dev.n_dev = NULL; a->dev.cam_src = NULL;
        if (rc != 0) break;
        if (lh_pt_launch_shade(S, &a->dev, (const double *)a->d_nrm9, (const double *)a->d_attr9[0], (const uint32_t *)a->d_prim_mesh,
                               a->d_materials, override_mat, env_rgb, d_env_map, env_w, env_h, (flags & LH_PT_REFERENCE_WEIGHTS) != 0,
                               depth, max_vertices, seed, s0, spp, x0, y0, w, band_rows, band_stride, cam->width, d_cam, counts, org, dir, (const uint32_t *)a->r_prim.p,
                               (const double *)a->r_t.p, (const double *)a->r_u.p, (const double *)a->r_v.p, path, thr, (unsigned long long *)a->p_rad.p,
                               org2, dir2, path2, thr2, a->ncus, s) != 0)
            return fail("pt shade launch failed: %s", hipGetErrorString(hipGetLastError()));
        if (depth == 0) {
            org = org2; dir = dir2; path = path2; thr = thr2;
            org2 = (double *)a->p_org2.p; dir2 = (double *)a->p_dir2.p; path2 = (uint32_t *)a->p_path2.p; thr2 = (float *)a->p_thr2.p;
        } else {
            { double *t1 = org; org = org2; org2 = t1; t1 = dir; dir = dir2; dir2 = t1; }
            { uint32_t *t2 = path; path = path2; path2 = t2; float *t3 = thr; thr = thr2; thr2 = t3; }
        }
        if ((depth & 7) == 7 && depth + 1 < nbounce) {
            uint32_t left = 0;
            HIPCHK(hipMemcpyAsync(&left, counts + depth + 1, sizeof(left), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (left == 0) break;
        }
    }
    if (rc != 0) return -1;
    if (lh_pt_launch_resolve(w, h, band_rows, 1.0f / (float)spp_total, (unsigned long long *)a->p_rad.p, (float *)d_rgb, s) != 0)
        return fail("pt resolve launch failed");
    std::vector<uint32_t> hcounts((size_t)nbounce + 1);
    HIPCHK(hipMemcpyAsync(hcounts.data(), counts, hcounts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (cnt) {
        unsigned long long hc[LH_CNT_N];
        HIPCHK(hipMemcpy(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost));
        a->stat[0] += hc[LH_CNT_NODES]; a->stat[1] += hc[LH_CNT_TRIS]; a->stat[2] += hc[LH_CNT_EXACT]; a->stat[3] += hc[LH_CNT_RAYS];
    }
    if (stats) {
        uint64_t rays = 0, depth = 0;
        for (int d = 0; d < nbounce; d++) { rays += hcounts[d]; if (hcounts[d]) depth = (uint64_t)d + 1; }
        stats->paths = S; stats->rays = rays; stats->max_depth_reached = depth;
    }
    return 0;
}

/* round-1 entry point: one diffuse reflectance for every mesh, constant environment */
extern "C" int lh_render_pt_tile(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int s0, int spp,
                                 int spp_total, int max_vertices, float kd, const float env[3], uint64_t seed,
                                 void *d_rgb, lh_pt_stats_t *stats, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_pt_tile: accel not committed");
    if (!cam || !d_rgb || !env) return fail("lh_render_pt_tile: NULL argument");
    if (!(kd > 0.0f) || kd > 1.0f) return fail("lh_render_pt_tile: bad arguments");
    lh_material_t m; memset(&m, 0, sizeof(m)); m.kd[0] = m.kd[1] = m.kd[2] = kd; m.ior = 1.0f;
    return pt_tile(a, cam, x0, y0, w, h, h, 0, s0, spp, spp_total, max_vertices, &m, env, NULL, 0, 0, 0, seed, d_rgb, stats, stream);
}

/* per-mesh materials (lh_accel_set_material) and the accelerator's environment (lh_accel_set_environment) */
extern "C" int lh_render_pt_tile2(lh_accel_t *a, const lh_camera_t *cam, int x0, int y0, int w, int h, int s0, int spp,
                                  int spp_total, int max_vertices, int flags, uint64_t seed, void *d_rgb, lh_pt_stats_t *stats, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_pt_tile2: accel not committed");
    if (!cam || !d_rgb) return fail("lh_render_pt_tile2: NULL argument");
    float one[3] = {1.0f, 1.0f, 1.0f};
    const float *rgb = a->env_set ? a->env.rgb : one;          /* never set: constant white; an explicit black environment stays black */
    return pt_tile(a, cam, x0, y0, w, h, h, 0, s0, spp, spp_total, max_vertices, NULL, rgb, a->d_env_map, a->env.width, a->env.height, flags,
                   seed, d_rgb, stats, stream);
}

/* a rank's interleaved full-width bands of a sharded frame as ONE pass: nbands bands of band_rows lines, band k starting at
 * frame line y0_first + k * band_stride (all inside the frame).  d_rgb: [nbands][band_rows][width][3], every band in image
 * orientation (lh_render_ao_bands' layout).  override_mat NULL: the accelerator's per-mesh materials. */
extern "C" int lh_render_pt_bands(lh_accel_t *a, const lh_camera_t *cam, int y0_first, int band_rows, int band_stride, int nbands,
                                  int s0, int spp, int spp_total, int max_vertices, int flags, const lh_material_t *override_mat,
                                  uint64_t seed, void *d_rgb, lh_pt_stats_t *stats, void *stream)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_pt_bands: accel not committed");
    if (!cam || !d_rgb) return fail("lh_render_pt_bands: NULL argument");
    if (nbands < 1 || band_rows < 1 || y0_first < 0 || (nbands > 1 && band_stride < band_rows) ||
        (long long)y0_first + (long long)(nbands - 1) * band_stride + band_rows > cam->height)
        return fail("lh_render_pt_bands: bands must be disjoint and inside the frame");
    float one[3] = {1.0f, 1.0f, 1.0f};
    const float *rgb = a->env_set ? a->env.rgb : one;          /* never set: constant white; an explicit black environment stays black */
    return pt_tile(a, cam, 0, y0_first, cam->width, nbands * band_rows, band_rows, band_stride, s0, spp, spp_total, max_vertices, override_mat, rgb,
                   a->d_env_map, a->env.width, a->env.height, flags, seed, d_rgb, stats, stream);
}

/* ------------------------------------------------------------------------ */
/* whole frame into host memory (render_frame_controller + bucket_write)     */
/* ------------------------------------------------------------------------ */

extern "C" int lh_render_ao_frame_host(lh_accel_t *a, const lh_camera_t *cam, int ps, int gather_nsamples,
                                       uint64_t seed, int tile, float *rgb, lh_tile_stats_t *stats)
{
    lh_guard guard(a);
    if (!a || !a->committed) return fail("lh_render_ao_frame_host: accel not committed");
    if (!cam || !rgb) return fail("lh_render_ao_frame_host: NULL argument");
    if (cam->width <= 0 || cam->height <= 0) return fail("lh_render_ao_frame_host: bad resolution");
    if (tile <= 0) {
        /* default tile: the largest power of two (<= 4096) whose scratch stays under ~6 GB.  Per sub-sample: ~200 B of ray /
         * hit / epilogue records, plus 49 B per AO ray when the AO stage has to materialise its rays (LH_AO_FUSED=0):
         * ambient_occlusion.rib's own 3 x 3 pixel samples x 64 AO rays would otherwise ask for 30 GB per 1024^2 tile */
        const int N = gather_nsamples > 0 ? gather_nsamples : 1;
        const double per_pixel = (double)(ps > 0 ? ps : 1) * (ps > 0 ? ps : 1) * (200.0 + (a->ao_fused ? 0.0 : 49.0 * N));
        tile = 4096;
        while (tile > 64 && (double)tile * tile * per_pixel > 6.0e9) tile /= 2;
    }
    HIPCHK(hipSetDevice(a->device));
    const int W = cam->width, H = cam->height;
    if (ensure_buf(&a->r_frame, (size_t)tile * tile * 3 * sizeof(float))) return -1;
    std::vector<float> host((size_t)tile * tile * 3);
    lh_tile_stats_t tot = {0, 0, 0, 0};
    for (int y0 = 0; y0 < H; y0 += tile)
        for (int x0 = 0; x0 < W; x0 += tile) {
            const int w = (x0 + tile <= W) ? tile : W - x0, h = (y0 + tile <= H) ? tile : H - y0;
            lh_tile_stats_t st;
            if (lh_render_ao_tile(a, cam, x0, y0, w, h, ps, gather_nsamples, seed, NULL, a->r_frame.p, &st, a->stream) != 0) return -1;
            HIPCHK(hipMemcpyAsync(host.data(), a->r_frame.p, (size_t)w * h * 3 * sizeof(float), hipMemcpyDeviceToHost, a->stream));
            HIPCHK(hipStreamSynchronize(a->stream));
            /* the tile comes back with its rows already flipped (row 0 = pixel row y0+h-1) */
            for (int r = 0; r < h; r++)
                memcpy(rgb + ((size_t)(H - (y0 + h) + r) * W + x0) * 3, host.data() + (size_t)r * w * 3, (size_t)w * 3 * sizeof(float));
            tot.primary_rays += st.primary_rays; tot.primary_hits += st.primary_hits;
            tot.ao_rays += st.ao_rays; tot.ao_occluded += st.ao_occluded;
        }
    if (stats) *stats = tot;
    return 0;
}
