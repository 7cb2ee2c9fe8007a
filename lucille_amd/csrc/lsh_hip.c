/*
 * lsh_hip -- `lsh RIBFILE` on the MI355X path: read a RIB (subset, lh_rib.c), build the HIP
 * accelerator over its geometry, render the frame the way lucille's renderer does (the AO
 * transport is what render.c:803 calls for every sample) and write the Display's .hdr.
 *
 * Mirrors the reference shell's command line (src/lsh/main.c:406-433): --output, --nthreads
 * (accepted, ignored: there are no render threads), --pixelsamples, --verbose, --help; plus
 * --resolution WxH, --gather N, --device N, --seed N which the reference sets through RIB only.
 * Plain C on the C ABI of include/lucille_hip.h; no CPU fallback: without a GPU it fails.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lucille_hip.h"

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static void usage(void)
{
    printf("lucille renderer shell, HIP (MI355X) ray-query path.\n\n  Usage: lsh_hip [OPTIONS] RIBFILE\n\n  [OPTIONS]\n\n"
           "    --help            Print this help.\n"
           "    --output     NAME Specify output name (.hdr).\n"
           "    --verbose         Verbose mode.\n"
           "    --nthreads     N  Accepted for compatibility, ignored.\n"
           "    --pixelsamples N  Samples per pixel axis (N x N).\n"
           "    --resolution WxH  Override Format.\n"
           "    --gather       N  Override Option \"gather\" \"nsamples\" (AO rays per hit).\n"
           "    --device       N  HIP device ordinal.\n"
           "    --gpus         N  Render on N GPUs of this node (one host build, replicated; tile queue; slabs\n"
           "                      gathered on the first device).  --devices a,b,.. picks them (repeats allowed).\n"
           "    --tile         N  Tile edge in pixels of the multi-GPU tile queue (default 256).\n"
           "    --rank R --world N --rendezvous PATH\n"
           "                      One process per GPU (lh_dist_*: RCCL over xGMI): start N copies of this command line, rank\n"
           "                      0 .. N-1, with the same fresh PATH on a shared file system.  Rank 0 builds the scene and\n"
           "                      broadcasts it, every rank renders its bands of the frame, rank 0 gathers and writes the\n"
           "                      .hdr.  --device defaults to the rank.  --bandrows N: lines per band (default 4).\n"
           "    --build host|device|auto  Where the traversal tree is built.  host: binned SAH on the CPU cores (best frame\n"
           "                      time).  device: both trees on the GPU -- Morton clusters + binned SAH for traversal,\n"
           "                      lucille's own tree level by level (21 M triangles in 0.3 s instead of 4 s; frames\n"
           "                      ~2 %% slower).  auto (default): device from 1 M triangles on -- this program renders\n"
           "                      one frame per scene set-up, like the reference's lsh.\n"
           "    --seed         N  Seed of the AO sample stream.\n"
           "    --parse-only      Read the RIB, print what was found, do not render.\n\n");
}

int main(int argc, char **argv)
{
    const char *rib = NULL, *output = NULL; int i, verbose = 0, ps = -1, gather = -1, device = 0, parse_only = 0, W = -1, H = -1;
    unsigned long long seed = 1; int build = -1, build_threads = 0;      /* build: -1 auto, 0 host, 1 device */
    int ngpus = 1, devices[64], ndev_listed = 0, tile = 256; lh_multi_t *multi = NULL; double dev_secs[64];
    int rank = -1, world = 0, band_rows = 4, device_given = 0; const char *rendezvous = NULL; lh_dist_t *dist = NULL;
    lh_rib_scene_t *scene = NULL; lh_rib_info_t info; lh_accel_t *accel = NULL; lh_accel_info_t ai; lh_tile_stats_t st;
    float *rgb; double t0, t1, t2, t3;

    for (i = 1; i < argc; i++) {
        const char *a = argv[i];
        while (*a == '-') a++;                    /* --opt and -opt spellings alike */
        if (argv[i][0] != '-') { rib = argv[i]; continue; }
        if (strcmp(a, "help") == 0) { usage(); return 0; }
        else if (strcmp(a, "verbose") == 0) verbose = 1;
        else if (strcmp(a, "parse-only") == 0) parse_only = 1;
        else if (strcmp(a, "recover") == 0 || strcmp(a, "progress") == 0 || strcmp(a, "debug") == 0) ;   /* main.c:292-298 */
        else if (i + 1 < argc && strcmp(a, "output") == 0) output = argv[++i];
        else if (i + 1 < argc && strcmp(a, "nthreads") == 0) ++i;
        else if (i + 1 < argc && strcmp(a, "maxraydepth") == 0) ++i;
        else if (i + 1 < argc && strcmp(a, "pixelsamples") == 0) ps = atoi(argv[++i]);
        else if (i + 1 < argc && strcmp(a, "gather") == 0) gather = atoi(argv[++i]);
        else if (i + 1 < argc && strcmp(a, "device") == 0) { device = atoi(argv[++i]); device_given = 1; }
        else if (i + 1 < argc && strcmp(a, "rank") == 0) rank = atoi(argv[++i]);
        else if (i + 1 < argc && strcmp(a, "world") == 0) world = atoi(argv[++i]);
        else if (i + 1 < argc && strcmp(a, "rendezvous") == 0) rendezvous = argv[++i];
        else if (i + 1 < argc && strcmp(a, "bandrows") == 0) band_rows = atoi(argv[++i]);
        else if (i + 1 < argc && strcmp(a, "gpus") == 0) ngpus = atoi(argv[++i]);
        else if (i + 1 < argc && strcmp(a, "tile") == 0) tile = atoi(argv[++i]);
        else if (i + 1 < argc && strcmp(a, "devices") == 0) {
            char *p = argv[++i];
            while (*p && ndev_listed < 64) { devices[ndev_listed++] = (int)strtol(p, &p, 10); if (*p == ',') p++; }
        }
        else if (i + 1 < argc && strcmp(a, "seed") == 0) seed = strtoull(argv[++i], NULL, 10);
        else if (i + 1 < argc && strcmp(a, "build") == 0) {
            const char *v = argv[++i];
            build = strcmp(v, "host") == 0 ? 0 : strcmp(v, "device") == 0 ? 1 : strcmp(v, "auto") == 0 ? -1 : -2;
            if (build == -2) { usage(); return 1; }
        }
        else if (i + 1 < argc && strcmp(a, "resolution") == 0) { if (sscanf(argv[++i], "%dx%d", &W, &H) != 2) { usage(); return 1; } }
        else { fprintf(stderr, "lsh_hip: unknown option %s\n", argv[i]); usage(); return 1; }
    }
    if (!rib) { usage(); return 1; }
    if ((world > 0 || rank >= 0 || rendezvous) && !(world > 0 && rank >= 0 && rank < world && rendezvous)) {
        fprintf(stderr, "lsh_hip: --rank R --world N --rendezvous PATH go together (0 <= R < N)\n"); return 1;
    }
    if (world > 0 && !device_given) device = rank % (lh_device_count() > 0 ? lh_device_count() : 1);

    t0 = now_s();
    if (lh_rib_load(rib, &scene) != 0) { fprintf(stderr, "lsh_hip: %s\n", lh_rib_last_error()); return 1; }
    lh_rib_info(scene, &info);
    fputs(lh_rib_messages(scene), stdout);
    if (W > 0 && H > 0) { info.camera.width = W; info.camera.height = H; }
    if (ps < 1) ps = info.pixel_samples[0];
    if (info.pixel_samples[0] != info.pixel_samples[1] && verbose)
        fprintf(stderr, "lsh_hip: PixelSamples %d x %d: the tile pipeline samples N x N, using %d\n", info.pixel_samples[0], info.pixel_samples[1], ps);
    if (gather < 1) gather = info.gather_nsamples;
    if (!output) {                                /* not a file display: there is no window here, write <name>.hdr */
        if (strcmp(info.display_type, "file") != 0 && strcmp(info.display_type, "hdr") != 0) {
            char *ext = strrchr(info.display_name, '.');
            if (ext) *ext = 0;
            strcat(info.display_name, ".hdr");
        }
        output = info.display_name;
    }
    t1 = now_s();
    printf("[lucille_hip] RIB parsing : %.3f s  (%u geoms, %llu triangles, %u of %u requests skipped)\n", t1 - t0, info.nmeshes,
           (unsigned long long)info.ntriangles, info.nskipped, info.nrequests);
    if (!info.world_complete) printf("[lucille_hip] warning: no complete WorldBegin/WorldEnd block\n");
    if (parse_only) {
        printf("[lucille_hip] Format %d x %d, fov %g, orientation %s, PixelSamples %d, gather %d, Display \"%s\" \"%s\"\n", info.camera.width,
               info.camera.height, info.fov, info.camera.rh ? "rh" : "lh", ps, gather, info.display_name, info.display_type);
        lh_rib_free(scene); return 0;
    }

    if (build == 1 || (build == -1 && info.ntriangles >= 1000000ull)) build_threads = LH_BUILD_ON_DEVICE;
    if (ndev_listed > 0) ngpus = ndev_listed;
    if (world > 0) {
        /* one process per GPU (lh_dist_*: SURVEY 8e): rank 0 builds, everybody receives the scene image */
        if (lh_dist_init_file(&dist, rendezvous, rank, world, device) != 0 || lh_accel_create(&accel, device) != 0) {
            fprintf(stderr, "lsh_hip: rank %d: %s\n", rank, lh_last_error()); lh_rib_free(scene); return 1;
        }
        /* rank 0's commit may fail: it says why, and STILL enters the broadcast, whose first word tells the other ranks to give up
         * (they would wait in ncclBroadcast for ever otherwise) */
        if (rank == 0 && (lh_accel_add_rib_scene(accel, scene) != 0 || lh_accel_commit(accel, build_threads) != 0))
            fprintf(stderr, "lsh_hip: rank 0: %s\n", lh_last_error());
        if (lh_dist_broadcast_scene(dist, accel) != 0) {
            fprintf(stderr, "lsh_hip: rank %d: %s\n", rank, lh_last_error()); lh_rib_free(scene); lh_dist_destroy(dist); return 1;
        }
    } else if (ngpus > 1 || ndev_listed > 0) {
        /* the G GPUs of this node from this one process (lh_multi_*: SURVEY 8b(4), 8e) */
        if (lh_multi_create(&multi, ngpus, ndev_listed ? devices : NULL) != 0 || lh_multi_add_rib_scene(multi, scene) != 0 ||
            lh_multi_commit(multi, build_threads) != 0) {
            fprintf(stderr, "lsh_hip: %s\n", lh_last_error()); lh_rib_free(scene); return 1;
        }
        accel = lh_multi_accel(multi, 0);
    } else if (lh_accel_create(&accel, device) != 0 || lh_accel_add_rib_scene(accel, scene) != 0 || lh_accel_commit(accel, build_threads) != 0) {
        fprintf(stderr, "lsh_hip: %s\n", lh_last_error()); lh_rib_free(scene); return 1;
    }
    lh_accel_info(accel, &ai);
    t2 = now_s();
    printf("[lucille_hip] BVH building: %.3f s  (%u nodes, depth %u; %s%s)\n", t2 - t1, ai.nnodes_traversal, ai.max_depth,
           dist && rank != 0 ? "received from rank 0" : build_threads == LH_BUILD_ON_DEVICE ? "built on the device" : "built on the host",
           multi ? ", replicated" : dist ? (lh_dist_transport(dist) == LH_DIST_RCCL ? ", broadcast over RCCL" : ", broadcast through shared memory") : "");

    rgb = (float *)malloc(sizeof(float) * 3 * (size_t)info.camera.width * info.camera.height);
    if (!rgb) { fprintf(stderr, "lsh_hip: out of memory\n"); return 1; }
    if (dist) {
        if (lh_dist_render_ao_frame_host(dist, accel, &info.camera, ps, gather, seed, band_rows, rgb, &st) != 0) { fprintf(stderr, "lsh_hip: rank %d: %s\n", rank, lh_last_error()); return 1; }
    } else if (multi) {
        if (lh_multi_render_ao_frame_host(multi, &info.camera, ps, gather, seed, tile, rgb, &st, dev_secs) != 0) { fprintf(stderr, "lsh_hip: %s\n", lh_last_error()); return 1; }
        if (verbose) for (i = 0; i < lh_multi_ndevices(multi); i++) printf("[lucille_hip] replica %d busy %.3f s\n", i, dev_secs[i]);
    } else if (lh_render_ao_frame_host(accel, &info.camera, ps, gather, seed, 0, rgb, &st) != 0) { fprintf(stderr, "lsh_hip: %s\n", lh_last_error()); return 1; }
    t3 = now_s();
    printf("[lucille_hip] Rendering   : %.3f s  (%llu primary + %llu AO rays, %.1f Mrays/s)\n", t3 - t2, (unsigned long long)st.primary_rays,
           (unsigned long long)st.ao_rays, 1e-6 * (double)(st.primary_rays + st.ao_rays) / (t3 - t2));
    if (!dist || rank == 0) {                     /* rank 0 owns the display (render.c:468-514) */
        if (lh_hdr_write(output, info.camera.width, info.camera.height, rgb) != 0) { fprintf(stderr, "lsh_hip: %s\n", lh_rib_last_error()); return 1; }
        printf("[lucille_hip] (Disp) Output written to \"%s\"\n", output);
    }
    free(rgb);
    if (multi) lh_multi_destroy(multi); else lh_accel_destroy(accel);
    if (dist) lh_dist_destroy(dist);
    lh_rib_free(scene);
    return 0;
}
