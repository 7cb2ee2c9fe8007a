/*
 * ri_render_hip.c -- the BATCHED frame loop a lucille maintainer adds to src/render/render.c so that, with the
 * MI355X accelerator bound (Option "raytrace" "accel_method" ["hip"]), a frame is rendered by the device tile
 * pipeline instead of one synchronous GPU launch per ri_raytrace() call.
 *
 * In lucille this is an edit INSIDE render.c (it needs that file's private bucket_t and bucket_write): the frame
 * controller render_frame_controller (render.c:1168-1207) gets a GPU branch in front of its thread launch.  To
 * build it here without touching the reference, this translation unit pulls render.c in unchanged from the
 * include path and hooks the branch in through the bucket queue (see "The hook" below); oracle/Makefile compiles
 * this file IN PLACE OF render.c for _ref/liblucille_ref_hip.so.  INTEGRATION.md section 3 shows the same change
 * as a patch to render.c.
 *
 * What the GPU branch replaces (all per-pixel work of a frame):
 *   render_bucket -> subsample -> ri_camera_get_pos_and_dir -> ri_transport_ambientocclusion -> ri_raytrace x (1 + N)
 *   (render.c:1107-1166, 715-823; ambientocclusion.c:332-415, 42-151)
 * by lh_render_ao_frame_host / lh_render_ao_tile_host (camera rays, closest hits, ri_intersection_state_build
 * subset, AO rays, any-hit, radiance, bucket_write's row order -- all on the device).  What it keeps: the bucket
 * queue (create_bucket_list's spiral order), bucket_write and the display driver: every bucket is handed to the
 * reference's own bucket_write, so drivers see the pixels in the order and format the CPU path delivers them.
 *
 * RI_HIP_RENDER selects the mode:
 *   "batched" (default)  whole frame on the device with the built-in counter-based sample stream;
 *   "replay"             bucket by bucket in queue order, AO samples drawn from the reference's own randomMT2(0)
 *                        stream exactly as its single-threaded render consumes it: reproduces the CPU frame
 *                        (up to the <= 20 / 65 536 pixels where device and glibc sin/cos differ);
 *   "rays"               the CPU controller: every ray through ri_raytrace -> accel->intersect (one launch per ray).
 * Anything the tile pipeline does not cover (sunsky light, xsamples != ysamples, a material texture on a geom)
 * falls back to the CPU controller, which still traces through the HIP accelerator.
 */
#include <stdint.h>
#include <pthread.h>
#include "queue.h"

/* The hook.  In lucille the GPU branch below sits at the top of render_frame_controller (render.c:1168-1207);
 * render.c is read-only here and that function is static, so this build interposes on the ONE place the frame's
 * worker threads take work -- ri_mt_queue_pop in render_bucket_thread_func (render.c:1057-1061): the first pop
 * of a frame (the bucket queue is still full) renders the whole frame on the device and drains the queue, after
 * which every worker finds it empty and returns; if the GPU branch declines, the pops go through unchanged and
 * the reference's own threads render the frame (still tracing through the HIP accelerator, one ray per call). */
static int hip_queue_pop(ri_mt_queue_t *queue, void **data, uint32_t *size);
#define ri_mt_queue_pop hip_queue_pop
#include "render.c"                 /* lucille's own src/render/render.c, found through -I src/render */
#undef ri_mt_queue_pop

#include "random.h"
#include "material.h"
#include "lucille_hip.h"

extern int         ri_hipbvh_intersect(void *accel, ri_ray_t *ray, ri_intersection_state_t *state, void *user);
extern lh_accel_t *ri_hipbvh_handle(void *accel);                         /* integration/ri_accel_hip.c */

static int hip_frame_supported(ri_render_t *render, int *ps_out)
{
    ri_display_t *disp = ri_option_get_curr_display(render->context->option);
    ri_list_t    *itr;
    const int xs = (int)disp->sampling_rates[0], ys = (int)disp->sampling_rates[1];
    if (!render->scene || !render->scene->accel || !render->scene->accel->data) return 0;
    if (render->scene->sunsky_light) return 0;                               /* gather_sunsky (ambientocclusion.c:369-374) */
    if (xs != ys || xs < 1) return 0;
    for (itr = ri_list_first(render->scene->geom_list); itr != NULL; itr = ri_list_next(itr)) {
        ri_geom_t *g = (ri_geom_t *)itr->data;
        if (g->material && g->material->texture) return 0;                   /* texture multiply (ambientocclusion.c:393-401) */
    }
    *ps_out = xs;
    return 1;
}

static void hip_camera(lh_camera_t *c, const ri_camera_t *cam)
{
    int i, j;
    c->width = cam->horizontal_resolution; c->height = cam->vertical_resolution;
    c->rh = cam->is_rh; c->ortho = (cam->camera_projection == RI_ORTHOGRAPHIC);
    c->flength = cam->flength;
    for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) c->cam2world[4 * i + j] = cam->camera_to_world.f[i][j];
}

/* one bucket's pixels out of `img` (rows in image orientation: row 0 = top) through the reference's bucket_write */
static void hip_emit_bucket(bucket_t *bucket, const float *img, int img_w, int img_x0, int img_top_row_of_bucket)
{
    const int w = bucket->w, h = bucket->h;
    int sx, sy;
    bucket->pixels = (ri_vector_t *)ri_mem_alloc_aligned(sizeof(ri_vector_t) * w * h, 32);
    for (sy = 0; sy < h; sy++)
        for (sx = 0; sx < w; sx++) {
            /* bucket row sy is screen line y + sy; the image holds it (h - 1 - sy) rows below the bucket's top row */
            const float *p = img + 3 * ((size_t)(img_top_row_of_bucket + (h - 1 - sy)) * img_w + (img_x0 + sx));
            ri_vector_t *o = &bucket->pixels[sy * w + sx];
            (*o)[0] = p[0]; (*o)[1] = p[1]; (*o)[2] = p[2]; (*o)[3] = 0.0;
        }
    ri_mutex_lock(ri_render_get()->mutex);
    bucket_write(bucket, ri_render_get()->display_drv, ri_option_get_curr_display(ri_render_get()->context->option));
    ri_mutex_unlock(ri_render_get()->mutex);
    ri_mem_free_aligned(bucket->pixels);
    bucket->pixels = NULL;
}

/* 1: the frame was rendered on the device and the queue is drained; 0: declined, the CPU threads do it */
static int hip_render_frame(ri_render_t *render)
{
    const char *mode = getenv("RI_HIP_RENDER");
    lh_accel_t *lh; lh_camera_t cam; lh_tile_stats_t st;
    int ps = 1, N, W, H, ret;
    bucket_t *bucket; uint32_t data_size;

    if (!render->scene || !render->scene->accel || render->scene->accel->intersect == NULL ||
        (mode && strcmp(mode, "rays") == 0) || !hip_frame_supported(render, &ps) ||
        (lh = ri_hipbvh_handle(render->scene->accel->data)) == NULL)
        return 0;
    hip_camera(&cam, render->context->option->camera);
    W = cam.width; H = cam.height;
    N = render->context->option->gather_nsamples;

    if (mode && strcmp(mode, "replay") == 0) {
        /* the reference's deterministic single-thread frame: buckets in queue order, each fed the next numbers of
         * thread 0's MT19937 stream; a bucket consumes 2 N per primary hit (calculate_occlusion draws z0, z1 per ray) */
        const int nphi = (int)sqrt((double)N), NN = nphi * nphi;
        double *fifo = NULL; size_t have = 0, cap = 0;
        float *rgb = NULL; size_t rgb_cap = 0;
        while ((ri_mt_queue_pop)(render->bucket_queue, (void **)&bucket, &data_size) == 0) {
            const size_t need = (size_t)2 * NN * bucket->w * bucket->h * ps * ps;
            if (need > cap) { fifo = (double *)realloc(fifo, sizeof(double) * need); cap = need; }
            while (have < need) fifo[have++] = randomMT2(0);
            if ((size_t)bucket->w * bucket->h * 3 > rgb_cap) { rgb_cap = (size_t)bucket->w * bucket->h * 3; rgb = (float *)realloc(rgb, sizeof(float) * rgb_cap); }
            ret = lh_render_ao_tile_host(lh, &cam, bucket->x, bucket->y, bucket->w, bucket->h, ps, N, 0, fifo, have, rgb, &st);
            if (ret != 0) {       /* put nothing back: this bucket stays black, the rest go to the CPU threads */
                ri_log(LOG_ERROR, "(HIPBVH) %s", lh_last_error());
                free(fifo); free(rgb);
                return 0;
            }
            {   /* keep what the bucket did not consume for the next one */
                const size_t used = (size_t)2 * NN * st.primary_hits;
                memmove(fifo, fifo + used, sizeof(double) * (have - used));
                have -= used;
            }
            hip_emit_bucket(bucket, rgb, bucket->w, 0, 0);
        }
        free(fifo); free(rgb);
        return 1;
    }

    {   /* batched: the frame on the device (tiles sized by its scratch budget), then bucket by bucket to the display */
        float *img = (float *)malloc(sizeof(float) * 3 * (size_t)W * H);
        if (!img) return 0;
        ret = lh_render_ao_frame_host(lh, &cam, ps, N, 1, 0, img, &st);
        if (ret != 0) {
            ri_log(LOG_ERROR, "(HIPBVH) %s -- falling back to the one-ray path", lh_last_error());
            free(img);
            return 0;
        }
        while ((ri_mt_queue_pop)(render->bucket_queue, (void **)&bucket, &data_size) == 0)
            hip_emit_bucket(bucket, img, W, bucket->x, H - (bucket->y + bucket->h));
        free(img);
        ri_log(LOG_INFO, "(HIPBVH) frame on the device: %llu primary + %llu AO rays",
               (unsigned long long)st.primary_rays, (unsigned long long)st.ao_rays);
    }
    return 1;
}

static pthread_mutex_t g_hip_frame_mu = PTHREAD_MUTEX_INITIALIZER;
static int             g_hip_declined = 0;

static int hip_queue_pop(ri_mt_queue_t *queue, void **data, uint32_t *size)
{
    ri_render_t *render = ri_render_get();
    if (render && queue == render->bucket_queue) {
        pthread_mutex_lock(&g_hip_frame_mu);
        if (ri_mt_queue_len(queue) != render->nbuckets) g_hip_declined = 0;      /* the frame is under way */
        else if (!g_hip_declined && !hip_render_frame(render)) g_hip_declined = 1; /* first pop of a frame */
        pthread_mutex_unlock(&g_hip_frame_mu);
    }
    return (ri_mt_queue_pop)(queue, data, size);
}
