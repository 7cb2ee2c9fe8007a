/*
 * host_api_prog.c -- a lucille-style C program against include/lucille_accel.h, the way the
 * reference's testbed drives the accelerator (src/testbed/main.cpp:53-65, controller.cpp:155-273):
 * ri_geom_new / add_positions / add_indices -> ri_scene_add_geom -> ri_accel_bind ->
 * ri_scene_build_accel -> ri_raytrace per ray.  Test infrastructure: reads a scene + rays +
 * beams file written by tests/test_gpu_hostapi.py, writes every record it gets back; the
 * python side compares them with the oracle.
 *
 * usage: host_api_prog <in.bin> <out.bin>
 * in : u32 nmesh; per mesh {u32 npos; double pos[npos][4]; u32 nidx; u32 idx[nidx]; u32 has_normals;
 *      double nrm[npos][4] (if has_normals)}; u32 nrays; double org[nrays][3]; double dir[nrays][3];
 *      u32 nbeams; double borg[nbeams][3]; double bdir[nbeams][4][3]
 * out: per ray, single path then batch path: i32 hit, i32 geom ordinal, u32 index, double t,u,v,
 *      P[3], Ng[3], Ns[3], i32 inside; then per beam: i32 ri_beam_set rc, i32 class;
 *      then u64 stat[5]; then i32 bind_unknown_rc, i32 empty_scene_hit; then the 24 x 16 x 3 float tile;
 *      then u32 nraster and, per raster beam (the last accepted beams, at most 8), i32 beam ordinal + the 16 x 16 doubles of
 *      its raster plane's t (ri_raster_plane_setup + ri_hipbvh_intersect_beam)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lucille_accel.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "host_api_prog: %s failed (line %d)\n", #c, __LINE__); return 2; } } while (0)

static int geom_ordinal(const ri_scene_t *scene, const ri_geom_t *g)
{
    unsigned int i;
    for (i = 0; i < scene->ngeoms; i++) if (scene->geom_list[i] == g) return (int)i;
    return -1;
}

static void put_record(FILE *f, const ri_scene_t *scene, int hit, const ri_intersection_state_t *st)
{
    int32_t h = hit, go = hit ? geom_ordinal(scene, st->geom) : -1, inside = hit ? st->inside : 0;
    uint32_t index = hit ? st->index : 0;
    double d[12]; int k;
    memset(d, 0, sizeof(d));
    if (hit) {
        d[0] = st->t; d[1] = st->u; d[2] = st->v;
        for (k = 0; k < 3; k++) { d[3 + k] = st->P[k]; d[6 + k] = st->Ng[k]; d[9 + k] = st->Ns[k]; }
    }
    fwrite(&h, 4, 1, f); fwrite(&go, 4, 1, f); fwrite(&index, 4, 1, f); fwrite(d, 8, 12, f); fwrite(&inside, 4, 1, f);
}

int main(int argc, char **argv)
{
    FILE *in, *out; uint32_t nmesh, nrays, nbeams, m, i; int k;
    ri_render_t *render; ri_scene_t *scene; ri_geom_t **geoms;
    double *org, *dir, *borg, *bdir; ri_ray_t *rays; ri_intersection_state_t *states; int *hits;
    uint64_t stat[5]; int32_t rc_unknown, empty_hit;

    CHECK(argc == 3);
    in = fopen(argv[1], "rb"); CHECK(in != NULL);
    out = fopen(argv[2], "wb"); CHECK(out != NULL);

    ri_render_init();
    render = ri_render_get(); CHECK(render != NULL);
    scene = render->scene; CHECK(scene != NULL);

    /* an empty scene builds an accelerator that always misses (bvh.c:311-315,446-449) */
    {
        ri_ray_t ray; ri_intersection_state_t st;
        memset(&ray, 0, sizeof(ray)); memset(&st, 0, sizeof(st));
        ray.org[2] = -5.0; ray.dir[2] = 1.0; ray.dir[1] = 0.25;
        /* unknown method: -1 and the accelerator stays unbound (accel.c:102-106) */
        rc_unknown = ri_accel_bind(scene->accel, 77);
        CHECK(ri_accel_bind(scene->accel, RI_ACCEL_HIP) == 0);
        CHECK(ri_scene_build_accel(scene) == 0);
        empty_hit = ri_raytrace(render, &ray, &st);
    }

    CHECK(fread(&nmesh, 4, 1, in) == 1);
    geoms = (ri_geom_t **)calloc(nmesh ? nmesh : 1, sizeof(*geoms));
    for (m = 0; m < nmesh; m++) {
        uint32_t npos, nidx, has_n; ri_vector_t *pos, *nrm = NULL; unsigned int *idx;
        CHECK(fread(&npos, 4, 1, in) == 1);
        pos = (ri_vector_t *)malloc(sizeof(ri_vector_t) * npos); CHECK(fread(pos, sizeof(ri_vector_t), npos, in) == npos);
        CHECK(fread(&nidx, 4, 1, in) == 1);
        idx = (unsigned int *)malloc(4 * nidx); CHECK(fread(idx, 4, nidx, in) == nidx);
        CHECK(fread(&has_n, 4, 1, in) == 1);
        if (has_n) { nrm = (ri_vector_t *)malloc(sizeof(ri_vector_t) * npos); CHECK(fread(nrm, sizeof(ri_vector_t), npos, in) == npos); }
        geoms[m] = ri_geom_new();
        ri_geom_add_positions(geoms[m], npos, (const ri_vector_t *)pos);
        ri_geom_add_indices(geoms[m], nidx, idx);
        if (nrm) ri_geom_add_normals(geoms[m], npos, (const ri_vector_t *)nrm);
        ri_scene_add_geom(scene, geoms[m]);
        free(pos); free(idx); free(nrm);           /* the geom keeps copies (geom.c:78-99) */
    }
    CHECK(ri_scene_build_accel(scene) == 0);       /* rebuild, like ri_scene_setup every frame (scene.c:84-98) */

    CHECK(fread(&nrays, 4, 1, in) == 1);
    org = (double *)malloc(24 * (size_t)nrays); dir = (double *)malloc(24 * (size_t)nrays);
    CHECK(fread(org, 24, nrays, in) == nrays); CHECK(fread(dir, 24, nrays, in) == nrays);
    CHECK(fread(&nbeams, 4, 1, in) == 1);
    borg = (double *)malloc(24 * (size_t)nbeams + 8); bdir = (double *)malloc(96 * (size_t)nbeams + 8);
    CHECK(fread(borg, 24, nbeams, in) == nbeams); CHECK(fread(bdir, 96, nbeams, in) == nbeams);

    ri_hipbvh_trace_statistics(1);
    ri_hipbvh_clear_stat_traversal();

    /* 1. the reference's calling pattern: one ri_raytrace per ray */
    for (i = 0; i < nrays; i++) {
        ri_ray_t ray; ri_intersection_state_t st; int hit;
        memset(&ray, 0, sizeof(ray)); memset(&st, 0, sizeof(st));
        for (k = 0; k < 3; k++) { ray.org[k] = org[3 * i + k]; ray.dir[k] = dir[3 * i + k]; }
        hit = ri_raytrace(render, &ray, &st);
        put_record(out, scene, hit, &st);
    }
    CHECK(render->stat.nrays == (uint64_t)nrays + 1);      /* raytrace.c:43 (+1: the empty-scene ray) */

    /* 2. the batched path */
    rays = (ri_ray_t *)calloc(nrays ? nrays : 1, sizeof(*rays));
    states = (ri_intersection_state_t *)calloc(nrays ? nrays : 1, sizeof(*states));
    hits = (int *)calloc(nrays ? nrays : 1, sizeof(int));
    for (i = 0; i < nrays; i++)
        for (k = 0; k < 3; k++) { rays[i].org[k] = org[3 * i + k]; rays[i].dir[k] = dir[3 * i + k]; }
    CHECK(ri_raytrace_batch(render, nrays, rays, states, hits) >= 0);
    for (i = 0; i < nrays; i++) put_record(out, scene, hits[i], &states[i]);

    ri_hipbvh_get_stat_traversal(stat);
    ri_hipbvh_trace_statistics(0);

    /* 3. beams: ri_beam_set + ri_bvh_intersect_beam_visibility (testbed simplerender.cpp:566) */
    for (i = 0; i < nbeams; i++) {
        ri_beam_t beam; ri_vector_t o, d[4]; int32_t rc, cls = -1; int j;
        memset(o, 0, sizeof(o)); memset(d, 0, sizeof(d));
        for (k = 0; k < 3; k++) o[k] = borg[3 * i + k];
        for (j = 0; j < 4; j++) for (k = 0; k < 3; k++) d[j][k] = bdir[12 * i + 3 * j + k];
        rc = ri_beam_set(&beam, o, d);
        if (rc == 0) cls = ri_hipbvh_intersect_beam_visibility(scene->accel->data, &beam, NULL);
        fwrite(&rc, 4, 1, out); fwrite(&cls, 4, 1, out);
    }
    ri_hipbvh_invalidate_cache(scene->accel->data);

    fwrite(stat, 8, 5, out);
    fwrite(&rc_unknown, 4, 1, out); fwrite(&empty_hit, 4, 1, out);

    /* 4. the tile-level entry point: a 24 x 16 tile of a 32 x 32 ambient-occlusion frame, camera on the +z side looking at the soup */
    {
        ri_tile_camera_t cam; static float rgb[24 * 16 * 3];
        memset(&cam, 0, sizeof(cam));
        cam.width = 32; cam.height = 32; cam.rh = 1; cam.ortho = 0; cam.flength = 2.0;
        cam.cam2world[0] = cam.cam2world[5] = cam.cam2world[10] = cam.cam2world[15] = 1.0;
        cam.cam2world[12] = 0.5; cam.cam2world[13] = 0.5; cam.cam2world[14] = 3.0;
        CHECK(ri_render_tile_ao(scene->accel->data, &cam, 4, 8, 24, 16, 2, 16, 77ull, rgb) == 0);
        CHECK(ri_render_tile_ao(NULL, &cam, 4, 8, 24, 16, 2, 16, 77ull, rgb) == -1);
        fwrite(rgb, sizeof(float), 24 * 16 * 3, out);
    }
    /* 5. the beam-raster path (bvh.h:203-206): a 16 x 16 window per beam -- eye = the beam's origin, axis-aligned frame,
     * lower-left corner = the beam's first corner direction, 45 degrees */
    {
        ri_raster_plane_t *plane = ri_raster_plane_new(); uint32_t nraster = 0, done = 0; long pos;
        CHECK(plane != NULL);
        pos = ftell(out); fwrite(&nraster, 4, 1, out);
        uint32_t bi;
        for (bi = nbeams; bi > 0 && done < 8; bi--) {         /* from the end: the wide beams */
            ri_beam_t beam; ri_vector_t o, d[4], frame[3]; int32_t ord; int j;
            i = bi - 1; ord = (int32_t)i;
            memset(o, 0, sizeof(o)); memset(d, 0, sizeof(d)); memset(frame, 0, sizeof(frame));
            for (k = 0; k < 3; k++) o[k] = borg[3 * i + k];
            for (j = 0; j < 4; j++) for (k = 0; k < 3; k++) d[j][k] = bdir[12 * i + 3 * j + k];
            if (ri_beam_set(&beam, o, d) != 0) continue;
            frame[0][0] = 1.0; frame[1][1] = 1.0; frame[2][2] = 1.0;
            CHECK(ri_raster_plane_setup(plane, 16, 16, frame, d[0], o, 45.0) == 0);
            CHECK(ri_hipbvh_intersect_beam(scene->accel->data, &beam, plane, NULL) == 0);
            fwrite(&ord, 4, 1, out); fwrite(plane->t, 8, 256, out);
            done++;
        }
        nraster = done;
        fseek(out, pos, SEEK_SET); fwrite(&nraster, 4, 1, out); fseek(out, 0, SEEK_END);
        CHECK(ri_raster_plane_free(plane) == 0);
        free(plane);
    }
    ri_hipbvh_report_stat_traversal();

    fclose(in); fclose(out);
    free(org); free(dir); free(borg); free(bdir); free(rays); free(states); free(hits);
    ri_render_free();
    for (m = 0; m < nmesh; m++) ri_geom_free(geoms[m]);
    free(geoms);
    return 0;
}
