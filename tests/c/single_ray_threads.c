/*
 * single_ray_threads.c -- lucille's render threads in miniature (render.c:1043-1105): T pthreads, each calling
 * accel->intersect for ONE ray at a time (raytrace.c:31-69 -> ri_hipbvh_intersect -> lh_accel_intersect1) over its share of a
 * ray file.  Test infrastructure: writes every record it got back and prints the rays/s, with the coalescing path on and off.
 *
 * usage: single_ray_threads <in.bin> <out.bin> <threads> <combine 0|1> [host_walk 0|1 (default 0: the device path)]
 * in : u32 npos; double pos[npos][3]; u32 nidx; u32 idx[nidx]; u32 nrays; double org[nrays][3]; double dir[nrays][3]
 * out: u32 prim[nrays]; double t[nrays], u[nrays], v[nrays]; then double seconds; u64 launches, rays (combine statistics)
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lucille_hip.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "single_ray_threads: %s failed (line %d): %s\n", #c, __LINE__, lh_last_error()); exit(2); } } while (0)

static lh_accel_t *acc; static uint32_t nrays; static double *org, *dir, *t, *u, *v; static uint32_t *prim; static int nthreads;

static void *worker(void *arg)
{
    const int k = (int)(size_t)arg; uint32_t i;
    for (i = (uint32_t)k; i < nrays; i += (uint32_t)nthreads)           /* interleaved: neighbouring rays are in flight together */
        if (lh_accel_intersect1(acc, org + 3 * i, dir + 3 * i, prim + i, t + i, u + i, v + i) < 0) { fprintf(stderr, "intersect1: %s\n", lh_last_error()); exit(3); }
    return NULL;
}

int main(int argc, char **argv)
{
    FILE *in, *out; uint32_t npos, nidx; double *pos; uint32_t *idx; pthread_t th[64]; struct timespec a, b; double secs; uint64_t st[2]; int k;
    if (argc != 5 && argc != 6) return 1;
    nthreads = atoi(argv[3]); if (nthreads < 1 || nthreads > 64) return 1;
    CHECK((in = fopen(argv[1], "rb")) != NULL);
    CHECK(fread(&npos, 4, 1, in) == 1); pos = (double *)malloc(24 * (size_t)npos); CHECK(fread(pos, 24, npos, in) == npos);
    CHECK(fread(&nidx, 4, 1, in) == 1); idx = (uint32_t *)malloc(4 * (size_t)nidx); CHECK(fread(idx, 4, nidx, in) == nidx);
    CHECK(fread(&nrays, 4, 1, in) == 1);
    org = (double *)malloc(24 * (size_t)nrays); dir = (double *)malloc(24 * (size_t)nrays);
    CHECK(fread(org, 24, nrays, in) == nrays); CHECK(fread(dir, 24, nrays, in) == nrays); fclose(in);
    prim = (uint32_t *)calloc(nrays, 4); t = (double *)calloc(nrays, 8); u = (double *)calloc(nrays, 8); v = (double *)calloc(nrays, 8);
    CHECK(lh_accel_create(&acc, 0) == 0);
    CHECK(lh_accel_add_mesh(acc, npos, pos, 24, nidx, idx) == 0);
    CHECK(lh_accel_commit(acc, 0) == 0);
    CHECK(lh_accel_set_param(acc, "combine", atoi(argv[4])) == 0);
    CHECK(lh_accel_set_param(acc, "host_walk", argc == 6 ? atoi(argv[5]) : 0) == 0);        /* lh_hostwalk.c: one ray on the calling thread */
    { uint32_t p; double tt, uu, vv; CHECK(lh_accel_intersect1(acc, org, dir, &p, &tt, &uu, &vv) >= 0); }      /* first-launch costs outside the timing */
    CHECK(lh_accel_combine_statistics(acc, st, 1) == 0);
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (k = 0; k < nthreads; k++) CHECK(pthread_create(&th[k], NULL, worker, (void *)(size_t)k) == 0);
    for (k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    secs = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
    CHECK(lh_accel_combine_statistics(acc, st, 0) == 0);
    CHECK((out = fopen(argv[2], "wb")) != NULL);
    fwrite(prim, 4, nrays, out); fwrite(t, 8, nrays, out); fwrite(u, 8, nrays, out); fwrite(v, 8, nrays, out);
    fwrite(&secs, 8, 1, out); fwrite(st, 8, 2, out); fclose(out);
    printf("%d threads, combine %s, host walk %s: %u rays in %.3f s = %.0f rays/s; %llu launches (%.2f rays each)\n", nthreads, argv[4], argc == 6 ? argv[5] : "0", nrays, secs, nrays / secs,
           (unsigned long long)st[0], st[0] ? (double)st[1] / (double)st[0] : 0.0);
    lh_accel_destroy(acc);
    return 0;
}
