/*
 * lh_tpool.h -- a small fork-join thread pool for the host builders (lh_bvh.c, lh_refbvh.c): the passes over the long
 * primitive ranges at the top of a tree (bounds, binning, partition) run on it while the subtree tasks are collected.
 * pthreads, not OpenMP: libgomp may bind the calling thread to one core (OMP_PROC_BIND) and the subtree workers created
 * afterwards would inherit that mask (measured: the subtree phase 5x slower).
 */
#ifndef LH_TPOOL_H
#define LH_TPOOL_H

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

/* ranges at least this long are processed by the pool (LH_PAR_MIN_LOG2 overrides: tuning) */
static uint32_t lh_par_min(void)
{
    static uint32_t v = 0;
    if (!v) { const char *e = getenv("LH_PAR_MIN_LOG2"); int l = e ? atoi(e) : 17; if (l < 10) l = 10; if (l > 30) l = 30; v = 1u << l; }
    return v;
}
#define LH_PAR_MIN lh_par_min()
#define LH_POOL_MAX 64

typedef struct lh_tpool {
    int nt; pthread_t th[LH_POOL_MAX];
    pthread_mutex_t mu; pthread_cond_t go, done;
    void (*fn)(void *, int, int); void *arg;
    unsigned gen; int pending, stop;
} lh_tpool_t;
typedef struct { lh_tpool_t *p; int t; } tpool_arg_t;

static void *tpool_main(void *a_)
{
    tpool_arg_t *a = (tpool_arg_t *)a_; lh_tpool_t *p = a->p; const int t = a->t; unsigned seen = 0;
    free(a);
    for (;;) {
        void (*fn)(void *, int, int); void *arg;
        pthread_mutex_lock(&p->mu);
        while (p->gen == seen && !p->stop) pthread_cond_wait(&p->go, &p->mu);
        if (p->stop) { pthread_mutex_unlock(&p->mu); return NULL; }
        seen = p->gen; fn = p->fn; arg = p->arg;
        pthread_mutex_unlock(&p->mu);
        fn(arg, t, p->nt);
        pthread_mutex_lock(&p->mu);
        if (--p->pending == 0) pthread_cond_signal(&p->done);
        pthread_mutex_unlock(&p->mu);
    }
}

static lh_tpool_t *tpool_new(int nt)
{
    lh_tpool_t *p; int t;
    if (nt < 2) return NULL;
    if (nt > LH_POOL_MAX) nt = LH_POOL_MAX;
    /* these passes are memory-bound: 16 threads saturate them, and two builders run next to each other (measured on 256 cores,
     * 21 M triangles: commit 3.9 s with 16 pool threads, 5.2 s with 64) */
    { const char *e = getenv("LH_POOL_THREADS"); const int cap = (e && atoi(e) >= 2) ? atoi(e) : 16; if (cap < nt) nt = cap; }
    p = (lh_tpool_t *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->go, NULL); pthread_cond_init(&p->done, NULL);
    for (t = 1; t < nt; t++) {           /* thread 0 is the caller */
        tpool_arg_t *a = (tpool_arg_t *)malloc(sizeof(*a));
        if (!a) break;
        a->p = p; a->t = t;
        if (pthread_create(&p->th[t], NULL, tpool_main, a) != 0) { free(a); break; }
    }
    p->nt = t;
    return p;
}

static void tpool_run(lh_tpool_t *p, void (*fn)(void *, int, int), void *arg)
{
    pthread_mutex_lock(&p->mu);
    p->fn = fn; p->arg = arg; p->pending = p->nt - 1; p->gen++;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
    fn(arg, 0, p->nt);
    pthread_mutex_lock(&p->mu);
    while (p->pending) pthread_cond_wait(&p->done, &p->mu);
    pthread_mutex_unlock(&p->mu);
}

static void tpool_free(lh_tpool_t *p)
{
    int t;
    if (!p) return;
    pthread_mutex_lock(&p->mu); p->stop = 1; pthread_cond_broadcast(&p->go); pthread_mutex_unlock(&p->mu);
    for (t = 1; t < p->nt; t++) pthread_join(p->th[t], NULL);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->go); pthread_cond_destroy(&p->done);
    free(p);
}

static void chunk_of(uint32_t first, uint32_t count, int t, int nt, uint32_t *a0, uint32_t *a1)
{
    const uint32_t chunk = (count + (uint32_t)nt - 1u) / (uint32_t)nt;
    uint64_t b0 = (uint64_t)first + (uint64_t)t * chunk, b1 = b0 + chunk, end = (uint64_t)first + count;
    if (b0 > end) b0 = end;
    if (b1 > end) b1 = end;
    *a0 = (uint32_t)b0; *a1 = (uint32_t)b1;
}

#endif
