/*
 * lh_refbvh.c -- the REFERENCE-ORDER tree: lucille's own binned-SAH BVH
 * (src/render/bvh.c:276-379 ri_bvh_build, :1328-1564 bvh_construct, :1571-1692
 * bin_triangle_edge, :1230-1326 find_cut_from_bin, :1210-1228 SAH, :1697-1731
 * bbox_add_margin, :1897-1917 gather_triangles) rebuilt on the host, in double
 * precision and with the reference's exact arithmetic, so that the two results of
 * the reference that DEPEND ON ITS TREE can be reproduced on the GPU:
 *
 *   1. ri_bvh_intersect_beam_visibility (bvh.c:612-667): the answer is the class of
 *      the first non-missing triangle in the reference's traversal and leaf order,
 *      and which triangles are tested at all depends on its (<=16-triangle) leaves;
 *   2. exact-t ties of ri_bvh_intersect: inside a leaf the LAST equal-t triangle wins
 *      (bvh.c:780 rejects only t > t_best), across leaves the FIRST VISITED leaf wins
 *      (bvh.c:850 strict <), visiting order = ray->dir_sign[node->axis0] (bvh.c:1080).
 *
 * The fast fp32 traversal tree (lh_bvh.c) stays the structure rays walk; this tree is
 * consulted only by the beam kernel and by the fp64 resolve when two candidates have
 * bit-equal t.  Subtrees are built in parallel (every recursion owns its index range
 * and the reference's scratch copy is per call), which changes node numbering only.
 */
#include "lh_refbvh.h"

#include <math.h>
#include <pthread.h>
#include "lh_tpool.h"
#include <stdio.h>
#include <time.h>
static double rnow(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#include <stdlib.h>
#include <string.h>

#define REF_NTRIS_LEAF 16       /* BVH_NTRIS_LEAF bvh.c:81 */
#define REF_BIN_SIZE   64       /* BVH_BIN_SIZE   bvh.c:82 */
#define REF_EPS        1.0e-14  /* RI_EPS src/base/common.h:27 */
#define REF_INFINITY   1.0e38   /* RI_INFINITY include/ri.h:47 */

typedef struct { double bmin[3], bmax[3]; uint32_t index; } tbox_t;

typedef struct rnode {
    double box[2][6];
    struct rnode *c[2];
    int axis, is_leaf;
    uint32_t first, count;
} rnode_t;

typedef struct chunk { struct chunk *next; size_t used, cap; rnode_t *n; } chunk_t;
typedef struct { chunk_t *head; } pool_t;

static rnode_t *pool_new(pool_t *p)
{
    rnode_t *r;
    if (!p->head || p->head->used == p->head->cap) {
        chunk_t *c = (chunk_t *)malloc(sizeof(*c));
        c->cap = 1 << 13; c->used = 0; c->n = (rnode_t *)malloc(sizeof(rnode_t) * c->cap);
        c->next = p->head; p->head = c;
    }
    r = &p->head->n[p->head->used++];
    memset(r, 0, sizeof(*r));
    return r;
}

static void pool_free(pool_t *p)
{
    chunk_t *c = p->head;
    while (c) { chunk_t *nx = c->next; free(c->n); free(c); c = nx; }
    p->head = NULL;
}

/* bbox_add_margin bvh.c:1697-1731 */
static void add_margin(double bmin[3], double bmax[3])
{
    double sc[3]; int i;
    for (i = 0; i < 3; i++) {
        double scale = bmax[i] - bmin[i];
        sc[i] = (scale < REF_EPS) ? REF_EPS : REF_EPS * scale;
    }
    for (i = 0; i < 3; i++) { bmin[i] -= sc[i]; bmax[i] += sc[i]; }
}

static void bounds_of(double bmin[3], double bmax[3], const tbox_t *b, uint32_t n)
{   /* calc_bbox_of_triangles bvh.c:1872-1895 */
    uint32_t i; int k;
    for (k = 0; k < 3; k++) { bmin[k] = b[0].bmin[k]; bmax[k] = b[0].bmax[k]; }
    for (i = 1; i < n; i++)
        for (k = 0; k < 3; k++) {
            bmin[k] = (bmin[k] < b[i].bmin[k]) ? bmin[k] : b[i].bmin[k];
            bmax[k] = (bmax[k] > b[i].bmax[k]) ? bmax[k] : b[i].bmax[k];
        }
}

static double area_of(const double lo[3], const double hi[3])
{   /* calc_surface_area bvh.c:1190-1208 */
    double sa = (hi[0] - lo[0]) * (hi[1] - lo[1]) + (hi[1] - lo[1]) * (hi[2] - lo[2]) + (hi[2] - lo[2]) * (hi[0] - lo[0]);
    sa *= 2.0;
    return sa;
}

static double sah_cost(int ns1, double a1, int ns2, double a2, double s)
{   /* SAH bvh.c:1210-1228: double sum narrowed to float */
    const float Taabb = 0.2f, Ttri = 0.8f;
    float T = 2.0f * Taabb + (a1 / s) * (double)ns1 * Ttri + (a2 / s) * (double)ns2 * Ttri;
    return T;
}

typedef struct {
    tbox_t *boxes;              /* partitioned in place                       */
    uint32_t threshold;         /* ranges <= threshold become parallel tasks  */
    int collecting;
    struct rtask { rnode_t *node; double bmin[3], bmax[3]; uint32_t il, ir; } *tasks;
    size_t ntasks, cap;
    lh_tpool_t *tp;             /* the long ranges at the top of the tree: bins, partition and bounds on the thread pool.  Every one
                                 * of these passes has an order-independent result (integer histograms, min / max) or a layout
                                 * that can be computed from per-chunk counts (the partition), so the tree stays the reference's */
} rctx_t;

typedef struct { const tbox_t *b; uint32_t n; double part[LH_POOL_MAX][6]; } rbounds_job_t;
static void rbounds_part(void *j_, int t, int nt)
{
    rbounds_job_t *j = (rbounds_job_t *)j_; uint32_t a0, a1, i; int k; double *o = j->part[t];
    chunk_of(0, j->n, t, nt, &a0, &a1);
    if (a0 >= a1) { for (k = 0; k < 3; k++) { o[k] = j->b[0].bmin[k]; o[3 + k] = j->b[0].bmax[k]; } return; }
    for (k = 0; k < 3; k++) { o[k] = j->b[a0].bmin[k]; o[3 + k] = j->b[a0].bmax[k]; }
    for (i = a0 + 1; i < a1; i++)
        for (k = 0; k < 3; k++) {
            o[k] = (o[k] < j->b[i].bmin[k]) ? o[k] : j->b[i].bmin[k];
            o[3 + k] = (o[3 + k] > j->b[i].bmax[k]) ? o[3 + k] : j->b[i].bmax[k];
        }
}

/* bounds_of on the pool (min / max of doubles: exact, whatever the order) */
static void bounds_of_par(lh_tpool_t *tp, double bmin[3], double bmax[3], const tbox_t *b, uint32_t n)
{
    rbounds_job_t *j = (tp && n >= LH_PAR_MIN) ? (rbounds_job_t *)malloc(sizeof(*j)) : NULL; int t, k;
    if (!j) { bounds_of(bmin, bmax, b, n); return; }
    j->b = b; j->n = n;
    tpool_run(tp, rbounds_part, j);
    for (k = 0; k < 3; k++) { bmin[k] = j->part[0][k]; bmax[k] = j->part[0][3 + k]; }
    for (t = 1; t < tp->nt; t++)
        for (k = 0; k < 3; k++) {
            bmin[k] = (bmin[k] < j->part[t][k]) ? bmin[k] : j->part[t][k];
            bmax[k] = (bmax[k] > j->part[t][3 + k]) ? bmax[k] : j->part[t][3 + k];
        }
    free(j);
}

typedef struct { const tbox_t *b; uint32_t n; double bmin[3], inv[3]; uint32_t part[LH_POOL_MAX][2][3][REF_BIN_SIZE]; } rbin_job_t;
static void rbin_part(void *j_, int t, int nt)
{
    rbin_job_t *j = (rbin_job_t *)j_; uint32_t a0, a1, i; int k;
    chunk_of(0, j->n, t, nt, &a0, &a1);
    memset(j->part[t], 0, sizeof(j->part[t]));
    for (i = a0; i < a1; i++)
        for (k = 0; k < 3; k++) {
            double qmin = (j->b[i].bmin[k] - j->bmin[k]) * j->inv[k], qmax = (j->b[i].bmax[k] - j->bmin[k]) * j->inv[k];
            uint32_t imin = (uint32_t)qmin, imax = (uint32_t)qmax;
            if (imin >= REF_BIN_SIZE) imin = REF_BIN_SIZE - 1;
            if (imax >= REF_BIN_SIZE) imax = REF_BIN_SIZE - 1;
            j->part[t][0][k][imin]++; j->part[t][1][k][imax]++;
        }
}

/* the reference's partition (bvh.c:1437-1478): left elements keep their order, right elements are written from the end
 * backwards -- element i's place follows from how many lefts / rights precede it: per-chunk counts, prefix sums, scatter */
typedef struct { tbox_t *dst; tbox_t *scratch; uint32_t n; int axis; double pos; uint32_t lcount[LH_POOL_MAX], loff[LH_POOL_MAX], roff[LH_POOL_MAX]; int phase; } rpart_job_t;
static void rpart_part(void *j_, int t, int nt)
{
    rpart_job_t *j = (rpart_job_t *)j_; uint32_t a0, a1, i;
    chunk_of(0, j->n, t, nt, &a0, &a1);
    if (j->phase == 0) {
        uint32_t c = 0;
        if (a1 > a0) memcpy(j->scratch + a0, j->dst + a0, sizeof(tbox_t) * (size_t)(a1 - a0));
        for (i = a0; i < a1; i++) c += (j->scratch[i].bmax[j->axis] < j->pos);
        j->lcount[t] = c;
    } else {
        uint32_t wl = j->loff[t], wr = j->n - 1u - j->roff[t];
        for (i = a0; i < a1; i++) {
            if (j->scratch[i].bmax[j->axis] < j->pos) j->dst[wl++] = j->scratch[i];
            else                                      j->dst[wr--] = j->scratch[i];
        }
    }
}


static void construct(rctx_t *cx, pool_t *pool, tbox_t *scratch, rnode_t *me, const double bmin[3],
                      const double bmax[3], uint32_t il, uint32_t ir)
{
    const uint32_t n = ir - il;
    tbox_t *boxes = cx->boxes;
    uint32_t bin[2][3][REF_BIN_SIZE];
    double cut_pos = 0.0; int cut_axis = 0;
    uint32_t i, nl = 0, nr;
    int j, k;

    if (n <= REF_NTRIS_LEAF) {      /* bvh.c:1345-1403; gather_triangles = keep the box order */
        me->is_leaf = 1; me->first = il; me->count = n;
        return;
    }
    if (cx->collecting && n <= cx->threshold) {
        if (cx->ntasks == cx->cap) { cx->cap = cx->cap ? cx->cap * 2 : 256; cx->tasks = (struct rtask *)realloc(cx->tasks, sizeof(*cx->tasks) * cx->cap); }
        cx->tasks[cx->ntasks].node = me; cx->tasks[cx->ntasks].il = il; cx->tasks[cx->ntasks].ir = ir;
        for (k = 0; k < 3; k++) { cx->tasks[cx->ntasks].bmin[k] = bmin[k]; cx->tasks[cx->ntasks].bmax[k] = bmax[k]; }
        cx->ntasks++;
        return;
    }

    /* bin_triangle_edge bvh.c:1571-1692 */
    {
        double size[3], inv[3];
        for (k = 0; k < 3; k++) { size[k] = bmax[k] - bmin[k]; inv[k] = (size[k] > REF_EPS) ? (double)REF_BIN_SIZE / size[k] : 0.0; }
        memset(bin, 0, sizeof(bin));
        if (cx->collecting && cx->tp && n >= LH_PAR_MIN) {
            rbin_job_t *jb = (rbin_job_t *)malloc(sizeof(*jb));
            if (jb) {
                int t, h, b2;
                jb->b = boxes + il; jb->n = n;
                for (k = 0; k < 3; k++) { jb->bmin[k] = bmin[k]; jb->inv[k] = inv[k]; }
                tpool_run(cx->tp, rbin_part, jb);
                for (t = 0; t < cx->tp->nt; t++) for (h = 0; h < 2; h++) for (k = 0; k < 3; k++) for (b2 = 0; b2 < REF_BIN_SIZE; b2++) bin[h][k][b2] += jb->part[t][h][k][b2];
                free(jb);
                goto binned;
            }
        }
        for (i = 0; i < n; i++)
            for (k = 0; k < 3; k++) {
                double qmin = (boxes[il + i].bmin[k] - bmin[k]) * inv[k], qmax = (boxes[il + i].bmax[k] - bmin[k]) * inv[k];
                uint32_t imin = (uint32_t)qmin, imax = (uint32_t)qmax;
                if (imin >= REF_BIN_SIZE) imin = REF_BIN_SIZE - 1;
                if (imax >= REF_BIN_SIZE) imax = REF_BIN_SIZE - 1;
                bin[0][k][imin]++; bin[1][k][imax]++;
            }
        binned: ;
    }
    /* find_cut_from_bin bvh.c:1230-1326 */
    {
        double bstep[3], min_cost = REF_INFINITY, sa_total = area_of(bmin, bmax);
        for (k = 0; k < 3; k++) bstep[k] = (bmax[k] - bmin[k]) / (double)REF_BIN_SIZE;
        for (j = 0; j < 3; j++) {
            uint64_t left = 0, right = n; double lmin[3], lmax[3], rmin[3], rmax[3]; int b;
            for (k = 0; k < 3; k++) { lmin[k] = rmin[k] = bmin[k]; lmax[k] = rmax[k] = bmax[k]; }
            for (b = 0; b < REF_BIN_SIZE - 1; b++) {
                double pos, cost;
                left += bin[0][j][b]; right -= bin[1][j][b];
                pos = bmin[j] + (b + 1) * bstep[j];
                lmax[j] = pos; rmin[j] = pos;
                cost = sah_cost((int)left, area_of(lmin, lmax), (int)right, area_of(rmin, rmax), sa_total);
                if (cost < min_cost) { min_cost = cost; cut_axis = j; cut_pos = pos; }
            }
        }
    }
    /* partition bvh.c:1437-1478 */
    nr = n - 1;
    {
        rpart_job_t *jp = (cx->collecting && cx->tp && n >= LH_PAR_MIN) ? (rpart_job_t *)malloc(sizeof(*jp)) : NULL;
        if (jp) {
            int t; uint32_t cl = 0, cr = 0;
            jp->dst = boxes + il; jp->scratch = scratch; jp->n = n; jp->axis = cut_axis; jp->pos = cut_pos;
            jp->phase = 0; tpool_run(cx->tp, rpart_part, jp);
            for (t = 0; t < cx->tp->nt; t++) {
                uint32_t a0, a1; chunk_of(0, n, t, cx->tp->nt, &a0, &a1);
                jp->loff[t] = cl; jp->roff[t] = cr; cl += jp->lcount[t]; cr += (a1 - a0) - jp->lcount[t];
            }
            jp->phase = 1; tpool_run(cx->tp, rpart_part, jp);
            nl = cl;
            free(jp);
        } else {
            memcpy(scratch, boxes + il, sizeof(tbox_t) * n);
            for (i = 0; i < n; i++) {
                if (scratch[i].bmax[cut_axis] < cut_pos) boxes[il + nl++] = scratch[i];
                else                                     boxes[il + nr--] = scratch[i];
            }
        }
    }
    if (nl == 0 || nl == n) nl = n / 2;

    me->axis = cut_axis;
    me->c[0] = pool_new(pool); me->c[1] = pool_new(pool);
    {
        double lmin[3], lmax[3], rmin[3], rmax[3];
        bounds_of_par(cx->collecting ? cx->tp : NULL, lmin, lmax, boxes + il, nl); add_margin(lmin, lmax);
        for (k = 0; k < 3; k++) { me->box[0][k] = lmin[k]; me->box[0][3 + k] = lmax[k]; }
        construct(cx, pool, scratch, me->c[0], lmin, lmax, il, il + nl);
        bounds_of_par(cx->collecting ? cx->tp : NULL, rmin, rmax, boxes + il + nl, n - nl); add_margin(rmin, rmax);
        for (k = 0; k < 3; k++) { me->box[1][k] = rmin[k]; me->box[1][3 + k] = rmax[k]; }
        construct(cx, pool, scratch, me->c[1], rmin, rmax, il + nl, ir);
    }
}

typedef struct { rctx_t *cx; pool_t pool; volatile uint32_t *next; uint32_t max_n; } rworker_t;

static void *rworker_main(void *arg)
{
    rworker_t *w = (rworker_t *)arg;
    tbox_t *scratch = (tbox_t *)malloc(sizeof(tbox_t) * (size_t)(w->max_n ? w->max_n : 1));
    for (;;) {
        uint32_t t = __sync_fetch_and_add(w->next, 1);
        if (t >= w->cx->ntasks) break;
        construct(w->cx, &w->pool, scratch, w->cx->tasks[t].node, w->cx->tasks[t].bmin, w->cx->tasks[t].bmax,
                  w->cx->tasks[t].il, w->cx->tasks[t].ir);
    }
    free(scratch);
    return NULL;
}

static uint32_t count_nodes(const rnode_t *n) { return n->is_leaf ? 1 : 1 + count_nodes(n->c[0]) + count_nodes(n->c[1]); }

static void flatten(lh_refbvh_t *o, const rnode_t *n, uint32_t idx, int32_t parent, int32_t depth, uint32_t *next,
                    const tbox_t *boxes)
{
    lh_refnode_t *d = &o->nodes[idx]; int k;
    memcpy(d->box, n->box, sizeof(d->box));
    d->axis = n->axis; d->is_leaf = n->is_leaf; d->first = n->first; d->count = n->count;
    d->parent = parent; d->depth = depth;
    if ((uint32_t)depth > o->max_depth) o->max_depth = (uint32_t)depth;
    if (n->is_leaf) {
        uint32_t i;
        d->child[0] = d->child[1] = -1;
        for (i = 0; i < n->count; i++) {
            uint32_t p = boxes[n->first + i].index;
            o->leaf_prims[n->first + i] = p; o->prim_leaf[p] = idx; o->prim_pos[p] = i;
        }
        return;
    }
    for (k = 0; k < 2; k++) d->child[k] = (int32_t)(*next)++;
    flatten(o, n->c[0], (uint32_t)d->child[0], (int32_t)idx, depth + 1, next, boxes);
    flatten(o, n->c[1], (uint32_t)d->child[1], (int32_t)idx, depth + 1, next, boxes);
}

int lh_refbvh_build(lh_refbvh_t *o, const lh_tri64_t *tri64, uint32_t ntris, int nthreads)
{
    rctx_t cx; pool_t main_pool; rnode_t *root; tbox_t *scratch; uint32_t i; int k;
    const double t0 = rnow(); double t1, t2, t3;
    memset(o, 0, sizeof(*o));
    o->ntris = ntris;
    if (ntris == 0) { o->empty = 1; return 0; }     /* bvh.c:311-315 */
    memset(&cx, 0, sizeof(cx)); memset(&main_pool, 0, sizeof(main_pool));
    cx.boxes = (tbox_t *)malloc(sizeof(tbox_t) * ntris);
    scratch = (tbox_t *)malloc(sizeof(tbox_t) * ntris);
    o->leaf_prims = (uint32_t *)malloc(sizeof(uint32_t) * ntris);
    o->prim_leaf = (uint32_t *)malloc(sizeof(uint32_t) * ntris);
    o->prim_pos = (uint32_t *)malloc(sizeof(uint32_t) * ntris);
    if (!cx.boxes || !scratch || !o->leaf_prims || !o->prim_leaf || !o->prim_pos) { free(cx.boxes); free(scratch); lh_refbvh_release(o); return -1; }
    for (i = 0; i < ntris; i++) {       /* get_bbox_of_triangle bvh.c:1852-1869 */
        const lh_tri64_t *t = &tri64[i];
        for (k = 0; k < 3; k++) {
            double lo = t->v[0][k], hi = t->v[0][k];
            lo = (lo < t->v[1][k]) ? lo : t->v[1][k]; lo = (lo < t->v[2][k]) ? lo : t->v[2][k];
            hi = (hi > t->v[1][k]) ? hi : t->v[1][k]; hi = (hi > t->v[2][k]) ? hi : t->v[2][k];
            cx.boxes[i].bmin[k] = lo; cx.boxes[i].bmax[k] = hi;
        }
        cx.boxes[i].index = i;
    }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (nthreads > 1 && ntris > 50000) {
        cx.collecting = 1; cx.threshold = ntris / (uint32_t)(nthreads * 8); if (cx.threshold < 2048) cx.threshold = 2048;
        cx.tp = tpool_new(nthreads);
        /* no serial pass over more than LH_PAR_MIN boxes at the top */
        if (cx.tp && cx.threshold < LH_PAR_MIN && ntris / LH_PAR_MIN >= (uint32_t)(2 * nthreads)) cx.threshold = LH_PAR_MIN;
    }
    t1 = rnow();
    bounds_of_par(cx.tp, o->bmin, o->bmax, cx.boxes, ntris);
    add_margin(o->bmin, o->bmax);
    root = pool_new(&main_pool);
    construct(&cx, &main_pool, scratch, root, o->bmin, o->bmax, 0, ntris);
    cx.collecting = 0;
    tpool_free(cx.tp); cx.tp = NULL;
    t2 = rnow();
    {
        rworker_t *w = NULL; pthread_t *th = NULL; volatile uint32_t next = 0; int t; uint32_t max_n = 0; size_t q;
        for (q = 0; q < cx.ntasks; q++) if (cx.tasks[q].ir - cx.tasks[q].il > max_n) max_n = cx.tasks[q].ir - cx.tasks[q].il;
        if (cx.ntasks) {
            w = (rworker_t *)calloc((size_t)nthreads, sizeof(*w)); th = (pthread_t *)calloc((size_t)nthreads, sizeof(*th));
            for (t = 0; t < nthreads; t++) { w[t].cx = &cx; w[t].next = &next; w[t].max_n = max_n; pthread_create(&th[t], NULL, rworker_main, &w[t]); }
            for (t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        }
        t3 = rnow();
        o->nnodes = count_nodes(root);
        o->nodes = (lh_refnode_t *)calloc(o->nnodes, sizeof(lh_refnode_t));
        if (!o->nodes) { free(cx.boxes); free(scratch); lh_refbvh_release(o); return -1; }
        { uint32_t nx = 1; flatten(o, root, 0, -1, 0, &nx, cx.boxes); }
        if (w) for (t = 0; t < nthreads; t++) pool_free(&w[t].pool);
        free(w); free(th);
    }
    pool_free(&main_pool);
    if (getenv("LH_BUILD_TIMING")) fprintf(stderr, "[lh_refbvh] boxes %.3f top %.3f subtrees %.3f flatten %.3f (tasks %zu)\n", t1 - t0, t2 - t1, t3 - t2, rnow() - t3, cx.ntasks);
    free(cx.tasks); free(cx.boxes); free(scratch);
    return 0;
}

void lh_refbvh_release(lh_refbvh_t *o)
{
    free(o->nodes); free(o->leaf_prims); free(o->prim_leaf); free(o->prim_pos);
    memset(o, 0, sizeof(*o));
}

/* The reference's winner among two primitives whose hits have bit-equal t (host version of
 * the device routine in lh_kernels.hip; used by the CPU model in tests/). dir_sign[k] =
 * dir[k] < 0 (bvh.c:473-475). */
uint32_t lh_refbvh_tie_winner(const lh_refbvh_t *o, uint32_t a, uint32_t b, const int dir_sign[3])
{
    uint32_t la = o->prim_leaf[a], lb = o->prim_leaf[b], ca, cb;
    if (la == lb) return (o->prim_pos[a] > o->prim_pos[b]) ? a : b;      /* bvh.c:780: last equal t in the leaf */
    ca = la; cb = lb;
    while (o->nodes[ca].depth > o->nodes[cb].depth) ca = (uint32_t)o->nodes[ca].parent;
    while (o->nodes[cb].depth > o->nodes[ca].depth) cb = (uint32_t)o->nodes[cb].parent;
    while (o->nodes[ca].parent != o->nodes[cb].parent) { ca = (uint32_t)o->nodes[ca].parent; cb = (uint32_t)o->nodes[cb].parent; }
    {
        const lh_refnode_t *l = &o->nodes[o->nodes[ca].parent];
        const int order = dir_sign[l->axis];                             /* bvh.c:1080,1171-1178 */
        return ((uint32_t)l->child[order] == ca) ? a : b;               /* first visited leaf wins (bvh.c:850) */
    }
}
