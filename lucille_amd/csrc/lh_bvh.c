/*
 * lh_bvh.c -- host BVH builder for the gfx950 traversal kernels.
 *
 * Role in the drop-in: this is what accel->build() runs for RI_ACCEL_HIP, in
 * place of ri_bvh_build (reference src/render/bvh.c:276-379).  Input is the
 * scene's geom list flattened exactly like create_triangle_list
 * (bvh.c:1736-1826) so primitive ids agree with the reference; output is the
 * SoA form described in lh_bvh.h / DESIGN.md.
 *
 * Algorithm (this project's own): top-down binned SAH over centroid bounds
 * (32 bins x 3 axes), leaves of <= 4 triangles, boxes stored as fp32 rounded
 * OUTWARD from the fp64 vertices so that fp32 culling is conservative with
 * respect to the reference's fp64 geometry.  Large inputs are built in
 * parallel: the main thread splits the top of the tree, subtrees below a size
 * threshold become tasks for a pthread pool, and a final DFS pass flattens
 * the pointer tree so that sibling inner nodes are adjacent in memory (one
 * 128-byte line when both are visited).
 */
#include "lh_bvh.h"
#include "lh_tpool.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NBINS 32

typedef struct tnode {
    float lo[3], hi[3];
    struct tnode *c[2];
    uint32_t first, count;   /* leaf range in order[] (count>0 => leaf) */
    int axis;
    uint32_t task;           /* 1 + index of the subtree task rooted here, 0: none (the flatten pass hands these subtrees to threads) */
} tnode_t;

/* chunked arena so tnode pointers stay valid */
typedef struct arena_chunk { struct arena_chunk *next; size_t used, cap; tnode_t *nodes; } arena_chunk_t;
typedef struct { arena_chunk_t *head; } arena_t;

static tnode_t *arena_new(arena_t *a)
{
    if (!a->head || a->head->used == a->head->cap) {
        arena_chunk_t *c = (arena_chunk_t *)malloc(sizeof(*c));
        c->cap = 1 << 14; c->used = 0;
        c->nodes = (tnode_t *)malloc(sizeof(tnode_t) * c->cap);
        c->next = a->head; a->head = c;
    }
    tnode_t *n = &a->head->nodes[a->head->used++];
    memset(n, 0, sizeof(*n));
    return n;
}

static void arena_free(arena_t *a)
{
    arena_chunk_t *c = a->head;
    while (c) { arena_chunk_t *nx = c->next; free(c->nodes); free(c); c = nx; }
    a->head = NULL;
}

typedef struct {
    uint32_t  n;
    float    *plo, *phi;    /* per-primitive fp32 outward box, [n][3] */
    float    *cen;          /* centroids [n][3] */
    uint32_t *order;        /* permutation being partitioned */
    float     ci, ct;       /* SAH constants */
    uint32_t  task_threshold;
    /* deferred subtree tasks */
    struct task { tnode_t *node; uint32_t first, count; int depth;
                  uint32_t ninner;                                   /* inner nodes of the finished subtree */
                  uint32_t idx, node_base, tri_base, fdepth;         /* flatten: where the subtree goes */
                  uint32_t max_depth, nleaves; } *tasks;
    size_t ntasks, captasks;
    int collecting;
    lh_tpool_t *pool;      /* the top of the tree (ranges of >= LH_PAR_MIN primitives while tasks are collected) runs its passes on it */
    uint32_t *tmp;          /* [n]: scratch of the parallel partition */
    int drop_dead;          /* triangles the reference can never report stay out of the tree (tri_dead_class) */
} build_ctx_t;

static inline float down32(double d) { float f = (float)d; if ((double)f > d) f = nextafterf(f, -INFINITY); return f; }
static inline float up32(double d)   { float f = (float)d; if ((double)f < d) f = nextafterf(f,  INFINITY); return f; }

static inline float half_area(const float lo[3], const float hi[3])
{
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

typedef struct { const build_ctx_t *b; uint32_t first, count; float part[LH_POOL_MAX][12]; } bounds_job_t;

static void bounds_part(void *j_, int t, int nt)
{
    bounds_job_t *j = (bounds_job_t *)j_; const build_ctx_t *b = j->b; float *o = j->part[t];
    uint32_t a0, a1, i; int k;
    chunk_of(j->first, j->count, t, nt, &a0, &a1);
    for (k = 0; k < 3; k++) { o[k] = o[6 + k] = INFINITY; o[3 + k] = o[9 + k] = -INFINITY; }
    for (i = a0; i < a1; i++) {
        const uint32_t p = b->order[i];
        for (k = 0; k < 3; k++) {
            const float l = b->plo[3 * (size_t)p + k], h = b->phi[3 * (size_t)p + k], c = b->cen[3 * (size_t)p + k];
            if (l < o[k]) o[k] = l;
            if (h > o[3 + k]) o[3 + k] = h;
            if (c < o[6 + k]) o[6 + k] = c;
            if (c > o[9 + k]) o[9 + k] = c;
        }
    }
}

static void range_bounds(const build_ctx_t *b, uint32_t first, uint32_t count,
                         float lo[3], float hi[3], float clo[3], float chi[3])
{
    int k; uint32_t i;
    for (k = 0; k < 3; k++) { lo[k] = clo[k] = INFINITY; hi[k] = chi[k] = -INFINITY; }
    if (b->collecting && b->pool && count >= LH_PAR_MIN) {        /* min / max: the same result in any order */
        bounds_job_t *j = (bounds_job_t *)malloc(sizeof(*j)); int t;
        if (j) {
            j->b = b; j->first = first; j->count = count;
            tpool_run(b->pool, bounds_part, j);
            for (t = 0; t < b->pool->nt; t++)
                for (k = 0; k < 3; k++) {
                    if (j->part[t][k] < lo[k]) lo[k] = j->part[t][k];
                    if (j->part[t][3 + k] > hi[k]) hi[k] = j->part[t][3 + k];
                    if (j->part[t][6 + k] < clo[k]) clo[k] = j->part[t][6 + k];
                    if (j->part[t][9 + k] > chi[k]) chi[k] = j->part[t][9 + k];
                }
            free(j);
            return;
        }
    }
    for (i = first; i < first + count; i++) {
        uint32_t p = b->order[i];
        for (k = 0; k < 3; k++) {
            float l = b->plo[3 * (size_t)p + k], h = b->phi[3 * (size_t)p + k], c = b->cen[3 * (size_t)p + k];
            if (l < lo[k]) lo[k] = l;
            if (h > hi[k]) hi[k] = h;
            if (c < clo[k]) clo[k] = c;
            if (c > chi[k]) chi[k] = c;
        }
    }
}

typedef struct { uint32_t cnt[3][NBINS]; float lo[3][NBINS][3], hi[3][NBINS][3]; } bins_t;

static void bins_clear(bins_t *B)
{
    int a, j, k;
    for (a = 0; a < 3; a++) for (j = 0; j < NBINS; j++) { B->cnt[a][j] = 0; for (k = 0; k < 3; k++) { B->lo[a][j][k] = INFINITY; B->hi[a][j][k] = -INFINITY; } }
}

static void bins_fill(const build_ctx_t *b, uint32_t i0, uint32_t i1, const int live[3], const float scale[3], const float clo[3], bins_t *B)
{
    uint32_t i; int a, k;
    for (i = i0; i < i1; i++) {
        const uint32_t p = b->order[i];
        const float *pl = &b->plo[3 * (size_t)p], *ph = &b->phi[3 * (size_t)p];
        for (a = 0; a < 3; a++) {
            int bin;
            if (!live[a]) continue;
            bin = (int)((b->cen[3 * (size_t)p + a] - clo[a]) * scale[a]);
            if (bin < 0) bin = 0;
            if (bin >= NBINS) bin = NBINS - 1;
            B->cnt[a][bin]++;
            for (k = 0; k < 3; k++) { if (pl[k] < B->lo[a][bin][k]) B->lo[a][bin][k] = pl[k]; if (ph[k] > B->hi[a][bin][k]) B->hi[a][bin][k] = ph[k]; }
        }
    }
}

typedef struct { const build_ctx_t *b; uint32_t first, count; int live[3]; float scale[3], clo[3]; bins_t part[LH_POOL_MAX]; } bin_job_t;

static void bin_part(void *j_, int t, int nt)
{
    bin_job_t *j = (bin_job_t *)j_; uint32_t a0, a1;
    chunk_of(j->first, j->count, t, nt, &a0, &a1);
    bins_clear(&j->part[t]);
    bins_fill(j->b, a0, a1, j->live, j->scale, j->clo, &j->part[t]);
}

/* stable partition of order[first, first + count) by bin <= best_bin, by chunks: left counts, prefix sums, scatter, copy back */
typedef struct { build_ctx_t *b; uint32_t first, count, nleft; int axis, best_bin; float c0, scale; uint32_t lcount[LH_POOL_MAX], loff[LH_POOL_MAX], roff[LH_POOL_MAX]; int phase; } part_job_t;

static void part_part(void *j_, int t, int nt)
{
    part_job_t *j = (part_job_t *)j_; build_ctx_t *b = j->b; uint32_t a0, a1, i;
    chunk_of(j->first, j->count, t, nt, &a0, &a1);
    if (j->phase == 0) {
        uint32_t c = 0;
        for (i = a0; i < a1; i++) {
            int bin = (int)((b->cen[3 * (size_t)b->order[i] + j->axis] - j->c0) * j->scale);
            if (bin < 0) bin = 0;
            if (bin >= NBINS) bin = NBINS - 1;
            c += (bin <= j->best_bin);
        }
        j->lcount[t] = c;
    } else if (j->phase == 1) {
        uint32_t wl = j->first + j->loff[t], wr = j->first + j->nleft + j->roff[t];
        for (i = a0; i < a1; i++) {
            const uint32_t p = b->order[i];
            int bin = (int)((b->cen[3 * (size_t)p + j->axis] - j->c0) * j->scale);
            if (bin < 0) bin = 0;
            if (bin >= NBINS) bin = NBINS - 1;
            if (bin <= j->best_bin) b->tmp[wl++] = p; else b->tmp[wr++] = p;
        }
    } else if (a1 > a0) memcpy(&b->order[a0], &b->tmp[a0], sizeof(uint32_t) * (size_t)(a1 - a0));
}

static void build_range(build_ctx_t *b, arena_t *ar, tnode_t *node, uint32_t first,
                        uint32_t count, int depth);

static void make_leaf(tnode_t *node, uint32_t first, uint32_t count)
{
    node->first = first; node->count = count; node->c[0] = node->c[1] = NULL;
}

static void split_children(build_ctx_t *b, arena_t *ar, tnode_t *node, uint32_t first,
                           uint32_t nl, uint32_t count, int depth)
{
    node->count = 0;
    node->c[0] = arena_new(ar); node->c[1] = arena_new(ar);
    build_range(b, ar, node->c[0], first, nl, depth + 1);
    build_range(b, ar, node->c[1], first + nl, count - nl, depth + 1);
}

static void build_range(build_ctx_t *b, arena_t *ar, tnode_t *node, uint32_t first,
                        uint32_t count, int depth)
{
    float clo[3], chi[3];
    int k, axis, best_axis = -1, best_bin = -1;
    float best_cost = INFINITY;

    if (b->collecting && count <= b->task_threshold && count > LH_MAX_LEAF_TRIS) {
        if (b->ntasks == b->captasks) {
            b->captasks = b->captasks ? b->captasks * 2 : 256;
            b->tasks = (struct task *)realloc(b->tasks, sizeof(*b->tasks) * b->captasks);
        }
        b->tasks[b->ntasks].node = node; b->tasks[b->ntasks].first = first;
        b->tasks[b->ntasks].count = count; b->tasks[b->ntasks].depth = depth;
        b->ntasks++;
        node->task = (uint32_t)b->ntasks;
        return;
    }

    range_bounds(b, first, count, node->lo, node->hi, clo, chi);

    if (count == 1) { make_leaf(node, first, count); return; }

    if (depth >= 48) {   /* degenerate input guard: bounded depth */
        if (count <= LH_MAX_LEAF_TRIS) { make_leaf(node, first, count); return; }
        node->axis = 0;
        split_children(b, ar, node, first, count / 2, count, depth);
        return;
    }

    /* binned SAH: one pass fills the bins of all three axes (counts and min / max boxes: the same in any order, so the
     * long ranges at the top of the tree are binned by the pool's threads), then a sweep per axis */
    {
        float parent_area = half_area(node->lo, node->hi);
        bins_t bins, *B = &bins;          /* 2.7 KB on the stack per level of the recursion */
        int live[3]; float scale3[3];
        int j;
        for (axis = 0; axis < 3; axis++) {
            const float ext = chi[axis] - clo[axis];
            live[axis] = ext > 0.0f;
            scale3[axis] = live[axis] ? (float)NBINS * (1.0f - 1e-6f) / ext : 0.0f;
        }
        if (b->collecting && b->pool && count >= LH_PAR_MIN) {
            bin_job_t *jb = (bin_job_t *)malloc(sizeof(*jb)); int t, a, kk;
            bins_clear(B);
            if (jb) {
                jb->b = b; jb->first = first; jb->count = count;
                for (a = 0; a < 3; a++) { jb->live[a] = live[a]; jb->scale[a] = scale3[a]; jb->clo[a] = clo[a]; }
                tpool_run(b->pool, bin_part, jb);
                for (t = 0; t < b->pool->nt; t++)
                    for (a = 0; a < 3; a++) for (j = 0; j < NBINS; j++) {
                        B->cnt[a][j] += jb->part[t].cnt[a][j];
                        for (kk = 0; kk < 3; kk++) {
                            if (jb->part[t].lo[a][j][kk] < B->lo[a][j][kk]) B->lo[a][j][kk] = jb->part[t].lo[a][j][kk];
                            if (jb->part[t].hi[a][j][kk] > B->hi[a][j][kk]) B->hi[a][j][kk] = jb->part[t].hi[a][j][kk];
                        }
                    }
                free(jb);
            } else bins_fill(b, first, first + count, live, scale3, clo, B);
        } else {
            bins_clear(B);
            bins_fill(b, first, first + count, live, scale3, clo, B);
        }
        for (axis = 0; axis < 3; axis++) {
            float rarea[NBINS]; uint32_t rcnt[NBINS];
            float lo[3], hi[3]; uint32_t n;
            if (!live[axis]) continue;
            /* right-to-left sweep */
            for (k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
            n = 0;
            for (j = NBINS - 1; j >= 1; j--) {
                n += B->cnt[axis][j];
                for (k = 0; k < 3; k++) { if (B->lo[axis][j][k] < lo[k]) lo[k] = B->lo[axis][j][k]; if (B->hi[axis][j][k] > hi[k]) hi[k] = B->hi[axis][j][k]; }
                rcnt[j] = n; rarea[j] = n ? half_area(lo, hi) : 0.0f;
            }
            for (k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
            n = 0;
            for (j = 0; j < NBINS - 1; j++) {
                float cost;
                n += B->cnt[axis][j];
                for (k = 0; k < 3; k++) { if (B->lo[axis][j][k] < lo[k]) lo[k] = B->lo[axis][j][k]; if (B->hi[axis][j][k] > hi[k]) hi[k] = B->hi[axis][j][k]; }
                if (n == 0 || rcnt[j + 1] == 0) continue;
                cost = half_area(lo, hi) * (float)n + rarea[j + 1] * (float)rcnt[j + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = j; }
            }
        }

        if (best_axis >= 0 && count <= LH_MAX_LEAF_TRIS) {
            float split_cost = b->ci + b->ct * best_cost / (parent_area > 0.0f ? parent_area : 1e-30f);
            if ((float)count * b->ct <= split_cost) { make_leaf(node, first, count); return; }
        }
    }

    if (best_axis < 0) {
        /* all centroids coincide: cannot separate spatially */
        if (count <= LH_MAX_LEAF_TRIS) { make_leaf(node, first, count); return; }
        node->axis = 0;
        split_children(b, ar, node, first, count / 2, count, depth);
        return;
    }

    /* partition order[first, first+count) by bin <= best_bin */
    {
        float ext = chi[best_axis] - clo[best_axis];
        float scale = (float)NBINS * (1.0f - 1e-6f) / ext;
        uint32_t l = first, r = first + count;
        if (b->collecting && b->pool && b->tmp && count >= LH_PAR_MIN) {
            /* (the serial branch below swaps in place; which order the primitives of a child end up in only decides the order
             * of the triangles inside a leaf -- the tree itself depends on the child SETS alone) */
            part_job_t *j = (part_job_t *)malloc(sizeof(*j));
            if (j) {
                int t; uint32_t nl = 0, nr = 0;
                j->b = b; j->first = first; j->count = count; j->axis = best_axis; j->best_bin = best_bin; j->c0 = clo[best_axis]; j->scale = scale;
                j->phase = 0; tpool_run(b->pool, part_part, j);
                for (t = 0; t < b->pool->nt; t++) {
                    uint32_t a0, a1; chunk_of(first, count, t, b->pool->nt, &a0, &a1);
                    j->loff[t] = nl; j->roff[t] = nr; nl += j->lcount[t]; nr += (a1 - a0) - j->lcount[t];
                }
                j->nleft = nl;
                j->phase = 1; tpool_run(b->pool, part_part, j);
                j->phase = 2; tpool_run(b->pool, part_part, j);
                free(j);
                l = r = first + nl;
            }
        }
        while (l < r) {
            uint32_t p = b->order[l];
            int bin = (int)((b->cen[3 * (size_t)p + best_axis] - clo[best_axis]) * scale);
            if (bin < 0) bin = 0;
            if (bin >= NBINS) bin = NBINS - 1;
            if (bin <= best_bin) l++;
            else { r--; b->order[l] = b->order[r]; b->order[r] = p; }
        }
        node->axis = best_axis;
        {
            uint32_t nl = l - first;
            if (nl == 0 || nl == count) nl = count / 2;   /* cannot happen; guard */
            split_children(b, ar, node, first, nl, count, depth);
        }
    }
}

/* inner nodes below (and including) n; a subtree task's root contributes the count its worker stored */
static uint32_t count_inner(const tnode_t *n) { return n->count ? 0 : 1 + count_inner(n->c[0]) + count_inner(n->c[1]); }

static uint32_t count_inner_top(const build_ctx_t *b, const tnode_t *n)
{
    if (n->count) return 0;
    if (n->task) return b->tasks[n->task - 1].ninner;
    return 1 + count_inner_top(b, n->c[0]) + count_inner_top(b, n->c[1]);
}

typedef struct { build_ctx_t *b; arena_t arena; volatile uint32_t *next; } worker_t;

static void *worker_main(void *arg)
{
    worker_t *w = (worker_t *)arg;
    for (;;) {
        uint32_t t = __sync_fetch_and_add(w->next, 1);
        if (t >= w->b->ntasks) break;
        build_range(w->b, &w->arena, w->b->tasks[t].node, w->b->tasks[t].first,
                    w->b->tasks[t].count, w->b->tasks[t].depth);
        w->b->tasks[t].ninner = count_inner(w->b->tasks[t].node);
    }
    return NULL;
}

/* ------------------------------------------------------------- flatten */

typedef struct {
    const build_ctx_t *b; lh_bvh_t *out; const lh_tri64_t *tri64;
    uint32_t next_node, next_tri, max_depth, nleaves;
    int defer;               /* the serial pass over the top of the tree: subtree tasks are only given their place */
} flat_t;

/* the 48-byte filter record of primitive p */
static void make_tri32(lh_tri32_t *o, const lh_tri64_t *t, uint32_t p)
{
    double e1[3], e2[3], n1, n2; int k;
    for (k = 0; k < 3; k++) {
        e1[k] = t->v[1][k] - t->v[0][k]; e2[k] = t->v[2][k] - t->v[0][k];
        o->v0[k] = (float)t->v[0][k];
    }
    o->e1x = (float)e1[0]; o->e1y = (float)e1[1]; o->e1z = (float)e1[2];
    o->e2x = (float)e2[0]; o->e2y = (float)e2[1]; o->e2z = (float)e2[2];
    n1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    n2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
    o->prim = p;
    o->ne1 = up32(n1 * (1.0 + 1e-6));
    o->ne2 = up32(n2 * (1.0 + 1e-6));
}

static int32_t emit_leaf(flat_t *f, const tnode_t *n)
{
    uint32_t first = f->next_tri, i;
    for (i = 0; i < n->count; i++) {
        uint32_t p = f->b->order[n->first + i];
        make_tri32(&f->out->tri32[f->next_tri++], &f->tri64[p], p);
    }
    f->nleaves++;
    return ~(int32_t)((first << 2) | (n->count - 1));
}

static void emit_inner(flat_t *f, const tnode_t *n, uint32_t idx, uint32_t depth)
{
    lh_node_t *o = &f->out->nodes[idx];
    const tnode_t *c0 = n->c[0], *c1 = n->c[1];
    uint32_t k0 = 0, k1 = 0; int k;
    if (depth > f->max_depth) f->max_depth = depth;
    for (k = 0; k < 3; k++) { o->lo0[k] = c0->lo[k]; o->hi0[k] = c0->hi[k]; o->lo1[k] = c1->lo[k]; o->hi1[k] = c1->hi[k]; }
    o->axis = n->axis; o->pad = 0;
    if (c0->count == 0) k0 = f->next_node++;
    if (c1->count == 0) k1 = f->next_node++;
    if (c0->count) o->ref0 = emit_leaf(f, c0); else { o->ref0 = (int32_t)k0; }
    if (c1->count) o->ref1 = emit_leaf(f, c1); else { o->ref1 = (int32_t)k1; }
    {
        const tnode_t *c[2] = { c0, c1 }; const uint32_t kk[2] = { k0, k1 }; int s;
        for (s = 0; s < 2; s++) {
            if (c[s]->count) continue;
            if (f->defer && c[s]->task) {
                /* a finished subtree: its nodes (all but its root) and its triangles are the next contiguous blocks of the
                 * depth-first layout; a thread fills them in later (flatten_task) */
                struct task *tk = &((build_ctx_t *)f->b)->tasks[c[s]->task - 1];
                tk->idx = kk[s]; tk->fdepth = depth + 1; tk->node_base = f->next_node; tk->tri_base = f->next_tri;
                f->next_node += tk->ninner - 1; f->next_tri += tk->count;
            } else emit_inner(f, c[s], kk[s], depth + 1);
        }
    }
}

typedef struct { flat_t base; build_ctx_t *b; volatile uint32_t *next; } flat_worker_t;

static void *flatten_worker(void *arg)
{
    flat_worker_t *w = (flat_worker_t *)arg;
    for (;;) {
        uint32_t t = __sync_fetch_and_add(w->next, 1);
        struct task *tk; flat_t f;
        if (t >= w->b->ntasks) break;
        tk = &w->b->tasks[t];
        f = w->base; f.defer = 0; f.next_node = tk->node_base; f.next_tri = tk->tri_base; f.max_depth = 0; f.nleaves = 0;
        emit_inner(&f, tk->node, tk->idx, tk->fdepth);
        tk->max_depth = f.max_depth; tk->nleaves = f.nleaves;
    }
    return NULL;
}

/* the scene's 16-bit grid: the 4-wide and 8-wide node formats live on it */
static void setup_grid(lh_bvh_t *o)
{
    int k;
    for (k = 0; k < 3; k++) {
        double ext = (double)o->bmax[k] - (double)o->bmin[k];
        o->grid_lo[k] = o->bmin[k];
        o->grid_step[k] = up32(ext > 0.0 ? ext / 65535.0 * (1.0 + 1e-6) : 1e-30);
    }
}

/* ---- 4-wide collapse of the flat binary tree (see lh_q4node_t) ------------------------- */
typedef struct { const float *lo, *hi; int32_t ref; } child4_t;

static float area3(const float *lo, const float *hi)
{
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

static void quant_box(const lh_bvh_t *o, const float *lo, const float *hi, uint16_t q[6])
{
    int k;
    for (k = 0; k < 3; k++) {
        const double g = o->grid_lo[k], st = o->grid_step[k];
        double ql = floor(((double)lo[k] - g) / st), qh = ceil(((double)hi[k] - g) / st);
        if (ql < 0.0) ql = 0.0;
        if (ql > 65535.0) ql = 65535.0;
        if (qh < 0.0) qh = 0.0;
        if (qh > 65535.0) qh = 65535.0;
        while (ql > 0.0 && g + ql * st > (double)lo[k]) ql -= 1.0;
        while (qh < 65535.0 && g + qh * st < (double)hi[k]) qh += 1.0;
        q[k] = (uint16_t)ql; q[3 + k] = (uint16_t)qh;
    }
}

/* Which binary nodes become 4-wide nodes.  A ray pays one record per 4-wide node whose box it enters, so the expected cost of
 * a collapse is the sum of the surface areas of the binary nodes that are kept as 4-wide nodes (leaves are fixed by the
 * binary build).  cost[n][k-1] = the cheapest way to hang subtree n into k free child slots of its 4-wide parent: either
 * as one 4-wide node (area(n) + the best split of ITS four slots over its two children), or dissolved into its two
 * children with the k slots split i : k - i.  (Ylitie et al. 2017 do the same for 8-wide nodes.)  The greedy rule used
 * before -- open the child of largest area until four slots are full -- is what LH_COLLAPSE=greedy still selects. */
typedef struct { double cost[8]; uint8_t split[8]; } dp4_t;      /* split[k-1]: 0 = a node of its own, i = left subtree gets i slots (widths 4 and 8) */

static double dp_child(const dp4_t *dp, int32_t ref, int k) { return ref >= 0 ? dp[ref].cost[k - 1] : 0.0; }

static int dp4_fill(const lh_bvh_t *o, dp4_t *dp, const int W)
{
    uint32_t ii;
    for (ii = o->nnodes; ii-- > 0;) {              /* children sit behind their parent in the flat tree */
        const lh_node_t *nd = &o->nodes[ii];
        float lo[3], hi[3]; int k, i;
        double g[9], area; uint8_t gi[9];
        if ((nd->ref0 >= 0 && (uint32_t)nd->ref0 <= ii) || (nd->ref1 >= 0 && (uint32_t)nd->ref1 <= ii)) return -1;
        for (k = 0; k < 3; k++) { lo[k] = nd->lo0[k]; hi[k] = nd->hi0[k]; }
        if (nd->ref1 != LH_REF_EMPTY)
            for (k = 0; k < 3; k++) { if (nd->lo1[k] < lo[k]) lo[k] = nd->lo1[k]; if (nd->hi1[k] > hi[k]) hi[k] = nd->hi1[k]; }
        area = (double)area3(lo, hi);
        if (nd->ref1 == LH_REF_EMPTY) {             /* a single child (one-leaf scenes): nothing to decide */
            for (k = 1; k <= W; k++) { dp[ii].cost[k - 1] = area + dp_child(dp, nd->ref0, W); dp[ii].split[k - 1] = 0; }
            continue;
        }
        for (k = 2; k <= W; k++) {
            g[k] = 1e300; gi[k] = 1;
            for (i = 1; i < k; i++) {
                const double c = dp_child(dp, nd->ref0, i) + dp_child(dp, nd->ref1, k - i);
                if (c < g[k]) { g[k] = c; gi[k] = (uint8_t)i; }
            }
        }
        dp[ii].cost[0] = area + g[W];
        for (k = 2; k <= W; k++) {
            if (dp[ii].cost[0] <= g[k]) { dp[ii].cost[k - 1] = dp[ii].cost[0]; dp[ii].split[k - 1] = 0; }
            else { dp[ii].cost[k - 1] = g[k]; dp[ii].split[k - 1] = gi[k]; }
        }
        /* the split of this node's own W slots, used when it IS a wide node */
        dp[ii].split[0] = gi[W];
    }
    return 0;
}

/* children of a wide node: subtree `ref` (box lo/hi, as stored in its binary parent) into k slots */
static void dp4_emit(const lh_bvh_t *o, const dp4_t *dp, const float *lo, const float *hi, int32_t ref, int k, child4_t *ch, int *n)
{
    if (ref >= 0 && k >= 2 && dp[ref].split[k - 1] != 0) {
        const lh_node_t *g = &o->nodes[ref];
        const int i = dp[ref].split[k - 1];
        dp4_emit(o, dp, g->lo0, g->hi0, g->ref0, i, ch, n);
        dp4_emit(o, dp, g->lo1, g->hi1, g->ref1, k - i, ch, n);
        return;
    }
    ch[*n].lo = lo; ch[*n].hi = hi; ch[*n].ref = ref; (*n)++;
}

/* above: the stack entries a ray can hold when it steps at this node = the sum over its ancestors of (children - 1); the
 * largest such sum + 5 (sentinel, the step's own writes) is the LDS rows the walk needs on this tree (q4_stack) */
static void build4(lh_bvh_t *o, const dp4_t *dp, uint32_t i2, uint32_t k4, uint32_t depth, uint32_t above, uint32_t *next)
{
    child4_t ch[4]; int n = 2, c, k; uint32_t kid[4];
    const lh_node_t *nd = &o->nodes[i2];
    if (dp && nd->ref1 != LH_REF_EMPTY) {
        const int i = dp[i2].split[0];
        n = 0;
        dp4_emit(o, dp, nd->lo0, nd->hi0, nd->ref0, i, ch, &n);
        dp4_emit(o, dp, nd->lo1, nd->hi1, nd->ref1, 4 - i, ch, &n);
    } else {
        ch[0].lo = nd->lo0; ch[0].hi = nd->hi0; ch[0].ref = nd->ref0;
        ch[1].lo = nd->lo1; ch[1].hi = nd->hi1; ch[1].ref = nd->ref1;
        if (ch[1].ref == LH_REF_EMPTY) n = 1;
        while (n < 4) {
            int best = -1; float ba = -1.0f;
            for (c = 0; c < n; c++)
                if (ch[c].ref >= 0) { float a = area3(ch[c].lo, ch[c].hi); if (a > ba) { ba = a; best = c; } }
            if (best < 0) break;
            {
                const lh_node_t *g = &o->nodes[ch[best].ref];
                ch[best].lo = g->lo0; ch[best].hi = g->hi0; ch[best].ref = g->ref0;
                ch[n].lo = g->lo1; ch[n].hi = g->hi1; ch[n].ref = g->ref1;
                n++;
            }
        }
    }
    if (depth + 1 > o->q4_depth) o->q4_depth = depth + 1;
    if (above + 5 > o->q4_stack) o->q4_stack = above + 5;
    for (c = 0; c < n; c++) if (ch[c].ref >= 0) kid[c] = (*next)++;     /* inner children adjacent */
    {
        lh_q4node_t *q = &o->q4nodes[k4];
        for (c = 0; c < 4; c++) {
            if (c < n) {
                uint16_t qb[6];
                quant_box(o, ch[c].lo, ch[c].hi, qb);
                for (k = 0; k < 3; k++) q->w[c][k] = (uint32_t)qb[k] | ((uint32_t)qb[3 + k] << 16);
                q->ref[c] = (ch[c].ref >= 0) ? (int32_t)kid[c] : ch[c].ref;
            } else {
                for (k = 0; k < 3; k++) q->w[c][k] = 65535u;          /* lo = 65535, hi = 0: inverted */
                q->ref[c] = LH_REF_EMPTY;
            }
        }
    }
    for (c = 0; c < n; c++) if (ch[c].ref >= 0) build4(o, dp, (uint32_t)ch[c].ref, kid[c], depth + 1, above + (uint32_t)(n - 1), next);
}

/* The 4-wide nodes in LEVEL order (breadth-first, a node's inner children still adjacent) instead of build4's depth-first
 * order: the rays of a wave are at similar depths at the same time, and the upper levels -- what every ray reads -- become a
 * dense prefix of the array instead of being scattered over it with whole subtrees in between.  (The device builder emits
 * level order by construction; on its trees depth-first order of the sibling groups cost config 5 +3 %.)  LH_Q4_LAYOUT=dfs
 * keeps build4's order. */
static int q4_level_order(lh_bvh_t *o)
{
    const uint32_t n = o->nq4nodes, cap = o->nnodes > n ? o->nnodes : n; uint32_t head = 0, tail = 0, i, real = 0; int c;
    uint32_t *order, *newidx; lh_q4node_t *nq;
    /* a sibling group of two or more starts at an even index: two 64-byte nodes share a 128-byte line (what the L2 fetches from
     * the fabric), so a group of two costs one line instead of possibly two, a group of four two instead of possibly three; the
     * skipped slots hold empty nodes nobody refers to.  S-soup-1M 2 232 -> 2 264 Mrays/s, config 5 85.0 -> 84.9 ms.
     * LH_Q4_PAIRS=0: dense. */
    const char *pe = getenv("LH_Q4_PAIRS"); const int pairs = !(pe && atoi(pe) == 0);
    if (n < 3) return 0;
    order = (uint32_t *)malloc(sizeof(uint32_t) * cap); newidx = (uint32_t *)malloc(sizeof(uint32_t) * n);
    nq = (lh_q4node_t *)malloc(sizeof(lh_q4node_t) * cap);
    if (!order || !newidx || !nq) { free(order); free(newidx); free(nq); return -1; }
    order[tail++] = 0;
    while (head < tail) {
        const uint32_t old = order[head];
        int m = 0;
        if (old == 0xffffffffu) { head++; continue; }
        newidx[old] = head++; real++;
        for (c = 0; c < 4; c++) m += o->q4nodes[old].ref[c] >= 0;
        if (pairs && m >= 2 && (tail & 1u) && tail + (uint32_t)m < cap) order[tail++] = 0xffffffffu;
        for (c = 0; c < 4; c++) { const int32_t r = o->q4nodes[old].ref[c]; if (r >= 0 && tail < cap) order[tail++] = (uint32_t)r; }
    }
    if (real != n) { free(order); free(newidx); free(nq); return 0; }          /* not a tree over all nodes (or no room for the pads): leave it */
    for (i = 0; i < tail; i++) {
        if (order[i] == 0xffffffffu) {
            int k;
            for (c = 0; c < 4; c++) { for (k = 0; k < 3; k++) nq[i].w[c][k] = 65535u; nq[i].ref[c] = LH_REF_EMPTY; }
            continue;
        }
        nq[i] = o->q4nodes[order[i]];
        for (c = 0; c < 4; c++) if (nq[i].ref[c] >= 0) nq[i].ref[c] = (int32_t)newidx[nq[i].ref[c]];
    }
    memcpy(o->q4nodes, nq, sizeof(lh_q4node_t) * tail);
    o->nq4nodes = tail;
    free(order); free(newidx); free(nq);
    return 0;
}

static int collapse4(lh_bvh_t *o)
{
    uint32_t next = 1;
    dp4_t *dp = NULL;
    const char *e = getenv("LH_COLLAPSE");
    o->q4nodes = (lh_q4node_t *)calloc(o->nnodes ? o->nnodes : 1, sizeof(lh_q4node_t));   /* <= nnodes records */
    if (!o->q4nodes) return -1;
    if (!(e && !strcmp(e, "greedy")) && o->nnodes) {
        dp = (dp4_t *)malloc(sizeof(dp4_t) * (size_t)o->nnodes);
        if (!dp) return -1;
        if (dp4_fill(o, dp, 4) != 0) { free(dp); dp = NULL; }      /* not a parent-first order: the greedy rule */
    }
    o->q4_depth = 0; o->q4_stack = 0;
    build4(o, dp, 0, 0, 0, 0, &next);
    o->nq4nodes = next;
    free(dp);
    { const char *l = getenv("LH_Q4_LAYOUT"); if (!(l && !strcmp(l, "dfs")) && q4_level_order(o) != 0) return -1; }
    return 0;
}

/* ---- 8-wide collapse onto the 16-bit grid (see lh_q8node_t): one 128-byte record = one cache line -------------- */
static void build8q(lh_bvh_t *o, const dp4_t *dp, uint32_t i2, uint32_t k8, uint32_t depth, uint32_t *next)
{
    child4_t ch[8]; int n = 0, c, k, s; uint32_t kid[8]; int slot_of[8], used[8];
    const lh_node_t *nd = &o->nodes[i2];
    float nlo[3], nhi[3]; double cen[3];
    lh_q8node_t *q = &o->q8nodes[k8];
    if (nd->ref1 != LH_REF_EMPTY) {
        const int i = dp[i2].split[0];
        dp4_emit(o, dp, nd->lo0, nd->hi0, nd->ref0, i, ch, &n);
        dp4_emit(o, dp, nd->lo1, nd->hi1, nd->ref1, 8 - i, ch, &n);
    } else { ch[0].lo = nd->lo0; ch[0].hi = nd->hi0; ch[0].ref = nd->ref0; n = 1; }
    if (depth + 1 > o->q8_depth) o->q8_depth = depth + 1;
    for (k = 0; k < 3; k++) {
        nlo[k] = ch[0].lo[k]; nhi[k] = ch[0].hi[k];
        for (c = 1; c < n; c++) { if (ch[c].lo[k] < nlo[k]) nlo[k] = ch[c].lo[k]; if (ch[c].hi[k] > nhi[k]) nhi[k] = ch[c].hi[k]; }
        cen[k] = 0.5 * ((double)nlo[k] + (double)nhi[k]);
    }
    /* octant slots (as for the compressed 8-wide node): the walk visits slot s with priority s ^ (ray octant), so a child
     * goes to the free slot whose diagonal its centroid offset points along most */
    for (s = 0; s < 8; s++) used[s] = 0;
    for (c = 0; c < n; c++) slot_of[c] = -1;
    for (k = 0; k < n; k++) {
        int bc = -1, bs = -1; double bv = -1.0e300;
        for (c = 0; c < n; c++) {
            double d[3]; int a;
            if (slot_of[c] >= 0) continue;
            for (a = 0; a < 3; a++) d[a] = 0.5 * ((double)ch[c].lo[a] + (double)ch[c].hi[a]) - cen[a];
            for (s = 0; s < 8; s++) {
                double v;
                if (used[s]) continue;
                v = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 4) ? d[2] : -d[2]);
                if (v > bv) { bv = v; bc = c; bs = s; }
            }
        }
        slot_of[bc] = bs; used[bs] = 1;
    }
    for (s = 0; s < 8; s++) { for (k = 0; k < 3; k++) q->w[s][k] = 65535u; q->ref[s] = LH_REF_EMPTY; }     /* lo = 65535, hi = 0: inverted */
    for (s = 0; s < 8; s++)                       /* inner children adjacent, in slot order */
        for (c = 0; c < n; c++)
            if (slot_of[c] == s) {
                uint16_t qb[6];
                quant_box(o, ch[c].lo, ch[c].hi, qb);
                for (k = 0; k < 3; k++) q->w[s][k] = (uint32_t)qb[k] | ((uint32_t)qb[3 + k] << 16);
                if (ch[c].ref >= 0) { kid[c] = (*next)++; q->ref[s] = (int32_t)kid[c]; } else q->ref[s] = ch[c].ref;
            }
    for (c = 0; c < n; c++) if (ch[c].ref >= 0) build8q(o, dp, (uint32_t)ch[c].ref, kid[c], depth + 1, next);
}

/* built the first time a large scene's ray dump asks for it (not thread-safe: callers lock); needs the binary nodes */
/* the 8-wide nodes in level order (see q4_level_order) */
static int q8_level_order(lh_bvh_t *o)
{
    const uint32_t n = o->nq8nodes; uint32_t head = 0, tail = 0, i; int c;
    uint32_t *order, *newidx; lh_q8node_t *nq;
    if (n < 3) return 0;
    order = (uint32_t *)malloc(sizeof(uint32_t) * n); newidx = (uint32_t *)malloc(sizeof(uint32_t) * n);
    nq = (lh_q8node_t *)malloc(sizeof(lh_q8node_t) * n);
    if (!order || !newidx || !nq) { free(order); free(newidx); free(nq); return -1; }
    order[tail++] = 0;
    while (head < tail) {
        const uint32_t old = order[head];
        newidx[old] = head++;
        for (c = 0; c < 8; c++) { const int32_t r = o->q8nodes[old].ref[c]; if (r >= 0 && tail < n) order[tail++] = (uint32_t)r; }
    }
    if (tail != n) { free(order); free(newidx); free(nq); return 0; }
    for (i = 0; i < n; i++) {
        nq[i] = o->q8nodes[order[i]];
        for (c = 0; c < 8; c++) if (nq[i].ref[c] >= 0) nq[i].ref[c] = (int32_t)newidx[nq[i].ref[c]];
    }
    memcpy(o->q8nodes, nq, sizeof(lh_q8node_t) * n);
    free(order); free(newidx); free(nq);
    return 0;
}

int lh_bvh_ensure_q8(lh_bvh_t *o)
{
    uint32_t next = 1;
    dp4_t *dp;
    if (o->q8nodes || o->ntris == 0) return 0;
    if (!o->nodes) return -1;
    o->q8nodes = (lh_q8node_t *)calloc(o->nnodes ? o->nnodes : 1, sizeof(lh_q8node_t));
    dp = (dp4_t *)malloc(sizeof(dp4_t) * (size_t)(o->nnodes ? o->nnodes : 1));
    if (!o->q8nodes || !dp || dp4_fill(o, dp, 8) != 0) { free(dp); free(o->q8nodes); o->q8nodes = NULL; return -1; }
    o->q8_depth = 0;
    build8q(o, dp, 0, 0, 0, &next);
    o->nq8nodes = next;
    free(dp);
    {
        lh_q8node_t *sh = (lh_q8node_t *)realloc(o->q8nodes, sizeof(lh_q8node_t) * (size_t)next);
        if (sh) o->q8nodes = sh;
    }
    { const char *l = getenv("LH_Q4_LAYOUT"); if (!(l && !strcmp(l, "dfs")) && q8_level_order(o) != 0) { free(o->q8nodes); o->q8nodes = NULL; return -1; } }
    return 0;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

typedef struct { const lh_mesh_view_t *m; uint32_t g, p0; lh_bvh_t *out; build_ctx_t *b; volatile int err; volatile int dead, dead_noise;
                 double s2keep[LH_POOL_MAX]; } prep_job_t;          /* s2keep[t]: thread t's largest zero-area triangle that stays in the tree (tri_zero_area_s2) */

/* Triangles the reference can NEVER report (triangle_isect, bvh.c:730-791: `if (fabs(a) > 1e-14) ... else return 0`, with
 * a = e1 . (dir x e2) in fp64, no FMA), decided from the vertices alone -- they stay out of the TRAVERSAL tree (lucille's own tree,
 * lh_refbvh.c, keeps them: its shape depends on them).  Why it matters: midpoint-tessellating a mesh with a collapsed edge (the ten
 * apex "quads" of the example scene's cone) leaves 65 536 zero-area triangles along ONE segment per source triangle -- coincident
 * boxes, hundreds of records per ray that comes near, and every one of them through the fp64 test, because an fp32 determinant
 * of zero decides nothing (round 5: 2 000 pixels of the config-5 frame cost 100-1200 triangle records per AO ray, against 2.8).
 *   class 1  v0 == v1 (e1 = 0): a = 0 * px + 0 * py + 0 * pz = 0 (or NaN if p overflowed): never > 1e-14.  Any ray.
 *            v0 == v2 (e2 = 0): p = dir x 0 = 0 (NaN for an infinite dir), a = e1 . 0 = 0: never > 1e-14.  Any ray.
 *   class 2  v1 == v2 (e1 == e2 =: e): a is rounding noise around the exact e . (dir x e) = 0.  With u = 2^-53, D = max |dir_k|,
 *            S = |ex| + |ey| + |ez|: every component of the computed p is off by <= 2u(1+u) D S, so e . p_computed is off by
 *            <= 2u(1+u) D S^2, and the three-term dot product adds <= 3u/(1-3u) (1+u)^2 D S^2: |a| < 5.01 u D S^2.  Dropped when
 *            S^2 <= LH_DEG_S2CAP, so that |a| <= 1e-15 D S^2 <= 1e-14 for every ray with D <= LH_DEG_DCAP = 1024; a ray with a
 *            larger direction component (unnormalised directions are legal, ray.h:22-68) is decided by the reference's own walk
 *            on its own tree (lh_dev_scene_t.deg_dcap; lh_kernels.hip).
 * Equality is numeric (== on doubles): +0 and -0 give differences of +-0, which the argument covers. */
#define LH_DEG_DCAP  1024.0
#define LH_DEG_S2CAP (1.0e-14 / (1.0e-15 * LH_DEG_DCAP))
static int tri_dead_class(const lh_tri64_t *t)
{
    const double *a = t->v[0], *b = t->v[1], *c = t->v[2];
    if ((a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) || (a[0] == c[0] && a[1] == c[1] && a[2] == c[2])) return 1;
    if (b[0] == c[0] && b[1] == c[1] && b[2] == c[2]) {
        const double s = fabs(b[0] - a[0]) + fabs(b[1] - a[1]) + fabs(b[2] - a[2]);
        if (s * s * (1.0 + 1e-9) <= LH_DEG_S2CAP) return 2;
    }
    return 0;
}

/* A zero-area triangle that STAYS in the tree (v1 == v2 beyond LH_DEG_S2CAP; three different points on one line; builds with
 * LH_DROP_DEGENERATE=0): the reference's determinant for it is rounding noise, and that noise clears 1e-14 once D S^2 is large
 * enough -- for ANY ray that reaches the triangle's leaf in lucille's own tree, whether it comes near the triangle or not (its
 * leaves hold several triangles; the traversal tree's boxes would never lead there).  Found by tools/fuzz_parity.py: a scene of
 * scale 28 with zero-area triangles of 17 units, directions of 1e5: 174 of 52 000 rays "hit" one at t = +-0.  So such triangles
 * bound deg_dcap too.  Numerically zero-area: the computed normal n = e1 x e2 is no larger than its own rounding error,
 * max |n_k| <= 8u s2 with s2 = |e1|_1 |e2|_1 (the exact normal is then below ~12u s2); the reference's
 * a = e1 . (dir x e2) = dir . n is below 3 D 12u s2 exact + ~15u D s2 of evaluation noise < 6e-15 D s2: never above 1e-14 while
 * D <= 1 / s2.  Returns s2 (0: the triangle has area, or an edge of length zero -- class 1, exactly a = 0). */
static double tri_zero_area_s2(const lh_tri64_t *t)
{
    return lh_zero_area_weight(t->v[0], t->v[1], t->v[2]);          /* round 6: lh_bvh.h -- also near-collinear triangles, with the bound that fits them */
}

/* triangles [a0, a1) of one mesh: fp64 vertices, primitive -> (geom, index), fp32 outward box, centroid */
static void prep_part(void *j_, int t, int nt)
{
    prep_job_t *j = (prep_job_t *)j_; const lh_mesh_view_t *m = j->m; build_ctx_t *b = j->b; lh_bvh_t *out = j->out;
    uint32_t a0, a1, i; int c, k;
    chunk_of(0, m->nindices / 3, t, nt, &a0, &a1);
    for (i = a0; i < a1; i++) {
        const uint32_t p = j->p0 + i;
        lh_tri64_t *tr = &out->tri64[p];
        for (c = 0; c < 3; c++) {
            const uint32_t vi = m->indices[3 * (size_t)i + c];
            const double *P;
            if (vi >= m->npositions) { j->err = -1; return; }
            P = (const double *)((const char *)m->positions + (size_t)vi * m->stride_bytes);
            for (k = 0; k < 3; k++) {
                /* NaN, inf or beyond what the fp32 filter can bound: refuse the scene (return -2) */
                if (!(fabs(P[k]) <= 1.0e30)) { j->err = -2; return; }
                tr->v[c][k] = P[k];
            }
        }
        out->prim_geom[p] = j->g; out->prim_index[p] = 3 * i;
        for (k = 0; k < 3; k++) {
            double lo = tr->v[0][k], hi = tr->v[0][k];
            if (tr->v[1][k] < lo) lo = tr->v[1][k];
            if (tr->v[2][k] < lo) lo = tr->v[2][k];
            if (tr->v[1][k] > hi) hi = tr->v[1][k];
            if (tr->v[2][k] > hi) hi = tr->v[2][k];
            b->plo[3 * (size_t)p + k] = down32(lo); b->phi[3 * (size_t)p + k] = up32(hi);
            b->cen[3 * (size_t)p + k] = (float)(0.5 * (lo + hi));
        }
        b->order[p] = p;
        {
            const int dc = b->drop_dead ? tri_dead_class(tr) : 0;
            if (dc) { b->order[p] = 0xFFFFFFFFu; j->dead = 1; if (dc == 2) j->dead_noise = 1; }
            else { const double s2 = tri_zero_area_s2(tr); if (s2 > j->s2keep[t]) j->s2keep[t] = s2; }
        }
    }
}

int lh_bvh_build(lh_bvh_t *out, const lh_mesh_view_t *meshes, uint32_t nmeshes, int nthreads)
{
    return lh_bvh_build_hook(out, meshes, nmeshes, nthreads, NULL, NULL);
}

/* after_flatten (may be NULL) is called once tri64 / prim_geom / prim_index are complete and will not move: the caller can
 * start work that only needs the flattened triangles (lucille's own tree, lh_refbvh.c) next to the tree build */
int lh_bvh_build_hook(lh_bvh_t *out, const lh_mesh_view_t *meshes, uint32_t nmeshes, int nthreads,
                      void (*after_flatten)(void *), void *hook_arg)
{
    uint64_t n64 = 0; uint32_t g, n; int k;
    build_ctx_t b; arena_t main_arena; tnode_t *root;
    double t0 = now_s();
    const char *env;

    memset(out, 0, sizeof(*out));
    out->deg_dcap = INFINITY;
    for (g = 0; g < nmeshes; g++) {
        const lh_mesh_view_t *m = &meshes[g];
        if (m->nindices && (!m->indices || !m->positions)) return -1;
        n64 += m->nindices / 3;
    }
    if (n64 >= (1u << 29)) return -1;   /* leaf reference encoding limit */
    n = (uint32_t)n64;
    out->ntris = n;
    if (n == 0) return 0;                /* empty scene: always-miss accel (bvh.c:311-315) */

    out->tri64 = (lh_tri64_t *)malloc(sizeof(lh_tri64_t) * n);
    out->tri32 = (lh_tri32_t *)malloc(sizeof(lh_tri32_t) * n);
    out->prim_geom = (uint32_t *)malloc(sizeof(uint32_t) * n);
    out->prim_index = (uint32_t *)malloc(sizeof(uint32_t) * n);
    memset(&b, 0, sizeof(b));
    b.n = n;
    b.plo = (float *)malloc(sizeof(float) * 3 * (size_t)n);
    b.phi = (float *)malloc(sizeof(float) * 3 * (size_t)n);
    b.cen = (float *)malloc(sizeof(float) * 3 * (size_t)n);
    b.order = (uint32_t *)malloc(sizeof(uint32_t) * n);
    if (!out->tri64 || !out->tri32 || !out->prim_geom || !out->prim_index || !b.plo || !b.phi || !b.cen || !b.order) {
        free(b.plo); free(b.phi); free(b.cen); free(b.order); lh_bvh_release(out); return -1;
    }

    /* flatten in create_triangle_list order: primitive id = running index (the meshes' triangles cut into chunks for the pool) */
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (nthreads > 1 && n > 100000) b.pool = tpool_new(nthreads);
    b.drop_dead = !((env = getenv("LH_DROP_DEGENERATE")) != NULL && atoi(env) == 0);
    out->nlive = n; out->deg_dcap = INFINITY;
    {
        uint32_t p = 0; int any_dead = 0, any_noise = 0; double s2keep = 0.0;
        for (g = 0; g < nmeshes; g++) {
            prep_job_t pj; int tk;
            pj.m = &meshes[g]; pj.g = g; pj.p0 = p; pj.out = out; pj.b = &b; pj.err = 0; pj.dead = 0; pj.dead_noise = 0;
            for (tk = 0; tk < LH_POOL_MAX; tk++) pj.s2keep[tk] = 0.0;
            if (b.pool && pj.m->nindices / 3 >= LH_PAR_MIN) tpool_run(b.pool, prep_part, &pj);
            else prep_part(&pj, 0, 1);
            if (pj.err) { tpool_free(b.pool); free(b.plo); free(b.phi); free(b.cen); free(b.order); lh_bvh_release(out); return pj.err; }
            p += pj.m->nindices / 3;
            any_dead |= pj.dead; any_noise |= pj.dead_noise;
            for (tk = 0; tk < LH_POOL_MAX; tk++) if (pj.s2keep[tk] > s2keep) s2keep = pj.s2keep[tk];
        }
        if (any_dead) {
            /* the tree is built over the live primitives: order[0 .. nlive); the dead ones keep their ids, their tri64 records
             * (lucille's own tree and the reference walk read those) and the unreferenced tail of tri32 */
            uint32_t m = 0, q;
            for (q = 0; q < n; q++) if (b.order[q] != 0xFFFFFFFFu) b.order[m++] = q;
            if (m == 0) {                                                                       /* nothing but zero-area triangles: as handed over */
                for (q = 0; q < n; q++) { const double s2 = tri_zero_area_s2(&out->tri64[q]); b.order[q] = q; if (s2 > s2keep) s2keep = s2; }
                m = n; any_noise = 0;
            }
            else { uint32_t w = m; for (q = 0; q < n; q++) if (tri_dead_class(&out->tri64[q])) b.order[w++] = q; }
            out->nlive = m; b.n = m;
            if (any_noise) out->deg_dcap = LH_DEG_DCAP;
        }
        /* zero-area triangles in the tree: rays beyond 1 / s2 are decided by the reference's own walk (tri_zero_area_s2) */
        if (s2keep > 0.0 && 1.0 / s2keep < out->deg_dcap) out->deg_dcap = 1.0 / s2keep;
    }

    if (after_flatten) after_flatten(hook_arg);

    b.ci = 1.0f; b.ct = 1.3f;          /* a triangle step against a node step (tools/experiments/host_sah_probe.py, r03: S-soup-1M 2110 -> 2132 Mrays/s with 4.6 instead of
                                          5.1 triangle tests per ray; config 5 unchanged) */
    if ((env = getenv("LH_BVH_CI")) != NULL) b.ci = (float)atof(env);
    if ((env = getenv("LH_BVH_CT")) != NULL) b.ct = (float)atof(env);
    memset(&main_arena, 0, sizeof(main_arena));
    root = arena_new(&main_arena);
    if (nthreads > 1 && n > 100000) {
        b.tmp = (uint32_t *)malloc(sizeof(uint32_t) * n);       /* NULL: the top of the tree partitions serially */
        b.collecting = 1;
        b.task_threshold = out->nlive / (uint32_t)(nthreads * 8);
        if (b.task_threshold < 4096) b.task_threshold = 4096;
        /* every range above the threshold is processed by the pool: no serial pass over more than LH_PAR_MIN primitives */
        if (b.pool && b.task_threshold < LH_PAR_MIN && n / LH_PAR_MIN >= (uint32_t)(2 * nthreads)) b.task_threshold = LH_PAR_MIN;
    }
    double t_prep = now_s();
    build_range(&b, &main_arena, root, 0, out->nlive, 0);
    b.collecting = 0;
    tpool_free(b.pool); b.pool = NULL; free(b.tmp); b.tmp = NULL;
    double t_top = now_s();

    {
        worker_t *w = NULL; pthread_t *th = NULL; volatile uint32_t next = 0; int t;
        if (b.ntasks) {
            w = (worker_t *)calloc((size_t)nthreads, sizeof(*w));
            th = (pthread_t *)calloc((size_t)nthreads, sizeof(*th));
            for (t = 0; t < nthreads; t++) { w[t].b = &b; w[t].next = &next; pthread_create(&th[t], NULL, worker_main, &w[t]); }
            for (t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        }

        double t_par = now_s();
        if (getenv("LH_BUILD_TIMING")) fprintf(stderr, "[lh_bvh] prep %.3f top %.3f subtrees %.3f (tasks %zu)\n", t_prep - t0, t_top - t_prep, t_par - t_top, b.ntasks);
        /* flatten */
        {
            flat_t f; uint32_t ninner;
            memset(&f, 0, sizeof(f));
            f.b = &b; f.out = out; f.tri64 = out->tri64;
            if (root->count) {   /* single-leaf scene: synthesize an inner root */
                out->nodes = (lh_node_t *)calloc(1, sizeof(lh_node_t));
                out->nnodes = 1;
                for (k = 0; k < 3; k++) { out->nodes[0].lo0[k] = root->lo[k]; out->nodes[0].hi0[k] = root->hi[k]; out->nodes[0].lo1[k] = INFINITY; out->nodes[0].hi1[k] = -INFINITY; }
                out->nodes[0].ref0 = emit_leaf(&f, root);
                out->nodes[0].ref1 = LH_REF_EMPTY;
            } else {
                ninner = count_inner_top(&b, root);
                out->nodes = (lh_node_t *)calloc(ninner, sizeof(lh_node_t));
                out->nnodes = ninner;
                f.next_node = 1;
                f.defer = (b.ntasks > 0 && nthreads > 1);
                emit_inner(&f, root, 0, 0);
                if (f.defer) {
                    /* the subtrees, each into its own block of the layout */
                    flat_worker_t fw[LH_POOL_MAX]; pthread_t fth[LH_POOL_MAX]; volatile uint32_t fnext = 0; int t, nt = nthreads > LH_POOL_MAX ? LH_POOL_MAX : nthreads, started = 0;
                    size_t q;
                    for (t = 0; t < nt; t++) {
                        fw[t].base = f; fw[t].b = &b; fw[t].next = &fnext;
                        if (pthread_create(&fth[t], NULL, flatten_worker, &fw[t]) != 0) break;
                        started++;
                    }
                    if (started == 0) flatten_worker(&fw[0]);
                    for (t = 0; t < started; t++) pthread_join(fth[t], NULL);
                    for (q = 0; q < b.ntasks; q++) {
                        if (b.tasks[q].max_depth > f.max_depth) f.max_depth = b.tasks[q].max_depth;
                        f.nleaves += b.tasks[q].nleaves;
                    }
                }
            }
            out->max_depth = f.max_depth + 1; out->nleaves = f.nleaves;
            for (k = 0; k < 3; k++) { out->bmin[k] = root->lo[k]; out->bmax[k] = root->hi[k]; }
        }

        if (w) { int t2; for (t2 = 0; t2 < nthreads; t2++) arena_free(&w[t2].arena); }
        free(w); free(th);
    }
    arena_free(&main_arena);
    { uint32_t q; for (q = out->nlive; q < n; q++) make_tri32(&out->tri32[q], &out->tri64[b.order[q]], b.order[q]); }     /* the dropped triangles: records no leaf refers to */
    free(b.tasks); free(b.plo); free(b.phi); free(b.cen); free(b.order);
    {
        double t1 = now_s(), t2, t3;
        setup_grid(out);
        t2 = now_s();
        if (collapse4(out) != 0) { lh_bvh_release(out); return -1; }
        t3 = now_s();
        if (getenv("LH_BUILD_TIMING")) fprintf(stderr, "[lh_bvh] flatten+free ..%.3f quantize %.3f collapse4 %.3f \n", t1 - t0, t2 - t1, t3 - t2);
    }
    out->build_seconds = now_s() - t0;
    return 0;
}

/* primitive flattening only (create_triangle_list order: tri64, prim_geom, prim_index), no tree: the input of the
 * device builder (lh_build.hip).  Same return codes as lh_bvh_build. */
/* primitives in lucille's order (create_triangle_list, bvh.c:1736-1826: geoms in list order, triangles in index order), no
 * tree.  The device-side commit waits for exactly this (0.33 s single-threaded for 21 M triangles), so the copy is cut into
 * ranges of the global primitive order, one thread each */
typedef struct {
    lh_bvh_t *out; const lh_mesh_view_t *meshes; const uint32_t *first_prim; uint32_t nmeshes, p0, p1; int rc, started;
} soup_job_t;

static void *soup_worker(void *arg)
{
    soup_job_t *j = (soup_job_t *)arg;
    uint32_t g = 0, p; int c, k;
    while (g + 1 < j->nmeshes && j->first_prim[g + 1] <= j->p0) g++;
    for (p = j->p0; p < j->p1; p++) {
        const lh_mesh_view_t *m; uint32_t i; lh_tri64_t *t = &j->out->tri64[p];
        while (g + 1 < j->nmeshes && j->first_prim[g + 1] <= p) g++;
        m = &j->meshes[g]; i = p - j->first_prim[g];
        for (c = 0; c < 3; c++) {
            const uint32_t vi = m->indices[3 * i + c]; const double *P;
            if (vi >= m->npositions) { j->rc = -1; return NULL; }
            P = (const double *)((const char *)m->positions + (size_t)vi * m->stride_bytes);
            for (k = 0; k < 3; k++) {
                if (!(fabs(P[k]) <= 1.0e30)) { j->rc = -2; return NULL; }
                t->v[c][k] = P[k];
            }
        }
        j->out->prim_geom[p] = g; j->out->prim_index[p] = 3 * i;
    }
    return NULL;
}

int lh_bvh_flatten(lh_bvh_t *out, const lh_mesh_view_t *meshes, uint32_t nmeshes, int nthreads)
{
    uint64_t n64 = 0; uint32_t g, *first_prim; int t, nt, rc = 0;
    soup_job_t *jobs; pthread_t *th;
    memset(out, 0, sizeof(*out));
    out->deg_dcap = INFINITY;
    for (g = 0; g < nmeshes; g++) {
        const lh_mesh_view_t *m = &meshes[g];
        if (m->nindices && (!m->indices || !m->positions)) return -1;
        n64 += m->nindices / 3;
    }
    if (n64 >= (1u << 29)) return -1;
    out->ntris = (uint32_t)n64; out->nlive = (uint32_t)n64;
    if (n64 == 0) return 0;
    out->tri64 = (lh_tri64_t *)malloc(sizeof(lh_tri64_t) * (size_t)n64);
    out->prim_geom = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n64);
    out->prim_index = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n64);
    first_prim = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)nmeshes + 1));
    nt = nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads);
    if ((uint64_t)nt * 65536u > n64) nt = (int)(n64 / 65536u) + 1;          /* a thread per >= 64 K primitives */
    jobs = (soup_job_t *)calloc((size_t)nt, sizeof(*jobs)); th = (pthread_t *)calloc((size_t)nt, sizeof(*th));
    if (!out->tri64 || !out->prim_geom || !out->prim_index || !first_prim || !jobs || !th) {
        free(first_prim); free(jobs); free(th); lh_bvh_release(out); return -1;
    }
    first_prim[0] = 0;
    for (g = 0; g < nmeshes; g++) first_prim[g + 1] = first_prim[g] + meshes[g].nindices / 3;
    for (t = 0; t < nt; t++) {
        jobs[t].out = out; jobs[t].meshes = meshes; jobs[t].first_prim = first_prim; jobs[t].nmeshes = nmeshes;
        jobs[t].p0 = (uint32_t)(n64 * (uint64_t)t / (uint64_t)nt); jobs[t].p1 = (uint32_t)(n64 * (uint64_t)(t + 1) / (uint64_t)nt);
    }
    for (t = 1; t < nt; t++) {
        jobs[t].started = pthread_create(&th[t], NULL, soup_worker, &jobs[t]) == 0;
        if (!jobs[t].started) soup_worker(&jobs[t]);                              /* no thread: this one does the range */
    }
    soup_worker(&jobs[0]);
    for (t = 1; t < nt; t++) if (jobs[t].started) pthread_join(th[t], NULL);
    for (t = 0; t < nt; t++) if (jobs[t].rc != 0 && (rc == 0 || jobs[t].rc == -1)) rc = jobs[t].rc;
    free(first_prim); free(jobs); free(th);
    if (rc != 0) lh_bvh_release(out);
    return rc;
}

void lh_bvh_release(lh_bvh_t *bvh)
{
    free(bvh->nodes); free(bvh->q4nodes); free(bvh->q8nodes); free(bvh->tri32); free(bvh->tri64); free(bvh->prim_geom); free(bvh->prim_index);
    memset(bvh, 0, sizeof(*bvh));
}
