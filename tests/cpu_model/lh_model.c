/*
 * lh_model.c -- HOST MODEL of the gfx950 traversal kernel (test infrastructure).
 *
 * Not part of the product and never loaded by it.  It runs, on the CPU, the
 * same algorithm as lucille_amd/csrc/lh_kernels.hip over the SAME flattened
 * BVH (lh_bvh.c) with the SAME arithmetic (lh_filter.h is shared source):
 * fp32 conservative slab test, fp32 tolerance-carrying Moeller-Trumbore
 * filter, pending list, certain-hit bound shrinking, fp64 resolve.  The
 * not-gpu tests use it to show, without a GPU, that the filter never loses a
 * hit the fp64 oracle finds, and to count node visits / triangle tests of the
 * product's own BVH for the roofline formula.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "lh_bvh.h"
#include "lh_refbvh.h"
#include "lh_filter.h"
#include "lh_reftrace.h"

#define MISS 0xFFFFFFFFu
#define DONE ((int32_t)0x80000000)
#define T_INF 1.0e38
#define PEND 4

typedef struct { double t, u, v; uint32_t prim; uint32_t frag; } best_t;

static const lh_refbvh_t *g_ref = NULL;    /* reference-order tree for exact-t ties (optional) */

static void resolve(const lh_bvh_t *b, uint32_t prim, const double *o, const double *d, best_t *best)
{
    double t, u, v;
    if (lh_exact_isect(&b->tri64[prim].v[0][0], o[0], o[1], o[2], d[0], d[1], d[2], &t, &u, &v)) {
        int take = t < best->t;
        if (best->prim != MISS && prim != best->prim && t != best->t && fabs(t - best->t) <= LH_FRAGILE_REL * fabs(t)) best->frag |= 2u;
        if (!take && t == best->t && best->prim != MISS && prim != best->prim) {
            if (g_ref) { int sg[3] = { d[0] < 0.0, d[1] < 0.0, d[2] < 0.0 }; take = lh_refbvh_tie_winner(g_ref, prim, best->prim, sg) == prim; }
            else take = prim > best->prim;
        }
        if (take && t < T_INF) {
            best->t = t; best->u = u; best->v = v; best->prim = prim;
            best->frag = (best->frag & 2u) | (uint32_t)lh_hit_fragile(&b->tri64[prim].v[0][0], o[0], o[1], o[2], d[0], d[1], d[2], t);
        }
    }
}

typedef struct {
    const lh_bvh_t *b; size_t begin, end; const double *org, *dir;
    uint32_t *prim; double *t, *u, *v; uint8_t *occ; int anyhit; int qnodes;
    uint64_t c[4];
    uint32_t *diag;            /* NULL, or n x 4: this ray's node visits, leaf visits, triangle records through the filter, fp64 tests */
} job_t;

static void trace_one(job_t *j, size_t i)
{
    const lh_bvh_t *b = j->b;
    const double *o = &j->org[3 * i], *d = &j->dir[3 * i];
    best_t best = { T_INF, 0.0, 0.0, MISS, 0u };
    int certain = 0;
    const uint64_t c0 = j->c[0], c1 = j->c[1], c2 = j->c[2]; uint32_t leaves = 0;
    j->c[3]++;
    if (b->ntris) {
        lh_ray32_t r; float tb = 1.0e38f, scene_r = 0.0f;
        int32_t stack[7 * LH_MAX_DEPTH + 16]; int sp = 1, cur = 0, np = 0, k;
        uint32_t pend[PEND];
        for (k = 0; k < 3; k++) { scene_r = fmaxf(scene_r, fabsf(b->bmin[k])); scene_r = fmaxf(scene_r, fabsf(b->bmax[k])); }
        lh_ray_setup(&r, o[0], o[1], o[2], d[0], d[1], d[2], scene_r);
        lh_ray_setup_grid(&r, b->grid_lo, b->grid_step, scene_r);
        stack[0] = DONE;
        while (cur != DONE) {
            while (cur >= 0 && j->qnodes == 4) {      /* 8-wide 16-bit grid nodes (lh_q8node_t), octant order */
                const lh_q8node_t *n = &b->q8nodes[cur]; int pr;
                const int oct = r.ngx | (r.ngy << 1) | (r.ngz << 2);
                j->c[0]++;
                for (pr = 7; pr >= 0; pr--) {              /* far to near: the nearest ends on top */
                    const int s = pr ^ oct; float tn;
                    if (n->ref[s] == DONE) continue;
                    if (!lh_slab_w(&r, n->w[s][0], n->w[s][1], n->w[s][2], tb, &tn)) continue;
                    stack[sp++] = n->ref[s];
                }
                cur = stack[--sp];
            }
            while (cur >= 0 && j->qnodes == 2) {      /* 4-wide 16-bit grid nodes */
                const lh_q4node_t *n = &b->q4nodes[cur]; float tn[4]; int h[4], c, nh = 0, order[4], m;
                j->c[0]++;
                for (c = 0; c < 4; c++) {
                    h[c] = lh_slab_w(&r, n->w[c][0], n->w[c][1], n->w[c][2], tb, &tn[c]) && n->ref[c] != DONE;
                    if (h[c]) order[nh++] = c;
                }
                for (c = 1; c < nh; c++) {                 /* nearest first (same key as the kernel: low 2 bits = slot) */
                    int x = order[c]; uint32_t kx, km; union { float f; uint32_t u; } cv;
                    cv.f = tn[x]; kx = (cv.u & ~3u) | (uint32_t)x;
                    for (m = c - 1; m >= 0; m--) { cv.f = tn[order[m]]; km = (cv.u & ~3u) | (uint32_t)order[m]; if (km <= kx) break; order[m + 1] = order[m]; }
                    order[m + 1] = x;
                }
                if (nh == 0) cur = stack[--sp];
                else { for (c = nh - 1; c >= 1; c--) stack[sp++] = n->ref[order[c]]; cur = n->ref[order[0]]; }
            }
            while (cur >= 0) {
                float tn0, tn1; int h0, h1; int32_t r0, r1;
                j->c[0]++;
                {
                    const lh_node_t *n = &b->nodes[cur];
                    h0 = lh_slab(&r, n->lo0[0], n->lo0[1], n->lo0[2], n->hi0[0], n->hi0[1], n->hi0[2], tb, &tn0);
                    h1 = lh_slab(&r, n->lo1[0], n->lo1[1], n->lo1[2], n->hi1[0], n->hi1[1], n->hi1[2], tb, &tn1);
                    r0 = n->ref0; r1 = n->ref1;
                }
                if (h0 | h1) {
                    int second = h1 && (!h0 || tn1 < tn0);
                    cur = second ? r1 : r0;
                    if (h0 & h1) stack[sp++] = second ? r0 : r1;
                } else cur = stack[--sp];
            }
            if (cur == DONE) break;
            {
                uint32_t x = ~(uint32_t)cur, first = x >> 2, cnt = (x & 3u) + 1u, q; int finished = 0;
                leaves++;
                for (q = 0; q < cnt; q++) {
                    const lh_tri32_t *T = &b->tri32[first + q]; float t_hi; int cls;
                    j->c[1]++;
                    cls = lh_tri_filter(&r, T->v0[0], T->v0[1], T->v0[2], T->e1x, T->e1y, T->e1z,
                                        T->e2x, T->e2y, T->e2z, T->ne1, T->ne2, tb, &t_hi);
                    if (cls != LH_TRI_REJECT) {
                        int sure = (cls == LH_TRI_CERTAIN);
                        if (j->anyhit && sure) { certain = 1; finished = 1; break; }
                        if (sure) tb = fminf(tb, t_hi);
                        if (np == PEND) {
                            for (k = 0; k < PEND; k++) resolve(b, pend[k], o, d, &best);
                            j->c[2] += PEND; np = 0;
                            if (j->anyhit && best.prim != MISS) { finished = 1; break; }
                        }
                        pend[np++] = T->prim;
                    }
                }
                if (finished) { cur = DONE; break; }
                cur = stack[--sp];
            }
        }
        if (!(j->anyhit && (certain || best.prim != MISS))) {
            for (k = 0; k < np; k++) resolve(b, pend[k], o, d, &best);
            j->c[2] += (uint64_t)np;
        }
    }
    if (j->diag) { uint32_t *dg = j->diag + 4 * i; dg[0] = (uint32_t)(j->c[0] - c0); dg[1] = leaves; dg[2] = (uint32_t)(j->c[1] - c1); dg[3] = (uint32_t)(j->c[2] - c2); }
    if (g_ref && best.prim != MISS && best.frag != 0u && !(j->anyhit && certain)) {
        /* the kernel's retrace pass (k_ref_retrace): the reference's own walk decides */
        uint32_t p; double tt, uu, vv;
        const int hit = lh_ref_trace(g_ref->nodes, g_ref->leaf_prims, &b->tri64[0].v[0][0], g_ref->empty, g_ref->bmin, g_ref->bmax,
                                     o[0], o[1], o[2], d[0], d[1], d[2], &p, &tt, &uu, &vv);
        j->c[2]++;
        if (j->anyhit) j->occ[i] = hit ? 1 : 0;
        else { j->prim[i] = p; j->t[i] = tt; j->u[i] = uu; j->v[i] = vv; }
        return;
    }
    if (j->anyhit) j->occ[i] = (certain || best.prim != MISS) ? 1 : 0;
    else { j->prim[i] = best.prim; j->t[i] = best.t; j->u[i] = best.u; j->v[i] = best.v; }
}

static void *run(void *arg) { job_t *j = (job_t *)arg; size_t i; for (i = j->begin; i < j->end; i++) trace_one(j, i); return NULL; }

static uint32_t *g_diag = NULL;      /* per-ray counts of the next lhm_trace call (lhm_set_diag) */
void lhm_set_diag(uint32_t *diag) { g_diag = diag; }

int lhm_trace(const lh_bvh_t *b, size_t n, const double *org, const double *dir, uint32_t *prim,
              double *t, double *u, double *v, uint8_t *occ, int anyhit, uint64_t counters[4], int nthreads)
{
    int i; job_t *jobs; pthread_t *th;
    if (nthreads < 1) nthreads = 1;
    jobs = (job_t *)calloc((size_t)nthreads, sizeof(*jobs)); th = (pthread_t *)calloc((size_t)nthreads, sizeof(*th));
    for (i = 0; i < nthreads; i++) {
        jobs[i].b = b; jobs[i].begin = n * (size_t)i / (size_t)nthreads; jobs[i].end = n * (size_t)(i + 1) / (size_t)nthreads;
        jobs[i].org = org; jobs[i].dir = dir; jobs[i].prim = prim; jobs[i].t = t; jobs[i].u = u; jobs[i].v = v;
        jobs[i].occ = occ; jobs[i].anyhit = anyhit & 1; jobs[i].qnodes = (anyhit >> 1) & 7; jobs[i].diag = g_diag;
    }
    if (nthreads == 1) run(&jobs[0]);
    else { for (i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, run, &jobs[i]); for (i = 0; i < nthreads; i++) pthread_join(th[i], NULL); }
    if (counters) { int k; for (k = 0; k < 4; k++) { counters[k] = 0; for (i = 0; i < nthreads; i++) counters[k] += jobs[i].c[k]; } }
    free(jobs); free(th);
    return 0;
}

/* host-only construction of the product's BVH (no device needed) */
lh_bvh_t *lhm_build(uint32_t npos, const double *pos_xyz, uint32_t nidx, const uint32_t *idx, int nthreads)
{
    lh_bvh_t *b = (lh_bvh_t *)calloc(1, sizeof(*b)); lh_mesh_view_t m;
    m.npositions = npos; m.positions = pos_xyz; m.stride_bytes = 24; m.nindices = nidx; m.indices = idx;
    if (lh_bvh_build(b, &m, 1, nthreads) != 0 || lh_bvh_ensure_q8(b) != 0) { free(b); return NULL; }   /* the model checks every format */
    return b;
}
void lhm_free(lh_bvh_t *b) { if (b) { lh_bvh_release(b); free(b); } }
void lhm_info(const lh_bvh_t *b, uint32_t out[4]) { out[0] = b->ntris; out[1] = b->nnodes; out[2] = b->max_depth; out[3] = b->nleaves; }
void lhm_info4(const lh_bvh_t *b, uint32_t out[2]) { out[0] = b->nq4nodes; out[1] = b->q4_depth; }
void lhm_infoq8(const lh_bvh_t *b, uint32_t out[2]) { out[0] = b->nq8nodes; out[1] = b->q8_depth; }
const void *lhm_q8nodes(const lh_bvh_t *b) { return b->q8nodes; }
const void *lhm_q4nodes(const lh_bvh_t *b) { return b->q4nodes; }
double lhm_build_seconds(const lh_bvh_t *b) { return b->build_seconds; }
const void *lhm_nodes(const lh_bvh_t *b) { return b->nodes; }
const void *lhm_tri32(const lh_bvh_t *b) { return b->tri32; }
void lhm_grid(const lh_bvh_t *b, float out[6]) { int k; for (k = 0; k < 3; k++) { out[k] = b->grid_lo[k]; out[3 + k] = b->grid_step[k]; } }

/* reference-order tree of the same scene */
lh_refbvh_t *lhm_ref_build(const lh_bvh_t *b, int nthreads)
{
    lh_refbvh_t *r = (lh_refbvh_t *)calloc(1, sizeof(*r));
    if (lh_refbvh_build(r, b->tri64, b->ntris, nthreads) != 0) { free(r); return NULL; }
    return r;
}
void lhm_ref_free(lh_refbvh_t *r) { if (r) { lh_refbvh_release(r); free(r); } }
void lhm_ref_use(const lh_refbvh_t *r) { g_ref = r; }
void lhm_ref_info(const lh_refbvh_t *r, uint32_t out[4])
{
    uint32_t i, ninner = 0, nleaf = 0;
    for (i = 0; i < r->nnodes; i++) { if (r->nodes[i].is_leaf) nleaf++; else ninner++; }
    out[0] = ninner; out[1] = nleaf; out[2] = r->max_depth; out[3] = r->ntris;
}
void lhm_ref_leaf_order(const lh_refbvh_t *r, uint32_t *leaf_prims, uint32_t *prim_leaf_first)
{
    uint32_t i;
    memcpy(leaf_prims, r->leaf_prims, sizeof(uint32_t) * r->ntris);
    for (i = 0; i < r->ntris; i++) prim_leaf_first[i] = r->nodes[r->prim_leaf[i]].first;
}
void lhm_ref_bbox(const lh_refbvh_t *r, double out[6]) { int k; for (k = 0; k < 3; k++) { out[k] = r->bmin[k]; out[3 + k] = r->bmax[k]; } }

/* the PRODUCT's one-ray host walk (lucille_amd/csrc/lh_hostwalk.c, linked into this test library as it is): n rays, one after the
 * other, over this model's trees -- so that the not-gpu suite pins its records on the oracle without a device */
int lh_host_walk_closest(const lh_bvh_t *b, const lh_refbvh_t *ref, const double o[3], const double d[3], uint32_t *prim, double *t, double *u, double *v);
int lhm_hostwalk(const lh_bvh_t *b, const lh_refbvh_t *ref, size_t n, const double *org, const double *dir, uint32_t *prim, double *t, double *u, double *v)
{
    size_t i; int hits = 0;
    for (i = 0; i < n; i++) hits += lh_host_walk_closest(b, ref, org + 3 * i, dir + 3 * i, prim + i, t + i, u + i, v + i) > 0;
    return hits;
}
