/*
 * Counting tool (test infrastructure, like the oracle it includes; nothing in the product path uses it).
 *
 * Question: how many node visits of the closest-hit k-d walk lie in subtrees that cannot hold the winner?
 * The reference crosses every cell of the root interval (js/8-kd_tree.js:862-962). A face listed by a leaf may reach
 * outside that leaf's cell, but only so far: D-[a] / D+[a] = the largest amount by which any listed face reaches below /
 * above the cell of a leaf that lists it, per axis, measured on the float32 vertices and the shader's float32 cells.
 * A subtree entered across a split plane of axis a at ray distance t_in then holds no hit nearer than
 * t_in - D[a] / |d[a]|. This file replays every closest-hit search of a frame as a stack walk, with and without that
 * rule, counts the visits, and checks that the winner never changes.
 */
#include "../../oracle/trace_oracle.c"

#define MAX_MODES 8
typedef struct {
    int prune;            /* 0 = visit everything */
    float abs_slack;      /* added to every D[a] */
    float rel_slack;      /* bound = (t_in - D/|d|) - rel_slack * |t_in| */
    int per_node;         /* 1: D = the overhang of the popped subtree (isotropic), not the tree's; 2: D = 0 (NOT exact: the usual early exit, as an upper limit of what pruning can save) */
} probe_mode;

typedef struct {
    uint64_t rays[8], skipped_whole[MAX_MODES][8];
    uint64_t visits[MAX_MODES][8], pruned_pops[MAX_MODES][8], tri_tests[MAX_MODES][8];
    uint64_t mismatches[MAX_MODES], ref_mismatch;
    double min_margin[MAX_MODES];   /* smallest (t of a hit in a skipped subtree would have had - bound): not computed, reserved */
} probe_counts;

static probe_mode g_modes[MAX_MODES];
static int g_n_modes;
static float g_dlo[3], g_dhi[3];
static float *g_sub_ov;             /* per heap index: overhang of the subtree (max over axes and sides) */
static probe_counts g_cnt;
static uint64_t g_hist[64];         /* per listed (leaf, face): overhang in units of 1/4 */

typedef struct { int64_t idx; float bmin[3], bmax[3]; int depth; float tmax; int axis; } Ent;

static float overhang_rec(const oracle_scene *sc, int64_t idx, const float *bmin, const float *bmax, int depth) {
    uint32_t n = kdNode(sc, idx);
    uint32_t ty = n >> 30;
    float ov = 0.0f;
    if (ty == RT_KD_INTERNAL) {
        int a = depth % 3;
        float split = bmin[a] + 0.5f * (bmax[a] - bmin[a]);
        float lmax[3] = { bmax[0], bmax[1], bmax[2] }, umin[3] = { bmin[0], bmin[1], bmin[2] };
        lmax[a] = split; umin[a] = split;
        float o1 = overhang_rec(sc, 2 * idx + 1, bmin, lmax, depth + 1);
        float o2 = overhang_rec(sc, 2 * idx + 2, umin, bmax, depth + 1);
        ov = o1 > o2 ? o1 : o2;
    } else if (ty == RT_KD_LEAF) {
        const int32_t *fi = sc->kd_leaves + 4 * (size_t)(n & 0x3fffffffu);
        for (int i = 0; i < 4; i++) if (fi[i] > -1) {
            f3 tv[3]; fetchTri(sc, fi[i], tv);
            float m = 0.0f;
            for (int a = 0; a < 3; a++) {
                float lo = fminf(fminf(comp(tv[0], a), comp(tv[1], a)), comp(tv[2], a));
                float hi = fmaxf(fmaxf(comp(tv[0], a), comp(tv[1], a)), comp(tv[2], a));
                float dl = bmin[a] - lo, dh = hi - bmax[a];
                if (dl > g_dlo[a]) g_dlo[a] = dl;
                if (dh > g_dhi[a]) g_dhi[a] = dh;
                if (dl > m) m = dl;
                if (dh > m) m = dh;
            }
            int b = (int)(m * 4.0f); if (b > 63) b = 63; if (b < 0) b = 0;
            g_hist[b]++;
            if (m > ov) ov = m;
        }
    }
    if (g_sub_ov && idx < sc->kd_dense_len) g_sub_ov[idx] = ov;
    return ov;
}

static void walk(const oracle_scene *sc, f3 ro, f3 rd, float t_seed, int m, int lv, float *t_out, int *face_out, probe_counts *c) {
    const probe_mode *M = &g_modes[m];
    float t0, t1;
    intersectRayAABB(sc, ro, rd, sc->kd_box_min, sc->kd_box_max, &t0, &t1);
    float best = t_seed; int face = -1;
    *t_out = best; *face_out = face;
    if (!(t0 < t1)) return;
    float w[3];
    for (int a = 0; a < 3; a++) {
        float d = comp(rd, a);
        float D = (d > 0.0f ? g_dlo[a] : g_dhi[a]) + M->abs_slack;
        w[a] = D / fabsf(d);
    }
    Ent st[64]; int sp = 0;
    int64_t idx = 0; int depth = 0;
    float bmin[3] = { sc->kd_box_min[0], sc->kd_box_min[1], sc->kd_box_min[2] };
    float bmax[3] = { sc->kd_box_max[0], sc->kd_box_max[1], sc->kd_box_max[2] };
    float tmin = t0, tmax = t1; const float maxDist = t1;
    HitInfo h; memset(&h, 0, sizeof h);
    for (;;) {
        uint32_t n = kdNode(sc, idx);
        uint32_t ty = n >> 30;
        c->visits[m][lv]++;
        if (ty == RT_KD_INTERNAL) {
            int a = depth % 3;
            float split = bmin[a] + 0.5f * (bmax[a] - bmin[a]);
            float t = (split - comp(ro, a)) / comp(rd, a);
            int neg = comp(rd, a) <= 0.0f;
            int64_t nearI = 2 * idx + 1 + (neg ? 1 : 0), farI = 2 * idx + 1 + (neg ? 0 : 1);
            /* near box / far box */
            float nmin[3] = { bmin[0], bmin[1], bmin[2] }, nmax[3] = { bmax[0], bmax[1], bmax[2] };
            float fmin_[3] = { bmin[0], bmin[1], bmin[2] }, fmax_[3] = { bmax[0], bmax[1], bmax[2] };
            if (neg) { nmin[a] = split; fmax_[a] = split; } else { nmax[a] = split; fmin_[a] = split; }
            int go_far = 0;
            if (t >= tmax) go_far = 0;
            else if (t <= tmin) go_far = 1;
            else {
                Ent *e = &st[sp++];
                e->idx = farI; memcpy(e->bmin, fmin_, sizeof fmin_); memcpy(e->bmax, fmax_, sizeof fmax_);
                e->depth = depth + 1; e->tmax = tmax; e->axis = a;
                tmax = t;
            }
            if (go_far) { idx = farI; memcpy(bmin, fmin_, sizeof fmin_); memcpy(bmax, fmax_, sizeof fmax_); }
            else { idx = nearI; memcpy(bmin, nmin, sizeof nmin); memcpy(bmax, nmax, sizeof nmax); }
            depth++;
            continue;
        }
        if (ty == RT_KD_LEAF) {
            const int32_t *fi = sc->kd_leaves + 4 * (size_t)(n & 0x3fffffffu);
            for (int i = 0; i < 4; i++) if (fi[i] > -1) {
                f3 tv[3]; fetchTri(sc, fi[i], tv);
                c->tri_tests[m][lv]++;
                if (intersectTriangle(sc, ro, rd, &h, tv) && h.t < best) { best = h.t; face = fi[i]; }
            }
        }
        /* next cell */
        int done = 0;
        for (;;) {
            if (!(tmax < maxDist) || sp == 0) { done = 1; break; }
            Ent *e = &st[--sp];
            tmin = tmax; tmax = e->tmax;
            if (M->prune) {
                float ww = w[e->axis];
                if (M->per_node == 1) ww = (g_sub_ov[e->idx] + M->abs_slack) / fabsf(comp(rd, e->axis));
                if (M->per_node == 2) ww = 0.0f;
                float bound = (tmin - ww) - M->rel_slack * fabsf(tmin);
                if (best <= bound) { c->pruned_pops[m][lv]++; continue; }
            }
            idx = e->idx; depth = e->depth; memcpy(bmin, e->bmin, sizeof bmin); memcpy(bmax, e->bmax, sizeof bmax);
            break;
        }
        if (done) break;
    }
    *t_out = best; *face_out = face;
}

static void hook(const oracle_scene *sc, const float *ro3, const float *rd3, float t_seed, float t_best, int face_best, int level) {
    f3 ro = F3(ro3[0], ro3[1], ro3[2]), rd = F3(rd3[0], rd3[1], rd3[2]);
    probe_counts c; memset(&c, 0, sizeof c);
    float t0, t1;
    intersectRayAABB(sc, ro, rd, sc->kd_box_min, sc->kd_box_max, &t0, &t1);
    if (!(t0 < t1)) return;
    c.rays[level]++;
    for (int m = 0; m < g_n_modes; m++) {
        float t; int face;
        walk(sc, ro, rd, t_seed, m, level, &t, &face, &c);
        int same = (face == face_best || (face < 0 && face_best < 0)) && (t == t_best || (face < 0 && face_best < 0));
        /* the reference leaves best.face at -1 when an analytic primitive wins */
        if (!same) { if (m == 0) c.ref_mismatch++; c.mismatches[m]++; }
    }
    uint64_t *dst = (uint64_t *)&g_cnt; const uint64_t *src = (const uint64_t *)&c;
    size_t nwords = offsetof(probe_counts, min_margin) / sizeof(uint64_t);
    for (size_t i = 0; i < nwords; i++) if (src[i]) {
#pragma omp atomic
        dst[i] += src[i];
    }
}

void probe_setup(const oracle_scene *sc, const probe_mode *modes, int n_modes, float *overhang6, uint64_t *hist64) {
    g_n_modes = n_modes > MAX_MODES ? MAX_MODES : n_modes;
    memcpy(g_modes, modes, sizeof(probe_mode) * (size_t)g_n_modes);
    memset(&g_cnt, 0, sizeof g_cnt);
    memset(g_hist, 0, sizeof g_hist);
    for (int a = 0; a < 3; a++) { g_dlo[a] = 0.0f; g_dhi[a] = 0.0f; }
    free(g_sub_ov);
    g_sub_ov = (float *)calloc((size_t)sc->kd_dense_len + 1, sizeof(float));
    overhang_rec(sc, 0, sc->kd_box_min, sc->kd_box_max, 0);
    for (int a = 0; a < 3; a++) { overhang6[a] = g_dlo[a]; overhang6[3 + a] = g_dhi[a]; }
    memcpy(hist64, g_hist, sizeof g_hist);
    oracle_set_closest_hook(hook);
}
void probe_counts_get(probe_counts *out) { *out = g_cnt; }
size_t probe_sizeof_counts(void) { return sizeof(probe_counts); }
