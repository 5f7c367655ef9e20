/*
 * ORACLE - test infrastructure only. Nothing in the product path may call this file.
 *
 * Scalar single-precision CPU restatement of the reference's generated GLSL ES 3.00
 * fragment shader, function by function, in the reference's order of operations:
 *   main / castRayRecursive      /root/reference/js/8-kd_tree.js:1060-1069, :822-1057
 *   closest hit, analytic        js/8-kd_tree.js:1202-1214 (generated per object, scene order)
 *   closest hit, k-d pushdown    js/8-kd_tree.js:850-963   (restart from the pushed node)
 *   castShadow                   js/8-kd_tree.js:575-714
 *   illuminate                   js/8-kd_tree.js:716-820
 *   intersectSphere/Plane/Rect/Cube/Cylinder/Triangle   js/8-kd_tree.js:327-536
 *   intersectRayAABB             js/8-kd_tree.js:545-573
 *   tracer vertex shader         js/8-kd_tree.js:177-190 (bilinear mix of the corner rays)
 *   luminance loop, Lbar, sf     js/8-kd_tree.js:2558-2569
 *   present shader               js/8-kd_tree.js:85-143
 * Checkpoint 6 (js/6-transmission.js:145-728) is the same algorithm with INFINITY = 100000.1,
 * no triangles in the scene and no tone stage; it is a configuration of this file.
 *
 * Choices where GLSL leaves the result to the implementation (each one is also made by the
 * HIP kernels, so that the two agree bit for bit):
 *   - no fused multiply-add anywhere (compile with -ffp-contract=off)
 *   - dot(a,b) = (ax*bx + ay*by) + az*bz ; mat4*vec4(p,1) = ((m0*x + m4*y) + m8*z) + m12 ;
 *     mat4*vec4(d,0) drops the translation term
 *   - normalize(v) = v * (1 / sqrt(dot(v,v))) ; pow() = rt_pow below (exp2(y*log2(x)) in f64)
 *   - locals are zero-initialised (what ANGLE does for WebGL): `specularCoeff` of the checkered
 *     material (js/8-kd_tree.js:797-807) is 0, stale `localHitPoint` of a triangle hit is whatever
 *     the last analytic test wrote
 *   - out-of-range writes rayTree[2n+1|2n+2].valid = false (js/8-kd_tree.js:1025,1035) are dropped
 *   - primaryRayDir is the analytic bilinear mix at the pixel centre, not a rasteriser's
 *
 * PARITY STATUS: the pixel half is "parity unpinned" by any reference execution: the reference's
 * GLSL cannot be compiled or run here (no GL implementation, no GLSL compiler) and the reference
 * has no tests or golden images. It is pinned by hand-derivable known answers (tests/test_oracle_kat.py)
 * and its inputs (uniforms, vertices, k-d arrays) are pinned bit-exactly by running the reference's
 * JavaScript under Node (tests/golden). The tone half (Lbar, sf) IS pinned against reference output.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/rt_hip.h"

typedef struct { float x, y, z; } f3;

typedef struct oracle_scene {
    rt_config cfg;
    int32_t n_prims;
    const rt_prim *prims;
    int32_t n_mats;
    const rt_material *mats;
    int32_t n_lights_loop;          /* trip count of the light loop */
    const rt_light *lights;         /* n_lights_loop entries (unset ones zero) */
    float ambient[3], bg[3];
    float cam[3], ray00[3], ray01[3], ray10[3], ray11[3];
    /* mesh as uploaded to the data textures */
    const float *verts;             /* f32, 3 per vertex */
    const uint32_t *faces;          /* 3 per face */
    const uint32_t *face_mat;       /* 2 per face: type, id */
    int32_t n_verts, n_faces;
    /* k-d tree in the reference's texture layout */
    const uint32_t *kd_dense;       /* metadata by heap index, 0 = INVALID */
    int64_t kd_dense_len;
    const int32_t *kd_leaves;       /* 4 per leaf */
    float kd_box_min[3], kd_box_max[3];
} oracle_scene;

/* Work counters of the REFERENCE algorithm (restart traversal), split by ray-tree level [0..7]. */
typedef struct oracle_counters {
    uint64_t rays;                  /* valid ray-tree nodes, all levels */
    uint64_t shadow_rays;           /* all levels */
    uint64_t rays_level[8];
    uint64_t rays_in_box[8];        /* closest-hit rays whose k-d loop ran at least once */
    uint64_t kd_steps[8], kd_leaf_visits[8], kd_tri_tests[8];       /* closest-hit traversal */
    uint64_t shadow_level[8], shadow_in_box[8];
    uint64_t sh_steps[8], sh_leaf_visits[8], sh_tri_tests[8];       /* any-hit traversal */
    uint64_t prim_tests[8];         /* analytic primitive tests of closest-hit rays */
    uint64_t sh_prim_tests[8];      /* analytic primitive tests of shadow rays */
    uint64_t sh_mesh_hits;          /* shadow rays stopped by a mesh triangle */
    uint64_t sh_hit_after_leaves[8];/* of those: non-empty leaves visited up to the hit: 1, 2, 3-4, 5-8, 9-16, 17-32, 33-64, 65+ */
} oracle_counters;

/* ------------------------------------------------------------------------------------------ */
static inline f3 F3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 add3(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
static inline f3 mulv(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 neg3(f3 a) { return F3(-a.x, -a.y, -a.z); }
static inline float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline f3 cross3(f3 a, f3 b) {
    return F3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline f3 normalize3(f3 v) { float inv = 1.0f / sqrtf(dot3(v, v)); return mul3(v, inv); }
static inline f3 xform_point(const float *m, f3 p) {
    return F3(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12],
              ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
              ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}
static inline f3 xform_dir(const float *m, f3 d) {
    return F3((m[0] * d.x + m[4] * d.y) + m[8] * d.z,
              (m[1] * d.x + m[5] * d.y) + m[9] * d.z,
              (m[2] * d.x + m[6] * d.y) + m[10] * d.z);
}
static inline float comp(f3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
static inline void setcomp(f3 *v, int a, float s) { if (a == 0) v->x = s; else if (a == 1) v->y = s; else v->z = s; }
static inline float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static inline float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
static inline float glsl_mod(float x, float y) { return x - y * floorf(x / y); }

/* pow(x, y) for x >= 0, y > 0: exp2(y * log2(x)) evaluated in double with fixed polynomials,
 * so that every platform that has IEEE double add/mul/div produces the same float. */
static float rt_pow(float xf, float yf) {
    if (!(xf > 0.0f)) return 0.0f;           /* x == 0 (or NaN/negative: GLSL leaves it undefined) */
    if (xf == 1.0f) return 1.0f;
    double x = (double)xf;
    uint64_t bits; memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m; memcpy(&m, &bits, 8);           /* m in [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    /* atanh series: ln m = 2 (s + s^3/3 + ... + s^23/23) */
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0; p = p * s2 + 1.0 / 19.0; p = p * s2 + 1.0 / 17.0; p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0; p = p * s2 + 1.0 / 11.0; p = p * s2 + 1.0 / 9.0;  p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;  p = p * s2 + 1.0 / 3.0;  p = p * s2 + 1.0;
    double ln_m = 2.0 * s * p;
    double lg = (double)e + ln_m * 1.4426950408889634;     /* log2(x) */
    double t = (double)yf * lg;
    if (t < -160.0) return 0.0f;
    if (t > 130.0) return INFINITY;
    double n = floor(t + 0.5), f = (t - n) * 0.6931471805599453;   /* |f| <= 0.347 */
    double q = 1.0 / 87178291200.0;                         /* 1/14! */
    q = q * f + 1.0 / 6227020800.0; q = q * f + 1.0 / 479001600.0; q = q * f + 1.0 / 39916800.0;
    q = q * f + 1.0 / 3628800.0;    q = q * f + 1.0 / 362880.0;    q = q * f + 1.0 / 40320.0;
    q = q * f + 1.0 / 5040.0;       q = q * f + 1.0 / 720.0;       q = q * f + 1.0 / 120.0;
    q = q * f + 1.0 / 24.0;         q = q * f + 1.0 / 6.0;         q = q * f + 0.5;
    q = q * f + 1.0;                q = q * f + 1.0;
    uint64_t sb = (uint64_t)((int64_t)n + 1023) << 52;      /* 2^n, n in [-160,130] */
    double sc; memcpy(&sc, &sb, 8);
    return (float)(q * sc);
}

/* ------------------------------------------------------------------------------------------ */
typedef struct { int hit; f3 hitPoint, localHitPoint, normal; float t; int materialType, materialId; } HitInfo;

static int intersectSphere(const oracle_scene *sc, f3 ro, f3 rd, HitInfo *h, const float *M, const float *IM) {
    f3 lo = xform_point(IM, ro), ld = xform_dir(IM, rd);
    float a = dot3(ld, ld);
    float b = 2.0f * dot3(lo, ld);
    float c = dot3(lo, lo) - 1.0f;
    float disc = b * b - 4.0f * a * c;
    if (disc > 0.0f) {
        float e = sqrtf(disc), denom = 2.0f * a;
        float t0 = (-b - e) / denom;
        if (t0 > sc->cfg.sphere_epsilon) {
            h->t = t0;
            h->localHitPoint = add3(lo, mul3(ld, h->t));
            h->normal = h->localHitPoint;                 /* object space, js/8-kd_tree.js:345 */
            return 1;
        }
        t0 = (-b + e) / denom;
        if (t0 > sc->cfg.sphere_epsilon) {
            h->t = t0;
            h->localHitPoint = add3(lo, mul3(ld, h->t));
            h->normal = xform_dir(M, h->localHitPoint);
            return 1;
        }
    }
    return 0;
}

static int intersectPlane(const oracle_scene *sc, f3 ro, f3 rd, HitInfo *h, const float *M, const float *IM) {
    f3 lo = xform_point(IM, ro), ld = xform_dir(IM, rd);
    f3 n = F3(0.0f, 1.0f, 0.0f);
    float t0 = dot3(neg3(lo), n) / dot3(ld, n);
    if (t0 > sc->cfg.plane_rect_epsilon) {
        h->t = t0;
        h->localHitPoint = add3(lo, mul3(ld, h->t));
        h->normal = xform_dir(M, n);
        return 1;
    }
    return 0;
}

static int intersectRect(const oracle_scene *sc, f3 ro, f3 rd, HitInfo *h, const float *M, const float *IM) {
    f3 lo = xform_point(IM, ro), ld = xform_dir(IM, rd);
    f3 A = F3(1.0f, 0.0f, 0.0f), B = F3(0.0f, 0.0f, -1.0f);
    f3 n = cross3(A, B);
    f3 p0 = F3(-0.5f, 0.0f, 0.5f);
    float t0 = dot3(sub3(p0, lo), n) / dot3(ld, n);
    if (t0 > sc->cfg.plane_rect_epsilon) {
        f3 p = add3(lo, mul3(ld, t0));
        f3 d = sub3(p, p0);
        float da = dot3(d, A), db = dot3(d, B);
        if (da > 0.0f && da < 1.0f && db > 0.0f && db < 1.0f) {
            h->t = t0;
            h->localHitPoint = add3(lo, mul3(ld, h->t));
            h->normal = xform_dir(M, n);
            return 1;
        }
    }
    return 0;
}

static int intersectCube(const oracle_scene *sc, f3 ro, f3 rd, HitInfo *h, const float *M, const float *IM) {
    (void)sc;
    f3 lo = xform_point(IM, ro), ld = xform_dir(IM, rd);
    f3 od = F3(lo.x / ld.x, lo.y / ld.y, lo.z / ld.z);
    f3 t0 = F3(-0.5f / fabsf(ld.x) - od.x, -0.5f / fabsf(ld.y) - od.y, -0.5f / fabsf(ld.z) - od.z);
    f3 t1 = F3(0.5f / fabsf(ld.x) - od.x, 0.5f / fabsf(ld.y) - od.y, 0.5f / fabsf(ld.z) - od.z);
    float tMin = fmaxf(fmaxf(t0.x, t0.y), t0.z);
    float tMax = fminf(fminf(t1.x, t1.y), t1.z);
    if (tMin > tMax || tMax < 0.0f) return 0;
    h->t = tMin;
    h->localHitPoint = add3(lo, mul3(ld, h->t));
    /* -sign(d) * step(t0.yzx, t0.xyz) * step(t0.zxy, t0.xyz) */
    f3 n = F3(-signf(ld.x) * stepf(t0.y, t0.x) * stepf(t0.z, t0.x),
              -signf(ld.y) * stepf(t0.z, t0.y) * stepf(t0.x, t0.y),
              -signf(ld.z) * stepf(t0.x, t0.z) * stepf(t0.y, t0.z));
    h->normal = xform_dir(M, n);
    return 1;
}

static int intersectCylinder(const oracle_scene *sc, f3 ro, f3 rd, HitInfo *h, const float *M, const float *IM) {
    (void)sc;
    f3 lo = xform_point(IM, ro), ld = xform_dir(IM, rd);
    float ox = lo.x, oy = lo.y, oz = lo.z, dx = ld.x, dy = ld.y, dz = ld.z;
    float a = dx * dx + dz * dz;
    float b = 2.0f * (ox * dx + oz * dz);
    float c = ox * ox + oz * oz - 1.0f;
    float disc = b * b - 4.0f * a * c;
    if (disc > 0.0f) {
        float e = sqrtf(disc), denom = 2.0f * a;
        float t0 = (-b - e) / denom, t1 = (-b + e) / denom;
        if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
        float y0 = oy + t0 * dy, y1 = oy + t1 * dy;
        if (y0 < -0.5f) {
            if (y1 >= -0.5f) {
                float th = t0 + (t1 - t0) * (y0 + 0.5f) / (y0 - y1);
                if (th > 0.0f) {
                    h->t = th;
                    h->localHitPoint = add3(lo, mul3(ld, h->t));
                    h->normal = xform_dir(M, F3(0.0f, -1.0f, 0.0f));
                    return 1;
                }
            }
        } else if (y0 >= -0.5f && y0 <= 0.5f) {
            if (t0 > 0.0f) {
                h->t = t0;
                h->localHitPoint = add3(lo, mul3(ld, h->t));
                h->normal = xform_dir(M, F3(h->localHitPoint.x, 0.0f, h->localHitPoint.z));
                return 1;
            }
        } else if (y0 > 0.5f) {
            if (y1 <= 0.5f) {
                float th = t0 + (t1 - t0) * (y0 - 0.5f) / (y0 - y1);
                if (th > 0.0f) {
                    h->t = th;
                    h->localHitPoint = add3(lo, mul3(ld, h->t));
                    h->normal = xform_dir(M, F3(0.0f, 1.0f, 0.0f));
                    return 1;
                }
            }
        }
    }
    return 0;
}

static int intersectPrim(const oracle_scene *sc, const rt_prim *p, f3 ro, f3 rd, HitInfo *h) {
    switch (p->kind) {
    case RT_PRIM_PLANE: return intersectPlane(sc, ro, rd, h, p->model, p->inv_model);
    case RT_PRIM_RECT: return intersectRect(sc, ro, rd, h, p->model, p->inv_model);
    case RT_PRIM_SPHERE: return intersectSphere(sc, ro, rd, h, p->model, p->inv_model);
    case RT_PRIM_CUBE: return intersectCube(sc, ro, rd, h, p->model, p->inv_model);
    case RT_PRIM_CYLINDER: return intersectCylinder(sc, ro, rd, h, p->model, p->inv_model);
    }
    return 0;
}

static int intersectTriangle(const oracle_scene *sc, f3 ro, f3 rd, HitInfo *h, const f3 *tv) {
    f3 e1 = sub3(tv[1], tv[0]), e2 = sub3(tv[2], tv[0]);
    f3 T = sub3(ro, tv[0]);
    f3 P = cross3(rd, e2), Q = cross3(T, e1);
    float PdotE1 = dot3(P, e1);
    if (fabsf(PdotE1) < sc->cfg.triangle_epsilon) return 0;
    float inv = 1.0f / PdotE1;
    float lx = inv * dot3(Q, e2), ly = inv * dot3(P, T), lz = inv * dot3(Q, rd);
    h->t = lx;
    if (h->t < 0.0f) return 0;
    float u = ly, v = lz;
    if (u < 0.0f || v < 0.0f || u + v > 1.0f) return 0;
    h->normal = normalize3(cross3(e1, e2));
    return 1;
}

static void fetchTri(const oracle_scene *sc, int32_t face, f3 *tv) {
    for (int j = 0; j < 3; j++) {
        const float *p = sc->verts + 3 * (size_t)sc->faces[3 * (size_t)face + j];
        tv[j] = F3(p[0], p[1], p[2]);
    }
}

static void intersectRayAABB(const oracle_scene *sc, f3 ro, f3 rd, const float *bmin, const float *bmax,
                             float *tMin, float *tMax) {
    *tMin = -sc->cfg.infinity;
    *tMax = sc->cfg.infinity;
    for (int i = 0; i < 3; i++) {
        float lo = (bmin[i] - comp(ro, i)) / comp(rd, i);
        float hi = (bmax[i] - comp(ro, i)) / comp(rd, i);
        if (lo > hi) { float tmp = lo; lo = hi; hi = tmp; }
        if (lo > *tMin) *tMin = lo;
        if (hi < *tMax) *tMax = hi;
    }
}

static inline uint32_t kdNode(const oracle_scene *sc, int64_t idx) {
    return (idx >= 0 && idx < sc->kd_dense_len) ? sc->kd_dense[idx] : 0u;
}

/* k-d pushdown traversal (Hapala & Havran restart variant), literal: js/8-kd_tree.js:851-963 for
 * closest == 1, js/8-kd_tree.js:588-692 for closest == 0 (returns 1 at the first triangle hit). */
typedef struct { float tMin; f3 normal, hitPoint, localHitPoint; int materialType, materialId, face; } Closest;

static int kdTraverse(const oracle_scene *sc, f3 ro, f3 rd, HitInfo *h, int closest, Closest *best,
                      oracle_counters *cn, int lv) {
    float kdTMin, kdTMax;
    intersectRayAABB(sc, ro, rd, sc->kd_box_min, sc->kd_box_max, &kdTMin, &kdTMax);
    float maxDist = kdTMax;
    kdTMax = kdTMin;
    uint32_t pushedNode = kdNode(sc, 0);
    int64_t pushedNodeIndex = 0;
    f3 pushedBoxMin = F3(sc->kd_box_min[0], sc->kd_box_min[1], sc->kd_box_min[2]);
    f3 pushedBoxMax = F3(sc->kd_box_max[0], sc->kd_box_max[1], sc->kd_box_max[2]);
    int pushedDepth = 0;
    int entered = 0;
    uint64_t leaves_seen = 0;
    if ((pushedNode >> 30) != RT_KD_INVALID) {
        int pushdown = 1;
        while (kdTMax < maxDist) {
            entered = 1;
            uint32_t currNode = pushedNode;
            int64_t currNodeIndex = pushedNodeIndex;
            f3 currBoxMin = pushedBoxMin, currBoxMax = pushedBoxMax;
            int currDepth = pushedDepth;
            kdTMin = kdTMax;
            kdTMax = maxDist;
            while ((currNode >> 30) != RT_KD_EMPTY_LEAF && (currNode >> 30) != RT_KD_LEAF) {
                if (closest) cn->kd_steps[lv]++; else cn->sh_steps[lv]++;
                int axis = currDepth % 3;
                float split = comp(currBoxMin, axis) + 0.5f * (comp(currBoxMax, axis) - comp(currBoxMin, axis));
                float t = (split - comp(ro, axis)) / comp(rd, axis);
                int64_t nearIndex = 2 * currNodeIndex + 1, farIndex = 2 * currNodeIndex + 2;
                uint32_t nearN = kdNode(sc, nearIndex), farN = kdNode(sc, farIndex);
                f3 nearMin = currBoxMin, nearMax = currBoxMax, farMin = currBoxMin, farMax = currBoxMax;
                setcomp(&nearMax, axis, split);
                setcomp(&farMin, axis, split);
                if (comp(rd, axis) <= 0.0f) {
                    uint32_t tn = nearN; nearN = farN; farN = tn;
                    int64_t ti = nearIndex; nearIndex = farIndex; farIndex = ti;
                    f3 tb = nearMin; nearMin = farMin; farMin = tb;
                    tb = nearMax; nearMax = farMax; farMax = tb;
                }
                if (t >= kdTMax) {
                    currNode = nearN; currNodeIndex = nearIndex; currBoxMin = nearMin; currBoxMax = nearMax;
                } else if (t <= kdTMin) {
                    currNode = farN; currNodeIndex = farIndex; currBoxMin = farMin; currBoxMax = farMax;
                } else {
                    currNode = nearN; currNodeIndex = nearIndex; currBoxMin = nearMin; currBoxMax = nearMax;
                    kdTMax = t;
                    pushdown = 0;
                }
                ++currDepth;
                if (pushdown) {
                    pushedNode = currNode; pushedNodeIndex = currNodeIndex;
                    pushedBoxMin = currBoxMin; pushedBoxMax = currBoxMax; pushedDepth = currDepth;
                }
                if ((currNode >> 30) == RT_KD_INVALID) {
                    /* cannot happen on a tree written by fillDataArray: an INTERNAL node always has
                     * both children; a GLSL loop would spin on texel 0s. Stop instead of hanging. */
                    return 0;
                }
            }
            if ((currNode >> 30) != RT_KD_EMPTY_LEAF) {
                if (closest) cn->kd_leaf_visits[lv]++; else cn->sh_leaf_visits[lv]++;
                leaves_seen++;
                const int32_t *fi = sc->kd_leaves + 4 * (size_t)(currNode & 0x3fffffffu);
                for (int i = 0; i < 4; i++) {
                    if (fi[i] > -1) {
                        f3 tv[3];
                        fetchTri(sc, fi[i], tv);
                        if (closest) cn->kd_tri_tests[lv]++; else cn->sh_tri_tests[lv]++;
                        if (!closest) {
                            if (intersectTriangle(sc, ro, rd, h, tv)) {
                                int b = leaves_seen <= 1 ? 0 : leaves_seen <= 2 ? 1 : leaves_seen <= 4 ? 2 : leaves_seen <= 8 ? 3 :
                                        leaves_seen <= 16 ? 4 : leaves_seen <= 32 ? 5 : leaves_seen <= 64 ? 6 : 7;
                                cn->sh_mesh_hits++;
                                cn->sh_hit_after_leaves[b]++;
                                if (entered) cn->shadow_in_box[lv]++;
                                return 1;
                            }
                        } else if (intersectTriangle(sc, ro, rd, h, tv) && h->t < best->tMin) {
                            h->hit = 1;
                            best->hitPoint = add3(ro, mul3(rd, h->t));
                            best->tMin = h->t;
                            best->normal = h->normal;
                            best->localHitPoint = h->localHitPoint;
                            best->materialType = (int)sc->face_mat[2 * (size_t)fi[i]];
                            best->materialId = (int)sc->face_mat[2 * (size_t)fi[i] + 1];
                            best->face = fi[i];
                        }
                    }
                }
            }
        }
    }
    if (entered) { if (closest) cn->rays_in_box[lv]++; else cn->shadow_in_box[lv]++; }
    return 0;
}

static int castShadow(const oracle_scene *sc, f3 ro, f3 rd, oracle_counters *cn, int lv) {
    rd = normalize3(rd);
    HitInfo h; memset(&h, 0, sizeof h);
    cn->shadow_rays++;
    cn->shadow_level[lv]++;
    for (int i = 0; i < sc->n_prims; i++) {
        cn->sh_prim_tests[lv]++;
        if (intersectPrim(sc, &sc->prims[i], ro, rd, &h)) return 1;
    }
    if (sc->cfg.render_triangles) {
        if (sc->cfg.use_kd_tree) {
            if (kdTraverse(sc, ro, rd, &h, 0, NULL, cn, lv)) return 1;
        } else {
            for (int32_t i = 0; i < sc->n_faces; i++) {
                f3 tv[3]; fetchTri(sc, i, tv);
                cn->sh_tri_tests[lv]++;
                if (intersectTriangle(sc, ro, rd, &h, tv)) return 1;
            }
        }
    }
    return 0;
}

static const rt_material *findMat(const oracle_scene *sc, int type, int id) {
    for (int i = 0; i < sc->n_mats; i++) if (sc->mats[i].type == type && sc->mats[i].id == id) return &sc->mats[i];
    return NULL;
}

static f3 illuminate(const oracle_scene *sc, const HitInfo *h, f3 rd, float *reflMix, float *refrMix,
                     float *refrIdx, oracle_counters *cn, int lv) {
    float ka = 0, kd = 0, ks = 0, ke = 0, kReflect = 0, kRefract = 0, ior = 0;
    f3 Co = F3(0, 0, 0), Cs = F3(0, 0, 0);
    const rt_material *m = findMat(sc, h->materialType, h->materialId);
    if (m) {
        ka = m->ka; kd = m->kd; ks = m->ks; ke = m->ke; kReflect = m->kReflect; kRefract = m->kRefract; ior = m->ior;
        Cs = F3(m->Cs[0], m->Cs[1], m->Cs[2]);
        if (m->type == RT_MAT_PHONG_CHECKERED) {
            float x = h->localHitPoint.x, z = h->localHitPoint.z;
            float u = x + 0.5f, v = z + 0.5f;
            float sizeU = 1.0f / 12.0f;
            float sizeV = 450.0f / (1600.0f * 12.0f);
            float row = (float)(int)(u / sizeU);
            float col = (float)(int)(v / sizeV);
            f3 Co1 = F3(m->Co[0], m->Co[1], m->Co[2]), Co2 = F3(m->Co2[0], m->Co2[1], m->Co2[2]);
            if ((int)glsl_mod(row, 2.0f) == 0) Co = ((int)glsl_mod(col, 2.0f) == 0) ? Co1 : Co2;
            else Co = ((int)glsl_mod(col, 2.0f) == 0) ? Co2 : Co1;
        } else {
            Co = F3(m->Co[0], m->Co[1], m->Co[2]);
        }
    }
    f3 N = normalize3(h->normal);
    f3 V = normalize3(neg3(rd));
    f3 amb = F3(sc->ambient[0], sc->ambient[1], sc->ambient[2]);
    f3 ambient = mulv(mul3(amb, ka), Co);            /* ka * ambientLight * Co, left to right */
    f3 diffuse = F3(0, 0, 0), specular = F3(0, 0, 0);
    for (int i = 0; i < sc->n_lights_loop; i++) {
        f3 Lp = F3(sc->lights[i].position[0], sc->lights[i].position[1], sc->lights[i].position[2]);
        f3 Lc = F3(sc->lights[i].color[0], sc->lights[i].color[1], sc->lights[i].color[2]);
        f3 S = normalize3(sub3(Lp, h->hitPoint));
        int inShadow = castShadow(sc, add3(h->hitPoint, mul3(h->normal, sc->cfg.epsilon)), S, cn, lv);
        if (!inShadow) {
            float SdotN = fmaxf(0.0f, dot3(S, N));
            float dc = clamp01(kd * SdotN);
            diffuse = add3(diffuse, mulv(mul3(Lc, dc), Co));
            if (SdotN > 0.0f) {
                float sp = 0.0f;
                if (h->materialType == RT_MAT_PHONG) {
                    /* reflect(-S, N) = I - 2 dot(N, I) N */
                    f3 I = neg3(S);
                    f3 R = normalize3(sub3(I, mul3(N, 2.0f * dot3(N, I))));
                    float RdotV = fmaxf(0.0f, dot3(R, V));
                    sp = ks * rt_pow(RdotV, ke);
                } else if (h->materialType == RT_MAT_PHONG_BLINN) {
                    f3 H = normalize3(add3(S, V));
                    float NdotH = fmaxf(0.0f, dot3(N, H));
                    sp = ks * rt_pow(NdotH, ke);
                }
                sp = clamp01(sp);
                specular = add3(specular, mulv(mul3(Lc, sp), Cs));
            }
        }
    }
    *reflMix = kReflect; *refrMix = kRefract; *refrIdx = ior;
    /* accumulatedColor(0) += ambient + specular + diffuse */
    return add3(F3(0, 0, 0), add3(add3(ambient, specular), diffuse));
}

/* Observer of every closest-hit k-d search, for counting tools that include this file (tools/prune_probe): called after
 * the traversal with the ray, the distance the search started from (the analytic winner's or INFINITY) and the result. */
typedef void (*oracle_closest_hook_fn)(const oracle_scene *sc, const float *ro3, const float *rd3, float t_seed, float t_best,
                                       int face_best, int level);
static oracle_closest_hook_fn g_closest_hook;
void oracle_set_closest_hook(oracle_closest_hook_fn f) { g_closest_hook = f; }

/* The closest-hit search of one ray-tree node (js/8-kd_tree.js:846-990): analytic primitives in scene order, then the
 * mesh. Fills h (hit flag) and best; *prim_hit = index of the winning analytic primitive or -1. */
static void findClosest(const oracle_scene *sc, f3 ro, f3 rd, HitInfo *hp, Closest *bp, int *prim_hit_p, oracle_counters *cn, int level) {
    HitInfo h = *hp;
    Closest best = *bp;
    int prim_hit = -1;
        for (int i = 0; i < sc->n_prims; i++) {
            cn->prim_tests[level]++;
            if (intersectPrim(sc, &sc->prims[i], ro, rd, &h) && h.t < best.tMin) {
                h.hit = 1;
                best.hitPoint = add3(ro, mul3(rd, h.t));
                best.tMin = h.t;
                best.normal = h.normal;
                best.localHitPoint = h.localHitPoint;
                best.materialType = sc->prims[i].mat_type;
                best.materialId = sc->prims[i].mat_id;
                best.face = -1;
                prim_hit = i;
            }
        }
        if (sc->cfg.render_triangles) {
            if (sc->cfg.use_kd_tree) {
                const float t_seed = best.tMin;
                kdTraverse(sc, ro, rd, &h, 1, &best, cn, level);
                if (g_closest_hook) {
                    const float o3[3] = { ro.x, ro.y, ro.z }, d3[3] = { rd.x, rd.y, rd.z };
                    g_closest_hook(sc, o3, d3, t_seed, best.tMin, best.face, level);
                }
            } else {
                for (int32_t i = 0; i < sc->n_faces; i++) {
                    f3 tv[3]; fetchTri(sc, i, tv);
                    cn->kd_tri_tests[level]++;
                    if (intersectTriangle(sc, ro, rd, &h, tv) && h.t < best.tMin) {
                        h.hit = 1;
                        best.hitPoint = add3(ro, mul3(rd, h.t));
                        best.tMin = h.t; best.normal = h.normal; best.localHitPoint = h.localHitPoint;
                        best.materialType = (int)sc->face_mat[2 * (size_t)i];
                        best.materialId = (int)sc->face_mat[2 * (size_t)i + 1];
                        best.face = i;
                    }
                }
            }
        }
    *hp = h; *bp = best; *prim_hit_p = prim_hit;
}

typedef struct { int valid; f3 ro, rd, color; float kReflect, kRefract; } TreeNode;

#define MAX_TREE 255

/* One fragment. Returns the colour; *hit_id = id of what the primary ray hit. */
static f3 shadePixel(const oracle_scene *sc, f3 primaryDir, int32_t *hit_id, oracle_counters *cn) {
    TreeNode tree[MAX_TREE];
    int size = (1 << sc->cfg.max_depth) - 1;
    if (size > MAX_TREE) size = MAX_TREE;
    memset(tree, 0, sizeof(TreeNode) * (size_t)size);
    tree[0].valid = 1;
    tree[0].ro = F3(sc->cam[0], sc->cam[1], sc->cam[2]);
    tree[0].rd = normalize3(primaryDir);
    *hit_id = -1;
    for (int node = 0; node < size; ++node) {
        if (!tree[node].valid) continue;
        f3 ro = tree[node].ro, rd = tree[node].rd;
        int level = 0; for (int n = node + 1; n > 1; n >>= 1) level++;
        if (level > 7) level = 7;
        cn->rays++; cn->rays_level[level]++;
        Closest best; memset(&best, 0, sizeof best);
        best.tMin = sc->cfg.infinity; best.face = -1;
        int prim_hit = -1;
        HitInfo h; memset(&h, 0, sizeof h);
        findClosest(sc, ro, rd, &h, &best, &prim_hit, cn, level);
        if (h.hit) {
            h.t = best.tMin; h.normal = best.normal; h.localHitPoint = best.localHitPoint;
            h.hitPoint = best.hitPoint; h.materialType = best.materialType; h.materialId = best.materialId;
            if (node == 0) *hit_id = (best.face >= 0) ? sc->n_prims + best.face : prim_hit;
        } else {
            tree[node].color = F3(sc->bg[0], sc->bg[1], sc->bg[2]);
            continue;
        }
        float kReflect = 0.0f, kRefract = 0.0f, ior = 0.0f;
        tree[node].color = illuminate(sc, &h, rd, &kReflect, &kRefract, &ior, cn, level);

        int outside = dot3(rd, h.normal) < 0.0f;
        f3 bias = mul3(h.normal, sc->cfg.epsilon);
        if (kRefract > sc->cfg.epsilon && ior > sc->cfg.epsilon && (node * 2 + 2) < size) {
            TreeNode *c = &tree[node * 2 + 2];
            tree[node].kRefract = kRefract;
            c->valid = 1;
            c->ro = outside ? sub3(h.hitPoint, bias) : add3(h.hitPoint, bias);
            f3 N = outside ? h.normal : neg3(h.normal);
            float eta = outside ? (1.0f / ior) : ior;
            float cosi = dot3(neg3(N), rd);
            float k = 1.0f + eta * eta * (cosi * cosi - 1.0f);
            c->rd = add3(mul3(rd, eta), mul3(N, eta * cosi - sqrtf(k)));    /* not normalised: js/8-kd_tree.js:1022 */
        } else if ((node * 2 + 2) < size) {
            tree[node * 2 + 2].valid = 0;
        }
        if (kReflect > sc->cfg.epsilon && (node * 2 + 1) < size) {
            TreeNode *c = &tree[node * 2 + 1];
            tree[node].kReflect = kReflect;
            c->valid = 1;
            c->ro = outside ? add3(h.hitPoint, bias) : sub3(h.hitPoint, bias);
            /* reflect(rayDir, normal) with the un-normalised normal */
            c->rd = sub3(rd, mul3(h.normal, 2.0f * dot3(h.normal, rd)));
        } else if ((node * 2 + 1) < size) {
            tree[node * 2 + 1].valid = 0;
        }
    }
    for (int node = size - 1; node > 0; --node) {
        if (tree[node].valid) {
            int parent = (node - 1) / 2;
            float mix = (node & 1) ? tree[parent].kReflect : tree[parent].kRefract;
            tree[parent].color = add3(tree[parent].color, mul3(tree[node].color, mix));
        }
    }
    return tree[0].color;
}

/* Test access to the pieces of one ray-tree node, for known-answer tests: closest hit of an arbitrary ray, the local
 * colour illuminate() gives it and the two child rays. out[0] = 1 on a hit, out[1] = analytic primitive index or
 * n_prims + face, out[2] = t, out[3..5] = hit point, out[6..8] = normal as the shader holds it (not normalised),
 * out[9..11] = object-space hit point, out[12..14] = local colour, out[15..17] = kReflect, kRefract, ior,
 * out[18..23] = origin and direction of the refracted child, out[24..29] = of the reflected child (computed whatever
 * the coefficients are), out[30] = number of lights whose shadow ray was blocked. */
void oracle_probe(const oracle_scene *sc, const float *ro3, const float *rd3, float *out) {
    oracle_counters cn; memset(&cn, 0, sizeof cn);
    f3 ro = F3(ro3[0], ro3[1], ro3[2]), rd = F3(rd3[0], rd3[1], rd3[2]);
    Closest best; memset(&best, 0, sizeof best);
    best.tMin = sc->cfg.infinity; best.face = -1;
    int prim_hit = -1;
    HitInfo h; memset(&h, 0, sizeof h);
    for (int i = 0; i < 31; i++) out[i] = 0.0f;
    findClosest(sc, ro, rd, &h, &best, &prim_hit, &cn, 0);
    if (!h.hit) return;
    h.t = best.tMin; h.normal = best.normal; h.localHitPoint = best.localHitPoint;
    h.hitPoint = best.hitPoint; h.materialType = best.materialType; h.materialId = best.materialId;
    out[0] = 1.0f; out[1] = (float)((best.face >= 0) ? sc->n_prims + best.face : prim_hit); out[2] = h.t;
    out[3] = h.hitPoint.x; out[4] = h.hitPoint.y; out[5] = h.hitPoint.z;
    out[6] = h.normal.x; out[7] = h.normal.y; out[8] = h.normal.z;
    out[9] = h.localHitPoint.x; out[10] = h.localHitPoint.y; out[11] = h.localHitPoint.z;
    float kReflect = 0.0f, kRefract = 0.0f, ior = 0.0f;
    f3 c = illuminate(sc, &h, rd, &kReflect, &kRefract, &ior, &cn, 0);
    out[12] = c.x; out[13] = c.y; out[14] = c.z; out[15] = kReflect; out[16] = kRefract; out[17] = ior;
    /* the child rays exactly as shadePixel forms them */
    int outside = dot3(rd, h.normal) < 0.0f;
    f3 bias = mul3(h.normal, sc->cfg.epsilon);
    f3 o1 = outside ? sub3(h.hitPoint, bias) : add3(h.hitPoint, bias);
    f3 N = outside ? h.normal : neg3(h.normal);
    float eta = outside ? (1.0f / ior) : ior;
    float cosi = dot3(neg3(N), rd);
    float k = 1.0f + eta * eta * (cosi * cosi - 1.0f);
    f3 d1 = add3(mul3(rd, eta), mul3(N, eta * cosi - sqrtf(k)));
    f3 o2 = outside ? add3(h.hitPoint, bias) : sub3(h.hitPoint, bias);
    f3 d2 = sub3(rd, mul3(h.normal, 2.0f * dot3(h.normal, rd)));
    out[18] = o1.x; out[19] = o1.y; out[20] = o1.z; out[21] = d1.x; out[22] = d1.y; out[23] = d1.z;
    out[24] = o2.x; out[25] = o2.y; out[26] = o2.z; out[27] = d2.x; out[28] = d2.y; out[29] = d2.z;
    /* castShadow counts a mesh hit itself; analytic blockers are told apart by re-running the shadow rays */
    int blocked = 0;
    for (int i = 0; i < sc->n_lights_loop; i++) {
        f3 Lp = F3(sc->lights[i].position[0], sc->lights[i].position[1], sc->lights[i].position[2]);
        f3 S = normalize3(sub3(Lp, h.hitPoint));
        blocked += castShadow(sc, add3(h.hitPoint, mul3(h.normal, sc->cfg.epsilon)), S, &cn, 0);
    }
    out[30] = (float)blocked;
}

static inline f3 mix3(f3 a, f3 b, float t) {   /* GLSL mix: a*(1-t) + b*t */
    float s = 1.0f - t;
    return F3(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t);
}

static inline uint8_t quant8(float c) {          /* GL fixed-point conversion of a colour channel */
    if (!(c == c)) return 0;
    float f = clamp01(c);
    return (uint8_t)(int)(f * 255.0f + 0.5f);
}

/* Renders the pixel rectangle [x0,x1) x [y0,y1) of the (width*S) x (height*S) target; y = 0 is the
 * bottom row. rgb / rgba8 / hit_ids are FULL-frame buffers (any may be NULL). step > 1 renders only
 * pixels with x % step == 0 and y % step == 0 (sampling for counters). */
void oracle_trace(const oracle_scene *sc, int x0, int y0, int x1, int y1, int step, float *rgb, uint8_t *rgba8,
                  int32_t *hit_ids, oracle_counters *out, int n_threads) {
    int W = sc->cfg.width * sc->cfg.supersampling, H = sc->cfg.height * sc->cfg.supersampling;
    f3 r00 = F3(sc->ray00[0], sc->ray00[1], sc->ray00[2]), r01 = F3(sc->ray01[0], sc->ray01[1], sc->ray01[2]);
    f3 r10 = F3(sc->ray10[0], sc->ray10[1], sc->ray10[2]), r11 = F3(sc->ray11[0], sc->ray11[1], sc->ray11[2]);
    oracle_counters total; memset(&total, 0, sizeof total);
    if (step < 1) step = 1;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    {
        oracle_counters cn; memset(&cn, 0, sizeof cn);
#pragma omp for schedule(dynamic, 1)
        for (int y = y0; y < y1; y++) {
            if (y % step) continue;
            for (int x = x0; x < x1; x++) {
                if (x % step) continue;
                float fx = ((float)x + 0.5f) / (float)W, fy = ((float)y + 0.5f) / (float)H;
                f3 dir = mix3(mix3(r00, r01, fy), mix3(r10, r11, fy), fx);
                int32_t id;
                f3 c = shadePixel(sc, dir, &id, &cn);
                size_t p = (size_t)y * (size_t)W + (size_t)x;
                if (rgb) { rgb[3 * p] = c.x; rgb[3 * p + 1] = c.y; rgb[3 * p + 2] = c.z; }
                if (rgba8) { rgba8[4 * p] = quant8(c.x); rgba8[4 * p + 1] = quant8(c.y); rgba8[4 * p + 2] = quant8(c.z); rgba8[4 * p + 3] = 255; }
                if (hit_ids) hit_ids[p] = id;
            }
        }
#pragma omp critical
        {
            uint64_t *a = (uint64_t *)&total; const uint64_t *b = (const uint64_t *)&cn;
            for (size_t i = 0; i < sizeof(oracle_counters) / sizeof(uint64_t); i++) a[i] += b[i];
        }
    }
    if (out) *out = total;
}

/* js/8-kd_tree.js:2559-2569, sequential double accumulation in pixel order */
void oracle_tone(const oracle_scene *sc, const uint8_t *rgba8, int64_t n_pixels, double *Lbar, double *sf) {
    double L_MAX = (double)sc->cfg.L_max, L_DMAX = (double)sc->cfg.L_dmax, EPS = 0.0001;
    double acc = 0.0;
    for (int64_t i = 0; i < n_pixels; i++) {
        double r = rgba8[i * 4 + 0] * L_MAX / 256.0;
        double g = rgba8[i * 4 + 1] * L_MAX / 256.0;
        double b = rgba8[i * 4 + 2] * L_MAX / 256.0;
        double L = 0.27 * r + 0.67 * g + 0.07 * b;
        acc += log(EPS + L) / (double)n_pixels;
    }
    *Lbar = exp(acc);
    *sf = pow((1.219 + pow(L_DMAX * 0.5, 0.4)) / (1.219 + pow(*Lbar, 0.4)), 2.5);
}

/* Present shader, js/8-kd_tree.js:85-143. The sample offsets divide by the WIDTH/HEIGHT the script
 * had when the shader string was built (800 x 600, js/8-kd_tree.js:55,110-111), not by the canvas
 * size; texture() is NEAREST with CLAMP_TO_EDGE (js/8-kd_tree.js:2322-2325). Lbar and sf arrive as
 * float uniforms (js/8-kd_tree.js:2579-2581). */
void oracle_present(const oracle_scene *sc, const uint8_t *rgba8, float Lbar, float sf, uint8_t *canvas) {
    int S = sc->cfg.supersampling, W = sc->cfg.width, H = sc->cfg.height, TW = W * S, TH = H * S;
    float offW = (float)(800 * S), offH = (float)(600 * S);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        /* texCoord is the varying vertex.xy*0.5+0.5 interpolated to the pixel centre */
        float tx = ((float)x + 0.5f) / (float)W, ty = ((float)y + 0.5f) / (float)H;
        float acc[3] = { 0, 0, 0 };
        int cnt = 0;
        float lo, hi;
        if (glsl_mod((float)S, 2.0f) > 0.0f) { lo = (float)(-(S / 2)); hi = (float)(S / 2); }
        else { float e = (float)(S / 2) - 0.5f; lo = -e; hi = e; }
        for (float i = lo; i <= hi; i += 1.0f) for (float j = lo; j <= hi; j += 1.0f) {
            float cx = clamp01(tx + i / offW), cy = clamp01(ty + j / offH);
            int px = (int)floorf(cx * (float)TW), py = (int)floorf(cy * (float)TH);
            if (px > TW - 1) px = TW - 1;
            if (py > TH - 1) py = TH - 1;
            const uint8_t *t = rgba8 + 4 * ((size_t)py * TW + px);
            for (int k = 0; k < 3; k++) acc[k] += (float)t[k] / 255.0f;
            cnt++;
        }
        (void)cnt;
        uint8_t *o = canvas + 4 * ((size_t)y * W + x);
        for (int k = 0; k < 3; k++) {
            float c = acc[k] / (float)(S * S);
            c *= sc->cfg.L_max;
            if (sc->cfg.tr_method == 0) {
                c *= sf;
            } else {
                c = c * sc->cfg.A / Lbar;
                c = c * sc->cfg.L_dmax / (1.0f + c);
            }
            c /= sc->cfg.L_dmax;
            o[k] = quant8(c);
        }
        o[3] = 255;
    }
}

size_t oracle_sizeof_scene(void) { return sizeof(oracle_scene); }
size_t oracle_sizeof_counters(void) { return sizeof(oracle_counters); }
float oracle_pow(float x, float y) { return rt_pow(x, y); }
