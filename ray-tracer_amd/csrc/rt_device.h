// Device-side data layout and the per-ray building blocks (gfx950, wave64).
//
// Arithmetic contract (shared with the CPU checker so that results agree bit for bit): IEEE float32,
// no fused multiply-add (build with -ffp-contract=off), correctly rounded divide/sqrt,
// dot(a,b) = (ax*bx + ay*by) + az*bz, mat4*vec4(p,1) = ((m0*x + m4*y) + m8*z) + m12,
// normalize(v) = v * (1/sqrt(dot(v,v))), pow = exp2(y*log2 x) through fixed double polynomials.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kd_build.h"

namespace rt {

constexpr int WAVE = 64;
constexpr int KD_STACK = 24;      // a root-to-leaf path has at most 23 internal nodes (depth cap 24)
constexpr int MAX_LEVELS = 8;

struct PrimDev {
    int32_t kind, mat;            // mat = index into the material table (-1: none)
    float bound_r2;               // squared radius of a sphere round the model's origin (m[9..11]) that holds the
                                  // primitive, with a margin (see surely_misses); 0 = unbounded (plane, rectangle)
    int32_t pad1;
    float m[12];                  // model: columns 0..3, rows 0..2  (m[3*c + r])
    float im[12];                 // inverse model, same packing
};

struct MatDev {
    int32_t type, pad;
    float ka, kd, ks, ke, kReflect, kRefract, ior, pad2;
    float Co[4], Co2[4], Cs[4];
};

struct LightDev {
    float pos[4], col[4];
};

// Triangle as the traversal wants it: vertex 0 and the two edges the reference's Moller-Trumbore
// forms first (e1 = v1 - v0, e2 = v2 - v0 in float32, js/8-kd_tree.js:505-506), then material and face index.
// `tris` holds these records indexed by face.
struct TriDev {
    float v0[3], e1[3], e2[3];
    int32_t mat;
    int32_t face;
    uint32_t pad;
};
// The walk reads the triangles in leaf order - the (at most four) faces of leaf 0, then those of leaf 1, ... - so
// that a leaf visit can name its triangles as a slot range without reading a leaf record. That table is by far the
// largest thing the walk touches (a face is listed by every leaf it overlaps), so it holds only what a test needs:
// 36 bytes a slot in `lgeo`. What only the winner of a ray needs sits beside it in `ltag`.
struct TriGeo {
    float v0[3], e1[3], e2[3];
};
struct TriTag {
    int32_t mat;
    int32_t face;
    uint32_t code;                // path code of the leaf that lists this slot (see path_code)
    uint32_t leaf;                // which entry of the leaf-cell table (6 floats: lo xyz, hi xyz) is this slot's leaf
};

// Position of a node in the tree as a bit string: one bit per level below the root, 0 = lower child, 1 = upper
// child, left-aligned in PATH_BITS bits and padded with zeros. Leaves are prefix-free, so the code names a leaf.
// For one ray the traversal always enters the NEAR child of a node first (js/8-kd_tree.js:888-930), and which child
// is near depends only on the sign of the direction component of the node's axis; the axis of level i is i % 3
// (js/8-kd_tree.js:872). Flipping the bits of the levels whose axis the ray runs down therefore turns the code into
// a number that grows in the order in which the ray visits the leaves, whatever order they are processed in.
constexpr int PATH_BITS = 23;   // the depth cap of 24 (root = 1) leaves at most 23 decisions
constexpr uint32_t axis_levels(int axis) {
    uint32_t m = 0;
    for (int i = axis; i < PATH_BITS; i += 3) m |= 1u << (PATH_BITS - 1 - i);
    return m;
}
constexpr uint32_t PATH_X = axis_levels(0), PATH_Y = axis_levels(1), PATH_Z = axis_levels(2);
// heap index (children 2h+1, 2h+2, js/8-kd_tree.js:1632-1645): h+1 in binary is a leading 1 followed by the path
inline uint32_t path_code(uint32_t heap) {
    const uint32_t x = heap + 1u;
    int depth = 0;
    while ((x >> depth) > 1u) depth++;
    return depth > PATH_BITS ? 0u : (x ^ (1u << depth)) << (PATH_BITS - depth);
}

// Device copy of a k-d node (see kd_build.h for internal nodes). Leaf info differs from the host form:
// [31:30] = 3, [25:24] = number of faces - 1 (a leaf with faces has 1..4), [23:0] = first slot in `lgeo` / `ltag`.
// The low 26 bits are, as they stand, the record a walking lane puts into its wave's ring of pending triangle tests.
constexpr uint32_t LEAF_COUNT_SHIFT = 24;
constexpr uint32_t LEAF_SLOT_MASK = (1u << LEAF_COUNT_SHIFT) - 1u;
constexpr uint32_t LEAF_RECORD_MASK = (1u << 26) - 1u;
__host__ __device__ inline uint32_t leaf_info(uint32_t n_faces, uint32_t first_slot) {
    return (3u << 30) | ((n_faces - 1u) << LEAF_COUNT_SHIFT) | first_slot;
}

struct SceneDev {
    int32_t n_prims, n_mats, n_lights, n_faces;
    int32_t render_triangles, use_kd, pad0, pad1;
    float inf, eps, sphere_eps, plane_eps, tri_eps, pad2, pad3, pad4;
    float ambient[4], bg[4];
    float cam[4], r00[4], r01[4], r10[4], r11[4];
    float box_lo[4], box_hi[4];
    // Overhang of the tree: the largest amount by which a face listed by a leaf reaches below (over_lo) / above (over_hi)
    // that leaf's cell, per axis, on the float32 vertices and the float32 cells the walk uses, plus a safety margin for
    // the rounding of the triangle test (1/256 of the root box's largest side; the walk adds 1/256 of the distance).
    // Infinity switches pruning off.
    float over_lo[4], over_hi[4];
    const PrimDev *prims;
    const MatDev *mats;
    const LightDev *lights;
    const TriDev *tris;
    const KdNodeDev *nodes;
    const TriGeo *lgeo;
    const TriTag *ltag;
    const uint32_t *path_slot;    // path code of a non-empty leaf -> its first slot in `lgeo` (2^PATH_BITS entries)
};

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 mulv(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ V3 normalize(V3 v) {
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
__device__ __forceinline__ V3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 xpoint(const float *m, V3 p) {
    return v3(((m[0] * p.x + m[3] * p.y) + m[6] * p.z) + m[9], ((m[1] * p.x + m[4] * p.y) + m[7] * p.z) + m[10],
              ((m[2] * p.x + m[5] * p.y) + m[8] * p.z) + m[11]);
}
__device__ __forceinline__ V3 xdir(const float *m, V3 d) {
    return v3((m[0] * d.x + m[3] * d.y) + m[6] * d.z, (m[1] * d.x + m[4] * d.y) + m[7] * d.z,
              (m[2] * d.x + m[5] * d.y) + m[8] * d.z);
}
__device__ __forceinline__ float comp(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
__device__ __forceinline__ float sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
__device__ __forceinline__ float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
__device__ __forceinline__ float glsl_mod(float x, float y) { return x - y * floorf(x / y); }

// pow(x,y), x >= 0, y > 0. Double-precision atanh-series log and Taylor exp with fixed coefficients:
// only IEEE add/mul/div, so host and device produce the same float.
__device__ __forceinline__ float pow_det(float xf, float yf) {
    if (!(xf > 0.0f)) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    double x = (double)xf;
    long long bits = __double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL;
    double m = __longlong_as_double(bits);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0; p = p * s2 + 1.0 / 19.0; p = p * s2 + 1.0 / 17.0; p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0; p = p * s2 + 1.0 / 11.0; p = p * s2 + 1.0 / 9.0;  p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;  p = p * s2 + 1.0 / 3.0;  p = p * s2 + 1.0;
    double ln_m = 2.0 * s * p;
    double lg = (double)e + ln_m * 1.4426950408889634;
    double t = (double)yf * lg;
    if (t < -160.0) return 0.0f;
    if (t > 130.0) return __int_as_float(0x7f800000);
    double n = floor(t + 0.5), f = (t - n) * 0.6931471805599453;
    double q = 1.0 / 87178291200.0;
    q = q * f + 1.0 / 6227020800.0; q = q * f + 1.0 / 479001600.0; q = q * f + 1.0 / 39916800.0;
    q = q * f + 1.0 / 3628800.0;    q = q * f + 1.0 / 362880.0;    q = q * f + 1.0 / 40320.0;
    q = q * f + 1.0 / 5040.0;       q = q * f + 1.0 / 720.0;       q = q * f + 1.0 / 120.0;
    q = q * f + 1.0 / 24.0;         q = q * f + 1.0 / 6.0;         q = q * f + 0.5;
    q = q * f + 1.0;                q = q * f + 1.0;
    double sc = __longlong_as_double((long long)((long long)n + 1023) << 52);
    return (float)(q * sc);
}

// ---- analytic primitives (js/8-kd_tree.js:327-502). On a hit they set t, the object-space hit point
//      `loc` and the (un-normalised) normal `nrm`, exactly the three HitInfo fields the shader writes. ----
struct PHit {
    float t;
    V3 loc, nrm;
};

__device__ __forceinline__ bool hit_sphere(const SceneDev &S, const PrimDev &P, V3 ro, V3 rd, PHit &h) {
    V3 lo = xpoint(P.im, ro), ld = xdir(P.im, rd);
    float a = dot(ld, ld);
    float b = 2.0f * dot(lo, ld);
    float c = dot(lo, lo) - 1.0f;
    float disc = b * b - 4.0f * a * c;
    if (disc > 0.0f) {
        float e = sqrtf(disc), denom = 2.0f * a;
        float t0 = (-b - e) / denom;
        if (t0 > S.sphere_eps) {
            h.t = t0;
            h.loc = lo + ld * t0;
            h.nrm = h.loc;   // near root: object-space normal, not transformed (js/8-kd_tree.js:341-347)
            return true;
        }
        t0 = (-b + e) / denom;
        if (t0 > S.sphere_eps) {
            h.t = t0;
            h.loc = lo + ld * t0;
            h.nrm = xdir(P.m, h.loc);
            return true;
        }
    }
    return false;
}

__device__ __forceinline__ bool hit_plane(const SceneDev &S, const PrimDev &P, V3 ro, V3 rd, PHit &h) {
    V3 lo = xpoint(P.im, ro), ld = xdir(P.im, rd);
    V3 n = v3(0.0f, 1.0f, 0.0f);
    float t0 = dot(-lo, n) / dot(ld, n);
    if (t0 > S.plane_eps) {
        h.t = t0;
        h.loc = lo + ld * t0;
        h.nrm = xdir(P.m, n);
        return true;
    }
    return false;
}

__device__ __forceinline__ bool hit_rect(const SceneDev &S, const PrimDev &P, V3 ro, V3 rd, PHit &h) {
    V3 lo = xpoint(P.im, ro), ld = xdir(P.im, rd);
    V3 A = v3(1.0f, 0.0f, 0.0f), B = v3(0.0f, 0.0f, -1.0f);
    V3 n = cross(A, B);
    V3 p0 = v3(-0.5f, 0.0f, 0.5f);
    float t0 = dot(p0 - lo, n) / dot(ld, n);
    if (t0 > S.plane_eps) {
        V3 p = lo + ld * t0;
        V3 d = p - p0;
        float da = dot(d, A), db = dot(d, B);
        if (da > 0.0f && da < 1.0f && db > 0.0f && db < 1.0f) {
            h.t = t0;
            h.loc = lo + ld * t0;
            h.nrm = xdir(P.m, n);
            return true;
        }
    }
    return false;
}

__device__ __forceinline__ bool hit_cube(const SceneDev &, const PrimDev &P, V3 ro, V3 rd, PHit &h) {
    V3 lo = xpoint(P.im, ro), ld = xdir(P.im, rd);
    V3 od = v3(lo.x / ld.x, lo.y / ld.y, lo.z / ld.z);
    V3 t0 = v3(-0.5f / fabsf(ld.x) - od.x, -0.5f / fabsf(ld.y) - od.y, -0.5f / fabsf(ld.z) - od.z);
    V3 t1 = v3(0.5f / fabsf(ld.x) - od.x, 0.5f / fabsf(ld.y) - od.y, 0.5f / fabsf(ld.z) - od.z);
    float tMin = fmaxf(fmaxf(t0.x, t0.y), t0.z);
    float tMax = fminf(fminf(t1.x, t1.y), t1.z);
    if (tMin > tMax || tMax < 0.0f) return false;
    h.t = tMin;
    h.loc = lo + ld * tMin;
    V3 n = v3(-sgn(ld.x) * stepf(t0.y, t0.x) * stepf(t0.z, t0.x), -sgn(ld.y) * stepf(t0.z, t0.y) * stepf(t0.x, t0.y),
              -sgn(ld.z) * stepf(t0.x, t0.z) * stepf(t0.y, t0.z));
    h.nrm = xdir(P.m, n);
    return true;
}

__device__ __forceinline__ bool hit_cylinder(const SceneDev &, const PrimDev &P, V3 ro, V3 rd, PHit &h) {
    V3 lo = xpoint(P.im, ro), ld = xdir(P.im, rd);
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
                    h.t = th;
                    h.loc = lo + ld * th;
                    h.nrm = xdir(P.m, v3(0.0f, -1.0f, 0.0f));
                    return true;
                }
            }
        } else if (y0 >= -0.5f && y0 <= 0.5f) {
            if (t0 > 0.0f) {
                h.t = t0;
                h.loc = lo + ld * t0;
                h.nrm = xdir(P.m, v3(h.loc.x, 0.0f, h.loc.z));
                return true;
            }
        } else if (y0 > 0.5f) {
            if (y1 <= 0.5f) {
                float th = t0 + (t1 - t0) * (y0 - 0.5f) / (y0 - y1);
                if (th > 0.0f) {
                    h.t = th;
                    h.loc = lo + ld * th;
                    h.nrm = xdir(P.m, v3(0.0f, 1.0f, 0.0f));
                    return true;
                }
            }
        }
    }
    return false;
}

// True only if the ray's line passes the primitive's bounding sphere at a distance: by more than 1 % of the radius
// (in bound_r2) plus 1 % of the distance between the ray's origin and the sphere's centre. The intersection routines
// above then return false - also in float32: what separates hit from miss there (a discriminant, slab distances) is
// then at least 1e-4 of the terms it is computed from, a thousand times their rounding error. Their results are
// unchanged; a wave whose rays all pass the sphere skips the routine. A NaN or a zero direction never says "misses".
// The cube's slab test treats an exactly zero direction component as an unbounded slab (0.5 / 0 - o / 0 = NaN, which
// min and max drop) and can then report a hit far outside the cube: such rays are never filtered.
__device__ __forceinline__ bool surely_misses(const PrimDev &P, V3 ro, V3 rd) {
    const V3 v = v3(P.m[9] - ro.x, P.m[10] - ro.y, P.m[11] - ro.z);
    const V3 cr = cross(v, rd);
    bool far_off = P.bound_r2 > 0.0f && dot(cr, cr) > dot(rd, rd) * (P.bound_r2 + 1.0e-4f * dot(v, v));
    if (P.kind == 3) {   // wave-uniform
        const V3 ld = xdir(P.im, rd);
        far_off = far_off && ld.x != 0.0f && ld.y != 0.0f && ld.z != 0.0f;
    }
    return far_off;
}

__device__ __forceinline__ bool hit_prim(const SceneDev &S, const PrimDev &P, V3 ro, V3 rd, PHit &h) {
    switch (P.kind) {   // wave-uniform
    case 0: return hit_plane(S, P, ro, rd, h);
    case 1: return hit_rect(S, P, ro, rd, h);
    case 2: return hit_sphere(S, P, ro, rd, h);
    case 3: return hit_cube(S, P, ro, rd, h);
    case 4: return hit_cylinder(S, P, ro, rd, h);
    }
    return false;
}

// Moller-Trumbore as js/8-kd_tree.js:504-536 (two-sided, |det| < eps rejected, t < 0 rejected).
__device__ __forceinline__ bool hit_tri_regs(V3 v0, V3 e1, V3 e2, float tri_eps, V3 ro, V3 rd, float &t) {
    V3 tv = ro - v0;
    V3 P = cross(rd, e2), Q = cross(tv, e1);
    float det = dot(P, e1);
    if (fabsf(det) < tri_eps) return false;
    float inv = 1.0f / det;
    t = inv * dot(Q, e2);
    if (t < 0.0f) return false;
    float u = inv * dot(P, tv), v = inv * dot(Q, rd);
    if (u < 0.0f || v < 0.0f || u + v > 1.0f) return false;
    return true;
}
__device__ __forceinline__ bool hit_tri(const TriDev &T, float tri_eps, V3 ro, V3 rd, float &t) {
    return hit_tri_regs(ld3(T.v0), ld3(T.e1), ld3(T.e2), tri_eps, ro, rd, t);
}
__device__ __forceinline__ bool hit_tri(const TriGeo &T, float tri_eps, V3 ro, V3 rd, float &t) {
    return hit_tri_regs(ld3(T.v0), ld3(T.e1), ld3(T.e2), tri_eps, ro, rd, t);
}

// the levels at which this ray's near child is the upper one (`dir <= 0`, js/8-kd_tree.js:888)
__device__ __forceinline__ uint32_t near_upper_levels(V3 rd) {
    return (rd.x <= 0.0f ? PATH_X : 0u) | (rd.y <= 0.0f ? PATH_Y : 0u) | (rd.z <= 0.0f ? PATH_Z : 0u);
}

// intersectRayAABB, js/8-kd_tree.js:545-573 (its boolean result is never used by the callers)
__device__ __forceinline__ void ray_box(const SceneDev &S, V3 ro, V3 rd, float &tMin, float &tMax) {
    tMin = -S.inf;
    tMax = S.inf;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float lo = (S.box_lo[i] - comp(ro, i)) / comp(rd, i);
        float hi = (S.box_hi[i] - comp(ro, i)) / comp(rd, i);
        if (lo > hi) { float tmp = lo; lo = hi; hi = tmp; }
        if (lo > tMin) tMin = lo;
        if (hi < tMax) tMax = hi;
    }
}

}  // namespace rt
