// C ABI (include/rt_hip.h) + the HIP kernels behind it. gfx950 only.
//
// Frame structure. The reference evaluates a per-pixel ray TREE inside one fragment-shader invocation
// (js/8-kd_tree.js:822-1057): slot n spawns a reflected ray into slot 2n+1 and a refracted ray into
// 2n+2, then colours are folded bottom-up. Here the tree is processed one LEVEL at a time over compact
// ray queues: the level-k kernel traces and shades every level-k ray of the whole image and appends the
// children it spawns to the level-(k+1) queue (wave-aggregated append: one atomic per wave, ranks from
// the ballot). Each parent remembers the queue slots of its two children, and the composite pass then
// folds colours level by level in the reference's order (refracted child first, then reflected:
// js/8-kd_tree.js:1041-1054 walks the slots downwards, and 2n+2 > 2n+1), so the float result is the same
// expression tree the shader evaluates.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "kd_build.h"
#include "rt_device.h"

using namespace rt;

namespace {

thread_local std::string g_create_error;

// ------------------------------------------------------------------------------------------------
// device-side queues
struct LevelBuf {
    float *ox, *oy, *oz, *dx, *dy, *dz;  // ray (unused at level 0: primary rays are generated)
    float *cr, *cg, *cb;                 // local colour, later the folded colour
    float *kr, *kt;                      // kReflect / kRefract of this node
    int32_t *refl, *refr;                // queue slot of the child in the next level, -1 = none
};

struct FrameGeom {
    int32_t W, H, tiles_x, n_tiles, shard_index, shard_count;
};

struct TraceArgs {
    const SceneDev *S;
    LevelBuf cur, nxt;
    uint32_t n_rays;      // entries of `cur`
    uint32_t nxt_cap;
    uint32_t *nxt_count;
    unsigned long long *shadow_count;
    int32_t *hit_ids;     // level 0 only, full frame
    FrameGeom g;
    int32_t level, last_level;
};

__device__ __forceinline__ bool pixel_of(const FrameGeom &g, uint32_t tid, int &x, int &y) {
    uint32_t tile = (uint32_t)g.shard_index + (tid >> 6) * (uint32_t)g.shard_count;
    if (tile >= (uint32_t)g.n_tiles) return false;
    uint32_t lane = tid & 63u;
    x = (int)(tile % (uint32_t)g.tiles_x) * 8 + (int)(lane & 7u);
    y = (int)(tile / (uint32_t)g.tiles_x) * 8 + (int)(lane >> 3);
    return x < g.W && y < g.H;
}

__device__ __forceinline__ V3 mix3(V3 a, V3 b, float t) {
    float s = 1.0f - t;
    return v3(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t);
}

// castShadow, js/8-kd_tree.js:575-714
__device__ __forceinline__ bool in_shadow(const SceneDev &S, V3 ro, V3 rd, uint2 *stack) {
    rd = normalize(rd);
    for (int i = 0; i < S.n_prims; i++) {
        PHit h;
        if (hit_prim(S, S.prims[i], ro, rd, h)) return true;
    }
    if (S.render_triangles) {
        if (S.use_kd) {
            float bt = 0.0f;
            int32_t bf = -1;
            if (kd_walk<true>(S, ro, rd, stack, bt, bf)) return true;
        } else {
            for (int f = 0; f < S.n_faces; f++) {
                float t;
                if (hit_tri(S.tris[f], S.tri_eps, ro, rd, t)) return true;
            }
        }
    }
    return false;
}

struct Surface {
    V3 point, normal, local;
    int32_t mat;
};

// illuminate, js/8-kd_tree.js:716-820
__device__ __forceinline__ V3 illuminate(const SceneDev &S, const Surface &sf, V3 rd, uint2 *stack, float &kReflect,
                                         float &kRefract, float &ior) {
    float ka = 0.0f, kd = 0.0f, ks = 0.0f, ke = 0.0f;
    int mtype = -1;
    V3 Co = v3(0, 0, 0), Cs = v3(0, 0, 0);
    kReflect = kRefract = ior = 0.0f;
    if (sf.mat >= 0) {
        const MatDev &m = S.mats[sf.mat];
        mtype = m.type;
        ka = m.ka; kd = m.kd; ks = m.ks; ke = m.ke;
        kReflect = m.kReflect; kRefract = m.kRefract; ior = m.ior;
        Cs = ld3(m.Cs);
        if (m.type == RT_MAT_PHONG_CHECKERED) {
            float u = sf.local.x + 0.5f, v = sf.local.z + 0.5f;
            float sizeU = 1.0f / 12.0f;
            float sizeV = 450.0f / (1600.0f * 12.0f);
            float row = (float)(int)(u / sizeU);
            float col = (float)(int)(v / sizeV);
            bool r0 = (int)glsl_mod(row, 2.0f) == 0, c0 = (int)glsl_mod(col, 2.0f) == 0;
            Co = (r0 == c0) ? ld3(m.Co) : ld3(m.Co2);
        } else {
            Co = ld3(m.Co);
        }
    }
    V3 N = normalize(sf.normal);
    V3 V = normalize(-rd);
    V3 ambient = mulv(ld3(S.ambient) * ka, Co);
    V3 diffuse = v3(0, 0, 0), specular = v3(0, 0, 0);
    for (int i = 0; i < S.n_lights; i++) {
        V3 Lc = ld3(S.lights[i].col);
        V3 Sd = normalize(ld3(S.lights[i].pos) - sf.point);
        if (!in_shadow(S, sf.point + sf.normal * S.eps, Sd, stack)) {
            float SdotN = fmaxf(0.0f, dot(Sd, N));
            float dc = clamp01(kd * SdotN);
            diffuse = diffuse + mulv(Lc * dc, Co);
            if (SdotN > 0.0f) {
                float sp = 0.0f;   // stays 0 for the checkered material (no branch assigns it)
                if (mtype == RT_MAT_PHONG) {
                    V3 I = -Sd;
                    V3 R = normalize(I - N * (2.0f * dot(N, I)));
                    sp = ks * pow_det(fmaxf(0.0f, dot(R, V)), ke);
                } else if (mtype == RT_MAT_PHONG_BLINN) {
                    V3 Hh = normalize(Sd + V);
                    sp = ks * pow_det(fmaxf(0.0f, dot(N, Hh)), ke);
                }
                sp = clamp01(sp);
                specular = specular + mulv(Lc * sp, Cs);
            }
        }
    }
    return v3(0, 0, 0) + ((ambient + specular) + diffuse);
}

__global__ __launch_bounds__(WAVE) void k_trace_level(TraceArgs A) {
    __shared__ uint2 s_stack[KD_STACK * WAVE];
    const SceneDev &S = *A.S;
    const uint32_t tid = blockIdx.x * WAVE + threadIdx.x;
    uint2 *stack = s_stack + threadIdx.x;

    bool active;
    V3 ro, rd;
    int px = 0, py = 0;
    if (A.level == 0) {
        active = tid < A.n_rays && pixel_of(A.g, tid, px, py);
        ro = ld3(S.cam);
        float fx = ((float)px + 0.5f) / (float)A.g.W, fy = ((float)py + 0.5f) / (float)A.g.H;
        V3 dir = mix3(mix3(ld3(S.r00), ld3(S.r01), fy), mix3(ld3(S.r10), ld3(S.r11), fy), fx);
        rd = normalize(dir);
    } else {
        active = tid < A.n_rays;
        uint32_t i = active ? tid : 0;
        ro = v3(A.cur.ox[i], A.cur.oy[i], A.cur.oz[i]);
        rd = v3(A.cur.dx[i], A.cur.dy[i], A.cur.dz[i]);
    }

    V3 color = v3(0, 0, 0);
    float kReflect = 0.0f, kRefract = 0.0f, ior = 0.0f;
    bool want_refl = false, want_refr = false, shaded = false;
    V3 c_refl_o, c_refl_d, c_refr_o, c_refr_d;
    c_refl_o = c_refl_d = c_refr_o = c_refr_d = v3(0, 0, 0);
    int32_t hit_id = -1;

    if (active) {
        // ---- closest hit: analytic primitives in scene order, then the mesh (js/8-kd_tree.js:835-990)
        float tMin = S.inf;
        int32_t best_prim = -1, best_face = -1;
        V3 b_nrm = v3(0, 0, 0), b_loc = v3(0, 0, 0), last_loc = v3(0, 0, 0);
        for (int i = 0; i < S.n_prims; i++) {
            PHit h;
            if (hit_prim(S, S.prims[i], ro, rd, h)) {
                last_loc = h.loc;   // HitInfo.localHitPoint keeps the last successful analytic test
                if (h.t < tMin) {
                    tMin = h.t;
                    best_prim = i;
                    b_nrm = h.nrm;
                    b_loc = h.loc;
                }
            }
        }
        if (S.render_triangles) {
            if (S.use_kd) {
                kd_walk<false>(S, ro, rd, stack, tMin, best_face);
            } else {
                for (int f = 0; f < S.n_faces; f++) {
                    float t;
                    if (hit_tri(S.tris[f], S.tri_eps, ro, rd, t) && t < tMin) { tMin = t; best_face = f; }
                }
            }
        }
        if (best_prim < 0 && best_face < 0) {
            color = ld3(S.bg);
        } else {
            Surface sf;
            sf.point = ro + rd * tMin;
            if (best_face >= 0) {
                const TriDev &T = S.tris[best_face];
                sf.normal = normalize(cross(ld3(T.e1), ld3(T.e2)));
                sf.local = last_loc;   // intersectTriangle never writes localHitPoint (js/8-kd_tree.js:954)
                sf.mat = T.mat;
                hit_id = S.n_prims + best_face;
            } else {
                sf.normal = b_nrm;
                sf.local = b_loc;
                sf.mat = S.prims[best_prim].mat;
                hit_id = best_prim;
            }
            shaded = true;
            color = illuminate(S, sf, rd, stack, kReflect, kRefract, ior);
            // ---- children (js/8-kd_tree.js:1010-1036); neither direction is normalised
            if (!A.last_level) {
                bool outside = dot(rd, sf.normal) < 0.0f;
                V3 bias = sf.normal * S.eps;
                if (kRefract > S.eps && ior > S.eps) {
                    want_refr = true;
                    c_refr_o = outside ? (sf.point - bias) : (sf.point + bias);
                    V3 N = outside ? sf.normal : -sf.normal;
                    float eta = outside ? (1.0f / ior) : ior;
                    float cosi = dot(-N, rd);
                    float k = 1.0f + eta * eta * (cosi * cosi - 1.0f);
                    c_refr_d = rd * eta + N * (eta * cosi - sqrtf(k));
                }
                if (kReflect > S.eps) {
                    want_refl = true;
                    c_refl_o = outside ? (sf.point + bias) : (sf.point - bias);
                    c_refl_d = rd - sf.normal * (2.0f * dot(sf.normal, rd));
                }
            }
        }
    }

    // ---- wave-aggregated append of the children to the next level's queue
    int32_t slot_refl = -1, slot_refr = -1;
    if (!A.last_level) {
        const unsigned long long m_refr = __ballot(want_refr), m_refl = __ballot(want_refl);
        const uint32_t n_refr = (uint32_t)__popcll(m_refr), n_refl = (uint32_t)__popcll(m_refl);
        if (n_refr + n_refl) {
            uint32_t base = 0;
            if (threadIdx.x == 0) base = atomicAdd(A.nxt_count, n_refr + n_refl);
            base = __shfl(base, 0);
            const unsigned long long below = (1ull << threadIdx.x) - 1ull;
            if (want_refr) slot_refr = (int32_t)(base + (uint32_t)__popcll(m_refr & below));
            if (want_refl) slot_refl = (int32_t)(base + n_refr + (uint32_t)__popcll(m_refl & below));
            if (want_refr && (uint32_t)slot_refr < A.nxt_cap) {
                A.nxt.ox[slot_refr] = c_refr_o.x; A.nxt.oy[slot_refr] = c_refr_o.y; A.nxt.oz[slot_refr] = c_refr_o.z;
                A.nxt.dx[slot_refr] = c_refr_d.x; A.nxt.dy[slot_refr] = c_refr_d.y; A.nxt.dz[slot_refr] = c_refr_d.z;
            } else {
                slot_refr = -1;
            }
            if (want_refl && (uint32_t)slot_refl < A.nxt_cap) {
                A.nxt.ox[slot_refl] = c_refl_o.x; A.nxt.oy[slot_refl] = c_refl_o.y; A.nxt.oz[slot_refl] = c_refl_o.z;
                A.nxt.dx[slot_refl] = c_refl_d.x; A.nxt.dy[slot_refl] = c_refl_d.y; A.nxt.dz[slot_refl] = c_refl_d.z;
            } else {
                slot_refl = -1;
            }
        }
    }
    const unsigned long long m_sh = __ballot(shaded);
    if (m_sh && threadIdx.x == 0) atomicAdd(A.shadow_count, (unsigned long long)__popcll(m_sh) * (unsigned long long)S.n_lights);

    if (tid < A.n_rays) {
        A.cur.cr[tid] = color.x; A.cur.cg[tid] = color.y; A.cur.cb[tid] = color.z;
        A.cur.kr[tid] = kReflect; A.cur.kt[tid] = kRefract;
        A.cur.refl[tid] = slot_refl; A.cur.refr[tid] = slot_refr;
        if (A.level == 0 && active && A.hit_ids) A.hit_ids[(size_t)py * A.g.W + px] = hit_id;
    }
}

// parent.color += kRefract * refracted.color, then += kReflect * reflected.color (js/8-kd_tree.js:1041-1054)
__global__ __launch_bounds__(256) void k_composite(LevelBuf cur, LevelBuf nxt, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r = cur.cr[i], g = cur.cg[i], b = cur.cb[i];
    int32_t c = cur.refr[i];
    if (c >= 0) {
        float k = cur.kt[i];
        r = r + k * nxt.cr[c]; g = g + k * nxt.cg[c]; b = b + k * nxt.cb[c];
    }
    c = cur.refl[i];
    if (c >= 0) {
        float k = cur.kr[i];
        r = r + k * nxt.cr[c]; g = g + k * nxt.cg[c]; b = b + k * nxt.cb[c];
    }
    cur.cr[i] = r; cur.cg[i] = g; cur.cb[i] = b;
}

__device__ __forceinline__ uint32_t quant8(float c) {   // GL float -> unorm8: round(clamp(c,0,1)*255)
    if (!(c == c)) return 0u;
    return (uint32_t)(int)(clamp01(c) * 255.0f + 0.5f);
}

// level-0 colours -> float target, RGBA8 target (both full-frame, bottom-up rows) and the shard-compact
// RGBA8 copy (tile order; alpha 0 marks lanes that fall outside the image)
__global__ __launch_bounds__(256) void k_resolve(LevelBuf l0, uint32_t n, FrameGeom g, float *fb, uint32_t *rgba,
                                                 uint32_t *compact) {
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    if (tid >= n) return;
    int x, y;
    if (!pixel_of(g, tid, x, y)) { compact[tid] = 0u; return; }
    float r = l0.cr[tid], gg = l0.cg[tid], b = l0.cb[tid];
    size_t p = (size_t)y * g.W + x;
    fb[3 * p] = r; fb[3 * p + 1] = gg; fb[3 * p + 2] = b;
    uint32_t v = quant8(r) | (quant8(gg) << 8) | (quant8(b) << 16) | 0xff000000u;
    rgba[p] = v;
    compact[tid] = v;
}

// sum over this shard of log(EPSILON + L), L = 0.27 r + 0.67 g + 0.07 b with r = byte * L_MAX / 256
// (js/8-kd_tree.js:2561-2566), double precision. Two deterministic stages: per-block partials, then one block.
__global__ __launch_bounds__(256) void k_tone_partial(const uint32_t *compact, uint32_t n, double L_max, double *part_sum,
                                                      unsigned long long *part_cnt) {
    __shared__ double s_sum[4];
    __shared__ unsigned long long s_cnt[4];
    double acc = 0.0;
    unsigned long long cnt = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint32_t v = compact[i];
        if (v >> 24) {
            double r = (double)(v & 255u) * L_max / 256.0;
            double g = (double)((v >> 8) & 255u) * L_max / 256.0;
            double b = (double)((v >> 16) & 255u) * L_max / 256.0;
            double L = 0.27 * r + 0.67 * g + 0.07 * b;
            acc += log(0.0001 + L);
            cnt++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_down(acc, o);
        cnt += __shfl_down(cnt, o);
    }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = acc; s_cnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
        part_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
}

__global__ __launch_bounds__(64) void k_tone_final(const double *part_sum, const unsigned long long *part_cnt, int n_parts,
                                                   double *out_sum, unsigned long long *out_cnt) {
    double acc = 0.0;
    unsigned long long cnt = 0;
    for (int i = threadIdx.x; i < n_parts; i += 64) { acc += part_sum[i]; cnt += part_cnt[i]; }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_down(acc, o);
        cnt += __shfl_down(cnt, o);
    }
    if (threadIdx.x == 0) { *out_sum = acc; *out_cnt = cnt; }
}

// Present pass, js/8-kd_tree.js:85-143: S x S NEAREST taps (offsets divide by 800*S / 600*S, the WIDTH and
// HEIGHT the reference script holds when it builds the shader string, js/8-kd_tree.js:55,110-111), scale by
// L_MAX, Ward (x sf) or Reinhard, divide by L_DMAX, quantise.
__global__ __launch_bounds__(256) void k_present(const uint32_t *rgba, int W, int H, int S, int method, float L_max,
                                                 float L_dmax, float Akey, float Lbar, float sf, uint32_t *canvas) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const int TW = W * S, TH = H * S;
    const float offW = (float)(800 * S), offH = (float)(600 * S);
    float tx = ((float)x + 0.5f) / (float)W, ty = ((float)y + 0.5f) / (float)H;
    float lo, hi;
    if (glsl_mod((float)S, 2.0f) > 0.0f) { lo = (float)(-(S / 2)); hi = (float)(S / 2); }
    else { float e = (float)(S / 2) - 0.5f; lo = -e; hi = e; }
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (float i = lo; i <= hi; i += 1.0f)
        for (float j = lo; j <= hi; j += 1.0f) {
            float cx = clamp01(tx + i / offW), cy = clamp01(ty + j / offH);
            int sx = (int)floorf(cx * (float)TW), sy = (int)floorf(cy * (float)TH);
            sx = sx > TW - 1 ? TW - 1 : sx;
            sy = sy > TH - 1 ? TH - 1 : sy;
            uint32_t v = rgba[(size_t)sy * TW + sx];
            acc[0] += (float)(v & 255u) / 255.0f;
            acc[1] += (float)((v >> 8) & 255u) / 255.0f;
            acc[2] += (float)((v >> 16) & 255u) / 255.0f;
        }
    uint32_t out = 0xff000000u;
    for (int k = 0; k < 3; k++) {
        float c = acc[k] / (float)(S * S);
        c *= L_max;
        if (method == 0) {
            c *= sf;
        } else {
            c = c * Akey / Lbar;
            c = c * L_dmax / (1.0f + c);
        }
        c /= L_dmax;
        out |= quant8(c) << (8 * k);
    }
    canvas[(size_t)y * W + x] = out;
}

// scatter gathered shard-compact buffers (each `stride` pixels long) into the full-frame RGBA8 target
__global__ __launch_bounds__(256) void k_assemble(const uint32_t *gathered, uint32_t stride, int n_shards, FrameGeom g,
                                                  uint32_t *rgba) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= stride * (uint32_t)n_shards) return;
    FrameGeom gs = g;
    gs.shard_index = (int)(i / stride);
    gs.shard_count = n_shards;
    int x, y;
    if (!pixel_of(gs, i % stride, x, y)) return;
    rgba[(size_t)y * g.W + x] = gathered[i];
}

// ------------------------------------------------------------------------------------------------
struct Level {
    void *mem = nullptr;
    uint32_t cap = 0;
    LevelBuf b{};
};

}  // namespace

struct rt_ctx {
    rt_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2 * MAX_LEVELS + 4]{};
    // scene (host copies)
    std::vector<PrimDev> prims;
    std::vector<rt_material> mats_host;
    std::vector<MatDev> mats;
    std::vector<LightDev> lights;
    std::vector<rt_prim> prims_host;
    std::vector<double> verts;
    std::vector<uint32_t> faces, face_mat;
    KdTreeHost kd;
    bool kd_built = false, scene_dirty = true, mesh_dirty = true;
    SceneDev sh{};   // host mirror of the device scene header
    // device
    SceneDev *d_scene = nullptr;
    PrimDev *d_prims = nullptr;
    MatDev *d_mats = nullptr;
    LightDev *d_lights = nullptr;
    TriDev *d_tris = nullptr;
    KdNodeDev *d_nodes = nullptr;
    KdLeafDev *d_leaves = nullptr;
    size_t cap_prims = 0, cap_mats = 0, cap_lights = 0, cap_tris = 0, cap_nodes = 0, cap_leaves = 0;
    Level lv[MAX_LEVELS];
    uint32_t *d_counts = nullptr;                // per-level ray counts
    unsigned long long *d_shadow = nullptr;
    uint32_t n_rays[MAX_LEVELS]{};
    // frame
    FrameGeom g{};
    uint32_t n0 = 0;                             // level-0 entries = tiles of this shard * 64
    float *d_fb = nullptr;
    uint32_t *d_rgba = nullptr, *d_compact = nullptr, *d_canvas = nullptr;
    int32_t *d_hit = nullptr;
    double *d_part_sum = nullptr, *d_tone_out = nullptr;
    unsigned long long *d_part_cnt = nullptr, *d_tone_cnt = nullptr;
    double Lbar = 0.0, sf = 0.0;
    bool traced = false;
    bool host_only = false;
    rt_stats st{};
};

namespace {

constexpr int TONE_BLOCKS = 1024;

#define HIPCK(call)                                                                                            \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                                      \
            return -1;                                                                                         \
        }                                                                                                      \
    } while (0)

int fail(rt_ctx *ctx, const char *msg) {
    ctx->err = msg;
    return -1;
}

template <class T>
int ensure(rt_ctx *ctx, T *&ptr, size_t &cap, size_t n) {
    if (n <= cap && ptr) return 0;
    if (ptr) HIPCK(hipFree(ptr));
    ptr = nullptr;
    cap = std::max<size_t>(n, 1);
    HIPCK(hipMalloc((void **)&ptr, cap * sizeof(T)));
    return 0;
}

int ensure_level(rt_ctx *ctx, int l, uint32_t cap) {
    Level &L = ctx->lv[l];
    if (L.mem && L.cap >= cap) return 0;
    if (L.mem) HIPCK(hipFree(L.mem));
    L.mem = nullptr;
    cap = std::max<uint32_t>((cap + 255u) & ~255u, 256u);
    HIPCK(hipMalloc(&L.mem, (size_t)cap * 13 * sizeof(float)));
    L.cap = cap;
    float *p = (float *)L.mem;
    float **f[] = {&L.b.ox, &L.b.oy, &L.b.oz, &L.b.dx, &L.b.dy, &L.b.dz, &L.b.cr, &L.b.cg, &L.b.cb, &L.b.kr, &L.b.kt};
    for (auto q : f) { *q = p; p += cap; }
    L.b.refl = (int32_t *)p; p += cap;
    L.b.refr = (int32_t *)p;
    return 0;
}

int upload_scene(rt_ctx *ctx) {
    // material index per primitive / per face is resolved here: (type, id) -> slot in the table
    auto find_mat = [&](int type, int id) -> int32_t {
        for (size_t i = 0; i < ctx->mats_host.size(); i++)
            if (ctx->mats_host[i].type == type && ctx->mats_host[i].id == id) return (int32_t)i;
        return -1;
    };
    ctx->prims.resize(ctx->prims_host.size());
    for (size_t i = 0; i < ctx->prims_host.size(); i++) {
        const rt_prim &s = ctx->prims_host[i];
        PrimDev &d = ctx->prims[i];
        d.kind = s.kind;
        d.mat = find_mat(s.mat_type, s.mat_id);
        d.pad0 = d.pad1 = 0;
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 3; r++) { d.m[3 * c + r] = s.model[4 * c + r]; d.im[3 * c + r] = s.inv_model[4 * c + r]; }
    }
    ctx->mats.resize(ctx->mats_host.size());
    for (size_t i = 0; i < ctx->mats_host.size(); i++) {
        const rt_material &s = ctx->mats_host[i];
        MatDev &d = ctx->mats[i];
        memset(&d, 0, sizeof d);
        d.type = s.type;
        d.ka = s.ka; d.kd = s.kd; d.ks = s.ks; d.ke = s.ke; d.kReflect = s.kReflect; d.kRefract = s.kRefract; d.ior = s.ior;
        for (int k = 0; k < 3; k++) { d.Co[k] = s.Co[k]; d.Co2[k] = s.Co2[k]; d.Cs[k] = s.Cs[k]; }
    }
    if (ensure(ctx, ctx->d_prims, ctx->cap_prims, ctx->prims.size())) return -1;
    if (ensure(ctx, ctx->d_mats, ctx->cap_mats, ctx->mats.size())) return -1;
    if (ensure(ctx, ctx->d_lights, ctx->cap_lights, ctx->lights.size())) return -1;
    if (!ctx->prims.empty()) HIPCK(hipMemcpyAsync(ctx->d_prims, ctx->prims.data(), ctx->prims.size() * sizeof(PrimDev), hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->mats.empty()) HIPCK(hipMemcpyAsync(ctx->d_mats, ctx->mats.data(), ctx->mats.size() * sizeof(MatDev), hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->lights.empty()) HIPCK(hipMemcpyAsync(ctx->d_lights, ctx->lights.data(), ctx->lights.size() * sizeof(LightDev), hipMemcpyHostToDevice, ctx->stream));

    if (ctx->mesh_dirty) {
        // triangles: float32 narrowing of the vertices (js/8-kd_tree.js:2462-2464), edges formed in float32
        const size_t nf = ctx->faces.size() / 3;
        std::vector<TriDev> tris(nf);
        for (size_t f = 0; f < nf; f++) {
            float v[3][3];
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) v[j][k] = (float)ctx->verts[3 * (size_t)ctx->faces[3 * f + j] + k];
            TriDev &t = tris[f];
            for (int k = 0; k < 3; k++) { t.v0[k] = v[0][k]; t.e1[k] = v[1][k] - v[0][k]; t.e2[k] = v[2][k] - v[0][k]; }
            t.mat = find_mat((int)ctx->face_mat[2 * f], (int)ctx->face_mat[2 * f + 1]);
            t.pad[0] = t.pad[1] = 0;
        }
        if (ensure(ctx, ctx->d_tris, ctx->cap_tris, nf)) return -1;
        if (nf) HIPCK(hipMemcpyAsync(ctx->d_tris, tris.data(), nf * sizeof(TriDev), hipMemcpyHostToDevice, ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));   // `tris` is a local
        ctx->mesh_dirty = false;
    }

    SceneDev &h = ctx->sh;
    const rt_config &c = ctx->cfg;
    h.n_prims = (int32_t)ctx->prims.size();
    h.n_mats = (int32_t)ctx->mats.size();
    h.n_lights = (int32_t)ctx->lights.size();
    h.n_faces = (int32_t)(ctx->faces.size() / 3);
    h.render_triangles = (c.render_triangles && h.n_faces > 0) ? 1 : 0;
    h.use_kd = c.use_kd_tree ? 1 : 0;
    h.inf = c.infinity; h.eps = c.epsilon; h.sphere_eps = c.sphere_epsilon; h.plane_eps = c.plane_rect_epsilon;
    h.tri_eps = c.triangle_epsilon;
    if (h.render_triangles && h.use_kd) {
        if (!ctx->kd_built) return fail(ctx, "rt_trace: mesh set but rt_build_kdtree was not called");
        for (int k = 0; k < 3; k++) {   // uniform3fv narrowing of root.min / root.max (js/8-kd_tree.js:1114, :2519-2520)
            h.box_lo[k] = (float)ctx->kd.root_box[k];
            h.box_hi[k] = (float)ctx->kd.root_box[3 + k];
        }
    }
    h.prims = ctx->d_prims; h.mats = ctx->d_mats; h.lights = ctx->d_lights; h.tris = ctx->d_tris;
    h.nodes = ctx->d_nodes; h.leaves = ctx->d_leaves;
    HIPCK(hipMemcpyAsync(ctx->d_scene, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    HIPCK(hipStreamSynchronize(ctx->stream));
    ctx->scene_dirty = false;
    return 0;
}

}  // namespace

// ================================================================================================
extern "C" {

void rt_default_config(rt_config *c) {
    memset(c, 0, sizeof *c);
    c->width = 640; c->height = 480; c->supersampling = 1; c->max_depth = 1; c->tr_method = 0;
    c->render_triangles = 1; c->use_kd_tree = 1; c->max_kd_depth = 24;
    c->L_max = 1000.0f; c->L_dmax = 500.0f; c->A = 0.18f;
    c->infinity = 100000.0f; c->epsilon = 0.0001f; c->sphere_epsilon = 0.095f; c->plane_rect_epsilon = 0.095f;
    c->triangle_epsilon = 0.0001f;
    c->shard_index = 0; c->shard_count = 1; c->device = 0;
}

const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_create(const rt_config *cfg, rt_ctx **out) {
    if (!cfg || !out) { g_create_error = "rt_create: null argument"; return -1; }
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->supersampling <= 0 || cfg->supersampling > 8) {
        g_create_error = "rt_create: width, height must be positive and supersampling in 1..8"; return -1;
    }
    if (cfg->max_depth < 1 || cfg->max_depth > MAX_LEVELS) { g_create_error = "rt_create: max_depth must be in 1..8"; return -1; }
    if (cfg->max_kd_depth < 1) { g_create_error = "rt_create: max_kd_depth must be >= 1"; return -1; }
    int sc = cfg->shard_count <= 1 ? 1 : cfg->shard_count;
    if (cfg->shard_index < 0 || cfg->shard_index >= sc) { g_create_error = "rt_create: shard_index out of range"; return -1; }
    if (cfg->device == -1) {   // host-only context: mesh + k-d build + export, nothing that touches a GPU
        rt_ctx *h = new rt_ctx();
        h->cfg = *cfg;
        h->cfg.shard_count = sc;
        h->host_only = true;
        *out = h;
        return 0;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "rt_create: no HIP device available"; return -1; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "rt_create: device ordinal out of range"; return -1; }
    rt_ctx *ctx = new rt_ctx();
    ctx->cfg = *cfg;
    ctx->cfg.shard_count = sc;
    auto bail = [&](const char *what, hipError_t e) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(e);
        rt_destroy(ctx);
        return -1;
    };
    hipError_t e;
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess) return bail("hipStreamCreate", e);
    for (auto &ev : ctx->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    FrameGeom &g = ctx->g;
    g.W = cfg->width * cfg->supersampling;
    g.H = cfg->height * cfg->supersampling;
    g.tiles_x = (g.W + 7) / 8;
    g.n_tiles = g.tiles_x * ((g.H + 7) / 8);
    g.shard_index = cfg->shard_index;
    g.shard_count = sc;
    uint32_t my_tiles = (uint32_t)((g.n_tiles - g.shard_index + sc - 1) / sc);
    ctx->n0 = my_tiles * 64u;
    const size_t npx = (size_t)g.W * g.H;
#define ALLOC(p, bytes)                                                                    \
    if ((e = hipMalloc((void **)&(p), (bytes))) != hipSuccess) return bail("hipMalloc", e); \
    if ((e = hipMemset((p), 0, (bytes))) != hipSuccess) return bail("hipMemset", e)
    ALLOC(ctx->d_scene, sizeof(SceneDev));
    ALLOC(ctx->d_fb, npx * 3 * sizeof(float));
    ALLOC(ctx->d_rgba, npx * 4);
    ALLOC(ctx->d_hit, npx * 4);
    ALLOC(ctx->d_compact, std::max<size_t>(ctx->n0, 64) * 4);
    ALLOC(ctx->d_canvas, (size_t)cfg->width * cfg->height * 4);
    ALLOC(ctx->d_counts, MAX_LEVELS * sizeof(uint32_t));
    ALLOC(ctx->d_shadow, sizeof(unsigned long long));
    ALLOC(ctx->d_part_sum, TONE_BLOCKS * sizeof(double));
    ALLOC(ctx->d_part_cnt, TONE_BLOCKS * sizeof(unsigned long long));
    ALLOC(ctx->d_tone_out, sizeof(double));
    ALLOC(ctx->d_tone_cnt, sizeof(unsigned long long));
#undef ALLOC
    *out = ctx;
    return 0;
}

void rt_destroy(rt_ctx *ctx) {
    if (!ctx) return;
    if (ctx->host_only) { delete ctx; return; }
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    void *ptrs[] = {ctx->d_scene, ctx->d_prims, ctx->d_mats, ctx->d_lights, ctx->d_tris, ctx->d_nodes, ctx->d_leaves,
                    ctx->d_fb, ctx->d_rgba, ctx->d_hit, ctx->d_compact, ctx->d_canvas, ctx->d_counts, ctx->d_shadow,
                    ctx->d_part_sum, ctx->d_part_cnt, ctx->d_tone_out, ctx->d_tone_cnt};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &l : ctx->lv)
        if (l.mem) (void)hipFree(l.mem);
    for (auto &ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rt_set_primitives(rt_ctx *ctx, const rt_prim *prims, int32_t n) {
    if (!ctx) return -1;
    if (n < 0 || (n > 0 && !prims)) return fail(ctx, "rt_set_primitives: bad arguments");
    for (int i = 0; i < n; i++)
        if (prims[i].kind < RT_PRIM_PLANE || prims[i].kind > RT_PRIM_CYLINDER) return fail(ctx, "rt_set_primitives: unknown primitive kind");
    ctx->prims_host.assign(prims, prims + n);
    ctx->scene_dirty = true;
    return 0;
}

int rt_set_materials(rt_ctx *ctx, const rt_material *mats, int32_t n) {
    if (!ctx) return -1;
    if (n < 0 || (n > 0 && !mats)) return fail(ctx, "rt_set_materials: bad arguments");
    for (int i = 0; i < n; i++)
        if (mats[i].type < RT_MAT_PHONG || mats[i].type > RT_MAT_PHONG_CHECKERED) return fail(ctx, "rt_set_materials: unknown material type");
    ctx->mats_host.assign(mats, mats + n);
    ctx->scene_dirty = true;
    ctx->mesh_dirty = true;   // per-face material slots are resolved against this table
    return 0;
}

int rt_set_lights(rt_ctx *ctx, const rt_light *lights, int32_t n, int32_t n_loop) {
    if (!ctx) return -1;
    if (n < 0 || (n > 0 && !lights) || n_loop < n) return fail(ctx, "rt_set_lights: bad arguments (need n >= 0, n_loop >= n)");
    ctx->lights.assign((size_t)n_loop, LightDev{{0, 0, 0, 0}, {0, 0, 0, 0}});
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { ctx->lights[i].pos[k] = lights[i].position[k]; ctx->lights[i].col[k] = lights[i].color[k]; }
    ctx->scene_dirty = true;
    return 0;
}

int rt_set_globals(rt_ctx *ctx, const float ambient[3], const float bg[3]) {
    if (!ctx) return -1;
    if (!ambient || !bg) return fail(ctx, "rt_set_globals: null argument");
    for (int k = 0; k < 3; k++) { ctx->sh.ambient[k] = ambient[k]; ctx->sh.bg[k] = bg[k]; }
    ctx->scene_dirty = true;
    return 0;
}

int rt_set_camera(rt_ctx *ctx, const float pos[3], const float r00[3], const float r01[3], const float r10[3],
                  const float r11[3]) {
    if (!ctx) return -1;
    if (!pos || !r00 || !r01 || !r10 || !r11) return fail(ctx, "rt_set_camera: null argument");
    for (int k = 0; k < 3; k++) {
        ctx->sh.cam[k] = pos[k]; ctx->sh.r00[k] = r00[k]; ctx->sh.r01[k] = r01[k]; ctx->sh.r10[k] = r10[k]; ctx->sh.r11[k] = r11[k];
    }
    ctx->scene_dirty = true;
    return 0;
}

int rt_set_mesh(rt_ctx *ctx, const double *verts, int32_t n_verts, const uint32_t *faces, const uint32_t *face_mat,
                int32_t n_faces) {
    if (!ctx) return -1;
    if (n_verts < 0 || n_faces < 0 || (n_verts > 0 && !verts) || (n_faces > 0 && (!faces || !face_mat)))
        return fail(ctx, "rt_set_mesh: bad arguments");
    for (int64_t i = 0; i < (int64_t)n_faces * 3; i++)
        if (faces[i] >= (uint32_t)n_verts) return fail(ctx, "rt_set_mesh: face index out of range");
    ctx->verts.assign(verts, verts + (size_t)n_verts * 3);
    ctx->faces.assign(faces, faces + (size_t)n_faces * 3);
    ctx->face_mat.assign(face_mat, face_mat + (size_t)n_faces * 2);
    ctx->kd_built = false;
    ctx->mesh_dirty = true;
    ctx->scene_dirty = true;
    return 0;
}

int rt_build_kdtree(rt_ctx *ctx) {
    if (!ctx) return -1;
    auto t0 = std::chrono::steady_clock::now();
    build_kdtree(ctx->verts.data(), (int64_t)ctx->verts.size() / 3, ctx->faces.data(), (int64_t)ctx->faces.size() / 3,
                 ctx->cfg.max_kd_depth, (double)ctx->cfg.infinity, 0, ctx->kd);
    auto t1 = std::chrono::steady_clock::now();
    if (ctx->host_only) {
        ctx->st.kd_build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        ctx->kd_built = true;
        return 0;
    }
    HIPCK(hipSetDevice(ctx->cfg.device));
    if (ensure(ctx, ctx->d_nodes, ctx->cap_nodes, ctx->kd.nodes.size())) return -1;
    if (ensure(ctx, ctx->d_leaves, ctx->cap_leaves, ctx->kd.leaves.size())) return -1;
    HIPCK(hipMemcpyAsync(ctx->d_nodes, ctx->kd.nodes.data(), ctx->kd.nodes.size() * sizeof(KdNodeDev), hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->kd.leaves.empty())
        HIPCK(hipMemcpyAsync(ctx->d_leaves, ctx->kd.leaves.data(), ctx->kd.leaves.size() * sizeof(KdLeafDev), hipMemcpyHostToDevice, ctx->stream));
    HIPCK(hipStreamSynchronize(ctx->stream));
    auto t2 = std::chrono::steady_clock::now();
    ctx->st.kd_build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    ctx->st.kd_upload_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    ctx->kd_built = true;
    ctx->scene_dirty = true;
    return 0;
}

int rt_kdtree_counts(rt_ctx *ctx, int64_t *n_nodes, int64_t *n_leaves) {
    if (!ctx) return -1;
    if (!ctx->kd_built) return fail(ctx, "rt_kdtree_counts: no tree built");
    if (n_nodes) *n_nodes = (int64_t)ctx->kd.nodes.size();
    if (n_leaves) *n_leaves = (int64_t)ctx->kd.leaves.size();
    return 0;
}

int rt_export_kdtree_reference(rt_ctx *ctx, uint32_t *nodes, int32_t *leaves, double root_box[6]) {
    if (!ctx) return -1;
    if (!ctx->kd_built) return fail(ctx, "rt_export_kdtree_reference: no tree built");
    if (nodes) export_reference(ctx->kd, nodes);
    if (leaves && !ctx->kd.leaves.empty()) memcpy(leaves, ctx->kd.leaves.data(), ctx->kd.leaves.size() * sizeof(KdLeafDev));
    if (root_box) memcpy(root_box, ctx->kd.root_box, sizeof ctx->kd.root_box);
    return 0;
}

int rt_trace(rt_ctx *ctx) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_trace: host-only context (created with device = -1)");
    HIPCK(hipSetDevice(ctx->cfg.device));
    if (ctx->scene_dirty && upload_scene(ctx)) return -1;
    const int D = ctx->cfg.max_depth;
    hipStream_t s = ctx->stream;
    HIPCK(hipMemsetAsync(ctx->d_counts, 0, MAX_LEVELS * sizeof(uint32_t), s));
    HIPCK(hipMemsetAsync(ctx->d_shadow, 0, sizeof(unsigned long long), s));
    memset(ctx->n_rays, 0, sizeof ctx->n_rays);
    ctx->n_rays[0] = ctx->n0;
    if (ensure_level(ctx, 0, ctx->n0)) return -1;
    hipEvent_t ev_begin = ctx->ev[2 * MAX_LEVELS], ev_end = ctx->ev[2 * MAX_LEVELS + 1], ev_comp = ctx->ev[2 * MAX_LEVELS + 2];
    HIPCK(hipEventRecord(ev_begin, s));
    int levels_run = 0;
    for (int l = 0; l < D; l++) {
        const uint32_t n = ctx->n_rays[l];
        if (n == 0) break;
        const bool last = (l == D - 1);
        if (!last) {
            uint64_t want = std::min<uint64_t>(2ull * n, 0x7fffff00ull);
            if (ensure_level(ctx, l + 1, (uint32_t)want)) return -1;
        }
        TraceArgs A{};
        A.S = ctx->d_scene;
        A.cur = ctx->lv[l].b;
        if (!last) { A.nxt = ctx->lv[l + 1].b; A.nxt_cap = ctx->lv[l + 1].cap; }
        A.n_rays = n;
        A.nxt_count = ctx->d_counts + (l + 1 < MAX_LEVELS ? l + 1 : 0);
        A.shadow_count = ctx->d_shadow;
        A.hit_ids = ctx->d_hit;
        A.g = ctx->g;
        A.level = l;
        A.last_level = last ? 1 : 0;
        HIPCK(hipEventRecord(ctx->ev[2 * l], s));
        hipLaunchKernelGGL(k_trace_level, dim3((n + WAVE - 1) / WAVE), dim3(WAVE), 0, s, A);
        HIPCK(hipGetLastError());
        HIPCK(hipEventRecord(ctx->ev[2 * l + 1], s));
        levels_run = l + 1;
        if (!last) {
            uint32_t cnt = 0;
            HIPCK(hipMemcpyAsync(&cnt, ctx->d_counts + l + 1, sizeof cnt, hipMemcpyDeviceToHost, s));
            HIPCK(hipStreamSynchronize(s));
            ctx->n_rays[l + 1] = std::min(cnt, ctx->lv[l + 1].cap);
        }
    }
    HIPCK(hipEventRecord(ev_comp, s));
    for (int l = levels_run - 2; l >= 0; l--) {
        if (ctx->n_rays[l + 1] == 0) continue;
        hipLaunchKernelGGL(k_composite, dim3((ctx->n_rays[l] + 255) / 256), dim3(256), 0, s, ctx->lv[l].b, ctx->lv[l + 1].b,
                           ctx->n_rays[l]);
    }
    hipLaunchKernelGGL(k_resolve, dim3((ctx->n0 + 255) / 256), dim3(256), 0, s, ctx->lv[0].b, ctx->n0, ctx->g, ctx->d_fb,
                       ctx->d_rgba, ctx->d_compact);
    HIPCK(hipGetLastError());
    HIPCK(hipEventRecord(ev_end, s));
    unsigned long long sh = 0;
    HIPCK(hipMemcpyAsync(&sh, ctx->d_shadow, sizeof sh, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));

    rt_stats &st = ctx->st;
    st.rays_total = 0;
    memset(st.rays_per_level, 0, sizeof st.rays_per_level);
    memset(st.level_kernel_ms, 0, sizeof st.level_kernel_ms);
    // level 0 holds whole tiles; lanes outside the image are not rays
    {
        uint64_t px = 0;
        const FrameGeom &g = ctx->g;
        for (int t = g.shard_index; t < g.n_tiles; t += g.shard_count) {
            int tx = t % g.tiles_x, ty = t / g.tiles_x;
            px += (uint64_t)std::min(8, g.W - tx * 8) * (uint64_t)std::min(8, g.H - ty * 8);
        }
        st.pixels = px;
        st.rays_per_level[0] = px;
    }
    for (int l = 1; l < levels_run; l++) st.rays_per_level[l] = ctx->n_rays[l];
    for (int l = 0; l < levels_run; l++) {
        st.rays_total += st.rays_per_level[l];
        float ms = 0;
        HIPCK(hipEventElapsedTime(&ms, ctx->ev[2 * l], ctx->ev[2 * l + 1]));
        st.level_kernel_ms[l] = ms;
    }
    st.shadow_rays = sh;
    float ms = 0;
    HIPCK(hipEventElapsedTime(&ms, ev_begin, ev_end));
    st.trace_ms = ms;
    HIPCK(hipEventElapsedTime(&ms, ev_comp, ev_end));
    st.composite_ms = ms;
    ctx->traced = true;
    return 0;
}

int rt_read_float(rt_ctx *ctx, float *rgb) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_read_float: host-only context (created with device = -1)");
    if (!rgb) return fail(ctx, "rt_read_float: null buffer");
    if (!ctx->traced) return fail(ctx, "rt_read_float: nothing traced yet");
    HIPCK(hipSetDevice(ctx->cfg.device));
    HIPCK(hipMemcpy(rgb, ctx->d_fb, (size_t)ctx->g.W * ctx->g.H * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int rt_read_rgba8(rt_ctx *ctx, uint8_t *rgba) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_read_rgba8: host-only context (created with device = -1)");
    if (!rgba) return fail(ctx, "rt_read_rgba8: null buffer");
    if (!ctx->traced) return fail(ctx, "rt_read_rgba8: nothing traced yet");
    HIPCK(hipSetDevice(ctx->cfg.device));
    HIPCK(hipMemcpy(rgba, ctx->d_rgba, (size_t)ctx->g.W * ctx->g.H * 4, hipMemcpyDeviceToHost));
    return 0;
}

int rt_read_hit_ids(rt_ctx *ctx, int32_t *ids) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_read_hit_ids: host-only context (created with device = -1)");
    if (!ids) return fail(ctx, "rt_read_hit_ids: null buffer");
    if (!ctx->traced) return fail(ctx, "rt_read_hit_ids: nothing traced yet");
    HIPCK(hipSetDevice(ctx->cfg.device));
    HIPCK(hipMemcpy(ids, ctx->d_hit, (size_t)ctx->g.W * ctx->g.H * 4, hipMemcpyDeviceToHost));
    return 0;
}

int rt_tone_partial(rt_ctx *ctx, double *sum_log, uint64_t *count) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_tone_partial: host-only context (created with device = -1)");
    if (!sum_log || !count) return fail(ctx, "rt_tone_partial: null argument");
    if (!ctx->traced) return fail(ctx, "rt_tone_partial: nothing traced yet");
    HIPCK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->stream;
    hipEvent_t e0 = ctx->ev[2 * MAX_LEVELS], e1 = ctx->ev[2 * MAX_LEVELS + 1];
    HIPCK(hipEventRecord(e0, s));
    int blocks = (int)std::min<uint32_t>((ctx->n0 + 255) / 256, TONE_BLOCKS);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_tone_partial, dim3(blocks), dim3(256), 0, s, ctx->d_compact, ctx->n0, (double)ctx->cfg.L_max,
                       ctx->d_part_sum, ctx->d_part_cnt);
    hipLaunchKernelGGL(k_tone_final, dim3(1), dim3(64), 0, s, ctx->d_part_sum, ctx->d_part_cnt, blocks, ctx->d_tone_out,
                       ctx->d_tone_cnt);
    HIPCK(hipGetLastError());
    HIPCK(hipEventRecord(e1, s));
    unsigned long long c = 0;
    HIPCK(hipMemcpyAsync(sum_log, ctx->d_tone_out, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCK(hipMemcpyAsync(&c, ctx->d_tone_cnt, sizeof c, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    *count = c;
    float ms = 0;
    HIPCK(hipEventElapsedTime(&ms, e0, e1));
    ctx->st.tone_ms = ms;
    return 0;
}

int rt_tone_finish(rt_ctx *ctx, double sum_log, uint64_t count, double *Lbar, double *sf) {
    if (!ctx) return -1;
    if (count == 0) return fail(ctx, "rt_tone_finish: zero pixels");
    // Lbar = exp(sum / N); sf = ((1.219 + (Ldmax/2)^0.4) / (1.219 + Lbar^0.4))^2.5  (js/8-kd_tree.js:2568-2569)
    ctx->Lbar = exp(sum_log / (double)count);
    ctx->sf = pow((1.219 + pow((double)ctx->cfg.L_dmax * 0.5, 0.4)) / (1.219 + pow(ctx->Lbar, 0.4)), 2.5);
    if (Lbar) *Lbar = ctx->Lbar;
    if (sf) *sf = ctx->sf;
    return 0;
}

int rt_tone(rt_ctx *ctx, double *Lbar, double *sf) {
    double s = 0;
    uint64_t n = 0;
    if (rt_tone_partial(ctx, &s, &n)) return -1;
    return rt_tone_finish(ctx, s, n, Lbar, sf);
}

int rt_present(rt_ctx *ctx, uint8_t *canvas) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_present: host-only context (created with device = -1)");
    if (!ctx->traced) return fail(ctx, "rt_present: nothing traced yet");
    HIPCK(hipSetDevice(ctx->cfg.device));
    const rt_config &c = ctx->cfg;
    dim3 grid((c.width + 15) / 16, (c.height + 15) / 16);
    // Lbar and sf reach the shader as float uniforms (js/8-kd_tree.js:2579-2581)
    hipLaunchKernelGGL(k_present, grid, dim3(256), 0, ctx->stream, ctx->d_rgba, c.width, c.height, c.supersampling, c.tr_method,
                       c.L_max, c.L_dmax, c.A, (float)ctx->Lbar, (float)ctx->sf, ctx->d_canvas);
    HIPCK(hipGetLastError());
    if (canvas) HIPCK(hipMemcpyAsync(canvas, ctx->d_canvas, (size_t)c.width * c.height * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int rt_device_rgba8(rt_ctx *ctx, void **dptr, size_t *bytes) {
    if (!ctx) return -1;
    if (dptr) *dptr = ctx->d_rgba;
    if (bytes) *bytes = (size_t)ctx->g.W * ctx->g.H * 4;
    return 0;
}

int rt_device_float(rt_ctx *ctx, void **dptr, size_t *bytes) {
    if (!ctx) return -1;
    if (dptr) *dptr = ctx->d_fb;
    if (bytes) *bytes = (size_t)ctx->g.W * ctx->g.H * 3 * sizeof(float);
    return 0;
}

int rt_device_canvas(rt_ctx *ctx, void **dptr, size_t *bytes) {
    if (!ctx) return -1;
    if (dptr) *dptr = ctx->d_canvas;
    if (bytes) *bytes = (size_t)ctx->cfg.width * ctx->cfg.height * 4;
    return 0;
}

int rt_device_shard_rgba8(rt_ctx *ctx, void **dptr, size_t *bytes) {
    if (!ctx) return -1;
    if (dptr) *dptr = ctx->d_compact;
    if (bytes) *bytes = (size_t)ctx->n0 * 4;
    return 0;
}

int rt_assemble_shards(rt_ctx *ctx, const void *gathered, size_t stride_bytes, int32_t n_shards) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_assemble_shards: host-only context (created with device = -1)");
    if (!gathered || n_shards < 1 || stride_bytes % 256) return fail(ctx, "rt_assemble_shards: bad arguments");
    HIPCK(hipSetDevice(ctx->cfg.device));
    uint32_t stride = (uint32_t)(stride_bytes / 4);
    uint64_t total = (uint64_t)stride * (uint64_t)n_shards;
    hipLaunchKernelGGL(k_assemble, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t *)gathered,
                       stride, n_shards, ctx->g, ctx->d_rgba);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(ctx->stream));
    ctx->traced = true;
    return 0;
}

int rt_get_stats(rt_ctx *ctx, rt_stats *out) {
    if (!ctx) return -1;
    if (!out) return fail(ctx, "rt_get_stats: null argument");
    *out = ctx->st;
    return 0;
}

}  // extern "C"
