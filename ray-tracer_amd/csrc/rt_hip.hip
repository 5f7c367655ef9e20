// C ABI (include/rt_hip.h) + the HIP kernels behind it. gfx950 only.
//
// Host side of the library: context, scene upload, per-frame orchestration of the kernels in rt_kernels.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <array>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"
#include "kd_build.h"
#include "kd_build_gpu.h"
#include "rt_device.h"
#include "rt_kernels.h"

using namespace rt;

namespace {

thread_local std::string g_create_error;

// Device form of a k-d tree (nodes, the two leaf-ordered triangle tables, path table), shared by every context of the process that
// uploads the same tree to the same GPU: a pipeline keeps several contexts with the same scene in flight, and their
// walks run side by side - one copy keeps the L2 caches for one tree instead of one per context. Immutable once
// registered; the last context to let go frees it.
struct SharedTree {
    int device = 0;
    unsigned long long hash = 0, sum = 0;        // two independent digests of the contents (see tree_digest)
    size_t n_nodes = 0, n_slots = 0, n_leaves = 0;
    size_t cap_nodes = 0, cap_slots = 0, cap_leaves = 0;
    KdNodeDev *nodes = nullptr;
    TriGeo *lgeo = nullptr;
    TriTag *ltag = nullptr;
    uint32_t *path_slot = nullptr;
    float *leaf_box = nullptr;                   // float32 cell of every leaf with faces, indexed by TriTag::leaf
    int refs = 0;
    bool findable = true;                        // hash / sum describe the contents: other contexts may take this copy for theirs
};
std::mutex g_shared_mu;
std::vector<SharedTree *> g_shared;

// Two digests over the same bytes, eight at a step (these are megabytes). A multiplication only carries a difference
// upwards, so the product's high half is folded back into the low half after every word: a flipped bit anywhere reaches
// every bit of the state. The second digest is a rotating sum, independent of the first: two trees are taken for the
// same one only when both digests and all three sizes agree.
struct Digest { unsigned long long h = 1469598103934665603ull, s = 0x9e3779b97f4a7c15ull; };
void tree_digest(const void *data, size_t n, Digest &d) {
    const unsigned long long *p = (const unsigned long long *)data;
    for (size_t i = 0; i < n / 8; i++) {
        d.h = (d.h ^ p[i]) * 1099511628211ull;
        d.h ^= d.h >> 32;
        d.s = ((d.s << 7) | (d.s >> 57)) + p[i] * 0xff51afd7ed558ccdull;
    }
    const unsigned char *q = (const unsigned char *)data + (n / 8) * 8;
    for (size_t i = 0; i < n % 8; i++) {
        d.h = (d.h ^ q[i]) * 1099511628211ull;
        d.h ^= d.h >> 32;
        d.s = ((d.s << 7) | (d.s >> 57)) + q[i];
    }
}

// Pipeline step k of a frame: k_gen_prims(level k), k_kd_walk (closest-hit rays of level k together with the shadow
// rays of level k-1, triangle tests included), k_shade_b(level k-1), k_shade_a(level k). A frame of depth D takes D+1
// steps; the last one only finishes the shadow rays of the last level.
constexpr int N_STAGES = 4;
constexpr int MAX_STEPS = MAX_LEVELS + 1;

// ------------------------------------------------------------------------------------------------
struct Level {
    void *mem = nullptr;
    uint32_t cap = 0;
    LevelBuf b{};
};

constexpr int MAX_PARTS = 8;

// One independent slice of the shard's tiles. Every part has its own stream, ray-tree levels, queues, record
// buffer and counters, so the parts of a frame run concurrently: while the last long rays of one part's walk
// drain (a persistent kernel ends only when its slowest ray does), the other parts' kernels fill the machine.
struct Part {
    hipStream_t stream = nullptr;
    hipEvent_t ev[MAX_STEPS][N_STAGES + 1]{};    // stage boundaries of every step
    hipEvent_t ev_begin = nullptr, ev_comp = nullptr, ev_end = nullptr, ev_counts = nullptr;
    uint32_t want[MAX_LEVELS]{};                 // capacity the next frame gives level l (0 = not known yet)
    Level lv[MAX_LEVELS];
    LevelCounters *d_counts = nullptr;           // one block of counters per level
    LevelCounters *h_counts = nullptr;           // pinned host copy
    uint4 *d_kdq = nullptr;                      // closest-hit traversal queue, reused by every level
    size_t cap_kdq = 0;
    void *d_shq_mem[2] = {nullptr, nullptr};     // shadow traversal queues: level l uses queue l & 1 (the queue of
    ShadowQ shq[2]{};                            // level l-1 is still being walked while level l fills its own)
    uint32_t cap_shq[2] = {0, 0};
    uint64_t rec_total[2] = {0, 0};              // leaf records tested this frame by the closest-hit / shadow walks (statistics)
    uint32_t n_rays[MAX_LEVELS]{};
    FrameGeom g{};
    uint32_t n0 = 0;                             // level-0 entries = tiles of this part * 64
    int levels_run = 0;
};

}  // namespace

struct rt_ctx {
    rt_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;                // scene uploads, tone, present (= the first part's stream)
    hipEvent_t ev_misc[2]{};                     // tone begin / end
    int n_cus = 256;
    int walk_waves_per_cu = RT_WALK_WAVES_PER_CU;
    uint32_t first_level_rays = 0;               // capacity of the deeper levels on the first frame (0 = level 0's), rt_set_option
    bool stage_timing = true;                    // events around every stage kernel (rt_stats.level_stage_ms)
    bool prune = true;                           // the closest-hit walk skips subtrees that cannot hold the nearest hit (rt_set_option "prune")
    bool stage_timed = false;                    // the frame in flight was queued with them
    int n_parts = 1;
    Part parts[MAX_PARTS];
    // scene (host copies)
    std::vector<PrimDev> prims;
    std::vector<rt_material> mats_host;
    std::vector<MatDev> mats;
    std::vector<LightDev> lights;
    std::vector<rt_prim> prims_host;
    std::vector<double> verts;
    std::vector<uint32_t> faces, face_mat;
    std::vector<TriDev> tris_host;
    KdTreeHost kd;
    GpuKdBuilder *gpu_kd = nullptr;              // device-side builder and its tree (rt_build_kdtree_device)
    bool kd_on_device = false;                   // the current tree is gpu_kd->tree, not `kd`
    uint64_t kd_device_fallbacks = 0;            // device builds that stalled and were redone on the host
    bool kd_built = false, scene_dirty = true, mesh_dirty = true;
    SceneDev sh{};   // host mirror of the device scene header, as the frame in flight (and a repeat of it) uses it
    struct { float ambient[4], bg[4], cam[4], r00[4], r01[4], r10[4], r11[4]; } next{};   // rt_set_globals / rt_set_camera: taken over by the next rt_trace_begin
    // device
    SceneDev *d_scene = nullptr;
    PrimDev *d_prims = nullptr;
    MatDev *d_mats = nullptr;
    LightDev *d_lights = nullptr;
    TriDev *d_tris = nullptr;
    KdNodeDev *d_nodes = nullptr;
    TriGeo *d_lgeo = nullptr;                    // leaf-ordered triangles: geometry for the tests ...
    TriTag *d_ltag = nullptr;                    // ... and material, face, leaf code for the winners
    const float *d_leaf_box = nullptr;           // float32 cells of the leaves, indexed by TriTag::leaf
    size_t n_slots = 0;
    // rt_update_vertices: float32 vertices and vertex indices on the device, a private copy of the leaf-ordered geometry
    // (the shared one belongs to every context that holds the tree), six overhang maxima
    float *d_v32 = nullptr;
    uint32_t *d_faces_u32 = nullptr;
    TriGeo *d_lgeo_own = nullptr;
    uint32_t *d_over_bits = nullptr;
    size_t cap_v32 = 0, cap_faces_u32 = 0, cap_lgeo_own = 0, cap_over_bits = 0;
    bool faces_on_device = false;                // d_faces_u32 holds the current mesh's indices
    bool tris_stale = true;                      // tris_host does not follow `verts` / the material table
    uint2 *d_path_first = nullptr;               // staging of (path code, first slot) pairs
    uint32_t *d_path_slot = nullptr;             // 2^PATH_BITS entries, SceneDev::path_slot
    float over[6] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY};   // SceneDev::over_lo / over_hi of the host-built tree
    bool over_valid = false;
    SharedTree *shared = nullptr;                // d_nodes, d_lgeo, d_ltag and d_path_slot belong to this shared copy, not to the context
    bool tree_dirty = true;
    size_t cap_prims = 0, cap_mats = 0, cap_lights = 0, cap_tris = 0, cap_path_first = 0;
    // frame
    FrameGeom g{};                               // the whole shard
    uint32_t n0 = 0;                             // level-0 entries = tiles of this shard * 64
    float *d_fb = nullptr;
    uint32_t *d_rgba = nullptr, *d_compact = nullptr, *d_canvas = nullptr;
    uint32_t *d_compact_canvas = nullptr;        // rt_present_shard: the shard's tiles after the present pass
    int32_t *d_hit = nullptr;
    double *d_part_sum = nullptr, *d_tone_out = nullptr;
    double *d_tone_pair = nullptr;               // {sum of log luminance, pixel count, 1.0 if the trace overflowed a buffer} of this shard, for a device-side all-reduce
    float *d_tone = nullptr;                     // Lbar, sf as the present pass reads them
    double *d_tone_d = nullptr;                  // the same in double precision
    bool tone_on_device = false;                 // Lbar / sf were computed by rt_tone_finish_device: the host copy is stale
    unsigned long long *d_part_cnt = nullptr, *d_tone_cnt = nullptr;
    double Lbar = 0.0, sf = 0.0;
    bool traced = false;
    bool in_flight = false;                      // rt_trace_begin without its rt_trace_end
    bool host_only = false;
    rt_stats st{};
};

namespace {

constexpr int TONE_BLOCKS = 1024;

// Working memory of rt_luminance_sums_noalloc: per-block partial sums and counts of the two-stage reduction and the two
// totals. One copy per GPU, held until the process ends; calls that overlap on one GPU would race on it.
struct ToneScratch {
    double part_sum[TONE_BLOCKS];
    unsigned long long part_cnt[TONE_BLOCKS];
    double out_sum;
    unsigned long long out_cnt;
};
static __device__ __align__(256) unsigned char scratch[sizeof(ToneScratch)];
constexpr int MAX_SCRATCH_DEVICES = 64;
std::atomic<ToneScratch *> g_scratch_ptr[MAX_SCRATCH_DEVICES];   // device address of `scratch` on every GPU that has asked for it (zero-initialised; threads that race to fill an entry store the same address)

// `scratch` names a host-side stand-in; the array's address on the CURRENT device comes from the runtime, once per device
hipError_t scratch_on_current_device(int device, ToneScratch **out) {
    if (device < 0 || device >= MAX_SCRATCH_DEVICES) return hipErrorInvalidDevice;
    ToneScratch *sp = g_scratch_ptr[device].load(std::memory_order_acquire);
    if (!sp) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        sp = (ToneScratch *)p;
        g_scratch_ptr[device].store(sp, std::memory_order_release);
    }
    *out = sp;
    return hipSuccess;
}

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

int ensure_level(rt_ctx *ctx, Part &P, int l, uint32_t cap) {
    Level &L = P.lv[l];
    if (L.mem && L.cap >= cap) return 0;
    if (L.mem) HIPCK(hipFree(L.mem));
    L.mem = nullptr;
    cap = std::max<uint32_t>((cap + 255u) & ~255u, 256u);
    HIPCK(hipMalloc(&L.mem, (size_t)cap * LEVEL_ARRAYS * sizeof(float)));
    L.cap = cap;
    L.b.ray = (float4 *)L.mem;   // widest alignment first; cap is a multiple of 256
    L.b.best = (unsigned long long *)((float *)L.mem + 8 * (size_t)cap);
    float *p = (float *)L.mem + 10 * (size_t)cap;
    float **f[] = {&L.b.ht, &L.b.cr, &L.b.cg, &L.b.cb, &L.b.kr, &L.b.kt};
    for (auto q : f) { *q = p; p += cap; }
    L.b.hp = (uint32_t *)p; p += cap;
    L.b.hf = (int32_t *)p; p += cap;
    L.b.shadow = (uint32_t *)p; p += cap;
    L.b.refl = (int32_t *)p; p += cap;
    L.b.refr = (int32_t *)p;
    return 0;
}

// a shadow-queue entry names its ray as (ray index << 5 | light): 27 bits of ray index
constexpr uint32_t MAX_LEVEL_RAYS = 1u << 27;

// traversal queues are shared by all levels: sized for the largest level seen so far
// lets go of a shared tree copy; the context's pointers are cleared, so that nothing of it is freed or resized through them
void detach_shared_tree(rt_ctx *ctx) {
    if (!ctx->shared) return;
    std::lock_guard<std::mutex> lock(g_shared_mu);
    SharedTree *t = ctx->shared;
    if (--t->refs == 0) {
        (void)hipFree(t->nodes);
        (void)hipFree(t->lgeo);
        (void)hipFree(t->ltag);
        (void)hipFree(t->path_slot);
        (void)hipFree(t->leaf_box);
        g_shared.erase(std::remove(g_shared.begin(), g_shared.end(), t), g_shared.end());
        delete t;
    }
    ctx->shared = nullptr;
    ctx->d_nodes = nullptr; ctx->d_lgeo = nullptr; ctx->d_ltag = nullptr; ctx->d_path_slot = nullptr; ctx->d_leaf_box = nullptr;
}

int ensure_queues(rt_ctx *ctx, Part &P, int level, uint32_t n_rays, uint32_t n_lights) {
    if (n_rays > MAX_LEVEL_RAYS) return fail(ctx, "rt_trace: a ray-tree level of more than 2^27 rays per part (raise rt_config.parts or shard the frame)");
    if (ensure(ctx, P.d_kdq, P.cap_kdq, n_rays)) return -1;
    const int q = level & 1;
    uint64_t want = (uint64_t)n_rays * std::max<uint32_t>(n_lights, 1);
    if (want > 0x7fffff00ull) return fail(ctx, "shadow queue too large");
    if (P.d_shq_mem[q] && P.cap_shq[q] >= want) return 0;
    if (P.d_shq_mem[q]) HIPCK(hipFree(P.d_shq_mem[q]));
    P.d_shq_mem[q] = nullptr;
    uint32_t cap = std::max<uint32_t>(((uint32_t)want + 255u) & ~255u, 256u);
    HIPCK(hipMalloc(&P.d_shq_mem[q], (size_t)cap * SHADOWQ_ARRAYS * sizeof(float)));
    P.cap_shq[q] = cap;
    P.shq[q].ray = (float4 *)P.d_shq_mem[q];
    P.shq[q].id = (uint32_t *)((float *)P.d_shq_mem[q] + 8 * (size_t)cap);
    return 0;
}

// builds the tree of the current mesh on the device (kd_build_gpu.h); per-face material slots are resolved here
int device_build(rt_ctx *ctx) {
    auto find_mat = [&](int type, int id) -> int32_t {
        for (size_t i = 0; i < ctx->mats_host.size(); i++)
            if (ctx->mats_host[i].type == type && ctx->mats_host[i].id == id) return (int32_t)i;
        return -1;
    };
    const size_t nf = ctx->faces.size() / 3;
    std::vector<int32_t> slots(nf);
    for (size_t f = 0; f < nf; f++) slots[f] = find_mat((int)ctx->face_mat[2 * f], (int)ctx->face_mat[2 * f + 1]);
    if (!ctx->gpu_kd) ctx->gpu_kd = new GpuKdBuilder();
    auto t0 = std::chrono::steady_clock::now();
    std::string err;
    if (ctx->gpu_kd->build(ctx->stream, ctx->verts.data(), (int64_t)ctx->verts.size() / 3, ctx->faces.data(), slots.data(),
                           (int64_t)nf, ctx->cfg.max_kd_depth, (double)ctx->cfg.infinity, err)) {
        ctx->err = "rt_build_kdtree_device: " + err;
        return -1;
    }
    if (ctx->gpu_kd->tree.n_slots > (1u << LEAF_COUNT_SHIFT)) return fail(ctx, "rt_build_kdtree_device: k-d tree has too many leaf slots for the device layout (2^24)");
    detach_shared_tree(ctx);   // an uploaded host-built tree is no longer this context's
    ctx->st.kd_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->st.kd_upload_ms = 0.0;
    return 0;
}

// SceneDev::over_lo / over_hi of a host-built tree: the cells are the float32 boxes the walk's planes cut out of the
// float32 root box, the vertices the float32 ones the triangle tests read.
// triangles in face order on the host: float32 narrowing of the vertices (js/8-kd_tree.js:2462-2464), edges formed in
// float32; per-face material slots resolved against the material table
void refresh_tris_host(rt_ctx *ctx) {
    if (!ctx->tris_stale) return;
    auto find_mat = [&](int type, int id) -> int32_t {
        for (size_t i = 0; i < ctx->mats_host.size(); i++)
            if (ctx->mats_host[i].type == type && ctx->mats_host[i].id == id) return (int32_t)i;
        return -1;
    };
    const size_t nf = ctx->faces.size() / 3;
    ctx->tris_host.resize(nf);
    for (size_t f = 0; f < nf; f++) {
        float v[3][3];
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) v[j][k] = (float)ctx->verts[3 * (size_t)ctx->faces[3 * f + j] + k];
        TriDev &t = ctx->tris_host[f];
        for (int k = 0; k < 3; k++) { t.v0[k] = v[0][k]; t.e1[k] = v[1][k] - v[0][k]; t.e2[k] = v[2][k] - v[0][k]; }
        t.mat = find_mat((int)ctx->face_mat[2 * f], (int)ctx->face_mat[2 * f + 1]);
        t.face = (int32_t)f;
        t.pad = 0;
    }
    ctx->tris_stale = false;
}

void tree_overhang(rt_ctx *ctx, const KdTreeHost &kd, std::vector<float> *leaf_boxes = nullptr) {
    // cells first (one pass over the nodes), then the listed faces against them, leaves spread over a few threads
    std::vector<float> own;
    std::vector<float> &boxes = leaf_boxes ? *leaf_boxes : own;
    boxes.assign(6 * kd.leaves.size(), 0.0f);
    struct Item { uint32_t node; float bmin[3], bmax[3]; };
    std::vector<Item> todo;
    todo.reserve(64);
    if (!kd.nodes.empty())
        todo.push_back(Item{0u, {(float)kd.root_box[0], (float)kd.root_box[1], (float)kd.root_box[2]},
                            {(float)kd.root_box[3], (float)kd.root_box[4], (float)kd.root_box[5]}});
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        const KdNodeDev &n = kd.nodes[it.node];
        const uint32_t ty = n.info >> 30;
        if (ty == 1u) {
            const uint32_t a = (n.info >> 28) & 3u, child = n.info & 0x0fffffffu;
            Item c0 = it, c1 = it;
            c0.node = child; c0.bmax[a] = n.split;
            c1.node = child + 1; c1.bmin[a] = n.split;
            todo.push_back(c0);
            todo.push_back(c1);
        } else if (ty == 3u) {
            float *b = &boxes[6 * (size_t)(n.info & 0x3fffffffu)];
            for (int c = 0; c < 3; c++) { b[c] = it.bmin[c]; b[3 + c] = it.bmax[c]; }
        }
    }
    std::vector<float> v32(ctx->verts.size());
    for (size_t i = 0; i < v32.size(); i++) v32[i] = (float)ctx->verts[i];
    const size_t n_leaves = kd.leaves.size();
    const int nt = (int)std::max<size_t>(1, std::min<size_t>({8, (size_t)std::max(1u, std::thread::hardware_concurrency()), n_leaves / 4096 + 1}));
    std::vector<std::array<float, 6>> part((size_t)nt, std::array<float, 6>{0, 0, 0, 0, 0, 0});
    auto work = [&](int t) {
        std::array<float, 6> m{0, 0, 0, 0, 0, 0};
        for (size_t li = n_leaves * (size_t)t / nt; li < n_leaves * (size_t)(t + 1) / nt; li++) {
            const float *b = &boxes[6 * li];
            for (int k = 0; k < 4; k++) {
                const int32_t f = kd.leaves[li].face[k];
                if (f < 0) continue;
                for (int j = 0; j < 3; j++) {
                    const float *v = &v32[3 * (size_t)ctx->faces[3 * (size_t)f + j]];
                    for (int c = 0; c < 3; c++) {
                        m[c] = std::max(m[c], b[c] - v[c]);
                        m[3 + c] = std::max(m[3 + c], v[c] - b[3 + c]);
                    }
                }
            }
        }
        part[(size_t)t] = m;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    float raw[6] = {0, 0, 0, 0, 0, 0};
    for (const auto &m : part)
        for (int c = 0; c < 6; c++) raw[c] = std::max(raw[c], m[c]);
    kd_overhang_with_margin(raw, kd.root_box, ctx->over);
    ctx->over_valid = true;
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
        d.pad1 = 0;
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 3; r++) { d.m[3 * c + r] = s.model[4 * c + r]; d.im[3 * c + r] = s.inv_model[4 * c + r]; }
        // bounding sphere round the model's origin: the farthest corner of the object-space box that holds the unit
        // shape (sphere: radius 1; cube: half side 0.5; cylinder: radius 1, half height 0.5), 1 % added, rounded up
        d.bound_r2 = 0.0f;
        const double ext[5][3] = {{0, 0, 0}, {0, 0, 0}, {1, 1, 1}, {0.5, 0.5, 0.5}, {1, 0.5, 1}};
        if (s.kind >= RT_PRIM_SPHERE && s.kind <= RT_PRIM_CYLINDER) {
            double r2 = 0.0;
            for (int corner = 0; corner < 8; corner++) {
                double w[3] = {0, 0, 0};
                for (int c = 0; c < 3; c++) {
                    const double e = ((corner >> c) & 1 ? 1.0 : -1.0) * ext[s.kind][c];
                    for (int r = 0; r < 3; r++) w[r] += (double)s.model[4 * c + r] * e;
                }
                r2 = std::max(r2, w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            }
            if (s.kind == RT_PRIM_SPHERE) {
                // the unit sphere itself: its image lies within the largest singular value of the linear part, whose
                // square is at most the largest absolute row sum of the Gram matrix (exact for a rotated uniform scale)
                double g = 0.0;
                for (int i = 0; i < 3; i++) {
                    double row = 0.0;
                    for (int j = 0; j < 3; j++) {
                        double e = 0.0;
                        for (int r = 0; r < 3; r++) e += (double)s.model[4 * i + r] * (double)s.model[4 * j + r];
                        row += std::fabs(e);
                    }
                    g = std::max(g, row);
                }
                r2 = std::min(r2, g);
            }
            const double b = r2 * 1.0201 * 1.0001;
            d.bound_r2 = (std::isfinite(b) && b < 1e30) ? (float)b : 0.0f;
        }
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

    if (ctx->mesh_dirty && ctx->kd_on_device) {
        // the material table changed under a device-built tree: its triangles carry material slots
        if (device_build(ctx)) return -1;
        ctx->mesh_dirty = false;
    }
    if (ctx->mesh_dirty) {
        ctx->tris_stale = true;
        refresh_tris_host(ctx);
        const size_t nf = ctx->tris_host.size();
        if (ensure(ctx, ctx->d_tris, ctx->cap_tris, nf)) return -1;
        if (nf) HIPCK(hipMemcpyAsync(ctx->d_tris, ctx->tris_host.data(), nf * sizeof(TriDev), hipMemcpyHostToDevice, ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));
        ctx->mesh_dirty = false;
        ctx->tree_dirty = true;
    }
    if (ctx->tree_dirty && ctx->kd_built && !ctx->kd_on_device) {
        // device form of the tree: leaves name a slot range of the leaf-ordered triangle array
        auto t0 = std::chrono::steady_clock::now();
        const KdTreeHost &kd = ctx->kd;
        refresh_tris_host(ctx);
        std::vector<uint32_t> first(kd.leaves.size() + 1, 0);
        for (size_t i = 0; i < kd.leaves.size(); i++) {
            uint32_t c = 0;
            for (int k = 0; k < 4; k++) c += kd.leaves[i].face[k] > -1;
            first[i + 1] = first[i] + c;
        }
        if (first.back() > (1u << LEAF_COUNT_SHIFT)) return fail(ctx, "k-d tree has too many leaf slots for the device layout (2^24)");
        std::vector<TriGeo> lgeo(first.back());
        std::vector<TriTag> ltag(first.back());
        for (size_t i = 0; i < kd.leaves.size(); i++) {
            uint32_t s0 = first[i];
            for (int k = 0; k < 4; k++)
                if (kd.leaves[i].face[k] > -1) {
                    const TriDev &t = ctx->tris_host[(size_t)kd.leaves[i].face[k]];
                    memcpy(lgeo[s0].v0, t.v0, sizeof(TriGeo));   // v0, e1, e2 lie in this order in both records
                    ltag[s0] = TriTag{t.mat, t.face, 0u, (uint32_t)i};
                    s0++;
                }
        }
        std::vector<KdNodeDev> nodes(kd.nodes);
        std::vector<uint2> path_first;   // (path code, first slot) of every non-empty leaf
        path_first.reserve(kd.leaves.size());
        for (size_t s = 0; s < nodes.size(); s++) {
            KdNodeDev &n = nodes[s];
            if ((n.info >> 30) == 3u) {
                const uint32_t li = n.info & 0x3fffffffu;
                n.info = leaf_info(first[li + 1] - first[li], first[li]);
                const uint32_t code = path_code(kd.heap_index[s]);
                for (uint32_t k = first[li]; k < first[li + 1]; k++) ltag[k].code = code;
                path_first.push_back(make_uint2(code, first[li]));
            }
        }
        std::vector<float> leaf_box;
        tree_overhang(ctx, kd, &leaf_box);
        const size_t n_leaves = kd.leaves.size();
        SharedTree *mine = nullptr;   // the copy this context may overwrite: it is its only holder and the new tree fits
        {
            std::lock_guard<std::mutex> lock(g_shared_mu);
            SharedTree *old = ctx->shared;
            if (old && old->refs == 1 && old->device == ctx->cfg.device && old->cap_nodes >= nodes.size() &&
                old->cap_slots >= lgeo.size() && old->cap_leaves >= n_leaves) {
                mine = old;
                mine->findable = false;   // its contents change now; a tree that is rebuilt over and over is not offered for sharing
            }
        }
        // The digests cost a pass over some fifteen megabytes; a context that rebuilds its own tree (moving geometry)
        // neither looks for a copy to share nor offers its own
        Digest dg;
        if (!mine) {
            tree_digest(nodes.data(), nodes.size() * sizeof(KdNodeDev), dg);
            tree_digest(lgeo.data(), lgeo.size() * sizeof(TriGeo), dg);
            tree_digest(ltag.data(), ltag.size() * sizeof(TriTag), dg);
        }
        if (!mine) detach_shared_tree(ctx);
        if (!mine) {   // another context of this process may hold this very tree on this GPU already
            std::lock_guard<std::mutex> lock(g_shared_mu);
            for (SharedTree *t : g_shared)
                if (t->findable && t->device == ctx->cfg.device && t->hash == dg.h && t->sum == dg.s && t->n_nodes == nodes.size() &&
                    t->n_slots == lgeo.size() && t->n_leaves == n_leaves) {
                    t->refs++;
                    ctx->shared = t;
                    break;
                }
        }
        if (mine || !ctx->shared) {
            SharedTree *t = mine;
            auto drop = [&]() {
                if (mine) { detach_shared_tree(ctx); return; }   // (the only holder lets go: the buffers are freed there)
                (void)hipFree(t->nodes); (void)hipFree(t->lgeo); (void)hipFree(t->ltag); (void)hipFree(t->path_slot); (void)hipFree(t->leaf_box);
                delete t;
            };
            hipError_t e = hipSuccess;
            if (!t) {
                // a tree that is rebuilt again and again (moving geometry) keeps its buffers: a quarter of room to grow into
                t = new SharedTree();
                t->device = ctx->cfg.device;
                t->cap_nodes = std::max<size_t>(nodes.size() + nodes.size() / 4, 1);
                t->cap_slots = std::max<size_t>(lgeo.size() + lgeo.size() / 4, 1);
                t->cap_leaves = std::max<size_t>(n_leaves + n_leaves / 4, 1);
                if ((e = hipMalloc((void **)&t->nodes, t->cap_nodes * sizeof(KdNodeDev))) != hipSuccess ||
                    (e = hipMalloc((void **)&t->lgeo, t->cap_slots * sizeof(TriGeo))) != hipSuccess ||
                    (e = hipMalloc((void **)&t->ltag, t->cap_slots * sizeof(TriTag))) != hipSuccess ||
                    (e = hipMalloc((void **)&t->leaf_box, t->cap_leaves * 6 * sizeof(float))) != hipSuccess ||
                    (e = hipMalloc((void **)&t->path_slot, sizeof(uint32_t) << PATH_BITS)) != hipSuccess) {
                    drop();
                    return fail(ctx, (std::string("hipMalloc (k-d tree): ") + hipGetErrorString(e)).c_str());
                }
            }
            t->hash = dg.h; t->sum = dg.s; t->n_nodes = nodes.size(); t->n_slots = lgeo.size(); t->n_leaves = n_leaves;
            if (ensure(ctx, ctx->d_path_first, ctx->cap_path_first, path_first.size())) { drop(); return -1; }
            e = hipMemcpyAsync(t->nodes, nodes.data(), nodes.size() * sizeof(KdNodeDev), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess && !lgeo.empty())
                e = hipMemcpyAsync(t->lgeo, lgeo.data(), lgeo.size() * sizeof(TriGeo), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess && !ltag.empty())
                e = hipMemcpyAsync(t->ltag, ltag.data(), ltag.size() * sizeof(TriTag), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess && !leaf_box.empty())
                e = hipMemcpyAsync(t->leaf_box, leaf_box.data(), leaf_box.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess && !path_first.empty()) {
                // only the entries of existing leaves are ever read, so the table is not cleared
                e = hipMemcpyAsync(ctx->d_path_first, path_first.data(), path_first.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(k_fill_path_slots, dim3((unsigned)((path_first.size() + 255) / 256)), dim3(256), 0, ctx->stream,
                                       ctx->d_path_first, (uint32_t)path_first.size(), t->path_slot);
                    e = hipGetLastError();
                }
            }
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // the sources are locals; the copy is complete before anyone shares it
            if (e != hipSuccess) { drop(); return fail(ctx, (std::string("k-d tree upload: ") + hipGetErrorString(e)).c_str()); }
            if (!mine) {
                t->refs = 1;
                std::lock_guard<std::mutex> lock(g_shared_mu);
                g_shared.push_back(t);
                ctx->shared = t;
            }
        }
        ctx->d_nodes = ctx->shared->nodes; ctx->d_lgeo = ctx->shared->lgeo; ctx->d_ltag = ctx->shared->ltag; ctx->d_path_slot = ctx->shared->path_slot;
        ctx->d_leaf_box = ctx->shared->leaf_box;
        ctx->n_slots = lgeo.size();
        HIPCK(hipStreamSynchronize(ctx->stream));
        ctx->st.kd_upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ctx->tree_dirty = false;
    }

    SceneDev &h = ctx->sh;
    const rt_config &c = ctx->cfg;
    memcpy(h.ambient, ctx->next.ambient, sizeof h.ambient); memcpy(h.bg, ctx->next.bg, sizeof h.bg);
    memcpy(h.cam, ctx->next.cam, sizeof h.cam); memcpy(h.r00, ctx->next.r00, sizeof h.r00); memcpy(h.r01, ctx->next.r01, sizeof h.r01);
    memcpy(h.r10, ctx->next.r10, sizeof h.r10); memcpy(h.r11, ctx->next.r11, sizeof h.r11);
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
        const double *rb = ctx->kd_on_device ? ctx->gpu_kd->tree.root_box : ctx->kd.root_box;
        for (int k = 0; k < 3; k++) {   // uniform3fv narrowing of root.min / root.max (js/8-kd_tree.js:1114, :2519-2520)
            h.box_lo[k] = (float)rb[k];
            h.box_hi[k] = (float)rb[3 + k];
        }
    }
    for (int k = 0; k < 3; k++) {
        const float *ov = ctx->kd_on_device ? ctx->gpu_kd->tree.over : ctx->over;
        h.over_lo[k] = ctx->prune ? ov[k] : INFINITY;
        h.over_hi[k] = ctx->prune ? ov[3 + k] : INFINITY;
    }
    h.prims = ctx->d_prims; h.mats = ctx->d_mats; h.lights = ctx->d_lights; h.tris = ctx->d_tris;
    h.nodes = ctx->d_nodes; h.lgeo = ctx->d_lgeo; h.ltag = ctx->d_ltag; h.path_slot = ctx->d_path_slot;
    if (ctx->kd_on_device) {
        const GpuKdTree &t = ctx->gpu_kd->tree;
        h.tris = t.tris; h.nodes = t.nodes; h.lgeo = t.lgeo; h.ltag = t.ltag; h.path_slot = t.path_slot;
    }
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
    {   // tile numbers and ray indices are 32-bit: the supersampled render target may have at most 2^30 pixels
        const uint64_t w = (uint64_t)cfg->width * (uint64_t)cfg->supersampling, h = (uint64_t)cfg->height * (uint64_t)cfg->supersampling;
        const uint64_t blocks = ((w + 127) / 128) * ((h + 127) / 128);
        if (w > (1u << 20) || h > (1u << 20) || blocks * 128 * 128 > (1ull << 30)) {
            g_create_error = "rt_create: render target (width x height x supersampling^2) beyond 2^30 pixels"; return -1;
        }
    }
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
    {
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, cfg->device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
        ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    for (auto &ev : ctx->ev_misc)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    {   // set-up for rt_luminance_sums_noalloc on this device, so that its first call there has nothing left to resolve
        ToneScratch *sp = nullptr;
        if ((e = scratch_on_current_device(cfg->device, &sp)) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
    }
    FrameGeom &g = ctx->g;
    g.W = cfg->width * cfg->supersampling;
    g.H = cfg->height * cfg->supersampling;
    g.tiles_x = (g.W + 7) / 8;
    g.tiles_y = (g.H + 7) / 8;
    g.blocks_x = (g.tiles_x + TILE_BLOCK - 1) / TILE_BLOCK;
    g.n_tiles = g.blocks_x * ((g.tiles_y + TILE_BLOCK - 1) / TILE_BLOCK) * TILE_BLOCK * TILE_BLOCK;   // whole blocks (tile_coords)
    g.shard_index = cfg->shard_index;
    g.shard_count = sc;
    g.compact_mul = 1;
    g.compact_add = 0;
    const uint32_t my_tiles = (uint32_t)((g.n_tiles - g.shard_index + sc - 1) / sc);
    ctx->n0 = my_tiles * 64u;
    // parts: cfg.parts, else the environment (RT_PARTS), else 1 (measured: with the current kernels a second part
    // neither helps nor hurts much - the walks are throughput-bound, not tail-bound); never more parts than tiles
    int np = cfg->parts;
    if (np <= 0) {
        const char *env = getenv("RT_PARTS");
        np = env ? atoi(env) : 1;
    }
    np = std::max(1, std::min({np, MAX_PARTS, (int)std::max<uint32_t>(my_tiles, 1u)}));
    ctx->n_parts = np;
    if (const char *env = getenv("RT_WALK_WAVES")) ctx->walk_waves_per_cu = std::max(1, std::min(atoi(env), 32));   // tuning hook
    for (int p = 0; p < np; p++) {
        Part &P = ctx->parts[p];
        if ((e = hipStreamCreate(&P.stream)) != hipSuccess) return bail("hipStreamCreate", e);
        for (auto &row : P.ev)
            for (auto &ev : row)
                if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
        for (hipEvent_t *ev : {&P.ev_begin, &P.ev_comp, &P.ev_end, &P.ev_counts})
            if ((e = hipEventCreate(ev)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipMalloc((void **)&P.d_counts, MAX_LEVELS * sizeof(LevelCounters))) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipMemset(P.d_counts, 0, MAX_LEVELS * sizeof(LevelCounters))) != hipSuccess) return bail("hipMemset", e);
        if ((e = hipHostMalloc((void **)&P.h_counts, MAX_LEVELS * sizeof(LevelCounters))) != hipSuccess) return bail("hipHostMalloc", e);
        // part p takes every np-th tile of the shard: tiles shard_index + shard_count * (p + np * j)
        P.g = g;
        P.g.shard_index = g.shard_index + g.shard_count * p;
        P.g.shard_count = g.shard_count * np;
        P.g.compact_mul = np;
        P.g.compact_add = p;
        const uint32_t part_tiles = my_tiles > (uint32_t)p ? (my_tiles - (uint32_t)p + (uint32_t)np - 1u) / (uint32_t)np : 0u;
        P.n0 = part_tiles * 64u;
        if (P.n0 > MAX_LEVEL_RAYS) {
            g_create_error = "rt_create: more than 2^27 rays per part at this size; raise rt_config.parts or shard the frame";
            rt_destroy(ctx);
            return -1;
        }
    }
    ctx->stream = ctx->parts[0].stream;
    const size_t npx = (size_t)g.W * g.H;
#define ALLOC(p, bytes)                                                                    \
    if ((e = hipMalloc((void **)&(p), (bytes))) != hipSuccess) return bail("hipMalloc", e); \
    if ((e = hipMemset((p), 0, (bytes))) != hipSuccess) return bail("hipMemset", e)
    ALLOC(ctx->d_scene, sizeof(SceneDev));
    ALLOC(ctx->d_fb, npx * 3 * sizeof(float));
    ALLOC(ctx->d_rgba, npx * 4);
    ALLOC(ctx->d_hit, npx * 4);
    if ((e = hipMemset(ctx->d_hit, 0xff, npx * 4)) != hipSuccess) return bail("hipMemset", e);   // -1: pixels of other shards are never written
    ALLOC(ctx->d_compact, std::max<size_t>(ctx->n0, 64) * 4);
    ALLOC(ctx->d_compact_canvas, std::max<size_t>(ctx->n0, 64) * 4);
    ALLOC(ctx->d_canvas, (size_t)cfg->width * cfg->height * 4);
    ALLOC(ctx->d_part_sum, TONE_BLOCKS * sizeof(double));
    ALLOC(ctx->d_part_cnt, TONE_BLOCKS * sizeof(unsigned long long));
    ALLOC(ctx->d_tone_out, sizeof(double));
    ALLOC(ctx->d_tone_cnt, sizeof(unsigned long long));
    ALLOC(ctx->d_tone_pair, 3 * sizeof(double));
    ALLOC(ctx->d_tone, 2 * sizeof(float));
    ALLOC(ctx->d_tone_d, 2 * sizeof(double));
#undef ALLOC
    *out = ctx;
    return 0;
}

void rt_destroy(rt_ctx *ctx) {
    if (!ctx) return;
    if (ctx->host_only) { delete ctx; return; }
    (void)hipSetDevice(ctx->cfg.device);
    (void)hipDeviceSynchronize();
    delete ctx->gpu_kd;
    detach_shared_tree(ctx);
    void *ptrs[] = {ctx->d_scene, ctx->d_prims, ctx->d_mats, ctx->d_lights, ctx->d_tris, ctx->d_v32, ctx->d_faces_u32, ctx->d_lgeo_own, ctx->d_over_bits,
                    ctx->d_path_first, ctx->d_fb, ctx->d_rgba, ctx->d_hit, ctx->d_compact, ctx->d_compact_canvas, ctx->d_canvas,
                    ctx->d_part_sum, ctx->d_part_cnt, ctx->d_tone_out, ctx->d_tone_cnt,
                    ctx->d_tone_pair, ctx->d_tone, ctx->d_tone_d};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &ev : ctx->ev_misc)
        if (ev) (void)hipEventDestroy(ev);
    for (Part &P : ctx->parts) {
        for (void *p : {(void *)P.d_counts, (void *)P.d_kdq, P.d_shq_mem[0], P.d_shq_mem[1]})
            if (p) (void)hipFree(p);
        if (P.h_counts) (void)hipHostFree(P.h_counts);
        for (auto &l : P.lv)
            if (l.mem) (void)hipFree(l.mem);
        for (auto &row : P.ev)
            for (auto &ev : row)
                if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : {P.ev_begin, P.ev_comp, P.ev_end, P.ev_counts})
            if (ev) (void)hipEventDestroy(ev);
        if (P.stream) (void)hipStreamDestroy(P.stream);
    }
    delete ctx;
}

int rt_set_primitives(rt_ctx *ctx, const rt_prim *prims, int32_t n) {
    if (!ctx) return -1;
    if (n < 0 || (n > 0 && !prims)) return fail(ctx, "rt_set_primitives: bad arguments");
    if (n > 60000) return fail(ctx, "rt_set_primitives: at most 60000 primitives");
    for (int i = 0; i < n; i++)
        if (prims[i].kind < RT_PRIM_PLANE || prims[i].kind > RT_PRIM_CYLINDER) return fail(ctx, "rt_set_primitives: unknown primitive kind");
    // callers that mirror RayTracer.update set every table on every frame: an unchanged table costs nothing
    if ((size_t)n == ctx->prims_host.size() && (n == 0 || memcmp(prims, ctx->prims_host.data(), (size_t)n * sizeof(rt_prim)) == 0)) return 0;
    ctx->prims_host.assign(prims, prims + n);
    ctx->scene_dirty = true;
    return 0;
}

int rt_set_materials(rt_ctx *ctx, const rt_material *mats, int32_t n) {
    if (!ctx) return -1;
    if (n < 0 || (n > 0 && !mats)) return fail(ctx, "rt_set_materials: bad arguments");
    for (int i = 0; i < n; i++)
        if (mats[i].type < RT_MAT_PHONG || mats[i].type > RT_MAT_PHONG_CHECKERED) return fail(ctx, "rt_set_materials: unknown material type");
    if ((size_t)n == ctx->mats_host.size() && (n == 0 || memcmp(mats, ctx->mats_host.data(), (size_t)n * sizeof(rt_material)) == 0)) return 0;
    ctx->mats_host.assign(mats, mats + n);
    ctx->scene_dirty = true;
    ctx->mesh_dirty = true;   // per-face material slots are resolved against this table
    return 0;
}

int rt_set_lights(rt_ctx *ctx, const rt_light *lights, int32_t n, int32_t n_loop) {
    if (!ctx) return -1;
    if (n < 0 || (n > 0 && !lights) || n_loop < n) return fail(ctx, "rt_set_lights: bad arguments (need n >= 0, n_loop >= n)");
    if (n_loop > MAX_LIGHTS) return fail(ctx, "rt_set_lights: at most 32 lights");
    ctx->lights.assign((size_t)n_loop, LightDev{{0, 0, 0, 0}, {0, 0, 0, 0}});
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { ctx->lights[i].pos[k] = lights[i].position[k]; ctx->lights[i].col[k] = lights[i].color[k]; }
    ctx->scene_dirty = true;
    return 0;
}

int rt_set_globals(rt_ctx *ctx, const float ambient[3], const float bg[3]) {
    if (!ctx) return -1;
    if (!ambient || !bg) return fail(ctx, "rt_set_globals: null argument");
    for (int k = 0; k < 3; k++) { ctx->next.ambient[k] = ambient[k]; ctx->next.bg[k] = bg[k]; }
    ctx->scene_dirty = true;
    return 0;
}

int rt_set_camera(rt_ctx *ctx, const float pos[3], const float r00[3], const float r01[3], const float r10[3],
                  const float r11[3]) {
    if (!ctx) return -1;
    if (!pos || !r00 || !r01 || !r10 || !r11) return fail(ctx, "rt_set_camera: null argument");
    for (int k = 0; k < 3; k++) {
        ctx->next.cam[k] = pos[k]; ctx->next.r00[k] = r00[k]; ctx->next.r01[k] = r01[k]; ctx->next.r10[k] = r10[k]; ctx->next.r11[k] = r11[k];
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
    ctx->kd_on_device = false;
    ctx->mesh_dirty = true;
    ctx->scene_dirty = true;
    ctx->faces_on_device = false;
    ctx->over_valid = false;
    return 0;
}

int rt_update_vertices(rt_ctx *ctx, const double *verts, int32_t n_verts) {
    if (!ctx) return -1;
    if (n_verts < 0 || (n_verts > 0 && !verts)) return fail(ctx, "rt_update_vertices: bad arguments");
    if ((size_t)n_verts * 3 != ctx->verts.size()) return fail(ctx, "rt_update_vertices: the mesh has another number of vertices (use rt_set_mesh)");
    if (ctx->in_flight) return fail(ctx, "rt_update_vertices: a frame is in flight, call rt_trace_end first");
    ctx->verts.assign(verts, verts + (size_t)n_verts * 3);
    ctx->tris_stale = true;
    ctx->over_valid = false;
    ctx->scene_dirty = true;
    const size_t nf = ctx->faces.size() / 3;
    // nothing of this mesh is on the device yet (or no device at all): the next upload takes the new vertices, through
    // whatever tree the context holds
    if (ctx->host_only || ctx->mesh_dirty || nf == 0) { ctx->mesh_dirty = true; return 0; }
    HIPCK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->stream;
    std::vector<float> v32((size_t)n_verts * 3);
    for (size_t i = 0; i < v32.size(); i++) v32[i] = (float)verts[i];   // vertsData[i*3+0] = g_verts[i].x, js/8-kd_tree.js:2462-2464
    if (ensure(ctx, ctx->d_v32, ctx->cap_v32, v32.size())) return -1;
    if (ensure(ctx, ctx->d_over_bits, ctx->cap_over_bits, 6)) return -1;
    HIPCK(hipMemcpyAsync(ctx->d_v32, v32.data(), v32.size() * sizeof(float), hipMemcpyHostToDevice, s));
    const bool dev_tree = ctx->kd_on_device && ctx->kd_built;
    const uint32_t *d_faces = dev_tree ? ctx->gpu_kd->tree.faces : nullptr;
    if (!d_faces) {
        if (!ctx->faces_on_device) {
            if (ensure(ctx, ctx->d_faces_u32, ctx->cap_faces_u32, ctx->faces.size())) return -1;
            HIPCK(hipMemcpyAsync(ctx->d_faces_u32, ctx->faces.data(), ctx->faces.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            ctx->faces_on_device = true;
        }
        d_faces = ctx->d_faces_u32;
    }
    TriDev *tris = dev_tree ? ctx->gpu_kd->tree.tris : ctx->d_tris;
    hipLaunchKernelGGL(k_refill_tris, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, tris, d_faces, ctx->d_v32, (uint32_t)nf);
    const bool have_tree = ctx->kd_built && !ctx->tree_dirty && (dev_tree || ctx->shared);
    if (have_tree) {
        // the leaf-ordered records through the tree as it stands (the reference's shader reads the new vertices through
        // the old leaves until the next rebuild, js/8-kd_tree.js:2494-2533), and the old tree's overhang for the new vertices
        const size_t n_slots = dev_tree ? ctx->gpu_kd->tree.n_slots : ctx->n_slots;
        TriGeo *lgeo;
        const TriTag *ltag;
        const float *boxes;
        if (dev_tree) {
            lgeo = ctx->gpu_kd->tree.lgeo; ltag = ctx->gpu_kd->tree.ltag; boxes = ctx->gpu_kd->tree.leaf_box;
        } else {
            if (ensure(ctx, ctx->d_lgeo_own, ctx->cap_lgeo_own, n_slots)) return -1;
            lgeo = ctx->d_lgeo_own; ltag = ctx->d_ltag; boxes = ctx->d_leaf_box;
            ctx->d_lgeo = ctx->d_lgeo_own;
        }
        HIPCK(hipMemsetAsync(ctx->d_over_bits, 0, 6 * sizeof(uint32_t), s));
        if (n_slots)
            hipLaunchKernelGGL(k_refill_slots, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, lgeo, ltag, d_faces, ctx->d_v32,
                               (uint32_t)n_slots, boxes, ctx->d_over_bits);
        HIPCK(hipGetLastError());
        float raw[6];
        HIPCK(hipMemcpyAsync(raw, ctx->d_over_bits, sizeof raw, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        if (dev_tree) kd_overhang_with_margin(raw, ctx->gpu_kd->tree.root_box, ctx->gpu_kd->tree.over);
        else { kd_overhang_with_margin(raw, ctx->kd.root_box, ctx->over); ctx->over_valid = true; }
    } else {
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(s));   // v32 is a local
    }
    return 0;
}

int rt_build_kdtree_device(rt_ctx *ctx) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_build_kdtree_device: host-only context (created with device = -1)");
    if (ctx->in_flight) return fail(ctx, "rt_build_kdtree_device: a frame is in flight, call rt_trace_end first");
    // a tree of one node (at most four faces, or no depth to split in) is not worth a launch
    if (ctx->faces.size() / 3 <= 4 || ctx->cfg.max_kd_depth <= 1) return rt_build_kdtree(ctx);
    HIPCK(hipSetDevice(ctx->cfg.device));
    if (device_build(ctx)) {
        // A stalled build (a workgroup's wait for another's count ran out: the scheme leans on the order in which the
        // hardware starts workgroups, which HIP does not promise) is no reason to fail a frame of an animation: the
        // host builder makes the same tree. Anything else is an error. RT_KD_NO_FALLBACK keeps the failure (tests).
        if (!(ctx->gpu_kd && ctx->gpu_kd->stalled) || getenv("RT_KD_NO_FALLBACK")) return -1;
        fprintf(stderr, "rt_build_kdtree_device: %s; building this tree on the host instead\n", ctx->err.c_str());
        ctx->kd_device_fallbacks++;
        ctx->kd_on_device = false;
        ctx->mesh_dirty = true;
        return rt_build_kdtree(ctx);
    }
    ctx->kd_on_device = true;
    ctx->kd_built = true;
    ctx->mesh_dirty = false;
    ctx->tree_dirty = false;
    ctx->scene_dirty = true;
    return 0;
}

int rt_build_kdtree(rt_ctx *ctx) {
    if (!ctx) return -1;
    if (ctx->kd_on_device) { ctx->kd_on_device = false; ctx->mesh_dirty = true; }
    auto t0 = std::chrono::steady_clock::now();
    build_kdtree(ctx->verts.data(), (int64_t)ctx->verts.size() / 3, ctx->faces.data(), (int64_t)ctx->faces.size() / 3,
                 ctx->cfg.max_kd_depth, (double)ctx->cfg.infinity, 0, ctx->kd);
    auto t1 = std::chrono::steady_clock::now();
    if (ctx->host_only) {
        ctx->st.kd_build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        ctx->kd_built = true;
        ctx->over_valid = false;
        return 0;
    }
    ctx->st.kd_build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    ctx->tree_dirty = true;
    ctx->over_valid = false;
    ctx->kd_built = true;
    ctx->scene_dirty = true;
    return 0;
}

int rt_kdtree_counts(rt_ctx *ctx, int64_t *n_nodes, int64_t *n_leaves) {
    if (!ctx) return -1;
    if (!ctx->kd_built) return fail(ctx, "rt_kdtree_counts: no tree built");
    if (ctx->kd_on_device) {
        if (n_nodes) *n_nodes = ctx->gpu_kd->tree.n_nodes;
        if (n_leaves) *n_leaves = ctx->gpu_kd->tree.n_leaves;
        return 0;
    }
    if (n_nodes) *n_nodes = ctx->kd.n_nodes;
    if (n_leaves) *n_leaves = (int64_t)ctx->kd.leaves.size();
    return 0;
}

int rt_kdtree_overhang(rt_ctx *ctx, float out[6]) {
    if (!ctx) return -1;
    if (!out) return fail(ctx, "rt_kdtree_overhang: null argument");
    if (!ctx->kd_built) return fail(ctx, "rt_kdtree_overhang: no tree built");
    if (!ctx->kd_on_device && !ctx->over_valid) tree_overhang(ctx, ctx->kd);
    memcpy(out, ctx->kd_on_device ? ctx->gpu_kd->tree.over : ctx->over, 6 * sizeof(float));
    return 0;
}

int rt_export_kdtree_reference(rt_ctx *ctx, uint32_t *nodes, int32_t *leaves, double root_box[6]) {
    if (!ctx) return -1;
    if (!ctx->kd_built) return fail(ctx, "rt_export_kdtree_reference: no tree built");
    if (ctx->kd_on_device) {
        const GpuKdTree &t = ctx->gpu_kd->tree;
        HIPCK(hipSetDevice(ctx->cfg.device));
        if (nodes) {   // the device lists the nodes in no particular order: ascending heap index, like the host export
            std::vector<std::pair<uint32_t, uint32_t>> v(t.n_nodes);
            HIPCK(hipMemcpy(v.data(), t.pairs, (size_t)t.n_nodes * sizeof(uint2), hipMemcpyDeviceToHost));
            std::sort(v.begin(), v.end());
            for (size_t i = 0; i < v.size(); i++) { nodes[2 * i] = v[i].first; nodes[2 * i + 1] = v[i].second; }
        }
        if (leaves && t.n_leaves) HIPCK(hipMemcpy(leaves, t.leaves, (size_t)t.n_leaves * 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (root_box) memcpy(root_box, t.root_box, sizeof t.root_box);
        return 0;
    }
    if (nodes) export_reference(ctx->kd, nodes);
    if (leaves && !ctx->kd.leaves.empty()) memcpy(leaves, ctx->kd.leaves.data(), ctx->kd.leaves.size() * sizeof(KdLeafDev));
    if (root_box) memcpy(root_box, ctx->kd.root_box, sizeof ctx->kd.root_box);
    return 0;
}

// queues pipeline step k of one part. Nothing here waits for the device: every kernel reads the number of rays of its
// level from the counters the previous step left in device memory and is launched over the level's capacity.
static int launch_step(rt_ctx *ctx, Part &P, int k) {
    const int D = ctx->cfg.max_depth;
    const SceneDev &sh = ctx->sh;
    const bool walk = sh.render_triangles && sh.use_kd;
    const bool have_cur = k < D;      // level k exists: its closest-hit half runs in this step
    const bool have_prev = k >= 1;    // level k-1 exists: its shadow half runs in this step
    const bool last = (k == D - 1);
    hipStream_t s = P.stream;
    auto level_args = [&](int l) {
        LevelArgs A{};
        A.S = ctx->sh;
        A.cur = P.lv[l].b;
        if (l < D - 1) { A.nxt = P.lv[l + 1].b; A.nxt_cap = P.lv[l + 1].cap; }
        A.shq = P.shq[l & 1];
        A.shq_cap = P.cap_shq[l & 1];
        A.kdq = P.d_kdq;
        A.cnt = P.d_counts + l;
        A.n_src = l > 0 ? &P.d_counts[l - 1].next_count : nullptr;
        A.n_rays = l > 0 ? P.lv[l].cap : P.n0;
        A.hit_ids = ctx->d_hit;
        A.g = P.g;
        A.level = l;
        A.last_level = (l == D - 1) ? 1 : 0;
        return A;
    };
    LevelArgs A{}, B{};
    WalkArgs Wk{};
    Wk.S = ctx->sh;
    if (have_cur) {
        A = level_args(k);
        WalkJob &J = Wk.job[1];
        J.rays = P.lv[k].b.ray;
        J.kdq = P.d_kdq;
        J.best = P.lv[k].b.best;
        J.cnt = P.d_counts + k;
        J.enabled = walk ? 1u : 0u;
    }
    if (have_prev) {
        B = level_args(k - 1);
        WalkJob &J = Wk.job[0];
        J.rays = P.shq[(k - 1) & 1].ray;
        J.shq_id = P.shq[(k - 1) & 1].id;
        J.shadow = P.lv[k - 1].b.shadow;
        J.cnt = P.d_counts + (k - 1);
        J.q_cap = P.cap_shq[(k - 1) & 1];
        J.enabled = walk ? 1u : 0u;
    }
    (void)last;
    const uint32_t bs = shade_block(A.n_rays), bs_prev = shade_block(B.n_rays);
    const dim3 grid((A.n_rays + bs - 1) / bs), grid_prev((B.n_rays + bs_prev - 1) / bs_prev);
    // persistent traversal waves: one 64-lane workgroup each (LDS: short stacks + mailbox)
    const dim3 walk_grid((unsigned)ctx->n_cus * (unsigned)ctx->walk_waves_per_cu);
    const bool any_walk = Wk.job[0].enabled || Wk.job[1].enabled;
    hipEvent_t *ev = P.ev[k];
    if (ctx->stage_timing) HIPCK(hipEventRecord(ev[0], s));
    if (have_cur) hipLaunchKernelGGL(k_gen_prims, grid, dim3(bs), 0, s, A);
    if (ctx->stage_timing) HIPCK(hipEventRecord(ev[1], s));
    if (any_walk) hipLaunchKernelGGL(k_kd_walk, walk_grid, dim3(WAVE), 0, s, Wk);
    if (ctx->stage_timing) HIPCK(hipEventRecord(ev[2], s));
    if (have_prev) hipLaunchKernelGGL(k_shade_b, grid_prev, dim3(bs_prev), 0, s, B);
    if (ctx->stage_timing) HIPCK(hipEventRecord(ev[3], s));
    if (have_cur) hipLaunchKernelGGL(k_shade_a, grid, dim3(bs), 0, s, A);
    HIPCK(hipGetLastError());
    if (ctx->stage_timing) HIPCK(hipEventRecord(ev[4], s));
    return 0;
}

// Queues one whole frame of a part: D+1 pipeline steps, the bottom-up colour fold, the resolve, and one copy of all
// counters to the host. The buffers are sized from what earlier frames needed (first frame: level 0's size for every
// level); finish_part checks afterwards that nothing was dropped.
static int enqueue_part(rt_ctx *ctx, Part &P) {
    const int D = ctx->cfg.max_depth;
    const SceneDev &sh = ctx->sh;
    hipStream_t s = P.stream;
    HIPCK(hipMemsetAsync(P.d_counts, 0, MAX_LEVELS * sizeof(LevelCounters), s));
    HIPCK(hipEventRecord(P.ev_begin, s));
    if (P.n0 == 0) {
        HIPCK(hipEventRecord(P.ev_comp, s));
        HIPCK(hipEventRecord(P.ev_end, s));
        return 0;
    }
    if (ensure_level(ctx, P, 0, P.n0)) return -1;
    for (int l = 1; l < D; l++) {
        if (P.want[l] == 0) P.want[l] = ctx->first_level_rays ? ctx->first_level_rays : P.n0;
        if (ensure_level(ctx, P, l, P.want[l])) return -1;
    }
    for (int l = 0; l < D; l++)
        if (ensure_queues(ctx, P, l, P.lv[l].cap, (uint32_t)sh.n_lights)) return -1;
    for (int k = 0; k <= D; k++)
        if (launch_step(ctx, P, k)) return -1;
    HIPCK(hipEventRecord(P.ev_comp, s));
    for (int l = D - 2; l >= 0; l--) {
        const uint32_t bound = l > 0 ? P.lv[l].cap : P.n0;
        hipLaunchKernelGGL(k_composite, dim3((bound + 255) / 256), dim3(256), 0, s, P.lv[l].b, P.lv[l + 1].b,
                           l > 0 ? &P.d_counts[l - 1].next_count : (const uint32_t *)nullptr, bound);
    }
    hipLaunchKernelGGL(k_resolve, dim3((P.n0 + 255) / 256), dim3(256), 0, s, P.lv[0].b, P.n0, P.g, ctx->d_fb, ctx->d_rgba,
                       ctx->d_compact);
    HIPCK(hipGetLastError());
    HIPCK(hipEventRecord(P.ev_end, s));
    HIPCK(hipMemcpyAsync(P.h_counts, P.d_counts, (size_t)D * sizeof(LevelCounters), hipMemcpyDeviceToHost, s));
    HIPCK(hipEventRecord(P.ev_counts, s));
    return 0;
}

// Waits for the part's frame and checks the counters: children beyond a level's capacity were dropped. That makes the
// frame wrong, so the levels are grown to what the counters ask for and 1 is returned: the caller queues the frame
// again. (Counts below a level that overflowed are too low; a later attempt sees them.)
static int finish_part(rt_ctx *ctx, Part &P) {
    const int D = ctx->cfg.max_depth;
    HIPCK(hipEventSynchronize(P.ev_counts));
    memset(P.n_rays, 0, sizeof P.n_rays);
    P.levels_run = 0;
    P.rec_total[0] = P.rec_total[1] = 0;
    if (P.n0 == 0) return 0;
    bool again = false;
    P.n_rays[0] = P.n0;
    for (int l = 0; l < D; l++) {
        if (P.n_rays[l] == 0) break;
        P.levels_run = l + 1;
        const LevelCounters &hc = P.h_counts[l];
        if (l + 1 < D) {
            if (hc.next_count > P.lv[l + 1].cap) {
                if (hc.next_count > MAX_LEVEL_RAYS) return fail(ctx, "rt_trace: a ray-tree level exceeds 2^27 rays per part (raise rt_config.parts or shard the frame)");
                // the deeper levels usually grow by the same factor
                const double f = (double)hc.next_count / (double)std::max<uint32_t>(P.lv[l + 1].cap, 1u);
                for (int m = l + 1; m < D; m++) {
                    const double w = (m == l + 1 ? (double)hc.next_count : (double)P.lv[m].cap * f) * 1.25;
                    P.want[m] = (uint32_t)std::min<double>(std::max<double>(w, (double)P.want[m]), (double)MAX_LEVEL_RAYS);
                }
                again = true;
            }
            P.n_rays[l + 1] = std::min(hc.next_count, P.lv[l + 1].cap);
        }
        P.rec_total[0] += hc.rec_count[0];
        P.rec_total[1] += hc.rec_count[1];
    }
    return again ? 1 : 0;
}

static int collect_stats(rt_ctx *ctx) {
    const SceneDev &sh = ctx->sh;
    const int NP = ctx->n_parts, D = ctx->cfg.max_depth;
    rt_stats &st = ctx->st;
    const double keep_build = st.kd_build_ms, keep_upload = st.kd_upload_ms, keep_tone = st.tone_ms;
    memset(&st, 0, sizeof st);
    st.kd_build_ms = keep_build; st.kd_upload_ms = keep_upload; st.tone_ms = keep_tone;
    {   // level 0 holds whole tiles; lanes outside the image are not rays
        uint64_t px = 0;
        const FrameGeom &g = ctx->g;
        for (int t = g.shard_index; t < g.n_tiles; t += g.shard_count) {
            int tx, ty;
            if (tile_coords(g, (uint32_t)t, tx, ty)) px += (uint64_t)std::min(8, g.W - tx * 8) * (uint64_t)std::min(8, g.H - ty * 8);
        }
        st.pixels = px;
        st.rays_per_level[0] = px;
    }
    // kernel times are summed over the parts; with more than one part the kernels of different parts share the
    // GPU, so these are occupancy-weighted durations, not exclusive ones
    for (int p = 0; p < NP; p++) {
        const Part &P = ctx->parts[p];
        st.tri_records[0] += P.rec_total[0];
        st.tri_records[1] += P.rec_total[1];
        for (int l = 1; l < P.levels_run; l++) st.rays_per_level[l] += P.n_rays[l];
        for (int l = 0; l < P.levels_run; l++) {
            const LevelCounters &hc = P.h_counts[l];
            st.shadow_rays += (uint64_t)hc.shaded * (uint64_t)sh.n_lights;
            st.kd_closest_rays += hc.kdq_count;
            st.kd_shadow_rays += hc.shq_count;
#ifdef RT_WALK_STATS
            fprintf(stderr, "walk stats part %d level %d: closest iters %llu visits %llu | shadow iters %llu visits %llu\n", p, l,
                    hc.walk_iters[0], hc.walk_visits[0], hc.walk_iters[1], hc.walk_visits[1]);
            for (int j = 0; j < 2; j++)   // 100 MHz ticks -> microseconds
                fprintf(stderr, "    %s: span %.1f us, mean wave lifetime %.1f us\n", j ? "shadow" : "closest",
                        (double)(hc.walk_t1[j] - ~hc.walk_t0[j]) * 0.01,
                        (double)hc.walk_tsum[j] * 0.01 / ((double)ctx->n_cus * ctx->walk_waves_per_cu));
#endif
        }
        if (P.n0 == 0) continue;
        for (int k = 0; ctx->stage_timed && k <= D && k < MAX_STEPS; k++) {
            float ms = 0;
            for (int j = 0; j < N_STAGES; j++) {
                HIPCK(hipEventElapsedTime(&ms, P.ev[k][j], P.ev[k][j + 1]));
                if (k < 8) st.level_stage_ms[k][j] += ms;
                st.stage_ms[j] += ms;
            }
            HIPCK(hipEventElapsedTime(&ms, P.ev[k][0], P.ev[k][N_STAGES]));
            if (k < 8) st.level_kernel_ms[k] += ms;
        }
        float ms = 0;
        HIPCK(hipEventElapsedTime(&ms, ctx->parts[0].ev_begin, P.ev_end));
        st.trace_ms = std::max(st.trace_ms, (double)ms);
        HIPCK(hipEventElapsedTime(&ms, P.ev_comp, P.ev_end));
        st.composite_ms += ms;
    }
    for (int l = 0; l < MAX_LEVELS; l++) st.rays_total += st.rays_per_level[l];
    return 0;
}

int rt_trace_begin(rt_ctx *ctx) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_trace: host-only context (created with device = -1)");
    if (ctx->in_flight) return fail(ctx, "rt_trace_begin: the previous frame was not finished with rt_trace_end");
    HIPCK(hipSetDevice(ctx->cfg.device));
    if (ctx->scene_dirty && upload_scene(ctx)) return -1;
    ctx->stage_timed = ctx->stage_timing;
    for (int p = 0; p < ctx->n_parts; p++)
        if (enqueue_part(ctx, ctx->parts[p])) {
            // part of the frame may be queued already: nothing of it may still run when the caller goes on
            for (int q = 0; q < ctx->n_parts; q++) (void)hipStreamSynchronize(ctx->parts[q].stream);
            return -1;
        }
    // everything queued on the main stream after this call (tone, present, a caller's exchange) follows the trace
    for (int p = 1; p < ctx->n_parts; p++) HIPCK(hipStreamWaitEvent(ctx->stream, ctx->parts[p].ev_end, 0));
    ctx->in_flight = true;
    ctx->traced = true;
    return 0;
}

int rt_trace_end(rt_ctx *ctx) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_trace: host-only context (created with device = -1)");
    if (!ctx->in_flight) return fail(ctx, "rt_trace_end: no frame in flight");
    HIPCK(hipSetDevice(ctx->cfg.device));
    ctx->in_flight = false;
    int reruns = 0;
    for (;;) {
        bool again = false;
        for (int p = 0; p < ctx->n_parts; p++) {
            const int r = finish_part(ctx, ctx->parts[p]);
            if (r < 0) return -1;
            again = again || r == 1;
        }
        if (!again) break;
        if (++reruns > 2 * MAX_LEVELS + 4) return fail(ctx, "rt_trace: buffers still too small after repeated growth");
        // whatever was queued behind the first attempt ran on a frame with dropped rays: drain it, then trace again
        HIPCK(hipStreamSynchronize(ctx->stream));
        for (int p = 0; p < ctx->n_parts; p++)
            if (enqueue_part(ctx, ctx->parts[p])) return -1;
        for (int p = 1; p < ctx->n_parts; p++) HIPCK(hipStreamWaitEvent(ctx->stream, ctx->parts[p].ev_end, 0));
    }
    if (collect_stats(ctx)) return -1;
    return reruns ? 1 : 0;
}

int rt_trace(rt_ctx *ctx) {
    if (rt_trace_begin(ctx)) return -1;
    return rt_trace_end(ctx) < 0 ? -1 : 0;
}

int rt_read_float(rt_ctx *ctx, float *rgb) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_read_float: host-only context (created with device = -1)");
    if (!rgb) return fail(ctx, "rt_read_float: null buffer");
    if (!ctx->traced) return fail(ctx, "rt_read_float: nothing traced yet");
    if (ctx->in_flight) return fail(ctx, "rt_read_float: a frame is in flight, call rt_trace_end first");
    HIPCK(hipSetDevice(ctx->cfg.device));
    HIPCK(hipMemcpy(rgb, ctx->d_fb, (size_t)ctx->g.W * ctx->g.H * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int rt_read_rgba8(rt_ctx *ctx, uint8_t *rgba) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_read_rgba8: host-only context (created with device = -1)");
    if (!rgba) return fail(ctx, "rt_read_rgba8: null buffer");
    if (!ctx->traced) return fail(ctx, "rt_read_rgba8: nothing traced yet");
    if (ctx->in_flight) return fail(ctx, "rt_read_rgba8: a frame is in flight, call rt_trace_end first");
    HIPCK(hipSetDevice(ctx->cfg.device));
    HIPCK(hipMemcpy(rgba, ctx->d_rgba, (size_t)ctx->g.W * ctx->g.H * 4, hipMemcpyDeviceToHost));
    return 0;
}

int rt_read_hit_ids(rt_ctx *ctx, int32_t *ids) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_read_hit_ids: host-only context (created with device = -1)");
    if (!ids) return fail(ctx, "rt_read_hit_ids: null buffer");
    if (!ctx->traced) return fail(ctx, "rt_read_hit_ids: nothing traced yet");
    if (ctx->in_flight) return fail(ctx, "rt_read_hit_ids: a frame is in flight, call rt_trace_end first");
    HIPCK(hipSetDevice(ctx->cfg.device));
    HIPCK(hipMemcpy(ids, ctx->d_hit, (size_t)ctx->g.W * ctx->g.H * 4, hipMemcpyDeviceToHost));
    return 0;
}

static int queue_tone_reduction(rt_ctx *ctx) {
    hipStream_t s = ctx->stream;
    HIPCK(hipEventRecord(ctx->ev_misc[0], s));
    int blocks = (int)std::min<uint32_t>((ctx->n0 + 255) / 256, TONE_BLOCKS);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_tone_partial, dim3(blocks), dim3(256), 0, s, ctx->d_compact, ctx->n0, (double)ctx->cfg.L_max,
                       ctx->d_part_sum, ctx->d_part_cnt);
    hipLaunchKernelGGL(k_tone_final, dim3(1), dim3(64), 0, s, ctx->d_part_sum, ctx->d_part_cnt, blocks, ctx->d_tone_out,
                       ctx->d_tone_cnt, ctx->d_tone_pair);
    {
        static_assert(MAX_PARTS <= 8, "OverflowArgs holds eight parts");
        OverflowArgs oa{};
        oa.n_parts = ctx->n_parts;
        oa.depth = ctx->cfg.max_depth;
        for (int p = 0; p < ctx->n_parts; p++) {
            oa.cnt[p] = ctx->parts[p].d_counts;
            for (int l = 0; l < MAX_LEVELS; l++) oa.cap[p][l] = ctx->parts[p].lv[l].cap;
        }
        hipLaunchKernelGGL(k_overflow_flag, dim3(1), dim3(1), 0, s, oa, ctx->d_tone_pair + 2);
    }
    HIPCK(hipGetLastError());
    HIPCK(hipEventRecord(ctx->ev_misc[1], s));
    return 0;
}

static thread_local std::string g_noalloc_error;

const char *rt_noalloc_last_error(void) { return g_noalloc_error.c_str(); }

int rt_luminance_sums_noalloc(const void *d_rgba8, uint64_t n_pixels, float L_max, void *d_sum_count, int32_t device, void *stream) {
    auto bad = [&](const std::string &m) { g_noalloc_error = "rt_luminance_sums_noalloc: " + m; return -1; };
    if (!d_rgba8 || !d_sum_count) return bad("null buffer");
    if (n_pixels == 0) return bad("zero pixels");
    if (n_pixels > RT_NOALLOC_MAX_PIXELS) return bad("more than RT_NOALLOC_MAX_PIXELS (2^30) pixels");
    // the kernels go to `device`; the caller's current device is put back on every way out (a serving loop or a capture
    // that finds another device current after this call would send its own launches to the wrong GPU)
    int prev = -1;
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return bad(std::string("hipGetDevice: ") + hipGetErrorString(e));
    if (prev != device && (e = hipSetDevice(device)) != hipSuccess) return bad(std::string("hipSetDevice: ") + hipGetErrorString(e));
    auto leave = [&](int rc) {
        if (prev != device) (void)hipSetDevice(prev);
        return rc;
    };
    ToneScratch *sp = nullptr;
    if ((e = scratch_on_current_device(device, &sp)) != hipSuccess) return leave(bad(std::string("hipGetSymbolAddress: ") + hipGetErrorString(e)));
    // every word of the scratch that the second stage reads is written by the first stage of the same call, so
    // nothing needs clearing between calls
    const uint32_t n = (uint32_t)n_pixels;
    int blocks = (int)std::min<uint32_t>((n + 255) / 256, TONE_BLOCKS);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_tone_partial, dim3(blocks), dim3(256), 0, s, (const uint32_t *)d_rgba8, n, (double)L_max, sp->part_sum, sp->part_cnt);
    hipLaunchKernelGGL(k_tone_final, dim3(1), dim3(64), 0, s, sp->part_sum, sp->part_cnt, blocks, &sp->out_sum, &sp->out_cnt,
                       (double *)d_sum_count);
    if ((e = hipGetLastError()) != hipSuccess) return leave(bad(std::string("launch: ") + hipGetErrorString(e)));
    return leave(0);
}

int rt_pick(rt_ctx *ctx, int32_t x, int32_t y, int32_t *id) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_pick: host-only context (created with device = -1)");
    if (!id) return fail(ctx, "rt_pick: null argument");
    if (!ctx->traced) return fail(ctx, "rt_pick: nothing traced yet");
    if (ctx->in_flight) return fail(ctx, "rt_pick: a frame is in flight, call rt_trace_end first");
    if (x < 0 || y < 0 || x >= ctx->g.W || y >= ctx->g.H) return fail(ctx, "rt_pick: pixel outside the render target");
    if ((int)(tile_index(ctx->g, x / 8, y / 8) % (uint32_t)ctx->g.shard_count) != ctx->g.shard_index)
        return fail(ctx, "rt_pick: the pixel belongs to another shard (primary hit ids are not gathered)");
    HIPCK(hipSetDevice(ctx->cfg.device));
    HIPCK(hipMemcpy(id, ctx->d_hit + (size_t)y * ctx->g.W + x, sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

int rt_tone_partial(rt_ctx *ctx, double *sum_log, uint64_t *count) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_tone_partial: host-only context (created with device = -1)");
    if (!sum_log || !count) return fail(ctx, "rt_tone_partial: null argument");
    if (!ctx->traced) return fail(ctx, "rt_tone_partial: nothing traced yet");
    HIPCK(hipSetDevice(ctx->cfg.device));
    if (queue_tone_reduction(ctx)) return -1;
    hipStream_t s = ctx->stream;
    unsigned long long c = 0;
    HIPCK(hipMemcpyAsync(sum_log, ctx->d_tone_out, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCK(hipMemcpyAsync(&c, ctx->d_tone_cnt, sizeof c, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    *count = c;
    float ms = 0;
    HIPCK(hipEventElapsedTime(&ms, ctx->ev_misc[0], ctx->ev_misc[1]));
    ctx->st.tone_ms = ms;
    return 0;
}

int rt_tone_partial_device(rt_ctx *ctx, void **d_sum_count) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_tone_partial_device: host-only context (created with device = -1)");
    if (!d_sum_count) return fail(ctx, "rt_tone_partial_device: null argument");
    if (!ctx->traced) return fail(ctx, "rt_tone_partial_device: nothing traced yet");
    HIPCK(hipSetDevice(ctx->cfg.device));
    if (queue_tone_reduction(ctx)) return -1;
    *d_sum_count = ctx->d_tone_pair;
    return 0;
}

int rt_tone_finish(rt_ctx *ctx, double sum_log, uint64_t count, double *Lbar, double *sf) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_tone_finish: host-only context (created with device = -1)");
    if (count == 0) return fail(ctx, "rt_tone_finish: zero pixels");
    // Lbar = exp(sum / N); sf = ((1.219 + (Ldmax/2)^0.4) / (1.219 + Lbar^0.4))^2.5  (js/8-kd_tree.js:2568-2569)
    ctx->Lbar = exp(sum_log / (double)count);
    ctx->sf = pow((1.219 + pow((double)ctx->cfg.L_dmax * 0.5, 0.4)) / (1.219 + pow(ctx->Lbar, 0.4)), 2.5);
    ctx->tone_on_device = false;
    if (Lbar) *Lbar = ctx->Lbar;
    if (sf) *sf = ctx->sf;
    HIPCK(hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(k_set_tone, dim3(1), dim3(1), 0, ctx->stream, ctx->Lbar, ctx->sf, ctx->d_tone, ctx->d_tone_d);
    HIPCK(hipGetLastError());
    return 0;
}

int rt_tone_finish_device(rt_ctx *ctx, const void *d_sum_count) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_tone_finish_device: host-only context (created with device = -1)");
    if (!d_sum_count) return fail(ctx, "rt_tone_finish_device: null argument");
    HIPCK(hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(k_tone_finish, dim3(1), dim3(1), 0, ctx->stream, (const double *)d_sum_count, (double)ctx->cfg.L_dmax,
                       ctx->d_tone, ctx->d_tone_d);
    HIPCK(hipGetLastError());
    ctx->tone_on_device = true;
    return 0;
}

int rt_tone_get(rt_ctx *ctx, double *Lbar, double *sf) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_tone_get: host-only context (created with device = -1)");
    if (ctx->tone_on_device) {
        double v[2];
        HIPCK(hipSetDevice(ctx->cfg.device));
        HIPCK(hipMemcpyAsync(v, ctx->d_tone_d, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));
        ctx->Lbar = v[0];
        ctx->sf = v[1];
        ctx->tone_on_device = false;
    }
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
                       c.L_max, c.L_dmax, c.A, ctx->d_tone, ctx->d_canvas);
    HIPCK(hipGetLastError());
    if (canvas) {
        HIPCK(hipMemcpyAsync(canvas, ctx->d_canvas, (size_t)c.width * c.height * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int rt_stream(rt_ctx *ctx, void **stream) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_stream: host-only context (created with device = -1)");
    if (!stream) return fail(ctx, "rt_stream: null argument");
    *stream = (void *)ctx->stream;
    return 0;
}

int rt_set_option(rt_ctx *ctx, const char *name, double value) {
    if (!ctx) return -1;
    if (!name) return fail(ctx, "rt_set_option: null name");
    if (!strcmp(name, "first_level_rays")) {
        if (!(value >= 0 && value <= 2147483392.0)) return fail(ctx, "rt_set_option: capacity out of range");
        ctx->first_level_rays = (uint32_t)value;
        return 0;
    }
    if (!strcmp(name, "prune")) {
        ctx->prune = value != 0.0;
        ctx->scene_dirty = true;
        return 0;
    }
    if (!strcmp(name, "stage_timing")) {
        ctx->stage_timing = value != 0.0;
        return 0;
    }
    if (!strcmp(name, "walk_waves_per_cu")) {
            if (!(value >= 1 && value <= 32)) return fail(ctx, "rt_set_option: walk_waves_per_cu must be in 1..32");
        ctx->walk_waves_per_cu = (int)value;
        return 0;
    }
    return fail(ctx, "rt_set_option: unknown option");
}

int rt_device_rgba8(rt_ctx *ctx, void **dptr, size_t *bytes) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_device_rgba8: host-only context (created with device = -1)");
    if (dptr) *dptr = ctx->d_rgba;
    if (bytes) *bytes = (size_t)ctx->g.W * ctx->g.H * 4;
    return 0;
}

int rt_device_float(rt_ctx *ctx, void **dptr, size_t *bytes) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_device_float: host-only context (created with device = -1)");
    if (dptr) *dptr = ctx->d_fb;
    if (bytes) *bytes = (size_t)ctx->g.W * ctx->g.H * 3 * sizeof(float);
    return 0;
}

int rt_device_canvas(rt_ctx *ctx, void **dptr, size_t *bytes) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_device_canvas: host-only context (created with device = -1)");
    if (dptr) *dptr = ctx->d_canvas;
    if (bytes) *bytes = (size_t)ctx->cfg.width * ctx->cfg.height * 4;
    return 0;
}

int rt_device_shard_rgba8(rt_ctx *ctx, void **dptr, size_t *bytes) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_device_shard_rgba8: host-only context (created with device = -1)");
    if (dptr) *dptr = ctx->d_compact;
    if (bytes) *bytes = (size_t)ctx->n0 * 4;   // 0 for a shard without tiles (the buffer itself is never smaller than one tile)
    return 0;
}

static int assemble_into(rt_ctx *ctx, const char *who, const void *gathered, size_t stride_bytes, int32_t n_shards, uint32_t *target) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, (std::string(who) + ": host-only context (created with device = -1)").c_str());
    if (!gathered || n_shards < 1 || stride_bytes % 256) return fail(ctx, (std::string(who) + ": bad arguments").c_str());
    HIPCK(hipSetDevice(ctx->cfg.device));
    // a shard of n_shards holds every n_shards-th tile: `stride` cannot be more than one tile beyond that, and the kernel
    // indexes the gathered pixels with 32 bits
    const uint64_t tiles_per_buffer = stride_bytes / 256;
    if (tiles_per_buffer * (uint64_t)n_shards > (uint64_t)ctx->g.n_tiles + (uint64_t)n_shards || tiles_per_buffer * 64 * (uint64_t)n_shards >= (1ull << 32))
        return fail(ctx, (std::string(who) + ": stride_bytes is larger than a shard of this frame").c_str());
    uint32_t stride = (uint32_t)(stride_bytes / 4);
    uint64_t total = (uint64_t)stride * (uint64_t)n_shards;
    hipLaunchKernelGGL(k_assemble, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t *)gathered,
                       stride, n_shards, ctx->g, target);
    HIPCK(hipGetLastError());
    return 0;
}

int rt_assemble_shards(rt_ctx *ctx, const void *gathered, size_t stride_bytes, int32_t n_shards) {
    if (assemble_into(ctx, "rt_assemble_shards", gathered, stride_bytes, n_shards, ctx ? ctx->d_rgba : nullptr)) return -1;
    ctx->traced = true;
    return 0;
}

int rt_assemble_shards_canvas(rt_ctx *ctx, const void *gathered, size_t stride_bytes, int32_t n_shards) {
    if (ctx && ctx->cfg.supersampling != 1) return fail(ctx, "rt_assemble_shards_canvas: needs supersampling 1 (canvas and render target have one geometry then)");
    return assemble_into(ctx, "rt_assemble_shards_canvas", gathered, stride_bytes, n_shards, ctx ? ctx->d_canvas : nullptr);
}

int rt_present_shard(rt_ctx *ctx) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_present_shard: host-only context (created with device = -1)");
    if (!ctx->traced) return fail(ctx, "rt_present_shard: nothing traced yet");
    if (ctx->cfg.supersampling != 1) return fail(ctx, "rt_present_shard: needs supersampling 1 (with more the shader reads across tiles: present the assembled frame on the root)");
    HIPCK(hipSetDevice(ctx->cfg.device));
    const rt_config &c = ctx->cfg;
    if (ctx->n0) {
        hipLaunchKernelGGL(k_present_compact, dim3((ctx->n0 + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_compact, ctx->n0, c.tr_method,
                           c.L_max, c.L_dmax, c.A, ctx->d_tone, ctx->d_compact_canvas);
        HIPCK(hipGetLastError());
    }
    return 0;
}

int rt_device_shard_canvas(rt_ctx *ctx, void **dptr, size_t *bytes) {
    if (!ctx) return -1;
    if (ctx->host_only) return fail(ctx, "rt_device_shard_canvas: host-only context (created with device = -1)");
    if (dptr) *dptr = ctx->d_compact_canvas;
    if (bytes) *bytes = (size_t)ctx->n0 * 4;
    return 0;
}

int rt_get_stats(rt_ctx *ctx, rt_stats *out) {
    if (!ctx) return -1;
    if (!out) return fail(ctx, "rt_get_stats: null argument");
    ctx->st.kd_device_fallbacks = ctx->kd_device_fallbacks;
    *out = ctx->st;
    return 0;
}

}  // extern "C"
