// HIP kernels of the frame (gfx950, wave64).
//
// The reference evaluates a per-pixel ray TREE inside one fragment-shader invocation
// (js/8-kd_tree.js:822-1057): slot n spawns a reflected ray into slot 2n+1 and a refracted ray into 2n+2,
// then colours are folded bottom-up. Here the tree is processed one LEVEL at a time over compact ray
// queues, and each level runs as a short pipeline of kernels whose lanes all do the same kind of work:
//
//   k_gen_prims    one lane per ray: (level 0: generate the primary ray) closest hit over the analytic
//                  primitives; rays whose interval overlaps the k-d root box are appended to the
//                  closest-hit traversal queue
//   k_kd_walk<0>   persistent waves: closest-hit k-d traversal; a lane that finishes its ray pulls the
//                  next one from the queue, so lanes stay busy although ray costs differ by 100x
//   k_shade_a      one lane per ray: miss -> background; hit -> surface point, one shadow ray per light:
//                  analytic any-hit test here, survivors that reach the k-d box go to the shadow queue
//   k_kd_walk<1>   persistent waves: any-hit k-d traversal of the shadow queue, sets a bit per light
//   k_shade_b      one lane per ray: Phong terms with the shadow bits, spawns the children into the next
//                  level's queue (wave-aggregated append: one atomic per wave, ranks from the ballot)
//
// Each parent remembers the queue slots of its two children; k_composite then folds colours level by
// level in the reference's order (refracted child first, then reflected: js/8-kd_tree.js:1041-1054 walks
// the slots downwards, and 2n+2 > 2n+1), so the float result is the same expression tree the shader
// evaluates.
#pragma once
#include "rt_device.h"

namespace rt {

// one ray-tree level, structure of arrays, `cap` entries each
struct LevelBuf {
    float *ox, *oy, *oz, *dx, *dy, *dz;  // ray
    float *ht;                           // closest hit distance (INFINITY constant = miss so far)
    uint32_t *hp;                        // analytic winner | last analytic hit << 16 (0xffff = none)
    int32_t *hf;                         // winning mesh face or -1
    uint32_t *shadow;                    // bit i: light i is occluded
    float *cr, *cg, *cb;                 // local colour, later the folded colour
    float *kr, *kt;                      // kReflect / kRefract of this node
    int32_t *refl, *refr;                // queue slot of the child in the next level, -1 = none
};
constexpr int LEVEL_ARRAYS = 17;

struct ShadowQ {                         // shadow rays that must walk the k-d tree
    float *ox, *oy, *oz, *dx, *dy, *dz;
    uint32_t *id;                        // ray index << 5 | light
};
constexpr int SHADOWQ_ARRAYS = 7;
constexpr int MAX_LIGHTS = 32;

struct FrameGeom {
    int32_t W, H, tiles_x, n_tiles, shard_index, shard_count;
};

// device counters of one level; every counter sits in its own 128-byte line so that atomics on different
// counters do not queue behind each other in one L2 channel
struct LevelCounters {
    uint32_t kdq_count, pad0[31];
    uint32_t kdq_head, pad1[31];
    uint32_t shq_count, pad2[31];
    uint32_t shq_head, pad3[31];
    uint32_t next_count, pad4[31];
    uint32_t shaded, pad5[31];
    unsigned long long dbg[16];   // RT_DBG_COUNT builds: [0] wave iterations [1] lane actions [2] batches [3] pairs
                                  // [4] refills [5] restarts [6] leaf actions [7] emitting iterations  (+8 for any-hit)
};

constexpr int BLOCK = 1024;   // lanes per workgroup of the one-lane-per-ray kernels

struct LevelArgs {
    SceneDev S;             // by value: lands in the kernel-argument segment, read through scalar loads
    LevelBuf cur, nxt;
    ShadowQ shq;
    uint32_t *kdq;          // ray indices for the closest-hit traversal
    LevelCounters *cnt;
    uint32_t n_rays, nxt_cap, shq_cap, pad;
    int32_t *hit_ids;       // level 0 only, full frame
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

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Reserves `c` consecutive slots per lane in a global queue with ONE atomic per workgroup (an atomic per
// wave on a single address serialises at roughly 10 ns each and was the floor of these kernels):
// wave-level inclusive scan by shuffles, wave totals through LDS, the workgroup total added once.
// Must be called by every lane of the workgroup. Returns the lane's first slot.
__device__ __forceinline__ uint32_t block_reserve(uint32_t c, uint32_t *counter, uint32_t *s_wave /* [BLOCK/64 + 1] */) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        const uint32_t nw = blockDim.x >> 6;
        for (uint32_t w = 0; w < nw; w++) { const uint32_t t = s_wave[w]; s_wave[w] = total; total += t; }
        const uint32_t base = total ? atomicAdd(counter, total) : 0u;
        s_wave[nw] = base;
    }
    __syncthreads();
    const uint32_t r = s_wave[blockDim.x >> 6] + s_wave[wave] + (incl - c);
    __syncthreads();   // s_wave may be reused by the next call
    return r;
}

__device__ __forceinline__ bool enters_box(const SceneDev &S, V3 ro, V3 rd) {
    float tmin, tmax;
    ray_box(S, ro, rd, tmin, tmax);
    return tmin < tmax;   // the reference's `while (kdTMax < maxDist)` runs at least once
}

// ---- stage 1 -----------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_gen_prims(LevelArgs A) {
    __shared__ uint32_t s_wave[BLOCK / WAVE + 1];
    const SceneDev &S = A.S;
    const uint32_t tid = blockIdx.x * BLOCK + threadIdx.x;
    bool active = tid < A.n_rays;
    V3 ro, rd;
    if (A.level == 0) {
        int px = 0, py = 0;
        active = active && pixel_of(A.g, tid, px, py);
        ro = ld3(S.cam);
        float fx = ((float)px + 0.5f) / (float)A.g.W, fy = ((float)py + 0.5f) / (float)A.g.H;
        V3 dir = mix3(mix3(ld3(S.r00), ld3(S.r01), fy), mix3(ld3(S.r10), ld3(S.r11), fy), fx);
        rd = normalize(dir);
        if (tid < A.n_rays) {
            A.cur.ox[tid] = ro.x; A.cur.oy[tid] = ro.y; A.cur.oz[tid] = ro.z;
            A.cur.dx[tid] = rd.x; A.cur.dy[tid] = rd.y; A.cur.dz[tid] = rd.z;
        }
    } else {
        uint32_t i = active ? tid : 0;
        ro = v3(A.cur.ox[i], A.cur.oy[i], A.cur.oz[i]);
        rd = v3(A.cur.dx[i], A.cur.dy[i], A.cur.dz[i]);
    }
    float tMin = S.inf;
    uint32_t best = 0xffffu, last = 0xffffu;
    int32_t face = -1;
    bool to_kd = false;
    if (active) {
        // analytic primitives in scene order, strict `t < tMin` (js/8-kd_tree.js:1202-1214)
        for (int i = 0; i < S.n_prims; i++) {
            PHit h;
            if (hit_prim(S, S.prims[i], ro, rd, h)) {
                last = (uint32_t)i;
                if (h.t < tMin) { tMin = h.t; best = (uint32_t)i; }
            }
        }
        if (S.render_triangles) {
            if (S.use_kd) {
                to_kd = enters_box(S, ro, rd);
            } else {
                for (int f = 0; f < S.n_faces; f++) {   // USE_KD_TREE = false: every face (js/8-kd_tree.js:965-988)
                    float t;
                    if (hit_tri(S.tris[f], S.tri_eps, ro, rd, t) && t < tMin) { tMin = t; face = f; }
                }
            }
        }
    }
    if (tid < A.n_rays) {
        A.cur.ht[tid] = tMin;
        A.cur.hp[tid] = active ? (best | (last << 16)) : 0xfffeffffu;   // 0xfffe in the high half: lane outside the image
        A.cur.hf[tid] = face;
    }
    const uint32_t slot = block_reserve(to_kd ? 1u : 0u, &A.cnt->kdq_count, s_wave);
    if (to_kd) A.kdq[slot] = tid;
}

// ---- stages 2 and 4: persistent k-d traversal ------------------------------------------------------
// Tree walk. The reference restarts from the "pushed" node for every cell the ray crosses (kd-pushdown,
// js/8-kd_tree.js:862-962); here the far child and the interval end are kept on a per-lane stack in LDS
// instead, which visits exactly the same leaves in the same order with the same comparisons: on a restart
// every ancestor decision is re-derived from the same plane distance t, the same kdTMax and a kdTMin that
// only grew, so it repeats, and the first node to decide differently is the one whose far child the stack
// holds. A NaN plane distance (0/0) ends the walk after the current leaf in both forms (`kdTMax < maxDist`
// is false for NaN). The plane distance uses a true IEEE division: a reciprocal multiply can differ in the
// last bit and flip a `t >= kdTMax` decision.
//
// Closest hit keeps the minimum t over EVERY leaf crossed - no early exit, because a leaf keeps only its
// first four faces and a later cell may list a triangle that an earlier, truncated cell dropped. That makes
// the walk independent of the triangle tests, so the two are decoupled inside the wave:
//   - every lane owns one ray and takes one tree action per iteration (a descent step, or at a leaf:
//     queue its faces and pop the stack); lanes whose ray is finished are refilled from the global queue;
//   - the faces go into a wave-local FIFO in LDS as (owner lane, face) pairs; whenever 64 pairs are waiting
//     the whole wave runs one Moller-Trumbore test per lane on them, whatever lane owns the ray;
//   - results come back through LDS: closest hit = 64-bit atomic min of (t bits, FIFO position) per owner,
//     so equal distances resolve to the face met first, as the shader's strict `hitInfo.t < tMin` does
//     (js/8-kd_tree.js:948-949); any hit = a flag per owner that stops that owner's walk.
// Before a lane takes a new ray the FIFO is drained, so no pair ever outlives its ray.
struct WalkArgs {
    SceneDev S;
    LevelBuf cur;
    ShadowQ shq;
    const uint32_t *kdq;
    LevelCounters *cnt;
    uint32_t shq_cap;
};

#ifndef RT_REFILL_MIN_IDLE
#define RT_REFILL_MIN_IDLE 16
#endif
#ifndef RT_SHORT_STACK
#define RT_SHORT_STACK 8
#endif
#ifndef RT_WALK_WAVES_PER_CU
#define RT_WALK_WAVES_PER_CU 16
#endif
constexpr int REFILL_MIN_IDLE = RT_REFILL_MIN_IDLE;   // drain + refill once this many lanes of the wave have finished walking
constexpr int REFILL_CHUNK = 256;     // rays a wave reserves from the global queue per atomic
constexpr int FIFO_CAP = 512;         // >= 63 waiting + 64 lanes * 4 faces; power of two
constexpr uint32_t OWNER_SHIFT = 26;  // pair = owner lane << 26 | face index
// Far children wait on a short per-lane stack (ring of SHORT_STACK entries in LDS; a push onto a full
// ring overwrites the oldest entry). When the ring runs dry although the ray has not reached the end of the
// root interval, the walk restarts at the root with kdTMin = the end of the cell just finished - which is
// what the reference does after EVERY cell, so a restart re-derives exactly the dropped decisions.
constexpr int SHORT_STACK = RT_SHORT_STACK;   // power of two

struct WalkLds {
    uint2 stack[SHORT_STACK * WAVE];
    float ro[3][WAVE], rd[3][WAVE];
    unsigned long long best[WAVE];    // closest: (t bits << 32 | FIFO position), ~0 = none
    uint32_t best_face[WAVE];
    uint32_t hit_flag[WAVE];          // any-hit: owner's ray is occluded
    uint32_t fifo[FIFO_CAP];
    uint32_t fifo_tail;               // total pairs ever queued by this wave
};

// One batch of up to 64 (owner, face) pairs, split in two so that the triangle loads of a batch travel
// together with the node loads of the walk: `batch_issue` reads the pairs and starts the loads,
// `batch_finish` (one loop iteration later) runs the tests and posts the results.
struct Batch {
    bool valid;
    uint32_t owner, face, pos;
    float4 a, b;   // v0.xyz e1.x | e1.yz e2.xy
    float c;       // e2.z
};

__device__ __forceinline__ void batch_issue(const SceneDev &S, WalkLds &L, uint32_t lane, uint32_t &q_head, uint32_t q_tail,
                                            Batch &B) {
    const uint32_t avail = q_tail - q_head;
    const uint32_t n = avail < (uint32_t)WAVE ? avail : (uint32_t)WAVE;
    B.valid = lane < n;
    if (B.valid) {
        const uint32_t e = L.fifo[(q_head + lane) & (FIFO_CAP - 1)];
        B.owner = e >> OWNER_SHIFT;
        B.face = e & ((1u << OWNER_SHIFT) - 1u);
        B.pos = q_head + lane;
        const float4 *t4 = reinterpret_cast<const float4 *>(S.tris + B.face);
        B.a = t4[0];
        B.b = t4[1];
        B.c = reinterpret_cast<const float *>(t4 + 2)[0];
    }
    q_head += n;
}

template <bool ANY>
__device__ __forceinline__ void batch_finish(const SceneDev &S, WalkLds &L, Batch &B) {
    unsigned long long key = ~0ull;
#if defined(RT_DBG_NO_TEST)
    if (B.valid && B.a.x == 1.2345e33f) L.hit_flag[B.owner] = 2u;
    B.valid = false;
    return;
#endif
    if (B.valid && (!ANY || !L.hit_flag[B.owner])) {
        const uint32_t o = B.owner;
        const V3 ro = v3(L.ro[0][o], L.ro[1][o], L.ro[2][o]);
        const V3 rd = v3(L.rd[0][o], L.rd[1][o], L.rd[2][o]);
        float t;
        if (hit_tri_regs(v3(B.a.x, B.a.y, B.a.z), v3(B.a.w, B.b.x, B.b.y), v3(B.b.z, B.b.w, B.c), S.tri_eps, ro, rd, t)) {
            if (ANY) {
                L.hit_flag[o] = 1u;
            } else {
                // t >= 0 here, so the float bits order like the values; -0 is folded onto +0
                const uint32_t tb = (t == 0.0f) ? 0u : __float_as_uint(t);
                key = ((unsigned long long)tb << 32) | (unsigned long long)B.pos;
                atomicMin(&L.best[o], key);
            }
        }
    }
    if (!ANY) {
        __syncthreads();
        if (key != ~0ull && L.best[B.owner] == key) L.best_face[B.owner] = B.face;   // keys are unique: one winner
    }
    __syncthreads();
    B.valid = false;
}

template <bool ANY>
__global__ __launch_bounds__(WAVE) void k_kd_walk(WalkArgs A) {
    __shared__ WalkLds L;
    const SceneDev &S = A.S;
    const KdNodeDev *__restrict__ nodes = S.nodes;
    const uint32_t lane = threadIdx.x;
    uint2 *stack = L.stack + lane;
    const uint32_t total = ANY ? min(A.cnt->shq_count, A.shq_cap) : A.cnt->kdq_count;
    uint32_t *head = ANY ? &A.cnt->shq_head : &A.cnt->kdq_head;
    const KdNodeDev root = nodes[0];

    enum { IDLE = 0, WALK = 1, WAIT = 2, FETCH = 3 };
    int state = IDLE;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    float tmin = 0.0f, tmax = 0.0f, maxDist = 0.0f, prim_t = 0.0f;
    uint32_t id = 0, info = 0;
    float split = 0.0f;
    uint32_t sp_top = 0, sp_cnt = 0;
    uint32_t q_head = 0, q_tail = 0;          // wave-uniform FIFO positions (grow without wrapping back)
    uint32_t chunk_pos = 0, chunk_end = 0;    // wave-uniform: rays reserved from the global queue
    bool more = true;
    bool leaf_pending = false;                // lf holds a leaf loaded in the previous iteration
    KdLeafDev lf = {{-1, -1, -1, -1}};
    Batch B;
    B.valid = false;
    B.owner = B.face = B.pos = 0;
    B.a = B.b = make_float4(0, 0, 0, 0);
    B.c = 0.0f;

#if defined(RT_DBG_COUNT)
    unsigned long long c_iter = 0, c_act = 0, c_batch = 0, c_pairs = 0, c_refill = 0, c_restart = 0, c_leaf = 0, c_emit_it = 0;
#define DBG(x) x
#else
#define DBG(x)
#endif
    for (;;) {
        DBG(c_iter++;)
        // ---- 1. finish the batch whose triangle loads were started one iteration ago
        if (__any(B.valid)) batch_finish<ANY>(S, L, B);

        // ---- 2. queue the faces of leaves reached in the previous iteration; the k-th faces of all lanes
        //         form one group, groups in slot order, so one owner's pairs keep the leaf's face order
        {
            uint32_t n_emit = 0;
            if (leaf_pending) n_emit = (lf.face[0] > -1) + (lf.face[1] > -1) + (lf.face[2] > -1) + (lf.face[3] > -1);
            leaf_pending = false;
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned long long m = __ballot(n_emit > (uint32_t)k);
                if (!m) break;
                DBG(if (k == 0) c_emit_it++;)
                if (n_emit > (uint32_t)k)
                    L.fifo[(q_tail + (uint32_t)__popcll(m & below)) & (FIFO_CAP - 1)] = (lane << OWNER_SHIFT) | (uint32_t)lf.face[k];
                q_tail += (uint32_t)__popcll(m);
            }
        }

        // ---- 3. while the global queue has rays: refill as soon as enough lanes are free; afterwards finished
        //         lanes just wait for the last walkers of the wave
        const int n_not_walking = __popcll(__ballot(state != WALK));
        if (more ? (n_not_walking >= REFILL_MIN_IDLE) : (n_not_walking == WAVE)) {
            __syncthreads();
            DBG(c_refill++;)
            while (q_head != q_tail) {   // drain: no pair may outlive its ray
                DBG(c_batch++; c_pairs += (q_tail - q_head < 64u ? q_tail - q_head : 64u);)
                batch_issue(S, L, lane, q_head, q_tail, B);
                batch_finish<ANY>(S, L, B);
            }
            if (state == WAIT) {
                if (ANY) {
                    if (L.hit_flag[lane]) atomicOr(&A.cur.shadow[id >> 5], 1u << (id & 31u));
                } else {
                    const unsigned long long key = L.best[lane];
                    if (key != ~0ull && __uint_as_float((uint32_t)(key >> 32)) < prim_t) {
                        const int32_t f = (int32_t)L.best_face[lane];
                        float t = 0.0f;
                        hit_tri(S.tris[f], S.tri_eps, v3(ox, oy, oz), v3(dx, dy, dz), t);   // exact t of the winner
                        A.cur.ht[id] = t;
                        A.cur.hf[id] = f;
                    }
                }
                state = IDLE;
            }
            if (more) {
                const unsigned long long idle = __ballot(state == IDLE);
                const uint32_t n_idle = (uint32_t)__popcll(idle);
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t left = chunk_end - chunk_pos;
                uint32_t q = 0xffffffffu;
                if (left < n_idle) {
                    // the rays left in the old chunk are taken first, the rest comes from a new chunk
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(head, (uint32_t)REFILL_CHUNK);
                    base = __shfl(base, 0);
                    q = rank < left ? chunk_pos + rank : base + (rank - left);
                    chunk_pos = base + (n_idle - left);
                    chunk_end = base + (uint32_t)REFILL_CHUNK;
                    if (base >= total) { more = false; chunk_pos = chunk_end = 0; }
                } else {
                    q = chunk_pos + rank;
                    chunk_pos += n_idle;
                }
                if (state == IDLE && q < total) {
                    if (ANY) {
                        ox = A.shq.ox[q]; oy = A.shq.oy[q]; oz = A.shq.oz[q];
                        dx = A.shq.dx[q]; dy = A.shq.dy[q]; dz = A.shq.dz[q];
                        id = A.shq.id[q];
                    } else {
                        id = A.kdq[q];
                        ox = A.cur.ox[id]; oy = A.cur.oy[id]; oz = A.cur.oz[id];
                        dx = A.cur.dx[id]; dy = A.cur.dy[id]; dz = A.cur.dz[id];
                        prim_t = A.cur.ht[id];
                    }
                    ray_box(S, v3(ox, oy, oz), v3(dx, dy, dz), tmin, maxDist);
                    tmax = maxDist;
                    sp_top = sp_cnt = 0;
                    info = root.info;
                    split = root.split;
                    L.ro[0][lane] = ox; L.ro[1][lane] = oy; L.ro[2][lane] = oz;
                    L.rd[0][lane] = dx; L.rd[1][lane] = dy; L.rd[2][lane] = dz;
                    L.best[lane] = ~0ull;
                    L.hit_flag[lane] = 0u;
                    // queued rays always satisfy tmin < maxDist; the test keeps the walk self-contained
                    state = (tmin < maxDist) ? WALK : WAIT;
                }
                __syncthreads();
            }
            if (!__any(state != IDLE)) {
                if (!more) break;
                continue;
            }
        }
        DBG(c_act += (unsigned long long)__popcll(__ballot(state == WALK));)

        // ---- 4. start the loads of the next batch
#if defined(RT_DBG_DROP_FIFO)
        q_head = q_tail;
#endif
        if (q_tail - q_head >= (uint32_t)WAVE) {
            __syncthreads();
            DBG(c_batch++; c_pairs += 64;)
            batch_issue(S, L, lane, q_head, q_tail, B);
        }

        // ---- 5. one tree action per walking lane; its loads are consumed in the next iteration
        if (state == WALK) {
            if (ANY && L.hit_flag[lane]) {
                state = WAIT;
            } else if ((info >> 30) == 1u) {
                const int a = (int)((info >> 28) & 3u);
                const float o = a == 0 ? ox : (a == 1 ? oy : oz), d = a == 0 ? dx : (a == 1 ? dy : dz);
                const float t = (split - o) / d;
                uint32_t nearc = info & 0x0fffffffu, farc = nearc + 1;
                if (d <= 0.0f) { const uint32_t tmp = nearc; nearc = farc; farc = tmp; }
                uint32_t next;
                if (t >= tmax) {
                    next = nearc;
                } else if (t <= tmin) {
                    next = farc;
                } else {
                    stack[(sp_top & (SHORT_STACK - 1)) * WAVE] = make_uint2(farc, __float_as_uint(tmax));
                    sp_top++;
                    sp_cnt = sp_cnt < (uint32_t)SHORT_STACK ? sp_cnt + 1 : sp_cnt;
                    next = nearc;
                    tmax = t;
                }
                const KdNodeDev n = nodes[next];
                info = n.info;
                split = n.split;
            } else {
                DBG(atomicAdd(&A.cnt->dbg[(ANY ? 8 : 0) + 6], 1ull);)
#if !defined(RT_DBG_NO_EMIT)
                if ((info >> 30) == 3u) {
                    lf = S.leaves[info & 0x3fffffffu];
                    leaf_pending = true;
                }
#endif
                if (!(tmax < maxDist)) {   // last cell of the ray (or NaN interval)
                    state = WAIT;
                } else if (sp_cnt) {
                    sp_top--;
                    sp_cnt--;
                    const uint2 e = stack[(sp_top & (SHORT_STACK - 1)) * WAVE];
                    tmin = tmax;
                    tmax = __uint_as_float(e.y);
                    const KdNodeDev n = nodes[e.x];
                    info = n.info;
                    split = n.split;
                } else {                   // ring ran dry: restart from the root for the rest of the ray
                    DBG(atomicAdd(&A.cnt->dbg[(ANY ? 8 : 0) + 5], 1ull);)
                    tmin = tmax;
                    tmax = maxDist;
                    info = root.info;
                    split = root.split;
                }
            }
        }
    }
#if defined(RT_DBG_COUNT)
    if (lane == 0) {
        unsigned long long *d = A.cnt->dbg + (ANY ? 8 : 0);
        atomicAdd(d + 0, c_iter); atomicAdd(d + 1, c_act); atomicAdd(d + 2, c_batch); atomicAdd(d + 3, c_pairs);
        atomicAdd(d + 4, c_refill); atomicAdd(d + 7, c_emit_it);
    }
    (void)c_restart; (void)c_leaf;
#endif
#undef DBG
}

// ---- surface reconstruction ------------------------------------------------------------------------
struct Surface {
    V3 point, normal, local;
    int32_t mat;
    int32_t hit_id;
};

// Rebuilds what the shader holds in HitInfo after the closest-hit search from the compact hit record.
// Re-running the winning analytic test reproduces its normal and object-space point bit for bit.
__device__ __forceinline__ bool make_surface(const SceneDev &S, V3 ro, V3 rd, float t, uint32_t hp, int32_t face, Surface &sf) {
    const uint32_t best = hp & 0xffffu, last = hp >> 16;
    if (face < 0 && best == 0xffffu) return false;
    sf.point = ro + rd * t;
    if (face >= 0) {
        const TriDev &T = S.tris[face];
        sf.normal = normalize(cross(ld3(T.e1), ld3(T.e2)));
        // intersectTriangle never writes localHitPoint (js/8-kd_tree.js:954): the shader still holds the
        // object-space point of the last analytic test that succeeded for this ray (zero if none did)
        sf.local = v3(0, 0, 0);
        if (last != 0xffffu) {
            PHit h;
            hit_prim(S, S.prims[last], ro, rd, h);
            sf.local = h.loc;
        }
        sf.mat = T.mat;
        sf.hit_id = S.n_prims + face;
    } else {
        PHit h;
        hit_prim(S, S.prims[best], ro, rd, h);
        sf.normal = h.nrm;
        sf.local = h.loc;
        sf.mat = S.prims[best].mat;
        sf.hit_id = (int32_t)best;
    }
    return true;
}

// ---- stage 3: background, shadow rays ------------------------------------------------------------------
// illuminate normalises S = L - P and castShadow normalises it again (js/8-kd_tree.js:788-789, :576)
__device__ __forceinline__ V3 shadow_dir(const SceneDev &S, int l, V3 point) {
    return normalize(normalize(ld3(S.lights[l].pos) - point));
}

__global__ __launch_bounds__(BLOCK) void k_shade_a(LevelArgs A) {
    __shared__ uint32_t s_wave[BLOCK / WAVE + 1];
    const SceneDev &S = A.S;
    const uint32_t tid = blockIdx.x * BLOCK + threadIdx.x;
    const bool in_range = tid < A.n_rays;
    const uint32_t i = in_range ? tid : 0;
    const uint32_t hp = A.cur.hp[i];
    const bool active = in_range && (hp >> 16) != 0xfffeu;
    V3 ro = v3(A.cur.ox[i], A.cur.oy[i], A.cur.oz[i]), rd = v3(A.cur.dx[i], A.cur.dy[i], A.cur.dz[i]);
    Surface sf;
    const bool hit = active && make_surface(S, ro, rd, A.cur.ht[i], hp, A.cur.hf[i], sf);
    if (in_range && !hit) {
        V3 c = active ? ld3(S.bg) : v3(0, 0, 0);
        A.cur.cr[tid] = c.x; A.cur.cg[tid] = c.y; A.cur.cb[tid] = c.z;
        A.cur.kr[tid] = 0.0f; A.cur.kt[tid] = 0.0f;
        A.cur.refl[tid] = -1; A.cur.refr[tid] = -1;
        if (A.level == 0 && active) {
            int px, py;
            pixel_of(A.g, tid, px, py);
            A.hit_ids[(size_t)py * A.g.W + px] = -1;
        }
    }
    uint32_t occluded_mask = 0, kd_mask = 0;
    const V3 so = hit ? sf.point + sf.normal * S.eps : v3(0, 0, 0);
    if (hit) {
        for (int l = 0; l < S.n_lights; l++) {   // castShadow per light (js/8-kd_tree.js:575-714)
            const V3 sd = shadow_dir(S, l, sf.point);
            bool occluded = false;
            for (int p = 0; p < S.n_prims && !occluded; p++) {
                PHit h;
                occluded = hit_prim(S, S.prims[p], so, sd, h);
            }
            if (!occluded && S.render_triangles) {
                if (S.use_kd) {
                    if (enters_box(S, so, sd)) kd_mask |= 1u << l;
                } else {
                    for (int f = 0; f < S.n_faces && !occluded; f++) {
                        float t;
                        occluded = hit_tri(S.tris[f], S.tri_eps, so, sd, t);
                    }
                }
            }
            if (occluded) occluded_mask |= 1u << l;
        }
    }
    if (in_range) A.cur.shadow[tid] = occluded_mask;
    // shadow rays that must walk the tree: one reservation for the whole workgroup
    uint32_t slot = block_reserve((uint32_t)__popc(kd_mask), &A.cnt->shq_count, s_wave);
    while (kd_mask) {
        const int l = __ffs((int)kd_mask) - 1;
        kd_mask &= kd_mask - 1;
        if (slot < A.shq_cap) {
            const V3 sd = shadow_dir(S, l, sf.point);
            A.shq.ox[slot] = so.x; A.shq.oy[slot] = so.y; A.shq.oz[slot] = so.z;
            A.shq.dx[slot] = sd.x; A.shq.dy[slot] = sd.y; A.shq.dz[slot] = sd.z;
            A.shq.id[slot] = (tid << 5) | (uint32_t)l;
        }
        slot++;
    }
}

// ---- stage 5: illuminate (js/8-kd_tree.js:716-820) + children (js/8-kd_tree.js:1010-1036) ---------------
__global__ __launch_bounds__(BLOCK) void k_shade_b(LevelArgs A) {
    __shared__ uint32_t s_wave[BLOCK / WAVE + 1];
    const SceneDev &S = A.S;
    const uint32_t tid = blockIdx.x * BLOCK + threadIdx.x;
    const bool in_range = tid < A.n_rays;
    const uint32_t i = in_range ? tid : 0;
    const uint32_t hp = A.cur.hp[i];
    const bool active = in_range && (hp >> 16) != 0xfffeu;
    V3 ro = v3(A.cur.ox[i], A.cur.oy[i], A.cur.oz[i]), rd = v3(A.cur.dx[i], A.cur.dy[i], A.cur.dz[i]);
    Surface sf;
    const bool hit = active && make_surface(S, ro, rd, A.cur.ht[i], hp, A.cur.hf[i], sf);
    bool want_refl = false, want_refr = false;
    V3 c_refl_o = v3(0, 0, 0), c_refl_d = v3(0, 0, 0), c_refr_o = v3(0, 0, 0), c_refr_d = v3(0, 0, 0);
    V3 color = v3(0, 0, 0);
    float kReflect = 0.0f, kRefract = 0.0f, ior = 0.0f;
    if (hit) {
        const uint32_t mask = A.cur.shadow[i];
        float ka = 0.0f, kd = 0.0f, ks = 0.0f, ke = 0.0f;
        int mtype = -1;
        V3 Co = v3(0, 0, 0), Cs = v3(0, 0, 0);
        if (sf.mat >= 0) {
            const MatDev &m = S.mats[sf.mat];
            mtype = m.type;
            ka = m.ka; kd = m.kd; ks = m.ks; ke = m.ke;
            kReflect = m.kReflect; kRefract = m.kRefract; ior = m.ior;
            Cs = ld3(m.Cs);
            if (m.type == RT_MAT_PHONG_CHECKERED) {   // js/8-kd_tree.js:755-776
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
        for (int l = 0; l < S.n_lights; l++) {
            if (mask & (1u << l)) continue;
            V3 Lc = ld3(S.lights[l].col);
            V3 Sd = normalize(ld3(S.lights[l].pos) - sf.point);
            float SdotN = fmaxf(0.0f, dot(Sd, N));
            float dc = clamp01(kd * SdotN);
            diffuse = diffuse + mulv(Lc * dc, Co);
            if (SdotN > 0.0f) {
                float sp = 0.0f;   // stays 0 for the checkered material: no branch assigns it (js/8-kd_tree.js:797-807)
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
        color = v3(0, 0, 0) + ((ambient + specular) + diffuse);
        if (!A.last_level) {   // neither child direction is normalised (js/8-kd_tree.js:1022, :1033)
            bool outside = dot(rd, sf.normal) < 0.0f;
            V3 bias = sf.normal * S.eps;
            if (kRefract > S.eps && ior > S.eps) {
                want_refr = true;
                c_refr_o = outside ? (sf.point - bias) : (sf.point + bias);
                V3 Nn = outside ? sf.normal : -sf.normal;
                float eta = outside ? (1.0f / ior) : ior;
                float cosi = dot(-Nn, rd);
                float k = 1.0f + eta * eta * (cosi * cosi - 1.0f);
                c_refr_d = rd * eta + Nn * (eta * cosi - sqrtf(k));
            }
            if (kReflect > S.eps) {
                want_refl = true;
                c_refl_o = outside ? (sf.point + bias) : (sf.point - bias);
                c_refl_d = rd - sf.normal * (2.0f * dot(sf.normal, rd));
            }
        }
    }
    int32_t slot_refl = -1, slot_refr = -1;
    if (!A.last_level) {
        // children: the refracted ray takes the lane's first slot, the reflected one the next
        uint32_t slot = block_reserve((want_refr ? 1u : 0u) + (want_refl ? 1u : 0u), &A.cnt->next_count, s_wave);
        if (want_refr) {
            if (slot < A.nxt_cap) {
                slot_refr = (int32_t)slot;
                A.nxt.ox[slot] = c_refr_o.x; A.nxt.oy[slot] = c_refr_o.y; A.nxt.oz[slot] = c_refr_o.z;
                A.nxt.dx[slot] = c_refr_d.x; A.nxt.dy[slot] = c_refr_d.y; A.nxt.dz[slot] = c_refr_d.z;
            }
            slot++;
        }
        if (want_refl && slot < A.nxt_cap) {
            slot_refl = (int32_t)slot;
            A.nxt.ox[slot] = c_refl_o.x; A.nxt.oy[slot] = c_refl_o.y; A.nxt.oz[slot] = c_refl_o.z;
            A.nxt.dx[slot] = c_refl_d.x; A.nxt.dy[slot] = c_refl_d.y; A.nxt.dz[slot] = c_refl_d.z;
        }
    }
    {   // shaded nodes of this level (shadow rays = shaded nodes x lights)
        const unsigned long long m = __ballot(hit);
        if (threadIdx.x == 0) s_wave[0] = 0u;
        __syncthreads();
        if (m && lane_id() == 0) atomicAdd(&s_wave[0], (uint32_t)__popcll(m));
        __syncthreads();
        if (threadIdx.x == 0 && s_wave[0]) atomicAdd(&A.cnt->shaded, s_wave[0]);
    }
    if (hit) {
        A.cur.cr[tid] = color.x; A.cur.cg[tid] = color.y; A.cur.cb[tid] = color.z;
        A.cur.kr[tid] = kReflect; A.cur.kt[tid] = kRefract;
        A.cur.refl[tid] = slot_refl; A.cur.refr[tid] = slot_refr;
        if (A.level == 0) {
            int px, py;
            pixel_of(A.g, tid, px, py);
            A.hit_ids[(size_t)py * A.g.W + px] = sf.hit_id;
        }
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

}  // namespace rt
