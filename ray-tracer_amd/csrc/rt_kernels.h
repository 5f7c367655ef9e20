// HIP kernels of the frame (gfx950, wave64).
//
// The reference evaluates a per-pixel ray TREE inside one fragment-shader invocation
// (js/8-kd_tree.js:822-1057): slot n spawns a reflected ray into slot 2n+1 and a refracted ray into 2n+2,
// then colours are folded bottom-up. Here the tree is processed one LEVEL at a time over compact ray
// queues, and each level runs as a short pipeline of kernels whose lanes all do the same kind of work:
//
//   k_gen_prims    one lane per ray: (level 0: generate the primary ray) closest hit over the analytic
//                  primitives; rays whose interval overlaps the k-d root box go to the walk queue
//   k_kd_walk      persistent waves walk the tree for the queued rays; the leaves a lane crosses are noted in a
//                  ring in LDS and their triangles tested 64 records at a time by the whole wave (Moller-Trumbore;
//                  closest hit: 64-bit atomic min per ray of (t, visit rank of the leaf, face position); shadow: any
//                  hit sets the light's shadow bit of the ray and ends its walk); one launch serves the closest-hit
//                  rays of a level and the shadow rays of the level before it
//   k_shade_a      one lane per ray: resolves the mesh hit, miss -> background, hit -> the two children into the
//                  next level's queue and one shadow ray per light: analytic any-hit test here, survivors that
//                  reach the k-d box go to the shadow queue
//   k_shade_b      one lane per ray, one step later: Phong terms with the finished shadow bits
//
// Walking and testing can be decoupled because the reference's closest-hit search never stops early: it
// crosses every cell of the root interval and keeps the minimum t (js/8-kd_tree.js:862-962), so which
// leaves are visited does not depend on any triangle test. For shadow rays the reference stops at the first
// hit (js/8-kd_tree.js:685-687); where the walk stops changes no result (any hit anywhere on the ray sets the same bit).
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
    float4 *ray;                         // ray[2i] = origin, ray[2i+1] = direction (w unused): one 32-byte record,
                                         // so that the gathers of the walk and of the triangle tests touch one line
    float *ht;                           // closest hit distance (INFINITY constant = miss so far)
    uint32_t *hp;                        // analytic winner | last analytic hit << 16 (0xffff = none)
    int32_t *hf;                         // winning mesh face or -1
    uint32_t *shadow;                    // bit i: light i is occluded
    unsigned long long *best;            // closest mesh hit: t bits << 32 | visit rank of the leaf << 2 | face position, ~0 = none
    float *cr, *cg, *cb;                 // local colour, later the folded colour
    float *kr, *kt;                      // kReflect / kRefract of this node
    int32_t *refl, *refr;                // queue slot of the child in the next level, -1 = none
};
constexpr int LEVEL_ARRAYS = 21;        // 32-bit words per ray

struct ShadowQ {                         // shadow rays that must walk the k-d tree
    float4 *ray;                         // ray[2q] = origin, w = entry distance of the root box; ray[2q+1] = direction, w = exit distance
    uint32_t *id;                        // ray index << 5 | light
};
constexpr int SHADOWQ_ARRAYS = 9;
constexpr int MAX_LIGHTS = 32;

struct FrameGeom {
    int32_t W, H, tiles_x, n_tiles;      // n_tiles counts the padded enumeration below
    int32_t tiles_y, blocks_x;
    int32_t shard_index, shard_count;   // level-0 entry tid belongs to tile shard_index + (tid >> 6) * shard_count
    int32_t compact_mul, compact_add;   // and to tile (tid >> 6) * compact_mul + compact_add of the shard's compact buffer
};

// device counters of one level; every counter sits in its own 128-byte line so that atomics on different
// counters do not queue behind each other in one L2 channel
struct LevelCounters {
    uint32_t kdq_count, pad0[31];
    uint32_t shq_count, pad2[31];
    uint32_t next_count, pad4[31];
    uint32_t shaded, pad5[31];
    uint32_t rec_count[2], pad6[30];      // leaf records tested by the closest / shadow walk (statistics)
    uint32_t xq[2][8][32];                // per job and eighth of the queue: cursor (one per 128-byte line)
    unsigned long long walk_iters[2], walk_visits[2], walk_t0[2], walk_t1[2], walk_tsum[2], walk_pruned[2], pad7[4];   // RT_WALK_STATS builds only: wave iterations, lane node visits, first start / last end / summed lifetime of the waves (100 MHz ticks)
};

// Lanes per workgroup of the one-lane-per-ray kernels: the largest size; the host picks the size of a launch
// (shade_block) and the kernels read it from blockDim. Every workgroup takes its queue slots with one atomic per counter
// (same-address atomics are served one at a time), which speaks for large workgroups; but all waves of a workgroup meet
// at two barriers, and with 1024 lanes a CU holds a single workgroup of k_gen_prims or k_shade_a at their register
// counts. Measured: 512 lanes (two or three workgroups side by side) best for levels of millions of rays, 256 for the
// short levels of a small shard, where whole workgroups are also too coarse a unit to spread over 256 CUs.
#ifndef RT_BLOCK
#define RT_BLOCK 512
#endif
constexpr int BLOCK = RT_BLOCK;
#ifndef RT_SMALL_BLOCK
#define RT_SMALL_BLOCK 256
#endif
#ifndef RT_SMALL_LEVEL
#define RT_SMALL_LEVEL (2u << 20)
#endif
inline uint32_t shade_block(uint32_t level_capacity) { return level_capacity < RT_SMALL_LEVEL ? (uint32_t)RT_SMALL_BLOCK : (uint32_t)BLOCK; }

struct LevelArgs {
    SceneDev S;             // by value: lands in the kernel-argument segment, read through scalar loads
    LevelBuf cur, nxt;
    ShadowQ shq;
    uint4 *kdq;             // closest-hit traversal queue: (ray index, root-box entry distance, exit distance, -)
    LevelCounters *cnt;
    const uint32_t *n_src;  // levels below 0: where the parent level counted its children (read on the device, so that
                            // the host can queue a whole frame without waiting for any count); null = n_rays is exact
    uint32_t n_rays;        // entries of this level, or with n_src the capacity that bounds the count
    uint32_t nxt_cap, shq_cap;
    int32_t *hit_ids;       // level 0 only, full frame
    FrameGeom g;
    int32_t level, last_level;
};

// Order of the 8x8-pixel tiles. Level 0 lists its rays tile by tile, every deeper level and every queue keeps that
// order, and the walk hands out neighbouring chunks of a queue to one XCD at a time: neighbours in this order should be
// neighbours in the image in BOTH directions, so that they cross the same nodes and triangles. The tiles are therefore
// numbered in blocks of 16 x 16 tiles (128 x 128 pixels), Z-order (Morton) inside a block, the blocks row by row;
// the numbering runs over whole blocks, tiles beyond the image edge exist in it and hold no pixels.
__host__ __device__ __forceinline__ uint32_t even_bits(uint32_t w) {   // bits 0, 2, 4, 6 of w -> a 4-bit number
    w &= 0x55u;
    w = (w | (w >> 1)) & 0x33u;
    return (w | (w >> 2)) & 0x0fu;
}
__host__ __device__ __forceinline__ uint32_t spread_bits(uint32_t v) {   // a 4-bit number -> bits 0, 2, 4, 6
    v = (v | (v << 2)) & 0x33u;
    return (v | (v << 1)) & 0x55u;
}
constexpr int TILE_BLOCK = 16;   // tiles per block side
__host__ __device__ __forceinline__ bool tile_coords(const FrameGeom &g, uint32_t tile, int &tx, int &ty) {
    const uint32_t block = tile >> 8, w = tile & 255u;
    tx = (int)((block % (uint32_t)g.blocks_x) * TILE_BLOCK + even_bits(w));
    ty = (int)((block / (uint32_t)g.blocks_x) * TILE_BLOCK + even_bits(w >> 1));
    return tile < (uint32_t)g.n_tiles && tx < g.tiles_x && ty < g.tiles_y;
}
__host__ __device__ __forceinline__ uint32_t tile_index(const FrameGeom &g, int tx, int ty) {
    const uint32_t block = (uint32_t)(ty / TILE_BLOCK) * (uint32_t)g.blocks_x + (uint32_t)(tx / TILE_BLOCK);
    return (block << 8) | spread_bits((uint32_t)tx & 15u) | (spread_bits((uint32_t)ty & 15u) << 1);
}

__device__ __forceinline__ bool pixel_of(const FrameGeom &g, uint32_t tid, int &x, int &y) {
    const uint32_t tile = (uint32_t)g.shard_index + (tid >> 6) * (uint32_t)g.shard_count;
    int tx, ty;
    if (!tile_coords(g, tile, tx, ty)) return false;
    const uint32_t lane = tid & 63u;
    x = tx * 8 + (int)(lane & 7u);
    y = ty * 8 + (int)(lane >> 3);
    return x < g.W && y < g.H;
}

__device__ __forceinline__ V3 mix3(V3 a, V3 b, float t) {
    float s = 1.0f - t;
    return v3(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t);
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// number of rays of the level a kernel works on (wave-uniform, scalar load)
__device__ __forceinline__ uint32_t level_rays(const uint32_t *n_src, uint32_t bound) {
    return n_src ? min(*n_src, bound) : bound;
}

__device__ __forceinline__ void load_ray(const float4 *ray, uint32_t i, V3 &ro, V3 &rd) {
    const float4 o = ray[2 * (size_t)i], d = ray[2 * (size_t)i + 1];
    ro = v3(o.x, o.y, o.z);
    rd = v3(d.x, d.y, d.z);
}
__device__ __forceinline__ void store_ray(float4 *ray, uint32_t i, V3 ro, V3 rd, float w0 = 0.0f, float w1 = 0.0f) {
    ray[2 * (size_t)i] = make_float4(ro.x, ro.y, ro.z, w0);
    ray[2 * (size_t)i + 1] = make_float4(rd.x, rd.y, rd.z, w1);
}

// Reserves one slot in a global queue for every lane that wants one, with ONE atomic per workgroup (an atomic per
// wave on a single address serialises at roughly 10 ns each and was the floor of these kernels): a ballot gives the
// lane's rank in its wave and the wave's count, the first lane adds up the waves and asks once.
// Must be called by every lane of the workgroup, once per kernel (s_wave is not recycled). Returns the lane's slot.
__device__ __forceinline__ uint32_t block_reserve(bool want, uint32_t *counter, uint32_t *s_wave /* [BLOCK/64 + 1] */) {
    const uint32_t wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(want);
    if (lane_id() == 0) s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        const uint32_t nw = blockDim.x >> 6;
        for (uint32_t w = 0; w < nw; w++) { const uint32_t t = s_wave[w]; s_wave[w] = total; total += t; }
        s_wave[nw] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    return s_wave[blockDim.x >> 6] + s_wave[wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// root-box interval of a ray (js/8-kd_tree.js:545-573); the walk is needed when it is not empty: the reference's
// `while (kdTMax < maxDist)` then runs at least once. The interval travels with the queue entry, so that a walking
// lane picks up a new ray without any arithmetic.
__device__ __forceinline__ bool enters_box(const SceneDev &S, V3 ro, V3 rd, float &tmin, float &tmax) {
    ray_box(S, ro, rd, tmin, tmax);
    return tmin < tmax;
}

// ---- stage 1 -----------------------------------------------------------------------------------
// six waves per SIMD (80 registers, no spills): three workgroups of a CU side by side
__global__ __launch_bounds__(BLOCK, 6) void k_gen_prims(LevelArgs A) {
    __shared__ uint32_t s_wave[BLOCK / WAVE + 1];
    const SceneDev &S = A.S;
    const uint32_t n_rays = level_rays(A.n_src, A.n_rays);
    if (blockIdx.x * blockDim.x >= n_rays) return;   // the grid covers the capacity of the level
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = tid < n_rays;
    V3 ro, rd;
    if (A.level == 0) {
        int px = 0, py = 0;
        active = active && pixel_of(A.g, tid, px, py);
        ro = ld3(S.cam);
        float fx = ((float)px + 0.5f) / (float)A.g.W, fy = ((float)py + 0.5f) / (float)A.g.H;
        V3 dir = mix3(mix3(ld3(S.r00), ld3(S.r01), fy), mix3(ld3(S.r10), ld3(S.r11), fy), fx);
        rd = normalize(dir);
        if (tid < n_rays) store_ray(A.cur.ray, tid, ro, rd);
    } else {
        uint32_t i = active ? tid : 0;
        load_ray(A.cur.ray, i, ro, rd);
    }
    float tMin = S.inf;
    uint32_t best = 0xffffu, last = 0xffffu;
    int32_t face = -1;
    bool to_kd = false;
    float kd_t0 = 0.0f, kd_t1 = 0.0f;
    if (active) {
        // analytic primitives in scene order, strict `t < tMin` (js/8-kd_tree.js:1202-1214)
        for (int i = 0; i < S.n_prims; i++) {
            PHit h;
            if (!surely_misses(S.prims[i], ro, rd) && hit_prim(S, S.prims[i], ro, rd, h)) {
                last = (uint32_t)i;
                if (h.t < tMin) { tMin = h.t; best = (uint32_t)i; }
            }
        }
        if (S.render_triangles) {
            if (S.use_kd) {
                to_kd = enters_box(S, ro, rd, kd_t0, kd_t1);
            } else {
                for (int f = 0; f < S.n_faces; f++) {   // USE_KD_TREE = false: every face (js/8-kd_tree.js:965-988)
                    float t;
                    if (hit_tri(S.tris[f], S.tri_eps, ro, rd, t) && t < tMin) { tMin = t; face = f; }
                }
            }
        }
    }
    if (tid < n_rays) {
        A.cur.ht[tid] = tMin;
        A.cur.hp[tid] = active ? (best | (last << 16)) : 0xfffeffffu;   // 0xfffe in the high half: lane outside the image
        A.cur.hf[tid] = face;
        A.cur.best[tid] = ~0ull;
    }
    const uint32_t slot = block_reserve(to_kd, &A.cnt->kdq_count, s_wave);
    if (to_kd) A.kdq[slot] = make_uint4(tid, __float_as_uint(kd_t0), __float_as_uint(kd_t1), __float_as_uint(tMin));
}

// ---- k-d walk ----------------------------------------------------------------------------------------
// The reference restarts from the "pushed" node for every cell the ray crosses (kd-pushdown,
// js/8-kd_tree.js:862-962). Here the far child and the interval end wait on a per-lane stack instead, which
// visits exactly the same leaves in the same order with the same comparisons: on a restart every ancestor
// decision is re-derived from the same plane distance t, the same kdTMax and a kdTMin that only grew, so it
// repeats, and the first node to decide differently is the one whose far child the stack holds. A NaN plane
// distance (0/0) ends the walk after the current leaf in both forms (`kdTMax < maxDist` is false for NaN).
// The plane distance uses a true IEEE division: a reciprocal multiply can differ in the last bit and flip a
// `t >= kdTMax` decision.
// The stack is a ring of SHORT_STACK entries per lane in LDS; a push onto a full ring overwrites the oldest
// entry. When the ring runs dry although the ray has not reached the end of the root interval, the walk
// restarts at the root with kdTMin = the end of the cell just finished - what the reference does after EVERY
// cell - so a restart re-derives exactly the dropped decisions.
// Every lane owns one ray and takes one tree action per iteration (a descent step, or at a leaf: note its
// triangles and pop); a lane whose ray is finished pulls the next one from the global queue.
//
// Triangle tests. A lane at a non-empty leaf does not test the leaf's faces itself (the other 63 lanes would idle
// meanwhile): it appends a 4-byte record (its lane, the leaf's slot range) to a ring in LDS that the wave shares.
// As soon as the ring holds 64 records, the whole wave stops walking for one round and every lane tests one
// record - the ray of the record's lane comes through ds_bpermute from that lane's registers, the triangles
// from the leaf-ordered table - so the Moller-Trumbore code always runs with full lanes and no record ever
// leaves the compute unit. Closest hit: the shader tests a leaf's faces in order and keeps a hit only if
// `hitInfo.t < tMin` (js/8-kd_tree.js:948-949), i.e. among equal distances the face met FIRST wins; a 64-bit key
// (t bits, visit rank of the leaf for this ray, face position in the leaf) ordered by an atomic min reproduces that
// without any assumption on the order of the tests, and names the winner: rank ^ near_upper_levels(direction) is the
// leaf's path code, SceneDev::path_slot maps it to the leaf's slots. Shadow rays: any hit sets the light's bit of the
// ray, and the lane that owns the ray is told through LDS and drops the ray - the reference's early return
// (js/8-kd_tree.js:685-687); where the walk stops changes no result because every hit sets the same bit.
// A lane whose records are still in the ring keeps its ray registers: it takes no new ray from the queue and no
// stretch from a neighbour until the ring's head has passed its last record.
//
// One launch serves two queues: job 0 = the shadow rays of the previous ray-tree level, job 1 = the closest-hit rays
// of the current one. The two are independent (the children of a level are spawned before its shadow rays are
// walked), and one launch lets the waves that run out of rays of the first queue move straight on to the second
// instead of idling until the slowest ray of the first is done.
struct WalkJob {
    const float4 *rays;        // job 0: shadow queue records (origin + entry distance, direction + exit distance); job 1: ray records of the level
    const uint4 *kdq;          // job 1: (ray index, root-box entry distance, exit distance, -)
    const uint32_t *shq_id;    // job 0: ray index << 5 | light
    uint32_t *shadow;          // job 0: shadow bits of the level the shadow rays belong to
    unsigned long long *best;  // job 1: closest mesh hit keys of the level
    LevelCounters *cnt;        // counters of the job's level
    uint32_t q_cap;
    uint32_t enabled;
};
struct WalkArgs {
    SceneDev S;
    WalkJob job[2];
};

#ifndef RT_REFILL_MIN_IDLE
#define RT_REFILL_MIN_IDLE 24
#endif
#ifndef RT_SHORT_STACK
#define RT_SHORT_STACK 8
#endif
#ifndef RT_WALK_WAVES_PER_CU
#define RT_WALK_WAVES_PER_CU 24
#endif
constexpr int REFILL_MIN_IDLE = RT_REFILL_MIN_IDLE;   // refill once this many lanes of the wave have no ray
#ifndef RT_STEAL_MIN_IDLE
#define RT_STEAL_MIN_IDLE 24
#endif
constexpr int STEAL_MIN_IDLE = RT_STEAL_MIN_IDLE;     // once the queue is dry: idle lanes that make the wave redistribute its rays
#ifndef RT_REFILL_CHUNK
#define RT_REFILL_CHUNK 64
#endif
// a refill hands out up to one ray per lane from a single reservation, so a chunk must cover a whole wave; a
// smaller chunk would let the wave's cursor run past its reservation and the wave would never see the queue end
static_assert(RT_REFILL_CHUNK >= 64, "REFILL_CHUNK must be at least the wave size");
constexpr int REFILL_CHUNK = RT_REFILL_CHUNK;        // rays a wave reserves from the global queue per atomic
constexpr int SHORT_STACK = RT_SHORT_STACK;          // power of two
#ifndef RT_MAIL_SLOTS
#define RT_MAIL_SLOTS 16
#endif
constexpr int MAIL_SLOTS = RT_MAIL_SLOTS;                      // stretches that change hands in one exchange of a wave
// leaf records waiting for their triangle tests: lane << 26 | the low 26 bits of the leaf's info ((faces - 1) << 24 | first slot in S.lgeo).
// The ring never holds more than 63 records when the lanes take their next actions.
constexpr int TEST_RING = 256;
#ifndef RT_WALK_ACTIONS
#define RT_WALK_ACTIONS 3
#endif
// tree actions per lane between two looks at the ring and at the number of idle lanes (the look costs about a sixth of
// an action); a look finds at most 63 records left over and WALK_ACTIONS * 64 new ones
constexpr int WALK_ACTIONS = RT_WALK_ACTIONS;
static_assert(63 + WALK_ACTIONS * 64 <= TEST_RING, "the ring must hold what the actions between two looks can list");
#ifndef RT_TEST_MIN_DRY
#define RT_TEST_MIN_DRY 64
#endif
constexpr uint32_t TEST_MIN_DRY = RT_TEST_MIN_DRY;   // records that trigger a test round once the global queue is dry
#ifndef RT_TEST_MIN_CLOSEST
#define RT_TEST_MIN_CLOSEST 64
#endif
constexpr uint32_t TEST_MIN_CLOSEST = RT_TEST_MIN_CLOSEST;   // the same while the queue still has rays, closest-hit job (a hit found earlier prunes earlier)

// wave-uniform values are passed through readfirstlane so that they live in scalar registers and
// conditions on them compile to scalar branches instead of lane-mask arithmetic
#define UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
// 32-bit population count of a wave mask (a 64-bit count would drag 64-bit compares, which have no scalar form)
#define COUNT_OF(mask) ((uint32_t)__builtin_popcount((uint32_t)(mask)) + (uint32_t)__builtin_popcount((uint32_t)((mask) >> 32)))
// rank of this lane among the set bits of a wave mask
#define RANK_IN(mask) __builtin_amdgcn_mbcnt_hi((uint32_t)((mask) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(mask), 0u))

// Pruning margins of a ray, one per axis: overhang of the listed faces on the side the ray comes from, as a distance
// along the ray. A zero or NaN direction component gives infinity or NaN: such an axis never prunes.
constexpr float PRUNE_KEEP = 1.0f - 1.0f / 256.0f;   // relative part of the safety margin (see SceneDev::over_lo)
__device__ __forceinline__ void prune_margins(const SceneDev &S, float dx, float dy, float dz, float &wx, float &wy, float &wz) {
    wx = (dx > 0.0f ? S.over_lo[0] : S.over_hi[0]) / fabsf(dx);
    wy = (dy > 0.0f ? S.over_lo[1] : S.over_hi[1]) / fabsf(dy);
    wz = (dz > 0.0f ? S.over_lo[2] : S.over_hi[2]) / fabsf(dz);
}

// One round of triangle tests: lanes 0 .. n-1 take the n oldest records of the ring. Every lane of the wave must
// call it (the permutes read the registers of lanes that test nothing themselves). Returns true in the lanes whose
// OWN shadow ray was found occluded in this round.
template <bool ANY>
__device__ __forceinline__ bool test_round(const SceneDev &S, const WalkJob &J, const uint32_t *ring, uint32_t *hit_flag,
                                           uint32_t head, uint32_t n, uint32_t lane, float ox, float oy, float oz, float dx,
                                           float dy, float dz, uint32_t ref) {
    const bool act = lane < n;
    const uint32_t rec = act ? ring[(head + lane) & (uint32_t)(TEST_RING - 1)] : 0u;
    const int src = (int)(rec >> 26);
    const uint32_t first = rec & LEAF_SLOT_MASK, count = ((rec >> LEAF_COUNT_SHIFT) & 3u) + 1u;
    // Lanes beyond n (only when the ring is drained with fewer than 64 records) run the same code on slot 0 and
    // lane 0's ray and drop the result: with no branch around the arithmetic the triangle loads are issued first and
    // the permutes and the other lanes' work run under their latency
    const TriGeo *T = S.lgeo + first;
    const V3 v0 = ld3(T->v0), e1 = ld3(T->e1), e2 = ld3(T->e2);
    const V3 ro = v3(__shfl(ox, src), __shfl(oy, src), __shfl(oz, src));
    const V3 rd = v3(__shfl(dx, src), __shfl(dy, src), __shfl(dz, src));
    const uint32_t rref = (uint32_t)__shfl((int)ref, src);
    unsigned long long key = ~0ull;
    float t;
    if (hit_tri_regs(v0, e1, e2, S.tri_eps, ro, rd, t)) {
        // t >= 0 here, so the float bits order like the values; -0 is folded onto +0
        key = (unsigned long long)((t == 0.0f) ? 0u : __float_as_uint(t)) << 32;
    }
    for (uint32_t k = 1; k < count && !(ANY && key != ~0ull); k++) {
        if (hit_tri(T[k], S.tri_eps, ro, rd, t)) {
            const uint32_t tb = (t == 0.0f) ? 0u : __float_as_uint(t);
            const unsigned long long kk = ((unsigned long long)tb << 32) | (unsigned long long)k;
            key = kk < key ? kk : key;
        }
    }
    if (act && key != ~0ull) {   // few records hit: the rest of the key is made only here
        if (ANY) {
            const uint32_t id = J.shq_id[rref];
            atomicOr(&J.shadow[id >> 5], 1u << (id & 31u));
            hit_flag[src] = 1u;
        } else {
            // low word: visit rank of this leaf for this ray (see path_code), then the face's position in the leaf
            atomicMin(&J.best[rref], key | (unsigned long long)((S.ltag[first].code ^ near_upper_levels(rd)) << 2));
#ifndef RT_NO_PRUNE
            atomicMin(&hit_flag[src], (uint32_t)(key >> 32));   // the owner's nearest distance so far (prunes its walk)
#endif
        }
    }
    bool mine = false;
    if (ANY) {
        __syncthreads();   // one wave per workgroup: orders the flag writes before the reads
        mine = hit_flag[lane] != 0u;
        if (mine) hit_flag[lane] = 0u;
        __syncthreads();
    }
#ifndef RT_NO_PRUNE
    else __syncthreads();   // one wave per workgroup: the minima above land before the owners read their slots
#endif
    return mine;
}

template <bool ANY>
__device__ __forceinline__ void walk_job(const SceneDev &S, const WalkJob &J, uint2 *stack, uint4 *mail, uint32_t *ring,
                                         uint32_t *hit_flag, const uint32_t lane) {
    const KdNodeDev *__restrict__ nodes = S.nodes;
    const uint32_t total = ANY ? min(J.cnt->shq_count, J.q_cap) : J.cnt->kdq_count;
    // The queue is cut into eight contiguous parts, one per XCD (workgroups go round the XCDs by index): the waves of
    // an XCD take the chunks of their own part first, so that each L2 sees the rays of one region of the image, and
    // move on to the other parts when theirs is dry.
    const uint32_t part_len = (((total + 7u) >> 3) + (uint32_t)REFILL_CHUNK - 1u) & ~((uint32_t)REFILL_CHUNK - 1u);
    uint32_t part = blockIdx.x & 7u, parts_seen = 0u;
    uint32_t part_begin = min(part * part_len, total), part_end = min(part_begin + part_len, total);
    uint32_t *head = &J.cnt->xq[ANY ? 1 : 0][part][0];
    // a part with no more chunks than its own XCD has waves is taken in one go by those: nobody else needs to ask
    const uint32_t home_waves = gridDim.x >> 3;
#define XQ_NEXT_PART()                                                                                          \
    do {                                                                                                        \
        last_chunk = 0u;                                                                                        \
        for (;;) {                                                                                              \
            parts_seen++;                                                                                       \
            if (parts_seen >= 8u) { more = 0u; break; }                                                         \
            part = (part + 1u) & 7u;                                                                            \
            part_begin = min(part * part_len, total);                                                           \
            part_end = min(part_begin + part_len, total);                                                       \
            head = &J.cnt->xq[ANY ? 1 : 0][part][0];                                                            \
            if (part_end - part_begin > home_waves * (uint32_t)REFILL_CHUNK) break;                             \
        }                                                                                                       \
    } while (0)
    const KdNodeDev root = nodes[0];

    // wave-uniform: the queue may still hold rays for this wave. A part has ceil(length / chunk) chunks, so at most that
    // many waves of its XCD can ever get one: the waves beyond that number skip the job without touching a cursor
    uint32_t more = (blockIdx.x >> 3) * (uint32_t)REFILL_CHUNK < part_end - part_begin ? 1u : 0u;
    uint32_t last_chunk = 0u;                // wave-uniform: the wave's chunk is the last of the queue
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    float tmin = 0.0f, tmax = 0.0f, maxDist = 0.0f;
    uint32_t ref = 0;
    float tbest = 0.0f, wx = 0.0f, wy = 0.0f, wz = 0.0f;   // closest-hit job: see "pruning" below
    uint32_t info = 0;                       // node the lane stands on; 0 (no node has it) = the lane has no ray
    float split = 0.0f;
    int32_t sp_top = 0, sp_lo = 0;           // ring stack: entries sp_lo .. sp_top-1 are valid (at most SHORT_STACK)
    uint32_t chunk_pos = 0, chunk_end = 0;   // wave-uniform: rays reserved from the global queue
    uint32_t ring_head = 0, ring_tail = 0;   // wave-uniform: records ring_head .. ring_tail-1 wait for their tests (counted on, never wrapped)
    uint32_t my_last = 0;                    // ring position just behind this lane's newest record
    uint32_t n_records = 0;                  // wave-uniform: records tested (statistics)
#ifdef RT_WALK_STATS
    unsigned long long st_iters = 0, st_visits = 0, st_pruned = 0;
    const unsigned long long st_begin = wall_clock64();
#endif

    for (;;) {
        // ---- triangle tests: full rounds as long as 64 records wait; what is left when no lane walks any more
        for (;;) {
            const uint32_t waiting = UNIFORM(ring_tail - ring_head);
            if (waiting == 0u) break;
            if (waiting < (UNIFORM(more) ? (ANY ? (uint32_t)WAVE : TEST_MIN_CLOSEST) : TEST_MIN_DRY) && __builtin_amdgcn_ballot_w64(info != 0u) != 0ull) break;
            const uint32_t n = min(waiting, (uint32_t)WAVE);
            const bool mine = test_round<ANY>(S, J, ring, hit_flag, ring_head, n, lane, ox, oy, oz, dx, dy, dz, ref);
            ring_head = UNIFORM(ring_head + n);
            n_records += n;
            if (ANY && mine) info = 0u;   // occluded: whatever else lies along the ray cannot change the bit
#ifndef RT_NO_PRUNE
            if (!ANY) tbest = __uint_as_float(hit_flag[lane]);
#endif
        }
        // ---- refill: while the global queue has rays, as soon as enough lanes are free. A lane is free when its ray
        //      is finished AND none of its records is still waiting in the ring (those name its ray registers)
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(info == 0u && (int32_t)(my_last - ring_head) <= 0);
        const uint32_t n_idle = COUNT_OF(idle);
        if (n_idle >= (UNIFORM(more) ? (uint32_t)REFILL_MIN_IDLE : (uint32_t)STEAL_MIN_IDLE)) {
            if (!UNIFORM(more)) {
                if (n_idle == (uint32_t)WAVE) break;
#ifndef RT_NO_STEAL
                // ---- the global queue is dry: idle lanes take over the far end of their neighbours' rays. A walking
                //      lane hands the OLDEST entry of its stack - the subtree that covers the last stretch of its ray -
                //      to an idle lane and ends its own ray where that stretch begins. Every entry a lane pushes is
                //      popped and walked sooner or later (a walk only ends early on a NaN plane distance, which needs
                //      a zero direction component: such rays do not give; and on a shadow ray found occluded, where
                //      nothing that is skipped matters), so the same leaves are visited; only who visits them changes,
                //      and the order of the triangle tests does not matter.
                //      If the ring had dropped entries, the taker inherits the old end of the ray and restarts at the
                //      root after its stretch, exactly as the giver would have.
                const bool is_idle = info == 0u && (int32_t)(my_last - ring_head) <= 0;
                const bool can_give = info != 0u && sp_top != sp_lo && dx != 0.0f && dy != 0.0f && dz != 0.0f &&
                                      dx == dx && dy == dy && dz == dz;
                const unsigned long long givers = __builtin_amdgcn_ballot_w64(can_give);
                if (givers != 0ull) {
                    if (can_give) {
                        const uint32_t g = RANK_IN(givers);
                        if (g < min(n_idle, (uint32_t)MAIL_SLOTS)) {
                            const uint2 bot = stack[((uint32_t)sp_lo & (SHORT_STACK - 1)) * WAVE];
                            // an entry's stretch begins where the entry above it ends, the newest one's at the lane's tmax
                            const float start = (sp_top - sp_lo > 1)
                                ? __uint_as_float(stack[((uint32_t)(sp_lo + 1) & (SHORT_STACK - 1)) * WAVE].y) : tmax;
                            mail[g] = make_uint4(bot.x, __float_as_uint(start), bot.y, ref);
                            mail[MAIL_SLOTS + g] = make_uint4(__float_as_uint(maxDist), __float_as_uint(ox), __float_as_uint(oy), __float_as_uint(oz));
                            mail[2 * MAIL_SLOTS + g] = make_uint4(__float_as_uint(dx), __float_as_uint(dy), __float_as_uint(dz), __float_as_uint(tbest));
                            sp_lo += 1;
                            maxDist = start;
                        }
                    }
                    __syncthreads();   // one wave per workgroup: orders the mailbox writes before the reads
                    const uint32_t taker = RANK_IN(idle);
                    if (is_idle && taker < min(COUNT_OF(givers), (uint32_t)MAIL_SLOTS)) {
                        const uint4 m0 = mail[taker], m1 = mail[MAIL_SLOTS + taker], m2 = mail[2 * MAIL_SLOTS + taker];
                        ref = m0.w;
                        maxDist = __uint_as_float(m1.x);
                        ox = __uint_as_float(m1.y); oy = __uint_as_float(m1.z); oz = __uint_as_float(m1.w);
                        dx = __uint_as_float(m2.x); dy = __uint_as_float(m2.y); dz = __uint_as_float(m2.z);
#ifndef RT_NO_PRUNE
                        if (!ANY) {   // the taker goes on with what the giver knows about the ray
                            tbest = __uint_as_float(m2.w);
                            hit_flag[lane] = m2.w;
                            prune_margins(S, dx, dy, dz, wx, wy, wz);
                        }
#endif
                        // the taker stands on a pretended empty leaf that ends where the stretch begins, with the
                        // stretch as the only entry of its stack: the tree action below pops it and fetches the node
                        tmin = tmax = __uint_as_float(m0.y);
                        stack[0] = make_uint2(m0.x, m0.z);
                        sp_lo = 0;
                        sp_top = 1;
                        info = 0x80000000u;
                        split = 0.0f;
                    }
                    __syncthreads();
                }
#endif
            } else {
                const bool is_idle = info == 0u && (int32_t)(my_last - ring_head) <= 0;
                const uint32_t rank = RANK_IN(idle);
                const uint32_t left = chunk_end - chunk_pos;   // rays of the wave's own chunk not handed out yet
                uint32_t q = 0xffffffffu;
                if (left >= n_idle) {
                    q = chunk_pos + rank;
                    chunk_pos += n_idle;
                } else if (UNIFORM(last_chunk)) {
                    // the wave holds the end of this part of the queue: what is left goes out, and no atomic is needed
                    // to know that nothing follows in it
                    if (rank < left) q = chunk_pos + rank;
                    chunk_pos = chunk_end;
                    XQ_NEXT_PART();
                } else {
                    // the rays left in the old chunk are taken first, the rest comes from a new chunk.
                    // Atomics on one address are served one at a time (about 10 ns each), so the ones that can only
                    // fail are avoided: a wave whose index in its XCD is beyond the number of chunks of the XCD's part
                    // never asks (see `more` above), the wave that got the last chunk of a part knows it, and a part that
                    // its own XCD empties in one go is not visited by the others. (A device-scope load of the cursor
                    // before the atomic instead measured twice as slow for the whole walk.)
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(head, (uint32_t)REFILL_CHUNK);
                    base = UNIFORM(base);
                    base += part_begin;
                    const uint32_t qend = part_end;
                    if (base >= qend) {
                        if (rank < left) q = chunk_pos + rank;
                        chunk_pos = chunk_end = 0u;
                        XQ_NEXT_PART();
                    } else {
                        const uint32_t end = min(base + (uint32_t)REFILL_CHUNK, qend);
                        const uint32_t fresh = base + (rank - left);   // lanes served from the old chunk: unused
                        q = rank < left ? chunk_pos + rank : (fresh < end ? fresh : 0xffffffffu);
                        chunk_pos = min(base + (n_idle - left), end);
                        chunk_end = end;
                        last_chunk = base + (uint32_t)REFILL_CHUNK >= qend ? 1u : 0u;
                    }
                }
                if (is_idle && q != 0xffffffffu) {
                    if (ANY) {
                        ref = q;
                        const float4 fo = J.rays[2 * (size_t)q], fd = J.rays[2 * (size_t)q + 1];
                        ox = fo.x; oy = fo.y; oz = fo.z; tmin = fo.w;
                        dx = fd.x; dy = fd.y; dz = fd.z; maxDist = fd.w;
                    } else {
                        const uint4 e = J.kdq[q];
                        ref = e.x;
                        tmin = __uint_as_float(e.y);
                        maxDist = __uint_as_float(e.z);
                        const float4 fo = J.rays[2 * (size_t)ref], fd = J.rays[2 * (size_t)ref + 1];
                        ox = fo.x; oy = fo.y; oz = fo.z;
                        dx = fd.x; dy = fd.y; dz = fd.z;
#ifndef RT_NO_PRUNE
                        // nearest distance known so far: the analytic winner's (a mesh hit must be strictly nearer,
                        // js/8-kd_tree.js:948-949). A negative one (inside a cube) beats every triangle: 0 prunes everything
                        const float seed = __uint_as_float(e.w);
                        tbest = seed > 0.0f ? seed : 0.0f;
                        hit_flag[lane] = __float_as_uint(tbest);
                        prune_margins(S, dx, dy, dz, wx, wy, wz);
#endif
                    }
                    tmax = maxDist;
                    sp_top = sp_lo = 0;
                    split = root.split;
                    info = tmin < maxDist ? root.info : 0u;   // always the root for queued rays; keeps the walk self-contained
                }
                if (__builtin_amdgcn_ballot_w64(info != 0u) == 0ull) continue;
            }
        }

#pragma unroll
        for (int rep = 0; rep < WALK_ACTIONS; rep++) {
            // ---- one tree action per walking lane, written as straight-line predicated code: nested branches cost
            //      a handful of scalar exec-mask instructions each, and a CU issues only about 1.5 instructions per
            //      cycle whatever their kind, so the instruction count of this block is what the walk costs.
            //      info: [31:30] type (1 internal, 2 empty leaf, 3 leaf) - the sign bit says "leaf of either kind"
#ifdef RT_WALK_STATS
            st_iters++;
            st_visits += (unsigned long long)__popcll(__ballot(info != 0u));
#endif
            const bool at_leaf = (int32_t)info < 0;
            const bool is_int = info != 0u && !at_leaf;
            // a lane at a leaf goes on with the newest entry of its stack. The entry is fetched first: the LDS read then
            // runs under the arithmetic of the lanes at internal nodes instead of in front of the node fetch
            const bool cont = at_leaf && tmax < maxDist;   // false on the last cell and on a NaN interval
            const bool pop = cont && sp_top != sp_lo;
            uint2 e = make_uint2(0u, __float_as_uint(maxDist));   // ring ran dry: restart at the root up to maxDist
            if (pop) e = stack[((uint32_t)(sp_top - 1) & (SHORT_STACK - 1)) * WAVE];
#ifndef RT_NO_PRUNE
            // pruning (closest-hit job): the entry is the far child of a node split on axis ea, entered at distance tmax.
            // No face listed below it reaches back across that plane by more than the tree's overhang on that axis, so
            // no hit below it is nearer than tmax - w[ea]; if the ray already holds a hit that near, the subtree is
            // skipped: the lane stands on a pretended empty leaf that ends where the subtree's stretch ends.
            // A restart (ring ran dry) has no plane; every cell still to come is entered across SOME plane beyond tmax.
            bool pruned = false, dead = false;
            if (!ANY) {
                const uint32_t ea = (e.x >> 28) & 3u;
                const float wa = ea == 0u ? wx : (ea == 1u ? wy : wz);
                const float lim = tmax * PRUNE_KEEP;
                pruned = pop && tbest <= lim - wa;
                dead = cont && !pop && tbest <= lim - fmaxf(fmaxf(wx, wy), wz);
            }
#endif
            // both children of an internal node sit in one aligned 16-byte slot pair (a popped node shares such a pair with
            // its sibling): the load is issued before the plane distance is known and the choice is made in registers,
            // which takes the division out of the dependent chain of a ray (what a launch over few rays waits for)
            // (every lane fetches: a lane without a ray or at the end of its ray has e.x = 0 and reads the root's pair)
            const uint4 pair = *reinterpret_cast<const uint4 *>(nodes + (is_int ? (info & 0x0fffffffu) : (e.x & 0x0ffffffeu)));
            // internal node: classify the children and pick the side(s) the ray interval touches
            const uint32_t a = (info >> 28) & 3u;
            const float o = a == 0u ? ox : (a == 1u ? oy : oz), d = a == 0u ? dx : (a == 1u ? dy : dz);
            const float t = (split - o) / d;
            const uint32_t child = info & 0x0fffffffu;
            const bool neg = d <= 0.0f;                              // the near child is the upper one
            const bool ge = t >= tmax, le = t <= tmin;
            const bool both = is_int && !ge && !le;
            // which of the pair the lane goes on with is decided on wave masks: the far child alone = the other one
            const bool upper = cont ? (e.x & 1u) != 0u : ((le && !ge) != neg);
            if (both) stack[((uint32_t)sp_top & (SHORT_STACK - 1)) * WAVE] = make_uint2((child + (neg ? 0u : 1u)) | (a << 28), __float_as_uint(tmax));
            // leaf with faces: note it (below), then go on with the next cell of the ray
            const bool is_listed = info >= 0xc0000000u;
            const unsigned long long listed = __builtin_amdgcn_ballot_w64(is_listed);
            const uint32_t at_info = info;
            sp_top += both ? 1 : (pop ? -1 : 0);                  // a push and a pop never meet in one action
            sp_lo = max(sp_lo, sp_top - SHORT_STACK);             // a push onto a full ring drops the oldest entry
            tmin = cont ? tmax : tmin;
            tmax = cont ? __uint_as_float(e.y) : (both ? t : tmax);
            info = 0u;
            if (is_int || cont) {
                split = __uint_as_float(upper ? pair.z : pair.x);
                info = upper ? pair.w : pair.y;
            }
#ifndef RT_NO_PRUNE
            if (!ANY) {
                if (pruned) info = 0x80000000u;
                if (dead) info = 0u;
#ifdef RT_WALK_STATS
                st_pruned += (unsigned long long)__popcll(__ballot(pruned));
#endif
            }
#endif

            // ---- the leaves just entered: one record per non-empty leaf into the wave's ring (some lane lists in
            //      nearly every action, so there is no branch around this)
            {
                const uint32_t pos = ring_tail + RANK_IN(listed);
                if (is_listed) {
                    ring[pos & (uint32_t)(TEST_RING - 1)] = (lane << 26) | (at_info & LEAF_RECORD_MASK);
                    my_last = pos + 1u;
                }
                ring_tail += (uint32_t)__builtin_popcountll(listed);
            }
        }
    }
    if (lane == 0 && n_records) atomicAdd(&J.cnt->rec_count[ANY ? 1 : 0], n_records);
#ifdef RT_WALK_STATS
    if (lane == 0) {
        atomicAdd(&J.cnt->walk_iters[ANY ? 1 : 0], st_iters);
        atomicAdd(&J.cnt->walk_visits[ANY ? 1 : 0], st_visits);
        atomicAdd(&J.cnt->walk_pruned[ANY ? 1 : 0], st_pruned);
        const unsigned long long st_end = wall_clock64();
        atomicMax(&J.cnt->walk_t0[ANY ? 1 : 0], ~st_begin);   // counters start at zero: the minimum is kept as the maximum of the complement
        atomicMax(&J.cnt->walk_t1[ANY ? 1 : 0], st_end);
        atomicAdd(&J.cnt->walk_tsum[ANY ? 1 : 0], st_end - st_begin);
    }
#endif
}
#undef RANK_IN
#undef COUNT_OF
#undef UNIFORM

__global__ __launch_bounds__(WAVE) void k_kd_walk(WalkArgs A) {
    __shared__ uint2 s_stack[SHORT_STACK * WAVE];
    __shared__ uint4 s_mail[3 * MAIL_SLOTS];      // stretches handed from walking lanes to idle ones
    __shared__ uint32_t s_ring[TEST_RING];        // leaf records waiting for their triangle tests
    __shared__ uint32_t s_hit[WAVE];              // shadow job: lane l's ray was found occluded
    const uint32_t lane = threadIdx.x;
    s_hit[lane] = 0u;
    __syncthreads();
    // the closest-hit job first: its rays are the longer ones (none of them stops early), so the launch ends with the
    // shorter shadow walks (measured: 3 % off the launch, 7 % for a 1/8 shard)
    if (A.job[1].enabled) {
        walk_job<false>(A.S, A.job[1], s_stack + lane, s_mail, s_ring, s_hit, lane);
        s_hit[lane] = 0u;   // the closest-hit job keeps each lane's nearest distance there
        __syncthreads();
    }
    if (A.job[0].enabled) walk_job<true>(A.S, A.job[0], s_stack + lane, s_mail, s_ring, s_hit, lane);
}

// (path code, first slot) of every non-empty leaf -> SceneDev::path_slot
__global__ __launch_bounds__(256) void k_fill_path_slots(const uint2 *pairs, uint32_t n, uint32_t *table) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) table[pairs[i].x] = pairs[i].y;
}

// ---- moved vertices under an unchanged tree (rt_update_vertices) ------------------------------------------------------
// The reference uploads the vertex array every frame and rebuilds the tree at most every 0.30 s
// (js/8-kd_tree.js:2459-2522): in between, its shader fetches fresh vertices through the old leaves. Here the triangle
// records hold vertices, so they are refilled: the face-ordered table ...
__global__ __launch_bounds__(256) void k_refill_tris(TriDev *tris, const uint32_t *faces, const float *verts, uint32_t n_faces) {
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n_faces) return;
    const float *a = verts + 3 * (size_t)faces[3 * (size_t)f], *b = verts + 3 * (size_t)faces[3 * (size_t)f + 1],
                *c = verts + 3 * (size_t)faces[3 * (size_t)f + 2];
    for (int k = 0; k < 3; k++) { tris[f].v0[k] = a[k]; tris[f].e1[k] = b[k] - a[k]; tris[f].e2[k] = c[k] - a[k]; }
}
// ... and the leaf-ordered one, together with the tree's new overhang (SceneDev::over_lo / over_hi): the moved faces may
// reach much farther out of the old cells than before. over_bits: six float maxima as bit patterns (all >= 0), zeroed
// by the caller.
__global__ __launch_bounds__(256) void k_refill_slots(TriGeo *lgeo, const TriTag *ltag, const uint32_t *faces, const float *verts,
                                                      uint32_t n_slots, const float *leaf_box, uint32_t *over_bits) {
    __shared__ uint32_t s_over[6][4];
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    float over[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (s < n_slots) {
        const TriTag tag = ltag[s];
        const float *box = leaf_box + 6 * (size_t)tag.leaf;
        const float *v[3];
        for (int j = 0; j < 3; j++) v[j] = verts + 3 * (size_t)faces[3 * (size_t)tag.face + j];
        TriGeo g;
        for (int k = 0; k < 3; k++) {
            g.v0[k] = v[0][k]; g.e1[k] = v[1][k] - v[0][k]; g.e2[k] = v[2][k] - v[0][k];
            for (int j = 0; j < 3; j++) {
                over[k] = fmaxf(over[k], box[k] - v[j][k]);
                over[3 + k] = fmaxf(over[3 + k], v[j][k] - box[3 + k]);
            }
        }
        lgeo[s] = g;
    }
    for (int c = 0; c < 6; c++) {
        float m = over[c];
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63u) == 0u) s_over[c][threadIdx.x >> 6] = __float_as_uint(m);
    }
    __syncthreads();
    if (threadIdx.x < 6u) {
        const uint32_t *w = s_over[threadIdx.x];
        const uint32_t m = max(max(w[0], w[1]), max(w[2], w[3]));
        if (m) atomicMax(&over_bits[threadIdx.x], m);
    }
}

// ---- surface reconstruction ------------------------------------------------------------------------
struct Surface {
    V3 point, normal, local;
    int32_t mat;
    int32_t hit_id;
};

// Rebuilds what the shader holds in HitInfo after the closest-hit search from the compact hit record.
// Re-running the winning analytic test reproduces its normal and object-space point bit for bit.
// `known`, `known_mat`: the winning triangle's edges and material if the caller holds them already (same values as
// S.tris[face]; spares a gather).
__device__ __forceinline__ bool make_surface(const SceneDev &S, V3 ro, V3 rd, float t, uint32_t hp, int32_t face, Surface &sf,
                                             const TriGeo *known = nullptr, int32_t known_mat = 0) {
    const uint32_t best = hp & 0xffffu, last = hp >> 16;
    if (face < 0 && best == 0xffffu) return false;
    sf.point = ro + rd * t;
    if (face >= 0) {
        V3 e1, e2;
        if (known) {
            e1 = ld3(known->e1); e2 = ld3(known->e2);
            sf.mat = known_mat;
        } else {
            const TriDev &F = S.tris[face];
            e1 = ld3(F.e1); e2 = ld3(F.e2);
            sf.mat = F.mat;
        }
        sf.normal = normalize(cross(e1, e2));
        // intersectTriangle never writes localHitPoint (js/8-kd_tree.js:954): the shader still holds the
        // object-space point of the last analytic test that succeeded for this ray (zero if none did)
        sf.local = v3(0, 0, 0);
        if (last != 0xffffu) {
            PHit h;
            hit_prim(S, S.prims[last], ro, rd, h);
            sf.local = h.loc;
        }
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
    __shared__ uint32_t s_cnt[(3 + MAX_LIGHTS) * (BLOCK / WAVE)];   // per kind of emitted item and wave: count, then exclusive sum
    __shared__ uint32_t s_base[2];                                  // first slot of the workgroup in the child / shadow queue
    const SceneDev &S = A.S;
    const uint32_t n_rays = level_rays(A.n_src, A.n_rays);
    if (blockIdx.x * blockDim.x >= n_rays) return;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = tid < n_rays;
    const uint32_t i = in_range ? tid : 0;
    const uint32_t hp = A.cur.hp[i];
    const bool active = in_range && (hp >> 16) != 0xfffeu;
    V3 ro, rd;
    load_ray(A.cur.ray, i, ro, rd);
    // mesh hit of the closest-hit walk: it replaces the analytic hit only if strictly nearer (js/8-kd_tree.js:948-949)
    float ht = A.cur.ht[i];
    int32_t hf = A.cur.hf[i];
    const TriGeo *winner = nullptr;
    int32_t winner_mat = 0;
    if (active) {
        const unsigned long long key = A.cur.best[i];
        if (key != ~0ull && __uint_as_float((uint32_t)(key >> 32)) < ht) {
            const uint32_t kr = (uint32_t)key;
            const uint32_t slot = S.path_slot[(kr >> 2) ^ near_upper_levels(rd)] + (kr & 3u);
            const TriGeo &T = S.lgeo[slot];
            const TriTag tag = S.ltag[slot];
            hit_tri(T, S.tri_eps, ro, rd, ht);   // the winner's exact t (the key folds -0 onto +0)
            hf = tag.face;
            winner = &T;
            winner_mat = tag.mat;
            A.cur.ht[i] = ht;
            A.cur.hf[i] = hf;
        }
    }
    Surface sf;
    const bool hit = active && make_surface(S, ro, rd, ht, hp, hf, sf, winner, winner_mat);
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
                occluded = !surely_misses(S.prims[p], so, sd) && hit_prim(S, S.prims[p], so, sd, h);
            }
            if (!occluded && S.render_triangles) {
                if (S.use_kd) {
                    float t0, t1;
                    if (enters_box(S, so, sd, t0, t1)) kd_mask |= 1u << l;
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
    // children (js/8-kd_tree.js:1010-1036). They depend on the hit only, not on the shadow rays, so they are spawned
    // here: the next level's closest-hit walk then runs together with this level's shadow walk.
    bool want_refl = false, want_refr = false;
    V3 c_refl_o = v3(0, 0, 0), c_refl_d = v3(0, 0, 0), c_refr_o = v3(0, 0, 0), c_refr_d = v3(0, 0, 0);
    float kReflect = 0.0f, kRefract = 0.0f;
    if (hit && sf.mat >= 0) {
        const MatDev &m = S.mats[sf.mat];
        kReflect = m.kReflect; kRefract = m.kRefract;
        const float ior = m.ior;
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
    // Queue slots for everything this workgroup emits, with two barriers in all: per wave a ballot per kind gives the
    // lane's rank and the wave's count; one lane adds up the waves of the children, another those of the shadow rays,
    // each with ONE atomic on its queue's counter. Kinds: 0 refracted child, 1 reflected child, 2 shaded node (count
    // only), 3 + l shadow ray towards light l that must walk the tree. The refracted rays of the workgroup take one run
    // of slots, the reflected ones the next, then (in the shadow queue) one run per light: neighbours in a queue are
    // rays of one kind from neighbouring pixels, which walk the same part of the tree.
    const uint32_t NW = blockDim.x / WAVE;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t n_kinds = 3u + (uint32_t)S.n_lights;
    const unsigned long long m_refr = __ballot(want_refr), m_refl = __ballot(want_refl), m_hit = __ballot(hit);
    if (lane == 0) {
        s_cnt[0 * NW + wave] = (uint32_t)__popcll(m_refr);
        s_cnt[1 * NW + wave] = (uint32_t)__popcll(m_refl);
        s_cnt[2 * NW + wave] = (uint32_t)__popcll(m_hit);
    }
    for (int l = 0; l < S.n_lights; l++) {
        const unsigned long long m = __ballot(((kd_mask >> l) & 1u) != 0u);
        if (lane == 0) s_cnt[(3u + (uint32_t)l) * NW + wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {             // children: exclusive sums over (kind 0 waves..., kind 1 waves...)
        uint32_t tot = 0;
        for (uint32_t k = 0; k < 2u * NW; k++) { const uint32_t t = s_cnt[k]; s_cnt[k] = tot; tot += t; }
        s_base[0] = tot ? atomicAdd(&A.cnt->next_count, tot) : 0u;
    } else if (threadIdx.x == WAVE) {   // shadow rays, light by light (another wave: runs beside the first)
        uint32_t tot = 0;
        for (uint32_t k = 3u * NW; k < n_kinds * NW; k++) { const uint32_t t = s_cnt[k]; s_cnt[k] = tot; tot += t; }
        s_base[1] = tot ? atomicAdd(&A.cnt->shq_count, tot) : 0u;
    } else if (threadIdx.x == 2 * WAVE) {   // shaded nodes of this level (shadow rays = shaded nodes x lights)
        uint32_t tot = 0;
        for (uint32_t w = 0; w < NW; w++) tot += s_cnt[2 * NW + w];
        if (tot) atomicAdd(&A.cnt->shaded, tot);
    }
    __syncthreads();
    int32_t slot_refl = -1, slot_refr = -1;
    if (want_refr) {
        const uint32_t cs = s_base[0] + s_cnt[0 * NW + wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_refr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_refr, 0u));
        if (cs < A.nxt_cap) {
            slot_refr = (int32_t)cs;
            store_ray(A.nxt.ray, cs, c_refr_o, c_refr_d);
        }
    }
    if (want_refl) {
        const uint32_t cs = s_base[0] + s_cnt[1 * NW + wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_refl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_refl, 0u));
        if (cs < A.nxt_cap) {
            slot_refl = (int32_t)cs;
            store_ray(A.nxt.ray, cs, c_refl_o, c_refl_d);
        }
    }
    if (hit) {
        A.cur.kr[tid] = kReflect; A.cur.kt[tid] = kRefract;
        A.cur.refl[tid] = slot_refl; A.cur.refr[tid] = slot_refr;
        if (A.level == 0) {
            int px, py;
            pixel_of(A.g, tid, px, py);
            A.hit_ids[(size_t)py * A.g.W + px] = sf.hit_id;
        }
    }
    // shadow rays that must walk the tree
    for (int l = 0; l < S.n_lights; l++) {
        const bool mine = ((kd_mask >> l) & 1u) != 0u;
        const unsigned long long m = __ballot(mine);
        const uint32_t slot = s_base[1] + s_cnt[(3u + (uint32_t)l) * NW + wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (mine && slot < A.shq_cap) {
            const V3 sd = shadow_dir(S, l, sf.point);
            float t0, t1;
            ray_box(S, so, sd, t0, t1);
            store_ray(A.shq.ray, slot, so, sd, t0, t1);
            A.shq.id[slot] = (tid << 5) | (uint32_t)l;
        }
    }
}

// ---- illuminate (js/8-kd_tree.js:716-820): the local colour of every hit, with the finished shadow bits ----
__global__ __launch_bounds__(BLOCK) void k_shade_b(LevelArgs A) {
    const SceneDev &S = A.S;
    const uint32_t n_rays = level_rays(A.n_src, A.n_rays);
    if (blockIdx.x * blockDim.x >= n_rays) return;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = tid < n_rays;
    const uint32_t i = in_range ? tid : 0;
    const uint32_t hp = A.cur.hp[i];
    const bool active = in_range && (hp >> 16) != 0xfffeu;
    V3 ro, rd;
    load_ray(A.cur.ray, i, ro, rd);
    Surface sf;
    const bool hit = active && make_surface(S, ro, rd, A.cur.ht[i], hp, A.cur.hf[i], sf);
    V3 color = v3(0, 0, 0);
    if (hit) {
        const uint32_t mask = A.cur.shadow[i];
        float ka = 0.0f, kd = 0.0f, ks = 0.0f, ke = 0.0f;
        int mtype = -1;
        V3 Co = v3(0, 0, 0), Cs = v3(0, 0, 0);
        if (sf.mat >= 0) {
            const MatDev &m = S.mats[sf.mat];
            mtype = m.type;
            ka = m.ka; kd = m.kd; ks = m.ks; ke = m.ke;
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
    }
    if (hit) {
        A.cur.cr[tid] = color.x; A.cur.cg[tid] = color.y; A.cur.cb[tid] = color.z;
    }
}

// parent.color += kRefract * refracted.color, then += kReflect * reflected.color (js/8-kd_tree.js:1041-1054)
__global__ __launch_bounds__(256) void k_composite(LevelBuf cur, LevelBuf nxt, const uint32_t *n_src, uint32_t bound) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= level_rays(n_src, bound)) return;
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
    const size_t ci = ((size_t)(tid >> 6) * (uint32_t)g.compact_mul + (uint32_t)g.compact_add) * 64u + (tid & 63u);
    if (!pixel_of(g, tid, x, y)) { compact[ci] = 0u; return; }
    float r = l0.cr[tid], gg = l0.cg[tid], b = l0.cb[tid];
    size_t p = (size_t)y * g.W + x;
    fb[3 * p] = r; fb[3 * p + 1] = gg; fb[3 * p + 2] = b;
    uint32_t v = quant8(r) | (quant8(gg) << 8) | (quant8(b) << 16) | 0xff000000u;
    rgba[p] = v;
    compact[ci] = v;
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
                                                   double *out_sum, unsigned long long *out_cnt, double *pair) {
    double acc = 0.0;
    unsigned long long cnt = 0;
    for (int i = threadIdx.x; i < n_parts; i += 64) { acc += part_sum[i]; cnt += part_cnt[i]; }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_down(acc, o);
        cnt += __shfl_down(cnt, o);
    }
    if (threadIdx.x == 0) {
        *out_sum = acc; *out_cnt = cnt;
        pair[0] = acc; pair[1] = (double)cnt;   // the form an all-reduce (sum) over ranks takes: exact up to 2^53 pixels
    }
}

// 1.0 when a ray-tree level of this context's last trace produced more children than the next level's buffer holds
// (rt_trace_end will trace that frame again), else 0.0: the third number of the exchange, so that after the all-reduce
// every rank knows whether ANY rank's sums and tiles of this frame are void.
struct OverflowArgs {
    const LevelCounters *cnt[8];   // per part
    uint32_t cap[8][MAX_LEVELS];   // capacity of every level of the part
    int32_t n_parts, depth;
};
__global__ void k_overflow_flag(OverflowArgs a, double *flag) {
    bool over = false;
    for (int p = 0; p < a.n_parts; p++)
        for (int l = 0; l + 1 < a.depth; l++) over = over || a.cnt[p][l].next_count > a.cap[p][l + 1];
    *flag = over ? 1.0 : 0.0;
}

// Lbar = exp(sum / N); sf = ((1.219 + (Ldmax/2)^0.4) / (1.219 + Lbar^0.4))^2.5  (js/8-kd_tree.js:2568-2569), from sums that
// stayed in device memory. tone[0..1] = Lbar, sf as the float uniforms the present shader receives (js/8-kd_tree.js:2579-2581),
// tone_d the same in double precision.
__global__ void k_tone_finish(const double *pair, double L_dmax, float *tone, double *tone_d) {
    const double Lbar = exp(pair[0] / pair[1]);
    const double sf = pow((1.219 + pow(L_dmax * 0.5, 0.4)) / (1.219 + pow(Lbar, 0.4)), 2.5);
    tone[0] = (float)Lbar; tone[1] = (float)sf;
    tone_d[0] = Lbar; tone_d[1] = sf;
}
__global__ void k_set_tone(double Lbar, double sf, float *tone, double *tone_d) {
    tone[0] = (float)Lbar; tone[1] = (float)sf;
    tone_d[0] = Lbar; tone_d[1] = sf;
}

// Present pass, js/8-kd_tree.js:85-143: S x S NEAREST taps (offsets divide by 800*S / 600*S, the WIDTH and
// HEIGHT the reference script holds when it builds the shader string, js/8-kd_tree.js:55,110-111), scale by
// L_MAX, Ward (x sf) or Reinhard, divide by L_DMAX, quantise.
// tone operator and quantisation of the present shader for one output pixel (js/8-kd_tree.js:118-141)
__device__ __forceinline__ uint32_t tone_pixel(const float (&acc)[3], int S, int method, float L_max, float L_dmax, float Akey,
                                               float Lbar, float sf) {
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
    return out;
}

__global__ __launch_bounds__(256) void k_present(const uint32_t *rgba, int W, int H, int S, int method, float L_max,
                                                 float L_dmax, float Akey, const float *tone, uint32_t *canvas) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const float Lbar = tone[0], sf = tone[1];
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
    canvas[(size_t)y * W + x] = tone_pixel(acc, S, method, L_max, L_dmax, Akey, Lbar, sf);
}

// The present pass on a shard's compact tiles, SUPERSAMPLING 1 only: there the shader reads exactly its own pixel of
// the render target (floor(((x + 0.5) / W) * W) = x), so every rank can present its tiles before the gather and
// the root only scatters finished canvas pixels.
__global__ __launch_bounds__(256) void k_present_compact(const uint32_t *compact, uint32_t n, int method, float L_max, float L_dmax,
                                                         float Akey, const float *tone, uint32_t *out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = compact[i];
    float acc[3] = {0.0f, 0.0f, 0.0f};
    acc[0] += (float)(v & 255u) / 255.0f;
    acc[1] += (float)((v >> 8) & 255u) / 255.0f;
    acc[2] += (float)((v >> 16) & 255u) / 255.0f;
    out[i] = tone_pixel(acc, 1, method, L_max, L_dmax, Akey, tone[0], tone[1]);
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
