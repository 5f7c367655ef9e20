// See kd_build_gpu.h. Reference: KdTree.buildTree / faceBoxOverlap, js/8-kd_tree.js:1647-1970.
//
// The reference builds depth-first with an explicit stack. The same tree falls out of a level-by-level build, because
// a node depends on its parent only: its box is the parent's box halved on the axis of its depth
// (js/8-kd_tree.js:1689-1702), its faces are the parent's faces that pass the triangle/box test, in the parent's order
// (js/8-kd_tree.js:1705-1711). What the depth-first order decides is the numbering of the leaves (the reference pops
// the upper child first, js/8-kd_tree.js:1677,:1718); it is restored at the end from the leaves' positions in the tree.
//
// One level = all nodes of one depth that still split ("active"). Their face lists sit back to back in a pool.
//   k_kd_test      one lane per (node, face): the two child boxes, two overlap tests, two flags - and the exclusive
//                  sums over either flag, from which every face's place in its child's list follows: a child's list is
//                  contiguous in the next pool and keeps the parent's order
//   k_kd_next      some workgroups move the faces to their places in the next pool; the others take one lane per
//                  child: size of its list -> splits again / leaf / empty leaf, rank of the children that split = their
//                  index among the next level's active nodes, node records in the treelet layout, boxes of the next
//                  level, leaves to a list
// and at the end the leaves are numbered and their triangles laid out in leaf order (k_kd_leaf_*).
// The exclusive sums of a level are computed inside the kernels that produce their input (chained_ranks below), so a
// level is two dependent launches.
#include "kd_build_gpu.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "kd_sat.h"

namespace rt {
namespace {

constexpr uint32_t T_INTERNAL = 1u << 30, T_EMPTY = 2u << 30, T_LEAF = 3u << 30;
constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t CODES = 1u << PATH_BITS;
enum { C_LEAVES = 0, C_PAIRS = 1, C_EMPTY = 2, C_STALL = 3, C_WORDS = 4 };
// state of one level in device memory: what the level before it produced. The kernels of a level read their sizes
// from here, so that the host can queue all levels without waiting for any of them (grids are sized by bounds).
enum { ST_POOL = 0, ST_ACTIVE = 1, ST_BLOCK = 2, ST_USED = 3, ST_OVER = 4, ST_LEAVES = 5, ST_PAIRS = 6, ST_EMPTY = 7, ST_CHILDREN = 8, ST_WORDS = 12 };
constexpr int MAX_BUILD_LEVELS = 26;

struct Nodes {
    double *blo, *bhi;
    float *flo, *fhi;
    uint32_t *heap, *start, *count, *pos;
};

// doubles as unsigned keys that order like the values (no NaN here)
__device__ __forceinline__ unsigned long long order_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
inline double key_value(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double x;
    memcpy(&x, &b, sizeof x);
    return x;
}
inline unsigned long long host_order_key(double x) {
    unsigned long long b;
    memcpy(&b, &x, sizeof b);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// root box: x from vertex 0, y from vertex 1, z from vertex 2 of each face (js/8-kd_tree.js:1657-1669).
// A few workgroups stride over the faces and end with one atomic pair per coordinate each: atomics on one address are
// served one after the other (about 10 ns each), so one pair per wave of a face-sized grid would cost more than the rest.
constexpr unsigned ROOT_BOX_BLOCKS = 64;
__global__ __launch_bounds__(256) void k_kd_root_box(const double *verts, const uint32_t *faces, uint32_t n_faces,
                                                     unsigned long long *keys /* lo xyz, hi xyz */) {
    __shared__ unsigned long long s_lo[3][4], s_hi[3][4];
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
    for (uint32_t f = blockIdx.x * 256 + threadIdx.x; f < n_faces; f += gridDim.x * 256) {
        for (int k = 0; k < 3; k++) {
            const unsigned long long c = order_key(verts[3 * (size_t)faces[3 * (size_t)f + k] + k]);
            lo[k] = c < lo[k] ? c : lo[k];
            hi[k] = c > hi[k] ? c : hi[k];
        }
    }
    for (int k = 0; k < 3; k++) {
        for (int o = 32; o; o >>= 1) {
            const unsigned long long a = (unsigned long long)__shfl_xor((long long)lo[k], o), b = (unsigned long long)__shfl_xor((long long)hi[k], o);
            lo[k] = a < lo[k] ? a : lo[k];
            hi[k] = b > hi[k] ? b : hi[k];
        }
        if ((threadIdx.x & 63u) == 0u) { s_lo[k][threadIdx.x >> 6] = lo[k]; s_hi[k][threadIdx.x >> 6] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3u) {
        const int k = (int)threadIdx.x;
        unsigned long long a = s_lo[k][0], b = s_hi[k][0];
        for (int w = 1; w < 4; w++) { a = s_lo[k][w] < a ? s_lo[k][w] : a; b = s_hi[k][w] > b ? s_hi[k][w] : b; }
        if (a <= b) { atomicMin(&keys[k], a); atomicMax(&keys[3 + k], b); }
    }
}

// Everything a build needs before its first level, in one launch: the triangles in face order - float32 narrowing of
// the vertices (js/8-kd_tree.js:2462-2464), edges formed in float32 -, the root's face list (every face, in index
// order, owned by node 0), and two work areas cleared.
__global__ __launch_bounds__(256) void k_kd_setup(const double *verts, const uint32_t *faces, const int32_t *face_mat, uint32_t n_faces,
                                                  TriDev *tris, uint32_t *pool_face, uint32_t *pool_owner, unsigned long long *clear_a,
                                                  size_t n_a, uint32_t *clear_b, size_t n_b) {
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    for (size_t i = f; i < n_a; i += (size_t)gridDim.x * 256) clear_a[i] = 0ull;
    for (size_t i = f; i < n_b; i += (size_t)gridDim.x * 256) clear_b[i] = 0u;
    if (f >= n_faces) return;
    float v[3][3];
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) v[j][k] = (float)verts[3 * (size_t)faces[3 * (size_t)f + j] + k];
    TriDev t;
    for (int k = 0; k < 3; k++) { t.v0[k] = v[0][k]; t.e1[k] = v[1][k] - v[0][k]; t.e2[k] = v[2][k] - v[0][k]; }
    t.mat = face_mat[f];
    t.face = (int32_t)f;
    t.pad = 0;
    tris[f] = t;
    pool_face[f] = f;
    pool_owner[f] = 0u;
}

// ---- exclusive sums over K flags per lane across all workgroups of a launch, in a single pass ----
// A launch of T workgroups owns K x (T + ceil(T / 256)) words that are zero when it starts. For each of the K flags,
// own[i] becomes CHAIN_VALID | the number of flags set in workgroup i as soon as that is known (value and mark travel in
// one word, so no fence is needed between them). Up to flat_limit (CHAIN_FLAT) workgroups, every workgroup simply adds
// up the own counts of ALL workgroups in front of it, each lane taking every 256th: T^2 / 2 loads in all, which is nothing
// for a few thousand workgroups, and the only wait is "everybody in front has counted". Larger launches go in two steps:
// group[g] becomes CHAIN_VALID | the number set in workgroups 256 g .. 256 g + 255, published by the last of them
// before it looks at the groups in front, and a workgroup adds the own counts in front of it in its group (one per
// lane) and the totals of the groups in front of that. Either way there is no chain of dependent hand-overs: a value
// takes about as long from one workgroup to another as a small kernel takes in all.
// A workgroup only ever waits for workgroups with a lower index. That those are running or done when it looks leans on
// the order in which the hardware starts workgroups (ascending index, round the XCDs); HIP does not promise that
// order. So a wait that lasts longer than any launch could (poll_limit polls, seconds by default) gives up and says so
// in *stall: no wave ever waits without end, the build is reported as stalled, and the caller falls back to the host
// builder (rt_build_kdtree_device).
constexpr uint32_t CHAIN_VALID = 0x80000000u;
constexpr uint32_t CHAIN_POLL_LIMIT = 1u << 22;   // default of GpuKdBuilder::poll_limit, several seconds: kernels of other streams may hold back a workgroup for a while
constexpr uint32_t CHAIN_FLAT = 2048;
constexpr int CHAIN_SHARED_WORDS_PER_FLAG = 12;

__device__ __forceinline__ uint32_t chain_wait(const uint32_t *word, uint32_t *stall, uint32_t poll_limit) {
    uint32_t v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t polls = 0; !(v & CHAIN_VALID); polls++) {
        if (polls >= poll_limit) { *stall = 1u; break; }
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v & ~CHAIN_VALID;
}
// sums over the 256 lanes of the workgroup, K at a time; slots = 4 K words of shared memory that nothing else uses
// meanwhile (they keep the sums of the four waves)
template <int K>
__device__ __forceinline__ void chain_block_sums(uint32_t (&x)[K], uint32_t *slots) {
    for (int k = 0; k < K; k++) {
        uint32_t v = x[k];
        for (int o = 32; o; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
        if ((threadIdx.x & 63u) == 0u) slots[4 * k + (threadIdx.x >> 6)] = v;
    }
    __syncthreads();
    for (int k = 0; k < K; k++) x[k] = slots[4 * k] + slots[4 * k + 1] + slots[4 * k + 2] + slots[4 * k + 3];
}

// every lane of the 256-lane workgroups tile = 0 .. tiles - 1 calls it (a tile with a higher number must not be dispatched
// before one with a lower number). rank[k] = number of set flags k in front of this lane's, over the whole
// launch; upto[k] = the same including the whole workgroup. sh: K * CHAIN_SHARED_WORDS_PER_FLAG words.
template <int K>
__device__ __forceinline__ void chained_ranks(uint32_t *chain, uint32_t tile, uint32_t tiles, uint32_t flat_limit, const bool (&flag)[K],
                                              uint32_t *sh, uint32_t *stall, uint32_t poll_limit, uint32_t (&rank)[K], uint32_t (&upto)[K]) {
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t stride = tiles + ((tiles + 255u) >> 8);   // words per flag: own counts, then group totals
    uint32_t own[K], part[K];
    for (int k = 0; k < K; k++) own[k] = flag[k] ? 1u : 0u;
    chain_block_sums<K>(own, sh);
    if (threadIdx.x == 0)
        for (int k = 0; k < K; k++)
            __hip_atomic_store(chain + k * stride + tile, CHAIN_VALID | own[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int k = 0; k < K; k++) part[k] = 0u;
    if (tiles <= flat_limit) {
        for (uint32_t i = threadIdx.x; i < tile; i += 256u)
            for (int k = 0; k < K; k++) part[k] += chain_wait(chain + k * stride + i, stall, poll_limit);
    } else {
        const uint32_t group = tile >> 8, first = group << 8;
        if (first + threadIdx.x < tile)
            for (int k = 0; k < K; k++) part[k] = chain_wait(chain + k * stride + first + threadIdx.x, stall, poll_limit);
        if ((tile & 255u) == 255u) {   // the last of its group: the group's totals must not wait for the groups in front
            uint32_t in_group[K];
            for (int k = 0; k < K; k++) in_group[k] = part[k];
            chain_block_sums<K>(in_group, sh + 4 * K);
            if (threadIdx.x == 0)
                for (int k = 0; k < K; k++)
                    __hip_atomic_store(chain + k * stride + tiles + group, CHAIN_VALID | (in_group[k] + own[k]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
        for (uint32_t g = threadIdx.x; g < group; g += 256u)
            for (int k = 0; k < K; k++) part[k] += chain_wait(chain + k * stride + tiles + g, stall, poll_limit);
    }
    chain_block_sums<K>(part, sh + 8 * K);
    for (int k = 0; k < K; k++) {
        const unsigned long long m = __ballot(flag[k]);
        uint32_t r = part[k];
        for (uint32_t w = 0; w < wave; w++) r += sh[4 * k + w];
        rank[k] = r + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        upto[k] = part[k] + own[k];
    }
}

__device__ __forceinline__ uint32_t owner_of(const uint32_t *pool_owner, const uint32_t *owner_map, uint32_t p) {
    const uint32_t raw = pool_owner[p];
    return owner_map ? owner_map[raw] : raw;
}

// One lane per face position p of the pool. flags[p]: bit i = the face goes on into child i of the node that owns it
// (0 for the faces of a leaf's list and for the unused tail of the grid). scanned[p] = (number of faces in front of p that
// go into their child 0, the same for child 1); scanned[bound] = the two totals.
// The next pool holds, node after node in the order of their lists here, first a node's child-0 list and then its
// child-1 list, each in the order of the parent's list. With S0, S1 the two sums, a list [start, start + cnt) therefore
// sends face p to S1[start] + S0[p] (child 0) and to S0[start + cnt] + S1[p] (child 1).
__global__ __launch_bounds__(256) void k_kd_test(const double *verts, const uint32_t *faces, const uint32_t *pool_face,
                                                 const uint32_t *pool_owner, const uint32_t *owner_map, const uint32_t *st,
                                                 uint32_t bound, Nodes cur, int axis, uint32_t *flags, uint2 *scanned,
                                                 uint32_t *chain, uint32_t flat_limit, uint32_t *stall, uint32_t poll_limit) {
    __shared__ uint32_t sh[2 * CHAIN_SHARED_WORDS_PER_FLAG];
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const bool in_grid = p < bound;
    bool into[2] = {false, false};
    if (in_grid) {
        const uint32_t o = p < st[ST_POOL] ? owner_of(pool_owner, owner_map, p) : NONE;
        if (o != NONE) {   // NONE: this face belongs to a leaf
            const uint32_t start = cur.start[o], cnt = cur.count[o];
            // (never false for a consistent level: stay inside the lists)
            if ((unsigned long long)start + cnt <= st[ST_POOL] && p - start < cnt) {
                double lo[3], hi[3];
                for (int k = 0; k < 3; k++) { lo[k] = cur.blo[3 * (size_t)o + k]; hi[k] = cur.bhi[3 * (size_t)o + k]; }
                const uint32_t *f = faces + 3 * (size_t)pool_face[p];
                const double *s0 = verts + 3 * (size_t)f[0], *s1 = verts + 3 * (size_t)f[1], *s2 = verts + 3 * (size_t)f[2];
                // child i spans [min + 0.5*i*ext, min + 0.5*(i+1)*ext] on the split axis (js/8-kd_tree.js:1692-1701)
                const double base = lo[axis], ext = hi[axis] - lo[axis];
                for (int i = 0; i < 2; i++) {
                    lo[axis] = base + 0.5 * (double)i * ext;
                    hi[axis] = base + 0.5 * (double)(i + 1) * ext;
                    into[i] = kd_tri_box_overlap(s0, s1, s2, lo, hi);
                }
            }
        }
        flags[p] = (into[0] ? 1u : 0u) | (into[1] ? 2u : 0u);
    }
    uint32_t rank[2], upto[2];
    chained_ranks<2>(chain, blockIdx.x, gridDim.x, flat_limit, into, sh, stall, poll_limit, rank, upto);
    if (in_grid) scanned[p] = make_uint2(rank[0], rank[1]);
    if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 0u) scanned[bound] = make_uint2(upto[0], upto[1]);
}

// what a level produced = the sizes of the next one, and the lengths of the lists so far. The children of the root and
// of every third level below it open new 16-slot blocks. A list that would run out stops the build here (sizes 0) and
// is named in the flags.
struct Advance {
    const uint2 *pool_total;   // the last element of the exclusive sums over the level's flags
    uint32_t bound_pool, bound_active, cap_pool, cap_nodes, node_cap;
    int next_opens_block;
};
__device__ __forceinline__ void kd_advance(const Advance &a, uint32_t active_total, uint32_t new_leaves, uint32_t new_pairs,
                                           uint32_t new_empty, const uint32_t *st, uint32_t *nxt, uint32_t *counters) {
    const uint32_t bound_pool = a.bound_pool, bound_active = a.bound_active, cap_pool = a.cap_pool, cap_nodes = a.cap_nodes,
                   node_cap = a.node_cap;
    const int next_opens_block = a.next_opens_block;
    const uint2 totals = *a.pool_total;
    uint32_t n_pool = totals.x + totals.y, n_active = active_total, over = st[ST_OVER];
    const uint32_t used = st[ST_USED];
    // this level was larger than the grids it ran on: it was processed in part only, nothing may build on it
    if (st[ST_POOL] > bound_pool || st[ST_ACTIVE] > bound_active) over |= 16u;
    if (n_pool > cap_pool) over |= 1u;
    if (n_active > cap_nodes) over |= 2u;
    uint32_t used_next = used;
    if (next_opens_block) {
        if ((unsigned long long)used + 16ull * n_active > 0x0fffffffull) over |= 8u;
        else used_next = used + 16u * n_active;
    }
    if (used_next > node_cap) over |= 4u;
    if (over) n_pool = n_active = 0u;
    nxt[ST_POOL] = n_pool; nxt[ST_ACTIVE] = n_active; nxt[ST_BLOCK] = used; nxt[ST_USED] = over ? used : used_next; nxt[ST_OVER] = over;
    // the lists of leaves and of (heap index, metadata) pairs grow level by level: the next level appends here
    nxt[ST_LEAVES] = st[ST_LEAVES] + new_leaves; nxt[ST_PAIRS] = st[ST_PAIRS] + new_pairs; nxt[ST_EMPTY] = st[ST_EMPTY] + new_empty;
    nxt[ST_CHILDREN] = st[ST_CHILDREN] + 2u * min(st[ST_ACTIVE], bound_active);
    counters[C_LEAVES] = nxt[ST_LEAVES]; counters[C_PAIRS] = nxt[ST_PAIRS]; counters[C_EMPTY] = nxt[ST_EMPTY];
}

struct EmitOut {
    KdNodeDev *nodes;
    uint32_t node_cap;
    uint2 *pairs;
    uint32_t cap_pairs;
    uint32_t *leaf_heap, *leaf_count, *leaf_pos, *leaf_child;
    float *leaf_box;         // the shader's float32 cell of every leaf with faces: lo xyz, hi xyz
    uint32_t cap_leaves;
    uint32_t *child_faces;   // the first four faces of every child's list, by the child's number over the whole build
    uint32_t cap_children;
    uint32_t *counters;
};

struct LevelIn {
    const uint32_t *pool_face, *pool_owner, *owner_map;   // owner_map: null at the root level
    const uint32_t *flags;
    const uint2 *scanned;
    const uint32_t *st;
    uint32_t bound_pool, bound_active;
    int axis, opens_block, child_depth, max_depth;
};
struct LevelOut {
    uint32_t *next_face, *next_owner, *owner_map;
    uint32_t next_cap, next_cap_nodes;
    uint32_t *st_next;
};

// child c of active node o is child 2 o + c of the level. Its list has n faces and begins at `first` in the next pool.
__device__ __forceinline__ void child_list(const uint2 a /* sums at the node's first face */, const uint2 b /* behind its last */,
                                           uint32_t c, uint32_t &first, uint32_t &n) {
    const uint32_t n0 = b.x - a.x, n1 = b.y - a.y;
    first = a.x + a.y + (c ? n0 : 0u);
    n = c ? n1 : n0;
}
// js/8-kd_tree.js:1683
__device__ __forceinline__ bool splits_again(uint32_t n, int child_depth, int max_depth) { return n > 4u && child_depth < max_depth; }

// faces to their places in the next pool; the first four faces of a list that becomes a leaf are also kept by child
__device__ __forceinline__ void kd_scatter(uint32_t block, const LevelIn &in, const Nodes &cur, const LevelOut &o_, const EmitOut &out) {
    const uint32_t p = block * 256 + threadIdx.x;
    if (p >= min(in.st[ST_POOL], in.bound_pool)) return;
    const uint32_t into = in.flags[p];
    if (into == 0u) return;   // (set only for a face of an active node's list)
    const uint32_t o = owner_of(in.pool_owner, in.owner_map, p);
    const uint32_t start = cur.start[o], cnt = cur.count[o], face = in.pool_face[p];
    const uint2 mine = in.scanned[p], a = in.scanned[start], b = in.scanned[start + cnt];
    for (uint32_t c = 0; c < 2u; c++) {
        if (!(into & (1u << c))) continue;
        uint32_t first, n;
        child_list(a, b, c, first, n);
        const uint32_t r = c ? mine.y - a.y : mine.x - a.x;   // place in the child's list
        const uint32_t dst = first + r;
        if (dst < o_.next_cap) { o_.next_face[dst] = face; o_.next_owner[dst] = 2 * o + c; }
        if (r < 4u && !splits_again(n, in.child_depth, in.max_depth)) {
            const uint32_t child = in.st[ST_CHILDREN] + 2 * o + c;
            if (child < out.cap_children) out.child_faces[4 * (size_t)child + r] = face;
        }
    }
}

// One lane per child: size of its list -> splits again / leaf / empty leaf; node records in the treelet layout, boxes of
// the next level, leaves to a list. tile = number of the workgroup among the `tiles` that do this.
__device__ __forceinline__ void kd_emit(uint32_t tile, uint32_t tiles, const LevelIn &in, const Nodes &cur, const Nodes &nxt, const LevelOut &o_,
                                        const EmitOut &out, const Advance &adv, uint32_t *chain, uint32_t flat_limit, uint32_t poll_limit, uint32_t *sh) {
    const uint32_t *st = in.st;
    const int axis = in.axis;
    const uint32_t t = tile * 256u + threadIdx.x;
    const bool live = t < 2 * min(st[ST_ACTIVE], in.bound_active);
    const uint32_t o = t >> 1, c = t & 1u;
    uint32_t first = 0, n = 0;
    if (live) {
        const uint32_t start = cur.start[o], cnt = cur.count[o];
        if ((unsigned long long)start + cnt <= st[ST_POOL]) child_list(in.scanned[start], in.scanned[start + cnt], c, first, n);
    }
    const bool splits = live && splits_again(n, in.child_depth, in.max_depth);
    // what this child adds to the lists: itself as a node of the next level, a leaf, its parent's record (written by
    // the first child), an empty leaf's record. Its places are its ranks among the lanes of the launch that add the same:
    // no atomics (those on one address are served one after the other, a microsecond per hundred waves)
    const bool adds[4] = {splits, live && !splits && n != 0u, live && c == 0u, live && !splits && n == 0u};
    uint32_t rank[4], upto[4];
    chained_ranks<4>(chain, tile, tiles, flat_limit, adds, sh, &out.counters[C_STALL], poll_limit, rank, upto);
    // the sizes of the next level (the sums of this one are complete in the last workgroup)
    if (tile == tiles - 1u && threadIdx.x == 0u)
        kd_advance(adv, upto[0], upto[1], upto[2] + upto[3], upto[3], st, o_.st_next, out.counters);
    if (!live) return;
    const uint32_t block_base = st[ST_BLOCK];
    const uint32_t pos = cur.pos[o], heap = cur.heap[o];
    // treelet layout (kd_build.cpp: relayout_treelets): a 16-slot block holds a pair, its four children and eight
    // grandchildren; the pairs below the third level open blocks of their own
    const uint32_t kids = in.opens_block ? block_base + 16u * o : (pos & ~15u) + 2u * (pos & 15u) + 2u;
    const float flo = cur.flo[3 * (size_t)o + axis], fhi = cur.fhi[3 * (size_t)o + axis];
    const float split = flo + 0.5f * (fhi - flo);   // the shader's plane (js/8-kd_tree.js:873, :885)
    const uint32_t cpos = kids + c, cheap = 2u * heap + 1u + c;
    // pair records: parents' and empty leaves' share one list, in lane order
    const uint32_t pi_parent = st[ST_PAIRS] + rank[2] + rank[3], pi_empty = pi_parent + (c == 0u ? 1u : 0u);
    if (c == 0u) {
        if (pos < out.node_cap) out.nodes[pos] = KdNodeDev{split, T_INTERNAL | ((uint32_t)axis << 28) | kids};
        if (pi_parent < out.cap_pairs) out.pairs[pi_parent] = make_uint2(heap, T_INTERNAL);
    }
    if (splits) {
        const uint32_t idx = rank[0];
        o_.owner_map[t] = idx;
        if (idx >= o_.next_cap_nodes) return;
        const double base = cur.blo[3 * (size_t)o + axis], ext = cur.bhi[3 * (size_t)o + axis] - base;
        for (int k = 0; k < 3; k++) {
            nxt.blo[3 * (size_t)idx + k] = cur.blo[3 * (size_t)o + k];
            nxt.bhi[3 * (size_t)idx + k] = cur.bhi[3 * (size_t)o + k];
            nxt.flo[3 * (size_t)idx + k] = cur.flo[3 * (size_t)o + k];
            nxt.fhi[3 * (size_t)idx + k] = cur.fhi[3 * (size_t)o + k];
        }
        nxt.blo[3 * (size_t)idx + axis] = base + 0.5 * (double)c * ext;
        nxt.bhi[3 * (size_t)idx + axis] = base + 0.5 * (double)(c + 1u) * ext;
        if (c == 0u) nxt.fhi[3 * (size_t)idx + axis] = split;
        else nxt.flo[3 * (size_t)idx + axis] = split;
        nxt.heap[idx] = cheap;
        nxt.start[idx] = first;
        nxt.count[idx] = n;
        nxt.pos[idx] = cpos;
        return;
    }
    o_.owner_map[t] = NONE;
    if (n == 0u) {
        if (cpos < out.node_cap) out.nodes[cpos] = KdNodeDev{0.0f, T_EMPTY};
        if (pi_empty < out.cap_pairs) out.pairs[pi_empty] = make_uint2(cheap, T_EMPTY);
        return;
    }
    // leaf: the first four faces of the list, the rest is dropped (KdLeaf, js/8-kd_tree.js:1610-1617); the faces
    // themselves are left under the child's number by the lanes that move them (kd_scatter)
    const uint32_t li = st[ST_LEAVES] + rank[1];
    if (li >= out.cap_leaves) return;
    out.leaf_heap[li] = cheap;
    out.leaf_count[li] = n < 4u ? n : 4u;
    out.leaf_pos[li] = cpos;
    out.leaf_child[li] = st[ST_CHILDREN] + t;
    for (int k = 0; k < 3; k++) {
        out.leaf_box[6 * (size_t)li + k] = (k == axis && c == 1u) ? split : cur.flo[3 * (size_t)o + k];
        out.leaf_box[6 * (size_t)li + 3 + k] = (k == axis && c == 0u) ? split : cur.fhi[3 * (size_t)o + k];
    }
}

// The second launch of a level. Moving the faces and making the nodes read the sums over the level's flags and nothing
// of each other: the first workgroups move, the others make nodes.
__global__ __launch_bounds__(256) void k_kd_next(uint32_t scatter_blocks, LevelIn in, Nodes cur, Nodes nxt, LevelOut o_, EmitOut out,
                                                 Advance adv, uint32_t *chain, uint32_t flat_limit, uint32_t poll_limit) {
    __shared__ uint32_t sh[4 * CHAIN_SHARED_WORDS_PER_FLAG];
    if (blockIdx.x < scatter_blocks) kd_scatter(blockIdx.x, in, cur, o_, out);
    else kd_emit(blockIdx.x - scatter_blocks, gridDim.x - scatter_blocks, in, cur, nxt, o_, out, adv, chain, flat_limit, poll_limit, sh);
}

__device__ __forceinline__ uint32_t code_of_heap(uint32_t heap) {
    const uint32_t x = heap + 1u;
    const int depth = 31 - __clz((int)x);
    return (x ^ (1u << depth)) << (PATH_BITS - depth);
}

// Leaves are numbered in the reference's creation order: depth first, upper child first. Of two leaves the one in
// the upper subtree of their common ancestor comes first, i.e. the one with the larger path code (the codes are
// prefix-free). With key = the complement of the code, a leaf's index is the number of leaves with a smaller key and its
// first slot in the leaf-ordered triangle table the number of faces those keep. The 2^PATH_BITS keys are cut into blocks
// of 64: per block a 64-bit mask of the keys that are leaves and 64 two-bit fields with their face counts - 1 (3 MB to
// clear instead of a 64 MB table), an exclusive sum over the blocks' (leaves, faces) totals, and bit counts inside a block.
constexpr uint32_t CODE_BLOCKS = CODES / 64u;

// sum of the 32 two-bit fields of x
__host__ __device__ __forceinline__ uint32_t sum_of_pairs(unsigned long long x) {
    return (uint32_t)__builtin_popcountll(x & 0x5555555555555555ull) + 2u * (uint32_t)__builtin_popcountll(x & 0xaaaaaaaaaaaaaaaaull);
}

__global__ __launch_bounds__(256) void k_kd_leaf_mark(const uint32_t *leaf_heap, const uint32_t *leaf_count, const uint32_t *n_leaves,
                                                      uint32_t bound, unsigned long long *mask, unsigned long long *fields) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(*n_leaves, bound)) return;
    const uint32_t key = ~code_of_heap(leaf_heap[i]) & (CODES - 1u), b = key >> 6, bit = key & 63u;
    atomicOr(&mask[b], 1ull << bit);
    atomicOr(&fields[2 * (size_t)b + (bit >> 5)], (unsigned long long)((leaf_count[i] - 1u) & 3u) << (2u * (bit & 31u)));
}

// input of the exclusive sum: leaves << 32 | faces of a block of keys; one more element (0) so that the last sum is the total
struct LeafBlockTotal {
    const unsigned long long *mask, *fields;
    __host__ __device__ __forceinline__ unsigned long long operator()(uint32_t b) const {
        if (b >= CODE_BLOCKS) return 0ull;
        const uint32_t n = (uint32_t)__builtin_popcountll(mask[b]);
        return ((unsigned long long)n << 32) | (unsigned long long)(n + sum_of_pairs(fields[2 * (size_t)b]) + sum_of_pairs(fields[2 * (size_t)b + 1]));
    }
};
using LeafBlockTotals = hipcub::TransformInputIterator<unsigned long long, LeafBlockTotal, hipcub::CountingInputIterator<uint32_t>>;

__global__ __launch_bounds__(256) void k_kd_leaf_emit(const uint32_t *leaf_heap, const uint32_t *leaf_count, const uint32_t *leaf_pos,
                                                      const uint32_t *leaf_child, const uint32_t *child_faces, uint32_t cap_children,
                                                      const uint32_t *n_leaves, uint32_t bound,
                                                      const unsigned long long *mask, const unsigned long long *fields,
                                                      const unsigned long long *block_scan, const TriDev *tris, uint32_t n_tris, KdNodeDev *nodes,
                                                      uint32_t node_cap, TriGeo *lgeo, TriTag *ltag, uint32_t cap_slots, uint32_t *path_slot,
                                                      int32_t *leaves, uint2 *pairs, uint32_t cap_pairs, uint32_t *counters,
                                                      const double *verts, const uint32_t *faces, const float *leaf_box, uint32_t *over_bits) {
    __shared__ uint32_t s_over[6][4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < min(*n_leaves, bound);
    // Overhang of the tree (SceneDev::over_lo / over_hi): how far the float32 vertices of a listed face reach below /
    // above the leaf's float32 cell. Non-negative floats order like their bits: one atomic maximum per workgroup and
    // component (atomics on one address are served one after the other).
    float over[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
        const uint32_t cnt = leaf_count[i], child = leaf_child[i];
        for (uint32_t k = 0; k < cnt && child < cap_children; k++) {
            const uint32_t f = child_faces[4 * (size_t)child + k];
            if (f >= n_tris) continue;
            for (int j = 0; j < 3; j++) {
                const double *v = verts + 3 * (size_t)faces[3 * (size_t)f + j];
                for (int c = 0; c < 3; c++) {
                    const float x = (float)v[c];
                    over[c] = fmaxf(over[c], leaf_box[6 * (size_t)i + c] - x);
                    over[3 + c] = fmaxf(over[3 + c], x - leaf_box[6 * (size_t)i + 3 + c]);
                }
            }
        }
    }
    for (int c = 0; c < 6; c++) {
        float v = over[c];
        for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        if ((threadIdx.x & 63u) == 0u) s_over[c][threadIdx.x >> 6] = __float_as_uint(v);
    }
    __syncthreads();
    if (threadIdx.x < 6u) {
        const uint32_t *w = s_over[threadIdx.x];
        const uint32_t m = max(max(w[0], w[1]), max(w[2], w[3]));
        if (m) atomicMax(&over_bits[threadIdx.x], m);
    }
    if (!live) return;
    const uint32_t heap = leaf_heap[i], code = code_of_heap(heap), cnt = leaf_count[i], child = leaf_child[i];
    const uint32_t key = ~code & (CODES - 1u), b = key >> 6, bit = key & 63u;
    const unsigned long long before = block_scan[b];
    // leaves with a smaller key in this block, and what their faces add beyond one each
    const uint32_t lower = (uint32_t)__builtin_popcountll(mask[b] & ((1ull << bit) - 1ull));
    const unsigned long long f0 = fields[2 * (size_t)b] & (bit >= 32u ? ~0ull : ((1ull << (2u * bit)) - 1ull));
    const unsigned long long f1 = bit <= 32u ? 0ull : (fields[2 * (size_t)b + 1] & ((1ull << (2u * (bit - 32u))) - 1ull));
    const uint32_t number = (uint32_t)(before >> 32) + lower;
    const uint32_t first = (uint32_t)before + lower + sum_of_pairs(f0) + sum_of_pairs(f1);
    if (leaf_pos[i] < node_cap) nodes[leaf_pos[i]] = KdNodeDev{0.0f, leaf_info(cnt, first)};
    path_slot[code] = first;
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t f = (k < cnt && child < cap_children) ? child_faces[4 * (size_t)child + k] : NONE;
        leaves[4 * (size_t)number + k] = k < cnt ? (int32_t)f : -1;
        if (k < cnt && first + k < cap_slots && f < n_tris) {
            const TriDev t = tris[f];
            TriGeo g;
            for (int c = 0; c < 3; c++) { g.v0[c] = t.v0[c]; g.e1[c] = t.e1[c]; g.e2[c] = t.e2[c]; }
            lgeo[first + k] = g;
            ltag[first + k] = TriTag{t.mat, t.face, code, i};
        }
    }
    // the leaves' pair records follow those of the inner nodes and empty leaves, in leaf order
    const uint32_t pi = counters[C_PAIRS] + number;
    if (pi < cap_pairs) pairs[pi] = make_uint2(heap, T_LEAF | number);
}

__device__ __forceinline__ double device_key_value(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// level 0: the root owns every face, in index order; its children open the block behind the root's own 16-slot line
__global__ void k_kd_init_root(const unsigned long long *box_keys, Nodes cur, uint32_t n_faces, uint32_t *st) {
    for (int k = 0; k < 3; k++) {
        const double lo = device_key_value(box_keys[k]), hi = device_key_value(box_keys[3 + k]);
        cur.blo[k] = lo; cur.bhi[k] = hi;
        cur.flo[k] = (float)lo; cur.fhi[k] = (float)hi;
    }
    cur.heap[0] = 0u; cur.start[0] = 0u; cur.count[0] = n_faces; cur.pos[0] = 0u;
    st[ST_POOL] = n_faces; st[ST_ACTIVE] = 1u; st[ST_BLOCK] = 16u; st[ST_USED] = 32u; st[ST_OVER] = 0u;
    st[ST_LEAVES] = 0u; st[ST_PAIRS] = 0u; st[ST_EMPTY] = 0u; st[ST_CHILDREN] = 0u;
}

#define KCK(call)                                                              \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                \
            err = std::string(#call) + ": " + hipGetErrorString(e_);           \
            return -1;                                                         \
        }                                                                      \
    } while (0)

template <class T>
int grow(T *&p, size_t n, std::string &err) {
    if (p) KCK(hipFree(p));
    p = nullptr;
    KCK(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)));
    return 0;
}

inline dim3 blocks_for(uint64_t n) { return dim3((unsigned)std::max<uint64_t>((n + 255) / 256, 1)); }
// words a launch over n lanes needs for its chained sums (see chained_ranks)
inline size_t chain_words(uint64_t n, int flags) { const size_t t = blocks_for(n).x; return (size_t)flags * (t + (t + 255) / 256); }

}  // namespace

GpuKdBuilder::~GpuKdBuilder() { release(); }

void GpuKdBuilder::release() {
    void *ptrs[] = {d_verts, d_faces, d_face_mat, pool_face[0], pool_face[1], pool_owner[0], pool_owner[1], flags, scanned,
                    owner_map[0], owner_map[1], leaf_bits, block_scan, leaf_heap, leaf_count,
                    leaf_pos, leaf_child, leaf_box, child_faces, counters, lstate, scan_tmp, chain, tree.nodes, tree.tris, tree.lgeo, tree.ltag, tree.path_slot, tree.pairs,
                    tree.leaves};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &l : lv)
        for (void *p : {(void *)l.blo, (void *)l.bhi, (void *)l.flo, (void *)l.fhi, (void *)l.heap, (void *)l.start, (void *)l.count, (void *)l.pos})
            if (p) (void)hipFree(p);
    if (h_counters) (void)hipHostFree(h_counters);
    if (h_lstate) (void)hipHostFree(h_lstate);
    if (h_total) (void)hipHostFree(h_total);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
}

// buffers that depend on the mesh size only; the lists are sized generously and checked after every level
int GpuKdBuilder::reserve(int64_t n_verts, int64_t n_faces, std::string &err) {
    if (!ev0) { KCK(hipEventCreate(&ev0)); KCK(hipEventCreate(&ev1)); }
    if (!counters) {
        KCK(hipMalloc((void **)&counters, (C_WORDS + 18) * sizeof(uint32_t)));   // counters, six 64-bit box keys, six overhang maxima (float bits)
        KCK(hipHostMalloc((void **)&h_counters, (C_WORDS + 18) * sizeof(uint32_t)));
        KCK(hipMalloc((void **)&lstate, (size_t)MAX_BUILD_LEVELS * ST_WORDS * sizeof(uint32_t)));
        KCK(hipHostMalloc((void **)&h_lstate, (size_t)MAX_BUILD_LEVELS * ST_WORDS * sizeof(uint32_t)));
        KCK(hipHostMalloc((void **)&h_total, sizeof(unsigned long long)));
    }
    if ((size_t)n_verts > cap_verts) { if (grow(d_verts, (size_t)n_verts * 3, err)) return -1; cap_verts = (size_t)n_verts; }
    if ((size_t)n_faces > cap_faces) {
        if (grow(d_faces, (size_t)n_faces * 3, err) || grow(d_face_mat, (size_t)n_faces, err)) return -1;
        cap_faces = (size_t)n_faces;
    }
    if ((size_t)n_faces > cap_tris) { if (grow(tree.tris, (size_t)n_faces, err)) return -1; cap_tris = (uint32_t)n_faces; }
    const uint32_t want_pool = (uint32_t)std::min<int64_t>(std::max<int64_t>(n_faces * 16, 1 << 20) * (int64_t)grow_pool, 0x3fffffff);
    if (want_pool > cap_pool) {
        for (int i = 0; i < 2; i++)
            if (grow(pool_face[i], want_pool, err) || grow(pool_owner[i], want_pool, err)) return -1;
        if (grow(flags, (size_t)want_pool + 1, err) || grow(scanned, (size_t)want_pool + 1, err)) return -1;
        cap_pool = want_pool;
    }
    const uint32_t want_nodes = (uint32_t)std::min<int64_t>(std::max<int64_t>(n_faces * 4, 1 << 16) * (int64_t)grow_nodes, 1 << 26);
    if (want_nodes > cap_level_nodes) {
        for (auto &l : lv)
            if (grow(l.blo, 3 * (size_t)want_nodes, err) || grow(l.bhi, 3 * (size_t)want_nodes, err) ||
                grow(l.flo, 3 * (size_t)want_nodes, err) || grow(l.fhi, 3 * (size_t)want_nodes, err) ||
                grow(l.heap, want_nodes, err) || grow(l.start, want_nodes, err) || grow(l.count, want_nodes, err) ||
                grow(l.pos, want_nodes, err))
                return -1;
        const size_t nc = 2 * (size_t)want_nodes + 1;
        if (grow(owner_map[0], nc, err) || grow(owner_map[1], nc, err)) return -1;
        cap_level_nodes = want_nodes;
    }
    const uint32_t want_leaves = (uint32_t)std::min<int64_t>(std::max<int64_t>(n_faces * 8, 1 << 16) * (int64_t)grow_leaves, 1 << 27);
    if (want_leaves > cap_leaves) {
        if (grow(leaf_heap, want_leaves, err) || grow(leaf_count, want_leaves, err) || grow(leaf_pos, want_leaves, err) ||
            grow(leaf_child, want_leaves, err) || grow(leaf_box, 6 * (size_t)want_leaves, err) || grow(tree.leaves, 4 * (size_t)want_leaves, err) ||
            grow(tree.lgeo, 4 * (size_t)want_leaves, err) || grow(tree.ltag, 4 * (size_t)want_leaves, err))
            return -1;
        cap_leaves = want_leaves;
        cap_leaf_table = want_leaves;
        cap_slots = 4 * want_leaves;
    }
    const uint32_t want_pairs = 4 * want_leaves;
    // (every node but the root is a child: the pairs' bound is the children's)
    if (want_pairs > cap_pairs) {
        if (grow(tree.pairs, want_pairs, err) || grow(child_faces, 4 * (size_t)want_pairs, err)) return -1;
        cap_pairs = want_pairs;
    }
    if (!leaf_bits) {
        if (grow(leaf_bits, 3 * (size_t)CODE_BLOCKS, err) || grow(block_scan, (size_t)CODE_BLOCKS + 1, err) || grow(tree.path_slot, (size_t)CODES, err)) return -1;
    }
    // states of the chained sums: two launches per level
    const size_t want_chain = (size_t)MAX_BUILD_LEVELS * (chain_words(cap_pool, 2) + chain_words(2 * (uint64_t)cap_level_nodes, 4));
    if (want_chain > cap_chain) { if (grow(chain, want_chain, err)) return -1; cap_chain = want_chain; }
    // temporary storage of the library scan that numbers the leaves
    size_t need = 0;
    {
        LeafBlockTotals totals(hipcub::CountingInputIterator<uint32_t>(0u), LeafBlockTotal{leaf_bits, leaf_bits + CODE_BLOCKS});
        KCK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, totals, block_scan, (int)(CODE_BLOCKS + 1)));
    }
    if (need > scan_tmp_bytes) {
        if (scan_tmp) KCK(hipFree(scan_tmp));
        scan_tmp = nullptr;
        KCK(hipMalloc(&scan_tmp, need));
        scan_tmp_bytes = need;
    }
    return 0;
}

// the node array grows by whole blocks; what is there already is kept
int GpuKdBuilder::ensure_nodes(hipStream_t s, uint32_t slots, std::string &err) {
    if (slots <= cap_node_slots) return 0;
    const uint32_t want = std::max<uint32_t>(slots + slots / 2, 1u << 16);
    KdNodeDev *nn = nullptr;
    KCK(hipMalloc((void **)&nn, (size_t)want * sizeof(KdNodeDev)));
    if (tree.nodes) {
        KCK(hipMemcpyAsync(nn, tree.nodes, (size_t)cap_node_slots * sizeof(KdNodeDev), hipMemcpyDeviceToDevice, s));
        KCK(hipStreamSynchronize(s));
        KCK(hipFree(tree.nodes));
    }
    tree.nodes = nn;
    cap_node_slots = want;
    return 0;
}

// The lists are sized from the face count; a mesh whose faces spread over many more cells than usual asks for more:
// the list that ran out is made four times as long and the build starts over.
//
// Two ways through a build. The first build of a mesh waits for the sizes of every level (one small read-back per
// level), because nothing is known about the tree. It leaves the sizes behind, and the next build of a mesh with the
// same number of faces - the same mesh moved, which is what the reference rebuilds for (js/8-kd_tree.js:2494-2522) -
// queues all levels at once over grids a quarter larger than last time's, with the true sizes read on the device;
// one read-back at the end tells whether every bound held. If one did not, the build is repeated the first way.
int GpuKdBuilder::build(hipStream_t s, const double *verts, int64_t n_verts, const uint32_t *faces, const int32_t *face_mat,
                        int64_t n_faces, int max_depth, double infinity, std::string &err) {
    // (for tests: the size from which the sums inside the kernels go in two steps, so that a small mesh can take that way)
    if (const char *e = getenv("RT_KD_CHAIN_FLAT")) chain_flat = (uint32_t)strtoul(e, nullptr, 10);
    else chain_flat = CHAIN_FLAT;
    // how long a workgroup waits for the count of one in front of it before the build gives up (see chained_ranks)
    if (const char *e = getenv("RT_KD_CHAIN_POLL_LIMIT")) poll_limit = (uint32_t)strtoul(e, nullptr, 10);
    else poll_limit = CHAIN_POLL_LIMIT;
    stalled = false;
    if (hist_levels > 0 && hist_faces == n_faces && hist_max_depth == max_depth && !getenv("RT_KD_BUILD_SYNC")) {
        const int r = build_once(s, verts, n_verts, faces, face_mat, n_faces, max_depth, infinity, true, err);
        if (r <= 0) return r;
        predicted_misses++;
    }
    for (int attempt = 0; attempt < 24; attempt++) {   // (three lists, each up to 2^20 times its first size in steps of 4)
        const int r = build_once(s, verts, n_verts, faces, face_mat, n_faces, max_depth, infinity, false, err);
        if (r <= 0) return r;
    }
    err = "device k-d build: lists still too small after repeated growth (" + err + ")";
    return -1;
}

// returns 0, -1 (error), 1 (a list ran out and was enlarged, or a predicted bound did not hold: build again)
int GpuKdBuilder::build_once(hipStream_t s, const double *verts, int64_t n_verts, const uint32_t *faces, const int32_t *face_mat,
                             int64_t n_faces, int max_depth, double infinity, bool predicted, std::string &err) {
    if (max_depth > 24) max_depth = 24;
    if (n_faces <= 4 || max_depth <= 1) { err = "device k-d build needs more than four faces and a depth limit above 1"; return -1; }
    if (n_faces > 0x0fffffff) { err = "device k-d build: too many faces"; return -1; }
    if (reserve(n_verts, n_faces, err)) return -1;
    const uint32_t nf = (uint32_t)n_faces;
    auto with_margin = [](uint32_t n, uint32_t extra, uint32_t cap) { return (uint32_t)std::min<uint64_t>((uint64_t)n + n / 4 + extra, cap); };
    if (predicted && ensure_nodes(s, with_margin(hist_used, 4096, 0x0fffffffu), err)) return -1;
    KCK(hipMemcpyAsync(d_verts, verts, (size_t)n_verts * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    KCK(hipMemcpyAsync(d_faces, faces, (size_t)n_faces * 3 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    KCK(hipMemcpyAsync(d_face_mat, face_mat, (size_t)n_faces * sizeof(int32_t), hipMemcpyHostToDevice, s));
    KCK(hipEventRecord(ev0, s));

    // counters, the root box, level 0
    unsigned long long *box_keys = (unsigned long long *)(counters + C_WORDS);
    unsigned long long *hk = (unsigned long long *)(h_counters + C_WORDS);
    Nodes cur{lv[0].blo, lv[0].bhi, lv[0].flo, lv[0].fhi, lv[0].heap, lv[0].start, lv[0].count, lv[0].pos};
    Nodes nxt{lv[1].blo, lv[1].bhi, lv[1].flo, lv[1].fhi, lv[1].heap, lv[1].start, lv[1].count, lv[1].pos};
    memset(h_counters, 0, (C_WORDS + 18) * sizeof(uint32_t));
    for (int k = 0; k < 3; k++) { hk[k] = host_order_key(infinity); hk[3 + k] = host_order_key(-infinity); }
    KCK(hipMemcpyAsync(counters, h_counters, (C_WORDS + 18) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_kd_root_box, dim3(std::min(blocks_for(nf).x, ROOT_BOX_BLOCKS)), dim3(256), 0, s, d_verts, d_faces, nf, box_keys);
    hipLaunchKernelGGL(k_kd_init_root, dim3(1), dim3(1), 0, s, box_keys, cur, nf, lstate);
    // the chained sums of a launch start from zeroed words: all of them at once when the grids are known beforehand
    size_t chain_clear = 0;
    if (predicted)
        for (int d = 0; d < hist_levels; d++)
            chain_clear += chain_words(with_margin(hist_pool[d], 4096, cap_pool), 2) +
                           chain_words(2 * (uint64_t)with_margin(hist_active[d], 256, cap_level_nodes), 4);
    hipLaunchKernelGGL(k_kd_setup, blocks_for(nf), dim3(256), 0, s, d_verts, d_faces, d_face_mat, nf, tree.tris, pool_face[0], pool_owner[0],
                       leaf_bits, 3 * (size_t)CODE_BLOCKS, chain, chain_clear);

    uint32_t n_active = 1, n_pool = nf, used_slots = 32;   // (waiting build only) the root keeps a 16-slot line of its own
    if (predicted) {
        KCK(hipMemsetAsync(tree.nodes, 0, (size_t)cap_node_slots * sizeof(KdNodeDev), s));
    } else {
        if (ensure_nodes(s, used_slots, err)) return -1;
        KCK(hipMemsetAsync(tree.nodes, 0, (size_t)used_slots * sizeof(KdNodeDev), s));
    }
    int side = 0, levels = 0;
    uint32_t grid_pool[MAX_BUILD_LEVELS], grid_active[MAX_BUILD_LEVELS];
    size_t chain_used = 0;
    EmitOut out{tree.nodes, cap_node_slots, tree.pairs, cap_pairs, leaf_heap, leaf_count, leaf_pos, leaf_child, leaf_box, cap_leaves, child_faces, cap_pairs, counters};

    for (int d = 0; predicted ? d < hist_levels : n_active > 0; d++) {   // d = depth of the active nodes, root = 0 (the reference counts from 1)
        if (d >= MAX_BUILD_LEVELS - 1) { err = "device k-d build: too many levels"; return -1; }
        const int axis = d % 3;            // js/8-kd_tree.js:1689-1702: x, y, z from the root down
        // the nodes a level makes get their numbers in one table while the faces still find their owners in the other
        const uint32_t *omap = d == 0 ? nullptr : owner_map[d & 1];
        const uint32_t gp = predicted ? with_margin(hist_pool[d], 4096, cap_pool) : n_pool;
        const uint32_t ga = predicted ? with_margin(hist_active[d], 256, cap_level_nodes) : n_active;
        grid_pool[d] = gp; grid_active[d] = ga;
        uint32_t *st = lstate + (size_t)d * ST_WORDS;
        uint32_t *chain_test = chain + chain_used, *chain_emit = chain_test + chain_words(gp, 2);
        chain_used += chain_words(gp, 2) + chain_words(2 * (uint64_t)ga, 4);
        if (!predicted) KCK(hipMemsetAsync(chain_test, 0, (size_t)(chain + chain_used - chain_test) * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_kd_test, blocks_for(gp), dim3(256), 0, s, d_verts, d_faces, pool_face[side], pool_owner[side],
                           omap, st, gp, cur, axis, flags, scanned, chain_test, chain_flat, counters + C_STALL, poll_limit);
        const unsigned sb = blocks_for(gp).x;
        // the children of the root and of every third level below it open new 16-slot blocks
        const bool opens_block = (d % 3 == 0);
        hipLaunchKernelGGL(k_kd_next, dim3(sb + blocks_for(2 * (uint64_t)ga).x), dim3(256), 0, s, sb,
                           LevelIn{pool_face[side], pool_owner[side], omap, flags, scanned, st, gp, ga, axis, opens_block ? 1 : 0, d + 2, max_depth},
                           cur, nxt,
                           LevelOut{pool_face[side ^ 1], pool_owner[side ^ 1], owner_map[(d + 1) & 1], cap_pool, cap_level_nodes, st + ST_WORDS},
                           out,
                           Advance{scanned + gp, gp, ga, cap_pool, cap_level_nodes,
                                   predicted ? cap_node_slots : 0xffffffffu, ((d + 1) % 3 == 0) ? 1 : 0},
                           chain_emit, chain_flat, poll_limit);
        KCK(hipGetLastError());
        levels = d + 1;
        if (!predicted) {
            KCK(hipMemcpyAsync(h_lstate + (size_t)(d + 1) * ST_WORDS, st + ST_WORDS, ST_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            KCK(hipStreamSynchronize(s));
            const uint32_t *hn = h_lstate + (size_t)(d + 1) * ST_WORDS;
            if (hn[ST_OVER] & 1u) { err = "face pool too small"; grow_pool = std::min(grow_pool * 4, 1 << 20); return 1; }
            if (hn[ST_OVER] & 2u) { err = "level node arrays too small"; grow_nodes = std::min(grow_nodes * 4, 1 << 20); return 1; }
            if (hn[ST_OVER] & 8u) { err = "device k-d build: node array too large"; return -1; }
            hist_pool[d] = n_pool; hist_active[d] = n_active;
            n_pool = hn[ST_POOL];
            n_active = hn[ST_ACTIVE];
            if (hn[ST_USED] > used_slots) {   // the blocks the next level's children will fill
                const uint32_t before = used_slots;
                used_slots = hn[ST_USED];
                if (ensure_nodes(s, used_slots, err)) return -1;
                KCK(hipMemsetAsync(tree.nodes + before, 0, (size_t)(used_slots - before) * sizeof(KdNodeDev), s));
                out.nodes = tree.nodes;
                out.node_cap = cap_node_slots;
            }
        }
        side ^= 1;
        std::swap(cur, nxt);
    }

    // leaves: numbering, leaf-ordered triangles, path table, reference leaf table
    uint32_t grid_leaves = 0;
    if (predicted) {
        grid_leaves = with_margin(hist_leaves, 1024, cap_leaves);
    } else {
        KCK(hipMemcpyAsync(h_counters, counters, C_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        KCK(hipStreamSynchronize(s));
        grid_leaves = h_counters[C_LEAVES];
        if (grid_leaves > cap_leaves) { err = "leaf list too small"; grow_leaves = std::min(grow_leaves * 4, 1 << 20); return 1; }
    }
    unsigned long long *leaf_mask = leaf_bits, *leaf_fields = leaf_bits + CODE_BLOCKS;   // (cleared by k_kd_setup)
    hipLaunchKernelGGL(k_kd_leaf_mark, blocks_for(grid_leaves), dim3(256), 0, s, leaf_heap, leaf_count, counters + C_LEAVES, grid_leaves,
                       leaf_mask, leaf_fields);
    size_t tmp = scan_tmp_bytes;
    LeafBlockTotals totals(hipcub::CountingInputIterator<uint32_t>(0u), LeafBlockTotal{leaf_mask, leaf_fields});
    KCK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, tmp, totals, block_scan, (int)(CODE_BLOCKS + 1), s));
    hipLaunchKernelGGL(k_kd_leaf_emit, blocks_for(grid_leaves), dim3(256), 0, s, leaf_heap, leaf_count, leaf_pos, leaf_child, child_faces, cap_pairs,
                       counters + C_LEAVES, grid_leaves, leaf_mask, leaf_fields, block_scan, tree.tris, nf, tree.nodes, cap_node_slots, tree.lgeo, tree.ltag, cap_slots,
                       tree.path_slot, tree.leaves, tree.pairs, cap_pairs, counters, d_verts, d_faces, leaf_box, counters + C_WORDS + 12);
    KCK(hipGetLastError());
    KCK(hipMemcpyAsync(h_total, block_scan + CODE_BLOCKS, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    KCK(hipMemcpyAsync(h_counters, counters, (C_WORDS + 18) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    KCK(hipMemcpyAsync(h_lstate, lstate, (size_t)(levels + 1) * ST_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    KCK(hipEventRecord(ev1, s));
    KCK(hipStreamSynchronize(s));
    const uint32_t n_leaves = h_counters[C_LEAVES];
    if (h_counters[C_STALL]) {
        stalled = true;
        err = "device k-d build: a workgroup waited for the flag count of another that never reported";
        return -1;
    }
    if (predicted) {
        // did every bound hold? A level larger than its grid was only partly processed: everything after it is void
        const uint32_t *last = h_lstate + (size_t)levels * ST_WORDS;
        bool ok = last[ST_OVER] == 0u && last[ST_ACTIVE] == 0u && n_leaves <= grid_leaves;
        for (int d = 0; d < levels && ok; d++) {
            const uint32_t *hs = h_lstate + (size_t)d * ST_WORDS;
            ok = hs[ST_POOL] <= grid_pool[d] && hs[ST_ACTIVE] <= grid_active[d];
        }
        if (!ok) { err = "a predicted size did not hold"; return 1; }
    }
    if (n_leaves > cap_leaves) { err = "leaf list too small"; grow_leaves = std::min(grow_leaves * 4, 1 << 20); return 1; }
    const unsigned long long total = *h_total;
    if ((uint32_t)(total >> 32) != n_leaves) { err = "device k-d build: two leaves share a path code"; return -1; }
    // (the leaves' records follow those of the inner nodes and empty leaves)
    if ((uint64_t)h_counters[C_PAIRS] + n_leaves > cap_pairs) { err = "node list too small"; grow_leaves = std::min(grow_leaves * 4, 1 << 20); return 1; }
    for (int k = 0; k < 6; k++) tree.root_box[k] = key_value(hk[k]);
    tree.leaf_box = leaf_box;
    tree.faces = d_faces;
    {
        float raw[6];
        memcpy(raw, h_counters + C_WORDS + 12, sizeof raw);
        kd_overhang_with_margin(raw, tree.root_box, tree.over);
    }
    tree.n_internal = 0;
    int real_levels = 0;   // a tree that got shallower leaves the last predicted levels without nodes
    for (int d = 0; d < levels; d++) {
        const uint32_t *hs = h_lstate + (size_t)d * ST_WORDS;
        tree.n_internal += hs[ST_ACTIVE];
        hist_pool[d] = hs[ST_POOL]; hist_active[d] = hs[ST_ACTIVE];
        if (hs[ST_ACTIVE] > 0u) real_levels = d + 1;
    }
    levels = real_levels;
    tree.depth = levels + 1;
    hist_levels = levels; hist_faces = n_faces; hist_max_depth = max_depth; hist_leaves = n_leaves;
    hist_used = h_lstate[(size_t)levels * ST_WORDS + ST_USED];
    tree.n_slots = (uint32_t)total;
    tree.n_leaves = n_leaves;
    tree.n_leaf = n_leaves;
    tree.n_empty = h_counters[C_EMPTY];
    tree.n_nodes = h_counters[C_PAIRS] + n_leaves;
    tree.n_node_slots = hist_used;
    float ms = 0;
    KCK(hipEventElapsedTime(&ms, ev0, ev1));
    tree.build_ms = ms;
    return 0;
}

}  // namespace rt
