// Device-side k-d tree builder: the same tree as kd_build.h (KdTree.buildTree, js/8-kd_tree.js:1647-1732), bit for bit,
// built level by level on the GPU and left in device memory in the layout the traversal kernels read. Meant for
// geometry that changes every frame (the reference rebuilds its tree while objects are dragged,
// js/8-kd_tree.js:2494-2522): nothing but the vertices, the faces and their materials crosses PCIe.
// Include from .hip sources only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "kd_build.h"
#include "rt_device.h"

namespace rt {

// result of a build; every pointer is device memory owned by the builder and valid until its next build
struct GpuKdTree {
    KdNodeDev *nodes = nullptr;     // treelet layout (see relayout_treelets in kd_build.cpp), leaves in device form
    uint32_t n_node_slots = 0;
    TriDev *tris = nullptr;         // face order
    TriGeo *lgeo = nullptr;         // leaf order: what a triangle test reads
    TriTag *ltag = nullptr;         // leaf order: what the winner of a ray needs
    uint32_t n_slots = 0;
    uint32_t *path_slot = nullptr;  // 2^PATH_BITS entries
    uint2 *pairs = nullptr;         // (heap index, reference metadata) of every node, unordered
    uint32_t n_nodes = 0;
    int32_t *leaves = nullptr;      // reference leaf table: n_leaves x 4 face indices, -1 = unused
    uint32_t n_leaves = 0;
    double root_box[6] = {0, 0, 0, 0, 0, 0};
    float over[6] = {0, 0, 0, 0, 0, 0};   // SceneDev::over_lo / over_hi
    const float *leaf_box = nullptr; // float32 cell of every leaf with faces (lo xyz, hi xyz), indexed by TriTag::leaf
    const uint32_t *faces = nullptr; // the mesh's vertex indices, 3 per face
    int64_t n_internal = 0, n_leaf = 0, n_empty = 0;
    int depth = 0;
    double build_ms = 0;            // device time of the last build
};

class GpuKdBuilder {
public:
    GpuKdBuilder() = default;
    ~GpuKdBuilder();
    GpuKdBuilder(const GpuKdBuilder &) = delete;
    GpuKdBuilder &operator=(const GpuKdBuilder &) = delete;

    // verts: n_verts*3 doubles; faces: n_faces*3 indices; face_mat: n_faces material slots (-1 = none), all in
    // host memory. Needs n_faces > 4 and max_depth > 1 (smaller inputs have a one-node tree: use the host builder).
    // Returns 0, or -1 with a message in `err`.
    int build(hipStream_t stream, const double *verts, int64_t n_verts, const uint32_t *faces, const int32_t *face_mat,
              int64_t n_faces, int max_depth, double infinity, std::string &err);

    GpuKdTree tree;
    int predicted_misses = 0;       // builds whose predicted sizes did not hold and that were repeated level by level
    bool stalled = false;           // the last build failed because a workgroup's wait ran out (see chained_ranks)

private:
    struct LevelNodes {
        double *blo = nullptr, *bhi = nullptr;   // builder's box, double precision: [3*i + axis]
        float *flo = nullptr, *fhi = nullptr;    // the shader's box: float32 halving sequence
        uint32_t *heap = nullptr, *start = nullptr, *count = nullptr, *pos = nullptr;
    };
    int build_once(hipStream_t stream, const double *verts, int64_t n_verts, const uint32_t *faces, const int32_t *face_mat,
                   int64_t n_faces, int max_depth, double infinity, bool predicted, std::string &err);   // 1 = build again
    int reserve(int64_t n_verts, int64_t n_faces, std::string &err);
    int grow_pool = 1, grow_nodes = 1, grow_leaves = 1;
    int ensure_nodes(hipStream_t s, uint32_t slots, std::string &err);
    void release();

    // inputs
    double *d_verts = nullptr;
    uint32_t *d_faces = nullptr;
    int32_t *d_face_mat = nullptr;
    size_t cap_verts = 0, cap_faces = 0;
    // working set
    LevelNodes lv[2];
    uint32_t cap_level_nodes = 0;
    uint32_t *pool_face[2] = {nullptr, nullptr}, *pool_owner[2] = {nullptr, nullptr};
    uint32_t cap_pool = 0;
    uint32_t *flags = nullptr;                               // cap_pool + 1: which children a face goes on into
    uint2 *scanned = nullptr;                                // cap_pool + 1: faces in front that go on into child 0 / child 1
    uint32_t *owner_map[2] = {nullptr, nullptr};             // 2 * cap_level_nodes + 1: child of a level -> node of the next, or none
    unsigned long long *leaf_bits = nullptr;                 // per block of 64 path codes: mask of the leaves, then their face counts
    unsigned long long *block_scan = nullptr;                // (leaves, faces) in front of each block
    uint32_t *leaf_heap = nullptr, *leaf_count = nullptr, *leaf_pos = nullptr, *leaf_child = nullptr;
    float *leaf_box = nullptr;                               // 6 * cap_leaves: float32 cell of every leaf with faces
    uint32_t *child_faces = nullptr;                         // 4 * cap_pairs: the first faces of every child's list
    uint32_t cap_leaves = 0, cap_pairs = 0;
    uint32_t *counters = nullptr;                            // device: leaves, pairs, empties, root box keys
    uint32_t *h_counters = nullptr;                          // pinned
    uint32_t *lstate = nullptr, *h_lstate = nullptr;         // sizes of every level (device, pinned copy)
    unsigned long long *h_total = nullptr;                   // pinned
    // sizes of the last build: bounds of the next one when the mesh has the same number of faces
    uint32_t hist_pool[26] = {}, hist_active[26] = {}, hist_leaves = 0, hist_used = 0;
    int hist_levels = 0, hist_max_depth = 0;
    int64_t hist_faces = -1;
    uint32_t *chain = nullptr;                               // states of the sums computed inside k_kd_test / k_kd_emit
    size_t cap_chain = 0;
    uint32_t chain_flat = 0;                                 // launches of up to this many workgroups sum in one step
    uint32_t poll_limit = 0;                                 // polls after which a workgroup stops waiting for another's count
    void *scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    uint32_t cap_node_slots = 0, cap_slots = 0, cap_tris = 0, cap_leaf_table = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

}  // namespace rt
