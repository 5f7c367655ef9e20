// Host-side k-d tree builder: the replacement for `new KdTree().buildTree(faceIndices)` +
// `fillDataArray` + the leaf copy of the reference (js/8-kd_tree.js:1647-1732, :1632-1645,
// :2497-2508). Same tree, bit for bit (round-robin axis, spatial-median split, Akenine-Moller
// triangle/box test in double precision with the reference's in-place cross product, leaves cut to
// four faces, leaves numbered in the reference's pop order), but built by a pool of threads over
// independent subtrees and emitted directly in a compact device layout.
#pragma once
#include <cstdint>
#include <vector>

namespace rt {

// Device node, 8 bytes. info bits: [31:30] type (1 internal, 2 empty leaf, 3 leaf);
// internal: [29:28] split axis, [27:0] index of the lower child (the upper child is the next slot),
//           `split` = the plane the reference SHADER computes for this node: the float32 halving
//           sequence cur.min[a] + 0.5f*(cur.max[a]-cur.min[a]) started from the float32-narrowed
//           root box (js/8-kd_tree.js:873, :885), NOT the builder's double-precision plane;
// leaf:     [29:0] leaf index into the leaf table.
struct KdNodeDev {
    float split;
    uint32_t info;
};

struct KdLeafDev {
    int32_t face[4];   // -1 = unused slot (js/8-kd_tree.js:1610-1617)
};

struct KdTreeHost {
    std::vector<KdNodeDev> nodes;     // nodes[0] = root; treelet layout, info == 0 marks an unused slot
    std::vector<uint32_t> heap_index; // reference heap index (2i+1 / 2i+2) of nodes[k]
    std::vector<KdLeafDev> leaves;    // reference leaf order
    double root_box[6];               // root.min, root.max as the builder computed them (f64)
    int depth = 0;                    // deepest node depth, root = 1
    int64_t n_internal = 0, n_leaf = 0, n_empty = 0;
    int64_t n_nodes = 0;              // real nodes (nodes.size() also counts the unused slots)
};

// verts: n_verts*3 doubles; faces: n_faces*3 indices. infinity = the script's INFINITY constant.
// n_threads <= 0 picks min(hardware threads, 16).
void build_kdtree(const double *verts, int64_t n_verts, const uint32_t *faces, int64_t n_faces, int max_depth,
                  double infinity, int n_threads, KdTreeHost &out);

// Overhang of a tree as the walk uses it (SceneDev::over_lo / over_hi in rt_device.h). raw = the largest amounts by which
// the float32 vertices of a listed face reach below (xyz) / above (xyz) the float32 cell of a leaf that lists it; the
// margin for the rounding of the triangle test is 1/256 of the float32 root box's largest side, and the result is
// rounded up (the subtractions behind raw can be half an ulp short).
inline void kd_overhang_with_margin(const float raw[6], const double root_box[6], float out[6]) {
    float side = 0.0f;
    for (int c = 0; c < 3; c++) {
        const float s = (float)root_box[3 + c] - (float)root_box[c];
        if (s > side) side = s;
    }
    const float slack = side * (1.0f / 256.0f);
    for (int c = 0; c < 6; c++) out[c] = (raw[c] + slack) * 1.000001f;
}

// canonical reference arrays: (heapIndex, metadata) pairs ascending by heap index
void export_reference(const KdTreeHost &t, uint32_t *pairs);

}  // namespace rt
