// See kd_build.h. Reference: KdTree.buildTree / faceBoxOverlap / axisTest_* / planeBoxOverlap,
// js/8-kd_tree.js:1647-1970.
#include "kd_build.h"
#include "kd_sat.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <utility>

namespace rt {
void relayout_treelets(KdTreeHost &t);
namespace {

struct BoxD { double lo[3], hi[3]; };
struct BoxF { float lo[3], hi[3]; };
struct MeshRef { const double *v; const uint32_t *f; };

// Triangle/box overlap: kd_sat.h (shared with the device builder)
static bool tri_box_overlap(const MeshRef &m, const BoxD &b, uint32_t face) {
    const uint32_t *f = m.f + 3 * (size_t)face;
    return kd_tri_box_overlap(m.v + 3 * (size_t)f[0], m.v + 3 * (size_t)f[1], m.v + 3 * (size_t)f[2], b.lo, b.hi);
}

constexpr uint32_t T_INTERNAL = 1u << 30, T_EMPTY = 2u << 30, T_LEAF = 3u << 30;

struct Arena {
    std::vector<KdNodeDev> nodes;
    std::vector<uint32_t> heap;
    std::vector<KdLeafDev> leaves;
    int depth = 0;
    int64_t n_internal = 0, n_leaf = 0, n_empty = 0;
};

struct Task {
    uint32_t top_slot;  // slot of the subtree root in the top arena
    uint32_t heap;
    int depth;
    BoxD bd;
    BoxF bf;
    std::vector<uint32_t> faces;
    Arena arena;        // nodes[0] = subtree root, then child pairs
};

struct Builder {
    MeshRef mesh;
    int max_depth;
    int split_depth;            // subtrees rooted at this depth become tasks (0 = never)
    std::vector<Task> *tasks;   // only the top pass appends
    // DFS (upper child first) event order of the top pass: >= 0 top-arena leaf number, < 0 task ~id
    std::vector<int64_t> *events;

    void node(Arena &A, uint32_t slot, uint32_t heap, int depth, const BoxD &bd, const BoxF &bf,
              std::vector<uint32_t> faces, bool top) {
        A.heap[slot] = heap;
        if (depth > A.depth) A.depth = depth;
        if (faces.size() > 4 && depth < max_depth) {
            if (top && split_depth > 0 && depth == split_depth) {
                events->push_back(~(int64_t)tasks->size());
                tasks->push_back(Task{slot, heap, depth, bd, bf, std::move(faces), Arena()});
                return;
            }
            const int a = (depth + 2) % 3;   // depth%3: 1 -> x, 2 -> y, 0 -> z (js/8-kd_tree.js:1689-1702)
            BoxD cb[2] = {bd, bd};
            BoxF cf[2] = {bf, bf};
            const double ext = bd.hi[a] - bd.lo[a];
            for (int i = 0; i < 2; i++) {
                cb[i].lo[a] = bd.lo[a] + 0.5 * i * ext;
                cb[i].hi[a] = bd.lo[a] + 0.5 * (i + 1) * ext;
            }
            const float split = bf.lo[a] + 0.5f * (bf.hi[a] - bf.lo[a]);
            cf[0].hi[a] = split;
            cf[1].lo[a] = split;
            std::vector<uint32_t> cfaces[2];
            for (int i = 0; i < 2; i++) {
                cfaces[i].reserve(faces.size());
                for (uint32_t f : faces)
                    if (tri_box_overlap(mesh, cb[i], f)) cfaces[i].push_back(f);
            }
            std::vector<uint32_t>().swap(faces);
            const uint32_t base = (uint32_t)A.nodes.size();
            A.nodes.resize(base + 2);
            A.heap.resize(base + 2);
            A.nodes[slot].split = split;
            A.nodes[slot].info = T_INTERNAL | ((uint32_t)a << 28) | base;
            A.n_internal++;
            // the reference pushes child 0 then child 1 and pops child 1 first (js/8-kd_tree.js:1677,:1718):
            // leaves are numbered upper subtree first
            node(A, base + 1, 2 * heap + 2, depth + 1, cb[1], cf[1], std::move(cfaces[1]), top);
            node(A, base + 0, 2 * heap + 1, depth + 1, cb[0], cf[0], std::move(cfaces[0]), top);
            return;
        }
        A.nodes[slot].split = 0.0f;
        if (faces.empty()) {
            A.nodes[slot].info = T_EMPTY;
            A.n_empty++;
            return;
        }
        KdLeafDev lf = {{-1, -1, -1, -1}};
        for (size_t i = 0; i < std::min<size_t>(faces.size(), 4); i++) lf.face[i] = (int32_t)faces[i];
        if (top) events->push_back((int64_t)A.leaves.size());
        A.nodes[slot].info = T_LEAF | (uint32_t)A.leaves.size();
        A.leaves.push_back(lf);
        A.n_leaf++;
    }
};

}  // namespace

void build_kdtree(const double *verts, int64_t n_verts, const uint32_t *faces, int64_t n_faces, int max_depth,
                  double infinity, int n_threads, KdTreeHost &out) {
    (void)n_verts;
    if (max_depth > 24) max_depth = 24;
    if (n_threads <= 0) n_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);

    // root box: x from vertex 0, y from vertex 1, z from vertex 2 of each face (js/8-kd_tree.js:1657-1669)
    BoxD rb;
    for (int k = 0; k < 3; k++) { rb.lo[k] = infinity; rb.hi[k] = -infinity; }
    for (int64_t i = 0; i < n_faces; i++) {
        for (int k = 0; k < 3; k++) {
            double c = verts[3 * (size_t)faces[3 * i + k] + k];
            rb.lo[k] = std::fmin(rb.lo[k], c);
            rb.hi[k] = std::fmax(rb.hi[k], c);
        }
    }
    BoxF rf;
    for (int k = 0; k < 3; k++) { rf.lo[k] = (float)rb.lo[k]; rf.hi[k] = (float)rb.hi[k]; }

    std::vector<Task> tasks;
    std::vector<int64_t> events;
    Builder B{{verts, faces}, max_depth, (n_threads > 1 && n_faces > 2048) ? 8 : 0, &tasks, &events};
    Arena top;
    top.nodes.resize(1);
    top.heap.resize(1);
    std::vector<uint32_t> all((size_t)n_faces);
    for (int64_t i = 0; i < n_faces; i++) all[(size_t)i] = (uint32_t)i;
    B.node(top, 0, 0, 1, rb, rf, std::move(all), true);

    // subtrees in parallel, largest first
    if (!tasks.empty()) {
        std::vector<size_t> order(tasks.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(),
                  [&](size_t a, size_t b) { return tasks[a].faces.size() > tasks[b].faces.size(); });
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            Builder W{{verts, faces}, max_depth, 0, nullptr, nullptr};
            for (;;) {
                size_t k = next.fetch_add(1);
                if (k >= order.size()) break;
                Task &t = tasks[order[k]];
                t.arena.nodes.resize(1);
                t.arena.heap.resize(1);
                W.node(t.arena, 0, t.heap, t.depth, t.bd, t.bf, std::move(t.faces), false);
            }
        };
        std::vector<std::thread> pool;
        int nt = (int)std::min<size_t>((size_t)n_threads, tasks.size());
        for (int i = 1; i < nt; i++) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
    }

    // stitch: leaf numbers follow the DFS event order; task nodes are appended behind the top arena
    std::vector<uint32_t> top_leaf_global(top.leaves.size());
    std::vector<uint32_t> task_leaf_base(tasks.size()), task_node_base(tasks.size());
    uint64_t n_leaves = 0, n_nodes = top.nodes.size();
    for (int64_t ev : events) {
        if (ev >= 0) top_leaf_global[(size_t)ev] = (uint32_t)n_leaves++;
        else { size_t k = (size_t)~ev; task_leaf_base[k] = (uint32_t)n_leaves; n_leaves += tasks[k].arena.leaves.size(); }
    }
    for (size_t k = 0; k < tasks.size(); k++) { task_node_base[k] = (uint32_t)n_nodes; n_nodes += tasks[k].arena.nodes.size() - 1; }

    out.nodes.assign((size_t)n_nodes, KdNodeDev{0.0f, 0u});
    out.heap_index.assign((size_t)n_nodes, 0u);
    out.leaves.assign((size_t)n_leaves, KdLeafDev{{-1, -1, -1, -1}});
    out.depth = top.depth;
    out.n_internal = top.n_internal; out.n_leaf = top.n_leaf; out.n_empty = top.n_empty;
    for (size_t s = 0; s < top.nodes.size(); s++) {
        KdNodeDev n = top.nodes[s];
        if ((n.info >> 30) == 3) {
            uint32_t li = n.info & 0x3fffffffu;
            n.info = T_LEAF | top_leaf_global[li];
            out.leaves[top_leaf_global[li]] = top.leaves[li];
        }
        out.nodes[s] = n;
        out.heap_index[s] = top.heap[s];
    }
    for (size_t k = 0; k < tasks.size(); k++) {
        const Arena &A = tasks[k].arena;
        const uint32_t nb = task_node_base[k], lb = task_leaf_base[k];
        auto fix = [&](KdNodeDev n) {
            uint32_t ty = n.info >> 30;
            if (ty == 1) n.info = (n.info & 0xf0000000u) | (nb + (n.info & 0x0fffffffu) - 1);
            else if (ty == 3) n.info = T_LEAF | (lb + (n.info & 0x3fffffffu));
            return n;
        };
        out.nodes[tasks[k].top_slot] = fix(A.nodes[0]);
        out.heap_index[tasks[k].top_slot] = A.heap[0];
        for (size_t j = 1; j < A.nodes.size(); j++) {
            out.nodes[nb + j - 1] = fix(A.nodes[j]);
            out.heap_index[nb + j - 1] = A.heap[j];
        }
        for (size_t j = 0; j < A.leaves.size(); j++) out.leaves[lb + j] = A.leaves[j];
        out.depth = std::max(out.depth, A.depth);
        out.n_internal += A.n_internal; out.n_leaf += A.n_leaf; out.n_empty += A.n_empty;
    }
    for (int k = 0; k < 3; k++) { out.root_box[k] = rb.lo[k]; out.root_box[3 + k] = rb.hi[k]; }
    out.n_nodes = (int64_t)out.nodes.size();
    relayout_treelets(out);
}

// Re-orders the node array into cache-line treelets. Children are addressed as a PAIR (lower child at
// `info & 0x0fffffff`, upper child in the next slot), so a treelet is rooted at a pair: 16 slots = 128 bytes =
// one L1/L2 line hold a pair, its four children and its eight grandchildren at fixed positions
//   [0,1] pair | [2,3] kids of 0 | [4,5] kids of 1 | [6,7] [8,9] kids of 2,3 | [10,11] [12,13] kids of 4,5 | [14,15] unused
// so three consecutive descent steps touch one line. Slots of children that do not exist stay INVALID holes.
// The pairs below the third level start treelets of their own, laid out depth-first. Only positions change:
// heap indices, leaf numbers and splits travel with their node.
void relayout_treelets(KdTreeHost &t) {
    const size_t n_old = t.nodes.size();
    if (n_old < 2) return;
    std::vector<KdNodeDev> nn;
    std::vector<uint32_t> nh;
    nn.reserve(n_old * 2);
    nh.reserve(n_old * 2);
    nn.push_back(t.nodes[0]);
    nh.push_back(t.heap_index[0]);
    for (int i = 1; i < 16; i++) { nn.push_back(KdNodeDev{0.0f, 0u}); nh.push_back(0xffffffffu); }   // root keeps a line of its own
    struct Job { uint32_t old_pair; uint32_t parent_new; };   // pair to place; new index of the node that owns it
    std::vector<Job> todo;
    auto is_int = [](const KdNodeDev &n) { return (n.info >> 30) == 1u; };
    if (is_int(t.nodes[0])) todo.push_back({t.nodes[0].info & 0x0fffffffu, 0u});
    while (!todo.empty()) {
        const Job j = todo.back();
        todo.pop_back();
        const uint32_t base = (uint32_t)nn.size();
        nn.resize(base + 16, KdNodeDev{0.0f, 0u});
        nh.resize(base + 16, 0xffffffffu);
        nn[j.parent_new].info = (nn[j.parent_new].info & 0xf0000000u) | base;
        // slot -> old node index for the three levels of this treelet
        uint32_t old_of[14];
        for (auto &o : old_of) o = 0xffffffffu;
        old_of[0] = j.old_pair;
        old_of[1] = j.old_pair + 1;
        static const int kid_slot[6] = {2, 4, 6, 8, 10, 12};   // first kid slot of slots 0..5
        for (int sl = 0; sl < 6; sl++) {
            const uint32_t o = old_of[sl];
            if (o == 0xffffffffu || !is_int(t.nodes[o])) continue;
            const uint32_t c = t.nodes[o].info & 0x0fffffffu;
            old_of[kid_slot[sl]] = c;
            old_of[kid_slot[sl] + 1] = c + 1;
        }
        for (int sl = 0; sl < 14; sl++) {
            const uint32_t o = old_of[sl];
            if (o == 0xffffffffu) continue;
            nn[base + sl] = t.nodes[o];
            nh[base + sl] = t.heap_index[o];
            if (sl < 6 && is_int(t.nodes[o])) nn[base + sl].info = (t.nodes[o].info & 0xf0000000u) | (base + (uint32_t)kid_slot[sl]);
        }
        // grandchildren that are internal own the pairs of the next treelets; push in reverse so that the
        // lowest slot is laid out first
        for (int sl = 13; sl >= 6; sl--) {
            const uint32_t o = old_of[sl];
            if (o != 0xffffffffu && is_int(t.nodes[o])) todo.push_back({t.nodes[o].info & 0x0fffffffu, base + (uint32_t)sl});
        }
    }
    t.nodes.swap(nn);
    t.heap_index.swap(nh);
}

void export_reference(const KdTreeHost &t, uint32_t *pairs) {
    std::vector<std::pair<uint32_t, uint32_t>> v;
    v.reserve(t.nodes.size());
    for (size_t i = 0; i < t.nodes.size(); i++) {
        uint32_t info = t.nodes[i].info, ty = info >> 30;
        if (ty == 0) continue;   // hole of the treelet layout
        uint32_t meta = ty == 1 ? T_INTERNAL : (ty == 2 ? T_EMPTY : (T_LEAF | (info & 0x3fffffffu)));
        v.push_back({t.heap_index[i], meta});
    }
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size(); i++) { pairs[2 * i] = v[i].first; pairs[2 * i + 1] = v[i].second; }
}

}  // namespace rt
