/*
 * ORACLE - test infrastructure only. Nothing in the product path may call this file.
 *
 * Sequential CPU restatement of the reference k-d tree builder, in the reference's own
 * order of operations (double precision, explicit node stack, upper child popped first):
 *   KdTree.buildTree      /root/reference/js/8-kd_tree.js:1647-1732
 *   KdTree.faceBoxOverlap js/8-kd_tree.js:1737-1827   (Akenine-Moller triangle/box SAT)
 *   axisTest_*            js/8-kd_tree.js:1829-1929
 *   planeBoxOverlap       js/8-kd_tree.js:1931-1970
 *   fillDataArray         js/8-kd_tree.js:1632-1645   (heap-index serialisation)
 *   KdLeaf                js/8-kd_tree.js:1610-1617   (at most 4 faces, rest dropped)
 *
 * Pinned by: tests/golden/kd_* (arrays and SHA-256 produced by running the reference
 * JavaScript itself under Node, oracle/gen_golden/ref_capture.js).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y, z; } v3;

typedef struct onode {
    uint32_t metadata;
    struct onode *child[2];
    v3 min, max;
    int depth;
    int32_t *faces;
    int32_t n_faces;
} onode;

typedef struct {
    const double *verts;
    const uint32_t *faces;
} mesh_t;

static int axis_x01(double a, double b, double fa, double fb, const v3 *v, v3 h) {
    double p0 = a * v[0].y - b * v[0].z, p2 = a * v[2].y - b * v[2].z, mn, mx;
    if (p0 < p2) { mn = p0; mx = p2; } else { mn = p2; mx = p0; }
    double rad = fa * h.y + fb * h.z;
    return !(mn > rad || mx < -rad);
}
static int axis_x2(double a, double b, double fa, double fb, const v3 *v, v3 h) {
    double p0 = a * v[0].y - b * v[0].z, p1 = a * v[1].y - b * v[1].z, mn, mx;
    if (p0 < p1) { mn = p0; mx = p1; } else { mn = p1; mx = p0; }
    double rad = fa * h.y + fb * h.z;
    return !(mn > rad || mx < -rad);
}
static int axis_y02(double a, double b, double fa, double fb, const v3 *v, v3 h) {
    double p0 = -a * v[0].x + b * v[0].z, p2 = -a * v[2].x + b * v[2].z, mn, mx;
    if (p0 < p2) { mn = p0; mx = p2; } else { mn = p2; mx = p0; }
    double rad = fa * h.x + fb * h.z;
    return !(mn > rad || mx < -rad);
}
static int axis_y1(double a, double b, double fa, double fb, const v3 *v, v3 h) {
    double p0 = -a * v[0].x + b * v[0].z, p1 = -a * v[1].x + b * v[1].z, mn, mx;
    if (p0 < p1) { mn = p0; mx = p1; } else { mn = p1; mx = p0; }
    double rad = fa * h.x + fb * h.z;
    return !(mn > rad || mx < -rad);
}
static int axis_z12(double a, double b, double fa, double fb, const v3 *v, v3 h) {
    double p1 = a * v[1].x - b * v[1].y, p2 = a * v[2].x - b * v[2].y, mn, mx;
    if (p2 < p1) { mn = p2; mx = p1; } else { mn = p1; mx = p2; }
    double rad = fa * h.x + fb * h.y;
    return !(mn > rad || mx < -rad);
}
static int axis_z0(double a, double b, double fa, double fb, const v3 *v, v3 h) {
    double p0 = a * v[0].x - b * v[0].y, p1 = a * v[1].x - b * v[1].y, mn, mx;
    if (p0 < p1) { mn = p0; mx = p1; } else { mn = p1; mx = p0; }
    double rad = fa * h.x + fb * h.y;
    return !(mn > rad || mx < -rad);
}

static int plane_box(v3 n, v3 vert, v3 mb) {
    v3 vmin, vmax;
    double v = vert.x;
    if (n.x > 0) { vmin.x = -mb.x - v; vmax.x = mb.x - v; } else { vmin.x = mb.x - v; vmax.x = -mb.x - v; }
    v = vert.y;
    if (n.y > 0) { vmin.y = -mb.y - v; vmax.y = mb.y - v; } else { vmin.y = mb.y - v; vmax.y = -mb.y - v; }
    v = vert.z;
    if (n.z > 0) { vmin.z = -mb.z - v; vmax.z = mb.z - v; } else { vmin.z = mb.z - v; vmax.z = -mb.z - v; }
    /* Vector3.dot: x*x' + y*y' + z*z' evaluated left to right */
    if (n.x * vmin.x + n.y * vmin.y + n.z * vmin.z > 0) return 0;
    if (n.x * vmax.x + n.y * vmax.y + n.z * vmax.z >= 0) return 1;
    return 0;
}

static v3 sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }

static int face_box_overlap(const mesh_t *m, v3 bmin, v3 bmax, int32_t face) {
    v3 c = { (bmin.x + bmax.x) * 0.5, (bmin.y + bmax.y) * 0.5, (bmin.z + bmax.z) * 0.5 };
    v3 h = { fabs(bmax.x - bmin.x) * 0.5, fabs(bmax.y - bmin.y) * 0.5, fabs(bmax.z - bmin.z) * 0.5 };
    v3 v[3], e[3];
    for (int k = 0; k < 3; k++) {
        const double *p = m->verts + 3 * (size_t)m->faces[3 * (size_t)face + k];
        v3 w = { p[0], p[1], p[2] };
        v[k] = sub(w, c);
    }
    e[0] = sub(v[1], v[0]);
    e[1] = sub(v[2], v[1]);
    e[2] = sub(v[0], v[2]);
    /* js/8-kd_tree.js:1760: r84 Vector3.cross() writes its result into `this`, so from here
     * on faceEdges[0] IS the normal and the first axis triple below runs on the normal. */
    {
        double ax = e[0].x, ay = e[0].y, az = e[0].z;
        double bx = e[1].x, by = e[1].y, bz = e[1].z;
        e[0].x = ay * bz - az * by;
        e[0].y = az * bx - ax * bz;
        e[0].z = ax * by - ay * bx;
    }
    v3 normal = e[0];

    double fex = fabs(e[0].x), fey = fabs(e[0].y), fez = fabs(e[0].z);
    if (!axis_x01(e[0].z, e[0].y, fez, fey, v, h)) return 0;
    if (!axis_y02(e[0].z, e[0].x, fez, fex, v, h)) return 0;
    if (!axis_z12(e[0].y, e[0].x, fey, fex, v, h)) return 0;

    fex = fabs(e[1].x); fey = fabs(e[1].y); fez = fabs(e[1].z);
    if (!axis_x01(e[1].z, e[1].y, fez, fey, v, h)) return 0;
    if (!axis_y02(e[1].z, e[1].x, fez, fex, v, h)) return 0;
    if (!axis_z0(e[1].y, e[1].x, fey, fex, v, h)) return 0;

    fex = fabs(e[2].x); fey = fabs(e[2].y); fez = fabs(e[2].z);
    if (!axis_x2(e[2].z, e[2].y, fez, fey, v, h)) return 0;
    if (!axis_y1(e[2].z, e[2].x, fez, fex, v, h)) return 0;
    if (!axis_z12(e[2].y, e[2].x, fey, fex, v, h)) return 0;

    double mn = fmin(v[0].x, fmin(v[1].x, v[2].x)), mx = fmax(v[0].x, fmax(v[1].x, v[2].x));
    if (mn > h.x || mx < -h.x) return 0;
    mn = fmin(v[0].y, fmin(v[1].y, v[2].y)); mx = fmax(v[0].y, fmax(v[1].y, v[2].y));
    if (mn > h.y || mx < -h.y) return 0;
    mn = fmin(v[0].z, fmin(v[1].z, v[2].z)); mx = fmax(v[0].z, fmax(v[1].z, v[2].z));
    if (mn > h.z || mx < -h.z) return 0;

    return plane_box(normal, v[0], h);
}

typedef struct {
    onode *root;
    int32_t *leaves;     /* 4 per leaf */
    int64_t n_leaves, cap_leaves;
    int64_t n_nodes;
    int depth;
} otree;

static onode *new_node(void) { return (onode *)calloc(1, sizeof(onode)); }

static void free_tree(onode *n) {
    if (!n) return;
    free_tree(n->child[0]);
    free_tree(n->child[1]);
    free(n->faces);
    free(n);
}

static void count_nodes(const onode *n, int64_t *c) {
    (*c)++;
    if ((n->metadata >> 30) == 1) { count_nodes(n->child[0], c); count_nodes(n->child[1], c); }
}

/* fillDataArray/traverseTree: pre-order walk assigning heap indices 2i+1 / 2i+2 */
static void emit(const onode *n, uint64_t idx, uint32_t *pairs, int64_t *k) {
    pairs[2 * *k] = (uint32_t)idx;
    pairs[2 * *k + 1] = n->metadata;
    (*k)++;
    if ((n->metadata >> 30) == 1) {
        emit(n->child[0], 2 * idx + 1, pairs, k);
        emit(n->child[1], 2 * idx + 2, pairs, k);
    }
}

static int cmp_pair(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* Builds the tree; returns an opaque handle. infinity = the script's INFINITY constant. */
void *kd_oracle_build(const double *verts, int32_t n_verts, const uint32_t *faces, int32_t n_faces,
                      int max_depth, double infinity) {
    (void)n_verts;
    mesh_t m = { verts, faces };
    otree *t = (otree *)calloc(1, sizeof(otree));
    if (max_depth > 24) max_depth = 24;
    onode *root = new_node();
    t->root = root;
    root->depth = 1;
    root->n_faces = n_faces;
    root->faces = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_faces > 0 ? n_faces : 1));
    for (int32_t i = 0; i < n_faces; i++) root->faces[i] = i;
    root->min.x = root->min.y = root->min.z = infinity;
    root->max.x = root->max.y = root->max.z = -infinity;
    /* js/8-kd_tree.js:1657-1669: x from vertex 0 only, y from vertex 1 only, z from vertex 2 only */
    for (int32_t i = 0; i < n_faces; i++) {
        const uint32_t *f = faces + 3 * (size_t)i;
        double x = verts[3 * (size_t)f[0]], y = verts[3 * (size_t)f[1] + 1], z = verts[3 * (size_t)f[2] + 2];
        root->min.x = fmin(root->min.x, x); root->max.x = fmax(root->max.x, x);
        root->min.y = fmin(root->min.y, y); root->max.y = fmax(root->max.y, y);
        root->min.z = fmin(root->min.z, z); root->max.z = fmax(root->max.z, z);
    }

    size_t cap = 64, sp = 0;
    onode **stack = (onode **)malloc(cap * sizeof(onode *));
    stack[sp++] = root;
    while (sp) {
        onode *n = stack[--sp];
        if (n->depth > t->depth) t->depth = n->depth;
        if (n->n_faces > 4 && n->depth < max_depth) {
            n->metadata = 1u << 30;
            for (int i = 0; i < 2; i++) {
                v3 cmin = n->min, cmax = n->max;
                onode *c = new_node();
                n->child[i] = c;
                int axis = n->depth % 3;
                if (axis == 1) {
                    cmin.x = n->min.x + 0.5 * i * (n->max.x - n->min.x);
                    cmax.x = n->min.x + 0.5 * (i + 1) * (n->max.x - n->min.x);
                } else if (axis == 2) {
                    cmin.y = n->min.y + 0.5 * i * (n->max.y - n->min.y);
                    cmax.y = n->min.y + 0.5 * (i + 1) * (n->max.y - n->min.y);
                } else {
                    cmin.z = n->min.z + 0.5 * i * (n->max.z - n->min.z);
                    cmax.z = n->min.z + 0.5 * (i + 1) * (n->max.z - n->min.z);
                }
                c->faces = (int32_t *)malloc(sizeof(int32_t) * (size_t)n->n_faces);
                c->n_faces = 0;
                for (int32_t j = 0; j < n->n_faces; j++) {
                    if (face_box_overlap(&m, cmin, cmax, n->faces[j])) c->faces[c->n_faces++] = n->faces[j];
                }
                c->min = cmin; c->max = cmax; c->depth = n->depth + 1;
                if (sp == cap) { cap *= 2; stack = (onode **)realloc(stack, cap * sizeof(onode *)); }
                stack[sp++] = c;
            }
            continue;
        }
        if (n->n_faces == 0) { n->metadata = 2u << 30; continue; }
        if (t->n_leaves == t->cap_leaves) {
            t->cap_leaves = t->cap_leaves ? t->cap_leaves * 2 : 1024;
            t->leaves = (int32_t *)realloc(t->leaves, sizeof(int32_t) * 4 * (size_t)t->cap_leaves);
        }
        int32_t *lf = t->leaves + 4 * t->n_leaves;
        lf[0] = lf[1] = lf[2] = lf[3] = -1;
        for (int i = 0; i < (n->n_faces < 4 ? n->n_faces : 4); i++) lf[i] = n->faces[i];
        n->metadata = (3u << 30) | (uint32_t)t->n_leaves;
        t->n_leaves++;
    }
    free(stack);
    count_nodes(root, &t->n_nodes);
    return t;
}

void kd_oracle_counts(void *h, int64_t *n_nodes, int64_t *n_leaves, int32_t *depth) {
    otree *t = (otree *)h;
    *n_nodes = t->n_nodes; *n_leaves = t->n_leaves; *depth = t->depth;
}

/* canonical arrays: (heapIndex, metadata) pairs ascending by heap index, leaves, root box */
void kd_oracle_export(void *h, uint32_t *pairs, int32_t *leaves, double *root_box) {
    otree *t = (otree *)h;
    int64_t k = 0;
    emit(t->root, 0, pairs, &k);
    qsort(pairs, (size_t)k, 2 * sizeof(uint32_t), cmp_pair);
    memcpy(leaves, t->leaves, sizeof(int32_t) * 4 * (size_t)t->n_leaves);
    root_box[0] = t->root->min.x; root_box[1] = t->root->min.y; root_box[2] = t->root->min.z;
    root_box[3] = t->root->max.x; root_box[4] = t->root->max.y; root_box[5] = t->root->max.z;
}

void kd_oracle_free(void *h) {
    otree *t = (otree *)h;
    if (!t) return;
    free_tree(t->root);
    free(t->leaves);
    free(t);
}
