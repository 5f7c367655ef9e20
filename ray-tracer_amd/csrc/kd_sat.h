// Triangle/box overlap test of the k-d builder, shared by the host builder (kd_build.cpp) and the device builder
// (kd_build_gpu.hip) so that both decide every face exactly alike. Double precision, no fused multiply-add.
// Reference: KdTree.faceBoxOverlap / axisTest_* / planeBoxOverlap, js/8-kd_tree.js:1737-1970.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define KD_HD __host__ __device__
#else
#define KD_HD
#endif

namespace rt {

// One separating-axis test of the reference's axisTest_* family (js/8-kd_tree.js:1829-1929):
// projections pa/pb of two vertices against the box radius on that axis.
KD_HD inline bool kd_separated(double pa, double pb, double rad) {
    return pa < pb ? (pa > rad || pb < -rad) : (pb > rad || pa < -rad);
}

// projections: X-type uses (y,z), Y-type (x,z) with the reference's signs, Z-type (x,y)
KD_HD inline double kd_px(const double *e, const double *p) { return e[2] * p[1] - e[1] * p[2]; }
KD_HD inline double kd_py(const double *e, const double *p) { return -e[2] * p[0] + e[0] * p[2]; }
KD_HD inline double kd_pz(const double *e, const double *p) { return e[1] * p[0] - e[0] * p[1]; }
KD_HD inline double kd_rx(const double *e, const double *H) { return fabs(e[2]) * H[1] + fabs(e[1]) * H[2]; }
KD_HD inline double kd_ry(const double *e, const double *H) { return fabs(e[2]) * H[0] + fabs(e[0]) * H[2]; }
KD_HD inline double kd_rz(const double *e, const double *H) { return fabs(e[1]) * H[0] + fabs(e[0]) * H[1]; }

// Triangle (s0, s1, s2) against the box [lo, hi], exactly as js/8-kd_tree.js:1737-1827 evaluates it. The reference
// forms the normal with Vector3.cross(), which in three.js r84 overwrites its receiver: after js/8-kd_tree.js:1760 the
// "first edge" is the normal, and the first three axis tests (js/8-kd_tree.js:1762-1773) use the normal's components.
// Kept as is - it decides which faces each child receives.
KD_HD inline bool kd_tri_box_overlap(const double *s0, const double *s1, const double *s2, const double *lo, const double *hi) {
    double H[3], Cn[3], P[3][3];
    for (int k = 0; k < 3; k++) {
        Cn[k] = (lo[k] + hi[k]) * 0.5;
        H[k] = fabs(hi[k] - lo[k]) * 0.5;
    }
    for (int k = 0; k < 3; k++) {
        P[0][k] = s0[k] - Cn[k];
        P[1][k] = s1[k] - Cn[k];
        P[2][k] = s2[k] - Cn[k];
    }
    double E0[3], E1[3], E2[3], N[3];
    for (int k = 0; k < 3; k++) {
        E0[k] = P[1][k] - P[0][k];
        E1[k] = P[2][k] - P[1][k];
        E2[k] = P[0][k] - P[2][k];
    }
    N[0] = E0[1] * E1[2] - E0[2] * E1[1];
    N[1] = E0[2] * E1[0] - E0[0] * E1[2];
    N[2] = E0[0] * E1[1] - E0[1] * E1[0];

    // "edge 0" = the normal: X01, Y02, Z12
    if (kd_separated(kd_px(N, P[0]), kd_px(N, P[2]), kd_rx(N, H))) return false;
    if (kd_separated(kd_py(N, P[0]), kd_py(N, P[2]), kd_ry(N, H))) return false;
    if (kd_separated(kd_pz(N, P[1]), kd_pz(N, P[2]), kd_rz(N, H))) return false;
    // edge 1: X01, Y02, Z0
    if (kd_separated(kd_px(E1, P[0]), kd_px(E1, P[2]), kd_rx(E1, H))) return false;
    if (kd_separated(kd_py(E1, P[0]), kd_py(E1, P[2]), kd_ry(E1, H))) return false;
    if (kd_separated(kd_pz(E1, P[0]), kd_pz(E1, P[1]), kd_rz(E1, H))) return false;
    // edge 2: X2, Y1, Z12
    if (kd_separated(kd_px(E2, P[0]), kd_px(E2, P[1]), kd_rx(E2, H))) return false;
    if (kd_separated(kd_py(E2, P[0]), kd_py(E2, P[1]), kd_ry(E2, H))) return false;
    if (kd_separated(kd_pz(E2, P[1]), kd_pz(E2, P[2]), kd_rz(E2, H))) return false;

    for (int k = 0; k < 3; k++) {
        double mn = fmin(P[0][k], fmin(P[1][k], P[2][k]));
        double mx = fmax(P[0][k], fmax(P[1][k], P[2][k]));
        if (mn > H[k] || mx < -H[k]) return false;
    }
    // planeBoxOverlap(normal, vert0, halfsize), js/8-kd_tree.js:1931-1970
    double vmin[3], vmax[3];
    for (int k = 0; k < 3; k++) {
        double v = P[0][k];
        if (N[k] > 0) { vmin[k] = -H[k] - v; vmax[k] = H[k] - v; }
        else { vmin[k] = H[k] - v; vmax[k] = -H[k] - v; }
    }
    if (N[0] * vmin[0] + N[1] * vmin[1] + N[2] * vmin[2] > 0) return false;
    return N[0] * vmax[0] + N[1] * vmax[1] + N[2] * vmax[2] >= 0;
}

}  // namespace rt
