// Exact IoU of oriented 3D boxes given by their 8 corners (N x M pairs): the quantity the reference takes from
// pytorch3d box3d_overlap / _C.iou_box3d [third-party, absent] at ProposalNetwork/utils/utils.py:194-210 (iou_3d),
// cubercnn/evaluation/omni3d_evaluation.py:155 (AP3D) and cubercnn/modeling/roi_heads/roi_heads.py:518,526,1563.
// Restated from the problem's definition (oracle/iou3d.py is the float64 statement of the same algorithm): the boundary
// of the intersection of two convex polyhedra consists of pieces of their faces; every quad face of one box is clipped
// (Sutherland-Hodgman) against the six half-spaces of the other and the volume is 1/3 * sum (n . p0) * area.
// Faces of box 1 are clipped inclusively; a face of box 2 exclusively against equally oriented planes of box 1 (coplanar
// faces of equal orientation count once, of opposite orientation cancel).
// One thread per pair, coordinates relative to box 1's centre (float32, error ~1e-5 of the box volume).
#include "cr_common.h"
#include <math.h>

__device__ __constant__ int IOU_FACES[6][4] = {{0, 1, 2, 3}, {3, 2, 6, 7}, {0, 1, 5, 4}, {0, 3, 7, 4}, {1, 2, 6, 5}, {4, 5, 6, 7}};
#define MAXV 12      // a quad clipped by 6 planes has at most 10 vertices

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ void box_planes(const V3* c, V3* n, float* off) {
    V3 ctr{0, 0, 0};
    for (int i = 0; i < 8; ++i) { ctr.x += c[i].x; ctr.y += c[i].y; ctr.z += c[i].z; }
    ctr.x *= 0.125f; ctr.y *= 0.125f; ctr.z *= 0.125f;
    for (int f = 0; f < 6; ++f) {
        const V3 a = c[IOU_FACES[f][0]];
        V3 m = cross3(sub(c[IOU_FACES[f][1]], a), sub(c[IOU_FACES[f][3]], a));
        const float l = sqrtf(dot3(m, m));
        const float s = (dot3(m, sub(ctr, a)) > 0.f ? -1.f : 1.f) / fmaxf(l, 1e-30f);
        m.x *= s; m.y *= s; m.z *= s;
        n[f] = m;
        off[f] = dot3(m, a);
    }
}

// clip `poly` (nv vertices) by n.x - off <= shift; returns the new vertex count
__device__ int clip_poly(V3* poly, int nv, V3 n, float off, float shift) {
    V3 out[MAXV];
    int m = 0;
    float d[MAXV];
    for (int i = 0; i < nv; ++i) d[i] = dot3(n, poly[i]) - off - shift;
    for (int i = 0; i < nv; ++i) {
        const int j = (i + 1 == nv) ? 0 : i + 1;
        const bool in_i = d[i] <= 0.f, in_j = d[j] <= 0.f;
        if (in_i && m < MAXV) out[m++] = poly[i];
        if (in_i != in_j && m < MAXV) {
            const float t = d[i] / (d[i] - d[j]);
            out[m++] = V3{poly[i].x + (poly[j].x - poly[i].x) * t, poly[i].y + (poly[j].y - poly[i].y) * t,
                          poly[i].z + (poly[j].z - poly[i].z) * t};
        }
    }
    for (int i = 0; i < m; ++i) poly[i] = out[i];
    return m;
}

__device__ float face_term(const V3* poly, int nv, V3 n) {
    if (nv < 3) return 0.f;
    V3 s{0, 0, 0};
    for (int i = 1; i + 1 < nv; ++i) {
        const V3 c = cross3(sub(poly[i], poly[0]), sub(poly[i + 1], poly[0]));
        s.x += c.x; s.y += c.y; s.z += c.z;
    }
    return dot3(n, poly[0]) * 0.5f * fabsf(dot3(s, n));
}

__device__ float box_vol(const V3* c) {
    return fabsf(dot3(sub(c[1], c[0]), cross3(sub(c[3], c[0]), sub(c[4], c[0]))));
}

__global__ __launch_bounds__(64) void k_box3d_overlap(const float* __restrict__ b1, const float* __restrict__ b2, int N, int M,
                                                      float* __restrict__ vol, float* __restrict__ iou) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * M) return;
    const int i = (int)(idx / M), j = (int)(idx - (int64_t)i * M);
    V3 c1[8], c2[8];
    V3 o{0, 0, 0};
    for (int k = 0; k < 8; ++k) {
        c1[k] = V3{b1[(i * 8 + k) * 3], b1[(i * 8 + k) * 3 + 1], b1[(i * 8 + k) * 3 + 2]};
        c2[k] = V3{b2[(j * 8 + k) * 3], b2[(j * 8 + k) * 3 + 1], b2[(j * 8 + k) * 3 + 2]};
        o.x += c1[k].x; o.y += c1[k].y; o.z += c1[k].z;
    }
    o.x *= 0.125f; o.y *= 0.125f; o.z *= 0.125f;
    float scale = 0.f;
    for (int k = 0; k < 8; ++k) {
        c1[k] = sub(c1[k], o); c2[k] = sub(c2[k], o);
        scale = fmaxf(scale, fmaxf(fmaxf(fabsf(c1[k].x), fabsf(c1[k].y)), fabsf(c1[k].z)));
        scale = fmaxf(scale, fmaxf(fmaxf(fabsf(c2[k].x), fabsf(c2[k].y)), fabsf(c2[k].z)));
    }
    const float tol = 2e-6f * scale;
    V3 n1[6], n2[6];
    float o1[6], o2[6];
    box_planes(c1, n1, o1);
    box_planes(c2, n2, o2);
    float v = 0.f;
    for (int f = 0; f < 6; ++f) {
        V3 poly[MAXV];
        int nv = 4;
        for (int k = 0; k < 4; ++k) poly[k] = c1[IOU_FACES[f][k]];
        for (int k = 0; k < 6 && nv > 0; ++k) nv = clip_poly(poly, nv, n2[k], o2[k], tol);
        v += face_term(poly, nv, n1[f]);
    }
    for (int f = 0; f < 6; ++f) {
        V3 poly[MAXV];
        int nv = 4;
        for (int k = 0; k < 4; ++k) poly[k] = c2[IOU_FACES[f][k]];
        // exclusive only against planes of box 1 with this face's orientation (a coplanar pair of equal orientation was
        // counted with box 1); faces of opposite orientation (touching boxes) both stay and cancel
        for (int k = 0; k < 6 && nv > 0; ++k)
            nv = clip_poly(poly, nv, n1[k], o1[k], dot3(n2[f], n1[k]) > 0.999f ? -tol : tol);
        v += face_term(poly, nv, n2[f]);
    }
    v = fmaxf(v * (1.f / 3.f), 0.f);
    const float va = box_vol(c1), vb = box_vol(c2);
    v = fminf(v, fminf(va, vb));
    vol[idx] = v;
    iou[idx] = v / (va + vb - v);
}

extern "C" int cr_box3d_overlap(cr_ctx* ctx, const float* boxes1, const float* boxes2, int N, int M, float* vol, float* iou) {
    CR_CHECK_ARG(ctx && N >= 0 && M >= 0, "cr_box3d_overlap: bad args");
    if ((int64_t)N * M == 0) return CR_OK;
    CR_CHECK_ARG(boxes1 && boxes2 && vol && iou, "cr_box3d_overlap: NULL pointer");
    hipLaunchKernelGGL(k_box3d_overlap, dim3((unsigned)cr_cdiv((int64_t)N * M, 64)), dim3(64), 0, ctx->stream, boxes1, boxes2,
                       N, M, vol, iou);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
