// Fused losses of the weakly supervised 3D head on the static (B, kf) foreground slots: ROIHeads3DScore._forward_cube in training
// mode (cubercnn/modeling/roi_heads/roi_heads.py:1366-1760 of the reference), which the reference -- and the torch composition
// in weak_losses.py that mirrors it -- evaluates as ~550 small tensor operations per direction.  Here:
//
//   k_weak_fwd     one workgroup per image, one lane per slot: decode (as k_cube_loss: deltas -> centre, exp dims x prior,
//                  allocentric -> egocentric pose, virtual depth), the cuboid in metres (:1547-1560), its 8 corners projected
//                  with K and clamped (Cubes.get_bube_corners, spaces.py:233-243), their XYXY hull (conversions.py:25-48), and
//                  the per-RoI terms  iou (GIoU, :1585-1590) | normal (pose_ground, :1610-1622) | z (50-step depth search,
//                  :1151-1194) | dims_w/h/l (prior hinge, :1234-1254); the pairwise pose alignment of the image (:1055-1074)
//                  from the rotations parked in LDS; the integer window of the depth median (:1196-1232)
//   cr_box_median  (weak.hip) median depth of every window
//   k_weak_reduce  one workgroup: pose alignment over the images, pseudo depth targets in the reference's order, uncertainty
//                  weighting (:1700-1712), safely_reduce_losses (:2843-2851) of every term over the valid slots, logged statistics
//   k_weak_bwd     one workgroup per image: the gradients of the reduced terms w.r.t. the selected head outputs (the class
//                  scatter and the 6D-rotation backward are cr_cube_select_bwd's)
//
// Empty slots (validf == 0) are computed and ignored.  Terms: 0 iou 1 pose 2 normal 3 z 4 pseudo_gt_z 5 dims_w 6 dims_h 7 dims_l;
// index 8 of the reduced vectors is the uncertainty itself.
#include "cr_common.h"
#include "cube_math.h"
#include <math.h>

#define WK_NT 8
#define WK_TAB 20            // per-image table: K (9, row-major, K / ratio with K[2][2] = 1) | clamp x_lo x_hi y_lo y_hi | ground
                             // confidence | image height, width | 4 columns the kernels do not read (host-side use)
#define WK_SQRT2F 1.41421356f
#define WK_STEPS 50
#define WK_MAXT 256          // slots per image (threads of the per-image workgroups)

struct WeakIn {
    const float *dxy, *zr, *dr, *Ra, *u, *v2r, *prior_mean;      // chunks of cr_cube_select's buf39
    const float* src_boxes;                                      // (n,4) proposal boxes
    const unsigned char* validf;                                 // (n)
    const int* clsc;                                             // (n) class of the slot, clamped into range
    const int64_t* gt_idx;                                       // (B,S) matched ground-truth object of every sampled slot
    const float* gt_boxes;                                       // (B,G,4)
    const float* prior_std;                                      // (K,3) or NULL
    const float* table;                                          // (B,WK_TAB)
    const float* normals;                                        // (B,3) ground normals or NULL
    int B, kf, S, G, allocentric, terms;                         // terms: bit k set = term k is part of the loss
};

struct WeakRow {            // what forward and backward both need of one slot
    float cux, cuy, sw, sh, z, sf, u, dims[3], R[9], M[9], K[9], bnd[4], gtb[4], x3d, y3d, pm[3], ps[3], gconf, nrm[3];
    bool rot, dclip[3], valid;
};

__device__ __forceinline__ float clamp_nan(float x, float lo, float hi) { return (x != x) ? x : fminf(fmaxf(x, lo), hi); }

__device__ __forceinline__ void weak_decode(const WeakIn& in, int i, int b, WeakRow& r) {
    const float* sb = in.src_boxes + (size_t)i * 4;
    const float* tab = in.table + (size_t)b * WK_TAB;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.K[k] = tab[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) r.bnd[k] = tab[9 + k];
    r.gconf = tab[13];
    r.valid = in.validf[i] != 0;
    r.sw = sb[2] - sb[0]; r.sh = sb[3] - sb[1];
    r.cux = (sb[0] + 0.5f * r.sw) + r.sw * in.dxy[i * 2];
    r.cuy = (sb[1] + 0.5f * r.sh) + r.sh * in.dxy[i * 2 + 1];
    const int c = in.clsc[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float d = in.dr[i * 3 + k];
        r.dclip[k] = !(d <= 5.0f);
        r.pm[k] = in.prior_mean[i * 3 + k];
        r.ps[k] = in.prior_std ? in.prior_std[c * 3 + k] : 1.0f;
        r.dims[k] = expf(fminf(d, 5.0f)) * r.pm[k];
    }
    const float K4[4] = {r.K[0], r.K[4], r.K[2], r.K[5]};
    r.rot = false;
    if (in.allocentric) r.rot = ray_rotation(r.cux, r.cuy, K4, r.M);
    const float* Ra = in.Ra + (size_t)i * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            r.R[a * 3 + q] = r.rot ? (r.M[a * 3] * Ra[q] + r.M[a * 3 + 1] * Ra[3 + q]) + r.M[a * 3 + 2] * Ra[6 + q] : Ra[a * 3 + q];
    r.z = in.zr[i] * in.v2r[i];
    r.u = in.u[i];
    r.sf = WK_SQRT2F * expf(-r.u);
    r.x3d = r.z * (r.cux - r.K[2]) / r.K[0];
    r.y3d = r.z * (r.cuy - r.K[5]) / r.K[4];
    const int64_t gi = in.gt_idx[(size_t)b * in.S + (i - b * in.kf)];
    const float* g = in.gt_boxes + ((size_t)b * in.G + (r.valid ? gi : 0)) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.gtb[k] = g[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) r.nrm[k] = in.normals ? in.normals[b * 3 + k] : 0.f;
}

// corners without the centre: rl[v][a] = sum_b R[a][b] loc[v][b]
__device__ __forceinline__ void corner_offsets(const float* dims, const float* R, float rl[8][3]) {
    const float zero[3] = {0.f, 0.f, 0.f};
    corners(zero, dims, R, rl);
}

// project the corners rl + c, clamp, hull.  uu / vv: the unclamped image coordinates (for the clamp's pass-through test)
__device__ __forceinline__ void project_box(const float rl[8][3], const float* c, const float* K, const float* bnd, float* box,
                                            float* uu, float* vv, float* pz, int* arg) {
    float lo_u = INFINITY, hi_u = -INFINITY, lo_v = INFINITY, hi_v = -INFINITY;
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    bool nan = false;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const float X = rl[v][0] + c[0], Y = rl[v][1] + c[1], Z = rl[v][2] + c[2];
        const float p0 = (K[0] * X + K[1] * Y) + K[2] * Z;
        const float p1 = (K[3] * X + K[4] * Y) + K[5] * Z;
        const float p2 = (K[6] * X + K[7] * Y) + K[8] * Z;
        const float fu = p0 / p2, fv = p1 / p2;
        if (uu) { uu[v] = fu; vv[v] = fv; pz[v] = p2; }
        const float cu = clamp_nan(fu, bnd[0], bnd[1]), cv = clamp_nan(fv, bnd[2], bnd[3]);
        nan |= (cu != cu) | (cv != cv);
        if (cu < lo_u) { lo_u = cu; a0 = v; }
        if (cv < lo_v) { lo_v = cv; a1 = v; }
        if (cu > hi_u) { hi_u = cu; a2 = v; }
        if (cv > hi_v) { hi_v = cv; a3 = v; }
    }
    const float qn = __uint_as_float(0x7fc00000u);
    box[0] = nan ? qn : lo_u; box[1] = nan ? qn : lo_v; box[2] = nan ? qn : hi_u; box[3] = nan ? qn : hi_v;
    if (arg) { arg[0] = a0; arg[1] = a1; arg[2] = a2; arg[3] = a3; }
}

// torchvision.ops.generalized_box_iou_loss(gt, pred), eps 1e-7; g != nullptr: gradient w.r.t. pred scaled by up
__device__ __forceinline__ float giou_loss(const float* gt, const float* p, float up, float* g) {
    const float eps = 1e-7f;
    const float x1 = gt[0], y1 = gt[1], x2 = gt[2], y2 = gt[3], a1 = p[0], b1 = p[1], a2 = p[2], b2 = p[3];
    const float xk1 = fmaxf(x1, a1), yk1 = fmaxf(y1, b1), xk2 = fminf(x2, a2), yk2 = fminf(y2, b2);
    const bool has = (yk2 > yk1) && (xk2 > xk1);
    const float inter = has ? (xk2 - xk1) * (yk2 - yk1) : 0.f;
    const float areaP = (a2 - a1) * (b2 - b1);
    const float uni = ((x2 - x1) * (y2 - y1) + areaP) - inter;
    const float iou = inter / (uni + eps);
    const float xc1 = fminf(x1, a1), yc1 = fminf(y1, b1), xc2 = fmaxf(x2, a2), yc2 = fmaxf(y2, b2);
    const float areaC = (xc2 - xc1) * (yc2 - yc1);
    const float loss = 1.f - (iou - (areaC - uni) / (areaC + eps));
    if (p[0] != p[0] || p[1] != p[1] || p[2] != p[2] || p[3] != p[3]) {      // NaN boxes: fmin / fmax above drop NaN, torch keeps it
        if (g) { g[0] = g[1] = g[2] = g[3] = 0.f; }
        return __uint_as_float(0x7fc00000u);
    }
    if (g) {
        const float d_areaC = up * (uni + eps) / ((areaC + eps) * (areaC + eps));
        const float d_uni = -up / (areaC + eps) + up * inter / ((uni + eps) * (uni + eps));     // d_iou = -up
        const float d_inter = -up / (uni + eps) - d_uni;
        float ga1 = -d_uni * (b2 - b1), ga2 = d_uni * (b2 - b1), gb1 = -d_uni * (a2 - a1), gb2 = d_uni * (a2 - a1);
        if (has) {
            if (a1 > x1) ga1 -= d_inter * (yk2 - yk1);
            if (a2 < x2) ga2 += d_inter * (yk2 - yk1);
            if (b1 > y1) gb1 -= d_inter * (xk2 - xk1);
            if (b2 < y2) gb2 += d_inter * (xk2 - xk1);
        }
        if (a1 < x1) ga1 -= d_areaC * (yc2 - yc1);
        if (a2 > x2) ga2 += d_areaC * (yc2 - yc1);
        if (b1 < y1) gb1 -= d_areaC * (xc2 - xc1);
        if (b2 > y2) gb2 += d_areaC * (xc2 - xc1);
        g[0] = ga1; g[1] = gb1; g[2] = ga2; g[3] = gb2;
    }
    return loss;
}

// torch.linspace(0, 4.9, 50)[i]
__device__ __forceinline__ float z_step(int i) {
    const float step = 4.9f / 49.0f;
    return i < WK_STEPS / 2 ? step * (float)i : 4.9f - step * (float)(WK_STEPS - 1 - i);
}

// z_search_loss (:1151-1194) without the uncertainty weight: 8 lanes per RoI walk the 50 depth steps (lane q takes steps q, q+8,
// ...), the best step is the first minimum of |gt area - projected area| over all of them (torch.argmin: a NaN beats any number,
// the first one wins).  Reads the decoded cuboid (dec) and its hull (pbox) written by k_weak_fwd.
__device__ __forceinline__ bool z_better(float da, int sa, float db, int sb) {       // is (da, sa) the argmin over (db, sb)?
    const bool na = da != da, nb = db != db;
    if (na != nb) return na;
    if (na) return sa < sb;
    if (da != db) return da < db;
    return sa < sb;
}

__global__ __launch_bounds__(256) void k_weak_zsearch(WeakIn in, const float* __restrict__ dec, const float* __restrict__ pbox,
                                                      float* __restrict__ Lraw) {
    const int t = blockIdx.x * 256 + threadIdx.x, n = in.B * in.kf;
    const int i = min(t >> 3, n - 1), q = t & 7, b = i / in.kf;          // (whole waves stay active for the shuffles)
    const float* d = dec + (size_t)i * 17;
    const float* tab = in.table + (size_t)b * WK_TAB;
    float K[9], bnd[4], rl[8][3], gt[4];
#pragma unroll
    for (int k = 0; k < 9; ++k) K[k] = tab[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) bnd[k] = tab[9 + k];
    corner_offsets(d + 3, d + 6, rl);
    const bool valid = in.validf[i] != 0;
    const int64_t gi = in.gt_idx[(size_t)b * in.S + (i - b * in.kf)];
    const float* g = in.gt_boxes + ((size_t)b * in.G + (valid ? gi : 0)) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) gt[k] = g[k];
    const float* box = pbox + (size_t)i * 4;
    const float z = d[2];
    const float gt_area = (gt[2] - gt[0]) * (gt[3] - gt[1]);
    const float pcx = (box[0] + box[2]) / 2.f, pcy = (box[1] + box[3]) / 2.f;
    const float pred_area = (box[2] - box[0]) * (box[3] - box[1]);
    // as in the reference: `(a <= c) <= b` compares a boolean with b
    const float bx = (gt[0] - WK_STEPS <= pcx) ? 1.f : 0.f, by = (gt[1] - WK_STEPS <= pcy) ? 1.f : 0.f;
    const bool within = (bx <= gt[2] + WK_STEPS) && (by <= gt[3] + WK_STEPS);
    const float sign = gt_area < pred_area ? 1.f : -1.f;
    float best = 0.f;
    int best_s = -1;
    for (int s = q; s < WK_STEPS; s += 8) {
        const float c[3] = {d[15], d[16], z + sign * z_step(s)};
        float bb[4];
        project_box(rl, c, K, bnd, bb, nullptr, nullptr, nullptr, nullptr);
        float area = (bb[2] - bb[0]) * (bb[3] - bb[1]);
        area = area + (area == 0.f ? 10000000.f : 0.f);
        const float dd = fabsf(gt_area - area);
        if (best_s < 0 || z_better(dd, s, best, best_s)) { best = dd; best_s = s; }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
        const float ob = __shfl_down(best, off, 64);
        const int os = __shfl_down(best_s, off, 64);
        if (z_better(ob, os, best, best_s)) { best = ob; best_s = os; }
    }
    if (q == 0 && (t >> 3) < n) {
        const float found = fabsf(z - (z + sign * z_step(best_s)));
        Lraw[(size_t)i * WK_NT + 3] = (within ? found : 0.1f * WK_STEPS) / 2.f;
    }
}

// block-wide sum of one float per thread (result in every thread); scratch >= blockDim.x / 64 floats
__device__ __forceinline__ float block_sum(float v, float* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < nw; ++k) s += scratch[k];
    return s;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WK_MAXT) void k_weak_fwd(WeakIn in, float* __restrict__ Lraw, float* __restrict__ dec,
                                                   float* __restrict__ pbox, int* __restrict__ ibox, float* __restrict__ pimg) {
    extern __shared__ float s_dyn[];                   // kf x 9 rotations | kf validity
    __shared__ float s_red[16];
    const int b = blockIdx.x, j = threadIdx.x, kf = in.kf;
    float* s_R = s_dyn;
    float* s_val = s_dyn + (size_t)kf * 9;
    const bool lane = j < kf;
    const int i = b * kf + (lane ? j : 0);
    WeakRow r;
    weak_decode(in, i, b, r);
    if (lane) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s_R[j * 9 + k] = r.R[k];
        s_val[j] = r.valid ? 1.f : 0.f;
    }
    __syncthreads();
    float L[WK_NT];
#pragma unroll
    for (int k = 0; k < WK_NT; ++k) L[k] = 0.f;
    float rl[8][3], box[4];
    corner_offsets(r.dims, r.R, rl);
    const float c[3] = {r.x3d, r.y3d, r.z};
    project_box(rl, c, r.K, r.bnd, box, nullptr, nullptr, nullptr, nullptr);
    if (in.terms & 1) L[0] = giou_loss(r.gtb, box, 0.f, nullptr);
    if (in.terms & 4) {
        const float* y = r.R + 3;
        const float nx = sqrtf((r.nrm[0] * r.nrm[0] + r.nrm[1] * r.nrm[1]) + r.nrm[2] * r.nrm[2]);
        const float ny = sqrtf((y[0] * y[0] + y[1] * y[1]) + y[2] * y[2]);
        const float ix = 1.f / fmaxf(nx, 1e-8f), iy = 1.f / fmaxf(ny, 1e-8f);
        const float cs = ((r.nrm[0] * ix) * (y[0] * iy) + (r.nrm[1] * ix) * (y[1] * iy)) + (r.nrm[2] * ix) * (y[2] * iy);
        L[2] = (1.f - fabsf(cs)) * r.gconf;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (in.terms & (32 << k)) L[5 + k] = fmaxf(fabsf(r.dims[k] - r.pm[k]) / r.ps[k] - 1.f, 0.f);

    // pose alignment of the image: mean over the pairs of valid slots of 1 - |cos of the relative angle|
    float mine = 0.f;
    if (in.terms & 2) {
        if (lane && r.valid) {
            for (int q = 0; q < j; ++q) {
                if (s_val[q] == 0.f) continue;
                float tr = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) tr += r.R[k] * s_R[q * 9 + k];
                mine += 1.f - fabsf((tr - 1.f) * 0.5f);
            }
        }
    }
    const float pair_sum = block_sum(mine, s_red);
    const float m = block_sum(lane && r.valid ? 1.f : 0.f, s_red);
    if (j == 0) {
        const float pairs = m * (m - 1.f) * 0.5f;
        pimg[b * 2] = pair_sum / pairs;                 // 0 / 0 = NaN for an image without (or with one) valid slot
        pimg[b * 2 + 1] = m;
    }
    if (lane) {
#pragma unroll
        for (int k = 0; k < WK_NT; ++k) Lraw[(size_t)i * WK_NT + k] = L[k];
        float* o = dec + (size_t)i * 17;
        o[0] = r.cux; o[1] = r.cuy; o[2] = r.z; o[3] = r.dims[0]; o[4] = r.dims[1]; o[5] = r.dims[2];
#pragma unroll
        for (int k = 0; k < 9; ++k) o[6 + k] = r.R[k];
        o[15] = r.x3d; o[16] = r.y3d;
#pragma unroll
        for (int k = 0; k < 4; ++k) pbox[(size_t)i * 4 + k] = box[k];
        // window of the depth median: the projected box clipped to the image, truncated to integers (:1205-1216)
        const float* tab = in.table + (size_t)b * WK_TAB;
        const float ih = tab[14], iw = tab[15];
        const float q0 = fminf(fmaxf(box[0], 0.f), iw), q1 = fminf(fmaxf(box[1], 0.f), ih);
        const float q2 = fminf(fmaxf(box[2], 0.f), iw), q3 = fminf(fmaxf(box[3], 0.f), ih);
        const bool inside = r.valid && !(box[0] != box[0]) && (q2 - q0) * (q3 - q1) > 0.f;
        int* ib = ibox + (size_t)i * 4;
        ib[0] = inside ? (int)q0 : 0; ib[1] = inside ? (int)q1 : 0; ib[2] = inside ? (int)q2 : 0; ib[3] = inside ? (int)q3 : 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct WeakRed {
    const float *u, *gt2d, *gtz, *gtdims;              // buf39 chunks (uncertainty, ground truth for the logged errors)
    const unsigned char* validf;
    const float* table;
    const float* depth; int H, W;                      // (B,H,W) padded depth maps or NULL
    const float* med;                                  // (n) window medians (pgz_mode 1)
    const float* gt_boxes; const int64_t* gt_idx;
    int B, kf, S, G, n, terms, pgz_mode;               // pgz_mode: 0 none, 1 window median, 2 depth under the centre
    float w[WK_NT]; float w_normal_extra;              // weights of the logged total
};

#define WR_T 256
#define WR_MAXN 8192          // slots of the whole batch (B * kf)
__device__ __forceinline__ float depth_at(const WeakRed& p, int b, float x, float y, float ih, float iw) {
    // clamp 10 px inside the image (:1225-1229, 1262-1270); NaN coordinates go to the lower bound instead of faulting
    const float cx = fminf(fmaxf(x == x ? x : 10.f, 10.f), iw - 11.f), cy = fminf(fmaxf(y == y ? y : 10.f, 10.f), ih - 11.f);
    const int xi = min(max((int)cx, 0), p.W - 1), yi = min(max((int)cy, 0), p.H - 1);
    return p.depth[((size_t)b * p.H + yi) * p.W + xi];
}

__global__ __launch_bounds__(WR_T) void k_weak_reduce(WeakRed p, float* __restrict__ Lraw, const float* __restrict__ dec,
                                                      const float* __restrict__ pbox, const int* __restrict__ ibox,
                                                      const float* __restrict__ pimg, float* __restrict__ ztgt,
                                                      float* __restrict__ red, float* __restrict__ cnt,
                                                      float* __restrict__ stats, float* __restrict__ aux) {
    __shared__ float s_acc[32][WR_T / 64];
    __shared__ float s_pose, s_fail1;
    __shared__ unsigned char s_flag[WR_MAXN];
    const int tid = threadIdx.x, n = p.n, kf = p.kf;
    // ---- pose alignment over the images (:1055-1074): images with exactly one slot are skipped and counted
    if (tid == 0) {
        float tot = 0.f;
        int fail = 0;
        for (int b = 0; b < p.B; ++b) {
            if (pimg[b * 2 + 1] == 1.f) ++fail;
            else tot += pimg[b * 2];
        }
        const bool none = fail == p.B;                  // the reference drops the term; here it is 0 with no gradient
        s_pose = none ? 0.f : tot * 1.f / (float)(fail + 1);
        s_fail1 = none ? 0.f : (float)(fail + 1);
        aux[0] = s_fail1;
    }
    for (int i = tid; i < n; i += WR_T) {
        const float* tab = p.table + (size_t)(i / kf) * WK_TAB;
        const float ih = tab[14], iw = tab[15];
        const float* bx = pbox + (size_t)i * 4;
        const float q0 = fminf(fmaxf(bx[0], 0.f), iw), q1 = fminf(fmaxf(bx[1], 0.f), ih);
        const float q2 = fminf(fmaxf(bx[2], 0.f), iw), q3 = fminf(fmaxf(bx[3], 0.f), ih);
        const bool area = !(bx[0] != bx[0]) && (q2 - q0) * (q3 - q1) > 0.f;
        s_flag[i] = (unsigned char)((p.validf[i] ? 1 : 0) | (area ? 2 : 0));
    }
    __syncthreads();
    const float pose = s_pose;
    // ---- pseudo depth targets (:1196-1232 / :1256-1279), then every slot's terms
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    for (int i = tid; i < n; i += WR_T) {
        const int b = i / kf;
        const bool valid = p.validf[i] != 0;
        const float* tab = p.table + (size_t)b * WK_TAB;
        const float ih = tab[14], iw = tab[15];
        const float* d = dec + (size_t)i * 17;
        float tz = 0.f;
        if (p.pgz_mode == 2) {
            tz = depth_at(p, b, d[0], d[1], ih, iw);
        } else if (p.pgz_mode == 1 && valid) {
            // as in the reference the targets of an image are ordered [windows with area..., windows without...], i.e. permuted
            // against the predictions when an image has both kinds: slot i with rank r among the valid slots takes entry r.
            // s_flag: bit 0 valid, bit 1 the (float) window has area
            int rnk = 0, nin = 0;
            for (int q = b * kf; q < (b + 1) * kf; ++q) {
                const int f = s_flag[q];
                rnk += (q < i) & f & 1;
                nin += f == 3 ? 1 : 0;
            }
            const bool want_in = rnk < nin;
            const int want = want_in ? rnk : rnk - nin, wflag = want_in ? 3 : 1;
            int src = -1, seen = 0;
            for (int q = b * kf; q < (b + 1) * kf && src < 0; ++q) {
                if (s_flag[q] != wflag) continue;
                if (seen == want) src = q;
                ++seen;
            }
            if (src < 0) src = i;
            if (want_in) {
                tz = p.med[src];
            } else {
                const float* bx = pbox + (size_t)src * 4;
                const float q0 = fminf(fmaxf(bx[0], 0.f), iw), q1 = fminf(fmaxf(bx[1], 0.f), ih);
                const float q2 = fminf(fmaxf(bx[2], 0.f), iw), q3 = fminf(fmaxf(bx[3], 0.f), ih);
                tz = depth_at(p, b, (q0 + q2) / 2.f, (q1 + q3) / 2.f, ih, iw);
            }
        }
        float* L = Lraw + (size_t)i * WK_NT;
        if (p.terms & 2) L[1] = pose;
        if (p.terms & 16) L[4] = fabsf(d[2] - tz);
        if (ztgt) ztgt[i] = tz;
        if (!valid) continue;
        const float u = p.u[i], sf = WK_SQRT2F * expf(-u);
        float total = 0.f;
#pragma unroll
        for (int k = 0; k < WK_NT; ++k) {
            if (!(p.terms & (1 << k))) continue;
            const float e = L[k] * sf;
            if (isfinite(e)) { acc[k] += e; acc[9 + k] += 1.f; }
            total += L[k] * p.w[k];
        }
        if (p.terms & 4) total += (L[2] * p.w_normal_extra) * tab[13];
        if (isfinite(u)) { acc[8] += u; acc[17] += 1.f; }
        // logged statistics over the valid slots (:1668-1687)
        const float zerr = fabsf(d[2] - p.gtz[i]);
        acc[18] += zerr;
        acc[19] += (fabsf(d[3] - p.gtdims[i * 3]) + fabsf(d[4] - p.gtdims[i * 3 + 1])) + fabsf(d[5] - p.gtdims[i * 3 + 2]);
        acc[20] += fabsf(d[0] - p.gt2d[i * 2]) + fabsf(d[1] - p.gt2d[i * 2 + 1]);
        acc[21] += zerr < 0.20f ? 1.f : 0.f;
        {
            const int64_t gi = p.gt_idx[(size_t)b * p.S + (i - b * kf)];
            const float* g = p.gt_boxes + ((size_t)b * p.G + gi) * 4;
            const float* bx = pbox + (size_t)i * 4;
            const float iwd = fmaxf(fminf(g[2], bx[2]) - fmaxf(g[0], bx[0]), 0.f), iht = fmaxf(fminf(g[3], bx[3]) - fmaxf(g[1], bx[1]), 0.f);
            const float inter = iwd * iht;
            const float un = ((g[2] - g[0]) * (g[3] - g[1]) + (bx[2] - bx[0]) * (bx[3] - bx[1])) - inter;
            acc[22] += inter > 0.f ? inter / un : 0.f;
        }
        acc[23] += expf(-u);
        if (isfinite(total)) { acc[24] += total; acc[25] += 1.f; }
        acc[26] += 1.f;
    }
    // ---- deterministic block reduction of the 27 accumulators
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        float v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) s_acc[k][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 27) {
        float v = 0.f;
        for (int w = 0; w < WR_T / 64; ++w) v += s_acc[tid][w];
        s_acc[tid][0] = v;
    }
    __syncthreads();
    if (tid == 0) {
        const float nv = s_acc[26][0];
        for (int k = 0; k < 9; ++k) {
            const bool on = k == 8 || (p.terms & (1 << k));
            const float c = s_acc[9 + k][0];
            cnt[k] = c;
            // safely_reduce_losses: mean over the finite entries; none finite -> mean * 0 (NaN); no valid slot at all -> 0
            red[k] = !on || nv == 0.f ? 0.f : (c > 0.f ? s_acc[k][0] / c : __uint_as_float(0x7fc00000u));
        }
        const float inv = nv > 0.f ? 1.f / nv : 0.f;
        stats[0] = s_acc[18][0] * inv;                  // z_error
        stats[1] = s_acc[19][0] * inv / 3.f;            // dims_error
        stats[2] = s_acc[20][0] * inv / 2.f;            // xy_error
        stats[3] = s_acc[21][0] * inv;                  // z_close
        stats[4] = s_acc[22][0] * inv;                  // 2D IoU
        stats[5] = s_acc[23][0] * inv;                  // conf
        stats[6] = s_acc[25][0] > 0.f ? s_acc[24][0] / s_acc[25][0] : 0.f;     // total_3D_loss / loss_w_3d
        stats[7] = nv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WK_MAXT) void k_weak_bwd(WeakIn in, const float* __restrict__ gred, const float* __restrict__ cnt,
                                                   const float* __restrict__ aux, const float* __restrict__ Lraw,
                                                   const float* __restrict__ dec, const float* __restrict__ ztgt,
                                                   const float* __restrict__ pimg, float* __restrict__ g_dxy,
                                                   float* __restrict__ g_zr, float* __restrict__ g_dr, float* __restrict__ g_Ra,
                                                   float* __restrict__ g_u) {
    extern __shared__ float s_dyn[];
    __shared__ float s_red[16];
    const int b = blockIdx.x, j = threadIdx.x, kf = in.kf, n = in.B * kf;
    float* s_R = s_dyn;
    float* s_val = s_dyn + (size_t)kf * 9;
    const bool lane = j < kf;
    const int i = b * kf + (lane ? j : 0);
    WeakRow r;
    weak_decode(in, i, b, r);
    if (lane) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s_R[j * 9 + k] = r.R[k];
        s_val[j] = r.valid ? 1.f : 0.f;
    }
    float gk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) gk[k] = cnt[k] > 0.f ? gred[k] / cnt[k] : 0.f;
    // d(total) / d(pose scalar): every valid slot carries the scalar times its own uncertainty factor
    float gp = 0.f;
    if (in.terms & 2) {
        for (int q = j; q < n; q += blockDim.x) {
            if (!in.validf[q]) continue;
            const float sfq = WK_SQRT2F * expf(-in.u[q]);
            if (isfinite(Lraw[(size_t)q * WK_NT + 1] * sfq)) gp += gk[1] * sfq;
        }
    }
    const float g_pose = block_sum(gp, s_red);           // (also the barrier after the LDS writes above)
    if (!lane) return;
    const float* L = Lraw + (size_t)i * WK_NT;
    float gL[WK_NT], du = 0.f;
#pragma unroll
    for (int k = 0; k < WK_NT; ++k) {
        const float e = L[k] * r.sf;
        const bool fin = r.valid && (in.terms & (1 << k)) && isfinite(e);
        gL[k] = fin ? gk[k] * r.sf : 0.f;
        du -= fin ? gk[k] * e : 0.f;
    }
    if (r.valid && isfinite(r.u)) du += gk[8];
    float d_cux = 0.f, d_cuy = 0.f, d_z = 0.f, d_dims[3] = {0, 0, 0}, d_R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // ---- iou: hull -> clamp -> perspective division -> K -> corners
    if (gL[0] != 0.f) {
        float rl[8][3], box[4], uu[8], vv[8], pz[8], gb[4];
        int arg[4];
        corner_offsets(r.dims, r.R, rl);
        const float c[3] = {r.x3d, r.y3d, r.z};
        project_box(rl, c, r.K, r.bnd, box, uu, vv, pz, arg);
        giou_loss(r.gtb, box, gL[0], gb);
        float dP[8][3];
#pragma unroll
        for (int v = 0; v < 8; ++v) dP[v][0] = dP[v][1] = dP[v][2] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = arg[e];
            const bool isu = (e & 1) == 0;
            const float f = isu ? uu[v] : vv[v];
            const float lo = isu ? r.bnd[0] : r.bnd[2], hi = isu ? r.bnd[1] : r.bnd[3];
            if (!(f >= lo && f <= hi)) continue;          // torch.clamp passes the gradient inside the bounds (inclusive)
            const float g = gb[e];
            // f = p_num / p2
            const float d_num = g / pz[v], d_p2 = -g * f / pz[v];
            const float* Kn = isu ? r.K : r.K + 3;
#pragma unroll
            for (int a = 0; a < 3; ++a) dP[v][a] += Kn[a] * d_num + r.K[6 + a] * d_p2;
        }
        float dc[3] = {0, 0, 0};
        corners_bwd(dP, r.dims, r.R, dc, d_dims, d_R);
        d_z += (dc[0] * (r.cux - r.K[2]) / r.K[0] + dc[1] * (r.cuy - r.K[5]) / r.K[4]) + dc[2];
        d_cux += dc[0] * r.z / r.K[0];
        d_cuy += dc[1] * r.z / r.K[4];
    }
    // ---- ground normal vs the cuboid's up axis (row 1 of R)
    if (gL[2] != 0.f) {
        const float* y = r.R + 3;
        const float nx = sqrtf((r.nrm[0] * r.nrm[0] + r.nrm[1] * r.nrm[1]) + r.nrm[2] * r.nrm[2]);
        const float ny = sqrtf((y[0] * y[0] + y[1] * y[1]) + y[2] * y[2]);
        const float ix = 1.f / fmaxf(nx, 1e-8f), iy = 1.f / fmaxf(ny, 1e-8f);
        const float xh[3] = {r.nrm[0] * ix, r.nrm[1] * ix, r.nrm[2] * ix};
        const float cs = (xh[0] * (y[0] * iy) + xh[1] * (y[1] * iy)) + xh[2] * (y[2] * iy);
        const float g = -gL[2] * r.gconf * sgn(cs);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // d cs / d y_k = xh_k / |y| - cs y_k / |y|^2 (|y| above eps); below eps the clamp makes cs linear in y
            const float dk = ny > 1e-8f ? xh[k] * iy - cs * y[k] * iy * iy : xh[k] * iy;
            d_R[3 + k] += g * dk;
        }
    }
    // ---- pseudo depth: |z - target|
    if (gL[4] != 0.f) d_z += gL[4] * sgn(r.z - ztgt[i]);
    // ---- dimension hinge
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (gL[5 + k] == 0.f) continue;
        const float a = fabsf(r.dims[k] - r.pm[k]) / r.ps[k] - 1.f;
        if (a > 0.f) d_dims[k] += gL[5 + k] * sgn(r.dims[k] - r.pm[k]) / r.ps[k];
    }
    // ---- pose alignment: d/dR_i of sum over the pairs of (1 - |(tr(Ri Rj^T) - 1) / 2|) / pairs / (fail + 1)
    if ((in.terms & 2) && r.valid && aux[0] > 0.f && g_pose != 0.f) {
        const float m = pimg[b * 2 + 1];
        if (m > 1.f) {
            const float scale = g_pose / (m * (m - 1.f) * 0.5f) / aux[0];
            for (int q = 0; q < kf; ++q) {
                if (q == j || s_val[q] == 0.f) continue;
                float tr = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) tr += r.R[k] * s_R[q * 9 + k];
                const float g = -scale * 0.5f * sgn((tr - 1.f) * 0.5f);
#pragma unroll
                for (int k = 0; k < 9; ++k) d_R[k] += g * s_R[q * 9 + k];
            }
        }
    }
    g_dxy[i * 2] = d_cux * r.sw;
    g_dxy[i * 2 + 1] = d_cuy * r.sh;
    g_zr[i] = d_z * in.v2r[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) g_dr[i * 3 + k] = r.dclip[k] ? 0.f : d_dims[k] * r.dims[k];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q)          // dRa = M^T dR
            g_Ra[(size_t)i * 9 + a * 3 + q] = r.rot ? (r.M[a] * d_R[q] + r.M[3 + a] * d_R[3 + q]) + r.M[6 + a] * d_R[6 + q] : d_R[a * 3 + q];
    g_u[i] = du;
}

// ---------------------------------------------------------------------------------------------------------------------
static int weak_args(WeakIn& in, const float* const* p, const unsigned char* validf, const int* clsc, const int64_t* gt_idx,
                     const float* gt_boxes, const float* prior_std, const float* table, const float* normals, int B, int kf,
                     int S, int G, int allocentric, int terms) {
    for (int k = 0; k < 8; ++k) CR_CHECK_ARG(p[k] != nullptr, "weak_loss: NULL input pointer #%d", k);
    CR_CHECK_ARG(validf && clsc && gt_idx && gt_boxes && table, "weak_loss: NULL pointer");
    CR_CHECK_ARG(B > 0 && kf > 0 && kf <= WK_MAXT && kf <= S && G > 0, "weak_loss: bad sizes (B=%d kf=%d S=%d G=%d)", B, kf, S, G);
    CR_CHECK_ARG(!(terms & 4) || normals, "weak_loss: the ground-normal term needs normals");
    CR_CHECK_ARG((terms & ~0xff) == 0, "weak_loss: unknown term bits");
    in.dxy = p[0]; in.zr = p[1]; in.dr = p[2]; in.Ra = p[3]; in.u = p[4]; in.v2r = p[5]; in.prior_mean = p[6]; in.src_boxes = p[7];
    in.validf = validf; in.clsc = clsc; in.gt_idx = gt_idx; in.gt_boxes = gt_boxes; in.prior_std = prior_std; in.table = table;
    in.normals = normals; in.B = B; in.kf = kf; in.S = S; in.G = G; in.allocentric = allocentric; in.terms = terms;
    return CR_OK;
}

// inputs: HOST array of 8 device pointers [dxy, zr, dr, Ra, u, v2r, prior_mean (chunks of cr_cube_select's buf39), src_boxes (n,4)].
// Lraw (n,8), dec (n,17), pbox (n,4), ibox (n,4) int32, pimg (B,2).
extern "C" int cr_weak_loss_fwd(cr_ctx* ctx, const float* const* inputs, const unsigned char* validf, const int32_t* clsc,
                                const int64_t* gt_idx, const float* gt_boxes, const float* prior_std, const float* table,
                                const float* normals, int B, int kf, int S, int G, int allocentric, int terms, float* Lraw,
                                float* dec, float* pbox, int32_t* ibox, float* pimg) {
    CR_CHECK_ARG(ctx && inputs && Lraw && dec && pbox && ibox && pimg, "cr_weak_loss_fwd: NULL pointer");
    WeakIn in;
    int rc = weak_args(in, inputs, validf, clsc, gt_idx, gt_boxes, prior_std, table, normals, B, kf, S, G, allocentric, terms);
    if (rc) return rc;
    const int T = (int)cr_cdiv(kf, 64) * 64;
    hipLaunchKernelGGL(k_weak_fwd, dim3((unsigned)B), dim3((unsigned)T), (size_t)kf * 10 * sizeof(float), ctx->stream, in, Lraw, dec,
                       pbox, ibox, pimg);
    if (terms & 8)
        hipLaunchKernelGGL(k_weak_zsearch, dim3((unsigned)cr_cdiv((int64_t)B * kf * 8, 256)), dim3(256), 0, ctx->stream, in, dec, pbox,
                           Lraw);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// red (9), cnt (9), stats (8), aux (1); red_inputs: HOST array of 4 device pointers [u, gt2d, gtz, gtdims] (buf39 chunks).
// weights (9): the eight term weights of the logged total, then the extra factor of the ground-normal term.
extern "C" int cr_weak_loss_reduce(cr_ctx* ctx, const float* const* red_inputs, const unsigned char* validf, const float* table,
                                   const float* depth, int H, int W, const float* med, const float* gt_boxes,
                                   const int64_t* gt_idx, int B, int kf, int S, int G, int terms, int pgz_mode,
                                   const float* weights, float* Lraw, const float* dec, const float* pbox, const int32_t* ibox,
                                   const float* pimg, float* ztgt, float* red, float* cnt, float* stats, float* aux) {
    CR_CHECK_ARG(ctx && red_inputs && validf && table && gt_boxes && gt_idx && weights && Lraw && dec && pbox && ibox && pimg && ztgt
                 && red && cnt && stats && aux, "cr_weak_loss_reduce: NULL pointer");
    CR_CHECK_ARG(B > 0 && kf > 0 && kf <= S && G > 0 && (int64_t)B * kf <= WR_MAXN, "cr_weak_loss_reduce: bad sizes");
    CR_CHECK_ARG(pgz_mode >= 0 && pgz_mode <= 2, "cr_weak_loss_reduce: pgz_mode %d", pgz_mode);
    CR_CHECK_ARG(pgz_mode == 0 || (depth && H > 0 && W > 0), "cr_weak_loss_reduce: the pseudo depth target needs the depth maps");
    CR_CHECK_ARG(pgz_mode != 1 || med, "cr_weak_loss_reduce: pgz_mode 1 needs the window medians");
    CR_CHECK_ARG(((terms >> 4) & 1) == (pgz_mode != 0), "cr_weak_loss_reduce: term 4 and pgz_mode disagree");
    WeakRed p;
    for (int k = 0; k < 4; ++k) CR_CHECK_ARG(red_inputs[k] != nullptr, "cr_weak_loss_reduce: NULL input #%d", k);
    p.u = red_inputs[0]; p.gt2d = red_inputs[1]; p.gtz = red_inputs[2]; p.gtdims = red_inputs[3];
    p.validf = validf; p.table = table; p.depth = depth; p.H = H; p.W = W; p.med = med; p.gt_boxes = gt_boxes; p.gt_idx = gt_idx;
    p.B = B; p.kf = kf; p.S = S; p.G = G; p.n = B * kf; p.terms = terms; p.pgz_mode = pgz_mode;
    for (int k = 0; k < WK_NT; ++k) p.w[k] = weights[k];
    p.w_normal_extra = weights[WK_NT];
    hipLaunchKernelGGL(k_weak_reduce, dim3(1), dim3(WR_T), 0, ctx->stream, p, Lraw, dec, pbox, ibox, pimg, ztgt, red, cnt, stats, aux);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// gred (9) = d(total) / d(red); outputs g_dxy (n,2) g_zr (n) g_dr (n,3) g_Ra (n,9) g_u (n) for cr_cube_select_bwd
extern "C" int cr_weak_loss_bwd(cr_ctx* ctx, const float* const* inputs, const unsigned char* validf, const int32_t* clsc,
                                const int64_t* gt_idx, const float* gt_boxes, const float* prior_std, const float* table,
                                const float* normals, int B, int kf, int S, int G, int allocentric, int terms, const float* gred,
                                const float* cnt, const float* aux, const float* Lraw, const float* dec, const float* ztgt,
                                const float* pimg, float* g_dxy, float* g_zr, float* g_dr, float* g_Ra, float* g_u) {
    CR_CHECK_ARG(ctx && inputs && gred && cnt && aux && Lraw && dec && ztgt && pimg && g_dxy && g_zr && g_dr && g_Ra && g_u,
                 "cr_weak_loss_bwd: NULL pointer");
    WeakIn in;
    int rc = weak_args(in, inputs, validf, clsc, gt_idx, gt_boxes, prior_std, table, normals, B, kf, S, G, allocentric, terms);
    if (rc) return rc;
    const int T = (int)cr_cdiv(kf, 64) * 64;
    hipLaunchKernelGGL(k_weak_bwd, dim3((unsigned)B), dim3((unsigned)T), (size_t)kf * 10 * sizeof(float), ctx->stream, in, gred, cnt,
                       aux, Lraw, dec, ztgt, pimg, g_dxy, g_zr, g_dr, g_Ra, g_u);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
