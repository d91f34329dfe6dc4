// K15/K16: fused Cube R-CNN 3D-head decode + disentangled corner losses, forward and backward, one lane per
// foreground RoI (n <= 128 per image).  Replaces ~350 elementwise torch launches per direction of
// ROIHeads3D._forward_cube (cubercnn/modeling/roi_heads/roi_heads.py:2353-2679 of the reference):
//   decode        cube_x/y = ctr + wh*delta; dims = exp(min(.,5))*prior_mean; z = z_raw*virtual_to_real   (:2371-2436)
//   allocentric   R = M(ray through cube_x,cube_y) @ R_alloc where angle > 0                              (math_util.py:802-830)
//   corners       get_cuboid_verts_faces                                                                  (math_util.py:142-245)
//   losses        disentangled L1 corner losses for xy / z / dims, chamfer for pose and joint             (:2471-2508,2575-2589)
//   uncertainty   every term x sqrt(2)*exp(-u)                                                            (:2633-2652)
// The reductions over RoIs (safely_reduce_losses) stay in the caller.
#include "cr_common.h"
#include "cube_math.h"
#include <math.h>

#define CH_T 64
#define SQRT2F 1.41421356f

struct CubeIn {
    const float *dxy, *zr, *dr, *Ra, *u;                 // head outputs for the RoI's class: (n,2) (n) (n,3) (n,9) (n)
    const float *src_boxes, *K4, *v2r, *prior_mean;      // (n,4) (n,4)=[fx,fy,cx,cy] (n) (n,3)
    const float *gt2d, *gtz, *gtdims, *gtR;              // (n,2) (n) (n,3) (n,9)
    int n, allocentric, chamfer_pose, use_conf, joint;
};

// mean over 24 of |P - G|; optionally its gradient w.r.t. P scaled by `up`
__device__ __forceinline__ float l1_corner(const float P[8][3], const float G[8][3], float dP[8][3], float up, bool grad) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float d = P[v][a] - G[v][a];
            s += fabsf(d);
            if (grad) dP[v][a] = sgn(d) * up / 24.f;
        }
    return s / 24.f;
}
// symmetric L1 chamfer over the 8x8 corner pairs (roi_heads.py:2209-2215); first minimum wins ties
__device__ __forceinline__ float chamfer(const float P[8][3], const float G[8][3], float dP[8][3], float up, bool grad) {
    float d[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            d[i][j] = (fabsf(P[i][0] - G[j][0]) + fabsf(P[i][1] - G[j][1])) + fabsf(P[i][2] - G[j][2]);
    if (grad) {
#pragma unroll
        for (int v = 0; v < 8; ++v) dP[v][0] = dP[v][1] = dP[v][2] = 0.f;
    }
    float s1 = 0.f, s2 = 0.f;
    // l1_dist.min(1): for each target j the closest prediction i
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int bi = 0;
        float bm = d[0][j];
#pragma unroll
        for (int i = 1; i < 8; ++i)
            if (d[i][j] < bm) { bm = d[i][j]; bi = i; }
        s1 += bm;
        if (grad) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i == bi) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) dP[i][a] += sgn(P[i][a] - G[j][a]) * up / 8.f;
                }
        }
    }
    // l1_dist.min(2): for each prediction i the closest target j
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int bj = 0;
        float bm = d[i][0];
#pragma unroll
        for (int j = 1; j < 8; ++j)
            if (d[i][j] < bm) { bm = d[i][j]; bj = j; }
        s2 += bm;
        if (grad) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j == bj) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) dP[i][a] += sgn(P[i][a] - G[j][a]) * up / 8.f;
                }
        }
    }
    return s1 / 8.f + s2 / 8.f;
}

// BWD = false: losses (n,5) = [dims, xy, z, pose, joint] (uncertainty-weighted), dec (n,17) = [cube_x, cube_y, z,
//              dims(3), R(9), x3d, y3d].   BWD = true: gradients of sum_k gl[i][k]*loss[i][k] w.r.t. the head outputs.
template <bool BWD>
__global__ __launch_bounds__(CH_T) void k_cube_loss(CubeIn in, const float* __restrict__ gl, float* __restrict__ losses,
                                                    float* __restrict__ dec, float* __restrict__ g_dxy,
                                                    float* __restrict__ g_zr, float* __restrict__ g_dr,
                                                    float* __restrict__ g_Ra, float* __restrict__ g_u) {
    const int i = blockIdx.x * CH_T + threadIdx.x;
    if (i >= in.n) return;
    const float* sb = in.src_boxes + i * 4;
    const float* K4 = in.K4 + i * 4;
    const float sw = sb[2] - sb[0], sh = sb[3] - sb[1];
    const float cux = (sb[0] + 0.5f * sw) + sw * in.dxy[i * 2], cuy = (sb[1] + 0.5f * sh) + sh * in.dxy[i * 2 + 1];
    float dims[3];
    bool dclip[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float r = in.dr[i * 3 + k];
        dclip[k] = !(r <= 5.0f);                          // clamp(max=5) passes the gradient where r <= 5
        dims[k] = expf(fminf(r, 5.0f)) * in.prior_mean[i * 3 + k];
    }
    float R[9], M[9];
    bool rot = false;
    if (in.allocentric) rot = ray_rotation(cux, cuy, K4, M);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float* Ra = in.Ra + i * 9;
            R[a * 3 + b] = rot ? (M[a * 3] * Ra[b] + M[a * 3 + 1] * Ra[3 + b]) + M[a * 3 + 2] * Ra[6 + b] : Ra[a * 3 + b];
        }
    const float z = in.zr[i] * in.v2r[i];
    const float u = in.u[i];
    const float sf = in.use_conf ? SQRT2F * expf(-u) : 1.0f;

    // ground truth
    const float g2x = in.gt2d[i * 2], g2y = in.gt2d[i * 2 + 1], gz = in.gtz[i];
    const float ga = (g2x - K4[2]) / K4[0], gb = (g2y - K4[3]) / K4[1];
    const float gc[3] = {gz * ga, gz * gb, gz};
    const float* gd = in.gtdims + i * 3;
    const float* gR = in.gtR + i * 9;
    float G[8][3], P[8][3], dP[8][3];
    corners(gc, gd, gR, G);

    const float pa = (cux - K4[2]) / K4[0], pb = (cuy - K4[3]) / K4[1];
    const float c_z[3] = {z * ga, z * gb, z};            // disentangled Z: predicted depth along the GT ray
    const float c_xy[3] = {gz * pa, gz * pb, gz};        // disentangled XY: GT depth along the predicted ray
    const float c_j[3] = {z * pa, z * pb, z};            // joint
    float up[5] = {0, 0, 0, 0, 0};
    if (BWD) {
#pragma unroll
        for (int k = 0; k < 5; ++k) up[k] = gl[i * 5 + k] * sf;
    }
    float d_cux = 0.f, d_cuy = 0.f, d_z = 0.f, d_dims[3] = {0, 0, 0}, d_R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float L[5];

    // dims
    corners(gc, dims, gR, P);
    L[0] = l1_corner(P, G, dP, up[0], BWD);
    if (BWD) corners_bwd(dP, dims, gR, nullptr, d_dims, nullptr);
    // xy
    corners(c_xy, gd, gR, P);
    L[1] = l1_corner(P, G, dP, up[1], BWD);
    if (BWD) {
        float dc[3] = {0, 0, 0};
        corners_bwd(dP, gd, gR, dc, nullptr, nullptr);
        d_cux += dc[0] * gz / K4[0];
        d_cuy += dc[1] * gz / K4[1];
    }
    // z
    corners(c_z, gd, gR, P);
    L[2] = l1_corner(P, G, dP, up[2], BWD);
    if (BWD) {
        float dc[3] = {0, 0, 0};
        corners_bwd(dP, gd, gR, dc, nullptr, nullptr);
        d_z += (dc[0] * ga + dc[1] * gb) + dc[2];
    }
    // pose
    corners(gc, gd, R, P);
    L[3] = in.chamfer_pose ? chamfer(P, G, dP, up[3], BWD) : l1_corner(P, G, dP, up[3], BWD);
    if (BWD) corners_bwd(dP, gd, R, nullptr, nullptr, d_R);
    // joint
    L[4] = 0.f;
    if (in.joint) {
        corners(c_j, dims, R, P);
        L[4] = in.chamfer_pose ? chamfer(P, G, dP, up[4], BWD) : l1_corner(P, G, dP, up[4], BWD);
        if (BWD) {
            float dc[3] = {0, 0, 0};
            corners_bwd(dP, dims, R, dc, d_dims, d_R);
            d_z += (dc[0] * pa + dc[1] * pb) + dc[2];
            d_cux += dc[0] * z / K4[0];
            d_cuy += dc[1] * z / K4[1];
        }
    }
    if (!BWD) {
#pragma unroll
        for (int k = 0; k < 5; ++k) losses[i * 5 + k] = L[k] * sf;
        float* o = dec + i * 17;
        o[0] = cux; o[1] = cuy; o[2] = z; o[3] = dims[0]; o[4] = dims[1]; o[5] = dims[2];
#pragma unroll
        for (int k = 0; k < 9; ++k) o[6 + k] = R[k];
        o[15] = z * pa; o[16] = z * pb;
    } else {
        g_dxy[i * 2] = d_cux * sw;
        g_dxy[i * 2 + 1] = d_cuy * sh;
        g_zr[i] = d_z * in.v2r[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) g_dr[i * 3 + k] = dclip[k] ? 0.f : d_dims[k] * dims[k];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)          // dRa = M^T dR
                g_Ra[i * 9 + a * 3 + b] = rot ? (M[a] * d_R[b] + M[3 + a] * d_R[3 + b]) + M[6 + a] * d_R[6 + b] : d_R[a * 3 + b];
        float du = 0.f;
        if (in.use_conf) {
#pragma unroll
            for (int k = 0; k < 5; ++k) du -= gl[i * 5 + k] * (L[k] * sf);
        }
        g_u[i] = du;
    }
}

static int cube_args(CubeIn& in, const float* const* p, int64_t n, int allocentric, int chamfer_pose, int use_conf,
                     int joint) {
    for (int k = 0; k < 13; ++k) CR_CHECK_ARG(p[k] != nullptr, "cube_loss: NULL input pointer #%d", k);
    in.dxy = p[0]; in.zr = p[1]; in.dr = p[2]; in.Ra = p[3]; in.u = p[4];
    in.src_boxes = p[5]; in.K4 = p[6]; in.v2r = p[7]; in.prior_mean = p[8];
    in.gt2d = p[9]; in.gtz = p[10]; in.gtdims = p[11]; in.gtR = p[12];
    in.n = (int)n; in.allocentric = allocentric; in.chamfer_pose = chamfer_pose; in.use_conf = use_conf; in.joint = joint;
    return CR_OK;
}

// inputs: HOST array of 13 device pointers in the order of CubeIn.  losses (n,5), dec (n,17).
extern "C" int cr_cube_loss_fwd(cr_ctx* ctx, const float* const* inputs, int64_t n, int allocentric, int chamfer_pose,
                                int use_conf, int joint, float* losses, float* dec) {
    CR_CHECK_ARG(ctx && inputs && n >= 0, "cr_cube_loss_fwd: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(losses && dec, "cr_cube_loss_fwd: NULL output");
    CubeIn in;
    int rc = cube_args(in, inputs, n, allocentric, chamfer_pose, use_conf, joint);
    if (rc) return rc;
    hipLaunchKernelGGL((k_cube_loss<false>), dim3((unsigned)cr_cdiv(n, CH_T)), dim3(CH_T), 0, ctx->stream, in,
                       (const float*)nullptr, losses, dec, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// gl (n,5) = d(total)/d(losses); outputs g_dxy (n,2) g_zr (n) g_dr (n,3) g_Ra (n,9) g_u (n)
extern "C" int cr_cube_loss_bwd(cr_ctx* ctx, const float* const* inputs, int64_t n, int allocentric, int chamfer_pose,
                                int use_conf, int joint, const float* gl, float* g_dxy, float* g_zr, float* g_dr,
                                float* g_Ra, float* g_u) {
    CR_CHECK_ARG(ctx && inputs && n >= 0, "cr_cube_loss_bwd: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(gl && g_dxy && g_zr && g_dr && g_Ra && g_u, "cr_cube_loss_bwd: NULL pointer");
    CubeIn in;
    int rc = cube_args(in, inputs, n, allocentric, chamfer_pose, use_conf, joint);
    if (rc) return rc;
    hipLaunchKernelGGL((k_cube_loss<true>), dim3((unsigned)cr_cdiv(n, CH_T)), dim3(CH_T), 0, ctx->stream, in, gl,
                       (float*)nullptr, (float*)nullptr, g_dxy, g_zr, g_dr, g_Ra, g_u);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Glue around the fused loss on the static-shape training path (ROIHeads3D._forward_cube, roi_heads.py:2237-2679):
//   cr_cube_select       per-RoI gather of the RoI's own class from the fused predictor output (n, 13K) =
//                        [deltas 2K | dims 3K | pose6d 6K | z K | uncert K], 6D -> rotation matrix (Gram-Schmidt,
//                        pytorch3d rotation_6d_to_matrix [third-party, restated]), uncertainty clip(0.01), the matched
//                        ground truth (sanitised on empty slots), camera constants and dimension priors
//   cr_cube_select_bwd   the transpose: dense gradient of the predictor output (zeros off the RoI's class)
//   cr_cube_reduce(_bwd) safely_reduce_losses (roi_heads.py:2843-2851) of the five losses + the uncertainty over the
//                        valid RoIs, optional inverse-z weighting (:2607-2611), and the logged error statistics.
// `buf` (39, n) chunks, each (n,d) contiguous, in CubeIn order without src_boxes:
//   dxy 2 | zr 1 | dr 3 | Ra 9 | u 1 | K4 4 | v2r 1 | prior_mean 3 | gt2d 2 | gtz 1 | gtdims 3 | gtR 9
// ---------------------------------------------------------------------------------------------------------------
struct CubeSel {
    const float* raw; int ld, o_d2, o_dims, o_pose, o_z, o_unc, K;
    const int64_t* cls; const unsigned char* valid; const int64_t* gt_idx;   // (B,S) rows, first kf columns used
    int S, kf, G, n;
    const float* gt3d; const float* gtpose; const float* priors; const float* meta;
    int z_type;          // 0 direct | 1 sigmoid: z = 100 sigmoid(raw) | 2 log: z = exp(raw) | 3 clusters: scaled sigmoid between
                         // mean -+ 3 std of the RoI's depth cluster   (roi_heads.py:2404-2436)
    int bins;            // MODEL.ROI_CUBE_HEAD.CLUSTER_BINS: the depth predictor has K * bins columns laid out [bin][class]
    const float* z_scales;   // (K, bins) 2D-scale centre of every (class, bin)  (priors_z_scales)   -- bins > 1
    const float* z_stats;    // (K, bins, 2) depth mean / std of every (class, bin) (priors_z_stats) -- z_type 3
    const float* boxes;      // (n, 4) proposal boxes: the bin of a RoI is the one whose scale is closest to the box diagonal
};

// depth parametrisation of MODEL.ROI_CUBE_HEAD.Z_TYPE (before the virtual-depth factor) and its derivative
__device__ __forceinline__ float z_decode(float raw, int z_type, float* dz, float mu = 0.f, float sd = 0.f) {
    if (z_type == 1) { const float sg = 1.f / (1.f + expf(-raw)); if (dz) *dz = 100.f * sg * (1.f - sg); return 100.f * sg; }
    if (z_type == 2) { const float e = expf(raw); if (dz) *dz = e; return e; }
    if (z_type == 3) {
        const float mn = fmaxf(mu - 3.f * sd, 0.f), mx = mu + 3.f * sd, sg = 1.f / (1.f + expf(-raw));
        if (dz) *dz = (mx - mn) * sg * (1.f - sg);
        return mn + (mx - mn) * sg;
    }
    if (dz) *dz = 1.f;
    return raw;
}

// cluster bin of RoI i with class c (roi_heads.py:2343-2356): argmin over the bins of |scale[c][b] - box diagonal|, first minimum
__device__ __forceinline__ int z_bin(const float* z_scales, const float* boxes, int bins, int i, int c) {
    if (bins <= 1) return 0;
    const float* b = boxes + (size_t)i * 4;
    const float w = b[2] - b[0], h = b[3] - b[1];
    const float diag = sqrtf(h * h + w * w);
    int best = 0;
    float bd = fabsf(z_scales[c * bins] - diag);
    for (int k = 1; k < bins; ++k) {
        const float d = fabsf(z_scales[c * bins + k] - diag);
        if (d < bd) { bd = d; best = k; }
    }
    return best;
}

// the RoI's depth: column o_z + bin * K + c of the predictor output, decoded
__device__ __forceinline__ float z_of(const float* r, int o_z, int K, int c, int i, int z_type, int bins, const float* z_scales,
                                      const float* z_stats, const float* boxes, float* dz, int* col) {
    const int bin = z_bin(z_scales, boxes, bins, i, c);
    const int cz = o_z + bin * K + c;
    if (col) *col = cz;
    float mu = 0.f, sd = 0.f;
    if (z_type == 3) { mu = z_stats[(c * bins + bin) * 2]; sd = z_stats[(c * bins + bin) * 2 + 1]; }
    return z_decode(r[cz], z_type, dz, mu, sd);
}
__device__ __constant__ int CUBE_OFF[12] = {0, 2, 3, 6, 15, 16, 20, 21, 24, 26, 27, 30};   // chunk starts (x n)

__device__ __forceinline__ void rot6d(const float* a, float* R, float* n1, float* nu) {
    const float l1 = fmaxf(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), 1e-12f);
    const float b1[3] = {a[0] / l1, a[1] / l1, a[2] / l1};
    const float d = b1[0] * a[3] + b1[1] * a[4] + b1[2] * a[5];
    const float u[3] = {a[3] - d * b1[0], a[4] - d * b1[1], a[5] - d * b1[2]};
    const float lu = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / lu, u[1] / lu, u[2] / lu};
    R[0] = b1[0]; R[1] = b1[1]; R[2] = b1[2];
    R[3] = b2[0]; R[4] = b2[1]; R[5] = b2[2];
    R[6] = b1[1] * b2[2] - b1[2] * b2[1]; R[7] = b1[2] * b2[0] - b1[0] * b2[2]; R[8] = b1[0] * b2[1] - b1[1] * b2[0];
    *n1 = l1; *nu = lu;
}

__global__ __launch_bounds__(64) void k_cube_select(CubeSel p, float* __restrict__ buf, unsigned char* __restrict__ validf,
                                                    int* __restrict__ clsc) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= p.n) return;
    const int b = i / p.kf, j = i - b * p.kf;
    const size_t bs = (size_t)b * p.S + j;
    const int64_t c0 = p.cls[bs];
    const bool v = p.valid[bs] && c0 >= 0 && c0 < p.K;
    const int c = (int)(c0 < 0 ? 0 : (c0 >= p.K ? p.K - 1 : c0));
    validf[i] = v ? 1 : 0;
    clsc[i] = c;
    const float* r = p.raw + (size_t)i * p.ld;
    const size_t n = p.n;
    float* o = buf;
    o[CUBE_OFF[0] * n + i * 2] = r[p.o_d2 + c * 2]; o[CUBE_OFF[0] * n + i * 2 + 1] = r[p.o_d2 + c * 2 + 1];
    o[CUBE_OFF[1] * n + i] = z_of(r, p.o_z, p.K, c, i, p.z_type, p.bins, p.z_scales, p.z_stats, p.boxes, nullptr, nullptr);
#pragma unroll
    for (int k = 0; k < 3; ++k) o[CUBE_OFF[2] * n + i * 3 + k] = r[p.o_dims + c * 3 + k];
    float a[6], R[9], l1, lu;
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = r[p.o_pose + c * 6 + k];
    rot6d(a, R, &l1, &lu);
#pragma unroll
    for (int k = 0; k < 9; ++k) o[CUBE_OFF[3] * n + i * 9 + k] = R[k];
    o[CUBE_OFF[4] * n + i] = fmaxf(r[p.o_unc + c], 0.01f);
#pragma unroll
    for (int k = 0; k < 4; ++k) o[CUBE_OFF[5] * n + i * 4 + k] = p.meta[b * 5 + k];
    o[CUBE_OFF[6] * n + i] = p.meta[b * 5 + 4];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[CUBE_OFF[7] * n + i * 3 + k] = p.priors ? p.priors[c * 3 + k] : 1.f;
    const int64_t gi = p.gt_idx[bs];
    const float* g3 = p.gt3d + ((size_t)b * p.G + gi) * 9;
    const float safe[6] = {256.f, 256.f, 5.f, 1.f, 1.f, 1.f};      // a unit cube 5 m in front of the camera
    o[CUBE_OFF[8] * n + i * 2] = v ? g3[0] : safe[0]; o[CUBE_OFF[8] * n + i * 2 + 1] = v ? g3[1] : safe[1];
    o[CUBE_OFF[9] * n + i] = v ? g3[2] : safe[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[CUBE_OFF[10] * n + i * 3 + k] = v ? g3[3 + k] : safe[3 + k];
    const float* gp = p.gtpose + ((size_t)b * p.G + gi) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[CUBE_OFF[11] * n + i * 9 + k] = gp[k];
}

// one wave per RoI: lane 0 back-propagates to the 13 selected predictor outputs, the wave writes the dense row
__global__ __launch_bounds__(64) void k_cube_select_bwd(CubeSel p, const unsigned char* __restrict__ validf,
                                                        const int* __restrict__ clsc, const float* __restrict__ g_dxy,
                                                        const float* __restrict__ g_zr, const float* __restrict__ g_dr,
                                                        const float* __restrict__ g_Ra, const float* __restrict__ g_u,
                                                        const float* __restrict__ g_usel, float* __restrict__ g_raw) {
    __shared__ float s[13];
    const int i = blockIdx.x, lane = threadIdx.x;
    const int c = clsc[i];
    const bool v = validf[i] != 0;
    const float* r = p.raw + (size_t)i * p.ld;
    if (lane == 0) {
        if (!v) {
            for (int k = 0; k < 13; ++k) s[k] = 0.f;
        } else {
            s[0] = g_dxy[i * 2]; s[1] = g_dxy[i * 2 + 1];
            s[2] = g_dr[i * 3]; s[3] = g_dr[i * 3 + 1]; s[4] = g_dr[i * 3 + 2];
            float a[6], R[9], l1, lu;
            for (int k = 0; k < 6; ++k) a[k] = r[p.o_pose + c * 6 + k];
            rot6d(a, R, &l1, &lu);
            const float* gR = g_Ra + (size_t)i * 9;
            const float* b1 = R; const float* b2 = R + 3;
            // b3 = b1 x b2
            float gb1[3] = {gR[0] + (b2[1] * gR[8] - b2[2] * gR[7]), gR[1] + (b2[2] * gR[6] - b2[0] * gR[8]),
                            gR[2] + (b2[0] * gR[7] - b2[1] * gR[6])};
            float gb2[3] = {gR[3] + (gR[7] * b1[2] - gR[8] * b1[1]), gR[4] + (gR[8] * b1[0] - gR[6] * b1[2]),
                            gR[5] + (gR[6] * b1[1] - gR[7] * b1[0])};
            // b2 = u / |u|
            const float dot2 = gb2[0] * b2[0] + gb2[1] * b2[1] + gb2[2] * b2[2];
            float gu[3] = {(gb2[0] - dot2 * b2[0]) / lu, (gb2[1] - dot2 * b2[1]) / lu, (gb2[2] - dot2 * b2[2]) / lu};
            // u = a2 - d b1, d = b1 . a2
            const float d = b1[0] * a[3] + b1[1] * a[4] + b1[2] * a[5];
            const float gd = -(gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2]);
            float ga2[3] = {gu[0] + gd * b1[0], gu[1] + gd * b1[1], gu[2] + gd * b1[2]};
            for (int k = 0; k < 3; ++k) gb1[k] += -d * gu[k] + gd * a[3 + k];
            // b1 = a1 / |a1|
            const float dot1 = gb1[0] * b1[0] + gb1[1] * b1[1] + gb1[2] * b1[2];
            for (int k = 0; k < 3; ++k) { s[5 + k] = (gb1[k] - dot1 * b1[k]) / l1; s[8 + k] = ga2[k]; }
            { float dz; z_of(r, p.o_z, p.K, c, i, p.z_type, p.bins, p.z_scales, p.z_stats, p.boxes, &dz, nullptr); s[11] = g_zr[i] * dz; }
            s[12] = r[p.o_unc + c] >= 0.01f ? g_u[i] + g_usel[i] : 0.f;      // clip(0.01) passes the gradient where raw >= 0.01
        }
    }
    __syncthreads();
    const int zcol = p.o_z + z_bin(p.z_scales, p.boxes, p.bins, i, c) * p.K + c;
    float* g = g_raw + (size_t)i * p.ld;
    for (int col = lane; col < p.ld; col += 64) {
        float val = 0.f;
        int e;
        if ((e = col - (p.o_d2 + c * 2)) >= 0 && e < 2) val = s[e];
        else if ((e = col - (p.o_dims + c * 3)) >= 0 && e < 3) val = s[2 + e];
        else if ((e = col - (p.o_pose + c * 6)) >= 0 && e < 6) val = s[5 + e];
        else if (col == zcol) val = s[11];
        else if (col == p.o_unc + c) val = s[12];
        g[col] = val;
    }
}

static int cube_sel_args(CubeSel& p, const float* raw, int ld, const int* layout5, int K, const int64_t* cls,
                         const unsigned char* valid, const int64_t* gt_idx, int B, int S, int kf, int G, const float* gt3d,
                         const float* gtpose, const float* priors, const float* meta, int z_type, int bins,
                         const float* z_scales, const float* z_stats, const float* boxes) {
    CR_CHECK_ARG(raw && layout5 && cls && valid && gt_idx && gt3d && gtpose && meta, "cube_select: NULL pointer");
    CR_CHECK_ARG(bins >= 1 && B > 0 && kf > 0 && kf <= S && G > 0 && K > 0 && ld >= (12 + bins) * K, "cube_select: bad sizes");
    CR_CHECK_ARG(bins == 1 || (z_scales && boxes), "cube_select: CLUSTER_BINS > 1 needs the scale centres and the RoI boxes");
    CR_CHECK_ARG(z_type != 3 || (bins > 1 && z_stats), "cube_select: Z_TYPE 'clusters' needs CLUSTER_BINS > 1 and the depth statistics");
    p.raw = raw; p.ld = ld; p.o_d2 = layout5[0]; p.o_dims = layout5[1]; p.o_pose = layout5[2]; p.o_z = layout5[3];
    p.o_unc = layout5[4]; p.K = K; p.cls = cls; p.valid = valid; p.gt_idx = gt_idx; p.S = S; p.kf = kf; p.G = G; p.n = B * kf;
    p.gt3d = gt3d; p.gtpose = gtpose; p.priors = priors; p.meta = meta; p.z_type = z_type;
    p.bins = bins; p.z_scales = z_scales; p.z_stats = z_stats; p.boxes = boxes;
    CR_CHECK_ARG(z_type >= 0 && z_type <= 3, "cube_select: z_type %d (0 direct, 1 sigmoid, 2 log, 3 clusters)", z_type);
    return CR_OK;
}

extern "C" int cr_cube_select(cr_ctx* ctx, const float* raw, int ld, const int* layout5, int K, const int64_t* cls,
                              const unsigned char* valid, const int64_t* gt_idx, int B, int S, int kf, int G,
                              const float* gt3d, const float* gtpose, const float* priors, const float* meta, float* buf39,
                              unsigned char* validf, int* clsc, int z_type, int bins, const float* z_scales,
                              const float* z_stats, const float* boxes) {
    CR_CHECK_ARG(ctx && buf39 && validf && clsc, "cr_cube_select: NULL pointer");
    CubeSel p;
    int rc = cube_sel_args(p, raw, ld, layout5, K, cls, valid, gt_idx, B, S, kf, G, gt3d, gtpose, priors, meta, z_type, bins, z_scales,
                           z_stats, boxes);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cube_select, dim3((unsigned)cr_cdiv(p.n, 64)), dim3(64), 0, ctx->stream, p, buf39, validf, clsc);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_cube_select_bwd(cr_ctx* ctx, const float* raw, int ld, const int* layout5, int K, int B, int kf,
                                  const unsigned char* validf, const int* clsc, const float* g_dxy, const float* g_zr,
                                  const float* g_dr, const float* g_Ra, const float* g_u, const float* g_usel, float* g_raw,
                                  int z_type, int bins, const float* z_scales, const float* z_stats, const float* boxes) {
    CR_CHECK_ARG(ctx && raw && layout5 && validf && clsc && g_dxy && g_zr && g_dr && g_Ra && g_u && g_usel && g_raw,
                 "cr_cube_select_bwd: NULL pointer");
    CR_CHECK_ARG(z_type >= 0 && z_type <= 3 && bins >= 1 && (bins == 1 || (z_scales && boxes)) && (z_type != 3 || z_stats),
                 "cr_cube_select_bwd: z_type %d / bins %d", z_type, bins);
    CubeSel p = {};
    p.z_type = z_type; p.bins = bins; p.z_scales = z_scales; p.z_stats = z_stats; p.boxes = boxes;
    p.raw = raw; p.ld = ld; p.o_d2 = layout5[0]; p.o_dims = layout5[1]; p.o_pose = layout5[2]; p.o_z = layout5[3];
    p.o_unc = layout5[4]; p.K = K; p.kf = kf; p.n = B * kf;
    if (p.n == 0) return CR_OK;
    hipLaunchKernelGGL(k_cube_select_bwd, dim3(p.n), dim3(64), 0, ctx->stream, p, validf, clsc, g_dxy, g_zr, g_dr, g_Ra, g_u,
                       g_usel, g_raw);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// red (6) = mean over the valid, finite entries of [dims, xy, z, pose, joint] (x inverse-z weight) and of the uncertainty
// (0 when there is none); cnt (6); stats (4) = mean |z|, |dims|, |xy| errors and mean exp(-u) over the valid RoIs.
__global__ __launch_bounds__(256) void k_cube_reduce(const float* __restrict__ L, const float* __restrict__ buf,
                                                     const float* __restrict__ dec, const unsigned char* __restrict__ validf,
                                                     int n, int inverse_z, float* __restrict__ red, float* __restrict__ cnt,
                                                     float* __restrict__ stats) {
    __shared__ float sm[256][17];
    const int t = threadIdx.x;
    float acc[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) acc[k] = 0.f;
    const size_t N = n;
    for (int i = t; i < n; i += 256) {
        if (!validf[i]) continue;
        const float gz = buf[CUBE_OFF[9] * N + i], u = buf[CUBE_OFF[4] * N + i];
        const float w = inverse_z ? 1.f / logf(fmaxf(gz, 2.71828183f)) : 1.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float x = L[(size_t)i * 5 + k] * w;
            if (isfinite(x)) { acc[k] += x; acc[6 + k] += 1.f; }
        }
        if (isfinite(u)) { acc[5] += u; acc[11] += 1.f; }
        acc[12] += fabsf(dec[(size_t)i * 17 + 2] - gz);
        acc[13] += (fabsf(dec[(size_t)i * 17 + 3] - buf[CUBE_OFF[10] * N + i * 3]) +
                    fabsf(dec[(size_t)i * 17 + 4] - buf[CUBE_OFF[10] * N + i * 3 + 1])) +
                   fabsf(dec[(size_t)i * 17 + 5] - buf[CUBE_OFF[10] * N + i * 3 + 2]);
        acc[14] += fabsf(dec[(size_t)i * 17] - buf[CUBE_OFF[8] * N + i * 2]) + fabsf(dec[(size_t)i * 17 + 1] - buf[CUBE_OFF[8] * N + i * 2 + 1]);
        acc[15] += expf(-u);
        acc[16] += 1.f;
    }
#pragma unroll
    for (int k = 0; k < 17; ++k) sm[t][k] = acc[k];
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {                 // fixed-order tree: bitwise reproducible
        if (t < s) {
#pragma unroll
            for (int k = 0; k < 17; ++k) sm[t][k] += sm[t + s][k];
        }
        __syncthreads();
    }
    if (t < 6) {
        const float c = sm[0][6 + t];
        red[t] = c > 0.f ? sm[0][t] / c : 0.f;
        cnt[t] = c;
    }
    if (t == 0) {
        const float nv = fmaxf(sm[0][16], 1.f);
        stats[0] = sm[0][12] / nv; stats[1] = sm[0][13] / (3.f * nv); stats[2] = sm[0][14] / (2.f * nv); stats[3] = sm[0][15] / nv;
    }
}

__global__ __launch_bounds__(256) void k_cube_reduce_bwd(const float* __restrict__ L, const float* __restrict__ buf,
                                                         const unsigned char* __restrict__ validf, int n, int inverse_z,
                                                         const float* __restrict__ cnt, const float* __restrict__ gred,
                                                         float* __restrict__ gL, float* __restrict__ gu) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t N = n;
    const bool v = validf[i] != 0;
    const float gz = buf[CUBE_OFF[9] * N + i], u = buf[CUBE_OFF[4] * N + i];
    const float w = inverse_z ? 1.f / logf(fmaxf(gz, 2.71828183f)) : 1.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float x = L[(size_t)i * 5 + k] * w;
        gL[(size_t)i * 5 + k] = (v && isfinite(x) && cnt[k] > 0.f) ? gred[k] * w / cnt[k] : 0.f;
    }
    gu[i] = (v && isfinite(u) && cnt[5] > 0.f) ? gred[5] / cnt[5] : 0.f;
}

extern "C" int cr_cube_reduce(cr_ctx* ctx, const float* L, const float* buf39, const float* dec, const unsigned char* validf,
                              int n, int inverse_z, float* red6, float* cnt6, float* stats4) {
    CR_CHECK_ARG(ctx && L && buf39 && dec && validf && red6 && cnt6 && stats4 && n > 0, "cr_cube_reduce: bad args");
    hipLaunchKernelGGL(k_cube_reduce, dim3(1), dim3(256), 0, ctx->stream, L, buf39, dec, validf, n, inverse_z, red6, cnt6, stats4);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_cube_reduce_bwd(cr_ctx* ctx, const float* L, const float* buf39, const unsigned char* validf, int n,
                                  int inverse_z, const float* cnt6, const float* gred6, float* gL, float* gu) {
    CR_CHECK_ARG(ctx && L && buf39 && validf && cnt6 && gred6 && gL && gu && n > 0, "cr_cube_reduce_bwd: bad args");
    hipLaunchKernelGGL(k_cube_reduce_bwd, dim3((unsigned)cr_cdiv(n, 256)), dim3(256), 0, ctx->stream, L, buf39, validf, n,
                       inverse_z, cnt6, gred6, gL, gu);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Inference: decode of the 3D head for the detections kept by the box head (roi_heads.py:2353-2436, 2682-2735), one
// lane per detection, straight from the fused predictor output: class gather, 6D -> R, allocentric -> egocentric,
// virtual depth, exp dims priors, back-projection of the centre, confidence, and the 8 corners (corners(): the
// get_cuboid_verts_faces order of math_util.py:198-207).
//   meta (B,6) = [fx, fy, cx, cy of K / ratio, virtual_to_real, ratio]; img (n) int32 image of each detection
//   out (n,42) = [x3d, y3d, z | w, h, l | cx2d*ratio, cy2d*ratio | conf | R (9) | corners (8x3)]
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_cube_decode_infer(const float* __restrict__ raw, int ld, int o_d2, int o_dims, int o_pose,
                                                          int o_z, int o_unc, int K, const int64_t* __restrict__ cls,
                                                          const int* __restrict__ img, const float* __restrict__ boxes,
                                                          const float* __restrict__ meta, const float* __restrict__ priors,
                                                          int n, int allocentric, float* __restrict__ out, int z_type, int bins,
                                                          const float* __restrict__ z_scales, const float* __restrict__ z_stats) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int64_t c0 = cls[i];
    const int c = (int)(c0 < 0 ? 0 : (c0 >= K ? K - 1 : c0));
    const float* r = raw + (size_t)i * ld;
    const float* m = meta + (size_t)img[i] * 6;
    const float K4[4] = {m[0], m[1], m[2], m[3]};
    const float* sb = boxes + (size_t)i * 4;
    const float sw = sb[2] - sb[0], sh = sb[3] - sb[1];
    const float cux = (sb[0] + 0.5f * sw) + sw * r[o_d2 + c * 2], cuy = (sb[1] + 0.5f * sh) + sh * r[o_d2 + c * 2 + 1];
    float dims[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dims[k] = expf(fminf(r[o_dims + c * 3 + k], 5.0f)) * (priors ? priors[c * 3 + k] : 1.f);
    float a[6], Ra[9], R[9], M[9], l1, lu;
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = r[o_pose + c * 6 + k];
    rot6d(a, Ra, &l1, &lu);
    const bool rot = allocentric ? ray_rotation(cux, cuy, K4, M) : false;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            R[p * 3 + q] = rot ? (M[p * 3] * Ra[q] + M[p * 3 + 1] * Ra[3 + q]) + M[p * 3 + 2] * Ra[6 + q] : Ra[p * 3 + q];
    const float z = z_of(r, o_z, K, c, i, z_type, bins, z_scales, z_stats, boxes, nullptr, nullptr) * m[4];
    const float ctr[3] = {z * (cux - K4[2]) / K4[0], z * (cuy - K4[3]) / K4[1], z};
    float* o = out + (size_t)i * 42;
    o[0] = ctr[0]; o[1] = ctr[1]; o[2] = ctr[2];
    o[3] = dims[0]; o[4] = dims[1]; o[5] = dims[2];
    o[6] = cux * m[5]; o[7] = cuy * m[5];
    o[8] = expf(-fmaxf(r[o_unc + c], 0.01f));
#pragma unroll
    for (int k = 0; k < 9; ++k) o[9 + k] = R[k];
    float P[8][3];
    corners(ctr, dims, R, P);
#pragma unroll
    for (int v = 0; v < 8; ++v) { o[18 + v * 3] = P[v][0]; o[19 + v * 3] = P[v][1]; o[20 + v * 3] = P[v][2]; }
}

extern "C" int cr_cube_decode_infer(cr_ctx* ctx, const float* raw, int ld, const int* layout5, int K, const int64_t* cls,
                                    const int* img, const float* boxes, const float* meta6, const float* priors, int n,
                                    int allocentric, float* out42, int z_type, int bins, const float* z_scales,
                                    const float* z_stats) {
    CR_CHECK_ARG(ctx && n >= 0 && z_type >= 0 && z_type <= 3 && bins >= 1, "cr_cube_decode_infer: bad args");
    CR_CHECK_ARG((bins == 1 || z_scales) && (z_type != 3 || (bins > 1 && z_stats)), "cr_cube_decode_infer: cluster tables missing");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(raw && layout5 && cls && img && boxes && meta6 && out42 && K > 0 && ld >= (12 + bins) * K, "cr_cube_decode_infer: bad args");
    hipLaunchKernelGGL(k_cube_decode_infer, dim3((unsigned)cr_cdiv(n, 64)), dim3(64), 0, ctx->stream, raw, ld, layout5[0],
                       layout5[1], layout5[2], layout5[3], layout5[4], K, cls, img, boxes, meta6, priors, n, allocentric, out42, z_type, bins, z_scales, z_stats);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
