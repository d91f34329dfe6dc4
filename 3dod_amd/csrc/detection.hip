// Detection-side kernels: ROIAlign (aligned, adaptive sampling) forward/backward over the FPN
// pyramid with the level assignment fused in, grouped NMS, and RPN anchor decoding.
//
// Reference call sites (paths into the reference tree); the arithmetic itself lives in
// third-party code that is absent from the reference and is restated from its published
// definition (parity unpinned, pinned here by a torch restatement in oracle/):
//   ROIPooler(ROIAlignV2 7x7, sampling_ratio 0)  cubercnn/modeling/roi_heads/roi_heads.py:2075-2080,2178,2273
//        = torchvision roi_align(aligned=True) + detectron2 assign_boxes_to_levels
//   nms / batched_nms                              cubercnn/modeling/roi_heads/fast_rcnn.py:105; detectron2 RPN
#include "cr_common.h"
#include <math.h>
#include <stdlib.h>

#include "cr_elem.h"

#define MAX_LEVELS 5
struct Pyramid {
    const void* feat[MAX_LEVELS];  // NHWC bf16 or f32 (template parameter of the kernels)
    float* grad[MAX_LEVELS];       // NHWC f32 (backward)
    int H[MAX_LEVELS], W[MAX_LEVELS];
    float scale[MAX_LEVELS];
    int nlev, C, min_level;        // min_level = log2(stride of level 0)
};

// detectron2 assign_boxes_to_levels: floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to the pyramid
__device__ __forceinline__ int roi_level(const float* b, const Pyramid& py) {
    const float area = (b[2] - b[0]) * (b[3] - b[1]);
    float lv = floorf(4.0f + log2f(sqrtf(area) / 224.0f + 1e-8f));
    const float lo = (float)py.min_level, hi = (float)(py.min_level + py.nlev - 1);
    lv = fminf(fmaxf(lv, lo), hi);
    return (int)lv - py.min_level;
}

struct Samp { int yl, yh, xl, xh; float w1, w2, w3, w4; bool ok; };
__device__ __forceinline__ Samp bilinear(float y, float x, int H, int W) {
    Samp s;
    s.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) && (y == y) && (x == x);   // NaN boxes sample nothing
    if (!s.ok) { s.yl = s.yh = s.xl = s.xh = 0; s.w1 = s.w2 = s.w3 = s.w4 = 0.f; return s; }
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    s.yl = (int)y; s.xl = (int)x;
    if (s.yl >= H - 1) { s.yh = s.yl = H - 1; y = (float)s.yl; } else s.yh = s.yl + 1;
    if (s.xl >= W - 1) { s.xh = s.xl = W - 1; x = (float)s.xl; } else s.xh = s.xl + 1;
    const float ly = y - s.yl, lx = x - s.xl, hy = 1.f - ly, hx = 1.f - lx;
    s.w1 = hy * hx; s.w2 = hy * lx; s.w3 = ly * hx; s.w4 = ly * lx;
    return s;
}

// one thread = (roi, ph, pw, 8 channels).  rois (R,5) = [batch, x1,y1,x2,y2].  out (R,PH,PW,C) in the storage type T
template <bool BWD, typename T>
__global__ __launch_bounds__(256) void k_roi_align(Pyramid py, const float* __restrict__ rois, int R, int PH, int PW,
                                                   T* __restrict__ out, const T* __restrict__ dout) {
    const int cg = py.C >> 3;
    const int64_t total = (int64_t)R * PH * PW * cg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg);
    const int pw = (int)((i / cg) % PW);
    const int ph = (int)((i / ((int64_t)cg * PW)) % PH);
    const int r = (int)(i / ((int64_t)cg * PW * PH));
    const float* rb = rois + (size_t)r * 5;
    const int n = (int)rb[0];
    const int lv = roi_level(rb + 1, py);
    const int H = py.H[lv], W = py.W[lv];
    const float sc = py.scale[lv];
    const float x1 = rb[1] * sc - 0.5f, y1 = rb[2] * sc - 0.5f;
    const float rw = (rb[3] - rb[1]) * sc, rh = (rb[4] - rb[2]) * sc;     // aligned: no 1-px floor
    const float bw = rw / (float)PW, bh = rh / (float)PH;
    const int gh = (int)ceilf(rh / (float)PH), gw = (int)ceilf(rw / (float)PW);
    const float cnt = fmaxf((float)(gh * gw), 1.f);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float g8[8];
    if (BWD) {
        load8<T>(dout, (size_t)i * 8, g8);
#pragma unroll
        for (int e = 0; e < 8; ++e) g8[e] /= cnt;
    }
    const size_t img = (size_t)n * H * W;
    for (int iy = 0; iy < gh; ++iy) {
        const float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
            const float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
            const Samp s = bilinear(y, x, H, W);
            if (!s.ok) continue;
            const size_t o1 = (img + (size_t)s.yl * W + s.xl) * py.C + c * 8, o2 = (img + (size_t)s.yl * W + s.xh) * py.C + c * 8;
            const size_t o3 = (img + (size_t)s.yh * W + s.xl) * py.C + c * 8, o4 = (img + (size_t)s.yh * W + s.xh) * py.C + c * 8;
            if (!BWD) {
                const T* f = (const T*)py.feat[lv];
                float a1[8], a2[8], a3[8], a4[8];
                load8<T>(f, o1, a1); load8<T>(f, o2, a2); load8<T>(f, o3, a3); load8<T>(f, o4, a4);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += s.w1 * a1[e] + s.w2 * a2[e] + s.w3 * a3[e] + s.w4 * a4[e];
            } else {
                float* gq = py.grad[lv];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    atomicAdd(gq + o1 + e, g8[e] * s.w1);
                    atomicAdd(gq + o2 + e, g8[e] * s.w2);
                    atomicAdd(gq + o3 + e, g8[e] * s.w3);
                    atomicAdd(gq + o4 + e, g8[e] * s.w4);
                }
            }
        }
    }
    if (!BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] /= cnt;
        store8<T>(out, (size_t)i * 8, acc);
    }
}

// backward: one thread = (roi, ph, pw, ONE channel), channel fastest, so every atomic wave-instruction adds
// 64 consecutive floats (256 contiguous bytes: the full-rate shape of MI355X_MICROARCH.md "Global float atomics").
template <typename T>
__global__ __launch_bounds__(256) void k_roi_align_bwd(Pyramid py, const float* __restrict__ rois, int R, int PH,
                                                       int PW, const T* __restrict__ dout) {
    const int C = py.C;
    const int64_t total = (int64_t)R * PH * PW * C;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int pw = (int)((i / C) % PW);
    const int ph = (int)((i / ((int64_t)C * PW)) % PH);
    const int r = (int)(i / ((int64_t)C * PW * PH));
    const float* rb = rois + (size_t)r * 5;
    const int n = (int)rb[0];
    const int lv = roi_level(rb + 1, py);
    const int H = py.H[lv], W = py.W[lv];
    const float sc = py.scale[lv];
    const float x1 = rb[1] * sc - 0.5f, y1 = rb[2] * sc - 0.5f;
    const float rw = (rb[3] - rb[1]) * sc, rh = (rb[4] - rb[2]) * sc;
    const float bw = rw / (float)PW, bh = rh / (float)PH;
    const int gh = (int)ceilf(rh / (float)PH), gw = (int)ceilf(rw / (float)PW);
    const float cnt = fmaxf((float)(gh * gw), 1.f);
    const float g = load1<T>(dout, (size_t)i) / cnt;
    if (g == 0.f) return;        // masked-out (padding) RoIs and dead channels add nothing: skip their 16 atomics
    float* gq = py.grad[lv] + (size_t)n * H * W * C + c;
    for (int iy = 0; iy < gh; ++iy) {
        const float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
            const float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
            const Samp s = bilinear(y, x, H, W);
            if (!s.ok) continue;
            atomicAdd(gq + ((size_t)s.yl * W + s.xl) * C, g * s.w1);
            atomicAdd(gq + ((size_t)s.yl * W + s.xh) * C, g * s.w2);
            atomicAdd(gq + ((size_t)s.yh * W + s.xl) * C, g * s.w3);
            atomicAdd(gq + ((size_t)s.yh * W + s.xh) * C, g * s.w4);
        }
    }
}

// Separable backward.  A sample's bilinear weight factors into a row part and a column part, so the gradient of one
// RoI w.r.t. its footprint of feature pixels is  Ay . G . Ax^T  per channel, with
//   Ay[y][ph] = sum over the gh samples of bin-row ph of their weight on feature row y, / gh      (Ax likewise),
//   G[ph][pw] = d(out)[r][ph][pw][c].
// One block per RoI builds Ay/Ax in LDS once; a thread owns one channel, keeps G in registers and issues ONE atomic per
// footprint pixel instead of 4 per sample (3-4x fewer atomics for the 2x2..4x4 sampling grids of FPN RoIs), still
// 64 consecutive floats per wave-instruction.
struct Lin1 { int lo, hi; float wlo, whi; bool ok; };
__device__ __forceinline__ Lin1 lin1(float y, int H) {
    Lin1 s;
    s.ok = (y >= -1.0f) && (y <= (float)H);             // false for NaN
    if (y <= 0.f) y = 0.f;
    s.lo = s.ok ? (int)y : 0;
    if (s.lo >= H - 1) { s.hi = s.lo = H - 1; y = (float)s.lo; } else s.hi = s.lo + 1;
    const float l = y - (float)s.lo;
    s.wlo = 1.f - l; s.whi = l;
    return s;
}

// XG = 8: the block index is (RoI, channel group blockIdx % 8) and a block covers C / 8 channels.  Workgroups are dealt
// round-robin to the 8 XCDs, so group x runs on XCD x and every 128-B line of the gradient maps (32 f32 channels of a pixel)
// is only ever added to from ONE XCD: the atomics stay in that XCD's L2 instead of the line migrating between the L2s of
// all the XCDs whose RoIs overlap there.  (Performance only: the L2s are coherent, any other dispatch order gives the same
// sums.)  The block's threads split the footprint rows: thread = (row phase t / CB, channel t % CB).
template <int P, typename T, int XG>
__global__ __launch_bounds__(256) void k_roi_align_bwd_sep(Pyramid py, const float* __restrict__ rois, int R,
                                                           const T* __restrict__ dout, int maxH) {
    extern __shared__ float sm[];                        // Ay [maxH][P] | Ax [maxW][P]
    __shared__ int s_lo[2], s_hi[2];
    float* Ay = sm;
    float* Ax = sm + (size_t)maxH * P;
    const int r = blockIdx.x / XG, t = threadIdx.x, C = py.C;
    const int CB = C / XG, cbase = (blockIdx.x % XG) * CB;   // this block's channels
    const int nph = XG == 1 ? 1 : (int)blockDim.x / CB;      // row phases (XG = 1: a thread walks every footprint row)
    const float* rb = rois + (size_t)r * 5;
    const int n = (int)rb[0];
    const int lv = roi_level(rb + 1, py);
    const int H = py.H[lv], W = py.W[lv];
    const float sc = py.scale[lv];
    const float x1 = rb[1] * sc - 0.5f, y1 = rb[2] * sc - 0.5f;
    const float rw = (rb[3] - rb[1]) * sc, rh = (rb[4] - rb[2]) * sc;
    const float bw = rw / (float)P, bh = rh / (float)P;
    const int gh = min((int)ceilf(rh / (float)P), 4096), gw = min((int)ceilf(rw / (float)P), 4096);
    if (t < 2) { s_lo[t] = 0x7fffffff; s_hi[t] = -1; }
    __syncthreads();
    // pass 1: footprint extent per axis (threads 0..P-1 rows, P..2P-1 columns)
    const bool builder = t < 2 * P;
    const int ax = t / P, p = t % P;                     // ax 0 = y, 1 = x
    const int gN = ax ? gw : gh, L = ax ? W : H;
    const float o1 = ax ? x1 : y1, bsz = ax ? bw : bh;
    if (builder) {
        int lo = 0x7fffffff, hi = -1;
        for (int i = 0; i < gN; ++i) {
            const Lin1 s = lin1(o1 + p * bsz + (i + 0.5f) * bsz / (float)gN, L);
            if (s.ok) { lo = min(lo, s.lo); hi = max(hi, s.hi); }
        }
        if (hi >= 0) { atomicMin(&s_lo[ax], lo); atomicMax(&s_hi[ax], hi); }
    }
    __syncthreads();
    const int y0 = s_lo[0], Py = s_hi[0] - y0 + 1, x0 = s_lo[1], Px = s_hi[1] - x0 + 1;
    if (s_hi[0] < 0 || s_hi[1] < 0 || n < 0) return;     // no valid sample at all (block-uniform)
    for (int i = t; i < Py * P; i += (int)blockDim.x) Ay[i] = 0.f;
    for (int i = t; i < Px * P; i += (int)blockDim.x) Ax[i] = 0.f;
    __syncthreads();
    if (builder) {                                       // pass 2: column p of Ay / Ax is owned by one thread
        float* A = ax ? Ax : Ay;
        const int base = ax ? x0 : y0;
        const float inv = 1.f / (float)gN;
        for (int i = 0; i < gN; ++i) {
            const Lin1 s = lin1(o1 + p * bsz + (i + 0.5f) * bsz / (float)gN, L);
            if (!s.ok) continue;
            A[(s.lo - base) * P + p] += s.wlo * inv;
            A[(s.hi - base) * P + p] += s.whi * inv;
        }
    }
    __syncthreads();
    for (int c = cbase + (XG == 1 ? t : t % CB); c < cbase + CB; c += (XG == 1 ? 256 : CB)) {
        const int ph0 = XG == 1 ? 0 : t / CB;
        float G[P][P];
        bool any = false;
#pragma unroll
        for (int ph = 0; ph < P; ++ph)
#pragma unroll
            for (int pw = 0; pw < P; ++pw) {
                G[ph][pw] = load1<T>(dout, (((size_t)r * P + ph) * P + pw) * C + c);
                any |= G[ph][pw] != 0.f;
            }
        if (!any) continue;                              // masked (padding) RoIs carry zero gradient
        float* gq = py.grad[lv] + (((size_t)n * H + y0) * W + x0) * C + c;
        for (int yy = ph0; yy < Py; yy += nph) {
            float tr[P];
#pragma unroll
            for (int pw = 0; pw < P; ++pw) tr[pw] = 0.f;
#pragma unroll
            for (int ph = 0; ph < P; ++ph) {
                const float a = Ay[yy * P + ph];         // LDS broadcast: uniform
                if (a != 0.f) {
#pragma unroll
                    for (int pw = 0; pw < P; ++pw) tr[pw] += a * G[ph][pw];
                }
            }
            float* grow = gq + (size_t)yy * W * C;
            for (int xx = 0; xx < Px; ++xx) {
                float v = 0.f;
#pragma unroll
                for (int pw = 0; pw < P; ++pw) v += Ax[xx * P + pw] * tr[pw];
                if (v != 0.f) atomicAdd(grow + (size_t)xx * C, v);
            }
        }
    }
}

static int fill_pyramid(Pyramid& py, const void* const* feats, float* const* grads, const int* Hs, const int* Ws,
                        const float* scales, int nlev, int C) {
    CR_CHECK_ARG(nlev >= 1 && nlev <= MAX_LEVELS, "roi_align: 1..%d levels", MAX_LEVELS);
    CR_CHECK_ARG(C % 8 == 0, "roi_align: C %% 8");
    py.nlev = nlev; py.C = C;
    for (int l = 0; l < nlev; ++l) {
        py.feat[l] = feats ? feats[l] : nullptr;
        py.grad[l] = grads ? grads[l] : nullptr;
        py.H[l] = Hs[l]; py.W[l] = Ws[l]; py.scale[l] = scales[l];
    }
    py.min_level = (int)lroundf(-log2f(scales[0]));
    return CR_OK;
}

// feats/Hs/Ws/scales are HOST arrays of length nlev (device pointers inside feats)
extern "C" int cr_roi_align_fwd(cr_ctx* ctx, const void* const* feats, const int* Hs, const int* Ws,
                                const float* scales, int nlev, int C, const float* rois, int64_t R, int PH, int PW,
                                void* out, int act_f32) {
    CR_CHECK_ARG(ctx && feats && Hs && Ws && scales, "cr_roi_align_fwd: NULL pointer");
    if (R == 0) return CR_OK;
    CR_CHECK_ARG(rois && out && PH > 0 && PW > 0, "cr_roi_align_fwd: bad args");
    Pyramid py;
    int rc = fill_pyramid(py, feats, nullptr, Hs, Ws, scales, nlev, C);
    if (rc) return rc;
    const int64_t total = R * PH * PW * (C / 8);
    if (act_f32)
        hipLaunchKernelGGL((k_roi_align<false, float>), dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, py,
                           rois, (int)R, PH, PW, (float*)out, (const float*)nullptr);
    else
        hipLaunchKernelGGL((k_roi_align<false, u16>), dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, py,
                           rois, (int)R, PH, PW, (u16*)out, (const u16*)nullptr);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// grads: HOST array of nlev device pointers to f32 NHWC maps (accumulated with atomics; zero them first)
extern "C" int cr_roi_align_bwd(cr_ctx* ctx, float* const* grads, const int* Hs, const int* Ws, const float* scales,
                                int nlev, int C, const float* rois, int64_t R, int PH, int PW, const void* dout,
                                int act_f32) {
    CR_CHECK_ARG(ctx && grads && Hs && Ws && scales, "cr_roi_align_bwd: NULL pointer");
    if (R == 0) return CR_OK;
    CR_CHECK_ARG(rois && dout && PH > 0 && PW > 0, "cr_roi_align_bwd: bad args");
    Pyramid py;
    int rc = fill_pyramid(py, nullptr, grads, Hs, Ws, scales, nlev, C);
    if (rc) return rc;
    if (PH == 7 && PW == 7 && !getenv("CR_ROI_BWD_PLAIN")) {
        int maxH = 0, maxW = 0;
        for (int l = 0; l < nlev; ++l) { maxH = Hs[l] > maxH ? Hs[l] : maxH; maxW = Ws[l] > maxW ? Ws[l] : maxW; }
        const size_t lds = (size_t)(maxH + maxW) * 7 * sizeof(float);
        if (lds <= 60 * 1024) {
            static const int xg_on = getenv("CR_ROI_BWD_XCD") ? atoi(getenv("CR_ROI_BWD_XCD")) : 1;
            static const int xg_threads = getenv("CR_ROI_BWD_T") ? atoi(getenv("CR_ROI_BWD_T")) : 128;   // 4 row phases x 32 channels
            if (xg_on && C % 256 == 0 && R * 8 < 0x7fffffff) {       // C / 8 channels per block, a multiple of a 128-B line
                const int nt = (C / 8) * (xg_threads / (C / 8) > 0 ? xg_threads / (C / 8) : 1);
                if (act_f32)
                    hipLaunchKernelGGL((k_roi_align_bwd_sep<7, float, 8>), dim3((unsigned)R * 8), dim3(nt), lds, ctx->stream, py,
                                       rois, (int)R, (const float*)dout, maxH);
                else
                    hipLaunchKernelGGL((k_roi_align_bwd_sep<7, u16, 8>), dim3((unsigned)R * 8), dim3(nt), lds, ctx->stream, py,
                                       rois, (int)R, (const u16*)dout, maxH);
            } else if (act_f32)
                hipLaunchKernelGGL((k_roi_align_bwd_sep<7, float, 1>), dim3((unsigned)R), dim3(256), lds, ctx->stream, py, rois,
                                   (int)R, (const float*)dout, maxH);
            else
                hipLaunchKernelGGL((k_roi_align_bwd_sep<7, u16, 1>), dim3((unsigned)R), dim3(256), lds, ctx->stream, py, rois,
                                   (int)R, (const u16*)dout, maxH);
            CR_LAUNCH_CHECK();
            return CR_OK;
        }
    }
    const int64_t total = R * PH * PW * (int64_t)C;
    CR_CHECK_ARG(cr_cdiv(total, 256) < 0x7fffffff, "cr_roi_align_bwd: too many RoIs");
    if (act_f32)
        hipLaunchKernelGGL(k_roi_align_bwd<float>, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, py, rois,
                           (int)R, PH, PW, (const float*)dout);
    else
        hipLaunchKernelGGL(k_roi_align_bwd<u16>, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, py, rois,
                           (int)R, PH, PW, (const u16*)dout);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// grouped NMS.  boxes [G][maxn][4] sorted by descending score inside each group, counts [G].
// keep [G][maxn] uint8.  Standard bitmask formulation: pass 1 builds the suppression matrix in
// parallel, pass 2 is one wave per group walking the rows in order (rows prefetched).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_nms_mask(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                 int maxn, float thresh, unsigned long long* __restrict__ mask) {
    const int g = blockIdx.z, n = counts[g];
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    if (row0 >= n || col0 >= n) return;
    if (col0 + 63 < row0) return;                       // only j > i matters
    __shared__ float sb[64 * 4];
    const float* gb = boxes + (size_t)g * maxn * 4;
    const int t = threadIdx.x;
    if (col0 + t < n) {
        sb[t * 4 + 0] = gb[(col0 + t) * 4 + 0]; sb[t * 4 + 1] = gb[(col0 + t) * 4 + 1];
        sb[t * 4 + 2] = gb[(col0 + t) * 4 + 2]; sb[t * 4 + 3] = gb[(col0 + t) * 4 + 3];
    }
    __syncthreads();
    const int i = row0 + t;
    if (i >= n) return;
    const float ax1 = gb[i * 4], ay1 = gb[i * 4 + 1], ax2 = gb[i * 4 + 2], ay2 = gb[i * 4 + 3];
    const float aa = (ax2 - ax1) * (ay2 - ay1);
    unsigned long long bits = 0;
    const int jn = min(64, n - col0);
    for (int j = 0; j < jn; ++j) {
        if (col0 + j <= i) continue;
        const float bx1 = sb[j * 4], by1 = sb[j * 4 + 1], bx2 = sb[j * 4 + 2], by2 = sb[j * 4 + 3];
        const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
        const float inter = w * h, ba = (bx2 - bx1) * (by2 - by1);
        if (inter / (aa + ba - inter) > thresh) bits |= 1ULL << j;
    }
    const int words = (maxn + 63) / 64;
    mask[((size_t)g * maxn + i) * words + blockIdx.x] = bits;
}

__global__ __launch_bounds__(64) void k_nms_scan(const int* __restrict__ counts, int maxn,
                                                 const unsigned long long* __restrict__ mask,
                                                 unsigned char* __restrict__ keep) {
    extern __shared__ unsigned long long removed[];     // words
    const int g = blockIdx.x, n = counts[g], t = threadIdx.x;
    const int words = (maxn + 63) / 64, nw = (n + 63) / 64;
    for (int w = t; w < words; w += 64) removed[w] = 0;
    __syncthreads();
    const unsigned long long* gm = mask + (size_t)g * maxn * words;
    unsigned char* gk = keep + (size_t)g * maxn;
    for (int i = 0; i < n; ++i) {
        const bool dead = (removed[i >> 6] >> (i & 63)) & 1ULL;     // uniform
        if (t == 0) gk[i] = dead ? 0 : 1;
        if (!dead) {
            // words left of the diagonal block were never written: start at i>>6
            for (int w = (i >> 6) + t; w < nw; w += 64) removed[w] |= gm[(size_t)i * words + w];
        }
        __syncthreads();
    }
    for (int i = n + t; i < maxn; i += 64) gk[i] = 0;
}

// Fast walk for maxn <= 2048 (words <= 32): 256 threads = 8 row-slots x 32 words.  Per 64-box chunk the rows are
// prefetched into registers while wave 0 resolves the diagonal block with register-only readlane steps, so no
// global-memory latency sits on the serial chain (the plain kernel above pays one load per box).
__global__ __launch_bounds__(256) void k_nms_scan32(const int* __restrict__ counts, int maxn,
                                                    const unsigned long long* __restrict__ mask,
                                                    unsigned char* __restrict__ keep) {
    __shared__ unsigned long long removed[32];
    __shared__ unsigned long long diag[64];
    __shared__ unsigned long long s_alive;
    const int g = blockIdx.x, n = counts[g], t = threadIdx.x;
    const int words = (maxn + 63) / 64, nch = (n + 63) / 64;
    const int w = t & 31, bs = t >> 5;
    if (t < 32) removed[t] = 0;
    const unsigned long long* gm = mask + (size_t)g * maxn * words;
    unsigned char* gk = keep + (size_t)g * maxn;
    unsigned long long cur[8], nxt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int row = bs + 8 * k;
        cur[k] = (w < words && row < n) ? gm[(size_t)row * words + w] : 0ULL;
    }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        if (w == c) {
#pragma unroll
            for (int k = 0; k < 8; ++k) diag[bs + 8 * k] = cur[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {                       // prefetch the next chunk's rows (words right of it only)
            const int row = (c + 1) * 64 + bs + 8 * k;
            nxt[k] = (c + 1 < nch && w > c && w < words && row < n) ? gm[(size_t)row * words + w] : 0ULL;
        }
        __syncthreads();
        if (t < 64) {
            const unsigned long long d = diag[t];
            const int dlo = (int)(unsigned)d, dhi = (int)(unsigned)(d >> 32);
            unsigned long long rem = removed[c];
#pragma unroll
            for (int b = 0; b < 64; ++b) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane(dlo, b), hi = (unsigned)__builtin_amdgcn_readlane(dhi, b);
                const unsigned long long row = ((unsigned long long)hi << 32) | lo;
                if (!((rem >> b) & 1ULL)) rem |= row;
            }
            const int left = n - c * 64;
            const unsigned long long valid = left >= 64 ? ~0ULL : ((1ULL << left) - 1ULL);
            const unsigned long long alive = ~rem & valid;
            if (c * 64 + t < maxn) gk[c * 64 + t] = (unsigned char)((alive >> t) & 1ULL);
            if (t == 0) s_alive = alive;
        }
        __syncthreads();
        const unsigned long long alive = s_alive;
        unsigned long long acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if ((alive >> (bs + 8 * k)) & 1ULL) acc |= cur[k];
        if (w > c && acc) atomicOr(&removed[w], acc);
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = nxt[k];
        __syncthreads();
    }
    for (int i = nch * 64 + t; i < maxn; i += 256) gk[i] = 0;
}

__global__ void k_fill_zero_u64(unsigned long long* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0ULL;
}

// mask_ws: workspace of G*maxn*ceil(maxn/64) u64.
extern "C" int cr_nms_grouped(cr_ctx* ctx, const float* boxes, const int* counts, int G, int maxn, float thresh,
                              void* mask_ws, unsigned char* keep) {
    CR_CHECK_ARG(ctx, "cr_nms_grouped: ctx is NULL");
    if (G == 0 || maxn == 0) return CR_OK;
    CR_CHECK_ARG(boxes && counts && mask_ws && keep && G > 0 && maxn > 0 && maxn <= 65536, "cr_nms_grouped: bad args");
    const int words = (maxn + 63) / 64;
    // rows below the diagonal block are skipped by the kernel -> clear the matrix first
    {   // a kernel, not hipMemsetAsync: memset nodes misbehave inside captured HIP graphs (ROCm 7.2)
        const int64_t nz = (int64_t)G * maxn * words;
        hipLaunchKernelGGL(k_fill_zero_u64, dim3((unsigned)cr_cdiv(nz, 256)), dim3(256), 0, ctx->stream,
                           (unsigned long long*)mask_ws, nz);
        CR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_nms_mask, dim3(words, words, G), dim3(64), 0, ctx->stream, boxes, counts, maxn, thresh,
                       (unsigned long long*)mask_ws);
    CR_LAUNCH_CHECK();
    if (words <= 32)
        hipLaunchKernelGGL(k_nms_scan32, dim3(G), dim3(256), 0, ctx->stream, counts, maxn,
                           (const unsigned long long*)mask_ws, keep);
    else
        hipLaunchKernelGGL(k_nms_scan, dim3(G), dim3(64), words * 8, ctx->stream, counts, maxn,
                           (const unsigned long long*)mask_ws, keep);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
