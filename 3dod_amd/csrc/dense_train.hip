// Static-shape (sync-free) training glue of the RPN and the RoI heads as a handful of fused kernels: proposal
// decoding, anchor / proposal <-> ground-truth matching, label assignment around the IoU-weighted sampling, and the
// RPN losses with their gradients.  Each kernel replaces dozens of elementwise launches of the tensor-op formulation
// (kept as the oracle: oracle/cpu_backend.py), which dominated the step once the convolutions were tuned.
//
// Reference code paths restated here (paths into the reference tree):
//   cubercnn/modeling/proposal_generator/rpn.py:41-110   RPNWithIgnore.label_and_sample_anchors (+ ignore regions)
//   cubercnn/modeling/proposal_generator/rpn.py:129-273  losses, "IoUness" objectness, uncertainty-weighted L1
//   cubercnn/modeling/proposal_generator/rpn.py:275-328  subsample_labels (IoU-weighted multinomial sampling)
//   cubercnn/modeling/roi_heads/roi_heads.py:2773-2840   ROIHeads3D.label_and_sample_proposals
//   detectron2 Matcher / Box2BoxTransform / find_top_rpn_proposals [third-party, restated in 3dod_amd/d2lite]
#include "cr_common.h"
#include <math.h>

#define NEG_IOU (-1.0f)

struct Box { float x1, y1, x2, y2; };
__device__ __forceinline__ Box ldbox(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return Box{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ float box_area(const Box& b) { return (b.x2 - b.x1) * (b.y2 - b.y1); }
__device__ __forceinline__ float box_inter(const Box& a, const Box& b) {
    const float w = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
    const float h = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
    return w * h;
}
// pairwise_iou: inter > 0 ? inter / (area_gt + area_box - inter) : 0
__device__ __forceinline__ float iou_gt_box(const Box& gt, float area_gt, const Box& b, float area_b) {
    const float inter = box_inter(gt, b);
    return inter > 0.f ? inter / (area_gt + area_b - inter) : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// 1. proposal decoding of the per-level top-k candidates, clip, validity (find_top_rpn_proposals before NMS)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rpn_decode_select(const float* __restrict__ anchors, const float* __restrict__ deltas,
                                                           const int64_t* __restrict__ idx, const float* __restrict__ scores,
                                                           int B, int A, int S, float wx, float wy, float ww, float wh,
                                                           float scale_clamp, const float* __restrict__ img_hw, float min_size,
                                                           float* __restrict__ boxes, float* __restrict__ nms_boxes,
                                                           unsigned char* __restrict__ valid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S) return;
    const int b = i / S;
    const int64_t a = idx[i];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    bool ok = false;
    if (a >= 0 && a < A) {
        const Box an = ldbox(anchors + a * 4);
        const float4 d = *reinterpret_cast<const float4*>(deltas + ((size_t)b * A + a) * 4);
        const float w = an.x2 - an.x1, h = an.y2 - an.y1;
        const float cx = an.x1 + 0.5f * w, cy = an.y1 + 0.5f * h;
        const float dx = d.x / wx, dy = d.y / wy;
        const float dw = fminf(d.z / ww, scale_clamp), dh = fminf(d.w / wh, scale_clamp);
        const float pcx = dx * w + cx, pcy = dy * h + cy;
        const float pw = expf(dw) * w, ph = expf(dh) * h;
        float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
        const float sc = scores[i];
        const bool fin = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2) && isfinite(sc);
        if (fin) {
            const float H = img_hw[b * 2], W = img_hw[b * 2 + 1];
            x1 = fminf(fmaxf(x1, 0.f), W); y1 = fminf(fmaxf(y1, 0.f), H);
            x2 = fminf(fmaxf(x2, 0.f), W); y2 = fminf(fmaxf(y2, 0.f), H);
            o = make_float4(x1, y1, x2, y2);
            ok = (x2 - x1) > min_size && (y2 - y1) > min_size;
        }
    }
    *reinterpret_cast<float4*>(boxes + (size_t)i * 4) = o;
    *reinterpret_cast<float4*>(nms_boxes + (size_t)i * 4) = ok ? o : make_float4(0.f, 0.f, 0.f, 0.f);
    valid[i] = ok ? 1 : 0;
}

extern "C" int cr_rpn_decode_select(cr_ctx* ctx, const float* anchors, const float* deltas, const int64_t* idx,
                                    const float* scores, int B, int A, int S, const float* weights4, float scale_clamp,
                                    const float* img_hw, float min_size, float* boxes, float* nms_boxes,
                                    unsigned char* valid) {
    CR_CHECK_ARG(ctx && anchors && deltas && idx && scores && weights4 && img_hw && boxes && nms_boxes && valid,
                 "cr_rpn_decode_select: NULL pointer");
    if (B * S == 0) return CR_OK;
    CR_CHECK_ARG(B > 0 && A > 0 && S > 0, "cr_rpn_decode_select: bad sizes");
    hipLaunchKernelGGL(k_rpn_decode_select, dim3((unsigned)cr_cdiv((int64_t)B * S, 256)), dim3(256), 0, ctx->stream,
                       anchors, deltas, idx, scores, B, A, S, weights4[0], weights4[1], weights4[2], weights4[3],
                       scale_clamp, img_hw, min_size, boxes, nms_boxes, valid);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 2. box <-> ground-truth matching.  boxes (R,4) shared by the batch (box_bstride = 0: anchors) or (B,R,4).
//    gt boxes (B,G,4), gt classes (B,G) int64: >= 0 valid, -1 ignore region, -2 padding.
//    max_iou (B,R): max over valid gt (NEG_IOU when none), argmax (first), max_ioa (B,R): max over ignore gt of
//    inter / area(box).  best (B,G) u64 (optional, zero-filled by the call): per valid gt the max IoU over boxes and the
//    lowest box index attaining it, packed (iou_bits << 32) | ~index.
// ---------------------------------------------------------------------------------------------------------------
#define MAXG 64      // ground-truth rows staged in LDS at a time: G itself is unbounded (crowded Omni3D images exceed 64)
__global__ __launch_bounds__(256) void k_box_match(const float* __restrict__ boxes, int64_t box_bstride,
                                                   const float* __restrict__ gtb, const int64_t* __restrict__ gtc, int B,
                                                   int R, int G, float* __restrict__ max_iou, int* __restrict__ argmax,
                                                   float* __restrict__ max_ioa, unsigned long long* __restrict__ best) {
    __shared__ float sg[MAXG * 4];
    __shared__ float sarea[MAXG];
    __shared__ int scls[MAXG];
    __shared__ unsigned long long sbest[MAXG];
    const int b = blockIdx.y, t = threadIdx.x;
    const int r = blockIdx.x * 256 + t;
    Box bx{0.f, 0.f, 0.f, 0.f};
    if (r < R) bx = ldbox(boxes + (size_t)b * box_bstride + (size_t)r * 4);
    const float ab = box_area(bx);
    float mi = NEG_IOU, ma = 0.f;
    int am = 0;
    for (int g0 = 0; g0 < G; g0 += MAXG) {                 // chunks of MAXG rows, in order (first arg-max is kept)
        const int gn = min(MAXG, G - g0);
        if (g0) __syncthreads();                          // the previous chunk has been consumed
        for (int g = t; g < gn; g += 256) {
            const float* p = gtb + ((size_t)b * G + g0 + g) * 4;
            sg[g * 4 + 0] = p[0]; sg[g * 4 + 1] = p[1]; sg[g * 4 + 2] = p[2]; sg[g * 4 + 3] = p[3];
            sarea[g] = (p[2] - p[0]) * (p[3] - p[1]);
            scls[g] = (int)gtc[(size_t)b * G + g0 + g];
            sbest[g] = 0ULL;
        }
        __syncthreads();
        if (r < R) {
            for (int g = 0; g < gn; ++g) {
                const int c = scls[g];
                if (c < -1) continue;
                const Box gb{sg[g * 4], sg[g * 4 + 1], sg[g * 4 + 2], sg[g * 4 + 3]};
                if (c >= 0) {
                    const float v = iou_gt_box(gb, sarea[g], bx, ab);
                    if (v > mi) { mi = v; am = g0 + g; }
                    if (best) {
                        const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(~(unsigned)r);
                        atomicMax(&sbest[g], key);         // v >= 0: its bit pattern orders like the float
                    }
                } else {
                    const float inter = box_inter(gb, bx);
                    const float v = inter > 0.f ? inter / ab : 0.f;
                    ma = fmaxf(ma, v);
                }
            }
        }
        if (best) {
            __syncthreads();
            for (int g = t; g < gn; g += 256)
                if (scls[g] >= 0) atomicMax(&best[(size_t)b * G + g0 + g], sbest[g]);
        }
    }
    if (r < R) {
        max_iou[(size_t)b * R + r] = mi;
        argmax[(size_t)b * R + r] = am;
        max_ioa[(size_t)b * R + r] = ma;
    }
}

__global__ void k_zero_u64(unsigned long long* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0ULL;
}

extern "C" int cr_box_match(cr_ctx* ctx, const float* boxes, int boxes_per_image, const float* gt_boxes,
                            const int64_t* gt_classes, int B, int R, int G, float* max_iou, int* argmax, float* max_ioa,
                            unsigned long long* best) {
    CR_CHECK_ARG(ctx && boxes && gt_boxes && gt_classes && max_iou && argmax && max_ioa, "cr_box_match: NULL pointer");
    if (B == 0 || R == 0) return CR_OK;
    CR_CHECK_ARG(B > 0 && R > 0 && G > 0, "cr_box_match: G >= 1 required (G=%d)", G);
    if (best) {
        hipLaunchKernelGGL(k_zero_u64, dim3((unsigned)cr_cdiv((int64_t)B * G, 256)), dim3(256), 0, ctx->stream, best,
                           (int64_t)B * G);
        CR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_box_match, dim3((unsigned)cr_cdiv(R, 256), B), dim3(256), 0, ctx->stream, boxes,
                       boxes_per_image ? (int64_t)R * 4 : (int64_t)0, gt_boxes, gt_classes, B, R, G, max_iou, argmax,
                       max_ioa, best);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 3. RPN anchor labels before sampling + the sampling keys.
//    labels_pre (B,A) int8: Matcher thresholds [lo, hi] -> {0, -1, 1} plus allow_low_quality_matches (every anchor
//    that attains a valid gt's best IoU is foreground); out (B,A) int32 initialised to -1, or 1 for the "forced"
//    anchors (rpn.py:75: the arg-max anchor of each gt); matched_iou = max(max_iou, 0);
//    keys (2,B,A): (matched_iou + eps) / e for positive / negative candidates, else 0  (e ~ Exp(1) from the caller:
//    top-k of these keys == multinomial sampling without replacement with weights iou + eps).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rpn_label(const float* __restrict__ anchors, const float* __restrict__ gtb,
                                                   const int64_t* __restrict__ gtc, const float* __restrict__ max_iou,
                                                   const unsigned long long* __restrict__ best, const float* __restrict__ e,
                                                   int B, int A, int G, float lo, float hi, int l0, int l1, int l2, float eps,
                                                   signed char* __restrict__ labels_pre, int* __restrict__ out,
                                                   float* __restrict__ matched_iou, float* __restrict__ keys) {
    __shared__ float sg[MAXG * 4];
    __shared__ float sarea[MAXG];
    __shared__ float sbest[MAXG];
    __shared__ int sbidx[MAXG];      // -1 = gt not valid
    const int b = blockIdx.y, t = threadIdx.x;
    const int a = blockIdx.x * 256 + t;
    Box bx{0.f, 0.f, 0.f, 0.f};
    if (a < A) bx = ldbox(anchors + (size_t)a * 4);
    const float ab = box_area(bx);
    bool lowq = false, is_best = false;
    for (int g0 = 0; g0 < G; g0 += MAXG) {                 // G is unbounded: MAXG rows staged at a time
        const int gn = min(MAXG, G - g0);
        if (g0) __syncthreads();
        for (int g = t; g < gn; g += 256) {
            const float* p = gtb + ((size_t)b * G + g0 + g) * 4;
            sg[g * 4 + 0] = p[0]; sg[g * 4 + 1] = p[1]; sg[g * 4 + 2] = p[2]; sg[g * 4 + 3] = p[3];
            sarea[g] = (p[2] - p[0]) * (p[3] - p[1]);
            const bool v = gtc[(size_t)b * G + g0 + g] >= 0;
            const unsigned long long k = best[(size_t)b * G + g0 + g];
            sbest[g] = __uint_as_float((unsigned)(k >> 32));
            sbidx[g] = v ? (int)(~(unsigned)(k & 0xffffffffULL)) : -1;
        }
        __syncthreads();
        if (a < A) {
            for (int g = 0; g < gn; ++g) {
                if (sbidx[g] < 0) continue;
                const Box gb{sg[g * 4], sg[g * 4 + 1], sg[g * 4 + 2], sg[g * 4 + 3]};
                const float v = iou_gt_box(gb, sarea[g], bx, ab);
                lowq |= v == sbest[g];
                is_best |= sbidx[g] == a;
            }
        }
    }
    if (a >= A) return;
    const size_t i = (size_t)b * A + a;
    const float vals = max_iou[i];
    int lab = l2;
    if (vals < hi) lab = l1;
    if (vals < lo) lab = l0;
    if (lowq) lab = 1;
    labels_pre[i] = (signed char)lab;
    const float mi = fmaxf(vals, 0.f);
    matched_iou[i] = mi;
    out[i] = (is_best && lab == 1) ? 1 : -1;
    const size_t BA = (size_t)B * A;
    keys[i] = lab == 1 ? (mi + eps) / e[i] : 0.f;
    keys[BA + i] = lab == 0 ? (mi + eps) / e[BA + i] : 0.f;
}

extern "C" int cr_rpn_label(cr_ctx* ctx, const float* anchors, const float* gt_boxes, const int64_t* gt_classes,
                            const float* max_iou, const unsigned long long* best, const float* expo, int B, int A, int G,
                            float lo, float hi, const int* labels3, float eps, signed char* labels_pre, int* out,
                            float* matched_iou, float* keys) {
    CR_CHECK_ARG(ctx && anchors && gt_boxes && gt_classes && max_iou && best && expo && labels3 && labels_pre && out &&
                 matched_iou && keys, "cr_rpn_label: NULL pointer");
    if (B == 0 || A == 0) return CR_OK;
    CR_CHECK_ARG(G > 0, "cr_rpn_label: G >= 1 required");
    hipLaunchKernelGGL(k_rpn_label, dim3((unsigned)cr_cdiv(A, 256), B), dim3(256), 0, ctx->stream, anchors, gt_boxes,
                       gt_classes, max_iou, best, expo, B, A, G, lo, hi, labels3[0], labels3[1], labels3[2], eps,
                       labels_pre, out, matched_iou, keys);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 4. scatter of the sampled picks into the label map (one block per image).
//    pos picks (B,KP): keys > 0 are real; neg picks (B,KN): real and rank < n_s - n_pos.  out[pos] = 1, out[neg] = 0,
//    except that a sampled negative inside an ignore region (ioa >= thresh) is dropped to -1 when the image has more
//    than one sampled negative (rpn.py:93-104).  Forced anchors keep their 1.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rpn_scatter(const int64_t* __restrict__ pidx, const float* __restrict__ pkey, int KP,
                                                     const int64_t* __restrict__ nidx, const float* __restrict__ nkey, int KN,
                                                     int n_s, const float* __restrict__ ioa, float ignore_thresh, int A,
                                                     int* __restrict__ out) {
    __shared__ int s_cnt[2];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < 2) s_cnt[t] = 0;
    __syncthreads();
    int c = 0;
    for (int i = t; i < KP; i += 256) c += pkey[(size_t)b * KP + i] > 0.f;
    if (c) atomicAdd(&s_cnt[0], c);
    __syncthreads();
    const int limit = n_s - s_cnt[0];
    c = 0;
    for (int i = t; i < KN; i += 256) c += (nkey[(size_t)b * KN + i] > 0.f) && (i < limit);
    if (c) atomicAdd(&s_cnt[1], c);
    __syncthreads();
    const bool many = s_cnt[1] > 1;
    int* o = out + (size_t)b * A;
    for (int i = t; i < KP; i += 256)
        if (pkey[(size_t)b * KP + i] > 0.f) o[pidx[(size_t)b * KP + i]] = 1;
    for (int i = t; i < KN; i += 256)
        if (nkey[(size_t)b * KN + i] > 0.f && i < limit) {
            const int64_t a = nidx[(size_t)b * KN + i];
            if (o[a] != 1) o[a] = (many && ioa[(size_t)b * A + a] >= ignore_thresh) ? -1 : 0;
        }
}

extern "C" int cr_rpn_scatter(cr_ctx* ctx, const int64_t* pos_idx, const float* pos_key, int KP, const int64_t* neg_idx,
                              const float* neg_key, int KN, int n_s, const float* ioa, float ignore_thresh, int B, int A,
                              int* out) {
    CR_CHECK_ARG(ctx && pos_idx && pos_key && neg_idx && neg_key && ioa && out, "cr_rpn_scatter: NULL pointer");
    if (B == 0) return CR_OK;
    hipLaunchKernelGGL(k_rpn_scatter, dim3(B), dim3(256), 0, ctx->stream, pos_idx, pos_key, KP, neg_idx, neg_key, KN, n_s,
                       ioa, ignore_thresh, A, out);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 5. RPN losses and their gradients in one pass (rpn.py:129-273, "IoUness" objectness, smooth_l1 beta 0 = L1):
//      t = IoU(anchor, matched gt) for positives, else 0
//      loss_cls = sum BCEWithLogits(logit, t) * t,   loss_loc = sum_pos |delta - target|_1 * t
//    sums[6] = [loss_cls, loss_loc, n_pos, n_neg, sum sigmoid(logit) over pos, sum sigmoid(logit) over non-pos]
//    (two-stage, fixed-order reduction: bitwise reproducible);  dlogits (B,A), ddeltas (B,A,4) unscaled.
// ---------------------------------------------------------------------------------------------------------------
#define LOSS_NV 6
__global__ __launch_bounds__(256) void k_rpn_loss(const float* __restrict__ logits, const float* __restrict__ deltas,
                                                  const float* __restrict__ anchors, const int* __restrict__ labels,
                                                  const int* __restrict__ midx, const float* __restrict__ gtb, int B, int A,
                                                  int G, float wx, float wy, float ww, float wh, float* __restrict__ partial,
                                                  float* __restrict__ dlogits, float* __restrict__ ddeltas) {
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    float v[LOSS_NV] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a < A) {
        const size_t i = (size_t)b * A + a;
        const int lab = labels[i];
        const float x = logits[i];
        const float sig = 1.f / (1.f + expf(-x));
        float dl = 0.f;
        float4 dd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lab == 1) {
            const Box an = ldbox(anchors + (size_t)a * 4);
            const Box g = ldbox(gtb + ((size_t)b * G + midx[i]) * 4);
            const float inter = box_inter(an, g);
            const float t = inter / (box_area(an) + box_area(g) - inter);
            const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
            v[0] = bce * t;
            dl = (sig - t) * t;
            const float sw = an.x2 - an.x1, sh = an.y2 - an.y1, sx = an.x1 + 0.5f * sw, sy = an.y1 + 0.5f * sh;
            const float tw = g.x2 - g.x1, th = g.y2 - g.y1, tx = g.x1 + 0.5f * tw, ty = g.y1 + 0.5f * th;
            const float4 d = *reinterpret_cast<const float4*>(deltas + i * 4);
            const float e0 = d.x - wx * (tx - sx) / sw, e1 = d.y - wy * (ty - sy) / sh;
            const float e2 = d.z - ww * logf(tw / sw), e3 = d.w - wh * logf(th / sh);
            v[1] = (((fabsf(e0) + fabsf(e1)) + fabsf(e2)) + fabsf(e3)) * t;
            auto sgn = [](float z) { return z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f); };
            dd = make_float4(sgn(e0) * t, sgn(e1) * t, sgn(e2) * t, sgn(e3) * t);
            v[2] = 1.f;
            v[4] = sig;
        } else {
            v[3] = lab == 0 ? 1.f : 0.f;
            v[5] = sig;
        }
        dlogits[i] = dl;
        *reinterpret_cast<float4*>(ddeltas + i * 4) = dd;
    }
    __shared__ float red[4][LOSS_NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < LOSS_NV; ++k) {
        float s = v[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < LOSS_NV) {
        const int k = threadIdx.x;
        partial[((size_t)b * gridDim.x + blockIdx.x) * LOSS_NV + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
}

__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ partial, int n, int nv, float* __restrict__ sums) {
    // one block, nv <= 8 columns: 32 threads per column take strided rows (f64), then a fixed-order LDS tree
    __shared__ double sm[8][32];
    const int k = threadIdx.x >> 5, j = threadIdx.x & 31;
    double s = 0.0;
    if (k < nv)
        for (int i = j; i < n; i += 32) s += (double)partial[(size_t)i * nv + k];
    sm[k][j] = s;
    __syncthreads();
    for (int w = 16; w >= 1; w >>= 1) {
        if (j < w) sm[k][j] += sm[k][j + w];
        __syncthreads();
    }
    if (j == 0 && k < nv) sums[k] = (float)sm[k][0];
}

extern "C" int cr_rpn_loss(cr_ctx* ctx, const float* logits, const float* deltas, const float* anchors, const int* labels,
                           const int* matched_idx, const float* gt_boxes, int B, int A, int G, const float* weights4,
                           float* partial_ws, float* sums6, float* dlogits, float* ddeltas) {
    CR_CHECK_ARG(ctx && logits && deltas && anchors && labels && matched_idx && gt_boxes && weights4 && partial_ws && sums6 &&
                 dlogits && ddeltas, "cr_rpn_loss: NULL pointer");
    CR_CHECK_ARG(B > 0 && A > 0 && G > 0, "cr_rpn_loss: bad sizes");
    const int nb = (int)cr_cdiv(A, 256);
    hipLaunchKernelGGL(k_rpn_loss, dim3(nb, B), dim3(256), 0, ctx->stream, logits, deltas, anchors, labels, matched_idx,
                       gt_boxes, B, A, G, weights4[0], weights4[1], weights4[2], weights4[3], partial_ws, dlogits, ddeltas);
    CR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, partial_ws, nb * B, LOSS_NV, sums6);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 6. RoI labels before sampling (roi_heads.py:2773-2840, one block per image).
//    cls (B,R) int64: matched gt class where IoU >= thr, K = background otherwise; -1 for invalid proposals and for
//    background inside an ignore region (only when the image has > 1 valid background proposals).
//    keys (2,B,R): (IoU + eps) / e for foreground / background candidates, 0 otherwise.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_roi_label(const float* __restrict__ max_iou, const int* __restrict__ argmax,
                                                   const float* __restrict__ max_ioa, const unsigned char* __restrict__ valid,
                                                   const int64_t* __restrict__ gtc, const float* __restrict__ e, int B, int R,
                                                   int G, int K, float thr, float ignore_thresh, float eps,
                                                   int64_t* __restrict__ cls, float* __restrict__ matched_iou,
                                                   float* __restrict__ keys) {
    __shared__ int s_bg;
    const int b = blockIdx.x, t = threadIdx.x;
    if (t == 0) s_bg = 0;
    __syncthreads();
    int c = 0;
    for (int r = t; r < R; r += 256) {
        const size_t i = (size_t)b * R + r;
        c += (!(max_iou[i] >= thr)) && valid[i];
    }
    if (c) atomicAdd(&s_bg, c);
    __syncthreads();
    const bool many = s_bg > 1;
    const size_t BR = (size_t)B * R;
    for (int r = t; r < R; r += 256) {
        const size_t i = (size_t)b * R + r;
        const float v = max_iou[i];
        const bool fg = v >= thr;
        const bool ign = !fg && many && max_ioa[i] >= ignore_thresh;
        int64_t cl = K;
        if (fg) { const int64_t gc = gtc[(size_t)b * G + argmax[i]]; cl = gc > 0 ? gc : 0; }
        if (ign || !valid[i]) cl = -1;
        cls[i] = cl;
        const float mi = fmaxf(v, 0.f);
        matched_iou[i] = mi;
        keys[i] = (cl >= 0 && cl < K) ? (mi + eps) / e[i] : 0.f;
        keys[BR + i] = cl == K ? (mi + eps) / e[BR + i] : 0.f;
    }
}

extern "C" int cr_roi_label(cr_ctx* ctx, const float* max_iou, const int* argmax, const float* max_ioa,
                            const unsigned char* valid, const int64_t* gt_classes, const float* expo, int B, int R, int G,
                            int K, float thr, float ignore_thresh, float eps, int64_t* cls, float* matched_iou, float* keys) {
    CR_CHECK_ARG(ctx && max_iou && argmax && max_ioa && valid && gt_classes && expo && cls && matched_iou && keys,
                 "cr_roi_label: NULL pointer");
    if (B == 0 || R == 0) return CR_OK;
    hipLaunchKernelGGL(k_roi_label, dim3(B), dim3(256), 0, ctx->stream, max_iou, argmax, max_ioa, valid, gt_classes, expo, B,
                       R, G, K, thr, ignore_thresh, eps, cls, matched_iou, keys);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 7. compaction of the sampled picks to n_s slots per image (one block per image, blockDim >= KF + KB, <= 1024):
//    entries = [foreground picks (KF) | background picks (KB)], valid = key > 0 (background also rank < n_s - n_fg);
//    a stable partition moves the valid entries to the front ("foreground first"), the first n_s slots are kept.
//    Outputs per slot: box, valid, class (-1 when invalid), matched gt index; counts (B,2) = [n_fg, n_bg].
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_roi_compact(const int64_t* __restrict__ fidx, const float* __restrict__ fkey, int KF,
                                                      const int64_t* __restrict__ bidx, const float* __restrict__ bkey, int KB,
                                                      int n_s, const float* __restrict__ boxes, const int64_t* __restrict__ cls,
                                                      const int* __restrict__ argmax, int R, float* __restrict__ o_boxes,
                                                      unsigned char* __restrict__ o_valid, int64_t* __restrict__ o_cls,
                                                      int64_t* __restrict__ o_gt, int* __restrict__ counts) {
    __shared__ int s_nfg, s_wv[16], s_wi[16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    if (t == 0) s_nfg = 0;
    __syncthreads();
    const bool isf = t < KF, isb = t >= KF && t < KF + KB;
    const bool fv = isf && fkey[(size_t)b * KF + t] > 0.f;
    const unsigned long long fm = __ballot(fv);
    if (lane == 0 && fm) atomicAdd(&s_nfg, __popcll(fm));
    __syncthreads();
    const int nfg = s_nfg, limit = n_s - nfg;
    const bool bv = isb && bkey[(size_t)b * KB + (t - KF)] > 0.f && (t - KF) < limit;
    const bool v = fv || bv, inv = (isf || isb) && !v;
    const unsigned long long vm = __ballot(v), im = __ballot(inv);
    if (lane == 0) { s_wv[wave] = __popcll(vm); s_wi[wave] = __popcll(im); }
    __syncthreads();
    int vbefore = 0, ibefore = 0, vtotal = 0;
    for (int w = 0; w < nw; ++w) {
        if (w < wave) { vbefore += s_wv[w]; ibefore += s_wi[w]; }
        vtotal += s_wv[w];
    }
    const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    vbefore += __popcll(vm & below);
    ibefore += __popcll(im & below);
    if (t == 0) { counts[b * 2] = nfg; counts[b * 2 + 1] = vtotal - nfg; }
    if (!(isf || isb)) return;
    const int pos = v ? vbefore : vtotal + ibefore;
    if (pos >= n_s) return;
    const int64_t src = isf ? fidx[(size_t)b * KF + t] : bidx[(size_t)b * KB + (t - KF)];
    const size_t o = (size_t)b * n_s + pos, si = (size_t)b * R + src;
    *reinterpret_cast<float4*>(o_boxes + o * 4) = *reinterpret_cast<const float4*>(boxes + si * 4);
    o_valid[o] = v ? 1 : 0;
    o_cls[o] = v ? cls[si] : -1;
    o_gt[o] = argmax[si];
}

extern "C" int cr_roi_compact(cr_ctx* ctx, const int64_t* fg_idx, const float* fg_key, int KF, const int64_t* bg_idx,
                              const float* bg_key, int KB, int n_s, const float* boxes, const int64_t* cls, const int* argmax,
                              int B, int R, float* o_boxes, unsigned char* o_valid, int64_t* o_cls, int64_t* o_gt, int* counts) {
    CR_CHECK_ARG(ctx && fg_idx && fg_key && bg_idx && bg_key && boxes && cls && argmax && o_boxes && o_valid && o_cls && o_gt &&
                 counts, "cr_roi_compact: NULL pointer");
    if (B == 0) return CR_OK;
    CR_CHECK_ARG(KF >= 0 && KB >= 0 && KF + KB >= n_s && KF + KB <= 1024, "cr_roi_compact: n_s <= KF + KB <= 1024 required");
    const int threads = (int)cr_cdiv(KF + KB, 64) * 64;
    hipLaunchKernelGGL(k_roi_compact, dim3(B), dim3(threads), 0, ctx->stream, fg_idx, fg_key, KF, bg_idx, bg_key, KB, n_s,
                       boxes, cls, argmax, R, o_boxes, o_valid, o_cls, o_gt, counts);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 8. Fast R-CNN losses on the padded sample (fast_rcnn.py:145-194): softmax cross-entropy over K+1 classes on the
//    valid rows, L1 (smooth_l1, beta 0) box regression on the foreground rows' own class, both as sums (the caller
//    divides by the number of valid rows), with gradients and the decoded boxes of the sampled class.  One wave / row.
//    sums3 = [sum ce, sum l1, n_valid];  dscores (N,K+1), ddeltas (N,K*4), pred (N,4).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_box_loss(const float* __restrict__ scores, const float* __restrict__ deltas,
                                                  const unsigned char* __restrict__ valid, const int64_t* __restrict__ cls,
                                                  const float* __restrict__ pboxes, const int64_t* __restrict__ gt_idx,
                                                  const float* __restrict__ gtb, int N, int S, int G, int K, float wx, float wy,
                                                  float ww, float wh, float scale_clamp, float* __restrict__ partial,
                                                  float* __restrict__ dscores, float* __restrict__ ddeltas,
                                                  float* __restrict__ pred) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    float v[3] = {0.f, 0.f, 0.f};
    if (row < N) {
        const int C = K + 1;
        const bool ok = valid[row] != 0;
        const int64_t c0 = cls[row];
        const int tcls = (int)(c0 > 0 ? c0 : 0);
        const float* sr = scores + (size_t)row * C;
        float m = -INFINITY;
        for (int j = lane; j < C; j += 64) m = fmaxf(m, sr[j]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        float se = 0.f;
        for (int j = lane; j < C; j += 64) se += expf(sr[j] - m);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) se += __shfl_xor(se, off, 64);
        const float lse = m + logf(se);
        for (int j = lane; j < C; j += 64) {
            const float p = expf(sr[j] - lse);
            dscores[(size_t)row * C + j] = ok ? (p - (j == tcls ? 1.f : 0.f)) : 0.f;
        }
        const bool fg = ok && c0 >= 0 && c0 < K;
        const int kcls = tcls < K ? tcls : K - 1;
        float* dr = ddeltas + (size_t)row * K * 4;
        for (int j = lane; j < K * 4; j += 64) dr[j] = 0.f;
        if (lane == 0) {
            v[0] = ok ? lse - sr[tcls] : 0.f;
            v[2] = ok ? 1.f : 0.f;
            const Box pb = ldbox(pboxes + (size_t)row * 4);
            const float4 d = *reinterpret_cast<const float4*>(deltas + ((size_t)row * K + kcls) * 4);
            const float sw = pb.x2 - pb.x1, sh = pb.y2 - pb.y1, sx = pb.x1 + 0.5f * sw, sy = pb.y1 + 0.5f * sh;
            if (fg) {
                const int b = row / S;
                const Box g = ldbox(gtb + ((size_t)b * G + gt_idx[row]) * 4);
                const float tw = g.x2 - g.x1, th = g.y2 - g.y1, tx = g.x1 + 0.5f * tw, ty = g.y1 + 0.5f * th;
                const float e0 = d.x - wx * (tx - sx) / sw, e1 = d.y - wy * (ty - sy) / sh;
                const float e2 = d.z - ww * logf(tw / sw), e3 = d.w - wh * logf(th / sh);
                v[1] = ((fabsf(e0) + fabsf(e1)) + fabsf(e2)) + fabsf(e3);
                auto sgn = [](float z) { return z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f); };
                *reinterpret_cast<float4*>(dr + kcls * 4) = make_float4(sgn(e0), sgn(e1), sgn(e2), sgn(e3));
            }
            // Box2BoxTransform.apply_deltas of the row's own class
            const float dx = d.x / wx, dy = d.y / wy, dw = fminf(d.z / ww, scale_clamp), dh = fminf(d.w / wh, scale_clamp);
            const float pcx = dx * sw + sx, pcy = dy * sh + sy, pw = expf(dw) * sw, ph = expf(dh) * sh;
            *reinterpret_cast<float4*>(pred + (size_t)row * 4) =
                make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
        }
    }
    __shared__ float red[4][3];
    if (lane == 0) { red[wave][0] = v[0]; red[wave][1] = v[1]; red[wave][2] = v[2]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        partial[(size_t)blockIdx.x * 3 + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
}

extern "C" int cr_box_loss(cr_ctx* ctx, const float* scores, const float* deltas, const unsigned char* valid,
                           const int64_t* cls, const float* prop_boxes, const int64_t* gt_idx, const float* gt_boxes, int B,
                           int S, int G, int K, const float* weights4, float scale_clamp, float* partial_ws, float* sums3,
                           float* dscores, float* ddeltas, float* pred) {
    CR_CHECK_ARG(ctx && scores && deltas && valid && cls && prop_boxes && gt_idx && gt_boxes && weights4 && partial_ws && sums3 &&
                 dscores && ddeltas && pred, "cr_box_loss: NULL pointer");
    const int N = B * S;
    if (N == 0) return CR_OK;
    CR_CHECK_ARG(K > 0 && G > 0, "cr_box_loss: bad sizes");
    const int nb = (int)cr_cdiv(N, 4);
    hipLaunchKernelGGL(k_box_loss, dim3(nb), dim3(256), 0, ctx->stream, scores, deltas, valid, cls, prop_boxes, gt_idx, gt_boxes,
                       N, S, G, K, weights4[0], weights4[1], weights4[2], weights4[3], scale_clamp, partial_ws, dscores, ddeltas,
                       pred);
    CR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, partial_ws, nb, 3, sums3);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// Ground truth of a batch -> the padded static-shape tensors of the training path in ONE launch (was a fill per tensor and a
// slice copy per image and field): boxes (B,G,4) f32 zero-padded, classes (B,G) int64 with -2 = padding, boxes3D (B,G,9)
// zero-padded, poses (B,G,3,3) identity-padded.  The per-image source pointers travel in the kernel argument (B <= 32).
// Host side of: RPNWithIgnore.label_and_sample_anchors / ROIHeads3D.label_and_sample_proposals taking `gt_instances`
// (cubercnn/modeling/proposal_generator/rpn.py:41-50, roi_heads/roi_heads.py:2773-2790 of the reference).
// ---------------------------------------------------------------------------
#define GT_MAXB 32
struct GtSrc {
    const float* boxes[GT_MAXB];
    const int64_t* classes[GT_MAXB];
    const float* boxes3d[GT_MAXB];
    const float* poses[GT_MAXB];
    int n[GT_MAXB];
};
__global__ __launch_bounds__(256) void k_gt_pack(GtSrc src, int B, int G, float* __restrict__ boxes, int64_t* __restrict__ classes,
                                                 float* __restrict__ boxes3d, float* __restrict__ poses) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i - b * G;
    const bool have = g < src.n[b];
#pragma unroll
    for (int e = 0; e < 4; ++e) boxes[(size_t)i * 4 + e] = have ? src.boxes[b][g * 4 + e] : 0.f;
    classes[i] = have ? src.classes[b][g] : (int64_t)-2;
    const bool h3 = have && src.boxes3d[b] != nullptr;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
        boxes3d[(size_t)i * 9 + e] = h3 ? src.boxes3d[b][g * 9 + e] : 0.f;
        poses[(size_t)i * 9 + e] = h3 ? src.poses[b][g * 9 + e] : ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
}

extern "C" int cr_gt_pack(cr_ctx* ctx, const float* const* boxes_ptrs, const int64_t* const* classes_ptrs,
                          const float* const* boxes3d_ptrs, const float* const* poses_ptrs, const int* counts, int B, int G,
                          float* boxes, int64_t* classes, float* boxes3d, float* poses) {
    CR_CHECK_ARG(ctx && boxes_ptrs && classes_ptrs && boxes3d_ptrs && poses_ptrs && counts, "cr_gt_pack: NULL table");
    CR_CHECK_ARG(B >= 1 && B <= GT_MAXB && G >= 1, "cr_gt_pack: B=%d must be 1..%d and G=%d >= 1", B, GT_MAXB, G);
    CR_CHECK_ARG(boxes && classes && boxes3d && poses, "cr_gt_pack: NULL output");
    GtSrc s;
    for (int b = 0; b < GT_MAXB; ++b) {
        const bool in = b < B;
        s.n[b] = in ? counts[b] : 0;
        CR_CHECK_ARG(!in || (counts[b] >= 0 && counts[b] <= G), "cr_gt_pack: image %d has %d objects, G = %d", b, in ? counts[b] : 0, G);
        s.boxes[b] = in ? boxes_ptrs[b] : nullptr;
        s.classes[b] = in ? classes_ptrs[b] : nullptr;
        s.boxes3d[b] = in ? boxes3d_ptrs[b] : nullptr;
        s.poses[b] = in ? poses_ptrs[b] : nullptr;
        CR_CHECK_ARG(!in || counts[b] == 0 || (s.boxes[b] && s.classes[b]), "cr_gt_pack: image %d: NULL boxes / classes", b);
        CR_CHECK_ARG(!in || (s.boxes3d[b] == nullptr) == (s.poses[b] == nullptr), "cr_gt_pack: image %d: boxes3D and poses go together", b);
    }
    hipLaunchKernelGGL(k_gt_pack, dim3((unsigned)cr_cdiv((int64_t)B * G, 256)), dim3(256), 0, ctx->stream, s, B, G, boxes, classes,
                       boxes3d, poses);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// RPN head output -> the tensors the training path consumes, one launch: the head evaluates objectness and anchor deltas as
// ONE 16-channel 1x1 convolution per level (y_l (B, H_l*W_l, 16) f32: A objectness logits, 4A deltas, padding); this kernel
// writes the level-concatenated logits (B, Atot) and deltas (B, Atot, 4) (detectron2 RPN.forward's permute / flatten / cat
// [third-party], rpn.py:153-170 here) and the per-level logits padded with -inf to the largest level, (B, L, amax), for the
// batched pre-NMS top-k.  The backward kernel scatters the two gradients back into dy_l (B, H_l*W_l, 16), zeros elsewhere.
// ---------------------------------------------------------------------------
#define RPN_MAXL 8
struct RpnLv { const float* y[RPN_MAXL]; float* dy[RPN_MAXL]; int n[RPN_MAXL]; int off[RPN_MAXL]; };   // n = H*W*A anchors of a level

__global__ __launch_bounds__(256) void k_rpn_unpack(RpnLv lv, int B, int L, int A, int C, int amax, int atot,
                                                    float* __restrict__ logits, float* __restrict__ deltas, float* __restrict__ padded) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * L * amax) return;
    const int j = (int)(i % amax), l = (int)((i / amax) % L), b = (int)(i / ((int64_t)amax * L));
    float lg = -INFINITY;
    if (j < lv.n[l]) {
        const int cell = j / A, a = j - cell * A;
        const float* y = lv.y[l] + ((size_t)b * (lv.n[l] / A) + cell) * C;
        lg = y[a];
        const size_t o = (size_t)b * atot + lv.off[l] + j;
        logits[o] = lg;
        *reinterpret_cast<float4*>(deltas + o * 4) = make_float4(y[A + 4 * a], y[A + 4 * a + 1], y[A + 4 * a + 2], y[A + 4 * a + 3]);
    }
    if (padded) padded[i] = lg;
}

// one thread per (b, cell, channel) of a level's dy
__global__ __launch_bounds__(256) void k_rpn_pack_grad(RpnLv lv, int B, int L, int A, int C, int atot, int64_t total,
                                                       const float* __restrict__ dlogits, const float* __restrict__ ddeltas) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // levels are laid out back to back in the index space: level l holds B * cells_l * C entries
    int64_t r = i;
    int l = 0;
    for (; l < L - 1; ++l) {
        const int64_t sz = (int64_t)B * (lv.n[l] / A) * C;
        if (r < sz) break;
        r -= sz;
    }
    const int cells = lv.n[l] / A;
    const int c = (int)(r % C), cell = (int)((r / C) % cells), b = (int)(r / ((int64_t)C * cells));
    float g = 0.f;
    if (c < A) {
        if (dlogits) g = dlogits[(size_t)b * atot + lv.off[l] + cell * A + c];
    } else if (c < 5 * A) {
        const int a = (c - A) >> 2, d = (c - A) & 3;
        if (ddeltas) g = ddeltas[((size_t)b * atot + lv.off[l] + cell * A + a) * 4 + d];
    }
    lv.dy[l][r] = g;
}

static int rpn_levels(RpnLv& lv, const float* const* y, float* const* dy, const int* cells, int L, int A, int* atot, int* amax) {
    CR_CHECK_ARG(L >= 1 && L <= RPN_MAXL && A >= 1 && cells, "cr_rpn_unpack: bad level table");
    int off = 0, mx = 0;
    for (int l = 0; l < RPN_MAXL; ++l) {
        const bool in = l < L;
        lv.y[l] = (in && y) ? y[l] : nullptr;
        lv.dy[l] = (in && dy) ? dy[l] : nullptr;
        lv.n[l] = in ? cells[l] * A : 0;
        lv.off[l] = off;
        if (in) { CR_CHECK_ARG(cells[l] > 0, "cr_rpn_unpack: empty level %d", l); off += lv.n[l]; if (lv.n[l] > mx) mx = lv.n[l]; }
    }
    *atot = off; *amax = mx;
    return CR_OK;
}

extern "C" int cr_rpn_unpack(cr_ctx* ctx, const float* const* y_ptrs, const int* cells, int L, int B, int A, int C,
                             float* logits, float* deltas, float* padded) {
    CR_CHECK_ARG(ctx && y_ptrs && logits && deltas && B >= 1 && C >= 5 * A, "cr_rpn_unpack: bad args");
    RpnLv lv; int atot, amax;
    int rc = rpn_levels(lv, y_ptrs, nullptr, cells, L, A, &atot, &amax);
    if (rc) return rc;
    for (int l = 0; l < L; ++l) CR_CHECK_ARG(y_ptrs[l], "cr_rpn_unpack: NULL level %d", l);
    const int64_t total = (int64_t)B * L * amax;
    hipLaunchKernelGGL(k_rpn_unpack, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, lv, B, L, A, C, amax, atot,
                       logits, deltas, padded);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_rpn_pack_grad(cr_ctx* ctx, const float* dlogits, const float* ddeltas, float* const* dy_ptrs, const int* cells,
                                int L, int B, int A, int C) {
    CR_CHECK_ARG(ctx && dy_ptrs && B >= 1 && C >= 5 * A, "cr_rpn_pack_grad: bad args");
    RpnLv lv; int atot, amax;
    int rc = rpn_levels(lv, nullptr, dy_ptrs, cells, L, A, &atot, &amax);
    if (rc) return rc;
    int64_t total = 0;
    for (int l = 0; l < L; ++l) { CR_CHECK_ARG(dy_ptrs[l], "cr_rpn_pack_grad: NULL level %d", l); total += (int64_t)B * cells[l] * C; }
    hipLaunchKernelGGL(k_rpn_pack_grad, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, lv, B, L, A, C, atot, total,
                       dlogits, ddeltas);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
