/*
 * cr3dod.h -- C ABI of libcr3dod.so: the MI355X (gfx950) kernels behind the
 * Cube R-CNN forward/loss path and the 1000-cube proposal-and-scoring geometry
 * of luchsonice/3dod.
 *
 * Conventions:
 *   - plain extern "C"; every call returns 0 on success or a negative
 *     errno-style code; cr_last_error() gives the thread-local message.
 *     Nothing throws across the boundary.
 *   - all data pointers are DEVICE pointers owned by the caller (contiguous,
 *     dtype as declared); the library allocates nothing persistent except the
 *     opaque cr_ctx (stream handle + small device workspace).
 *   - every call only ENQUEUES on the ctx stream and returns; no host syncs.
 *   - one ctx per process/GPU; calls on one ctx are not thread-safe.
 *   - empty inputs (n == 0) return 0 without launching, like the reference's
 *     early returns (cubercnn/modeling/roi_heads/roi_heads.py:332-333,2278-2279).
 *
 * Each entry point cites the reference interface it replaces (file:line into
 * the reference tree).
 */
#ifndef CR3DOD_H
#define CR3DOD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cr_ctx cr_ctx;

#define CR_OK 0
#define CR_EINVAL (-22)
#define CR_ENOMEM (-12)
#define CR_EHIP (-5)
#define CR_ERANGE (-34)

/* ---- context ----------------------------------------------------------- */
/* hip_stream: a hipStream_t (may be NULL for the default stream).  The ctx
 * does not own the stream. */
int cr_ctx_create(int device, void* hip_stream, cr_ctx** out);
int cr_ctx_destroy(cr_ctx* ctx);
int cr_ctx_set_stream(cr_ctx* ctx, void* hip_stream);
const char* cr_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int cr_abi_version(void);

/* ---- geometry (HBM-bound, no MFMA) ------------------------------------- */

/* get_cuboid_verts_faces -- cubercnn/util/math_util.py:142-245.
 * box6 (n,6) = [X,Y,Z,W,H,L], R (n,3,3) row-major -> verts (n,8,3). */
int cr_cuboid_corners(cr_ctx* ctx, const float* box6, const float* R, int64_t n, float* verts);

/* K17: fused project + score + argmax of the proposal method:
 *   Cubes.get_all_corners / get_bube_corners  ProposalNetwork/utils/spaces.py:192-245
 *   cubes_to_box                              ProposalNetwork/utils/conversions.py:25-48
 *   score_iou / score_dimensions / score_corners
 *                                             ProposalNetwork/scoring/scorefunction.py:47-85,144-160
 *   product + np.argmax                       cubercnn/modeling/roi_heads/roi_heads.py:492-505
 *
 * cubes (N,P,15) = [cx,cy,cz,w,h,l,R row-major]; K (3,3) when k_per_object==0
 * else (N,3,3); im_w/im_h = the clamp tuple (W,H); ref_boxes (N,4) XYXY;
 * prior_mu / prior_sigma (N,3) in (w,h,l); rect_pts (N,4,2) = the
 * cv2.boxPoints(minAreaRect(mask contour)) quad (cr_mask_rects computes it), or
 * NULL for the reference's no-contour fallback (scorefunction.py:69-75) on every
 * object; an object whose row starts with NaN (empty mask) takes it alone.
 * Outputs (any of the first six may be NULL = not written):
 *   out_corners (N,P,8,2) out_boxes (N,P,4) out_iou/out_dim/out_corner/
 *   out_combined (N,P); out_argmax (N) int64; out_best (N) = combined[argmax].
 * P <= 4096.  NaN/Inf are data (unguarded z<=0, 0/0 ratios) as in the reference;
 * argmax follows np.argmax (first maximal index, NaN maximal).
 * iou_boxes (N,4) or NULL: the box of the IoU term when it differs from ref_boxes (which then only gives the aspect ratio
 * of score_dimensions) -- the GT-box branches score IoU against the projected ground-truth cube, roi_heads.py:459,530. */
int cr_cubes_project_score(cr_ctx* ctx, const float* cubes, int64_t N, int64_t P,
                           const float* K, int k_per_object, float im_w, float im_h,
                           const float* ref_boxes, const float* prior_mu, const float* prior_sigma,
                           const float* rect_pts,
                           float* out_corners, float* out_boxes, float* out_iou, float* out_dim,
                           float* out_corner, float* out_combined, int64_t* out_argmax, float* out_best,
                           const float* iou_boxes);

/* Same arguments and contract on out_argmax / out_best (bit-equal to cr_cubes_project_score: the product + np.argmax of
 * roi_heads.py:492-505); the six optional planes are written from reciprocal / native-exp / float32-chamfer arithmetic and
 * agree with the exact planes to 1e-4 (relative, + 1e-5 absolute on the scores in [0, 1]) instead of bit for bit.  Per object the kernel keeps the
 * cubes whose fast ratio difference, chamfer term or combined score lies within a four-fold error interval of the object's
 * maximum, re-evaluates those with the exact sequence (exact normalisers, exact product) and takes the argmax among them.
 * Objects with anything non-finite, a zero normaliser, a best score below 1e-6, a projected box thinner than one pixel that
 * still overlaps the reference box, more than 256 candidates or without a
 * rectangle (the fallback rectangle is a float64 mean over the exact boxes) run the exact sequence as a whole.
 * stats (2) int64 device counters or NULL: [0] += objects that took the exact sequence, [1] += re-evaluated candidates. */
int cr_cubes_project_score_fast(cr_ctx* ctx, const float* cubes, int64_t N, int64_t P,
                                const float* K, int k_per_object, float im_w, float im_h,
                                const float* ref_boxes, const float* prior_mu, const float* prior_sigma,
                                const float* rect_pts,
                                float* out_corners, float* out_boxes, float* out_iou, float* out_dim,
                                float* out_corner, float* out_combined, int64_t* out_argmax, float* out_best,
                                const float* iou_boxes, int64_t* stats);

/* K18: proposals.propose -- ProposalNetwork/proposals/proposals.py:338-424 with
 * the random variates supplied by the caller (RNG streams cannot be made
 * identical across back-ends; parity = same outputs for the same draws):
 *   boxes (N,4); depth (H,W); prior_mu/prior_sigma (N,3); K (3,3);
 *   dim_normals (R,3,N,P) standard normals, round r of the rejection sampler
 *   (sample_normal_in_range, ProposalNetwork/utils/utils.py:42-60);
 *   ctr_normals (3,N,P); yaw_idx (N,P) int32 in [0,36);
 *   normal (3,) unit ground normal -> the 36-yaw table of
 *   orthobasis_from_normal_t (utils.py:112-146).
 * out_cubes (N,P,15).  out_exhausted (1) int32: number of (object,proposal,dim)
 * entries still outside their range after R rounds (the caller redraws).
 * P <= 1024. */
int cr_propose(cr_ctx* ctx, const float* boxes, int64_t N, const float* depth, int H, int W,
               const float* prior_mu, const float* prior_sigma, const float* K, int64_t P,
               const float* dim_normals, int rounds, const float* ctr_normals,
               const int32_t* yaw_idx, const float* normal, float* out_cubes, int32_t* out_exhausted);

/* cr_propose for the objects of B images in one launch (the per-image loop of
 * cubercnn/modeling/roi_heads/roi_heads.py:480-505): img_idx (N) int32 selects the
 * object's depth map of depth (B,H,W), intrinsics of K (B,3,3) and ground normal
 * of normals (B,3); everything else as cr_propose. */
int cr_propose_batched(cr_ctx* ctx, const float* boxes, const int32_t* img_idx, int64_t N, const float* depth,
                       int B, int H, int W, const float* prior_mu, const float* prior_sigma, const float* K,
                       int64_t P, const float* dim_normals, int rounds, const float* ctr_normals,
                       const int32_t* yaw_idx, const float* normals, float* out_cubes, int32_t* out_exhausted);

/* K21: Plane.fit_parallel -- ProposalNetwork/utils/plane.py:79-134 with the
 * sampled index triples given.  pts (Q,3); triples (T,3) int32;
 * out_neg_eq (4) = -(a,b,c,d) as the reference returns; out_counts (T) int32
 * inlier counts (workspace + diagnostic); out_best (2) int32 = {index, count}. */
int cr_ransac_plane(cr_ctx* ctx, const float* pts, int64_t Q, const int32_t* triples, int64_t T,
                    float thresh, float* out_neg_eq, int32_t* out_counts, int32_t* out_best);

/* B independent plane fits in one launch pair (one per image of a batch, roi_heads.py:400-410): pts (B,Q,3);
 * eligible (B,Q) uint8 or NULL = which points count as inliers (the image's ground pixels; the triples must index
 * eligible points); triples (B,T,3); out_neg_eq (B,4); out_counts (B,T); out_best (B,2). */
int cr_ransac_plane_batched(cr_ctx* ctx, const float* pts, const unsigned char* eligible, int B, int64_t Q,
                            const int32_t* triples, int64_t T, float thresh, float* out_neg_eq,
                            int32_t* out_counts, int32_t* out_best);

/* Lower median (torch.median's choice) of depth[img[i], y1:y2, x1:x2] for n integer windows (x1,y1,x2,y2), clipped to
 * the map like a Python slice; NaN for an empty window.  depth (B,H,W) f32 contiguous; boxes (n,4) int32; img (n) int32.
 * Replaces the per-box torch.median loop of ROIHeads3DScore.pseudo_gt_z_box_loss
 * (cubercnn/modeling/roi_heads/roi_heads.py:1196-1232, loop at :1216-1218).  Bit-exact (selection, no arithmetic). */
int cr_box_median(cr_ctx* ctx, const float* depth, int B, int H, int W, const int32_t* boxes, const int32_t* img, int n,
                  float* out);

/* Convex hull of each RoI's 8 projected corners in the reference's order and tie rules (jarvis_march,
 * ProposalNetwork/utils/utils.py:424-470): pts (n,8,2) f32 -> order (n,8) int32 (hull vertices first), count (n) int32,
 * bump (n,8) f32 = the constant the reference adds to both coordinates of duplicated points (:427-433). */
int cr_hull8(cr_ctx* ctx, const float* pts, int n, int32_t* order, int32_t* count, float* bump);
/* segment_loss of ROIHeads3DScore (roi_heads.py:1030-1053) for n RoIs: soft polygon mask of the hull (fill_polygon,
 * utils.py:472-502) against the object's mask through sigmoid_focal_loss(inputs = mask, targets = polygon), mean over the
 * H x W pixels.  hull (n,8,2) f32 ordered vertices, count (n); masks (Nm,H,W) uint8, mask_idx (n) int32, mask_ones (Nm)
 * int32 = set pixels per mask (only the hull's bounding box is visited; outside it the term is a constant per mask bit).
 * loss (n) f32 and grad (n,8,2) f32 (d loss / d hull vertex, or NULL) are written. */
int cr_polygon_focal(cr_ctx* ctx, const float* hull, const int32_t* count, const unsigned char* masks, const int32_t* mask_idx,
                     const int32_t* mask_ones, int n, int H, int W, float* loss, float* grad);

/* Raster counts behind score_segmentation / score_mod_segmentation (ProposalNetwork/scoring/scorefunction.py:88-126;
 * cv2.convexHull + cv2.fillPoly + [::stride, ::stride] + mask_iou, utils.py:230-250) for the P proposals of one object:
 * corners (P,8,2) f32 projected corners, mask (H,W) uint8 -> counts (P,2) int32 = {polygon samples, polygon AND mask
 * samples} on the grid (stride*i, stride*j); a sample is in the polygon iff inside or on the closed hull with vertices
 * truncated to int32. */
int cr_segment_counts(cr_ctx* ctx, const float* corners, int P, const unsigned char* mask, int H, int W, int stride,
                      int32_t* counts);

/* Minimum-area rectangle of the largest 8-connected component of each object mask: the mask -> 4-point box step of
 * score_corners (ProposalNetwork/scoring/scorefunction.py:58-68: cv2.findContours(RETR_EXTERNAL) -> max contourArea ->
 * cv2.minAreaRect -> cv2.boxPoints) for n masks in one call.  Masks are uint8 (H,W) each, given EITHER as one dense
 * (n,H,W) array `masks` OR as `mask_ptrs`, a device array of n device pointers (masks of several images without a
 * gather); the other argument is NULL.  rects (n,4,2) f32 (x,y) corners, a NaN row for an empty mask (the reference
 * then falls back to the mean extent of the projected cubes, :69-75; cr_cubes_project_score does the same on a NaN
 * row); valid (n) uint8.  Scratch: labels, sizes (n,H,W) int32 (touched only inside each mask's bounding window),
 * best (n) uint64, bbox (n,4) int32.  H <= 1900.  Ties between components of equal size go to the one that starts first
 * in raster order; between rectangles of equal area to the smaller edge angle mod pi/2. */
int cr_mask_rects(cr_ctx* ctx, const unsigned char* masks, const unsigned char* const* mask_ptrs, int n, int H, int W,
                  int32_t* labels, int32_t* sizes, unsigned long long* best, int32_t* bbox, float* rects,
                  unsigned char* valid);

/* ---- Depth-Anything-V2 forward (DINOv2 ViT + DPT head), the ops that are not GEMMs / convolutions --------------- */
/* softmax(q k^T * scale) v per (batch, head) on the packed output of the qkv linear: qkv (B,N,3,H,D) bf16, out (B,N,H,D)
 * bf16, D = 64.  Flash-attention schedule on MFMA (no N x N matrix in memory).  Replaces Attention.forward /
 * MemEffAttention.forward, depth/metric_depth/depth_anything_v2/dinov2_layers/attention.py:49-82 (xformers
 * memory_efficient_attention [third-party] when available). */
int cr_attention_fwd(cr_ctx* ctx, const void* qkv, void* out, int B, int N, int H, int D, float scale);
/* nn.LayerNorm over the last dimension, x / y (M,C) bf16, gamma / beta (C) f32 (dinov2.py:96, block.py:56,68). */
int cr_layernorm(cr_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, int64_t M, int C, float eps);
/* x_out = x + ls * y and h_out = LayerNorm(x_out) in one pass (the end of one residual branch and the norm that opens
 * the next, block.py:84-110); ls may be NULL (= 1).  Same values as cr_scale_residual followed by cr_layernorm. */
int cr_scale_residual_layernorm(cr_ctx* ctx, const void* x, const void* y, const float* ls, const float* gamma,
                                const float* beta, void* x_out, void* h_out, int64_t M, int C, float eps);
/* exact (erf) GELU in place on n bf16 values (dinov2_layers/mlp.py:36). */
int cr_gelu_inplace(cr_ctx* ctx, void* x, int64_t n);
/* out = x + gamma * y, (M,C) bf16, gamma (C) f32 or NULL for 1 (LayerScale + residual, block.py:84-110). */
int cr_scale_residual(cr_ctx* ctx, const void* x, const void* y, const float* gamma, void* out, int64_t M, int C);
/* F.interpolate(mode="bilinear", align_corners=True) on NHWC bf16: x (B,h,w,C) -> y (B,Ho,Wo,C)
 * (util/blocks.py:139, dpt.py:150). */
int cr_resize_bilinear_ac(cr_ctx* ctx, const void* x, void* y, int B, int h, int w, int Ho, int Wo, int C);

/* ---- convolution stack (MFMA, f32 accumulate, NHWC) --------------------- */
/* Two arithmetic modes, chosen per call by `act_f32` (the trailing argument of every entry point that touches
 * activations):
 *   act_f32 = 1  activations, compute weights and activation gradients are float32; contractions run on
 *                v_mfma_f32_16x16x4_f32 (exact f32 fmaf chains, 157 TFLOP/s peak).  This is the REFERENCE'S precision:
 *                tools/train_net.py:184-330 trains in float32, no autocast anywhere -- and the default of the Python host.
 *   act_f32 = 0  the same tensors are bfloat16 (v_mfma_f32_16x16x32_bf16, f32 accumulate, 2.5 PFLOP/s peak): the fast
 *                mode, opt-in; parameters, gradients of parameters and all statistics stay float32 in both modes.
 *   act_f32 = 2  "split" mode: storage exactly as act_f32 = 1 (everything float32); the large contractions run on the
 *                bf16 matrix cores with every f32 operand split EXACTLY into three bf16 values (8 + 8 + 8 significant
 *                bits) and six of the nine cross products accumulated in f32 -- the dropped three are below 2^-23 of
 *                the product, i.e. below f32 rounding -- 96 instead of 256 MFMA cycles per 32 of k.  Layers the split
 *                kernels do not cover (k extent not a multiple of 32, tiny maps, stride-2 backward-data) run the f32 MFMA
 *                kernels of act_f32 = 1.  Forward / backward-data take the pre-split weights (cr_weight_split3) in
 *                `w_split` (NULL = not prepared: f32 MFMA kernels).  Entry points without a contraction treat 2 as 1.
 * "bf16" in the comments below reads "bf16 or f32 by act_f32".
 * Activations are NHWC (channels padded to a multiple of 8).  Conv weights
 * are [Cout][ks*ks][Cin] = the physical (channels_last) layout of a
 * (Cout,Cin,ks,ks) parameter, so state-dict shapes stay the reference's.
 * Replaces torch ATen/cuDNN conv2d + BatchNorm2d + ReLU of
 * cubercnn/modeling/backbone/dla.py:40-68,156-174,233-321 and the detectron2 FPN /
 * StandardRPNHead convolutions wired at dla.py:484-507 (configs/Base.yaml:41-60). */

/* y = relu?(conv(x,w) + bias? + residual?)   ks in {1,3,7}, stride in {1,2}.
 * x (N,H,W,Cin) bf16; w (Cout, ks*ks*Cin) bf16; y (N,Ho,Wo,Cout) bf16 or f32 (out_f32).
 * stats: optional f32 [ceil(M/64)][2][Cout] receiving, per 64 output pixels, the per-channel sum / sum-of-squares
 * of the (pre-residual, pre-ReLU) conv output for BatchNorm (every entry is written; no atomics -> reproducible). */
int cr_conv2d_fwd(cr_ctx* ctx, const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout,
                  int ks, int stride, int pad, const float* bias, const void* residual, int relu,
                  float* stats, int out_f32, int act_f32, const void* w_split);

/* dx (N,H,W,Cin) bf16 from dy (N,Ho,Wo,Cout) bf16; wt = cr_weight_transpose(w); wt_split = cr_weight_split3(wt) or NULL.
 * accumulate: NULL, or a tensor of dx's shape and type that is added in the epilogue (dx = conv^T(dy) + accumulate): the
 * gradient another consumer of the same input already produced -- the fan-in sum of autograd without an add kernel. */
int cr_conv2d_bwd_data(cr_ctx* ctx, const void* dy, const void* wt, void* dx, int N, int H, int W, int Cin,
                       int Cout, int ks, int stride, int pad, int act_f32, const void* wt_split, const void* accumulate);
/* split-mode weight planes of an f32 matrix src (rows, K), K % 32 == 0: dst = bf16 [rows][K/32][3][32] (6 bytes per
 * element): plane 0 / 1 / 2 = the top / middle / low 8 significant bits of each value (their sum is the value, exactly);
 * inside a 64-byte plane row the 16-byte chunk c holds k = {4c..4c+3, 16+4c..16+4c+3} of the 32 (the order in which the
 * split kernels' lanes read the f32 activations).  No counterpart in the reference (cuDNN consumes f32 weights). */
int cr_weight_split3(cr_ctx* ctx, const float* src, void* dst, int64_t rows, int K);
/* the same for many matrices in ONE launch: descs_dev = ndesc records {int64 src_off (floats from src_base), int64 dst_off
 * (bf16 elements from dst_base), int64 item0, int32 rows, int32 K} (32 bytes), item0 = running sum of rows * K / 8 over
 * the preceding records, total_items = that sum over all records.  Offsets must keep 16-byte alignment. */
int cr_weights_split3(cr_ctx* ctx, const float* src_base, void* dst_base, const void* descs_dev, int ndesc,
                      int64_t total_items);
/* dw f32 (Cout, ks*ks*Cin); accumulate=0 zeroes it first (shared RPN-head weights accumulate over levels). */
int cr_conv2d_bwd_weight(cr_ctx* ctx, const void* dy, const void* x, float* dw, int N, int H, int W, int Cin,
                         int Cout, int ks, int stride, int pad, int accumulate, int act_f32);
/* RPN head output -> training tensors in one launch.  The head evaluates objectness and anchor deltas as ONE C = 16-channel
 * 1x1 convolution per level: y_l (B, cells_l, C) f32 with A objectness logits, 4 A deltas, padding per cell.  cr_rpn_unpack
 * writes logits (B, Atot) and deltas (B, Atot, 4) level-concatenated in (cell, anchor) order (detectron2 RPN.forward's permute
 * / flatten / cat [third-party]; rpn.py:153-170) and, when `padded` is non-NULL, the per-level logits padded with -inf to the
 * largest level, (B, L, amax).  cr_rpn_pack_grad: the backward -- dy_l from dlogits / ddeltas (either may be NULL = zero).
 * y_ptrs / dy_ptrs / cells: HOST arrays of L <= 8 entries. */
int cr_rpn_unpack(cr_ctx* ctx, const float* const* y_ptrs, const int* cells, int L, int B, int A, int C, float* logits,
                  float* deltas, float* padded);
int cr_rpn_pack_grad(cr_ctx* ctx, const float* dlogits, const float* ddeltas, float* const* dy_ptrs, const int* cells, int L,
                     int B, int A, int C);
/* Ground truth of a batch of B <= 32 images -> the padded tensors of the static-shape training path, one launch:
 * boxes (B,G,4) f32 zero-padded, classes (B,G) int64 (-2 = padding, -1 = ignore region as in the data), boxes3D (B,G,9)
 * zero-padded, poses (B,G,3,3) identity-padded.  *_ptrs / counts are HOST arrays of B device pointers / object counts
 * (boxes3d / poses entries may both be NULL for an image without 3D annotations).  Feeds what RPNWithIgnore.
 * label_and_sample_anchors (rpn.py:41-50) and ROIHeads3D.label_and_sample_proposals (roi_heads.py:2773-2790) read from
 * `gt_instances`. */
int cr_gt_pack(cr_ctx* ctx, const float* const* boxes_ptrs, const int64_t* const* classes_ptrs,
               const float* const* boxes3d_ptrs, const float* const* poses_ptrs, const int* counts, int B, int G,
               float* boxes, int64_t* classes, float* boxes3d, float* poses);
/* Row-wise top-k: x (rows, n) float32 -> vals (rows, k) sorted descending, idx (rows, k) int64; ties: lower index first; a
 * positive NaN is the largest value (torch.topk's order).  k <= 2048, k <= n, cr_topk_blocks(n, k) * k <= 16384.  ws:
 * rows * cr_topk_blocks(n, k) * k 64-bit words of scratch.  Two launches, no memset, deterministic.  Replaces torch.topk in
 * subsample_labels' multinomial-without-replacement (cubercnn/modeling/proposal_generator/rpn.py:275-328, as the top-k of
 * (IoU + eps) / Exp(1) keys), ROIHeads3D._sample_proposals (roi_heads.py:2737-2771) and detectron2's
 * find_top_rpn_proposals (pre- and post-NMS top-k) [third-party]. */
int cr_topk_blocks(int64_t n, int k);
int cr_topk(cr_ctx* ctx, const float* x, int rows, int64_t n, int k, void* ws, float* vals, int64_t* idx);
/* Loss-divergence guard of tools/train_net.py:202-220 on the device (no host sync in the step): vals (n) = this step's loss
 * terms summed over the ranks, scale = 1 / world size.  Writes red[i] = vals[i] * scale (red may be NULL), total = their sum,
 * flag = 1 when stabilize and (total is not finite or total > tolerance * rolling mean), else 0, and updates the rolling mean
 * `recent` (NaN = not started: starts at 2 * total; moves by gamma on good steps only).  cr_step_counters: the iteration
 * counters of train_net.py:259-266 from the final skip flag (after cr_nonfinite_flag). */
int cr_loss_guard(cr_ctx* ctx, const float* vals, int n, float scale, float* red, float* total, float* recent,
                  int stabilize, float tolerance, float gamma, int* flag);
int cr_step_counters(cr_ctx* ctx, const int* flag, float* explode, float* success);
/* Multi-segment copy (accumulate = 0: dst = src) or accumulate (1: dst += src) of float32 data in ONE launch: descs_dev =
 * ndesc records {const float* src; float* dst; int64 n; int64 item0} (32 bytes; item0 = sum of n over the preceding records),
 * total = sum of n.  Segments of one call must not overlap.  Stacks the five predictor weights / biases of CubeHead
 * (cube_head.py:113-149) and the two of FastRCNNOutputLayers into one GEMM operand and routes the stacked gradient back. */
int cr_multi_seg(cr_ctx* ctx, const void* descs_dev, int ndesc, int64_t total, int accumulate);
/* ReLU backward in one pass: g[i] = y[i] > 0 ? dy[i] : 0 with y the ReLU's output; y, dy, g in the activations' type
 * (act_f32), n elements, 16-byte aligned (nn.ReLU / F.relu backward behind dla.py:40-68, the FPN / RPN-head convolutions and
 * the FC layers of cube_head.py:75,161-168). */
int cr_relu_bwd(cr_ctx* ctx, const void* y, const void* dy, void* g, int64_t n, int act_f32);
/* Fold a frozen BatchNorm2d into the preceding convolution for inference: wf (Cout, K) bf16 = w * gamma / sqrt(var + eps),
 * bias (Cout) f32 = beta - mean * gamma / sqrt(var + eps); w (Cout, K) f32 in the kernels' [Cout][kh][kw][Cin] order.
 * conv(x, wf) + bias (+ residual, ReLU in the conv epilogue) == BatchNorm(conv(x, w)) in eval mode
 * (cubercnn/modeling/backbone/dla.py:40-68: conv -> bn -> relu with BatchNorm = nn.BatchNorm2d in eval()). */
int cr_fold_bn(cr_ctx* ctx, const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
               float eps, void* wf, float* bias, int Cout, int K, int act_f32);

/* same, plus dbias[Cout] += sum over output pixels of dy (bias gradient of the FPN / RPN-head convs), accumulated inside
 * the same kernel from the dy tiles it stages anyway (dbias must be zeroed or hold the running gradient). */
int cr_conv2d_bwd_weight_bias(cr_ctx* ctx, const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W,
                              int Cin, int Cout, int ks, int stride, int pad, int accumulate, int act_f32);
/* Winograd F(2x2, 3x3) transforms for stride-1, pad-1 3x3 convolutions on float32 NHWC maps (csrc/winograd.hip): the 16
 * products over channels in between are 1x1 grouped convolutions (cr_conv2d_fwd_group) on V / U / M -- the arithmetic replaces
 * torch.nn.functional.conv2d as called by detectron2's RPN head / FPN output convolutions in the reference's model.
 *   cr_wino_filter  w (O,3,3,C) KRSC -> U (16,O,C) = G g G^T; backward != 0: U (16,C,O) of the 180-degree-rotated taps
 *                   (backward-data = the same convolution with those filters on dY)
 *   cr_wino_input   xs: n host-array pointers to (N_i,H_i,W_i,C) maps (H_i, W_i even) -> V (16,T,C), T = sum N_i H_i W_i / 4
 *   cr_wino_output  M (16,T,O) -> ys[i] (N_i,H_i,W_i,O) = A^T M A + bias (O, or NULL), ReLU if relu, + accs[i] if given */
int cr_wino_filter(cr_ctx* ctx, const float* w_krsc, float* U, int O, int C, int backward);
/* the products in between as ONE launch: y[b] (R,O) = x[b] (R,K) @ w[b] (O,K)^T for b < batches, float32, operands of batch b
 * at base + b * stride_* elements (O % 128 == 0, K % 32 == 0) */
int cr_gemm_batched_f32(cr_ctx* ctx, const float* x, const float* w, float* y, int R, int K, int O, int batches,
                        int64_t stride_x, int64_t stride_w, int64_t stride_y);
int cr_wino_input(cr_ctx* ctx, int n, const float* const* xs, const int* Ns, const int* Hs, const int* Ws, int C,
                  float* V, int64_t T);
int cr_wino_output(cr_ctx* ctx, int n, const float* M, float* const* ys, const int* Ns, const int* Hs, const int* Ws,
                   int O, int64_t T, const float* bias, int relu, const float* const* accs);
/* weight gradient the same way: dU[k] (O,C) = dM[k]^T V[k] over the tiles (cr_wgrad_batched_f32: the 16 positions in one launch), with
 *   cr_wino_dy           dys: n maps (N_i,H_i,W_i,O) -> dM (16,T,O) = A dY A^T; db_part (16,O), zeroed by the caller, +=
 *                        partial channel sums of dY when given (atomics spread over 16 rows)
 *   cr_wino_filter_grad  dw (O,3,3,C) += G^T dU G from dU (16,O,C); db (O) += column sums of db_part when both given */
int cr_wino_dy(cr_ctx* ctx, int n, const float* const* dys, const int* Ns, const int* Hs, const int* Ws, int O,
               float* dM, int64_t T, float* db_part);
int cr_wino_filter_grad(cr_ctx* ctx, const float* dU, float* dw, int O, int C, const float* db_part, float* db);
/* dw[b] (O,K) += dy[b] (R,O)^T x[b] (R,K) for b < batches in one launch (float32, f32 atomics over the pixel splits; dw zeroed
 * by the caller or holding what is added to); operands of batch b at base + b * stride_* elements */
int cr_wgrad_batched_f32(cr_ctx* ctx, const float* dy, const float* x, float* dw, int R, int K, int O, int batches,
                         int64_t stride_dy, int64_t stride_x, int64_t stride_dw);

/* Grouped launches: n <= 8 independent stride-1 convolutions of one geometry class (same k in {1,3}, pad, Cin, Cout, precision)
 * in ONE grid of 128 x 128 tiles (csrc/conv.hip: k_conv_igemm_dma_grp, k_conv_wgrad_f32_grp) -- the five pyramid levels of
 * detectron2's FPN output convolutions and of the RPN head's convolution (StandardRPNHead applies ONE conv to every level).
 * xs / ws / ys ...: HOST arrays of n device pointers; Ns / Hs / Ws: HOST arrays of the input shapes.  Forward / backward-data:
 * fp32 or bf16 (not the split mode); Cout (backward-data: Cin) % 128 == 0.  cr_conv2d_bwd_weight_group: fp32 only, always
 * accumulates into dws[i] (entries may repeat: shared weights) and, when given, dbiases[i] += column sums of dys[i]. */
int cr_conv2d_fwd_group(cr_ctx* ctx, int n, const void* const* xs, const void* const* ws, void* const* ys, const int* Ns,
                        const int* Hs, const int* Ws, int Cin, int Cout, int ks, int pad, const float* const* biases,
                        const void* const* residuals, int relu, int act_f32);
int cr_conv2d_bwd_data_group(cr_ctx* ctx, int n, const void* const* dys, const void* const* wts, void* const* dxs,
                             const int* Ns, const int* Hs, const int* Ws, int Cin, int Cout, int ks, int pad, int act_f32,
                             const void* const* accumulates);
int cr_conv2d_bwd_weight_group(cr_ctx* ctx, int n, const void* const* dys, const void* const* xs, float* const* dws,
                               float* const* dbiases, const int* Ns, const int* Hs, const int* Ws, int Cin, int Cout, int ks,
                               int pad, int act_f32);
int cr_cast_f32_to_bf16(cr_ctx* ctx, const float* src, void* dst, int64_t n);
/* bias gradient: out[c] += sum_m x[m][c]; x (M,C) bf16 or f32; ws = 1024*C floats; deterministic. */
int cr_colsum_accum(cr_ctx* ctx, const void* x, int is_f32, int64_t M, int C, float* ws, float* out);
/* wt[c][tap][k] (bf16) = w[k][tap][c] (f32) */
int cr_weight_transpose(cr_ctx* ctx, const float* w, void* wt, int Cout, int ks, int Cin, int act_f32);

/* BatchNorm2d, training mode, per-GPU statistics (dla.py:17).  stats from cr_conv2d_fwd.
 * stats = [nparts][2][C] partial sums.  y = relu?((x-mean)*invstd*gamma + beta + residual?); writes
 * mean_invstd [2][C]; updates running stats (may be NULL). */
int cr_bn_fwd(cr_ctx* ctx, const void* x, const float* stats, int nparts, const float* gamma, const float* beta,
              const void* residual, void* y, int64_t M, int C, int relu, float eps, float momentum,
              float* mean_invstd, float* running_mean, float* running_var, int act_f32);
/* g = dy*(out>0 if relu); dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); dres = g (may be NULL);
 * dgamma/dbeta are ACCUMULATED; sums = f32 [1025][2][C] workspace (1024 block partials + the reduced row). */
int cr_bn_bwd(cr_ctx* ctx, const void* dy, const void* out, const void* x, const float* mean_invstd,
              const float* gamma, float* sums, void* dx, void* dres, float* dgamma, float* dbeta, int64_t M,
              int C, int relu, int act_f32);

/* window 2: MaxPool2d(2,2) (dla.py:208); window 1: max_pool2d(k=1,s=2) (dla.py:474). */
int cr_pool2x_fwd(cr_ctx* ctx, const void* x, void* y, int N, int H, int W, int C, int window, int act_f32);
/* accumulate (or NULL): a tensor of dx's shape added in the same pass (gradient fan-in without an add kernel) */
int cr_pool2x_bwd(cr_ctx* ctx, const void* x, const void* dy, void* dx, int N, int H, int W, int C, int window, int act_f32,
                  const void* accumulate);
/* FPN top-down path (detectron2 FPN, fuse_type "sum"): y = lat + nearest_up2x(top); and d/dtop. */
int cr_upsample2x_add(cr_ctx* ctx, const void* lat, const void* top, void* y, int N, int H, int W, int C, int act_f32);
int cr_sum2x2(cr_ctx* ctx, const void* dy, void* dtop, int N, int H, int W, int C, int act_f32);
/* preprocess_image: uint8 (N,3,H,W) -> (x-mean)/std -> NHWC, channels zero-padded to one 16-B chunk per pixel:
 * 3->8 (bf16) or 3->4 (f32).  mean3/std3: HOST floats. */
int cr_preprocess(cr_ctx* ctx, const unsigned char* img, void* y, int N, int H, int W, const float* mean3,
                  const float* std3, int act_f32);

/* ---- detection ops ------------------------------------------------------ */
/* ROIPooler(ROIAlignV2): torchvision roi_align(aligned=True, sampling_ratio=0) with detectron2's FPN level
 * assignment fused in.  feats/grads/Hs/Ws/scales: HOST arrays of length nlev.  rois (R,5) f32
 * [batch,x1,y1,x2,y2]; out / dout (R,PH,PW,C) bf16; grads f32 NHWC maps (atomics; zero first).
 * roi_heads.py:2075-2080,2178,2273. */
int cr_roi_align_fwd(cr_ctx* ctx, const void* const* feats, const int* Hs, const int* Ws, const float* scales,
                     int nlev, int C, const float* rois, int64_t R, int PH, int PW, void* out, int act_f32);
int cr_roi_align_bwd(cr_ctx* ctx, float* const* grads, const int* Hs, const int* Ws, const float* scales,
                     int nlev, int C, const float* rois, int64_t R, int PH, int PW, const void* dout, int act_f32);
/* cr_roi_align_bwd without atomics: every pixel of every level's map is WRITTEN once (no zero fill needed), RoI
 * contributions are summed in RoI order -> bit-reproducible.  One block owns a 16x16-pixel tile x 64 channels of a map.
 * 7x7 pooling, C % 64 == 0; N = images in the batch (rois[:,0] in [0,N)).  roi_heads.py:2178,2273 (backward of the pooler). */
int cr_roi_align_bwd_set(cr_ctx* ctx, float* const* grads, const int* Hs, const int* Ws, const float* scales,
                         int nlev, int C, int N, const float* rois, int64_t R, int PH, int PW, const void* dout,
                         int act_f32);
/* batched NMS over G independent groups; boxes (G,maxn,4) sorted by descending score per group, counts (G)
 * int32; keep (G,maxn) uint8; mask_ws: G*maxn*ceil(maxn/64)*8 bytes.  fast_rcnn.py:105, detectron2 RPN. */
int cr_nms_grouped(cr_ctx* ctx, const float* boxes, const int* counts, int G, int maxn, float thresh,
                   void* mask_ws, unsigned char* keep);

/* ---- Cube R-CNN 3D head: fused decode + disentangled corner losses (K15/K16) -------------
 * cubercnn/modeling/roi_heads/roi_heads.py:2353-2679 (decode, allocentric->egocentric pose, corner sets, L1 /
 * chamfer corner losses, uncertainty weighting), one lane per foreground RoI.
 * inputs: HOST array of 13 device pointers, all f32:
 *   [0] deltas (n,2) [1] z_raw (n) [2] dims_raw (n,3) [3] R_alloc (n,9) [4] uncert (n)      -- head outputs of the RoI's class
 *   [5] proposal boxes (n,4) [6] K (n,4)=[fx,fy,cx,cy] scaled [7] virtual_to_real (n) [8] prior dims mean (n,3)
 *   [9] gt (x,y) 2D centre (n,2) [10] gt z (n) [11] gt dims (n,3) (W,H,L) [12] gt pose (n,9)
 * losses (n,5) = [dims, xy, z, pose, joint], each already x sqrt(2)exp(-uncert) when use_conf;
 * dec (n,17) = [cube_x, cube_y, z, dims(3), R egocentric (9), x3d, y3d].
 * bwd: gl (n,5) = d total / d losses -> gradients w.r.t. inputs [0..4]. */
int cr_cube_loss_fwd(cr_ctx* ctx, const float* const* inputs, int64_t n, int allocentric, int chamfer_pose,
                     int use_conf, int joint, float* losses, float* dec);
int cr_cube_loss_bwd(cr_ctx* ctx, const float* const* inputs, int64_t n, int allocentric, int chamfer_pose,
                     int use_conf, int joint, const float* gl, float* g_dxy, float* g_zr, float* g_dr, float* g_Ra,
                     float* g_u);

/* nn.MaxPool2d(3, stride=2, padding=1) of the torchvision ResNet stem (cubercnn/modeling/backbone/resnet.py:33,49),
 * NHWC bf16, output ((H-1)/2+1, (W-1)/2+1); PyTorch's NaN and first-max tie rules. */
int cr_maxpool3x3s2_fwd(cr_ctx* ctx, const void* x, void* y, int N, int H, int W, int C, int act_f32);
int cr_maxpool3x3s2_bwd(cr_ctx* ctx, const void* x, const void* dy, void* dx, int N, int H, int W, int C, int act_f32);

/* FC weights of the RoI heads (roi_heads.py:2160-2204, cube_head.py:152-202).  w f32 (O, C, HW) in the checkpoint's
 * (c,h,w) column order -> wb bf16 (O, HW, C) for NHWC-flattened inputs (HW = 1: plain cast); and the transpose for the
 * gradient: acc f32 (O, C, HW) += g bf16 (O, HW, C). */
int cr_fc_weight_prepare(cr_ctx* ctx, const float* w, void* wb, int O, int C, int HW, int act_f32);
int cr_fc_grad_accum(cr_ctx* ctx, const void* g, float* acc, int O, int C, int HW, int act_f32);

/* Linear layers of the RoI heads on the hand-written implicit-GEMM kernels (a linear layer over R rows = a 1x1 convolution
 * over a (1,1,R,K) map): replaces nn.Linear / F.linear of FastRCNNConvFCHead + FastRCNNOutputLayers [detectron2, wired at
 * cubercnn/modeling/roi_heads/roi_heads.py:2160-2204] and of CubeHead (cubercnn/modeling/roi_heads/cube_head.py:75,113-149,
 * 161-168; the five predictors run as ONE GEMM over stacked weights).  x (R,K), w (O,K), wt (K,O), y / dy (R,O), dx (R,K) in
 * the activations' type (act_f32); bias, dw (O,K), dbias (O) float32.  K and O multiples of 16; K a multiple of 8.
 * cr_linear_bwd_weight: dw (+)= dy^T x (accumulate = 0 zeroes dw first), dbias += column sums of dy when non-NULL.
 * w_split / wt_split: cr_weight_split3 of w / wt for act_f32 = 2, or NULL.
 * cr_transpose2d: dst (cols,rows) = src (rows,cols)^T for 2-byte (act_f32 = 0) or 4-byte elements: makes wt from w. */
int cr_linear_fwd(cr_ctx* ctx, const void* x, const void* w, const float* bias, void* y, int R, int K, int O, int relu,
                  int out_f32, int act_f32, const void* w_split);
int cr_linear_bwd_data(cr_ctx* ctx, const void* dy, const void* wt, void* dx, int R, int K, int O, int act_f32,
                       const void* wt_split);
int cr_linear_bwd_weight(cr_ctx* ctx, const void* dy, const void* x, float* dw, float* dbias, int R, int K, int O,
                         int accumulate, int act_f32);
int cr_transpose2d(cr_ctx* ctx, const void* src, void* dst, int rows, int cols, int act_f32);

/* multi-tensor form of cr_cast_f32_to_bf16 + cr_weight_transpose: every conv weight of the model in one launch.
 * descs_dev: device array of cr_wdesc (offsets in ELEMENTS from the three base pointers); tiles_dev: device array of
 * ntiles int4 = (tensor index, filter tap, first cout, first cin) covering each tensor in 32x32 (cout x cin) tiles. */
typedef struct cr_wdesc {
    int64_t src_off, dst_off, dstT_off;
    int Cout, KK, Cin, need_T;          /* KK = k*k taps; need_T = 0 skips the transposed copy */
} cr_wdesc;
int cr_weights_prepare(cr_ctx* ctx, const float* src_base, void* dst_base, void* dstT_base, const cr_wdesc* descs_dev,
                       const int* tiles_dev, int ntiles, int act_f32);

/* ---- static-shape training glue of the RPN (3dod_amd/csrc/dense_train.hip) -----------------------------------
 * Batched, sync-free forms of cubercnn/modeling/proposal_generator/rpn.py:41-110 (label_and_sample_anchors with
 * ignore regions), :129-273 (losses) and :275-328 (subsample_labels), and of detectron2's Matcher /
 * Box2BoxTransform / find_top_rpn_proposals [third-party] that the reference calls there.  All pointers are device
 * pointers unless marked HOST.  gt classes: >= 0 object, -1 ignore region, -2 padding; any G >= 1 (the kernels stage the
 * ground-truth rows through LDS 64 at a time). */

/* decode + clip + validity of the per-level top-k candidates.  idx (B,S) int64 = anchor index (or -1 = empty slot),
 * scores (B,S); weights4 HOST [wx,wy,ww,wh]; img_hw (B,2) = (h,w).  boxes/nms_boxes (B,S,4), valid (B,S) u8. */
int cr_rpn_decode_select(cr_ctx* ctx, const float* anchors, const float* deltas, const int64_t* idx,
                         const float* scores, int B, int A, int S, const float* weights4, float scale_clamp,
                         const float* img_hw, float min_size, float* boxes, float* nms_boxes, unsigned char* valid);
/* IoU matching of boxes (R,4) [boxes_per_image = 0] or (B,R,4) [= 1] against gt (B,G,4): max_iou (-1 when the image
 * has no object), argmax (first), max_ioa over ignore regions; best (B,G) u64 or NULL: (iou bits << 32) | ~argmax box. */
int cr_box_match(cr_ctx* ctx, const float* boxes, int boxes_per_image, const float* gt_boxes, const int64_t* gt_classes,
                 int B, int R, int G, float* max_iou, int* argmax, float* max_ioa, unsigned long long* best);
/* Matcher labels + allow_low_quality_matches + sampling keys.  expo (2,B,A) ~ Exp(1); labels3 HOST = Matcher labels.
 * labels_pre (B,A) i8, out (B,A) i32 (-1, or 1 for each gt's arg-max anchor), matched_iou (B,A), keys (2,B,A). */
int cr_rpn_label(cr_ctx* ctx, const float* anchors, const float* gt_boxes, const int64_t* gt_classes,
                 const float* max_iou, const unsigned long long* best, const float* expo, int B, int A, int G, float lo,
                 float hi, const int* labels3, float eps, signed char* labels_pre, int* out, float* matched_iou,
                 float* keys);
/* writes the sampled picks (top-k of the keys; key > 0 = real) into out: positives 1, negatives 0 (rank < n_s - n_pos),
 * negatives inside an ignore region -1 when the image has more than one sampled negative. */
int cr_rpn_scatter(cr_ctx* ctx, const int64_t* pos_idx, const float* pos_key, int KP, const int64_t* neg_idx,
                   const float* neg_key, int KN, int n_s, const float* ioa, float ignore_thresh, int B, int A, int* out);
/* losses + gradients.  sums6 = [cls, loc, n_pos, n_neg, sum sigmoid | pos, sum sigmoid | not pos] (unnormalised);
 * partial_ws: B*ceil(A/256)*6 floats; dlogits (B,A), ddeltas (B,A,4) = d(cls)/d(logits), d(loc)/d(deltas). */
int cr_rpn_loss(cr_ctx* ctx, const float* logits, const float* deltas, const float* anchors, const int* labels,
                const int* matched_idx, const float* gt_boxes, int B, int A, int G, const float* weights4,
                float* partial_ws, float* sums6, float* dlogits, float* ddeltas);

/* ---- static-shape training glue of the RoI heads (3dod_amd/csrc/dense_train.hip) -------------------------------
 * cubercnn/modeling/roi_heads/roi_heads.py:2773-2840 (label_and_sample_proposals with ignore regions) and
 * cubercnn/modeling/roi_heads/fast_rcnn.py:145-194 (FastRCNNOutputs.losses), batched and sync-free. */

/* labels before sampling from cr_box_match's outputs.  valid (B,R) u8 = real proposal; K = number of classes
 * (= background label).  cls (B,R) int64 in {-1, 0..K}; keys (2,B,R) = fg / bg sampling keys (expo (2,B,R) ~ Exp(1)). */
int cr_roi_label(cr_ctx* ctx, const float* max_iou, const int* argmax, const float* max_ioa, const unsigned char* valid,
                 const int64_t* gt_classes, const float* expo, int B, int R, int G, int K, float thr, float ignore_thresh,
                 float eps, int64_t* cls, float* matched_iou, float* keys);
/* stable compaction of [fg picks (KF) | bg picks (KB)] (top-k of the keys) to n_s slots per image, valid picks first.
 * outputs (B,n_s,..): boxes, valid u8, cls (-1 = empty slot), matched gt index; counts (B,2) int32 = [n_fg, n_bg]. */
int cr_roi_compact(cr_ctx* ctx, const int64_t* fg_idx, const float* fg_key, int KF, const int64_t* bg_idx,
                   const float* bg_key, int KB, int n_s, const float* boxes, const int64_t* cls, const int* argmax, int B,
                   int R, float* o_boxes, unsigned char* o_valid, int64_t* o_cls, int64_t* o_gt, int* counts);
/* scores (B*S,K+1), deltas (B*S,K*4) f32.  sums3 = [sum CE over valid rows, sum L1 over fg rows, n_valid];
 * dscores/ddeltas = gradients of those sums; pred (B*S,4) = boxes decoded with each row's own class.
 * partial_ws: ceil(B*S/4)*3 floats. */
int cr_box_loss(cr_ctx* ctx, const float* scores, const float* deltas, const unsigned char* valid, const int64_t* cls,
                const float* prop_boxes, const int64_t* gt_idx, const float* gt_boxes, int B, int S, int G, int K,
                const float* weights4, float scale_clamp, float* partial_ws, float* sums3, float* dscores, float* ddeltas,
                float* pred);

/* ---- glue around the fused 3D-head loss on the static-shape path (roi_heads.py:2237-2679, :2843-2851) ---------
 * raw (n, ld) f32 = fused predictor output, layout5 HOST = column offsets of [deltas 2K, dims 3K, pose6d 6K, z K,
 * uncert K]; cls/valid/gt_idx are (B,S) arrays of which the first kf columns (foreground slots) are used, n = B*kf.
 * buf39 (39*n f32): chunks dxy 2|zr 1|dr 3|Ra 9|u 1|K4 4|v2r 1|prior 3|gt2d 2|gtz 1|gtdims 3|gtR 9, each (n,d):
 * the 12 pointers cr_cube_loss_fwd takes besides the RoI boxes.  validf (n) u8, clsc (n) i32 (clamped class).
 * z_type (ABI 5): MODEL.ROI_CUBE_HEAD.Z_TYPE -- 0 'direct', 1 'sigmoid' (z = 100 sigmoid(raw)), 2 'log' (z = exp(raw)),
 * 3 'clusters' (scaled sigmoid between mean -+ 3 std of the RoI's depth cluster), roi_heads.py:2404-2436; the `zr` chunk holds
 * the DECODED depth (before the virtual-depth factor).  bins = MODEL.ROI_CUBE_HEAD.CLUSTER_BINS: the depth predictor has
 * K * bins columns [bin][class] (layout5[4] = layout5[3] + K * bins) and a RoI reads the bin whose 2D-scale centre z_scales
 * (K, bins) is closest to the diagonal of its proposal box boxes (n,4) (roi_heads.py:2343-2356); z_stats (K, bins, 2) = depth
 * mean / std per (class, bin) for z_type 3; both NULL when bins = 1.  cr_cube_select_bwd / cr_cube_decode_infer take the same. */
int cr_cube_select(cr_ctx* ctx, const float* raw, int ld, const int* layout5, int K, const int64_t* cls,
                   const unsigned char* valid, const int64_t* gt_idx, int B, int S, int kf, int G, const float* gt3d,
                   const float* gtpose, const float* priors, const float* meta, float* buf39, unsigned char* validf,
                   int* clsc, int z_type, int bins, const float* z_scales, const float* z_stats, const float* boxes);
/* transpose of the gather: g_raw (n, ld) dense (zeros off each RoI's class), 6D rotation and clip(0.01) back-propagated;
 * g_usel (n) = extra gradient on the selected uncertainty (from the uncertainty loss term). */
int cr_cube_select_bwd(cr_ctx* ctx, const float* raw, int ld, const int* layout5, int K, int B, int kf,
                       const unsigned char* validf, const int* clsc, const float* g_dxy, const float* g_zr, const float* g_dr,
                       const float* g_Ra, const float* g_u, const float* g_usel, float* g_raw, int z_type,
                       int bins, const float* z_scales, const float* z_stats, const float* boxes);
/* red6 = safely_reduce_losses of [dims, xy, z, pose, joint] (losses (n,5) from cr_cube_loss_fwd) and of the uncertainty
 * over the valid RoIs; cnt6 = entries averaged; stats4 = mean |z|, |dims|, |xy| errors, mean exp(-u) (dec from the fwd). */
int cr_cube_reduce(cr_ctx* ctx, const float* L, const float* buf39, const float* dec, const unsigned char* validf, int n,
                   int inverse_z, float* red6, float* cnt6, float* stats4);
int cr_cube_reduce_bwd(cr_ctx* ctx, const float* L, const float* buf39, const unsigned char* validf, int n, int inverse_z,
                       const float* cnt6, const float* gred6, float* gL, float* gu);

/* ---- test-time filter of the box head on padded proposals, without a host round trip -----------------------------------
 * FastRCNNOutputs.inference -> fast_rcnn_inference_single_image (cubercnn/modeling/roi_heads/fast_rcnn.py:57-116): softmax,
 * score threshold, Box2BoxTransform.apply_deltas + Boxes.clip, class-wise NMS (torchvision batched_nms), the best `topk`
 * detections per image with the full score row of their proposal.  B images x P proposal slots; logits (B*P, ldl >= K+1),
 * deltas (B*P, ldd >= 4*nreg), nreg = K (class-specific regression) or 1; objectness (B*P) or NULL: slots whose objectness is
 * not finite are padding; rows with a non-finite logit / delta / proposal coordinate are dropped (fast_rcnn.py:75-78).
 * cr_det_scores: S (B, P*K) = class probability where it exceeds `thresh`, else -inf; ncand (B+2) int32 candidates per image.
 * (cr_topk of S viewed as (B, P*K) gives val / idx (B,Kc), descending.)
 * cr_det_gather: boxes (B,Kc,4), cls / row (B,Kc) int32 (-1 = empty slot), counts (B) int32.
 * cr_nms_grouped_cls: cr_nms_grouped where a box is only suppressed by a kept box of its own class (cls (G,maxn) int32).
 * cr_det_pick: out_boxes (B,topk,4), out_scores (B,topk), out_cls / out_row (B,topk) int64, out_full (B,topk,K) =
 * scores_full (:96,110), out_count (B,2) int32 = [detections, overflow]; overflow = 1 when the image has more candidates
 * than Kc and fewer than topk survivors among them: repeat it on an unbounded path.  topk <= 512. */
int cr_det_scores(cr_ctx* ctx, const float* logits, int ldl, const float* deltas, int ldd, const float* prop_boxes,
                  const float* objectness, int B, int P, int K, int nreg, float thresh, float* S, int32_t* ncand);
int cr_det_gather(cr_ctx* ctx, const float* val, const int64_t* idx, const float* deltas, int ldd, const float* prop_boxes,
                  const float* img_hw, int B, int P, int K, int nreg, int Kc, const float* weights4, float scale_clamp,
                  float* boxes, int32_t* cls, int32_t* row, int32_t* counts);
int cr_nms_grouped_cls(cr_ctx* ctx, const float* boxes, const int32_t* cls, const int* counts, int G, int maxn, float thresh,
                       void* mask_ws, unsigned char* keep);
int cr_det_pick(cr_ctx* ctx, const unsigned char* keep, const int32_t* counts, const int32_t* ncand, const float* val,
                const float* boxes, const int32_t* cls, const int32_t* row, const float* logits, int ldl, int B, int P, int K,
                int Kc, int topk, float* out_boxes, float* out_scores, int64_t* out_cls, int64_t* out_row, float* out_full,
                int32_t* out_count);

/* ---- fused losses of the weakly supervised 3D head on the static (B, kf) foreground slots --------------------------
 * ROIHeads3DScore._forward_cube in training mode (cubercnn/modeling/roi_heads/roi_heads.py:1366-1760): decode, cuboid,
 * projected + clamped corners and their hull, the per-RoI terms and their reduction.  Term index (bit of `terms`):
 * 0 iou (GIoU :1585-1590) | 1 pose (pairwise alignment :1055-1074) | 2 normal (pose_ground :1610-1622) | 3 z (depth search
 * :1151-1194) | 4 pseudo_gt_z (:1196-1232 window median, or :1256-1279 depth under the centre) | 5,6,7 dims_w/h/l (:1234-1254).
 * inputs: HOST array of 8 device pointers [dxy, zr, dr, Ra, u, v2r, prior_mean (chunks of cr_cube_select's buf39), src_boxes
 * (n,4)], n = B*kf; validf / clsc from cr_cube_select; gt_idx (B,S) int64, gt_boxes (B,G,4); prior_std (K,3) or NULL;
 * table (B,20) = [K / ratio row-major with K[2][2] = 1 (9) | clamp x_lo x_hi y_lo y_hi of Cubes.get_bube_corners (spaces.py:
 * 240-243) | ground-map confidence | image height, width | 4 columns the kernels do not read]; normals (B,3) or NULL (needed by term 2).  kf <= 256.
 * Outputs: Lraw (n,8) unweighted terms (1 and 4 are filled by cr_weak_loss_reduce), dec (n,17) = [cube_x, cube_y, z, dims(3),
 * R(9), x3d, y3d], pbox (n,4) projected hull, ibox (n,4) int32 window of the depth median (0,0,0,0 = none), pimg (B,2) =
 * [pose alignment mean, valid slots] per image. */
int cr_weak_loss_fwd(cr_ctx* ctx, const float* const* inputs, const unsigned char* validf, const int32_t* clsc,
                     const int64_t* gt_idx, const float* gt_boxes, const float* prior_std, const float* table,
                     const float* normals, int B, int kf, int S, int G, int allocentric, int terms, float* Lraw, float* dec,
                     float* pbox, int32_t* ibox, float* pimg);
/* uncertainty weighting (:1700-1712) and safely_reduce_losses (:2843-2851) of every term over the valid slots.
 * red_inputs: HOST array of 4 device pointers [u, gt2d, gtz, gtdims] (buf39 chunks); depth (B,H,W) padded depth maps (needed
 * when pgz_mode != 0); med (n) = cr_box_median of ibox (pgz_mode 1); pgz_mode 0 none / 1 window median, targets in the
 * reference's [with area..., without area...] order per image / 2 depth under the predicted centre; weights (9) HOST = weights
 * of the logged total (8 terms, then the extra ground-confidence factor of term 2).
 * red (9) = mean of term x sqrt(2) exp(-u) over the valid finite entries, [8] = mean uncertainty; cnt (9); stats (8) = z_error,
 * dims_error, xy_error, z_close, 2D IoU, conf, total_3D_loss / loss_w_3d, valid slots; aux (1) = images-with-one-slot + 1 (0 when
 * every image has exactly one: the reference drops the pose term then, here it is 0); ztgt (n) pseudo depth targets. */
int cr_weak_loss_reduce(cr_ctx* ctx, const float* const* red_inputs, const unsigned char* validf, const float* table,
                        const float* depth, int H, int W, const float* med, const float* gt_boxes, const int64_t* gt_idx, int B,
                        int kf, int S, int G, int terms, int pgz_mode, const float* weights, float* Lraw, const float* dec,
                        const float* pbox, const int32_t* ibox, const float* pimg, float* ztgt, float* red, float* cnt,
                        float* stats, float* aux);
/* gradients of sum_k gred[k] * red[k] w.r.t. the selected head outputs, in the form cr_cube_select_bwd takes. */
int cr_weak_loss_bwd(cr_ctx* ctx, const float* const* inputs, const unsigned char* validf, const int32_t* clsc,
                     const int64_t* gt_idx, const float* gt_boxes, const float* prior_std, const float* table,
                     const float* normals, int B, int kf, int S, int G, int allocentric, int terms, const float* gred,
                     const float* cnt, const float* aux, const float* Lraw, const float* dec, const float* ztgt,
                     const float* pimg, float* g_dxy, float* g_zr, float* g_dr, float* g_Ra, float* g_u);

/* inference decode of the 3D head for n kept detections (roi_heads.py:2353-2436,2682-2735) from the fused predictor
 * output raw (n, ld) (layout as in cr_cube_select).  cls (n) int64, img (n) int32, boxes (n,4); meta6 (B,6) =
 * [fx,fy,cx,cy of K/ratio, virtual_to_real, ratio]; priors (K,3) or NULL.
 * out42 (n,42) = [x3d,y3d,z | w,h,l | 2D centre x ratio | exp(-uncert) | R (9) | corners (8,3)]. */
int cr_cube_decode_infer(cr_ctx* ctx, const float* raw, int ld, const int* layout5, int K, const int64_t* cls,
                         const int* img, const float* boxes, const float* meta6, const float* priors, int n,
                         int allocentric, float* out42, int z_type, int bins, const float* z_scales, const float* z_stats);

/* ---- exact IoU of oriented 3D boxes -----------------------------------------------------------
 * replaces pytorch3d box3d_overlap / _C.iou_box3d [third-party] at ProposalNetwork/utils/utils.py:194-210,
 * cubercnn/evaluation/omni3d_evaluation.py:155, cubercnn/modeling/roi_heads/roi_heads.py:518,526,1563.
 * boxes1 (N,8,3), boxes2 (M,8,3) f32 corners in pytorch3d's order -> vol (N,M), iou (N,M). */
int cr_box3d_overlap(cr_ctx* ctx, const float* boxes1, const float* boxes2, int N, int M, float* vol, float* iou);

/* ---- optimizer (tools/train_net.py:233-266, cubercnn/solver/build.py:50-56) ------------- */
int cr_nonfinite_flag(cr_ctx* ctx, const float* g, int64_t n, int* flag);
/* SGD momentum on flat f32 buffers; skipped on device when *skip_flag != 0 (may be NULL).  The learning rate is
 * lr * (*lr_scale_dev) when lr_scale_dev (a device float: the schedule's current factor, WarmupMultiStepLR of
 * tools/train_net.py:410-414) is given, so a launch captured in a HIP graph follows the schedule. */
int cr_sgd_step(cr_ctx* ctx, float* p, const float* g, float* m, int64_t n, float lr, const float* lr_scale_dev,
                float momentum, float weight_decay, float grad_scale, const int* skip_flag);
/* the same update with nesterov = True (SOLVER.NESTEROV, solver/build.py:50-56): p -= lr * (g' + momentum * m_new) */
int cr_sgd_step_nesterov(cr_ctx* ctx, float* p, const float* g, float* m, int64_t n, float lr, const float* lr_scale_dev,
                float momentum, float weight_decay, float grad_scale, const int* skip_flag);

/* SOLVER.CLIP_GRADIENTS -- detectron2 maybe_add_gradient_clipping as called by cubercnn/solver/build.py:68 -- on the flat
 * gradient in place, BEFORE the update; grad_scale (1 / world size after the all-reduce) is folded in, the update that follows
 * runs with grad_scale 1.
 *   cr_grad_clip_value: g = clamp(g * grad_scale, -clip_value, clip_value)                  (CLIP_TYPE "value")
 *   cr_grad_clip_norm:  per parameter, norm = ||g * grad_scale||_norm_type (2, 1, inf or any p > 0),
 *                       g = g * grad_scale * min(1, max_norm / (norm + 1e-6))                (CLIP_TYPE "norm")
 *   starts / counts (nparam) int64 device arrays: offset and element count of every parameter inside g;
 *   partial: nparam * 16 floats of device scratch (fixed-order two-pass reduction: bit-reproducible). */
int cr_grad_clip_value(cr_ctx* ctx, float* g, int64_t n, float clip_value, float grad_scale);
int cr_grad_clip_norm(cr_ctx* ctx, float* g, const int64_t* starts, const int64_t* counts, int nparam, float max_norm,
                      float norm_type, float grad_scale, float* partial);
/* torch.optim.Adam / AdamW with the reference's eps (cubercnn/solver/build.py:57-64: 'adam', 'adam+amsgrad', 'adamw',
 * 'adamw+amsgrad'), fused like cr_sgd_step: one launch per hyper-parameter segment of the flat buffers; decoupled = 1: AdamW.
 * max_exp_avg_sq: amsgrad state or NULL.  step: device float = number of applied updates, advanced by cr_adam_tick once per
 * optimizer step before the segments; both kernels do nothing when *skip_flag != 0 (train_net.py:246). */
int cr_adam_tick(cr_ctx* ctx, float* step, const int* skip_flag);
int cr_adam_step(cr_ctx* ctx, float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, int64_t n,
                 float lr, const float* lr_scale_dev, float beta1, float beta2, float eps, float weight_decay,
                 float grad_scale, int decoupled, const float* step, const int* skip_flag);

#ifdef __cplusplus
}
#endif
#endif /* CR3DOD_H */
