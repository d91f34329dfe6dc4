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
 * cv2.boxPoints(minAreaRect(mask contour)) quad, or NULL for the reference's
 * no-contour fallback (scorefunction.py:69-75).
 * Outputs (any of the first six may be NULL = not written):
 *   out_corners (N,P,8,2) out_boxes (N,P,4) out_iou/out_dim/out_corner/
 *   out_combined (N,P); out_argmax (N) int64; out_best (N) = combined[argmax].
 * P <= 4096.  NaN/Inf are data (unguarded z<=0, 0/0 ratios) as in the reference;
 * argmax follows np.argmax (first maximal index, NaN maximal). */
int cr_cubes_project_score(cr_ctx* ctx, const float* cubes, int64_t N, int64_t P,
                           const float* K, int k_per_object, float im_w, float im_h,
                           const float* ref_boxes, const float* prior_mu, const float* prior_sigma,
                           const float* rect_pts,
                           float* out_corners, float* out_boxes, float* out_iou, float* out_dim,
                           float* out_corner, float* out_combined, int64_t* out_argmax, float* out_best);

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

/* K21: Plane.fit_parallel -- ProposalNetwork/utils/plane.py:79-134 with the
 * sampled index triples given.  pts (Q,3); triples (T,3) int32;
 * out_neg_eq (4) = -(a,b,c,d) as the reference returns; out_counts (T) int32
 * inlier counts (workspace + diagnostic); out_best (2) int32 = {index, count}. */
int cr_ransac_plane(cr_ctx* ctx, const float* pts, int64_t Q, const int32_t* triples, int64_t T,
                    float thresh, float* out_neg_eq, int32_t* out_counts, int32_t* out_best);

#ifdef __cplusplus
}
#endif
#endif /* CR3DOD_H */
