// Kernels of the weakly supervised 3D head (ROIHeads3DScore, cubercnn/modeling/roi_heads/roi_heads.py:664-1946).
//
// k_box_median: median depth inside each projected 2D box -- the pseudo ground truth of `pseudo_gt_z_box_loss`
// (roi_heads.py:1196-1232), which the reference computes with one torch.median call per box in a Python loop.
// One block per box; exact selection of the lower median (torch.median's choice for an even count) by a 4-pass
// 8-bit radix select over the order-preserving integer image of the float bits: every pass histograms the window's
// elements that still match the prefix found so far (LDS atomics on run-length-compressed counts), a single wave scans
// the 256 bins.  The window is
// re-read from L2 four times (a 512x512 map is 1 MB); no sorting, no scratch memory.  Integer work, bit-exact.
#include "cr_common.h"

__device__ __forceinline__ unsigned f2ord(float f) {      // monotone float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

#define MED_T 1024
__global__ __launch_bounds__(MED_T) void k_box_median(const float* __restrict__ depth, int B, int H, int W,
                                                    const int* __restrict__ boxes, const int* __restrict__ img, int n,
                                                    float* __restrict__ out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_k;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= n) return;
    int x1 = boxes[4 * b + 0], y1 = boxes[4 * b + 1], x2 = boxes[4 * b + 2], y2 = boxes[4 * b + 3];
    const int im = img[b];
    // python slice semantics of depth[y1:y2, x1:x2] for non-negative bounds
    x1 = min(max(x1, 0), W); x2 = min(max(x2, 0), W); y1 = min(max(y1, 0), H); y2 = min(max(y2, 0), H);
    const int bw = max(x2 - x1, 0), bh = max(y2 - y1, 0);
    const int cnt = bw * bh;
    if (cnt == 0 || im < 0 || im >= B) {
        if (tid == 0) out[b] = __uint_as_float(0x7fc00000u);       // empty window: NaN
        return;
    }
    const float* base = depth + ((size_t)im * H + y1) * W + x1;
    if (tid == 0) { s_prefix = 0u; s_k = (unsigned)((cnt - 1) >> 1); }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned hi_mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        // run-length accumulation: neighbouring depths share their high bytes, so in the first passes nearly every
        // element of a thread falls into the same bin -- counting the run in a register and touching LDS only when the
        // bin changes removes the (otherwise fully serialised) same-address atomics
        int cur = -1;
        unsigned run = 0;
        for (int r = tid >> 6; r < bh; r += MED_T / 64) {                 // one wave per row, lanes along the row
            const float* row = base + (size_t)r * W;
            for (int c = tid & 63; c < bw; c += 64) {
                const unsigned key = f2ord(row[c]);
                const int bin = (key & hi_mask) == prefix ? (int)((key >> shift) & 255u) : -1;
                if (bin == cur) {
                    ++run;
                } else {
                    if (cur >= 0) atomicAdd(&hist[cur], run);
                    cur = bin;
                    run = 1;
                }
            }
        }
        if (cur >= 0) atomicAdd(&hist[cur], run);
        __syncthreads();
        if (tid < 64) {
            // wave scan over 256 bins: each lane owns 4 consecutive bins
            const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const unsigned mine = h0 + h1 + h2 + h3;
            unsigned incl = mine;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off, 64);
                if (tid >= off) incl += v;
            }
            const unsigned excl = incl - mine, k = s_k;
            if (k >= excl && k < incl) {                   // exactly one lane
                unsigned run = excl, bin;
                if (k < run + h0) bin = 0;
                else if (k < (run += h0) + h1) bin = 1;
                else if (k < (run += h1) + h2) bin = 2;
                else { run += h2; bin = 3; }
                s_prefix = prefix | ((4u * tid + bin) << shift);
                s_k = k - run;
            }
        }
        __syncthreads();
    }
    if (tid == 0) out[b] = ord2f(s_prefix);
}

extern "C" int cr_box_median(cr_ctx* ctx, const float* depth, int B, int H, int W, const int32_t* boxes,
                             const int32_t* img, int n, float* out) {
    CR_CHECK_ARG(ctx && B >= 0 && H >= 0 && W >= 0 && n >= 0, "cr_box_median: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(depth && boxes && img && out, "cr_box_median: NULL pointer");
    CR_CHECK_ARG((int64_t)H * W < (1ll << 31), "cr_box_median: map too large");
    hipLaunchKernelGGL(k_box_median, dim3(n), dim3(MED_T), 0, ctx->stream, depth, B, H, W, boxes, img, n, out);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
