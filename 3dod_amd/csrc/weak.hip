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
        auto count = [&](float v) {
            const unsigned key = f2ord(v);
            const int bin = (key & hi_mask) == prefix ? (int)((key >> shift) & 255u) : -1;
            if (bin == cur) {
                ++run;
            } else {
                if (cur >= 0) atomicAdd(&hist[cur], run);
                cur = bin;
                run = 1;
            }
        };
        // one wave per row pair, lanes along the rows; the eight loads of a step (2 rows x 4 column chunks) are issued before the
        // first one is consumed -- the window comes out of L2 / HBM at ~1 us per dependent round trip otherwise
        const int lane = tid & 63;
        for (int r = (tid >> 6) * 2; r < bh; r += MED_T / 32) {
            const float* row0 = base + (size_t)r * W;
            const bool two = r + 1 < bh;
            const float* row1 = two ? row0 + W : row0;
            for (int c0 = lane; c0 < bw; c0 += 256) {
                float v0[4], v1[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + 64 * k;
                    const bool ok = c < bw;
                    v0[k] = ok ? row0[c] : 0.f;
                    v1[k] = ok && two ? row1[c] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c0 + 64 * k < bw) count(v0[k]);
                if (two) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c0 + 64 * k < bw) count(v1[k]);
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

// ---- the same selection with every window spread over MED_SPLIT workgroups ----------------------------------------------------
// A projected box clipped to the image is often most of the image (262k pixels at 512 x 512); one workgroup per window then
// leaves the chip to the few largest windows.  Here pass p of the radix select is one launch over (windows x MED_SPLIT) workgroups
// that histogram their rows in LDS and add the non-empty bins to the window's global histogram; the next launch starts by
// scanning that histogram (every workgroup for itself, workgroup 0 of the window records the state), the fifth one writes the
// result.  Workspace: hist [4][n][256] u32 (zeroed by k_med_zero) + state [5][n][2] u32.
#define MED_SPLIT 64
#define MED_PT 256
__global__ void k_med_zero(unsigned* __restrict__ p, long count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = 0u;
}

template <int PASS>
__global__ __launch_bounds__(MED_PT) void k_med_pass(const float* __restrict__ depth, int B, int H, int W, const int* __restrict__ boxes,
                                                     const int* __restrict__ img, int n, unsigned* __restrict__ hist,
                                                     unsigned* __restrict__ state, float* __restrict__ out) {
    __shared__ unsigned sh[256];
    __shared__ unsigned s_prefix, s_k;
    const int b = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    int x1 = boxes[4 * b + 0], y1 = boxes[4 * b + 1], x2 = boxes[4 * b + 2], y2 = boxes[4 * b + 3];
    const int im = img[b];
    x1 = min(max(x1, 0), W); x2 = min(max(x2, 0), W); y1 = min(max(y1, 0), H); y2 = min(max(y2, 0), H);
    const int bw = max(x2 - x1, 0), bh = max(y2 - y1, 0);
    const int cnt = bw * bh;
    if (cnt == 0 || im < 0 || im >= B) {
        if (PASS == 4 && tid == 0) out[b] = __uint_as_float(0x7fc00000u);       // empty window: NaN
        return;
    }
    // ---- state before this pass: the previous state refined by the previous pass's histogram
    unsigned prefix = 0u, k = (unsigned)((cnt - 1) >> 1);
    if (PASS > 0) {
        if (PASS > 1) { prefix = state[((size_t)(PASS - 1) * n + b) * 2]; k = state[((size_t)(PASS - 1) * n + b) * 2 + 1]; }
        if (tid < 64) {
            const unsigned* h = hist + ((size_t)(PASS - 1) * n + b) * 256;
            const unsigned h0 = h[4 * tid], h1 = h[4 * tid + 1], h2 = h[4 * tid + 2], h3 = h[4 * tid + 3];
            const unsigned mine = h0 + h1 + h2 + h3;
            unsigned incl = mine;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off, 64);
                if (tid >= off) incl += v;
            }
            const unsigned excl = incl - mine;
            if (k >= excl && k < incl) {                   // exactly one lane
                unsigned run = excl, bin;
                if (k < run + h0) bin = 0;
                else if (k < (run += h0) + h1) bin = 1;
                else if (k < (run += h1) + h2) bin = 2;
                else { run += h2; bin = 3; }
                s_prefix = prefix | ((4u * tid + bin) << (24 - 8 * (PASS - 1)));
                s_k = k - run;
            }
        }
        __syncthreads();
        prefix = s_prefix; k = s_k;
        if (part == 0 && tid == 0) { state[((size_t)PASS * n + b) * 2] = prefix; state[((size_t)PASS * n + b) * 2 + 1] = k; }
    }
    if (PASS == 4) {
        if (tid == 0) out[b] = ord2f(prefix);
        return;
    }
    // ---- histogram of this workgroup's rows
    sh[tid] = 0u;
    __syncthreads();
    const int shift = 24 - 8 * (PASS < 4 ? PASS : 3);
    const unsigned hi_mask = PASS == 0 ? 0u : (0xffffffffu << (shift + 8));
    const float* base = depth + ((size_t)im * H + y1) * W + x1;
    int cur = -1;
    unsigned run = 0;
    auto count = [&](float v) {
        const unsigned key = f2ord(v);
        const int bin = (key & hi_mask) == prefix ? (int)((key >> shift) & 255u) : -1;
        if (bin == cur) {
            ++run;
        } else {
            if (cur >= 0) atomicAdd(&sh[cur], run);
            cur = bin;
            run = 1;
        }
    };
    const int lane = tid & 63;
    for (int r = (part * (MED_PT / 64) + (tid >> 6)) * 2; r < bh; r += MED_SPLIT * (MED_PT / 64) * 2) {
        const float* row0 = base + (size_t)r * W;
        const bool two = r + 1 < bh;
        const float* row1 = two ? row0 + W : row0;
        for (int c0 = lane; c0 < bw; c0 += 256) {
            float v0[4], v1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + 64 * q;
                const bool ok = c < bw;
                v0[q] = ok ? row0[c] : 0.f;
                v1[q] = ok && two ? row1[c] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c0 + 64 * q < bw) count(v0[q]);
            if (two) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c0 + 64 * q < bw) count(v1[q]);
            }
        }
    }
    if (cur >= 0) atomicAdd(&sh[cur], run);
    __syncthreads();
    if (sh[tid]) atomicAdd(&hist[((size_t)PASS * n + b) * 256 + tid], sh[tid]);
}

extern "C" int cr_box_median(cr_ctx* ctx, const float* depth, int B, int H, int W, const int32_t* boxes,
                             const int32_t* img, int n, float* out) {
    CR_CHECK_ARG(ctx && B >= 0 && H >= 0 && W >= 0 && n >= 0, "cr_box_median: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(depth && boxes && img && out, "cr_box_median: NULL pointer");
    CR_CHECK_ARG((int64_t)H * W < (1ll << 31), "cr_box_median: map too large");
    const size_t hist_words = (size_t)4 * n * 256, state_words = (size_t)5 * n * 2;
    if ((hist_words + state_words) * sizeof(unsigned) > ctx->ws_bytes || n > 65535) {       // one workgroup per window
        hipLaunchKernelGGL(k_box_median, dim3(n), dim3(MED_T), 0, ctx->stream, depth, B, H, W, boxes, img, n, out);
        CR_LAUNCH_CHECK();
        return CR_OK;
    }
    unsigned* hist = (unsigned*)ctx->ws;
    unsigned* state = hist + hist_words;
    hipLaunchKernelGGL(k_med_zero, dim3((unsigned)cr_cdiv((int64_t)hist_words, 256)), dim3(256), 0, ctx->stream, hist, (long)hist_words);
    const dim3 grid((unsigned)n, MED_SPLIT), blk(MED_PT);
    hipLaunchKernelGGL((k_med_pass<0>), grid, blk, 0, ctx->stream, depth, B, H, W, boxes, img, n, hist, state, out);
    hipLaunchKernelGGL((k_med_pass<1>), grid, blk, 0, ctx->stream, depth, B, H, W, boxes, img, n, hist, state, out);
    hipLaunchKernelGGL((k_med_pass<2>), grid, blk, 0, ctx->stream, depth, B, H, W, boxes, img, n, hist, state, out);
    hipLaunchKernelGGL((k_med_pass<3>), grid, blk, 0, ctx->stream, depth, B, H, W, boxes, img, n, hist, state, out);
    hipLaunchKernelGGL((k_med_pass<4>), dim3((unsigned)n, 1), blk, 0, ctx->stream, depth, B, H, W, boxes, img, n, hist, state, out);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Convex hull of the 8 projected cuboid corners in the order the reference produces it (jarvis_march,
// ProposalNetwork/utils/utils.py:424-470, incl. its handling of duplicate points :427-433 and its tie rules), one thread
// per RoI.  out_order (n,8): indices into the (bumped) points, hull vertices first; out_count (n); out_bump (n,8): the
// constant the reference adds to BOTH coordinates of a duplicated point.  The march is capped at 8 steps (a hull of 8
// points has at most 8 vertices; the reference would loop forever on inputs that never return to the start).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_hull8(const float* __restrict__ pts, int n, int* __restrict__ out_order, int* __restrict__ out_count,
                        float* __restrict__ out_bump) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float x[8], y[8], bump[8];
    for (int i = 0; i < 8; ++i) { x[i] = pts[(r * 8 + i) * 2]; y[i] = pts[(r * 8 + i) * 2 + 1]; bump[i] = 0.f; }
    // duplicates: every index i that equals some later point; the k-th such index (ascending) gets + (k+1)
    int k = 0;
    for (int i = 0; i < 7; ++i) {
        bool dup = false;
        for (int j = i + 1; j < 8; ++j) dup |= (x[i] == x[j] && y[i] == y[j]);
        if (dup) { ++k; bump[i] = (float)k; }
    }
    for (int i = 0; i < 8; ++i) { x[i] += bump[i]; y[i] += bump[i]; out_bump[r * 8 + i] = bump[i]; }
    // start: smallest x, among equals the largest y (first one on a further tie)
    float minx = x[0];
    for (int i = 1; i < 8; ++i) minx = fminf(minx, x[i]);
    int start = -1, ncand = 0;
    for (int i = 0; i < 8; ++i)
        if (x[i] == minx) {
            ++ncand;
            if (start < 0 || y[i] > y[start]) start = i;
        }
    if (ncand == 1)
        for (int i = 0; i < 8; ++i) if (x[i] == minx) { start = i; break; }
    int res[9];
    int m = 0;
    res[m++] = start;
    int l = start;
    for (int step = 0; step < 8; ++step) {
        int q = (l + 1) % 8;
        for (int i = 0; i < 8; ++i) {
            if (i == l) continue;
            const float d = (x[i] - x[l]) * (y[q] - y[l]) - (y[i] - y[l]) * (x[q] - x[l]);
            const float di = (x[i] - x[l]) * (x[i] - x[l]) + (y[i] - y[l]) * (y[i] - y[l]);
            const float dq = (x[q] - x[l]) * (x[q] - x[l]) + (y[q] - y[l]) * (y[q] - y[l]);
            if (d > 0.f || (d == 0.f && di > dq)) q = i;
        }
        l = q;
        if (l == start) break;
        if (m < 8) res[m++] = q; else break;
    }
    out_count[r] = m;
    for (int i = 0; i < 8; ++i) out_order[r * 8 + i] = i < m ? res[m - 1 - i] : 0;      // the reference returns the flipped list
}

extern "C" int cr_hull8(cr_ctx* ctx, const float* pts, int n, int32_t* order, int32_t* count, float* bump) {
    CR_CHECK_ARG(ctx && n >= 0, "cr_hull8: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(pts && order && count && bump, "cr_hull8: NULL pointer");
    hipLaunchKernelGGL(k_hull8, dim3((unsigned)cr_cdiv(n, 64)), dim3(64), 0, ctx->stream, pts, n, order, count, bump);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Soft polygon mask + focal loss of `segment_loss` (roi_heads.py:1030-1053 with fill_polygon, utils.py:472-502):
//   m(px) = prod_e clamp((X - v1x)(v2y - v1y) - (Y - v1y)(v2x - v1x), 0, 1) over the hull edges e = (v1 -> v2),
//   loss  = mean_px sigmoid_focal_loss(inputs = gt mask (0/1 used as a logit), targets = m; alpha 0.25, gamma 2)
// -- yes, the ground-truth mask is the `inputs` argument in the reference.  One pass per RoI over the H x W pixels, forward
// and the gradient w.r.t. the (<= 8) hull vertices in the same kernel (the reference materialises 8 full-size float masks
// per RoI).  Ties of the clamp (argument exactly 0 or 1) pass half the gradient, like torch.max / torch.min.
// One workgroup per RoI over the hull's bounding box; mask_ones (Nm) = number of set pixels of every mask.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_polygon_focal(const float* __restrict__ hull, const int* __restrict__ count,
                                                       const unsigned char* __restrict__ masks, const int* __restrict__ mask_idx,
                                                       const int* __restrict__ mask_ones, int H, int W,
                                                       float* __restrict__ loss, float* __restrict__ grad, int want_grad) {
    __shared__ float red[4][18];
    const int r = blockIdx.x;
    const int k = count[r];
    float vx[8], vy[8];
    for (int e = 0; e < 8; ++e) { vx[e] = hull[(r * 8 + e) * 2]; vy[e] = hull[(r * 8 + e) * 2 + 1]; }
    const int mi = mask_idx[r];
    const unsigned char* mk = masks + (size_t)mi * H * W;
    // Outside the hull's bounding box some edge has a negative argument, so the polygon value is exactly 0 there and the
    // focal term depends on the mask bit alone (two constants) with zero gradient: only the box is visited.
    float fx0 = vx[0], fx1 = vx[0], fy0 = vy[0], fy1 = vy[0];
    for (int e = 1; e < 8; ++e)
        if (e < k) { fx0 = fminf(fx0, vx[e]); fx1 = fmaxf(fx1, vx[e]); fy0 = fminf(fy0, vy[e]); fy1 = fmaxf(fy1, vy[e]); }
    const bool finite = (fx0 == fx0) && (fx1 == fx1) && (fy0 == fy0) && (fy1 == fy1) && k >= 1;
    const int bx0 = finite ? max((int)floorf(fmaxf(fx0, -1.f)) - 1, 0) : 0, bx1 = finite ? min((int)ceilf(fminf(fx1, (float)W)) + 1, W - 1) : W - 1;
    const int by0 = finite ? max((int)floorf(fmaxf(fy0, -1.f)) - 1, 0) : 0, by1 = finite ? min((int)ceilf(fminf(fy1, (float)H)) + 1, H - 1) : H - 1;
    const int bw = max(bx1 - bx0 + 1, 0), bh = max(by1 - by0 + 1, 0);
    const int total = H * W;
    float acc = 0.f, ones_in = 0.f, g[16];
    for (int i = 0; i < 16; ++i) g[i] = 0.f;
    const float inv_total = 1.f / (float)total;
    for (int q = threadIdx.x; q < bw * bh; q += 256) {
        const int yy = by0 + q / bw, xx = bx0 + (q - (q / bw) * bw);
        const int p = yy * W + xx;
        const float Y = (float)yy, X = (float)xx;
        float c[8], dc[8], m = 1.f;
        for (int e = 0; e < 8; ++e) {
            c[e] = 1.f; dc[e] = 0.f;
            if (e < k) {
                const int e2 = e + 1 == k ? 0 : e + 1;
                const float raw = (X - vx[e]) * (vy[e2] - vy[e]) - (Y - vy[e]) * (vx[e2] - vx[e]);
                c[e] = fminf(fmaxf(raw, 0.f), 1.f);
                dc[e] = (raw > 0.f && raw < 1.f) ? 1.f : ((raw == 0.f || raw == 1.f) ? 0.5f : 0.f);
                m *= c[e];
            }
        }
        const float x = mk[p] ? 1.f : 0.f;
        ones_in += x;
        const float ps = 1.f / (1.f + __expf(-x));
        const float ce = fmaxf(x, 0.f) - x * m + log1pf(__expf(-fabsf(x)));
        const float pt = ps * m + (1.f - ps) * (1.f - m);
        const float om = 1.f - pt;
        const float at = 0.25f * m + 0.75f * (1.f - m);
        acc += at * ce * om * om * inv_total;
        if (want_grad) {
            const float dL = (-0.5f * ce * om * om + at * (-x) * om * om + at * ce * 2.f * om * (-(2.f * ps - 1.f))) * inv_total;
            for (int e = 0; e < 8; ++e) {
                if (e < k && dc[e] != 0.f) {
                    float others = 1.f;
                    for (int j = 0; j < 8; ++j) if (j != e && j < k) others *= c[j];
                    const float dr = dL * others * dc[e];
                    const int e2 = e + 1 == k ? 0 : e + 1;
                    g[2 * e] += dr * (-(vy[e2] - vy[e]) + (Y - vy[e]));
                    g[2 * e + 1] += dr * (-(X - vx[e]) + (vx[e2] - vx[e]));
                    g[2 * e2] += dr * (-(Y - vy[e]));
                    g[2 * e2 + 1] += dr * (X - vx[e]);
                }
            }
        }
    }
    // block reduction of 18 values: loss inside the box, 16 gradients, mask ones inside the box
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float vals[18];
    vals[0] = acc;
    for (int i = 0; i < 16; ++i) vals[1 + i] = g[i];
    vals[17] = ones_in;
    for (int i = 0; i < 18; ++i) {
        float v = vals[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 17) {
        float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x == 0) {
            // pixels outside the box: polygon value 0 -> focal(x, 0) = 0.75 * ce(x) * sigmoid(x)^2
            const float in_ones = red[0][17] + red[1][17] + red[2][17] + red[3][17];
            const float out_ones = (float)mask_ones[mi] - in_ones;
            const float out_zeros = (float)(total - bw * bh) - out_ones;
            const float L0 = 0.75f * 0.69314718056f * 0.25f;
            const float p1 = 1.f / (1.f + __expf(-1.f));
            const float L1 = 0.75f * (1.f + log1pf(__expf(-1.f))) * p1 * p1;
            v += (L0 * out_zeros + L1 * out_ones) * inv_total;
            loss[r] = v;
        } else if (want_grad) {
            grad[r * 16 + threadIdx.x - 1] = v;
        }
    }
}

extern "C" int cr_polygon_focal(cr_ctx* ctx, const float* hull, const int32_t* count, const unsigned char* masks,
                                const int32_t* mask_idx, const int32_t* mask_ones, int n, int H, int W, float* loss,
                                float* grad) {
    CR_CHECK_ARG(ctx && n >= 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 30), "cr_polygon_focal: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(hull && count && masks && mask_idx && mask_ones && loss, "cr_polygon_focal: NULL pointer");
    hipLaunchKernelGGL(k_polygon_focal, dim3((unsigned)n), dim3(256), 0, ctx->stream, hull, count, masks, mask_idx, mask_ones, H,
                       W, loss, grad, grad != nullptr);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Mask scores of the 1000 proposals of one object (MABO diagnostics): score_segmentation / score_mod_segmentation,
// ProposalNetwork/scoring/scorefunction.py:88-126 with mask_iou / mod_mask_iou, ProposalNetwork/utils/utils.py:230-250.
// The reference rasterises every proposal on the host: cv2.convexHull(8 projected corners) -> int32 -> cv2.fillPoly on a
// full-size canvas -> [::4, ::4] -> intersection / union with the object mask [cv2: third-party, absent; restated].
// Here one workgroup per proposal: gift-wrapped hull of the 8 points (float), vertices truncated to int32 like the numpy
// cast, and a sample (4i, 4j) belongs to the filled polygon iff it lies inside or ON the closed integer polygon (exact
// integer cross products); only the samples in the polygon's bounding box are visited.  cv2.fillPoly additionally paints
// the Bresenham lines of the edges, which can add samples up to half a pixel outside the exact polygon -- at a stride of 4
// these are rare; parity with cv2 is unpinned.
// counts (P,2) int32 = {samples in polygon, samples in polygon AND mask}; the IoU arithmetic is done by the caller.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_segment_counts(const float* __restrict__ corners, int P, const unsigned char* __restrict__ mask,
                                                        int H, int W, int stride, int* __restrict__ counts) {
    __shared__ int hx[8], hy[8], hk, bb[4];
    __shared__ int s_cnt[2];
    const int p = blockIdx.x;
    if (threadIdx.x == 0) {
        float x[8], y[8];
        bool finite = true;
        for (int i = 0; i < 8; ++i) {
            x[i] = corners[(p * 8 + i) * 2]; y[i] = corners[(p * 8 + i) * 2 + 1];
            finite &= (x[i] == x[i]) && (y[i] == y[i]) && fabsf(x[i]) < 1e9f && fabsf(y[i]) < 1e9f;
        }
        int m = 0;
        if (finite) {
            int start = 0;
            for (int i = 1; i < 8; ++i) if (x[i] < x[start] || (x[i] == x[start] && y[i] < y[start])) start = i;
            int l = start;
            for (int step = 0; step < 8; ++step) {
                hx[m] = (int)x[l]; hy[m] = (int)y[l]; ++m;                 // (int): truncation toward zero, like astype(int32)
                int q = -1;
                for (int i = 0; i < 8; ++i) {
                    if (x[i] == x[l] && y[i] == y[l]) continue;            // the point itself and its duplicates
                    if (q < 0) { q = i; continue; }
                    const float d = (x[i] - x[l]) * (y[q] - y[l]) - (y[i] - y[l]) * (x[q] - x[l]);
                    const float di = (x[i] - x[l]) * (x[i] - x[l]) + (y[i] - y[l]) * (y[i] - y[l]);
                    const float dq = (x[q] - x[l]) * (x[q] - x[l]) + (y[q] - y[l]) * (y[q] - y[l]);
                    if (d > 0.f || (d == 0.f && di > dq)) q = i;
                }
                if (q < 0 || (x[q] == x[start] && y[q] == y[start])) break;
                l = q;
            }
        }
        hk = m;
        int x0 = 1 << 30, x1 = -(1 << 30), y0 = 1 << 30, y1 = -(1 << 30);
        for (int i = 0; i < m; ++i) { x0 = min(x0, hx[i]); x1 = max(x1, hx[i]); y0 = min(y0, hy[i]); y1 = max(y1, hy[i]); }
        bb[0] = max(x0, 0); bb[1] = min(x1, W - 1); bb[2] = max(y0, 0); bb[3] = min(y1, H - 1);
        s_cnt[0] = 0; s_cnt[1] = 0;
    }
    __syncthreads();
    const int k = hk;
    int in_poly = 0, in_both = 0;
    if (k >= 1 && bb[0] <= bb[1] && bb[2] <= bb[3]) {
        const int i0 = (bb[0] + stride - 1) / stride, i1 = bb[1] / stride, j0 = (bb[2] + stride - 1) / stride, j1 = bb[3] / stride;
        const int nx = i1 - i0 + 1, ny = j1 - j0 + 1;
        if (nx > 0 && ny > 0) {
            for (int t = threadIdx.x; t < nx * ny; t += 256) {
                const int X = (i0 + t % nx) * stride, Y = (j0 + t / nx) * stride;
                bool pos = true, neg = true;
                if (k == 1) {
                    pos = neg = (X == hx[0] && Y == hy[0]);
                } else {
                    for (int e = 0; e < k; ++e) {
                        const int e2 = e + 1 == k ? 0 : e + 1;
                        const long long cr = (long long)(hx[e2] - hx[e]) * (Y - hy[e]) - (long long)(hy[e2] - hy[e]) * (X - hx[e]);
                        pos &= cr >= 0; neg &= cr <= 0;
                    }
                    if (k == 2 && pos && neg) {                        // on the line: inside the segment's box only
                        pos = neg = X >= min(hx[0], hx[1]) && X <= max(hx[0], hx[1]) && Y >= min(hy[0], hy[1]) && Y <= max(hy[0], hy[1]);
                    }
                }
                if (pos || neg) {
                    ++in_poly;
                    in_both += mask[(size_t)Y * W + X] ? 1 : 0;
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) { in_poly += __shfl_xor(in_poly, off, 64); in_both += __shfl_xor(in_both, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt[0], in_poly); atomicAdd(&s_cnt[1], in_both); }
    __syncthreads();
    if (threadIdx.x == 0) { counts[2 * p] = s_cnt[0]; counts[2 * p + 1] = s_cnt[1]; }
}

extern "C" int cr_segment_counts(cr_ctx* ctx, const float* corners, int P, const unsigned char* mask, int H, int W, int stride,
                                 int32_t* counts) {
    CR_CHECK_ARG(ctx && P >= 0 && H > 0 && W > 0 && stride > 0, "cr_segment_counts: bad args");
    if (P == 0) return CR_OK;
    CR_CHECK_ARG(corners && mask && counts, "cr_segment_counts: NULL pointer");
    hipLaunchKernelGGL(k_segment_counts, dim3((unsigned)P), dim3(256), 0, ctx->stream, corners, P, mask, H, W, stride, counts);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Object mask -> minimum-area rectangle of its largest 8-connected component: the host step of score_corners
// (ProposalNetwork/scoring/scorefunction.py:58-68: cv2.findContours(RETR_EXTERNAL) -> max contourArea -> cv2.minAreaRect
// -> cv2.boxPoints), kept on the device so the n masks never leave HBM.
//   1. k_ccl_init    label = first pixel of the pixel's run inside its 64-pixel wave segment (ballot), -1 for background
//   2. k_ccl_merge   union-find (atomicMin on the parent) across segment boundaries and with the row above; only the
//                    pixels at run starts of either row issue a union, the rest are implied
//   3. k_ccl_sizes   label := root; each run segment adds its length to sizes[root] (one atomic per segment)
//   4. k_ccl_best    per object max over roots of (size, lowest root) -- ties go to the component that starts first in
//                    raster order, as scipy.ndimage.label + argmax does in the host restatement (oracle/rect.py)
//   5. k_mask_rect   one workgroup per object: row extremes of the chosen component -> left / right convex chains (two
//                    lanes, integer turn tests) -> rotating calipers over the hull edges in f64
// Labels are pixel indices local to the object (H*W < 2^31).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ccl_find(const int* L, int a) {
    int p;
    while ((p = __hip_atomic_load(L + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != a) a = p;   // parents change under us: read at L2
    return a;
}

__device__ __forceinline__ void ccl_union(int* L, int a, int b) {
    bool done = false;
    while (!done) {
        a = ccl_find(L, a);
        b = ccl_find(L, b);
        if (a < b) { const int old = atomicMin(L + b, a); done = old == b; b = old; }
        else if (b < a) { const int old = atomicMin(L + a, b); done = old == a; a = old; }
        else done = true;
    }
}

// Window of an object = rows [y0, y1] x 256-pixel chunks from x0 covering [x0, x1] of its mask's bounding box (k_mask_bbox).
// The labelling passes walk only the window, so they cost the masks' area, not the frame's; pixels left of x0 / right of x1
// / above y0 are background by construction and are never read.
struct MaskSrc {
    const unsigned char* base;              // (n,H,W) dense, or
    const unsigned char* const* ptrs;       // n device pointers to (H,W) masks
    __device__ __forceinline__ const unsigned char* of(int obj, size_t hw) const { return ptrs ? ptrs[obj] : base + obj * hw; }
};

// bbox (n,4) int32 = {x0, y0, x1, y1}, preset to {large, large, -1, -1}.  grid (ceil(H/32), n), block 256: a wave scans 8 rows.
__global__ __launch_bounds__(256) void k_mask_bbox(MaskSrc src, int H, int W, int* __restrict__ bbox) {
    const int lane = threadIdx.x & 63, obj = blockIdx.y;
    const int yb = blockIdx.x * 32 + (threadIdx.x >> 6) * 8;
    const unsigned char* m = src.of(obj, (size_t)H * W);
    int lo = W, hi = -1, ylo = H, yhi = -1;
    const bool words = (W & 3) == 0 && (((size_t)m) & 3) == 0;
#pragma unroll 8
    for (int k = 0; k < 8; ++k) {
        const int y = yb + k;
        if (y >= H) break;
        const unsigned char* row = m + (size_t)y * W;
        bool any = false;
        if (words) {
            const unsigned int* r4 = (const unsigned int*)row;
            for (int i = lane; i < W / 4; i += 64) {
                const unsigned int v = r4[i];
                if (v) {
                    const int first = (v & 0xFFu) ? 0 : (v & 0xFF00u) ? 1 : (v & 0xFF0000u) ? 2 : 3;
                    const int last = (v & 0xFF000000u) ? 3 : (v & 0xFF0000u) ? 2 : (v & 0xFF00u) ? 1 : 0;
                    lo = min(lo, 4 * i + first); hi = max(hi, 4 * i + last); any = true;
                }
            }
        } else {
            for (int x = lane; x < W; x += 64)
                if (row[x]) { lo = min(lo, x); hi = max(hi, x); any = true; }
        }
        if (any) { ylo = min(ylo, y); yhi = max(yhi, y); }
    }
    for (int o = 32; o; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o));
        ylo = min(ylo, __shfl_xor(ylo, o)); yhi = max(yhi, __shfl_xor(yhi, o));
    }
    if (lane == 0 && hi >= 0) {
        int* b = bbox + 4 * obj;
        atomicMin(b + 0, lo); atomicMin(b + 1, ylo); atomicMax(b + 2, hi); atomicMax(b + 3, yhi);
    }
}

// The labelling passes: grid (CCL_ROWS, n), block 256.  Workgroup j of an object walks the window rows y0+j, y0+j+CCL_ROWS,
// ... in chunks of 256 pixels from x0, so a wave always covers the same 64 consecutive pixels of a row in every pass (the
// run segments of k_ccl_init) and no workgroup is spent outside the window.
#define CCL_ROWS 16
#define CCL_WINDOW_LOOP(...)                                                                                    \
    const int obj = blockIdx.y;                                                                                  \
    const int bx0 = bbox[4 * obj], by0 = bbox[4 * obj + 1], bx1 = bbox[4 * obj + 2], by1 = bbox[4 * obj + 3];   \
    for (int y = by0 + (int)blockIdx.x; y <= by1; y += CCL_ROWS)                                                 \
        for (int xc = bx0; xc <= bx1; xc += 256) {                                                               \
            const int x = xc + (int)threadIdx.x;                                                                 \
            __VA_ARGS__                                                                                          \
        }

__global__ __launch_bounds__(256) void k_ccl_init(MaskSrc src, int H, int W, const int* __restrict__ bbox,
                                                  int* __restrict__ labels, int* __restrict__ sizes) {
    const size_t hw = (size_t)H * W;
    const int lane = threadIdx.x & 63;
    CCL_WINDOW_LOOP({
        const size_t base = obj * hw;
        const bool fg = x < W && src.of(obj, hw)[(size_t)y * W + x] != 0;
        const unsigned long long b = __ballot(fg);
        if (x < W) {
            const unsigned long long below = ~b & ((1ull << lane) - 1ull);  // background lanes before this one
            const int start = below ? 64 - __clzll(below) : 0;
            labels[base + (size_t)y * W + x] = fg ? y * W + (x - lane + start) : -1;
            sizes[base + (size_t)y * W + x] = 0;
        }
    })
}

__global__ __launch_bounds__(256) void k_ccl_merge(int H, int W, const int* __restrict__ bbox, int* __restrict__ labels) {
    CCL_WINDOW_LOOP({
        int* L = labels + (size_t)obj * H * W;
        const int p = y * W + x;
        if (x < W && L[p] >= 0) {
            const bool w = x > bx0 && L[p - 1] >= 0;
            if (w && (threadIdx.x & 63) == 0) ccl_union(L, p, p - 1);      // run continues across the segment boundary
            if (y > by0) {
                const int up = p - W;
                const bool n = L[up] >= 0, nw = x > bx0 && L[up - 1] >= 0, ne = x < bx1 && L[up + 1] >= 0;
                if (n) { if (!w || !nw) ccl_union(L, p, up); }
                else {
                    if (nw && !w) ccl_union(L, p, up - 1);
                    if (ne) ccl_union(L, p, up + 1);
                }
            }
        }
    })
}

__global__ __launch_bounds__(256) void k_ccl_sizes(int H, int W, const int* __restrict__ bbox, int* __restrict__ labels,
                                                   int* __restrict__ sizes) {
    const int lane = threadIdx.x & 63;
    CCL_WINDOW_LOOP({
        const size_t base = (size_t)obj * H * W;
        int* L = labels + base;
        const int p = y * W + x;
        const bool fg = x < W && L[p] >= 0;
        const unsigned long long b = __ballot(fg);
        if (fg) {
            const int root = ccl_find(L, p);
            L[p] = root;
            if (lane == 0 || !((b >> (lane - 1)) & 1ull)) {                 // first pixel of a run segment
                const unsigned long long rest = ~(b >> lane);
                atomicAdd(sizes + base + root, rest ? __ffsll((long long)rest) - 1 : 64 - lane);
            }
        }
    })
}

__global__ __launch_bounds__(256) void k_ccl_best(int H, int W, const int* __restrict__ bbox, const int* __restrict__ labels,
                                                  const int* __restrict__ sizes, unsigned long long* __restrict__ best) {
    CCL_WINDOW_LOOP({
        const size_t base = (size_t)obj * H * W;
        const int p = y * W + x;
        if (x < W && labels[base + p] == p)
            atomicMax(best + obj, ((unsigned long long)(unsigned)sizes[base + p] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p));
    })
}

#define RECT_T 256
// row extremes of one component (sx0[r] = -1: the component has no pixel in row r) -> left / right convex chains ->
// rotating calipers over the hull edges in f64 -> the four corners (cv2.boxPoints order).  Shared by the two mask kernels.
__device__ __forceinline__ void rect_from_row_extremes(const int H, const int obj, int* sx0, int* sx1, int* stk, int* hx, int* hy,
                                                       int* s_rows, int* s_cnt, double* s_area, double* s_ang,
                                                       float* out, unsigned char* valid) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rmin = s_rows[0], rmax = s_rows[1];
    if (tid < 2) {                                                          // lane 0: left chain, lane 1: right chain
        int* st = stk + tid * H;
        const int* sx = tid ? sx1 : sx0;
        int top = 0;
        for (int r = rmin; r <= rmax; ++r) {
            const int x = sx[r];
            if (x < 0) continue;                                            // (a component covers contiguous rows)
            while (top >= 2) {
                const int ay = st[top - 2], by = st[top - 1], ax = sx[ay], bx = sx[by];
                const int cross = (bx - ax) * (r - ay) - (by - ay) * (x - ax);
                if (tid ? cross <= 0 : cross >= 0) --top; else break;
            }
            st[top++] = r;
        }
        s_cnt[tid] = top;
    }
    __syncthreads();
    if (tid == 0) {
        int h = 0;
        for (int i = 0; i < s_cnt[0]; ++i) { const int r = stk[i]; hx[h] = sx0[r]; hy[h] = r; ++h; }
        for (int i = s_cnt[1] - 1; i >= 0; --i) {
            const int r = stk[H + i];
            if ((r == rmax || r == rmin) && sx0[r] == sx1[r]) continue;     // chain end shared with the left chain
            hx[h] = sx1[r]; hy[h] = r; ++h;
        }
        s_cnt[2] = h;
    }
    __syncthreads();
    const int h = s_cnt[2];
    if (h == 1) {
        if (tid < 8) out[tid] = (float)((tid & 1) ? hy[0] : hx[0]);
        if (tid == 0) valid[obj] = 1;
        return;
    }
    const double HALF_PI = 1.5707963267948966;
    double b_area = 1.0e300, b_ang = 4.0;
    for (int e = tid; e < h; e += RECT_T) {
        const int e2 = e + 1 == h ? 0 : e + 1;
        double ang = fmod(atan2((double)(hy[e2] - hy[e]), (double)(hx[e2] - hx[e])), HALF_PI);
        if (ang < 0.0) ang += HALF_PI;
        const double c = cos(ang), s = sin(ang);
        double x0 = 1.0e300, x1 = -1.0e300, y0 = 1.0e300, y1 = -1.0e300;
        for (int v = 0; v < h; ++v) {
            const double vx = (double)hx[v], vy = (double)hy[v];
            const double rx = vx * c + vy * s, ry = vy * c - vx * s;
            x0 = fmin(x0, rx); x1 = fmax(x1, rx); y0 = fmin(y0, ry); y1 = fmax(y1, ry);
        }
        const double area = (x1 - x0) * (y1 - y0);
        if (area < b_area || (area == b_area && ang < b_ang)) { b_area = area; b_ang = ang; }
    }
    for (int o = 32; o; o >>= 1) {
        const double a2 = __shfl_xor(b_area, o), g2 = __shfl_xor(b_ang, o);
        if (a2 < b_area || (a2 == b_area && g2 < b_ang)) { b_area = a2; b_ang = g2; }
    }
    if (lane == 0) { s_area[wave] = b_area; s_ang[wave] = b_ang; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < RECT_T / 64; ++w)
            if (s_area[w] < b_area || (s_area[w] == b_area && s_ang[w] < b_ang)) { b_area = s_area[w]; b_ang = s_ang[w]; }
        const double c = cos(b_ang), s = sin(b_ang);
        double x0 = 1.0e300, x1 = -1.0e300, y0 = 1.0e300, y1 = -1.0e300;
        for (int v = 0; v < h; ++v) {
            const double vx = (double)hx[v], vy = (double)hy[v];
            const double rx = vx * c + vy * s, ry = vy * c - vx * s;
            x0 = fmin(x0, rx); x1 = fmax(x1, rx); y0 = fmin(y0, ry); y1 = fmax(y1, ry);
        }
        const double bx[4] = {x0, x1, x1, x0}, by[4] = {y0, y0, y1, y1};
        for (int k = 0; k < 4; ++k) {                                       // back to image axes: [bx by] @ [[c s] [-s c]]
            out[2 * k] = (float)(bx[k] * c - by[k] * s);
            out[2 * k + 1] = (float)(bx[k] * s + by[k] * c);
        }
        valid[obj] = 1;
    }
}

// dynamic LDS: int sx0[H], sx1[H], stack[2][H], hull_x[2H], hull_y[2H]
__global__ __launch_bounds__(RECT_T) void k_mask_rect(int H, int W, const int* __restrict__ bbox, const int* __restrict__ labels,
                                                      const unsigned long long* __restrict__ best, float* __restrict__ rects,
                                                      unsigned char* __restrict__ valid) {
    extern __shared__ int s_rect[];
    int* sx0 = s_rect;
    int* sx1 = sx0 + H;
    int* stk = sx1 + H;
    int* hx = stk + 2 * H;
    int* hy = hx + 2 * H;
    __shared__ int s_rows[2], s_cnt[3];
    __shared__ double s_area[RECT_T / 64], s_ang[RECT_T / 64];
    const int obj = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* out = rects + (size_t)obj * 8;
    if (bbox[4 * obj + 2] == -2) return;                                    // k_mask_rect_runs has written this object
    const unsigned long long key = best[obj];
    if (key == 0ull) {                                                      // empty mask
        if (tid < 8) out[tid] = __builtin_nanf("");                         // cr_cubes_project_score: fallback rectangle
        if (tid == 0) valid[obj] = 0;
        return;
    }
    const int root = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    const int* L = labels + (size_t)obj * H * W;
    if (tid == 0) { s_rows[0] = H; s_rows[1] = -1; }
    __syncthreads();
    const int bx0 = bbox[4 * obj], by0 = bbox[4 * obj + 1], bx1 = bbox[4 * obj + 2], by1 = bbox[4 * obj + 3];
    for (int r = by0 + wave; r <= by1; r += RECT_T / 64) {
        int lo = W, hi = -1;
        for (int x = bx0 + lane; x <= bx1; x += 64)
            if (L[(size_t)r * W + x] == root) { lo = min(lo, x); hi = max(hi, x); }
        for (int o = 32; o; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
        if (lane == 0) {
            sx0[r] = hi >= 0 ? lo : -1;
            sx1[r] = hi;
            if (hi >= 0) { atomicMin(&s_rows[0], r); atomicMax(&s_rows[1], r); }
        }
    }
    __syncthreads();
    rect_from_row_extremes(H, obj, sx0, sx1, stk, hx, hy, s_rows, s_cnt, s_area, s_ang, out, valid);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same result from RUNS instead of pixel labels, one workgroup per object, everything in LDS: a row of an object mask
// is a handful of runs, so the component labelling is a union-find over (rows x <= RUN_MAXR) run records with LDS atomics
// instead of five passes over two int32 planes in global memory with L2 atomics (0.9 ms for 1 024 objects: latency of the
// parent chains).  Objects with a row of more than RUN_MAXR runs (noise, combs) are left to the pixel-label kernels above:
// this kernel marks the objects it has finished with bbox x1 = y1 = -2, which makes their window loops empty.
//   1. runs of every window row (a wave per row: ballot over 64-pixel segments, run boundaries by bit scans)
//   2. union of the runs of adjacent rows that touch 8-connectedly: [s - 1, e + 1] overlap (atomicMin on the parent)
//   3. size[root] += run length;  4. best root = max (size, earliest first pixel);  5. row extremes of that component
//   -> rect_from_row_extremes
// ---------------------------------------------------------------------------------------------------------------------
#define RUN_MAXR 4
__device__ __forceinline__ int run_find(volatile int* P, int a) {
    int p;
    while ((p = P[a]) != a) a = p;
    return a;
}
__device__ __forceinline__ void run_union(int* P, int a, int b) {
    bool done = false;
    while (!done) {
        a = run_find(P, a);
        b = run_find(P, b);
        if (a < b) { const int old = atomicMin(P + b, a); done = old == b; b = old; }
        else if (b < a) { const int old = atomicMin(P + a, b); done = old == a; a = old; }
        else done = true;
    }
}

// dynamic LDS, H rows: short rs[H][4], re[H][4]; int parent[4H], size[4H]; int rc[H] (the 8H ints of the hull stage alias them)
__global__ __launch_bounds__(RECT_T) void k_mask_rect_runs(MaskSrc src, int H, int W, int* __restrict__ bbox,
                                                           float* __restrict__ rects, unsigned char* __restrict__ valid) {
    extern __shared__ int s_dyn[];
    short* rs = reinterpret_cast<short*>(s_dyn);
    short* re = rs + RUN_MAXR * H;
    int* parent = s_dyn + RUN_MAXR * H;                 // (2 x 4H shorts = 4H ints)
    int* size = parent + RUN_MAXR * H;
    int* rc = size + RUN_MAXR * H;
    // the arrays of the hull stage live where records that are dead by then were: the row extremes on `size` (written after
    // the best component is known), the chain stacks and hull vertices on the runs and parents (read last by step 5, a
    // barrier before rect_from_row_extremes) -- 13 ints of LDS per row instead of 21: six workgroups per CU at H = 512
    int* sx0 = size;
    int* sx1 = size + H;
    int* stk = s_dyn;
    int* hx = s_dyn + 2 * H;
    int* hy = parent;
    __shared__ int s_rows[2], s_cnt[3], s_over;
    __shared__ unsigned long long s_best;
    __shared__ double s_area[RECT_T / 64], s_ang[RECT_T / 64];
    const int obj = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* out = rects + (size_t)obj * 8;
    const int bx0 = bbox[4 * obj], by0 = bbox[4 * obj + 1], bx1 = bbox[4 * obj + 2], by1 = bbox[4 * obj + 3];
    if (bx1 < 0) {                                                           // empty mask
        if (tid < 8) out[tid] = __builtin_nanf("");                         // cr_cubes_project_score: fallback rectangle
        if (tid == 0) { valid[obj] = 0; bbox[4 * obj + 2] = -2; bbox[4 * obj + 3] = -2; }
        return;
    }
    if (tid == 0) { s_over = 0; s_best = 0ull; s_rows[0] = H; s_rows[1] = -1; }
    __syncthreads();
    const unsigned char* m = src.of(obj, (size_t)H * W);
    const int rows = by1 - by0 + 1;
    // ---- 1. runs (every quantity below is wave-uniform: the bit scans run on the scalar unit)
    for (int r = wave; r < rows; r += RECT_T / 64) {
        const unsigned char* row = m + (size_t)(by0 + r) * W;
        int cnt = 0, cur = -1;                                               // cur: start of a run still open at the segment end
        for (int x0 = bx0; x0 <= bx1; x0 += 64) {
            const int x = x0 + lane;
            unsigned long long b = __ballot(x <= bx1 && row[x] != 0);
            if (cur >= 0 && !(b & 1ull)) {                                   // the open run ended with the previous segment
                if (cnt < RUN_MAXR && lane == 0) { rs[r * RUN_MAXR + cnt] = (short)cur; re[r * RUN_MAXR + cnt] = (short)(x0 - 1); }
                ++cnt; cur = -1;
            }
            while (b) {
                const int st = __ffsll((long long)b) - 1;
                const unsigned long long rest = ~(b >> st);                  // zero bits above the run (and above bit 63 - st)
                const int len = rest ? __ffsll((long long)rest) - 1 : 64 - st;
                const int a = cur >= 0 ? cur : x0 + st;                      // (cur >= 0 implies st == 0 here)
                if (st + len == 64) { cur = a; break; }                      // reaches the segment end: maybe continues
                if (cnt < RUN_MAXR && lane == 0) { rs[r * RUN_MAXR + cnt] = (short)a; re[r * RUN_MAXR + cnt] = (short)(x0 + st + len - 1); }
                ++cnt; cur = -1;
                b &= ~(((len == 64 ? 0ull : (1ull << len)) - 1ull) << st);
            }
        }
        if (cur >= 0) {
            if (cnt < RUN_MAXR && lane == 0) { rs[r * RUN_MAXR + cnt] = (short)cur; re[r * RUN_MAXR + cnt] = (short)bx1; }
            ++cnt;
        }
        if (lane == 0) {
            rc[r] = min(cnt, RUN_MAXR);
            if (cnt > RUN_MAXR) s_over = 1;
        }
    }
    __syncthreads();
    if (s_over) return;                                                      // left to the pixel-label kernels
    for (int i = tid; i < rows * RUN_MAXR; i += RECT_T) { parent[i] = i; size[i] = 0; }
    __syncthreads();
    // ---- 2. unions between adjacent rows
    for (int r = 1 + tid; r < rows; r += RECT_T)
        for (int k = 0; k < rc[r]; ++k) {
            const int s0 = rs[r * RUN_MAXR + k] - 1, e0 = re[r * RUN_MAXR + k] + 1;
            for (int j = 0; j < rc[r - 1]; ++j)
                if (rs[(r - 1) * RUN_MAXR + j] <= e0 && re[(r - 1) * RUN_MAXR + j] >= s0) run_union(parent, r * RUN_MAXR + k, (r - 1) * RUN_MAXR + j);
        }
    __syncthreads();
    // ---- 3. sizes, 4. the largest component (ties: the one whose first pixel comes first in raster order)
    for (int r = tid; r < rows; r += RECT_T)
        for (int k = 0; k < rc[r]; ++k) {
            const int root = run_find(parent, r * RUN_MAXR + k);
            parent[r * RUN_MAXR + k] = root;                                 // (a root's entry stays itself)
            atomicAdd(size + root, re[r * RUN_MAXR + k] - rs[r * RUN_MAXR + k] + 1);
        }
    __syncthreads();
    for (int r = tid; r < rows; r += RECT_T)
        for (int k = 0; k < rc[r]; ++k) {
            const int i = r * RUN_MAXR + k;
            if (parent[i] == i)
                atomicMax(&s_best, ((unsigned long long)(unsigned)size[i] << 32) |
                                   (unsigned long long)(0xFFFFFFFFu - (unsigned)((by0 + r) * W + rs[i])));
        }
    __syncthreads();
    const unsigned long long key = s_best;
    const int p0 = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));       // first pixel of the chosen component
    const int root = (p0 / W - by0) * RUN_MAXR;                              // its first run is run 0..3 of that row: find which
    int rootidx = -1;
    for (int k = 0; k < RUN_MAXR; ++k)
        if (k < rc[p0 / W - by0] && rs[root + k] == p0 % W) rootidx = root + k;
    // ---- 5. row extremes of the chosen component (absolute rows, like k_mask_rect)
    for (int r = tid; r < rows; r += RECT_T) {
        int lo = W, hi = -1;
        for (int k = 0; k < rc[r]; ++k)
            if (parent[r * RUN_MAXR + k] == rootidx) { lo = min(lo, (int)rs[r * RUN_MAXR + k]); hi = max(hi, (int)re[r * RUN_MAXR + k]); }
        sx0[by0 + r] = hi >= 0 ? lo : -1;
        sx1[by0 + r] = hi;
        if (hi >= 0) { atomicMin(&s_rows[0], by0 + r); atomicMax(&s_rows[1], by0 + r); }
    }
    if (tid == 0) { bbox[4 * obj + 2] = -2; bbox[4 * obj + 3] = -2; }        // finished: nothing left for the pixel-label kernels
    __syncthreads();
    rect_from_row_extremes(H, obj, sx0, sx1, stk, hx, hy, s_rows, s_cnt, s_area, s_ang, out, valid);
}

extern "C" int cr_mask_rects(cr_ctx* ctx, const unsigned char* masks, const unsigned char* const* mask_ptrs, int n, int H, int W,
                             int32_t* labels, int32_t* sizes, unsigned long long* best, int32_t* bbox, float* rects,
                             unsigned char* valid) {
    CR_CHECK_ARG(ctx && n >= 0 && H > 0 && W > 0 && H <= 1900 && (int64_t)H * W < (1ll << 31) && n <= 65535,
                 "cr_mask_rects: bad args (H <= 1900: 32 B of LDS per row; n <= 65535)");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG((masks != nullptr) != (mask_ptrs != nullptr), "cr_mask_rects: give either masks or mask_ptrs");
    CR_CHECK_ARG(labels && sizes && best && bbox && rects && valid, "cr_mask_rects: NULL pointer");
    CR_HIP(hipMemsetAsync(best, 0, sizeof(unsigned long long) * (size_t)n, ctx->stream));
    CR_HIP(hipMemset2DAsync(bbox, 4 * sizeof(int32_t), 0x7f, 2 * sizeof(int32_t), (size_t)n, ctx->stream));       // x0, y0 = large
    CR_HIP(hipMemset2DAsync(bbox + 2, 4 * sizeof(int32_t), 0xff, 2 * sizeof(int32_t), (size_t)n, ctx->stream));   // x1, y1 = -1
    const MaskSrc src{masks, mask_ptrs};
    const dim3 grid(CCL_ROWS, (unsigned)n), block(256);
    hipLaunchKernelGGL(k_mask_bbox, dim3((unsigned)cr_cdiv(H, 32), (unsigned)n), block, 0, ctx->stream, src, H, W, bbox);
    // the run-based kernel first (LDS only; 13 ints of LDS per row); what it leaves goes through the pixel-label passes
    const size_t run_lds = sizeof(int) * (size_t)H * (3 * RUN_MAXR + 1);
    static const bool runs_on = []() { const char* e = getenv("CR_MASK_RUNS"); return !(e && e[0] == '0'); }();
    if (runs_on && run_lds <= 64 * 1024 && W < 32768)
        hipLaunchKernelGGL(k_mask_rect_runs, dim3((unsigned)n), dim3(RECT_T), run_lds, ctx->stream, src, H, W, bbox, rects, valid);
    hipLaunchKernelGGL(k_ccl_init, grid, block, 0, ctx->stream, src, H, W, (const int*)bbox, labels, sizes);
    hipLaunchKernelGGL(k_ccl_merge, grid, block, 0, ctx->stream, H, W, (const int*)bbox, labels);
    hipLaunchKernelGGL(k_ccl_sizes, grid, block, 0, ctx->stream, H, W, (const int*)bbox, labels, sizes);
    hipLaunchKernelGGL(k_ccl_best, grid, block, 0, ctx->stream, H, W, (const int*)bbox, (const int*)labels, (const int*)sizes, best);
    hipLaunchKernelGGL(k_mask_rect, dim3((unsigned)n), dim3(RECT_T), sizeof(int) * 8 * (size_t)H, ctx->stream, H, W,
                       (const int*)bbox, (const int*)labels, (const unsigned long long*)best, rects, valid);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
