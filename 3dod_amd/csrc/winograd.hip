// Winograd F(2x2, 3x3) transforms for the stride-1, pad-1, 3x3 convolutions of the pyramid heads (float32, NHWC):
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        2.25x fewer multiplies than the direct form
// The 16 element-wise products over channels are 16 GEMMs (tiles x C) @ (C x O), run by the existing 1x1 grouped
// convolution launches (cr_conv2d_fwd_group); this file holds the three transforms around them:
//   cr_wino_filter   U[16][O][C] = G g G^T of every (output, input) channel pair -- forward (g = w[o][.][.][c]) or
//                    backward-data (U[16][C][O] from the flipped taps: dX = conv(dY, rot180(w)^T))
//   cr_wino_input    V[16][T][C] = B^T d B of every 4 x 4 input window (2 x 2 output tile), zero padding at the borders;
//                    T = tiles of all maps of the call (pyramid levels x images), one row block per map
//   cr_wino_output   y = A^T M A (+ bias, ReLU, + accumulate) from M[16][T][O], written to each map's own tensor
// float32 error of F(2x2, 3x3) against float64: 6e-7 of the output range (direct: 2e-7; scripts/winograd_numerics.py).
// Reference call sites: the 3x3 convolutions of detectron2's FPN output / RPN head as used by cubercnn/modeling (the
// arithmetic they replace is torch.nn.functional.conv2d).
#include "cr_common.h"

#define WINO_MAX_MAPS 8
struct WinoMap { const float* src; float* dst; const float* acc; int N, H, W, tbase, blk0; };
struct WinoGeo { WinoMap m[WINO_MAX_MAPS]; int n, C, T; };

__device__ __forceinline__ int wino_find(const WinoGeo& g, int blk) {
    int i = 0;
    while (i + 1 < g.n && blk >= g.m[i + 1].blk0) ++i;
    return i;
}

// grid: per map N * (H/2) * (C/64) blocks; block 256 = 4 waves, wave w takes tile columns w, w+4, ...; lane = channel
__global__ __launch_bounds__(256) void k_wino_input(WinoGeo g, float* __restrict__ V) {
    const int mi = wino_find(g, (int)blockIdx.x);
    const WinoMap& m = g.m[mi];
    const int C = g.C, cgs = C >> 6;
    const int local = (int)blockIdx.x - m.blk0;
    const int cg = local % cgs, row = local / cgs;
    const int th = m.H >> 1, tw = m.W >> 1;
    const int n = row / th, ty = row % th;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = cg * 64 + lane;
    const size_t plane = (size_t)g.T * C;
    for (int tx = wave; tx < tw; tx += 4) {
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 2 * ty - 1 + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = 2 * tx - 1 + j;
                d[i][j] = ((unsigned)y < (unsigned)m.H && (unsigned)x < (unsigned)m.W)
                              ? m.src[(((size_t)n * m.H + y) * m.W + x) * C + c] : 0.f;
            }
        }
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
        }
        const size_t o = ((size_t)m.tbase + ((size_t)n * th + ty) * tw + tx) * C + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            V[(i * 4 + 0) * plane + o] = t[i][0] - t[i][2];
            V[(i * 4 + 1) * plane + o] = t[i][1] + t[i][2];
            V[(i * 4 + 2) * plane + o] = t[i][2] - t[i][1];
            V[(i * 4 + 3) * plane + o] = t[i][1] - t[i][3];
        }
    }
}

__global__ __launch_bounds__(256) void k_wino_output(WinoGeo g, const float* __restrict__ M, const float* __restrict__ bias, int relu) {
    const int mi = wino_find(g, (int)blockIdx.x);
    const WinoMap& m = g.m[mi];
    const int C = g.C, cgs = C >> 6;                    // C: OUTPUT channels of the product here
    const int local = (int)blockIdx.x - m.blk0;
    const int cg = local % cgs, row = local / cgs;
    const int th = m.H >> 1, tw = m.W >> 1;
    const int n = row / th, ty = row % th;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = cg * 64 + lane;
    const size_t plane = (size_t)g.T * C;
    const float b = bias ? bias[c] : 0.f;
    for (int tx = wave; tx < tw; tx += 4) {
        const size_t o = ((size_t)m.tbase + ((size_t)n * th + ty) * tw + tx) * C + c;
        float q[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) q[i][j] = M[(i * 4 + j) * plane + o];
        float s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[0][j] = (q[0][j] + q[1][j]) + q[2][j]; s[1][j] = (q[1][j] - q[2][j]) - q[3][j]; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float y0 = (s[i][0] + s[i][1]) + s[i][2] + b, y1 = (s[i][1] - s[i][2]) - s[i][3] + b;
            const size_t p = (((size_t)n * m.H + 2 * ty + i) * m.W + 2 * tx) * C + c;
            if (relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
            if (m.acc) { y0 += m.acc[p]; y1 += m.acc[p + C]; }
            m.dst[p] = y0;
            m.dst[p + C] = y1;
        }
    }
}

// w: [O][3][3][C] (KRSC).  mode 0: U[k][o][c]; mode 1 (backward-data): U[k][c][o] from the taps rotated by 180 degrees
__global__ __launch_bounds__(256) void k_wino_filter(const float* __restrict__ w, float* __restrict__ U, int O, int C, int mode) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= O * C) return;
    const int o = idx / C, c = idx - o * C;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = w[((size_t)(o * 3 + (mode ? 2 - r : r)) * 3 + (mode ? 2 - s : s)) * C + c];
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        t[0][s] = g[0][s];
        t[1][s] = 0.5f * ((g[0][s] + g[1][s]) + g[2][s]);
        t[2][s] = 0.5f * ((g[0][s] - g[1][s]) + g[2][s]);
        t[3][s] = g[2][s];
    }
    const size_t plane = (size_t)O * C;
    const size_t at = mode ? (size_t)c * O + o : (size_t)o * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        U[(i * 4 + 0) * plane + at] = t[i][0];
        U[(i * 4 + 1) * plane + at] = 0.5f * ((t[i][0] + t[i][1]) + t[i][2]);
        U[(i * 4 + 2) * plane + at] = 0.5f * ((t[i][0] - t[i][1]) + t[i][2]);
        U[(i * 4 + 3) * plane + at] = t[i][2];
    }
}

// the adjoint of the output transform for the weight gradient: dM = A dY A^T of every 2 x 2 tile of dY (16 planes (T, C)), and,
// when db is given, db[c] += sum of dY (f32 atomics: one per workgroup and channel)
__global__ __launch_bounds__(256) void k_wino_dy(WinoGeo g, float* __restrict__ dM, float* __restrict__ db) {
    __shared__ float s_sum[4][64];
    const int mi = wino_find(g, (int)blockIdx.x);
    const WinoMap& m = g.m[mi];
    const int C = g.C, cgs = C >> 6;
    const int local = (int)blockIdx.x - m.blk0;
    const int cg = local % cgs, row = local / cgs;
    const int th = m.H >> 1, tw = m.W >> 1;
    const int n = row / th, ty = row % th;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = cg * 64 + lane;
    const size_t plane = (size_t)g.T * C;
    float acc = 0.f;
    for (int tx = wave; tx < tw; tx += 4) {
        const size_t p = (((size_t)n * m.H + 2 * ty) * m.W + 2 * tx) * C + c;
        const float d00 = m.src[p], d01 = m.src[p + C], d10 = m.src[p + (size_t)m.W * C], d11 = m.src[p + (size_t)m.W * C + C];
        acc += (d00 + d01) + (d10 + d11);
        const float t[4][2] = {{d00, d01}, {d00 + d10, d01 + d11}, {d00 - d10, d01 - d11}, {-d10, -d11}};
        const size_t o = ((size_t)m.tbase + ((size_t)n * th + ty) * tw + tx) * C + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dM[(i * 4 + 0) * plane + o] = t[i][0];
            dM[(i * 4 + 1) * plane + o] = t[i][0] + t[i][1];
            dM[(i * 4 + 2) * plane + o] = t[i][0] - t[i][1];
            dM[(i * 4 + 3) * plane + o] = -t[i][1];
        }
    }
    if (db) {
        s_sum[wave][lane] = acc;
        __syncthreads();
        if (wave == 0) atomicAdd(db + c, (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]));
    }
}

// dW[o][r][s][c] += (G^T dU G)[r][s] from dU (16, O, C): the weight gradient back from the transformed domain, added into the
// flat-gradient view of the (O,3,3,C) weight (one thread per (o, c): no atomics)
__global__ __launch_bounds__(256) void k_wino_filter_grad(const float* __restrict__ dU, float* __restrict__ dw, int O, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= O * C) return;
    const int o = idx / C, c = idx - o * C;
    const size_t plane = (size_t)O * C;
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[i][j] = dU[(i * 4 + j) * plane + idx];
    float t[3][4];                                       // G^T u : G^T = [[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
        t[1][j] = 0.5f * (u[1][j] - u[2][j]);
        t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float g0 = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
        const float g1 = 0.5f * (t[r][1] - t[r][2]);
        const float g2 = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
        float* d = dw + ((size_t)(o * 3 + r) * 3) * C + c;
        d[0] += g0; d[C] += g1; d[2 * C] += g2;
    }
}

static int wino_geo(const char* who, WinoGeo& g, int n, const float* const* srcs, float* const* dsts, const float* const* accs,
                    const int* Ns, const int* Hs, const int* Ws, int C, int64_t T, unsigned* blocks) {
    CR_CHECK_ARG(n >= 1 && n <= WINO_MAX_MAPS, "%s: 1..%d maps", who, WINO_MAX_MAPS);
    CR_CHECK_ARG(C > 0 && C % 64 == 0, "%s: channels %% 64", who);
    g.n = n; g.C = C; g.T = (int)T;
    int tb = 0, blk = 0;
    for (int i = 0; i < n; ++i) {
        CR_CHECK_ARG(Ns[i] > 0 && Hs[i] > 0 && Ws[i] > 0 && Hs[i] % 2 == 0 && Ws[i] % 2 == 0, "%s: map %d needs even H and W", who, i);
        g.m[i].src = srcs ? srcs[i] : nullptr; g.m[i].dst = dsts ? dsts[i] : nullptr; g.m[i].acc = accs ? accs[i] : nullptr;
        g.m[i].N = Ns[i]; g.m[i].H = Hs[i]; g.m[i].W = Ws[i]; g.m[i].tbase = tb; g.m[i].blk0 = blk;
        tb += Ns[i] * (Hs[i] / 2) * (Ws[i] / 2);
        blk += Ns[i] * (Hs[i] / 2) * (C / 64);
    }
    CR_CHECK_ARG(tb == T && (int64_t)T * C * 16 < (1ll << 40), "%s: T = %lld does not match the maps (%d tiles)", who, (long long)T, tb);
    for (int i = n; i < WINO_MAX_MAPS; ++i) g.m[i] = g.m[n - 1];
    *blocks = (unsigned)blk;
    return CR_OK;
}

extern "C" int cr_wino_filter(cr_ctx* ctx, const float* w_krsc, float* U, int O, int C, int backward) {
    CR_CHECK_ARG(ctx && w_krsc && U && O > 0 && C > 0, "cr_wino_filter: bad args");
    hipLaunchKernelGGL(k_wino_filter, dim3((unsigned)cr_cdiv((int64_t)O * C, 256)), dim3(256), 0, ctx->stream, w_krsc, U, O, C, backward ? 1 : 0);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// xs: n HOST-array device pointers to (N_i, H_i, W_i, C) float32 maps; V (16, T, C) with T = sum N_i H_i W_i / 4
extern "C" int cr_wino_input(cr_ctx* ctx, int n, const float* const* xs, const int* Ns, const int* Hs, const int* Ws, int C,
                             float* V, int64_t T) {
    CR_CHECK_ARG(ctx && xs && Ns && Hs && Ws && V, "cr_wino_input: NULL pointer");
    WinoGeo g;
    unsigned blocks = 0;
    int rc = wino_geo("cr_wino_input", g, n, xs, nullptr, nullptr, Ns, Hs, Ws, C, T, &blocks);
    if (rc) return rc;
    hipLaunchKernelGGL(k_wino_input, dim3(blocks), dim3(256), 0, ctx->stream, g, V);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// M (16, T, O) -> ys[i] (N_i, H_i, W_i, O) = A^T M A + bias, ReLU if asked, + accs[i] (may be NULL / hold NULLs)
extern "C" int cr_wino_output(cr_ctx* ctx, int n, const float* M, float* const* ys, const int* Ns, const int* Hs, const int* Ws,
                              int O, int64_t T, const float* bias, int relu, const float* const* accs) {
    CR_CHECK_ARG(ctx && M && ys && Ns && Hs && Ws, "cr_wino_output: NULL pointer");
    WinoGeo g;
    unsigned blocks = 0;
    int rc = wino_geo("cr_wino_output", g, n, nullptr, ys, accs, Ns, Hs, Ws, O, T, &blocks);
    if (rc) return rc;
    hipLaunchKernelGGL(k_wino_output, dim3(blocks), dim3(256), 0, ctx->stream, g, M, bias, relu);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// dys: n host-array pointers to (N_i,H_i,W_i,O) maps -> dM (16,T,O) = A dY A^T; db (O) += per-channel sums of dY when given
extern "C" int cr_wino_dy(cr_ctx* ctx, int n, const float* const* dys, const int* Ns, const int* Hs, const int* Ws, int O,
                          float* dM, int64_t T, float* db) {
    CR_CHECK_ARG(ctx && dys && Ns && Hs && Ws && dM, "cr_wino_dy: NULL pointer");
    WinoGeo g;
    unsigned blocks = 0;
    int rc = wino_geo("cr_wino_dy", g, n, dys, nullptr, nullptr, Ns, Hs, Ws, O, T, &blocks);
    if (rc) return rc;
    hipLaunchKernelGGL(k_wino_dy, dim3(blocks), dim3(256), 0, ctx->stream, g, dM, db);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// dw (O,3,3,C) += G^T dU G, dU (16,O,C) = sum over tiles of dM^T V per transformed position
extern "C" int cr_wino_filter_grad(cr_ctx* ctx, const float* dU, float* dw, int O, int C) {
    CR_CHECK_ARG(ctx && dU && dw && O > 0 && C > 0, "cr_wino_filter_grad: bad args");
    hipLaunchKernelGGL(k_wino_filter_grad, dim3((unsigned)cr_cdiv((int64_t)O * C, 256)), dim3(256), 0, ctx->stream, dU, dw, O, C);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
