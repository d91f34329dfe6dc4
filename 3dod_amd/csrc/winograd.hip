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

// The three map-side transforms share one decomposition: a wave owns one 2 x 2 output tile at a time and its 64 lanes cover
// 64 * VEC consecutive channels with VEC floats each (16-byte accesses for C % 256 == 0: whole 1 KB channel rows per wave
// instruction); a block of 4 waves takes TPB consecutive tiles of one map and one channel group.
typedef float vf1 __attribute__((ext_vector_type(1)));
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
#define WINO_TPB 8          // tiles per block of the input / output transforms
#define WINO_TPB_DY 16      // ... of the dY transform (fewer bias atomics)
#define WINO_DB_SLOTS 16    // partial bias-gradient rows the dY transform spreads its atomics over

struct WinoTile { int n, ty, tx, c; bool ok; };
template <int VEC>
__device__ __forceinline__ WinoTile wino_tile(const WinoGeo& g, const WinoMap& m, int tile_in_block, int tpb) {
    const int cgs = g.C / (64 * VEC);
    const int local = (int)blockIdx.x - m.blk0;
    const int cg = local % cgs, chunk = local / cgs;
    const int th = m.H >> 1, tw = m.W >> 1;
    const int t = chunk * tpb + tile_in_block;
    WinoTile w;
    w.ok = t < m.N * th * tw;
    w.n = t / (th * tw);
    const int r = t - w.n * (th * tw);
    w.ty = r / tw; w.tx = r - w.ty * tw;
    w.c = (cg * 64 + (threadIdx.x & 63)) * VEC;
    return w;
}

template <typename VT>
__global__ __launch_bounds__(256) void k_wino_input(WinoGeo g, float* __restrict__ V) {
    constexpr int VEC = sizeof(VT) / 4;
    const WinoMap& m = g.m[wino_find(g, (int)blockIdx.x)];
    const int C = g.C, th = m.H >> 1, tw = m.W >> 1;
    const size_t plane = (size_t)g.T * C;
    for (int k = threadIdx.x >> 6; k < WINO_TPB; k += 4) {
        const WinoTile w = wino_tile<VEC>(g, m, k, WINO_TPB);
        if (!w.ok) break;
        VT d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 2 * w.ty - 1 + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = 2 * w.tx - 1 + j;
                d[i][j] = ((unsigned)y < (unsigned)m.H && (unsigned)x < (unsigned)m.W)
                              ? *(const VT*)(m.src + (((size_t)w.n * m.H + y) * m.W + x) * C + w.c) : (VT)(0.f);
            }
        }
        VT t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
        }
        float* o = V + ((size_t)m.tbase + ((size_t)w.n * th + w.ty) * tw + w.tx) * C + w.c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(VT*)(o + (i * 4 + 0) * plane) = t[i][0] - t[i][2];
            *(VT*)(o + (i * 4 + 1) * plane) = t[i][1] + t[i][2];
            *(VT*)(o + (i * 4 + 2) * plane) = t[i][2] - t[i][1];
            *(VT*)(o + (i * 4 + 3) * plane) = t[i][1] - t[i][3];
        }
    }
}

template <typename VT>
__global__ __launch_bounds__(256) void k_wino_output(WinoGeo g, const float* __restrict__ M, const float* __restrict__ bias, int relu) {
    constexpr int VEC = sizeof(VT) / 4;
    const WinoMap& m = g.m[wino_find(g, (int)blockIdx.x)];
    const int C = g.C, th = m.H >> 1, tw = m.W >> 1;     // C: OUTPUT channels of the product here
    const size_t plane = (size_t)g.T * C;
    for (int k = threadIdx.x >> 6; k < WINO_TPB; k += 4) {
        const WinoTile w = wino_tile<VEC>(g, m, k, WINO_TPB);
        if (!w.ok) break;
        VT b = (VT)(0.f);                               // (a parameter may sit at any 4-byte offset of a flat buffer)
        if (bias) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) b[e] = bias[w.c + e];
        }
        const float* o = M + ((size_t)m.tbase + ((size_t)w.n * th + w.ty) * tw + w.tx) * C + w.c;
        VT q[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) q[i][j] = *(const VT*)(o + (i * 4 + j) * plane);
        VT s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[0][j] = (q[0][j] + q[1][j]) + q[2][j]; s[1][j] = (q[1][j] - q[2][j]) - q[3][j]; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            VT y0 = (s[i][0] + s[i][1]) + s[i][2] + b, y1 = (s[i][1] - s[i][2]) - s[i][3] + b;
            const size_t p = (((size_t)w.n * m.H + 2 * w.ty + i) * m.W + 2 * w.tx) * C + w.c;
            if (relu) { y0 = __builtin_elementwise_max(y0, (VT)(0.f)); y1 = __builtin_elementwise_max(y1, (VT)(0.f)); }
            if (m.acc) { y0 += *(const VT*)(m.acc + p); y1 += *(const VT*)(m.acc + p + C); }
            *(VT*)(m.dst + p) = y0;
            *(VT*)(m.dst + p + C) = y1;
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
// when db_part (WINO_DB_SLOTS, C) is given, db_part[block % slots][c] += sum of dY (f32 atomics, one per workgroup and channel,
// spread over the slots: thousands of atomics on ONE address serialise in L2); k_wino_filter_grad folds the slots into db
template <typename VT>
__global__ __launch_bounds__(256) void k_wino_dy(WinoGeo g, float* __restrict__ dM, float* __restrict__ db) {
    constexpr int VEC = sizeof(VT) / 4;
    __shared__ VT s_sum[4][64];
    const WinoMap& m = g.m[wino_find(g, (int)blockIdx.x)];
    const int C = g.C, th = m.H >> 1, tw = m.W >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t plane = (size_t)g.T * C;
    VT acc = (VT)(0.f);
    int c = wino_tile<VEC>(g, m, 0, WINO_TPB_DY).c;
    for (int k = wave; k < WINO_TPB_DY; k += 4) {
        const WinoTile w = wino_tile<VEC>(g, m, k, WINO_TPB_DY);
        if (!w.ok) break;
        const float* p = m.src + (((size_t)w.n * m.H + 2 * w.ty) * m.W + 2 * w.tx) * C + w.c;
        const VT d00 = *(const VT*)p, d01 = *(const VT*)(p + C), d10 = *(const VT*)(p + (size_t)m.W * C),
                 d11 = *(const VT*)(p + (size_t)m.W * C + C);
        acc += (d00 + d01) + (d10 + d11);
        const VT t[4][2] = {{d00, d01}, {d00 + d10, d01 + d11}, {d00 - d10, d01 - d11}, {-d10, -d11}};
        float* o = dM + ((size_t)m.tbase + ((size_t)w.n * th + w.ty) * tw + w.tx) * C + w.c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(VT*)(o + (i * 4 + 0) * plane) = t[i][0];
            *(VT*)(o + (i * 4 + 1) * plane) = t[i][0] + t[i][1];
            *(VT*)(o + (i * 4 + 2) * plane) = t[i][0] - t[i][1];
            *(VT*)(o + (i * 4 + 3) * plane) = -t[i][1];
        }
    }
    if (db) {
        s_sum[wave][lane] = acc;
        __syncthreads();
        if (wave == 0) {
            const VT v = (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) atomicAdd(db + (size_t)(blockIdx.x % WINO_DB_SLOTS) * C + c + e, v[e]);
        }
    }
}

// dW[o][r][s][c] += (G^T dU G)[r][s] from dU (16, O, C): the weight gradient back from the transformed domain, added into the
// flat-gradient view of the (O,3,3,C) weight (one thread per (o, c): no atomics)
__global__ __launch_bounds__(256) void k_wino_filter_grad(const float* __restrict__ dU, float* __restrict__ dw, int O, int C,
                                                          const float* __restrict__ db_part, float* __restrict__ db) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= O * C) return;
    const int o = idx / C, c = idx - o * C;
    if (db && c == 0) {                                  // fixed-order sum of the partial bias-gradient rows
        float sum = 0.f;
        for (int k = 0; k < WINO_DB_SLOTS; ++k) sum += db_part[(size_t)k * O + o];
        db[o] += sum;
    }
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
                    const int* Ns, const int* Hs, const int* Ws, int C, int64_t T, int tpb, unsigned* blocks, int* vec) {
    CR_CHECK_ARG(n >= 1 && n <= WINO_MAX_MAPS, "%s: 1..%d maps", who, WINO_MAX_MAPS);
    CR_CHECK_ARG(C > 0 && C % 64 == 0, "%s: channels %% 64", who);
    g.n = n; g.C = C; g.T = (int)T;
    const int v = C % 256 == 0 ? 4 : C % 128 == 0 ? 2 : 1;           // floats per lane: 16-byte accesses when the channels allow
    *vec = v;
    int tb = 0, blk = 0;
    for (int i = 0; i < n; ++i) {
        CR_CHECK_ARG(Ns[i] > 0 && Hs[i] > 0 && Ws[i] > 0 && Hs[i] % 2 == 0 && Ws[i] % 2 == 0, "%s: map %d needs even H and W", who, i);
        g.m[i].src = srcs ? srcs[i] : nullptr; g.m[i].dst = dsts ? dsts[i] : nullptr; g.m[i].acc = accs ? accs[i] : nullptr;
        g.m[i].N = Ns[i]; g.m[i].H = Hs[i]; g.m[i].W = Ws[i]; g.m[i].tbase = tb; g.m[i].blk0 = blk;
        tb += Ns[i] * (Hs[i] / 2) * (Ws[i] / 2);
        blk += (int)cr_cdiv((int64_t)Ns[i] * (Hs[i] / 2) * (Ws[i] / 2), tpb) * (C / (64 * v));
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
    int vec = 1;
    int rc = wino_geo("cr_wino_input", g, n, xs, nullptr, nullptr, Ns, Hs, Ws, C, T, WINO_TPB, &blocks, &vec);
    if (rc) return rc;
    if (vec == 4) hipLaunchKernelGGL(k_wino_input<vf4>, dim3(blocks), dim3(256), 0, ctx->stream, g, V);
    else if (vec == 2) hipLaunchKernelGGL(k_wino_input<vf2>, dim3(blocks), dim3(256), 0, ctx->stream, g, V);
    else hipLaunchKernelGGL(k_wino_input<vf1>, dim3(blocks), dim3(256), 0, ctx->stream, g, V);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// M (16, T, O) -> ys[i] (N_i, H_i, W_i, O) = A^T M A + bias, ReLU if asked, + accs[i] (may be NULL / hold NULLs)
extern "C" int cr_wino_output(cr_ctx* ctx, int n, const float* M, float* const* ys, const int* Ns, const int* Hs, const int* Ws,
                              int O, int64_t T, const float* bias, int relu, const float* const* accs) {
    CR_CHECK_ARG(ctx && M && ys && Ns && Hs && Ws, "cr_wino_output: NULL pointer");
    WinoGeo g;
    unsigned blocks = 0;
    int vec = 1;
    int rc = wino_geo("cr_wino_output", g, n, nullptr, ys, accs, Ns, Hs, Ws, O, T, WINO_TPB, &blocks, &vec);
    if (rc) return rc;
    if (vec == 4) hipLaunchKernelGGL(k_wino_output<vf4>, dim3(blocks), dim3(256), 0, ctx->stream, g, M, bias, relu);
    else if (vec == 2) hipLaunchKernelGGL(k_wino_output<vf2>, dim3(blocks), dim3(256), 0, ctx->stream, g, M, bias, relu);
    else hipLaunchKernelGGL(k_wino_output<vf1>, dim3(blocks), dim3(256), 0, ctx->stream, g, M, bias, relu);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// dys: n host-array pointers to (N_i,H_i,W_i,O) maps -> dM (16,T,O) = A dY A^T; db_part (16,O), zeroed by the caller, += partial
// per-channel sums of dY when given (cr_wino_filter_grad adds their total to the bias gradient)
extern "C" int cr_wino_dy(cr_ctx* ctx, int n, const float* const* dys, const int* Ns, const int* Hs, const int* Ws, int O,
                          float* dM, int64_t T, float* db) {
    CR_CHECK_ARG(ctx && dys && Ns && Hs && Ws && dM, "cr_wino_dy: NULL pointer");
    WinoGeo g;
    unsigned blocks = 0;
    int vec = 1;
    int rc = wino_geo("cr_wino_dy", g, n, dys, nullptr, nullptr, Ns, Hs, Ws, O, T, WINO_TPB_DY, &blocks, &vec);
    if (rc) return rc;
    if (vec == 4) hipLaunchKernelGGL(k_wino_dy<vf4>, dim3(blocks), dim3(256), 0, ctx->stream, g, dM, db);
    else if (vec == 2) hipLaunchKernelGGL(k_wino_dy<vf2>, dim3(blocks), dim3(256), 0, ctx->stream, g, dM, db);
    else hipLaunchKernelGGL(k_wino_dy<vf1>, dim3(blocks), dim3(256), 0, ctx->stream, g, dM, db);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// dw (O,3,3,C) += G^T dU G, dU (16,O,C) = sum over tiles of dM^T V per transformed position; db (O) += the column sums of
// db_part (16,O) from cr_wino_dy when both are given
extern "C" int cr_wino_filter_grad(cr_ctx* ctx, const float* dU, float* dw, int O, int C, const float* db_part, float* db) {
    CR_CHECK_ARG(ctx && dU && dw && O > 0 && C > 0 && (!db == !db_part), "cr_wino_filter_grad: bad args");
    hipLaunchKernelGGL(k_wino_filter_grad, dim3((unsigned)cr_cdiv((int64_t)O * C, 256)), dim3(256), 0, ctx->stream, dU, dw, O, C,
                       db_part, db);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
