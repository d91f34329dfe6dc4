// Kernels of the Depth-Anything-V2 forward (DINOv2 ViT encoder + DPT head; reference:
// depth/metric_depth/depth_anything_v2/{dinov2.py, dinov2_layers/*.py, dpt.py, util/blocks.py}) that are not plain
// GEMMs / convolutions: fused multi-head attention, LayerNorm, exact GELU, LayerScale + residual, bilinear up-sampling.
// Token activations are (B*N, C) bf16 row-major; image-shaped activations NHWC bf16 like the rest of the library.
#include "cr_common.h"
#include <math.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __forceinline__ float vbf2f(u16 b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ u16 vf2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(u16, b);
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention forward, head_dim 64 (dinov2_layers/attention.py:49-62: softmax(q k^T / sqrt(d)) v on the packed qkv linear
// output).  Flash-attention schedule on 16x16x32 bf16 MFMA, written for the wave64 operand layout:
//   * one workgroup = 128 queries of one (batch, head), one wave = 16 queries; keys / values stream through LDS in tiles
//     of 64 (K row-major [key][d], V transposed [d][key]);
//   * S^T = K Q^T: the MFMA result holds, per lane, 16 scores of ONE query (column lane & 15) -> the row maximum and the
//     row sum need the lane's own values plus two xor-shuffles (lanes l, l^16, l^32, l^48 share a query);
//   * O^T += V^T P^T: the 8 probabilities a lane must supply as the B operand of a 32-key step are exactly the 8 it
//     already holds when the k index of the MFMA is mapped to keys as kk = 8g+j -> key 32s + 16(j/4) + 4g + (j%4); the
//     V^T fragments are read from LDS with the same mapping, so P never goes through memory;
//   * the 16 output accumulators of a lane all belong to its query, so the online-softmax rescale is one scalar per lane.
// qkv: (B, N, 3, H, 64) bf16 (row stride 3*H*64); out: (B, N, H, 64) bf16.
// ---------------------------------------------------------------------------------------------------------------------
#define ATT_D 64
#define ATT_TK 64
#define ATT_KPAD 8              // LDS row padding (elements): 144-byte rows, conflict-free 16-byte reads

// WAVES waves per workgroup share one K/V tile; every wave owns QB blocks of 16 queries (QB = 2: each K / V^T fragment
// fetched from LDS feeds two MFMAs).
template <int QB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_attention_fwd(const u16* __restrict__ qkv, u16* __restrict__ out, int B, int N,
                                                              int H, float scale, int qtiles) {
    constexpr int T = 64 * WAVES, TQ = 16 * QB * WAVES;
    __shared__ __attribute__((aligned(16))) u16 sK[ATT_TK][ATT_D + ATT_KPAD];      // [key][d]
    __shared__ __attribute__((aligned(16))) u16 sVt[ATT_D][ATT_TK + ATT_KPAD];     // [d][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    // workgroups are dispatched round-robin over the 8 XCDs (each with its own L2): renumber them so that the query
    // tiles of one (batch, head) -- which stream the same K / V -- are neighbours on ONE XCD instead of one per XCD
    const int nblk = gridDim.x, per = nblk >> 3, body = per << 3;
    const int lb = (int)blockIdx.x < body ? ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int bh = lb / qtiles, qt = lb - bh * qtiles;
    const int b = bh / H, h = bh - b * H;
    const int q0 = qt * TQ + wave * 16 * QB;
    const size_t row_stride = (size_t)3 * H * ATT_D;
    const u16* base = qkv + (size_t)b * N * row_stride + (size_t)h * ATT_D;
    const u16* Kb = base + (size_t)H * ATT_D;
    const u16* Vb = base + (size_t)2 * H * ATT_D;

    // Q fragments (B operand): query q0 + 16*qb + c, d = 32*s + 8*g .. +7
    bf16x8 qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + 16 * qb + c;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < N) v = *reinterpret_cast<const uint4*>(base + (size_t)q * row_stride + 32 * s + 8 * g);
            qf[qb][s] = __builtin_bit_cast(bf16x8, v);
        }
    }
    f32x4 o[QB][4];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[qb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (N + ATT_TK - 1) / ATT_TK;
    // staging: 64 keys x 64 d = 512 chunks of 8 elements over the workgroup; the global loads of tile t+1 are issued before
    // the MFMAs of tile t (registers), so their latency overlaps the compute instead of sitting between two barriers
    constexpr int NCH = 512 / T;
    uint4 rk[NCH], rv[NCH];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + T * i;
            const int key = t * ATT_TK + (idx >> 3), d8 = (idx & 7) * 8;
            rk[i] = make_uint4(0, 0, 0, 0);
            rv[i] = make_uint4(0, 0, 0, 0);
            if (key < N) {
                rk[i] = *reinterpret_cast<const uint4*>(Kb + (size_t)key * row_stride + d8);
                rv[i] = *reinterpret_cast<const uint4*>(Vb + (size_t)key * row_stride + d8);
            }
        }
    };
    load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * ATT_TK;
        __syncthreads();                               // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + T * i;
            const int key = idx >> 3, d8 = (idx & 7) * 8;
            *reinterpret_cast<uint4*>(&sK[key][d8]) = rk[i];
            const unsigned w[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
            // column swizzle key ^ 8*(d/8 % 8): the 8 lanes that hold the same key write rows 8 apart, which would all
            // fall on one LDS bank (row stride 36 dwords); the reads below apply the same XOR (it keeps aligned groups
            // of 4 keys together)
            const int kx = key ^ (((d8 >> 3) & 7) << 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sVt[d8 + 2 * e][kx] = (u16)(w[e] & 0xffff);
                sVt[d8 + 2 * e + 1][kx] = (u16)(w[e] >> 16);
            }
        }
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);

        // S^T blocks: rows = keys 16*kb + 4g + e, column = query c.  All K fragments are fetched from LDS first and the
        // MFMAs issued back to back over independent accumulators (a read -> wait -> MFMA chain per fragment exposes
        // the LDS latency 8 times per tile)
        bf16x8 kf[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                kf[s][kb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&sK[16 * kb + c][32 * s + 8 * g]));
        f32x4 s4[QB][4];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                s4[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0][kb], qf[qb][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                s4[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1][kb], qf[qb][1], s4[qb][kb], 0, 0, 0);
        // V^T fragments for the second product: issued now, consumed after the softmax arithmetic
        bf16x8 vf[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const int sw = ((2 * db + (c >> 3)) & 7) << 3;
                const uint2 v0 = *reinterpret_cast<const uint2*>(&sVt[16 * db + c][(32 * s + 4 * g) ^ sw]);
                const uint2 v1 = *reinterpret_cast<const uint2*>(&sVt[16 * db + c][(32 * s + 16 + 4 * g) ^ sw]);
                vf[s][db] = __builtin_bit_cast(bf16x8, make_uint4(v0.x, v0.y, v1.x, v1.y));
            }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + 16 * kb + 4 * g + e;
                    const float v = key < N ? s4[qb][kb][e] * scale : -INFINITY;
                    s4[qb][kb][e] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qb], mx);      // finite: every tile has at least one valid key
            const float alpha = __expf(m_run[qb] - m_new); // first tile: exp(-inf) = 0
            float psum = 0.f;
            unsigned pk[4][2];                             // P as bf16 pairs, [kb][e/2]
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                float p[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e] = __expf(s4[qb][kb][e] - m_new);
                    psum += p[e];
                }
                pk[kb][0] = (unsigned)vf2bf(p[0]) | ((unsigned)vf2bf(p[1]) << 16);
                pk[kb][1] = (unsigned)vf2bf(p[2]) | ((unsigned)vf2bf(p[3]) << 16);
            }
            l_run[qb] = l_run[qb] * alpha + psum;
            m_run[qb] = m_new;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[qb][i][0] *= alpha; o[qb][i][1] *= alpha; o[qb][i][2] *= alpha; o[qb][i][3] *= alpha;
            }
            // O^T[d, q] += sum_keys V^T[d, key] P^T[key, q]; step s covers key blocks 2s and 2s+1
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pk[2 * s][0], pk[2 * s][1], pk[2 * s + 1][0], pk[2 * s + 1][1]));
#pragma unroll
                for (int db = 0; db < 4; ++db) o[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[s][db], pf, o[qb][db], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l = l_run[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int q = q0 + 16 * qb + c;
        if (q < N) {
            const float inv = 1.f / l;
            u16* dst = out + ((size_t)b * N + q) * ((size_t)H * ATT_D) + (size_t)h * ATT_D;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 pkd;
                pkd.x = (unsigned)vf2bf(o[qb][db][0] * inv) | ((unsigned)vf2bf(o[qb][db][1] * inv) << 16);
                pkd.y = (unsigned)vf2bf(o[qb][db][2] * inv) | ((unsigned)vf2bf(o[qb][db][3] * inv) << 16);
                *reinterpret_cast<uint2*>(dst + 16 * db + 4 * g) = pkd;
            }
        }
    }
}

extern "C" int cr_attention_fwd(cr_ctx* ctx, const void* qkv, void* out, int B, int N, int H, int D, float scale) {
    CR_CHECK_ARG(ctx && B >= 0 && N >= 0 && H > 0, "cr_attention_fwd: bad args");
    CR_CHECK_ARG(D == ATT_D, "cr_attention_fwd: head dimension %d is not built (64 only)", D);
    if ((int64_t)B * N == 0) return CR_OK;
    CR_CHECK_ARG(qkv && out, "cr_attention_fwd: NULL pointer");
    // measured on ViT-L (4 x 16 heads x 1370 tokens): <QB=1, 8 waves> 86 us, <2, 4> 98 us, <2, 8> 102 us -- the second query
    // block per wave costs more in registers / occupancy than it saves in LDS fragment reads
    constexpr int QB = 1, WAVES = 8;
    const int qtiles = (int)cr_cdiv(N, 16 * QB * WAVES);
    CR_CHECK_ARG((int64_t)B * H * qtiles < (1ll << 31), "cr_attention_fwd: grid too large");
    hipLaunchKernelGGL((k_attention_fwd<QB, WAVES>), dim3((unsigned)(B * H * qtiles)), dim3(64 * WAVES), 0, ctx->stream,
                       (const u16*)qkv, (u16*)out, B, N, H, scale, qtiles);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dimension (nn.LayerNorm(eps=1e-6), dinov2.py:96): one wave per row, float32 statistics
// (mean, then the centred second moment), bf16 in / out.
// ---------------------------------------------------------------------------------------------------------------------
template <int NCH>      // 16-byte chunks per lane: C <= 512 * NCH
__global__ __launch_bounds__(256) void k_layernorm(const u16* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, u16* __restrict__ y, int64_t M, int C,
                                                   float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const u16* xr = x + row * C;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = lane * 8 + 512 * k;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (i < C) u = *reinterpret_cast<const uint4*>(xr + i);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[k][2 * e] = vbf2f((u16)(w[e] & 0xffff));
            v[k][2 * e + 1] = vbf2f((u16)(w[e] >> 16));
            s += v[k][2 * e] + v[k][2 * e + 1];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (lane * 8 + 512 * k < C) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = v[k][e] - mean;
                q += a * a;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float inv = rsqrtf(q / (float)C + eps);
    u16* yr = y + row * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = lane * 8 + 512 * k;
        if (i < C) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + i), g1 = *reinterpret_cast<const float4*>(gamma + i + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + i), b1 = *reinterpret_cast<const float4*>(beta + i + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            unsigned o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = (v[k][2 * e] - mean) * inv * gg[2 * e] + bb[2 * e];
                const float b2 = (v[k][2 * e + 1] - mean) * inv * gg[2 * e + 1] + bb[2 * e + 1];
                o[e] = (unsigned)vf2bf(a) | ((unsigned)vf2bf(b2) << 16);
            }
            *reinterpret_cast<uint4*>(yr + i) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// x_new = x + ls * y (LayerScale + residual) and h = LayerNorm(x_new) in one pass over the row: the statistics are taken
// from the bf16-rounded x_new, i.e. exactly what k_scale_residual followed by k_layernorm produces.
template <int NCH>
__global__ __launch_bounds__(256) void k_scale_residual_layernorm(const u16* __restrict__ x, const u16* __restrict__ yv,
                                                                  const float* __restrict__ ls, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, u16* __restrict__ xo,
                                                                  u16* __restrict__ ho, int64_t M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const u16* xr = x + row * C;
    const u16* yr = yv + row * C;
    u16* xor_ = xo + row * C;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = lane * 8 + 512 * k;
        if (i < C) {
            const uint4 a = *reinterpret_cast<const uint4*>(xr + i), b = *reinterpret_cast<const uint4*>(yr + i);
            const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
            unsigned o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float l0 = ls ? ls[i + 2 * e] : 1.f, l1 = ls ? ls[i + 2 * e + 1] : 1.f;
                const u16 r0 = vf2bf(vbf2f((u16)(aw[e] & 0xffff)) + l0 * vbf2f((u16)(bw[e] & 0xffff)));
                const u16 r1 = vf2bf(vbf2f((u16)(aw[e] >> 16)) + l1 * vbf2f((u16)(bw[e] >> 16)));
                o[e] = (unsigned)r0 | ((unsigned)r1 << 16);
                v[k][2 * e] = vbf2f(r0);
                v[k][2 * e + 1] = vbf2f(r1);
                s += v[k][2 * e] + v[k][2 * e + 1];
            }
            *reinterpret_cast<uint4*>(xor_ + i) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (lane * 8 + 512 * k < C) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = v[k][e] - mean;
                q += a * a;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float inv = rsqrtf(q / (float)C + eps);
    u16* hr = ho + row * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = lane * 8 + 512 * k;
        if (i < C) {
            unsigned o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = (v[k][2 * e] - mean) * inv * gamma[i + 2 * e] + beta[i + 2 * e];
                const float b2 = (v[k][2 * e + 1] - mean) * inv * gamma[i + 2 * e + 1] + beta[i + 2 * e + 1];
                o[e] = (unsigned)vf2bf(a) | ((unsigned)vf2bf(b2) << 16);
            }
            *reinterpret_cast<uint4*>(hr + i) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

extern "C" int cr_scale_residual_layernorm(cr_ctx* ctx, const void* x, const void* y, const float* ls, const float* gamma,
                                           const float* beta, void* x_out, void* h_out, int64_t M, int C, float eps) {
    CR_CHECK_ARG(ctx && M >= 0 && C > 0 && C % 8 == 0 && C <= 2048, "cr_scale_residual_layernorm: bad dims M=%lld C=%d", (long long)M, C);
    if (M == 0) return CR_OK;
    CR_CHECK_ARG(x && y && gamma && beta && x_out && h_out, "cr_scale_residual_layernorm: NULL pointer");
    const dim3 grid((unsigned)cr_cdiv(M, 4));
#define SRLN_LAUNCH(NCH_)                                                                                             \
    hipLaunchKernelGGL(k_scale_residual_layernorm<NCH_>, grid, dim3(256), 0, ctx->stream, (const u16*)x, (const u16*)y, ls, \
                       gamma, beta, (u16*)x_out, (u16*)h_out, M, C, eps)
    if (C <= 512) SRLN_LAUNCH(1);
    else if (C <= 1024) SRLN_LAUNCH(2);
    else SRLN_LAUNCH(4);
#undef SRLN_LAUNCH
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_layernorm(cr_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, int64_t M, int C,
                            float eps) {
    CR_CHECK_ARG(ctx && M >= 0 && C > 0 && C % 8 == 0, "cr_layernorm: bad dims M=%lld C=%d", (long long)M, C);
    if (M == 0) return CR_OK;
    CR_CHECK_ARG(x && gamma && beta && y, "cr_layernorm: NULL pointer");
    CR_CHECK_ARG(C <= 2048, "cr_layernorm: C=%d > 2048 is not built", C);
    const dim3 grid((unsigned)cr_cdiv(M, 4));
    if (C <= 512)
        hipLaunchKernelGGL(k_layernorm<1>, grid, dim3(256), 0, ctx->stream, (const u16*)x, gamma, beta, (u16*)y, M, C, eps);
    else if (C <= 1024)
        hipLaunchKernelGGL(k_layernorm<2>, grid, dim3(256), 0, ctx->stream, (const u16*)x, gamma, beta, (u16*)y, M, C, eps);
    else
        hipLaunchKernelGGL(k_layernorm<4>, grid, dim3(256), 0, ctx->stream, (const u16*)x, gamma, beta, (u16*)y, M, C, eps);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Element-wise: exact (erf) GELU in place (dinov2_layers/mlp.py: nn.GELU), and x + gamma * y (LayerScale + residual,
// dinov2_layers/block.py:84-110, layer_scale.py).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gelu(u16* __restrict__ x, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        uint4 v = *reinterpret_cast<uint4*>(x + i * 8);
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = vbf2f((u16)(w[e] & 0xffff)), b = vbf2f((u16)(w[e] >> 16));
            const float ga = 0.5f * a * (1.f + erff(a * 0.70710678118654752f));
            const float gb = 0.5f * b * (1.f + erff(b * 0.70710678118654752f));
            w[e] = (unsigned)vf2bf(ga) | ((unsigned)vf2bf(gb) << 16);
        }
        *reinterpret_cast<uint4*>(x + i * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

extern "C" int cr_gelu_inplace(cr_ctx* ctx, void* x, int64_t n) {
    CR_CHECK_ARG(ctx && n >= 0 && n % 8 == 0, "cr_gelu_inplace: n must be a multiple of 8");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(x, "cr_gelu_inplace: NULL pointer");
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(k_gelu, dim3((unsigned)(cr_cdiv(n8, 256) < 8192 ? cr_cdiv(n8, 256) : 8192)), dim3(256), 0, ctx->stream,
                       (u16*)x, n8);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

__global__ __launch_bounds__(256) void k_scale_residual(const u16* __restrict__ x, const u16* __restrict__ y,
                                                        const float* __restrict__ gamma, u16* __restrict__ out, int64_t M,
                                                        int C) {
    const int cg = C >> 3;
    const int64_t total = M * cg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cg) << 3;
        const uint4 xv = *reinterpret_cast<const uint4*>(x + i * 8), yv = *reinterpret_cast<const uint4*>(y + i * 8);
        const unsigned xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = vbf2f((u16)(xs[e] & 0xffff)) + (gamma ? gamma[c0 + 2 * e] : 1.f) * vbf2f((u16)(ys[e] & 0xffff));
            const float b = vbf2f((u16)(xs[e] >> 16)) + (gamma ? gamma[c0 + 2 * e + 1] : 1.f) * vbf2f((u16)(ys[e] >> 16));
            o[e] = (unsigned)vf2bf(a) | ((unsigned)vf2bf(b) << 16);
        }
        *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int cr_scale_residual(cr_ctx* ctx, const void* x, const void* y, const float* gamma, void* out, int64_t M, int C) {
    CR_CHECK_ARG(ctx && M >= 0 && C > 0 && C % 8 == 0, "cr_scale_residual: bad dims");
    if (M == 0) return CR_OK;
    CR_CHECK_ARG(x && y && out, "cr_scale_residual: NULL pointer");
    const int64_t total = M * (C >> 3);
    hipLaunchKernelGGL(k_scale_residual, dim3((unsigned)(cr_cdiv(total, 256) < 8192 ? cr_cdiv(total, 256) : 8192)), dim3(256), 0,
                       ctx->stream, (const u16*)x, (const u16*)y, gamma, (u16*)out, M, C);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Bilinear resize with align_corners=True (F.interpolate in util/blocks.py:139 and dpt.py:150), NHWC bf16, 8 channels per
// thread; optional ReLU on the way out (the consumer of a fusion block starts with an activation).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_bilinear_ac(const u16* __restrict__ x, u16* __restrict__ y, int B, int h,
                                                            int w, int Ho, int Wo, int C) {
    const int cg = C >> 3;
    const int64_t total = (int64_t)B * Ho * Wo * cg;
    const float sy = Ho > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cg) << 3;
        int64_t r = i / cg;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const u16* p = x + (size_t)b * h * w * C + c0;
        const uint4 v00 = *reinterpret_cast<const uint4*>(p + ((size_t)y0 * w + x0) * C);
        const uint4 v01 = *reinterpret_cast<const uint4*>(p + ((size_t)y0 * w + x1) * C);
        const uint4 v10 = *reinterpret_cast<const uint4*>(p + ((size_t)y1 * w + x0) * C);
        const uint4 v11 = *reinterpret_cast<const uint4*>(p + ((size_t)y1 * w + x1) * C);
        const unsigned a[4] = {v00.x, v00.y, v00.z, v00.w}, bq[4] = {v01.x, v01.y, v01.z, v01.w};
        const unsigned cq[4] = {v10.x, v10.y, v10.z, v10.w}, d[4] = {v11.x, v11.y, v11.z, v11.w};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r2[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int sh = hh * 16;
                const float t00 = vbf2f((u16)(a[e] >> sh)), t01 = vbf2f((u16)(bq[e] >> sh));
                const float t10 = vbf2f((u16)(cq[e] >> sh)), t11 = vbf2f((u16)(d[e] >> sh));
                const float top = t00 + (t01 - t00) * lx, bot = t10 + (t11 - t10) * lx;
                r2[hh] = top + (bot - top) * ly;
            }
            o[e] = (unsigned)vf2bf(r2[0]) | ((unsigned)vf2bf(r2[1]) << 16);
        }
        *reinterpret_cast<uint4*>(y + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int cr_resize_bilinear_ac(cr_ctx* ctx, const void* x, void* y, int B, int h, int w, int Ho, int Wo, int C) {
    CR_CHECK_ARG(ctx && B >= 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "cr_resize_bilinear_ac: bad dims");
    if (B == 0) return CR_OK;
    CR_CHECK_ARG(x && y, "cr_resize_bilinear_ac: NULL pointer");
    const int64_t total = (int64_t)B * Ho * Wo * (C >> 3);
    hipLaunchKernelGGL(k_resize_bilinear_ac, dim3((unsigned)(cr_cdiv(total, 256) < 16384 ? cr_cdiv(total, 256) : 16384)),
                       dim3(256), 0, ctx->stream, (const u16*)x, (u16*)y, B, h, w, Ho, Wo, C);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
