// Convolution stack of the Cube R-CNN DLA34/ResNet34 + FPN + RPN-head path as
// implicit GEMMs on the CDNA4 matrix cores (bf16 in, f32 accumulate), NHWC.
//
//   k_conv_igemm   forward conv and backward-data (transposed gather) in one
//                  template: out[pixel][ch] = sum_k  W[ch][k] * X_gather[pixel][k]
//                  with fused bias / residual add / ReLU / BatchNorm statistics.
//   k_conv_wgrad   backward-weight: dW[ch][k] = sum_pixels dY[pixel][ch] * X_gather[pixel][k]
//                  (both operands read column-wise from LDS with ds_read_b64_tr_b16),
//                  split over pixel ranges, f32 atomics.
//
// Reference modules covered (paths into the reference tree):
//   cubercnn/modeling/backbone/dla.py:40-68,156-174,233-321  (conv3x3/1x1/7x7 + BN + ReLU + residual)
//   detectron2 FPN / StandardRPNHead [third-party], wired at dla.py:484-507, configs/Base.yaml:41-60
//
// MFMA operand maps used (cdna_hip_programming.md section 3), v_mfma_f32_16x16x32_bf16:
//   A: lane l holds A[row l&15][k = 8(l>>4)+j]   B: lane l holds B[k = 8(l>>4)+j][col l&15]
//   D: col = l&15, row = 4(l>>4)+reg.
// We feed WEIGHTS as A (rows = output channels) and PIXELS as B (cols = pixels), so a lane
// ends up with 4 consecutive output channels of one pixel = one 8-byte NHWC store.
//
// Two arithmetic modes, selected per call by the storage type of the activations (act_f32 of the C ABI):
//   bf16  operands bf16, v_mfma_f32_16x16x32_bf16, f32 accumulate   (fast mode)
//   f32   operands f32,  v_mfma_f32_16x16x4_f32 (exact f32 fmaf chain, 157 TFLOP/s peak) -- the reference's precision
// Both share the kernels: an LDS row is 64 bytes = four 16-B chunks = 32 bf16 or 16 f32 of k, so the LDS image, the
// gathers and the fragment reads are byte-identical; only the k extent of a sub-step (SUB = 32 / 16) and the MFMA differ.
// In f32 mode lane l's 16-B fragment holds k = 4(l>>4) .. +3 of its row; MFMA e (0..3) of a sub-step multiplies element
// e of every lane's A and B fragments, i.e. the k set {e, 4+e, 8+e, 12+e} -- the same permutation on both operands.
#include "cr_common.h"
#include "cr_elem.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CONV_T 256
#define STAT_REPL 32        // (legacy name) -- BN statistics are now per-block partial sums, reduced in a fixed order

struct ConvP {
    const void* x;       // gather source, NHWC bf16 or f32 (fwd: input; bwd-data: dY)
    const void* w;       // [Cout][Kdim] in the activations' type, k = (r*KS + s)*Cin + c
    void* y;             // [M][Cout] bf16 or f32
    const void* res;     // optional residual [M][Cout] in the activations' type (added before ReLU)
    const float* bias;   // optional [Cout]
    float* stats;        // optional [ceil(M/128)][2][Cout]: per-M-tile sum and sum of squares of the conv output
    int N, Hin, Win, Cin, Hout, Wout, Cout;
    int stride, pad, Kdim, M, cshift, relu;
    unsigned x_bytes, w_bytes;   // extents for the buffer-load descriptors (out-of-range voffset reads 0)
    int xcd;                     // 1 = XCD-aware tile order
    int f32;                     // 1 = f32 operands / outputs (host-side dispatch only)
    // split-K over workgroups (k_conv_igemm_dma): ksplit > 1 -> block (tile, s) accumulates k stages [s*kstages, ...) and
    // stores its raw f32 accumulators to part[s][M][Cout]; k_splitk_epilogue sums them in a fixed order and applies the
    // epilogue (bias, BN statistics, residual, ReLU)
    float* part;
    int ksplit, kstages;
    // backward-data of a stride-2 convolution by output-pixel parity class (k_conv_igemm, MODE 1, cls = 1): the output pixels
    // (2h'+ph, 2w'+pw) of one class only receive the filter taps r = ph+pad (mod 2), s = pw+pad (mod 2) -- 4 + 2 + 2 + 1 of
    // the 9 taps of a 3x3 filter -- so each class is a stride-1 gather over its own taps and no MFMA multiplies the zeros
    // that the plain transposed gather inserts for the other 3/4 (tap, pixel) pairs.  Hout/Wout/M then describe ONE class
    // (H/2 x W/2 pixels); the grid holds the four classes back to back; Hfull/Wfull address the output.
    int cls, Hfull, Wfull, wstride;      // wstride = k extent of a weight row (= Kdim except in class mode)
    // split mode (k_conv_igemm_dma_s3): the weights as three bf16 planes, [row][Kdim/32][plane][32] (cr_weight_split3)
    const void* w3;
    unsigned w3_bytes;
};

// offset (in elements) of output pixel m's channel row
__device__ __forceinline__ size_t conv_out_row(const ConvP& p, int m, int ph, int pw) {
    if (!p.cls) return (size_t)m * p.Cout;
    const int hw = p.Hout * p.Wout;
    const int n = m / hw, rem = m - n * hw;
    const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
    return ((size_t)(n * p.Hfull + 2 * ho + ph) * p.Wfull + 2 * wo + pw) * p.Cout;
}

// LDS image of a [rows][32] bf16 tile (64-B rows, four 16-B chunks): chunk' = chunk ^ ((-(row>>2)) & 3)
// makes the 16x16x32 fragment read (lane -> row l&15, chunk l>>4) conflict-free per ds_read_b128 lane group.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  xcd_swizzle() renumbers them so that XCD x
// works on one contiguous range of logical tiles: neighbouring tiles (which share input rows / weight tiles / the pixel
// range of a wgrad split) then hit the same L2 instead of each XCD re-fetching the operand from the Infinity Cache.
__device__ __forceinline__ int xcd_swizzle(int b, int nblocks) {
    const int per = nblocks >> 3, body = per << 3;
    return b < body ? (b & 7) * per + (b >> 3) : b;
}

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 32 + ((chunk ^ ((-(row >> 2)) & 3)) << 3);
}

// one sub-step of MFMAs on 16-B fragments: T = u16 -> one 16x16x32 bf16 MFMA; T = float -> four 16x16x4 f32 MFMAs
template <typename T, int TC, int TP>
__device__ __forceinline__ void mfma_substep(const u32x4 (&wf)[TC], const u32x4 (&xf)[TP], f32x4 (&acc)[TC][TP]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i]),
                                                                    __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
    } else {
        // e outermost: consecutive MFMAs hit different accumulators (40-cycle dependent latency vs 32-cycle issue)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[i][e]), __uint_as_float(xf[j][e]),
                                                                     acc[i][j], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------
// f32 arithmetic on the bf16 matrix cores ("split" mode, act_f32 = 2).  An f32 value x is EXACTLY the sum of three bf16
// values: h = the top 8 significant bits (x truncated to bf16), m = the next 8 (x - h truncated), l = the last 8
// (x - h - m, exact).  A product w * x is then the sum of nine bf16 x bf16 products, each exact in the MFMA's f32
// datapath; the three smallest (wm*xl, wl*xm, wl*xl <= 2^-23 |w x|) are below the rounding of an f32 product and are
// dropped, the other six are accumulated in f32 by v_mfma_f32_16x16x32_bf16.  Six bf16 MFMAs per 32 of k cost 96 cycles
// where the f32 MFMA needs 8 x 32 = 256: 2.7x the f32 matrix peak at f32 accuracy (measured against f64 in
// tests/test_gpu_convops_f32.py: same error as the f32 MFMA path).  The split costs VALU work (2 and + 2 sub per element
// + v_perm packing), done once per staged element.
// ---------------------------------------------------------------------------
// 8 f32 (two 16-B chunks) -> the three bf16x8 planes; element order kept (element 2q, 2q+1 -> dword q)
__device__ __forceinline__ void split3(const u32x4& c0, const u32x4& c1, u32x4& H, u32x4& M, u32x4& L) {
    const unsigned xs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned a = xs[2 * q], b = xs[2 * q + 1];
        H[q] = __builtin_amdgcn_perm(b, a, 0x07060302u);                 // [b.hi16 : a.hi16]
        const float ra = __uint_as_float(a) - __uint_as_float(a & 0xffff0000u);
        const float rb = __uint_as_float(b) - __uint_as_float(b & 0xffff0000u);
        const unsigned ua = __float_as_uint(ra), ub = __float_as_uint(rb);
        M[q] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
        const float la = ra - __uint_as_float(ua & 0xffff0000u);
        const float lb = rb - __uint_as_float(ub & 0xffff0000u);
        L[q] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
    }
}

// Shared epilogue of the implicit-GEMM kernels: bias, BN statistics, residual, ReLU, store (4 consecutive channels per
// lane).  sStat: >= 4*2*BN floats of LDS that nothing else uses any more (the k loop ended with a barrier).
template <int BM, int BN, int TC, int TP, typename OutT, typename T = u16>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x4 (&acc)[TC][TP], float* sStat, int m0, int n0, int mt,
                                              int poff, int coff, int tid, int lane, int wave, bool lead, int ph = 0, int pw = 0) {
    const bool do_stats = p.stats != nullptr;
    if (do_stats) {
        if (lead) for (int i = tid; i < 4 * 2 * BN; i += CONV_T) sStat[i] = 0.f;
        __syncthreads();
    }
    const int g = lane >> 4, pl = lane & 15;
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int chl = coff + i * 16 + 4 * g;     // channel within the block tile
        const int ch = n0 + chl;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) b4[e] = p.bias[ch + e];
        }
        float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int m = m0 + poff + j * 16 + pl;
            if (m < p.M && lead) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[e];
                if (do_stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
                }
                const size_t o = conv_out_row(p, m, ph, pw) + ch;
                if (p.res) {
                    if constexpr (sizeof(T) == 2) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(p.res) + o);
                        v[0] += bf2f((u16)(rr.x & 0xffff)); v[1] += bf2f((u16)(rr.x >> 16));
                        v[2] += bf2f((u16)(rr.y & 0xffff)); v[3] += bf2f((u16)(rr.y >> 16));
                    } else {
                        const float4 rr = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + o);
                        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (sizeof(OutT) == 4) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 pk;
                    pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                    pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(p.y) + o) = pk;
                }
            }
        }
        if (do_stats) {
            // reduce over the 16 pixel lanes that share this lane's channel group
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    ssum[e] += __shfl_xor(ssum[e], off, 64);
                    ssq[e] += __shfl_xor(ssq[e], off, 64);
                }
            }
            if (pl == 0 && lead) {  // exactly one lane per (wave, channel): plain stores
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sStat[(wave * 2 + 0) * BN + chl + e] = ssum[e];
                    sStat[(wave * 2 + 1) * BN + chl + e] = ssq[e];
                }
            }
        }
    }
    if (do_stats) {
        __syncthreads();
        if (!lead) return;
        // statistics rows are per 64 pixels so that their number does not depend on the tile choice:
        // a 128-pixel tile writes its sums to row 2*mt and zeros to row 2*mt+1
        const int srow = (BM == 128) ? 2 * mt : mt;
        float* dst = p.stats + (size_t)srow * 2 * p.Cout;
        if (BM == 128 && (size_t)(srow + 1) * 64 < (size_t)p.M + 64) {
            float* dz = dst + 2 * p.Cout;
            if ((int64_t)(srow + 1) * 64 < (int64_t)p.M)
                for (int i = tid; i < BN; i += CONV_T) { dz[n0 + i] = 0.f; dz[p.Cout + n0 + i] = 0.f; }
        }
        for (int i = tid; i < BN; i += CONV_T) {
            const float a = ((sStat[(0 * 2 + 0) * BN + i] + sStat[(1 * 2 + 0) * BN + i]) + sStat[(2 * 2 + 0) * BN + i]) + sStat[(3 * 2 + 0) * BN + i];
            const float b = ((sStat[(0 * 2 + 1) * BN + i] + sStat[(1 * 2 + 1) * BN + i]) + sStat[(2 * 2 + 1) * BN + i]) + sStat[(3 * 2 + 1) * BN + i];
            dst[n0 + i] = a;
            dst[p.Cout + n0 + i] = b;
        }
    }
}

// KU = 32-wide k sub-steps per pipeline stage: small tiles on small grids are bound by the latency of one
// global->LDS round trip per stage, so they take 4 sub-steps (BK = 128) per round trip.
// KG = wave groups per block (intra-block split-K): layers whose grid is <= ~1 block per CU run 4 groups of 4 waves on
// interleaved k sub-steps of the same output tile (4 waves per SIMD to overlap the per-sub-step instruction/latency
// chain that a single wave per SIMD exposes) and reduce the accumulators through LDS before the shared epilogue.
template <int BM, int BN, int KS, int MODE, typename OutT, int KU, int KG, typename T = u16>
__global__ __launch_bounds__(CONV_T * KG) void k_conv_igemm(ConvP p) {
    constexpr int ES = (int)sizeof(T);               // bytes per operand element
    constexpr int KE = 16 / ES;                      // k elements per 16-B chunk (8 bf16 / 4 f32)
    constexpr int SUB = 4 * KE;                      // k extent of one sub-step = one 64-B LDS row (32 / 16)
    // 4 waves tile the BM x BN block: 2x2 for the square-ish tiles, 4x1 (pixels) for narrow channel tiles
    constexpr int WAVES_M = (BN == 16) ? 4 : ((BN == 128 || BM == 64) ? 2 : 4);
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int TP = BM / WAVES_M / 16;            // pixel tiles (16) per wave
    constexpr int TC = BN / WAVES_N / 16;            // channel tiles per wave
    constexpr int NA = BM * 4 / CONV_T;              // pixel-row chunks per thread (2 or 1)
    constexpr int NB = (BN * 4 + CONV_T - 1) / CONV_T;   // weight chunks per thread
    static_assert(TP >= 1 && TC >= 1 && NA >= 1, "tile too small for 4 waves");
    constexpr int XE = KU * BM * 32, WE = KU * BN * 32;          // LDS u16 units per group (rows of 64 B)
    __shared__ __attribute__((aligned(16))) u16 smem[KG * (XE + WE)];
    u16* sXall = smem;
    u16* sWall = smem + KG * XE;
    static_assert(4 * 2 * BN * sizeof(float) <= XE * sizeof(u16), "sStat must fit in sX");
    static_assert(KG == 1 || (KG - 1) * BM * BN * sizeof(float) <= KG * (XE + WE) * sizeof(u16), "reduction buffer");
    float* sStat = reinterpret_cast<float*>(smem);    // [wave][2][BN], epilogue only (the k loop ends with a barrier)

    const int grp = KG == 1 ? 0 : (int)(threadIdx.x >> 8);       // k group of this wave
    const int tid = threadIdx.x & (CONV_T - 1), lane = tid & 63, wave = tid >> 6;
    u16* sX = sXall + grp * XE;
    u16* sW = sWall + grp * WE;
    const int n_tiles = p.Cout / BN;
    int bid = p.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    // parity-class mode (stride-2 backward-data): the grid is 4 classes x tiles; this block's class, its taps and k extent
    int ph = 0, pw = 0, nsr = KS, kdim = p.Kdim;
    unsigned tapr = 0, taps_ = 0;                 // packed lists (2 bits each) of the filter rows / columns of the class
    if (MODE == 1 && p.cls) {
        const int per = (int)gridDim.x >> 2;
        const int c = bid / per;
        bid -= c * per;
        ph = c >> 1; pw = c & 1;
        int nr = 0, ns = 0;
        for (int r = 0; r < KS; ++r) {
            if (((ph + p.pad - r) & 1) == 0) tapr |= (unsigned)r << (2 * nr++);
            if (((pw + p.pad - r) & 1) == 0) taps_ |= (unsigned)r << (2 * ns++);
        }
        nsr = ns;
        kdim = nr * ns * p.Cin;
    }
    const int nt = bid % n_tiles, mt = bid / n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    // filter tap (row r, column s2) of tap index j: all KS*KS taps, or the class's own
    auto tap_rs = [&](int j, int& r, int& s2) {
        if (MODE == 1 && p.cls) {
            const int jr = j / nsr, js = j - jr * nsr;
            r = (int)((tapr >> (2 * jr)) & 3u); s2 = (int)((taps_ >> (2 * js)) & 3u);
        } else { r = j / KS; s2 = j - r * KS; }
    };

    // ---- per-thread gather bookkeeping (2 pixel rows, one 16-B k-chunk each)
    const int cA = tid & 3;
    int nimg[NA], hb[NA], wb[NA];
    bool rv[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        rv[i] = m < p.M;
        const int mm = rv[i] ? m : 0;
        const int hw = p.Hout * p.Wout;
        const int n = mm / hw;
        const int rem = mm - n * hw;
        const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
        nimg[i] = n;
        if (MODE == 0) { hb[i] = ho * p.stride - p.pad; wb[i] = wo * p.stride - p.pad; }
        else if (p.cls) { hb[i] = 2 * ho + ph + p.pad;  wb[i] = 2 * wo + pw + p.pad; }
        else           { hb[i] = ho + p.pad;            wb[i] = wo + p.pad; }
    }
    uint4 ra[KU][NA], rb[KU][NB];

    // All global reads are raw buffer loads: masked-out elements (padding taps, k tail, rows past M) get a voffset
    // past the descriptor's extent and read as zero, so the gather is branch-free.  (With flat loads under exec-mask
    // branches the compiler put `s_waitcnt vmcnt(0)` in front of every load of the loop, serialising them.)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    // Fast gather path (KS > 1, Cin % SUB == 0): a sub-step never straddles a filter tap, so the tap
    // decomposition, the bounds test and the pixel offset are recomputed only when the tap changes.
    const bool fast = (KS > 1) && ((p.Cin & (SUB - 1)) == 0);
    int cur_tap = -1;
    unsigned tb[NA];                                // byte offset of (pixel, current tap, chunk cA) or OOB
#pragma unroll
    for (int i = 0; i < NA; ++i) tb[i] = OOB;
    auto pix_off = [&](int i, int r, int s2, bool kin) -> unsigned {
        int hi, wi;
        bool ok = rv[i] && kin;
        if (MODE == 0) { hi = hb[i] + r; wi = wb[i] + s2; }
        else {
            const int th = hb[i] - r, tw = wb[i] - s2;
            if (p.stride == 2) { ok = ok && (((th | tw) & 1) == 0); hi = th >> 1; wi = tw >> 1; }
            else { hi = th; wi = tw; }
        }
        ok = ok && ((unsigned)hi < (unsigned)p.Hin) && ((unsigned)wi < (unsigned)p.Win);
        const unsigned off = (unsigned)(((nimg[i] * p.Hin + hi) * p.Win + wi) * p.Cin) * (unsigned)ES;
        return ok ? off : OOB;
    };

    auto load_tiles = [&](int stage) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int kt = (stage * KG + grp) * KU + u;     // groups take interleaved stages
        if (fast) {
            const int tap = (kt * SUB) >> p.cshift, cc = (kt * SUB) & (p.Cin - 1);
            if (tap != cur_tap) {
                int r, s2;
                tap_rs(tap, r, s2);
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const unsigned o = pix_off(i, r, s2, kt * SUB < kdim);
                    tb[i] = o == OOB ? OOB : o + cA * 16;
                }
                cur_tap = tap;
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[u][i] = buf_load16(rx, tb[i] + (unsigned)cc * (unsigned)ES);
        } else {
            const int k0 = kt * SUB + cA * KE;
            int r = 0, s2 = 0, c0 = k0;
            if (KS > 1) {
                const int tap = k0 >> p.cshift;
                c0 = k0 & (p.Cin - 1);
                tap_rs(tap, r, s2);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const unsigned o = pix_off(i, r, s2, k0 < kdim);
                ra[u][i] = buf_load16(rx, o == OOB ? OOB : o + (unsigned)c0 * (unsigned)ES);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + CONV_T * i;
            const int row = idx >> 2, kb = kt * SUB + (idx & 3) * KE;
            const bool ok = idx < BN * 4 && kb < kdim;
            int wcol = kb;                               // column of k index kb in the weight row
            if (MODE == 1 && p.cls) {
                int r, s2;
                tap_rs(kb >> p.cshift, r, s2);
                wcol = (r * KS + s2) * p.Cin + (kb & (p.Cin - 1));
            }
            rb[u][i] = buf_load16(rw, ok ? (unsigned)((n0 + row) * p.wstride + wcol) * (unsigned)ES : OOB);
        }
      }
    };
    auto store_tiles = [&]() {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            *reinterpret_cast<uint4*>(&sX[u * BM * 32 + lds_off((tid >> 2) + 64 * i, cA)]) = ra[u][i];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + CONV_T * i;
            if (idx < BN * 4) *reinterpret_cast<uint4*>(&sW[u * BN * 32 + lds_off(idx >> 2, idx & 3)]) = rb[u][i];
        }
      }
    };

    const int poff = (wave / WAVES_N) * (BM / WAVES_M);            // pixel offset of this wave in the tile
    const int coff = (wave % WAVES_N) * (BN / WAVES_N);            // channel offset

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (kdim + SUB - 1) / SUB;
    const int nstage = (nk + KU * KG - 1) / (KU * KG);
    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        load_tiles(st + 1);        // next stage in flight under the MFMAs (past the end: every offset is OOB -> zeros, no traffic)
        const int fr = lane & 15, fc = lane >> 4;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            if (KG == 1 && KU > 1 && st * KU + u >= nk) break;  // block-uniform k tail (KG > 1: the tail multiplies zeros)
            u32x4 xf[TP], wf[TC];
#pragma unroll
            for (int j = 0; j < TP; ++j)
                xf[j] = *reinterpret_cast<const u32x4*>(&sX[u * BM * 32 + lds_off(poff + j * 16 + fr, fc)]);
#pragma unroll
            for (int i = 0; i < TC; ++i)
                wf[i] = *reinterpret_cast<const u32x4*>(&sW[u * BN * 32 + lds_off(coff + i * 16 + fr, fc)]);
            mfma_substep<T, TC, TP>(wf, xf, acc);
        }
        __syncthreads();
        store_tiles();
        __syncthreads();
    }

    if (KG > 1) {
        // groups 1.. park their accumulators in LDS ([group-1][wave][reg][lane]: lane-contiguous, conflict-free),
        // group 0 adds them in a fixed order (bitwise reproducible) and runs the epilogue alone
        float* red = reinterpret_cast<float*>(smem);
        constexpr int PER_WAVE = TC * TP * 4 * 64;
        if (grp > 0) {
            float* dst = red + ((grp - 1) * 4 + wave) * PER_WAVE + lane;
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[((i * TP + j) * 4 + e) * 64] = acc[i][j][e];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
        for (int g2 = 1; g2 < KG; ++g2) {
            const float* src = red + ((g2 - 1) * 4 + wave) * PER_WAVE + lane;
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] += src[((i * TP + j) * 4 + e) * 64];
        }
        }
        __syncthreads();                 // `red` is dead from here on (sStat aliases it)
    }
    const bool lead = grp == 0;          // groups 1.. only keep the epilogue's barriers company
    conv_epilogue<BM, BN, TC, TP, OutT, T>(p, acc, sStat, m0, n0, mt, poff, coff, tid, lane, wave, lead, ph, pw);
}

// ---------------------------------------------------------------------------
// k_conv_igemm_dma: the implicit GEMM of k_conv_igemm for the LARGE 3x3 / 1x1 layers (forward and backward-data), 128-pixel
// x BN-channel tiles, with the operand tiles fetched by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers) into
// TWO 64-deep LDS buffers: a whole k stage is in flight under the MFMAs of the previous one and there is one barrier pair
// per 64 (not 32) of k.  3x3 256->256 on 4x128x128: 138 -> 90 us (561 -> 860 TFLOP/s), bit-identical output (same MFMA
// order: k in steps of 32, ascending).
//   * The DMA writes lane l's 16 bytes at base + 16 l, i.e. 16 rows x four 16-B chunks in order; the conflict-free image
//     of lds_off() (chunk' = chunk ^ f(row)) is produced on the SOURCE side: lane l fetches logical chunk (l & 3) ^ f(row).
//   * The compiler puts `s_waitcnt vmcnt(0)` in front of every LDS read it can see while a DMA is outstanding, so the
//     fragment reads are inline `ds_read_b128` and the waits are explicit: vmcnt(N) (this wave's N DMAs of the NEXT stage
//     may stay in flight) + barrier before a buffer is read, lgkmcnt(0) tied to the fragment registers before the MFMAs,
//     barrier before the buffer is refilled.
// Requires Cin % 64 == 0 (a 64-wide k stage never straddles a filter tap) and Cout % BN == 0; bf16 output.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// generic -> LDS address space.  Kept out of the kernel template: there the cast sits in a value-dependent expression, is
// re-checked when the template is instantiated, fails in the HOST pass (silently: device-code diagnostics are deferred)
// and the kernel's host stub is never emitted.
__device__ __forceinline__ lds_ptr_t to_lds(u16* p) { return (lds_ptr_t)p; }
__device__ __forceinline__ unsigned lds_addr(u16* p) { return (unsigned)(size_t)(lds_ptr_t)p; }
// one LDS-DMA: 64 lanes x 16 B from base + voff[lane] (0 past the descriptor's extent) to dst + 16 lane.  Also a plain
// function for the same reason: the target builtin inside an instantiation-dependent call is re-checked per instantiation.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, u16* dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ u32x4 lds_read16_asm(unsigned byte_addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}

// KG = 2: two wave groups per block (8 waves) on alternating k stages, each with its own pair of LDS buffers; their
// accumulators are summed through LDS before the epilogue.  For grids of <= one block per CU (the 64x64 ... 16x16 maps in
// f32): two waves per SIMD cover each other's LDS / barrier latencies, which one wave per SIMD leaves exposed (87 -> ~110
// TFLOP/s on the 3x3 128->128 layers), without the slab traffic of more split-K.
// BM = 64: half-height tiles for the f32 layers whose 128-pixel tiling gives fewer than one block per CU -- twice the
// blocks without split-K slabs (a 64 x 128 tile moves 0.047 B/flop, 6 TB/s at the f32 peak: still under the L2 rate).
// (the body is a device function of (parameters, block index, block count): k_conv_igemm_dma runs it on its own grid,
// k_conv_igemm_dma_grp on one problem of a group)
template <int BN, int KS, int MODE, typename T = u16, int KG = 1, int BM = 128>
__device__ __forceinline__ void conv_igemm_dma_body(const ConvP& p, const int blk_id, const int blk_count) {
    constexpr int ES = (int)sizeof(T), SUB = 64 / ES;                        // k per 64-B LDS row: 32 bf16 / 16 f32
    constexpr int BK = 2 * SUB, NBX = BM / 64;                               // pixel rows per thread (chunks of 64 rows)
    constexpr int WAVES_M = BN == 128 ? 2 : 4, WAVES_N = 4 / WAVES_M;      // as k_conv_igemm: same statistics order
    constexpr int TP = BM / WAVES_M / 16, TC = BN / WAVES_N / 16;
    constexpr int NBW = BN / 64;                                            // weight rows per thread (chunks of 64 rows)
    constexpr int XS = BM * 32, WS = BN * 32;                               // elements of one 32-deep sub-block
    constexpr int STAGE = 2 * XS + 2 * WS;                                  // X(u=0), X(u=1), W(u=0), W(u=1)
    constexpr int NDMA = 2 * NBX + 2 * NBW;                                       // DMA instructions per thread per stage
    __shared__ __attribute__((aligned(1024))) u16 smem_all[KG * 2 * STAGE]; // per group 64 KiB (BN=128) / 48 KiB (BN=64)
    const int grp = KG == 1 ? 0 : (int)(threadIdx.x >> 8);
    u16* smem = smem_all + grp * 2 * STAGE;
    static_assert(4 * 2 * BN * sizeof(float) <= 2 * STAGE * sizeof(u16), "sStat must fit");
    const int tid = threadIdx.x & (CONV_T - 1), lane = tid & 63, wave = tid >> 6;
    const int n_tiles = p.Cout / BN;
    const int tiles_total = blk_count / p.ksplit;                           // ksplit == 1: the whole grid
    const int ks = blk_id / tiles_total, bt = blk_id - ks * tiles_total;
    const int bid = (p.xcd && (p.ksplit == 1 || (tiles_total & 7) == 0)) ? xcd_swizzle(bt, tiles_total) : bt;
    const int nt = bid % n_tiles, mt = bid / n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // rows this lane fetches: (tid >> 2) + 64 i  (wave w: rows 16 w .. 16 w + 15 of each 64-row half)
    int nimg[NBX], hb[NBX], wb[NBX];
    bool rv[NBX];
    unsigned csrc[NBX];                                                        // logical chunk (bytes) fetched for row i
#pragma unroll
    for (int i = 0; i < NBX; ++i) {
        const int row = (tid >> 2) + 64 * i;
        const int m = m0 + row;
        rv[i] = m < p.M;
        const int mm = rv[i] ? m : 0;
        const int hw = p.Hout * p.Wout;
        const int n = mm / hw;
        const int rem = mm - n * hw;
        const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
        nimg[i] = n;
        if (MODE == 0) { hb[i] = ho * p.stride - p.pad; wb[i] = wo * p.stride - p.pad; }
        else           { hb[i] = ho + p.pad;            wb[i] = wo + p.pad; }
        csrc[i] = (unsigned)(((tid & 3) ^ ((-(row >> 2)) & 3)) * 16);
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned wrow[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) wrow[i] = (unsigned)((n0 + (tid >> 2) + 64 * i) * p.Kdim) * (unsigned)ES + csrc[0];   // (the swizzle repeats every 64 rows)

    const int nstage_all = p.Kdim / BK;
    const int st0 = ks * p.kstages, st1 = min(st0 + p.kstages, nstage_all);  // this block's k stages (all of them when ksplit == 1)
    int cur_tap = -1;
    unsigned tb[NBX];
#pragma unroll
    for (int i = 0; i < NBX; ++i) tb[i] = OOB;                     // byte offset of (pixel row i, current filter tap, this lane's chunk) or OOB
    auto issue = [&](int st, int buf) {
        const int k0 = st * BK;
        const int tap = KS == 1 ? 0 : (k0 >> p.cshift), cc = KS == 1 ? k0 : (k0 & (p.Cin - 1));
        if (tap != cur_tap) {                        // block-uniform: the tap decomposition and bounds test once per tap
            const int r = tap / KS, s2 = tap - r * KS;
#pragma unroll
            for (int i = 0; i < NBX; ++i) {
                int hi, wi;
                bool ok = rv[i];
                if (MODE == 0) { hi = hb[i] + r; wi = wb[i] + s2; }
                else {
                    const int th = hb[i] - r, tw = wb[i] - s2;
                    if (p.stride == 2) { ok = ok && (((th | tw) & 1) == 0); hi = th >> 1; wi = tw >> 1; }
                    else { hi = th; wi = tw; }
                }
                ok = ok && ((unsigned)hi < (unsigned)p.Hin) && ((unsigned)wi < (unsigned)p.Win);
                tb[i] = ok ? (unsigned)(((nimg[i] * p.Hin + hi) * p.Win + wi) * p.Cin) * (unsigned)ES + csrc[i] : OOB;
            }
            cur_tap = tap;
        }
        u16* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NBX; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                dma16(rx, base + u * XS + (wave * 16 + 64 * i) * 32, tb[i] == OOB ? OOB : tb[i] + (unsigned)(cc + u * SUB) * (unsigned)ES);
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                dma16(rw, base + 2 * XS + u * WS + (wave * 16 + 64 * i) * 32, wrow[i] + (unsigned)(k0 + u * SUB) * (unsigned)ES);
    };

    const int poff = (wave / WAVES_N) * (BM / WAVES_M), coff = (wave % WAVES_N) * (BN / WAVES_N);
    const int fr = lane & 15, fc = lane >> 4;
    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned lds0 = lds_addr(smem);                                   // byte address of smem in LDS
    unsigned xa[TP], wa[TC];                                                // fragment byte offsets inside a sub-block
#pragma unroll
    for (int j = 0; j < TP; ++j) xa[j] = (unsigned)lds_off(poff + j * 16 + fr, fc) * 2u;
#pragma unroll
    for (int i = 0; i < TC; ++i) wa[i] = (unsigned)lds_off(coff + i * 16 + fr, fc) * 2u;

    // group g takes stages st0 + g, st0 + g + KG, ...; every group runs the same number of iterations (shared barriers), a
    // group past its last stage neither fetches nor multiplies
    const int niter = (st1 - st0 + KG - 1) / KG;
    if (st0 + grp < st1) issue(st0 + grp, 0);
    for (int it = 0; it < niter; ++it) {
        const int st = st0 + grp + it * KG;
        const int buf = it & 1;
        if (st + KG < st1) {
            issue(st + KG, buf ^ 1);
            static_assert(NDMA == 8 || NDMA == 6 || NDMA == 4, "vmcnt literal");
            if (NDMA == 8)      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // stage st has landed (this wave's DMAs)
            else if (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                        // ... and every other wave's
        const unsigned sb = lds0 + (unsigned)(buf * STAGE) * 2u;
        if (KG == 1 || st < st1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            u32x4 xf[TP], wf[TC];
#pragma unroll
            for (int j = 0; j < TP; ++j) xf[j] = lds_read16_asm(sb + (unsigned)(u * XS) * 2u + xa[j]);
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[i] = lds_read16_asm(sb + (unsigned)(2 * XS + u * WS) * 2u + wa[i]);
            if (TP == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[TP - 2]), "+v"(xf[TP - 1]),
                                      "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]));
            else         asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0]), "+v"(xf[TP - 1]),
                                      "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]));
            mfma_substep<T, TC, TP>(wf, xf, acc);
        }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // everyone is done with `buf` before stage st+2 refills it
    }
    if constexpr (KG > 1) {
        // sum the groups' accumulators: register order through LDS (the operand buffers are dead), 16 B per lane
        float4* red = reinterpret_cast<float4*>(smem_all);
#pragma unroll
        for (int gg = 1; gg < KG; ++gg) {
            if (grp == gg) {
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
                        red[((wave * TC + i) * TP + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        const float4 v = red[((wave * TC + i) * TP + j) * 64 + lane];
                        acc[i][j][0] += v.x; acc[i][j][1] += v.y; acc[i][j][2] += v.z; acc[i][j][3] += v.w;
                    }
            }
            __syncthreads();
        }
    }
    if (p.ksplit > 1) {
        // raw partial sums of this k range: part[ks][m][ch], 4 consecutive channels per lane (16-B stores)
        if (grp != 0) return;
        const int g = lane >> 4, pl = lane & 15;
        float* dst = p.part + (size_t)ks * p.M * p.Cout;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int m = m0 + poff + j * 16 + pl;
                if (m < p.M)
                    *reinterpret_cast<float4*>(dst + (size_t)m * p.Cout + n0 + coff + i * 16 + 4 * g) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        return;
    }
    conv_epilogue<BM, BN, TC, TP, T, T>(p, acc, reinterpret_cast<float*>(smem_all), m0, n0, mt, poff, coff, tid, lane, wave, grp == 0);
}

template <int BN, int KS, int MODE, typename T = u16, int KG = 1, int BM = 128>
__global__ __launch_bounds__(CONV_T * KG) void k_conv_igemm_dma(ConvP p) {
    conv_igemm_dma_body<BN, KS, MODE, T, KG, BM>(p, (int)blockIdx.x, (int)gridDim.x);
}

// Grouped launch: up to CR_MAX_GROUP independent convolutions of one geometry class (same filter size, stride 1, Cin, Cout,
// precision) in ONE grid of 128 x 128 tiles -- the five levels of the FPN output convolutions / the RPN head convolution.
// Alone, the 32 x 32 ... 8 x 8 levels are a handful of tiles each: they take the split-K path (partial slabs + an epilogue
// launch) and stay latency-bound at 28 ... 45 us for 0.3 ... 4.8 GFLOP; inside the group's grid their tiles run beside the
// 128 x 128 level's at its rate.  Problem i owns blocks [start[i], start[i] + count[i]); starts are multiples of 8 so that
// every problem's XCD-aware tile order sees the round-robin deal it assumes (the padding blocks exit at once).
#define CR_MAX_GROUP 8
struct ConvGroup { ConvP p[CR_MAX_GROUP]; int start[CR_MAX_GROUP], count[CR_MAX_GROUP]; int n; };

template <int KS, int MODE, typename T>
__global__ __launch_bounds__(CONV_T) void k_conv_igemm_dma_grp(ConvGroup g) {
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.start[i + 1]) ++i;
    const int blk = (int)blockIdx.x - g.start[i];
    if (blk >= g.count[i]) return;                                            // alignment padding (block-uniform)
    conv_igemm_dma_body<128, KS, MODE, T, 1, 128>(g.p[i], blk, g.count[i]);
}

// k_conv_igemm_dma_s3: k_conv_igemm_dma in split mode (f32 activations and results, six bf16 MFMAs per tile pair and 32
// of k; see split3).  The pixels arrive as f32 by LDS-DMA exactly as in the f32 kernel (two 64-B sub-rows = 32 of k per
// stage); a lane's two 16-B fragments (k = 4g..4g+3 and 16+4g..16+4g+3 of the stage) are split in registers.  The
// weights arrive PRE-SPLIT (cr_weight_split3, once per optimizer step for the whole model): per (row, 32 of k) three
// 64-B rows, one per plane, whose element order is the pixels' fragment order (chunk c = k {4c..4c+3, 16+4c..16+4c+3}),
// so one ds_read_b128 per plane is a lane's MFMA operand.  LDS: 2 x (16 KB pixels + 3 x BN x 64 B weights) = 80 / 56 KB.
template <int BN, int KS, int MODE>
__global__ __launch_bounds__(CONV_T) void k_conv_igemm_dma_s3(ConvP p) {
    constexpr int BM = 128, BK = 32;
    // every wave takes ALL BN channels of 32 pixels: the split is per pixel fragment, so TP = 2 halves its VALU work per MFMA
    // against the 64 x 64 wave tile (the two waves of a SIMD share its vector issue: 2 x 176 VALU + 192 MFMA issue slots per
    // stage pair did not fit under the 192 x 16 MFMA cycles)
    constexpr int WAVES_M = 4, WAVES_N = 1;
    constexpr int TP = BM / WAVES_M / 16, TC = BN / WAVES_N / 16;
    constexpr int NBW = BN / 64;
    constexpr int XS = BM * 32, WS = BN * 32;                               // u16 units of one 64-B-row sub-block
    constexpr int STAGE = 2 * XS + 3 * WS;                                  // X(k 0..15), X(k 16..31), W planes h, m, l
    constexpr int NDMA = 4 + 3 * NBW;
    __shared__ __attribute__((aligned(1024))) u16 smem[2 * STAGE];
    static_assert(4 * 2 * BN * sizeof(float) <= 2 * STAGE * sizeof(u16), "sStat must fit");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = p.Cout / BN;
    const int tiles_total = (int)gridDim.x / p.ksplit;
    const int ks = (int)blockIdx.x / tiles_total, bt = (int)blockIdx.x - ks * tiles_total;
    const int bid = (p.xcd && (p.ksplit == 1 || (tiles_total & 7) == 0)) ? xcd_swizzle(bt, tiles_total) : bt;
    const int nt = bid % n_tiles, mt = bid / n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    int nimg[2], hb[2], wb[2];
    bool rv[2];
    unsigned csrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 2) + 64 * i;
        const int m = m0 + row;
        rv[i] = m < p.M;
        const int mm = rv[i] ? m : 0;
        const int hw = p.Hout * p.Wout;
        const int n = mm / hw;
        const int rem = mm - n * hw;
        const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
        nimg[i] = n;
        if (MODE == 0) { hb[i] = ho * p.stride - p.pad; wb[i] = wo * p.stride - p.pad; }
        else           { hb[i] = ho + p.pad;            wb[i] = wo + p.pad; }
        csrc[i] = (unsigned)(((tid & 3) ^ ((-(row >> 2)) & 3)) * 16);
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w3_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int nstage_all = p.Kdim / BK;
    unsigned wrow[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) wrow[i] = (unsigned)((n0 + (tid >> 2) + 64 * i) * nstage_all) * 192u + csrc[i];

    const int st0 = ks * p.kstages, st1 = min(st0 + p.kstages, nstage_all);
    int cur_tap = -1;
    unsigned tb[2] = {OOB, OOB};
    auto issue = [&](int st, int buf) {
        const int k0 = st * BK;
        const int tap = KS == 1 ? 0 : (k0 >> p.cshift), cc = KS == 1 ? k0 : (k0 & (p.Cin - 1));
        if (tap != cur_tap) {
            const int r = tap / KS, s2 = tap - r * KS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int hi, wi;
                bool ok = rv[i];
                if (MODE == 0) { hi = hb[i] + r; wi = wb[i] + s2; }
                else {
                    const int th = hb[i] - r, tw = wb[i] - s2;
                    if (p.stride == 2) { ok = ok && (((th | tw) & 1) == 0); hi = th >> 1; wi = tw >> 1; }
                    else { hi = th; wi = tw; }
                }
                ok = ok && ((unsigned)hi < (unsigned)p.Hin) && ((unsigned)wi < (unsigned)p.Win);
                tb[i] = ok ? (unsigned)(((nimg[i] * p.Hin + hi) * p.Win + wi) * p.Cin) * 4u + csrc[i] : OOB;
            }
            cur_tap = tap;
        }
        u16* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                dma16(rx, base + u * XS + (wave * 16 + 64 * i) * 32, tb[i] == OOB ? OOB : tb[i] + (unsigned)(cc + u * 16) * 4u);
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                dma16(rw, base + 2 * XS + pl * WS + (wave * 16 + 64 * i) * 32, wrow[i] + (unsigned)st * 192u + (unsigned)pl * 64u);
    };

    const int poff = (wave / WAVES_N) * (BM / WAVES_M), coff = (wave % WAVES_N) * (BN / WAVES_N);
    const int fr = lane & 15, fc = lane >> 4;
    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned lds0 = lds_addr(smem);
    unsigned xa[TP], wa[TC];
#pragma unroll
    for (int j = 0; j < TP; ++j) xa[j] = (unsigned)lds_off(poff + j * 16 + fr, fc) * 2u;
#pragma unroll
    for (int i = 0; i < TC; ++i) wa[i] = (unsigned)lds_off(coff + i * 16 + fr, fc) * 2u;

    issue(st0, 0);
    for (int st = st0; st < st1; ++st) {
        const int buf = (st - st0) & 1;
        if (st + 1 < st1) {
            issue(st + 1, buf ^ 1);
            if (NDMA == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else            asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const unsigned sb = lds0 + (unsigned)(buf * STAGE) * 2u;
        u32x4 wf[3][TC], x0[TP], x1[TP];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[pl][i] = lds_read16_asm(sb + (unsigned)(2 * XS + pl * WS) * 2u + wa[i]);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            x0[j] = lds_read16_asm(sb + xa[j]);
            x1[j] = lds_read16_asm(sb + (unsigned)XS * 2u + xa[j]);
        }
        // all fragment reads have returned (the asm ties the wait to the registers the MFMAs / the split consume)
        static_assert(TP == 2, "wait below names two pixel fragments");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0[0]), "+v"(x0[1]), "+v"(x1[0]), "+v"(x1[1]));
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < TC; i += 4)
                asm volatile("" : "+v"(wf[pl][i]), "+v"(wf[pl][i + 1]), "+v"(wf[pl][i + 2]), "+v"(wf[pl][i + 3]));
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            u32x4 xs[3];
            split3(x0[j], x1[j], xs[0], xs[1], xs[2]);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                constexpr int PW[6] = {2, 1, 0, 1, 0, 0}, PX[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < TC; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[PW[t]][i]),
                                                                        __builtin_bit_cast(bf16x8, xs[PX[t]]), acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (p.ksplit > 1) {
        const int g = lane >> 4, pl = lane & 15;
        float* dst = p.part + (size_t)ks * p.M * p.Cout;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int m = m0 + poff + j * 16 + pl;
                if (m < p.M)
                    *reinterpret_cast<float4*>(dst + (size_t)m * p.Cout + n0 + coff + i * 16 + 4 * g) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        return;
    }
    conv_epilogue<BM, BN, TC, TP, float, float>(p, acc, reinterpret_cast<float*>(smem), m0, n0, mt, poff, coff, tid, lane, wave, true);
}

// f32 [rows][K] (K % 32 == 0) -> split-mode weight planes [rows][K/32][3][32 bf16]; within a 64-B plane row, 16-B chunk c
// holds k = 32 kb + {4c..4c+3, 16+4c..16+4c+3}.  One thread per (row, kb, chunk).
__global__ __launch_bounds__(256) void k_weight_split3(const float* __restrict__ src, u16* __restrict__ dst, int64_t rows, int K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int KB = K >> 5;
    if (i >= rows * KB * 4) return;
    const int c = (int)(i & 3);
    const int64_t rk = i >> 2;                            // row * KB + kb
    const int64_t row = rk / KB;
    const int kb = (int)(rk - row * KB);
    const float* s0 = src + row * K + kb * 32 + 4 * c;
    const u32x4 a = *reinterpret_cast<const u32x4*>(s0), b = *reinterpret_cast<const u32x4*>(s0 + 16);
    u32x4 H, M, L;
    split3(a, b, H, M, L);
    u16* d = dst + rk * 96 + c * 8;
    *reinterpret_cast<u32x4*>(d) = H;
    *reinterpret_cast<u32x4*>(d + 32) = M;
    *reinterpret_cast<u32x4*>(d + 64) = L;
}

extern "C" int cr_weight_split3(cr_ctx* ctx, const float* src, void* dst, int64_t rows, int K) {
    CR_CHECK_ARG(ctx && rows >= 0 && K > 0 && (K & 31) == 0, "cr_weight_split3: K=%d must be a positive multiple of 32", K);
    if (rows == 0) return CR_OK;
    CR_CHECK_ARG(src && dst && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, "cr_weight_split3: NULL or misaligned pointer");
    const int64_t n = rows * (K >> 5) * 4;
    hipLaunchKernelGGL(k_weight_split3, dim3((unsigned)cr_cdiv(n, 256)), dim3(256), 0, ctx->stream, src, (u16*)dst, rows, K);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// the same for many matrices in one launch (all conv weights of the model, views into one flat buffer): desc d covers the
// items [item0, item0 + rows * K/32 * 4) -- an item is one 16-B chunk of one plane row
struct cr_s3desc { int64_t src_off, dst_off, item0; int rows, K; };
__global__ __launch_bounds__(256) void k_weights_split3(const float* __restrict__ src_base, u16* __restrict__ dst_base,
                                                        const cr_s3desc* __restrict__ descs, int ndesc, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = ndesc - 1;                           // last desc with item0 <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].item0 <= i) lo = mid; else hi = mid - 1;
    }
    const cr_s3desc d = descs[lo];
    const int64_t li = i - d.item0;
    const int KB = d.K >> 5;
    const int c = (int)(li & 3);
    const int64_t rk = li >> 2;
    const int64_t row = rk / KB;
    const int kb = (int)(rk - row * KB);
    const float* s0 = src_base + d.src_off + row * d.K + kb * 32 + 4 * c;
    const u32x4 a = *reinterpret_cast<const u32x4*>(s0), b = *reinterpret_cast<const u32x4*>(s0 + 16);
    u32x4 H, M, L;
    split3(a, b, H, M, L);
    u16* o = dst_base + d.dst_off + rk * 96 + c * 8;
    *reinterpret_cast<u32x4*>(o) = H;
    *reinterpret_cast<u32x4*>(o + 32) = M;
    *reinterpret_cast<u32x4*>(o + 64) = L;
}

extern "C" int cr_weights_split3(cr_ctx* ctx, const float* src_base, void* dst_base, const void* descs_dev, int ndesc,
                                 int64_t total_items) {
    CR_CHECK_ARG(ctx && ndesc >= 0 && total_items >= 0, "cr_weights_split3: bad args");
    if (ndesc == 0 || total_items == 0) return CR_OK;
    CR_CHECK_ARG(src_base && dst_base && descs_dev, "cr_weights_split3: NULL pointer");
    hipLaunchKernelGGL(k_weights_split3, dim3((unsigned)cr_cdiv(total_items, 256)), dim3(256), 0, ctx->stream, src_base,
                       (u16*)dst_base, (const cr_s3desc*)descs_dev, ndesc, total_items);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// second phase of the split-K launches: out[m][c] = epilogue( sum_s part[s][m][c] ), the sum in a fixed order (bitwise
// reproducible), the epilogue as conv_epilogue's: bias, BatchNorm statistics of the pre-residual value (one row of partial
// sums per 64 pixels), residual, ReLU.  A block = 64 pixels x 64 channels; a thread = 4 pixels x 4 channels.
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_epilogue(ConvP p) {
    __shared__ float sS[16][2][64];
    const int tid = threadIdx.x, cg4 = tid & 15, pr = tid >> 4;
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64 + cg4 * 4;
    const size_t MC = (size_t)p.M * p.Cout;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] = p.bias[c0 + e];
    }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + pr + 16 * q;
        if (m >= p.M) continue;
        const size_t o = (size_t)m * p.Cout + c0;
        float4 a = *reinterpret_cast<const float4*>(p.part + o);
        for (int s2 = 1; s2 < p.ksplit; ++s2) {
            const float4 t = *reinterpret_cast<const float4*>(p.part + (size_t)s2 * MC + o);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        float v[4] = {a.x + b4[0], a.y + b4[1], a.z + b4[2], a.w + b4[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
        if (p.res) {
            if constexpr (sizeof(T) == 2) {
                const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(p.res) + o);
                v[0] += bf2f((u16)(rr.x & 0xffff)); v[1] += bf2f((u16)(rr.x >> 16));
                v[2] += bf2f((u16)(rr.y & 0xffff)); v[3] += bf2f((u16)(rr.y >> 16));
            } else {
                const float4 rr = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + o);
                v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            }
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint2 pk;
            pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
            pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(p.y) + o) = pk;
        }
    }
    if (p.stats == nullptr) return;                      // block-uniform
#pragma unroll
    for (int e = 0; e < 4; ++e) { sS[pr][0][cg4 * 4 + e] = ssum[e]; sS[pr][1][cg4 * 4 + e] = ssq[e]; }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += sS[r][which][c];      // fixed order
        p.stats[((size_t)blockIdx.x * 2 + which) * p.Cout + blockIdx.y * 64 + c] = a;
    }
}


template <int BM, int BN, int KS, int MODE, int KU = 1, int KG = 1>
static int launch_igemm_t(cr_ctx* ctx, const ConvP& p, int out_f32) {
    const int grid = (int)(cr_cdiv(p.M, BM) * (p.Cout / BN)) * (p.cls ? 4 : 1);
    if (out_f32)
        hipLaunchKernelGGL((k_conv_igemm<BM, BN, KS, MODE, float, KU, KG>), dim3(grid), dim3(CONV_T * KG), 0, ctx->stream, p);
    else
        hipLaunchKernelGGL((k_conv_igemm<BM, BN, KS, MODE, u16, KU, KG>), dim3(grid), dim3(CONV_T * KG), 0, ctx->stream, p);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

template <int BM, int BN, int KS, int MODE, int KU>
static int launch_igemm_f32(cr_ctx* ctx, const ConvP& p) {
    const int grid = (int)(cr_cdiv(p.M, BM) * (p.Cout / BN)) * (p.cls ? 4 : 1);
    hipLaunchKernelGGL((k_conv_igemm<BM, BN, KS, MODE, float, KU, 1, float>), dim3(grid), dim3(CONV_T), 0, ctx->stream, p);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static int xcd_enabled() {
    static const int v = env_int("CR_XCD", 1);
    return v;
}

// literal template arguments from a plain function (the launch from inside a function template left the host stubs undefined)
static void launch_dma_kernel(int bn, int ks, int mode, hipStream_t stream, const ConvP& p, int bm = 128) {
    const dim3 grid((unsigned)(cr_cdiv(p.M, bm) * (p.Cout / bn) * p.ksplit)), block(CONV_T);
    if (bm == 64) {         // f32, 64 x 128 tiles, two wave groups (callers: try_launch_dma)
#define CR_BM64_CASE(K, M_) if (ks == K && mode == M_) { \
        hipLaunchKernelGGL((k_conv_igemm_dma<128, K, M_, float, 2, 64>), grid, dim3(CONV_T * 2), 0, stream, p); return; }
        CR_BM64_CASE(3, 0) CR_BM64_CASE(3, 1) CR_BM64_CASE(1, 0) CR_BM64_CASE(1, 1)
#undef CR_BM64_CASE
    }
    if (p.w3) {
#define CR_S3_CASE(B, K, M_) if (bn == B && ks == K && mode == M_) { \
        hipLaunchKernelGGL((k_conv_igemm_dma_s3<B, K, M_>), grid, block, 0, stream, p); return; }
        CR_S3_CASE(128, 3, 0) CR_S3_CASE(128, 3, 1) CR_S3_CASE(128, 1, 0) CR_S3_CASE(128, 1, 1)
        CR_S3_CASE(64, 3, 0) CR_S3_CASE(64, 3, 1) CR_S3_CASE(64, 1, 0) CR_S3_CASE(64, 1, 1)
#undef CR_S3_CASE
    }
    // f32, at most one block per CU: two wave groups per block (see the kernel)
    static const int kg_on = env_int("CR_CONV_KG2", 1);
    const bool kg2 = kg_on && p.f32 && grid.x <= 256 && (p.Kdim / (p.f32 ? 32 : 64)) / p.ksplit >= 4;
#define CR_DMA_CASE(B, K, M_) if (bn == B && ks == K && mode == M_) { \
        if (kg2) hipLaunchKernelGGL((k_conv_igemm_dma<B, K, M_, float, 2>), grid, dim3(CONV_T * 2), 0, stream, p); \
        else if (p.f32) hipLaunchKernelGGL((k_conv_igemm_dma<B, K, M_, float>), grid, block, 0, stream, p); \
        else hipLaunchKernelGGL((k_conv_igemm_dma<B, K, M_, u16>), grid, block, 0, stream, p); \
        return; }
    CR_DMA_CASE(128, 3, 0) CR_DMA_CASE(128, 3, 1) CR_DMA_CASE(128, 1, 0) CR_DMA_CASE(128, 1, 1)
    CR_DMA_CASE(64, 3, 0) CR_DMA_CASE(64, 3, 1) CR_DMA_CASE(64, 1, 0) CR_DMA_CASE(64, 1, 1)
#undef CR_DMA_CASE
}

static int dma_enabled() {
    static const int v = env_int("CR_CONV_DMA", 1);
    return v;
}

// true + launched if the layer takes the LDS-DMA kernel: the layers that would run <128,128> or <128,64> tiles with KU = 1
template <int KS, int MODE>
static bool try_launch_dma(cr_ctx* ctx, const ConvP& p, int out_f32, int* rc) {
    if constexpr (KS == 7) {
        return false;
    } else {
        // a k stage is two 64-B rows: 64 bf16 / 32 f32 of k, and must not straddle a filter tap
        if (p.cls || (!p.f32 && out_f32) || !dma_enabled() || (p.Cin & (p.f32 ? 31 : 63)) != 0 || p.Cout % 64 != 0) return false;
        static const int min_tiles = env_int("CR_CONV_DMA_MIN_TILES", 128), force_bn = env_int("CR_CONV_DMA_BN", 0);
        const int64_t big_tiles = cr_cdiv(p.M, 128) * (p.Cout >= 128 ? p.Cout / 128 : 1);
        static const int splitk_on = env_int("CR_CONV_SPLITK", 1);
        if (p.f32 && splitk_on && big_tiles < 192 && ctx->ws && (p.Kdim >= 2048 || (KS == 3 && p.Kdim >= 1152))) {
            // f32 mode, few big tiles: the small-tile kernels are L2-bandwidth bound there (a 64 x 32 tile moves 0.094 B
            // per flop = 14.7 TB/s at the f32 MFMA peak, a 128 x 128 tile 0.031), so keep 128-wide tiles and split K over
            // workgroups until every CU has one; partial sums go to the ctx workspace, k_splitk_epilogue finishes
            const int bn2 = p.Cout % 128 == 0 ? 128 : 64;
            int64_t tiles = cr_cdiv(p.M, 128) * (p.Cout / bn2);
            const int nstage_all = p.Kdim / 32;
            // half-height tiles first: twice the blocks before any k split (no slab traffic, often no epilogue launch)
            static const int bm64_on = env_int("CR_CONV_BM64", 1);
            // very long k with >= 128 big tiles (the box head's first FC layer: 2048 x 12544 -> 1024): 128 x 128 tiles, split k
            // until there are two blocks per CU (scripts/fc_bench.py: 509 -> 450 us; the slabs are 1 % of the operand reads)
            const bool long_k = KS == 1 && p.Kdim >= 8192 && tiles >= 128;
            const int bm = (bm64_on && !p.w3 && bn2 == 128 && p.M >= 128 && !long_k) ? 64 : 128;
            if (bm == 64) tiles = cr_cdiv(p.M, 64) * (p.Cout / 128);
            static const int sk_target = env_int("CR_SPLITK_TARGET", 256);
            int S = (int)cr_cdiv(long_k ? 2 * sk_target : sk_target, tiles);
            if (S > nstage_all / 4) S = nstage_all / 4;                      // >= 4 stages (128 of k) per block
            if (S > 16) S = 16;
            const int64_t cap = (int64_t)(ctx->ws_bytes / ((size_t)p.M * p.Cout * sizeof(float)));
            if (S > cap) S = (int)cap;
            if (S < 2 && bm == 64) {
                launch_dma_kernel(128, KS, MODE, ctx->stream, p, 64);
                hipError_t e3 = hipGetLastError();
                if (e3 != hipSuccess) { cr_set_error("k_conv_igemm_dma launch failed: %s", hipGetErrorString(e3)); *rc = CR_EHIP; }
                else *rc = CR_OK;
                return true;
            }
            if (S >= 2) {
                ConvP q = p;
                q.kstages = (int)cr_cdiv(nstage_all, S);
                q.ksplit = (int)cr_cdiv(nstage_all, q.kstages);
                q.part = (float*)ctx->ws;
                launch_dma_kernel(bn2, KS, MODE, ctx->stream, q, bm);
                const dim3 g2((unsigned)cr_cdiv(p.M, 64), (unsigned)(p.Cout / 64));
                hipLaunchKernelGGL(k_splitk_epilogue<float>, g2, dim3(256), 0, ctx->stream, q);
                hipError_t e2 = hipGetLastError();
                if (e2 != hipSuccess) { cr_set_error("split-K conv launch failed: %s", hipGetErrorString(e2)); *rc = CR_EHIP; }
                else *rc = CR_OK;
                return true;
            }
        }
        if (big_tiles < min_tiles) return false;             // small grids keep the 64x64 / split-K kernels
        int bn = p.Cout % 128 == 0 ? 128 : 64;
        if (force_bn == 64 || (force_bn == 0 && bn == 128 && big_tiles < 512)) bn = 64;   // more workgroups on mid-size maps
        launch_dma_kernel(bn, KS, MODE, ctx->stream, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cr_set_error("k_conv_igemm_dma launch failed: %s", hipGetErrorString(e)); *rc = CR_EHIP; }
    else *rc = CR_OK;
    return true;
}

// ---------------------------------------------------------------------------
// k_conv_patch_f32: forward (MODE 0) / backward-data (MODE 1) of the stem convolutions in f32 (16 output channels of the
// GEMM, stride 1, "same" padding, 512 x 512 maps: 7x7 on the 4-channel image, 3x3 16 -> 16).  Same idea as
// k_conv_wgrad_patch_f32: the im2col kernels gather every input pixel once per filter tap from L2 (98 / 87 / 79 us for
// 6.6 / 4.8 / 4.8 GFLOP); here a block stages the (16 + KS - 1)^2 input patch of a 16 x 16 pixel tile in LDS once and every
// tap reads its shifted window:  D[co 4g+e][pixel l15] += W[co][tap][c] * X[row + r][l15 + s][c].
//   A = weights, all of them in registers for the whole kernel (lane: row co = l & 15, k = l >> 4):
//       Cin = 16: a float4 per tap = w[co][tap][4g .. 4g+3]; k-step j of a tap multiplies channels 4g + j;
//       Cin = 4:  a float per tap  = w[co][tap][g]; one k-step per tap.
//   B = patch (lane: column pixel l15, k = g): ONE ds_read_b128 per (patch row, s) = the four k-steps' channels 4g .. 4g+3
//       of pixel l15 + s (a wave reads 1 KB contiguous: conflict-free), used by the up to KS output rows it belongs to;
//       Cin = 4: one ds_read_b32 per (patch row, s).
// Wave w owns tile rows 4w .. 4w+3 (four accumulators); a lane ends with four consecutive channels of a pixel: the tile
// row goes out as one 1-KB store per wave.  The next tile's patch is in flight under the MFMAs and lands in the other
// LDS stage.  BN statistics: per-tile sums to statistics row 4 * tile (rows are per 64 pixels), zeros to the next three.
// MODE 1 reads the transposed weights (cr_weight_transpose: [Cin][(r,s)][Cout]) with the taps mirrored, and adds the
// `accumulate` tensor (gradient fan-in) in the epilogue.
// ---------------------------------------------------------------------------
template <int KS, int CIN, int MODE>
__global__ __launch_bounds__(256) void k_conv_patch_f32(ConvP p, int tiles_x, int tiles_per_img, int tiles_total) {
    constexpr int PAD = KS / 2, PW = 16 + KS - 1, TAPS = KS * KS;
    constexpr int C4 = CIN / 4, NXL = PW * PW * C4, NXR = (NXL + 255) / 256;
    constexpr int STAGE = PW * PW * CIN;
    constexpr int NQ = 4 + KS - 1;                                   // patch rows a wave touches
    __shared__ float smem[2 * STAGE];
    __shared__ float sStatAll[2][4 * 2 * 16];                        // per LDS stage: a tile's sums are read after the barrier
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int H = p.Hout, W = p.Wout;
    const float* __restrict__ wsrc = reinterpret_cast<const float*>(p.w);
    // weights -> registers.  Patch offset (r', s') multiplies filter tap (r', s') forward, (KS-1-r', KS-1-s') backward.
    f32x4 w4[CIN == 16 ? TAPS : 1];
    float w1[CIN == 4 ? TAPS : 1];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const int r = t / KS, s2 = t - r * KS;
        const int tap = MODE == 0 ? t : (KS - 1 - r) * KS + (KS - 1 - s2);
        if (CIN == 16) w4[t] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)l15 * p.Kdim + tap * CIN + 4 * g);
        else w1[t] = wsrc[(size_t)l15 * p.Kdim + tap * CIN + g];
    }
    int xpy[NXR], xpx[NXR], xc4[NXR];
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
        const int i = tid + j * 256;
        const int pp = i / C4;
        xc4[j] = i - pp * C4; xpy[j] = pp / PW; xpx[j] = pp - xpy[j] * PW;
    }
    const __amdgpu_buffer_rsrc_t bx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    u32x4 rx[NXR];
    auto load_tile = [&](int tile) {
        const int n = tile / tiles_per_img, tr = tile - n * tiles_per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int y0 = ty * 16, x0 = tx * 16;
#pragma unroll
        for (int j = 0; j < NXR; ++j) {
            const int yy = y0 + xpy[j] - PAD, xx = x0 + xpx[j] - PAD;
            const bool ok = tid + j * 256 < NXL && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            rx[j] = __builtin_amdgcn_raw_buffer_load_b128(bx, ok ? (unsigned)((((n * H + yy) * W + xx) * CIN + xc4[j] * 4) * 4) : OOB, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NXR; ++j)
            if (tid + j * 256 < NXL) *reinterpret_cast<u32x4*>(&smem[buf * STAGE + (tid + j * 256) * 4]) = rx[j];
    };
    const int G = gridDim.x;
    int tile = blockIdx.x, buf = 0;
    load_tile(tile);
    store_tile(0);
    __syncthreads();
    for (; tile < tiles_total; tile += G, buf ^= 1) {
        const bool more = tile + G < tiles_total;
        if (more) load_tile(tile + G);
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // this lane's pixel column in the wave's first patch row; channel chunk g (Cin = 16) / channel g (Cin = 4)
        const float* xb = smem + buf * STAGE + ((wave * 4) * PW + l15) * CIN + (CIN == 16 ? 4 * g : g);
        if constexpr (CIN == 16) {
            f32x4 xv[2][KS];
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) xv[0][s2] = *reinterpret_cast<const f32x4*>(xb + s2 * CIN);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) {
#pragma unroll
                    for (int s2 = 0; s2 < KS; ++s2) xv[(q + 1) & 1][s2] = *reinterpret_cast<const f32x4*>(xb + ((q + 1) * PW + s2) * CIN);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
                    for (int r = 0; r < KS; ++r) {
                        const int row = q - r;                       // output row of this wave fed by patch row q through tap row r
                        if (row >= 0 && row < 4) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[row] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r * KS + s2][j], xv[q & 1][s2][j], acc[row], 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float xv[2][KS];
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) xv[0][s2] = xb[s2 * CIN];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) {
#pragma unroll
                    for (int s2 = 0; s2 < KS; ++s2) xv[(q + 1) & 1][s2] = xb[((q + 1) * PW + s2) * CIN];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
                    for (int r = 0; r < KS; ++r) {
                        const int row = q - r;
                        if (row >= 0 && row < 4)
                            acc[row] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r * KS + s2], xv[q & 1][s2], acc[row], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue: lane = channels 4g .. 4g+3 of pixel l15 of tile rows 4w .. 4w+3
        float* sStat = sStatAll[buf];
        const int n = tile / tiles_per_img, tr = tile - n * tiles_per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            const size_t o = ((size_t)(n * H + ty * 16 + wave * 4 + row) * W + tx * 16 + l15) * 16 + 4 * g;
            f32x4 v = acc[row];
            if (MODE == 1 && p.res != nullptr) {
                const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + o);
                v += rr;
            }
            if (MODE == 0 && p.stats != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + o) = v;
        }
        if (MODE == 0 && p.stats != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    ssum[e] += __shfl_xor(ssum[e], off, 64);
                    ssq[e] += __shfl_xor(ssq[e], off, 64);
                }
            if (l15 == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { sStat[(wave * 2 + 0) * 16 + 4 * g + e] = ssum[e]; sStat[(wave * 2 + 1) * 16 + 4 * g + e] = ssq[e]; }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();                                             // stage buf^1 complete, stage buf free, sStat visible
        if (MODE == 0 && p.stats != nullptr) {
            float* dst = p.stats + (size_t)(4 * tile) * 2 * 16;      // statistics rows are per 64 pixels: 4 per tile
            if (tid < 32) {
                const int which = tid >> 4, c = tid & 15;
                dst[which * 16 + c] = ((sStat[(0 * 2 + which) * 16 + c] + sStat[(1 * 2 + which) * 16 + c]) + sStat[(2 * 2 + which) * 16 + c]) + sStat[(3 * 2 + which) * 16 + c];
            } else if (tid < 128) {
                dst[tid] = 0.f;                                      // rows 4 tile + 1 .. + 3
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_conv_patch_bwd_s2_f32: backward-data of the stem's stride-2 3x3 convolution 16 -> 32 (dx 512 x 512 x 16 from dy
// 256 x 256 x 32) in f32.  The im2col kernel by output-parity class (k_conv_igemm<128,16,3,1>, cls) takes 78 us for 2.4 GFLOP:
// 16-wide tiles and a gather per tap.  Here a block owns a 16 x 16 tile of dx; the 9 x 9 x 32 patch of dy it depends on is
// staged in LDS once, as eight 4-channel planes ([chunk][row][col][4]: sixteen pixel lanes of one chunk read consecutive
// 16-B items).  A dx pixel (y0 + 2a + ph, x0 + 2b + pw) only receives the taps r = ph + 1 (mod 2), s likewise: 1 / 2 / 2 / 4
// taps for the four parity classes, dy pixel (a + dr, b + ds) with dr = 1 for r = 0, else 0.  MFMA column = 16 pixels of
// one class (two a-rows x eight b); wave w takes a-rows 2w, 2w+1 of EVERY class (four accumulators, 72 MFMAs: balanced).
//   A = transposed weights wt[ci][tap][co] in registers (lane: row ci = l & 15, k = l >> 4: co = 4g + j and 16 + 4g + j),
//   B = dy plane chunk g / 4 + g (two ds_read_b128 per tap).  A lane ends with 4 consecutive channels of a dx pixel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_patch_bwd_s2_f32(ConvP p, int tiles_x, int tiles_per_img, int tiles_total) {
    constexpr int PR = 9, PLANE = PR * PR * 4, STAGE = 8 * PLANE;   // floats
    constexpr int NXL = PR * PR * 8, NXR = (NXL + 255) / 256;       // 16-B items of a patch / per thread
    __shared__ float smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int H = p.Hout, W = p.Wout, Hd = p.Hin, Wd = p.Win;       // dx / dy extents
    const float* __restrict__ wsrc = reinterpret_cast<const float*>(p.w);
    f32x4 wlo[9], whi[9];                                           // wt[ci = l15][tap][4g ..] and [16 + 4g ..]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wlo[t] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)l15 * p.Kdim + t * 32 + 4 * g);
        whi[t] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)l15 * p.Kdim + t * 32 + 16 + 4 * g);
    }
    int xrow[NXR], xcol[NXR], xch[NXR];
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
        const int i = tid + j * 256;                                 // item = (pixel, chunk), chunk fastest in global memory
        const int pp = i >> 3;
        xch[j] = i & 7; xrow[j] = pp / PR; xcol[j] = pp - xrow[j] * PR;
    }
    const __amdgpu_buffer_rsrc_t bx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    u32x4 rx[NXR];
    auto load_tile = [&](int tile) {
        const int n = tile / tiles_per_img, tr = tile - n * tiles_per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
#pragma unroll
        for (int j = 0; j < NXR; ++j) {
            const int yy = ty * 8 + xrow[j], xx = tx * 8 + xcol[j];
            const bool ok = tid + j * 256 < NXL && yy < Hd && xx < Wd;
            rx[j] = __builtin_amdgcn_raw_buffer_load_b128(bx, ok ? (unsigned)((((n * Hd + yy) * Wd + xx) * 32 + xch[j] * 4) * 4) : OOB, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NXR; ++j)
            if (tid + j * 256 < NXL)
                *reinterpret_cast<u32x4*>(&smem[buf * STAGE + xch[j] * PLANE + (xrow[j] * PR + xcol[j]) * 4]) = rx[j];
    };
    const int G = gridDim.x;
    int tile = blockIdx.x, buf = 0;
    load_tile(tile);
    store_tile(0);
    __syncthreads();
    const int a = 2 * wave + (l15 >> 3), b = l15 & 7;                // this lane's pixel pair index inside the tile
    for (; tile < tiles_total; tile += G, buf ^= 1) {
        const bool more = tile + G < tiles_total;
        if (more) load_tile(tile + G);
        // planes g (channels 4g ..) and 4 + g (16 + 4g ..) at dy pixel (a, b) of the patch
        const float* plo = smem + buf * STAGE + g * PLANE + (a * PR + b) * 4;
        const float* phi = plo + 4 * PLANE;
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the four dy pixels a lane ever needs: (a + dr, b + ds), dr, ds in {0, 1}
        f32x4 dlo[2][2], dhi[2][2];
#pragma unroll
        for (int dr = 0; dr < 2; ++dr)
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
                dlo[dr][ds] = *reinterpret_cast<const f32x4*>(plo + (dr * PR + ds) * 4);
                dhi[dr][ds] = *reinterpret_cast<const f32x4*>(phi + (dr * PR + ds) * 4);
            }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2) {
                        // dx row 2a + ph takes tap r iff (2a + ph + 1 - r) is even: r = ph + 1 (mod 2); dy row a + (ph + 1 - r) / 2
                        if (((ph + 1 - r) & 1) == 0 && ((pw + 1 - s2) & 1) == 0) {
                            const int dr = (ph + 1 - r) / 2, ds = (pw + 1 - s2) / 2;      // 0 or 1 (r = 0 -> 1 for ph = 1)
                            const int t = r * 3 + s2, c = ph * 2 + pw;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wlo[t][j], dlo[dr][ds][j], acc[c], 0, 0, 0);
                                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(whi[t][j], dhi[dr][ds][j], acc[c], 0, 0, 0);
                            }
                        }
                    }
        const int n = tile / tiles_per_img, tr = tile - n * tiles_per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int pw = 0; pw < 2; ++pw) {
                const size_t o = ((size_t)(n * H + ty * 16 + 2 * a + ph) * W + tx * 16 + 2 * b + pw) * 16 + 4 * g;
                f32x4 v = acc[ph * 2 + pw];
                if (p.res != nullptr) v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + o);
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + o) = v;
            }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }
}

// true + launched if the layer takes the patch kernel
template <int KS, int MODE>
static bool try_launch_conv_patch_f32(cr_ctx* ctx, const ConvP& p, int* rc) {
    if constexpr (KS == 1 || (KS == 7 && MODE == 1)) {
        return false;
    } else {
        static const int on = env_int("CR_CONV_PATCH", 1);
        const int cin_ok = KS == 3 ? 16 : 4;
        if (!on || !p.f32 || p.w3 || p.cls || p.Cout != 16 || p.Cin != cin_ok || p.stride != 1 || p.pad != KS / 2 ||
            p.Hin != p.Hout || p.Win != p.Wout || (p.Hout & 15) || (p.Wout & 15) || p.bias || p.relu || p.ksplit != 1 ||
            (MODE == 0 && p.res) || (MODE == 1 && p.stats))
            return false;
        const int tiles_x = p.Wout / 16, tiles_per_img = tiles_x * (p.Hout / 16), tiles_total = tiles_per_img * p.N;
        static const int per_cu = env_int("CR_CONV_PATCH_BLOCKS", 2);
        const dim3 grid((unsigned)std::min(tiles_total, 256 * per_cu));
        if (KS == 3) hipLaunchKernelGGL((k_conv_patch_f32<3, 16, MODE>), grid, dim3(256), 0, ctx->stream, p, tiles_x, tiles_per_img, tiles_total);
        else hipLaunchKernelGGL((k_conv_patch_f32<7, 4, 0>), grid, dim3(256), 0, ctx->stream, p, tiles_x, tiles_per_img, tiles_total);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { cr_set_error("k_conv_patch_f32 launch failed: %s", hipGetErrorString(e)); *rc = CR_EHIP; }
        else *rc = CR_OK;
        return true;
    }
}

template <int KS, int MODE>
static int launch_igemm_ks(cr_ctx* ctx, const ConvP& p, int out_f32) {
    int rc_dma = CR_OK;
    if (try_launch_conv_patch_f32<KS, MODE>(ctx, p, &rc_dma)) return rc_dma;
    if (try_launch_dma<KS, MODE>(ctx, p, out_f32, &rc_dma)) return rc_dma;
    if (p.f32) {
        // f32 MFMA runs at 1/16 of the bf16 rate: the kernels are MFMA-bound, one wave per SIMD with >= 2 independent
        // accumulators already issues at the full rate, so the tile is chosen to put a block on every CU (intra-block
        // split-K adds waves, not CUs, and is not used); two sub-steps (32 of k) per stage cover the global round trip
        const int64_t b128 = p.Cout % 128 == 0 ? cr_cdiv(p.M, 128) * (p.Cout / 128) : 0;
        const int64_t b64 = p.Cout % 64 == 0 ? cr_cdiv(p.M, 64) * (p.Cout / 64) : 0;
        if (b128 >= 256) return launch_igemm_f32<128, 128, KS, MODE, 2>(ctx, p);
        if (b64 >= 192) return launch_igemm_f32<64, 64, KS, MODE, 2>(ctx, p);
        if (p.Cout % 32 == 0) {
            if (cr_cdiv(p.M, 128) * (p.Cout / 32) >= 512) return launch_igemm_f32<128, 32, KS, MODE, 2>(ctx, p);
            return launch_igemm_f32<64, 32, KS, MODE, 2>(ctx, p);
        }
        if (cr_cdiv(p.M, 128) * (p.Cout / 16) >= 512) return launch_igemm_f32<128, 16, KS, MODE, 2>(ctx, p);
        return launch_igemm_f32<64, 16, KS, MODE, 2>(ctx, p);
    }
    // the 32x32 ... 8x8 levels at 4 images/GPU give only 8-64 tiles of 128x128: use 64x64 tiles there so that the
    // launch covers the 256 CUs (MI355X: "a launch needs >> 256 workgroups")
    const int64_t big_tiles = cr_cdiv(p.M, 128) * (p.Cout >= 128 ? p.Cout / 128 : 1);
    static const int ku_small = env_int("CR_IGEMM_KU_SMALL", 4), ku_big = env_int("CR_IGEMM_KU_BIG", 1);
    static const int ku_tiny_blocks = env_int("CR_IGEMM_KU8_BLOCKS", 320);
    if (p.Cout % 64 == 0 && big_tiles < 512) {
        // <= ~1 block per CU: 16 waves per block on interleaved k (intra-block split-K)
        if (cr_cdiv(p.M, 64) * (p.Cout / 64) <= ku_tiny_blocks && p.Kdim >= 512 && (KS == 1 || (p.Cin & 31) == 0))
            return launch_igemm_t<64, 64, KS, MODE, 2, 4>(ctx, p, out_f32);
        if (ku_small == 4) return launch_igemm_t<64, 64, KS, MODE, 4>(ctx, p, out_f32);
        if (ku_small == 2) return launch_igemm_t<64, 64, KS, MODE, 2>(ctx, p, out_f32);
        return launch_igemm_t<64, 64, KS, MODE>(ctx, p, out_f32);
    }
    if (p.Cout % 128 == 0) {
        if (ku_big == 2) return launch_igemm_t<128, 128, KS, MODE, 2>(ctx, p, out_f32);
        return launch_igemm_t<128, 128, KS, MODE>(ctx, p, out_f32);
    }
    if (p.Cout % 64 == 0) return launch_igemm_t<128, 64, KS, MODE>(ctx, p, out_f32);
    if (p.Cout % 32 == 0) return launch_igemm_t<128, 32, KS, MODE>(ctx, p, out_f32);
    return launch_igemm_t<128, 16, KS, MODE>(ctx, p, out_f32);
}

static int ilog2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return ((1 << s) == v) ? s : -1;
}

static int conv_common_checks(const char* who, int N, int H, int W, int Cin, int Cout, int ks, int stride, int pad,
                              int act_f32 = 0) {
    CR_CHECK_ARG(N > 0 && H > 0 && W > 0, "%s: bad input dims", who);
    // a 16-B chunk holds 8 bf16 or 4 f32 channels of one pixel
    CR_CHECK_ARG(act_f32 ? (Cin % 4 == 0 && Cin >= 4) : (Cin % 8 == 0 && Cin >= 8),
                 "%s: Cin=%d must be a multiple of %d (NHWC 16-B chunks)", who, Cin, act_f32 ? 4 : 8);
    CR_CHECK_ARG(Cout % 16 == 0, "%s: Cout=%d must be a multiple of 16", who, Cout);
    CR_CHECK_ARG(ks == 1 || ks == 3 || ks == 7, "%s: kernel size %d not supported (1,3,7)", who, ks);
    CR_CHECK_ARG(stride == 1 || stride == 2, "%s: stride %d not supported (1,2)", who, stride);
    CR_CHECK_ARG(ks == 1 || ilog2_exact(Cin) >= 0, "%s: Cin=%d must be a power of two for %dx%d kernels", who, Cin, ks, ks);
    CR_CHECK_ARG(pad >= 0 && pad <= ks / 2, "%s: pad %d", who, pad);
    const int64_t lim = act_f32 ? (int64_t)0x1fffffff : (int64_t)0x3fffffff;
    CR_CHECK_ARG((int64_t)N * H * W * (int64_t)(Cin > Cout ? Cin : Cout) < lim && (int64_t)Cout * ks * ks * Cin < lim,
                 "%s: tensor too large for 31-bit byte offsets (buffer loads)", who);
    return CR_OK;
}

extern "C" int cr_conv2d_fwd(cr_ctx* ctx, const void* x, const void* w, void* y, int N, int H, int W, int Cin,
                             int Cout, int ks, int stride, int pad, const float* bias, const void* residual,
                             int relu, float* stats, int out_f32, int act_f32, const void* w_split) {
    CR_CHECK_ARG(ctx && x && w && y, "cr_conv2d_fwd: NULL pointer");
    int rc = conv_common_checks("cr_conv2d_fwd", N, H, W, Cin, Cout, ks, stride, pad, act_f32);
    if (rc) return rc;
    const size_t es = act_f32 ? 4 : 2;
    ConvP p;
    p.x = x; p.w = w; p.y = y; p.res = residual; p.bias = bias; p.stats = stats;
    p.N = N; p.Hin = H; p.Win = W; p.Cin = Cin; p.Cout = Cout;
    p.Hout = (H + 2 * pad - ks) / stride + 1;
    p.Wout = (W + 2 * pad - ks) / stride + 1;
    p.stride = stride; p.pad = pad; p.Kdim = ks * ks * Cin; p.M = N * p.Hout * p.Wout;
    p.cshift = ks == 1 ? 0 : ilog2_exact(Cin); p.relu = relu; p.xcd = xcd_enabled(); p.f32 = act_f32 ? 1 : 0;
    p.part = nullptr; p.ksplit = 1; p.kstages = p.Kdim; p.cls = 0; p.Hfull = p.Hout; p.Wfull = p.Wout; p.wstride = p.Kdim;
    p.x_bytes = (unsigned)((size_t)N * H * W * Cin * es); p.w_bytes = (unsigned)((size_t)Cout * p.Kdim * es);
    p.w3 = (act_f32 == 2 && (p.Kdim & 31) == 0) ? w_split : nullptr; p.w3_bytes = (unsigned)((size_t)Cout * p.Kdim * 6);
    if (act_f32) out_f32 = 1;
    if (ks == 1) return launch_igemm_ks<1, 0>(ctx, p, out_f32);
    if (ks == 3) return launch_igemm_ks<3, 0>(ctx, p, out_f32);
    return launch_igemm_ks<7, 0>(ctx, p, out_f32);
}

// dX[n,h,w,c] = sum_{r,s,k} dY[n,(h+pad-r)/stride,(w+pad-s)/stride,k] * W[k,r,s,c]
// wt = weights re-laid as [Cin][(r*KS+s)*Cout + k]  (cr_weight_transpose)
extern "C" int cr_conv2d_bwd_data(cr_ctx* ctx, const void* dy, const void* wt, void* dx, int N, int H, int W,
                                  int Cin, int Cout, int ks, int stride, int pad, int act_f32, const void* wt_split,
                                  const void* accumulate) {
    CR_CHECK_ARG(ctx && dy && wt && dx, "cr_conv2d_bwd_data: NULL pointer");
    int rc = conv_common_checks("cr_conv2d_bwd_data", N, H, W, Cout, Cin, ks, stride, pad, act_f32);
    if (rc) return rc;
    CR_CHECK_ARG(Cin % 16 == 0, "cr_conv2d_bwd_data: Cin=%d must be a multiple of 16", Cin);
    const size_t es = act_f32 ? 4 : 2;
    ConvP p;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    p.x = dy; p.w = wt; p.y = dx; p.res = accumulate; p.bias = nullptr; p.stats = nullptr;   // res: dx = conv^T(dy) + accumulate
    p.N = N; p.Hin = Ho; p.Win = Wo; p.Cin = Cout;      // gather source = dY
    p.Hout = H; p.Wout = W; p.Cout = Cin;               // GEMM output = dX
    p.stride = stride; p.pad = pad; p.Kdim = ks * ks * Cout; p.M = N * H * W;
    p.cshift = ks == 1 ? 0 : ilog2_exact(Cout); p.relu = 0; p.xcd = xcd_enabled(); p.f32 = act_f32 ? 1 : 0;
    p.part = nullptr; p.ksplit = 1; p.kstages = p.Kdim; p.cls = 0; p.Hfull = p.Hout; p.Wfull = p.Wout; p.wstride = p.Kdim;
    p.x_bytes = (unsigned)((size_t)N * Ho * Wo * Cout * es); p.w_bytes = (unsigned)((size_t)Cin * p.Kdim * es);
    p.w3 = (act_f32 == 2 && (p.Kdim & 31) == 0) ? wt_split : nullptr; p.w3_bytes = (unsigned)((size_t)Cin * p.Kdim * 6);
    static const int patch_s2 = env_int("CR_CONV_PATCH", 1);
    if (patch_s2 && act_f32 && stride == 2 && ks == 3 && pad == 1 && Cout == 32 && Cin == 16 && (H & 15) == 0 && (W & 15) == 0 &&
        Ho * 2 == H && Wo * 2 == W) {
        const int tiles_x = W / 16, tiles_per_img = tiles_x * (H / 16), tiles_total = tiles_per_img * N;
        static const int per_cu = env_int("CR_CONV_PATCH_BLOCKS", 2);
        const dim3 grid((unsigned)std::min(tiles_total, 256 * per_cu));
        hipLaunchKernelGGL(k_conv_patch_bwd_s2_f32, grid, dim3(256), 0, ctx->stream, p, tiles_x, tiles_per_img, tiles_total);
        CR_LAUNCH_CHECK();
        return CR_OK;
    }
    static const int cls_on = env_int("CR_BWD_S2_CLASSES", 1);
    if (cls_on && stride == 2 && ks == 3 && (H & 1) == 0 && (W & 1) == 0 && (Cout & (act_f32 ? 15 : 31)) == 0) {
        // by output-pixel parity class (see ConvP): H/2 x W/2 pixels per class, 4 classes in one grid
        p.cls = 1; p.Hfull = H; p.Wfull = W;
        p.Hout = H / 2; p.Wout = W / 2; p.M = N * p.Hout * p.Wout;
    }
    if (ks == 1) return launch_igemm_ks<1, 1>(ctx, p, act_f32 ? 1 : 0);
    if (ks == 3) return launch_igemm_ks<3, 1>(ctx, p, act_f32 ? 1 : 0);
    return launch_igemm_ks<7, 1>(ctx, p, act_f32 ? 1 : 0);
}

// ---- grouped forward / backward-data (k_conv_igemm_dma_grp) ------------------------------------------------------------
static int group_starts(int n, const int* counts, int* starts) {
    int total = 0;
    for (int i = 0; i < n; ++i) { starts[i] = total; total += (counts[i] + 7) & ~7; }
    return total;
}

// Tile quantisation of a grouped launch: T tiles of equal duration over 2 x 256 resident slots run as ceil(T / 512) rounds (the
// pyramid at 4 images: 1 364 tiles = 2.66 rounds, paid as 3).  The problems that do not fit into the full rounds (the tail:
// the smaller levels) are split S ways along k (the split-K mode of the body: partial slabs in the ctx workspace, finished by
// k_splitk_epilogue in a fixed order), so the last round is made of 1/S-duration blocks: 1 024 + 3 x 340 blocks = 2 + 2/3 rounds.
// f32 only (bf16 tiles are 6 x shorter: the epilogue launches would cost more than the partial round).  Returns the number of
// problems that were split; their indices are the trailing ones of `split[]`.
static int group_tail_split(cr_ctx* ctx, ConvGroup& g, int* counts, int n, bool* split) {
    static const int S_env = env_int("CR_GRP_KSPLIT", 3);
    for (int i = 0; i < n; ++i) split[i] = false;
    if (S_env < 2 || !ctx->ws) return 0;
    const int slots = 512;
    int total = 0;
    for (int i = 0; i < n; ++i) total += counts[i];
    const int full = (total / slots) * slots, rem = total - full;
    int bulk = 0, nsplit = 0;
    size_t need = 0;
    int S = S_env;
    if (total * 2 <= slots) {
        // a group that fills less than half of ONE round (the three small FPN levels: 88 tiles on 256 CUs): every problem is
        // split along k, as many ways as still fit the resident slots
        static const int S_small = env_int("CR_GRP_KSPLIT_SMALL", 3);
        S = S_small < slots / total ? S_small : slots / total;
        if (S < 2) return 0;
        for (int i = 0; i < n; ++i) {
            split[i] = true;
            need += (size_t)g.p[i].M * g.p[i].Cout * sizeof(float);
            ++nsplit;
        }
    } else {
        if (full == 0 || rem < slots / 16 || rem > (slots * 7) / 8) return 0;    // nothing to win / the last round is nearly full anyway
        for (int i = 0; i < n; ++i) {                                             // problems arrive largest first
            if (bulk + counts[i] <= full) { bulk += counts[i]; continue; }
            split[i] = true;
            need += (size_t)g.p[i].M * g.p[i].Cout * sizeof(float);
            ++nsplit;
        }
    }
    while (S >= 2 && need * S > ctx->ws_bytes) --S;
    if (S < 2) { for (int i = 0; i < n; ++i) split[i] = false; return 0; }
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (!split[i]) continue;
        ConvP& p = g.p[i];
        const int nstage_all = p.Kdim / 32;
        int s = S;
        if (s > nstage_all / 4) s = nstage_all / 4;                              // >= 4 stages per block
        if (s < 2) { split[i] = false; --nsplit; continue; }
        p.kstages = (int)cr_cdiv(nstage_all, s);
        p.ksplit = (int)cr_cdiv(nstage_all, p.kstages);
        p.part = (float*)((char*)ctx->ws + off);
        off += (size_t)p.M * p.Cout * sizeof(float) * p.ksplit;
        counts[i] *= p.ksplit;
    }
    return nsplit;
}

static void group_tail_epilogues(cr_ctx* ctx, const ConvGroup& g, int n, const bool* split) {
    for (int i = 0; i < n; ++i) {
        if (!split[i]) continue;
        const ConvP& p = g.p[i];
        const dim3 g2((unsigned)cr_cdiv(p.M, 64), (unsigned)(p.Cout / 64));
        hipLaunchKernelGGL(k_splitk_epilogue<float>, g2, dim3(256), 0, ctx->stream, p);
    }
}

// n convolutions (stride 1, k in {1, 3}, Cin % 64 == 0 (f32: % 32), Cout % 128 == 0) in one launch.  Pointer tables are HOST
// arrays of device pointers; biases / residuals (fwd) and accumulates (bwd-data) may be NULL or hold NULL entries.
extern "C" int cr_conv2d_fwd_group(cr_ctx* ctx, int n, const void* const* xs, const void* const* ws, void* const* ys,
                                   const int* Ns, const int* Hs, const int* Ws, int Cin, int Cout, int ks, int pad,
                                   const float* const* biases, const void* const* residuals, int relu, int act_f32) {
    CR_CHECK_ARG(ctx && xs && ws && ys && Ns && Hs && Ws, "cr_conv2d_fwd_group: NULL pointer");
    CR_CHECK_ARG(n >= 1 && n <= CR_MAX_GROUP, "cr_conv2d_fwd_group: 1..%d problems", CR_MAX_GROUP);
    CR_CHECK_ARG((ks == 1 || ks == 3) && act_f32 != 2 && Cout % 128 == 0 && (Cin & (act_f32 ? 31 : 63)) == 0,
                 "cr_conv2d_fwd_group: k in {1,3}, Cout %% 128 == 0, Cin %% 64 (32 in f32) == 0, fp32 or bf16");
    const size_t es = act_f32 ? 4 : 2;
    ConvGroup g;
    int counts[CR_MAX_GROUP];
    g.n = n;
    for (int i = 0; i < n; ++i) {
        CR_CHECK_ARG(xs[i] && ws[i] && ys[i], "cr_conv2d_fwd_group: NULL tensor %d", i);
        int rc = conv_common_checks("cr_conv2d_fwd_group", Ns[i], Hs[i], Ws[i], Cin, Cout, ks, 1, pad, act_f32);
        if (rc) return rc;
        ConvP& p = g.p[i];
        p.x = xs[i]; p.w = ws[i]; p.y = ys[i]; p.res = residuals ? residuals[i] : nullptr; p.bias = biases ? biases[i] : nullptr;
        p.stats = nullptr;
        p.N = Ns[i]; p.Hin = Hs[i]; p.Win = Ws[i]; p.Cin = Cin; p.Cout = Cout;
        p.Hout = Hs[i] + 2 * pad - ks + 1; p.Wout = Ws[i] + 2 * pad - ks + 1;
        p.stride = 1; p.pad = pad; p.Kdim = ks * ks * Cin; p.M = p.N * p.Hout * p.Wout;
        p.cshift = ks == 1 ? 0 : ilog2_exact(Cin); p.relu = relu; p.xcd = xcd_enabled(); p.f32 = act_f32 ? 1 : 0;
        p.part = nullptr; p.ksplit = 1; p.kstages = p.Kdim; p.cls = 0; p.Hfull = p.Hout; p.Wfull = p.Wout; p.wstride = p.Kdim;
        p.x_bytes = (unsigned)((size_t)p.N * p.Hin * p.Win * Cin * es); p.w_bytes = (unsigned)((size_t)Cout * p.Kdim * es);
        p.w3 = nullptr; p.w3_bytes = 0;
        counts[i] = (int)(cr_cdiv(p.M, 128) * (Cout / 128));
    }
    bool split[CR_MAX_GROUP];
    const int nsplit = act_f32 == 1 ? group_tail_split(ctx, g, counts, n, split) : 0;
    const int total = group_starts(n, counts, g.start);
    for (int i = 0; i < n; ++i) g.count[i] = counts[i];
    for (int i = n; i < CR_MAX_GROUP; ++i) { g.start[i] = total; g.count[i] = 0; }
    const dim3 grid((unsigned)total), block(CONV_T);
#define CR_GRP_CASE(K) if (ks == K) { \
        if (act_f32) hipLaunchKernelGGL((k_conv_igemm_dma_grp<K, 0, float>), grid, block, 0, ctx->stream, g); \
        else hipLaunchKernelGGL((k_conv_igemm_dma_grp<K, 0, u16>), grid, block, 0, ctx->stream, g); }
    CR_GRP_CASE(1) CR_GRP_CASE(3)
#undef CR_GRP_CASE
    if (nsplit) group_tail_epilogues(ctx, g, n, split);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// `batches` float32 GEMMs of one shape in ONE launch: y[b] (R,O) = x[b] (R,K) @ w[b] (O,K)^T, operands of batch b at
// base + b * stride elements (the 16 products of a Winograd F(2x2,3x3) convolution: csrc/winograd.hip).  blockIdx.y = batch.
__global__ __launch_bounds__(CONV_T) void k_gemm_batched_f32(ConvP p, long long sx, long long sw, long long sy, int count) {
    const long long b = blockIdx.y;
    p.x = (const float*)p.x + b * sx;
    p.w = (const float*)p.w + b * sw;
    p.y = (float*)p.y + b * sy;
    conv_igemm_dma_body<128, 1, 0, float, 1, 128>(p, (int)blockIdx.x, count);
}

extern "C" int cr_gemm_batched_f32(cr_ctx* ctx, const float* x, const float* w, float* y, int R, int K, int O, int batches,
                                   int64_t stride_x, int64_t stride_w, int64_t stride_y) {
    CR_CHECK_ARG(ctx && x && w && y && R > 0 && batches >= 1 && batches <= 65535, "cr_gemm_batched_f32: bad args");
    CR_CHECK_ARG(O % 128 == 0 && K % 32 == 0, "cr_gemm_batched_f32: O %% 128 == 0, K %% 32 == 0");
    int rc = conv_common_checks("cr_gemm_batched_f32", 1, 1, R, K, O, 1, 1, 0, 1);
    if (rc) return rc;
    ConvP p;
    p.x = x; p.w = w; p.y = y; p.res = nullptr; p.bias = nullptr; p.stats = nullptr;
    p.N = 1; p.Hin = 1; p.Win = R; p.Cin = K; p.Cout = O; p.Hout = 1; p.Wout = R;
    p.stride = 1; p.pad = 0; p.Kdim = K; p.M = R; p.cshift = 0; p.relu = 0; p.xcd = xcd_enabled(); p.f32 = 1;
    p.part = nullptr; p.ksplit = 1; p.kstages = p.Kdim; p.cls = 0; p.Hfull = 1; p.Wfull = R; p.wstride = K;
    p.x_bytes = (unsigned)((size_t)R * K * 4); p.w_bytes = (unsigned)((size_t)O * K * 4);
    p.w3 = nullptr; p.w3_bytes = 0;
    const int count = (int)(cr_cdiv(R, 128) * (O / 128));
    hipLaunchKernelGGL(k_gemm_batched_f32, dim3((unsigned)count, (unsigned)batches), dim3(CONV_T), 0, ctx->stream, p,
                       (long long)stride_x, (long long)stride_w, (long long)stride_y, count);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// dX_i = conv^T(dY_i) (+ accumulate_i).  wts: the [Cin][k*k*Cout] layouts (cr_weight_transpose); Cin % 128 == 0 here.
extern "C" int cr_conv2d_bwd_data_group(cr_ctx* ctx, int n, const void* const* dys, const void* const* wts, void* const* dxs,
                                        const int* Ns, const int* Hs, const int* Ws, int Cin, int Cout, int ks, int pad,
                                        int act_f32, const void* const* accumulates) {
    CR_CHECK_ARG(ctx && dys && wts && dxs && Ns && Hs && Ws, "cr_conv2d_bwd_data_group: NULL pointer");
    CR_CHECK_ARG(n >= 1 && n <= CR_MAX_GROUP, "cr_conv2d_bwd_data_group: 1..%d problems", CR_MAX_GROUP);
    CR_CHECK_ARG((ks == 1 || ks == 3) && act_f32 != 2 && Cin % 128 == 0 && (Cout & (act_f32 ? 31 : 63)) == 0,
                 "cr_conv2d_bwd_data_group: k in {1,3}, Cin %% 128 == 0, Cout %% 64 (32 in f32) == 0, fp32 or bf16");
    const size_t es = act_f32 ? 4 : 2;
    ConvGroup g;
    int counts[CR_MAX_GROUP];
    g.n = n;
    for (int i = 0; i < n; ++i) {
        CR_CHECK_ARG(dys[i] && wts[i] && dxs[i], "cr_conv2d_bwd_data_group: NULL tensor %d", i);
        int rc = conv_common_checks("cr_conv2d_bwd_data_group", Ns[i], Hs[i], Ws[i], Cout, Cin, ks, 1, pad, act_f32);
        if (rc) return rc;
        const int H = Hs[i], W = Ws[i], Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
        ConvP& p = g.p[i];
        p.x = dys[i]; p.w = wts[i]; p.y = dxs[i]; p.res = accumulates ? accumulates[i] : nullptr; p.bias = nullptr; p.stats = nullptr;
        p.N = Ns[i]; p.Hin = Ho; p.Win = Wo; p.Cin = Cout;      // gather source = dY
        p.Hout = H; p.Wout = W; p.Cout = Cin;                   // GEMM output = dX
        p.stride = 1; p.pad = pad; p.Kdim = ks * ks * Cout; p.M = p.N * H * W;
        p.cshift = ks == 1 ? 0 : ilog2_exact(Cout); p.relu = 0; p.xcd = xcd_enabled(); p.f32 = act_f32 ? 1 : 0;
        p.part = nullptr; p.ksplit = 1; p.kstages = p.Kdim; p.cls = 0; p.Hfull = H; p.Wfull = W; p.wstride = p.Kdim;
        p.x_bytes = (unsigned)((size_t)p.N * Ho * Wo * Cout * es); p.w_bytes = (unsigned)((size_t)Cin * p.Kdim * es);
        p.w3 = nullptr; p.w3_bytes = 0;
        counts[i] = (int)(cr_cdiv(p.M, 128) * (Cin / 128));
    }
    bool split[CR_MAX_GROUP];
    const int nsplit = act_f32 == 1 ? group_tail_split(ctx, g, counts, n, split) : 0;
    const int total = group_starts(n, counts, g.start);
    for (int i = 0; i < n; ++i) g.count[i] = counts[i];
    for (int i = n; i < CR_MAX_GROUP; ++i) { g.start[i] = total; g.count[i] = 0; }
    const dim3 grid((unsigned)total), block(CONV_T);
#define CR_GRP_CASE(K) if (ks == K) { \
        if (act_f32) hipLaunchKernelGGL((k_conv_igemm_dma_grp<K, 1, float>), grid, block, 0, ctx->stream, g); \
        else hipLaunchKernelGGL((k_conv_igemm_dma_grp<K, 1, u16>), grid, block, 0, ctx->stream, g); }
    CR_GRP_CASE(1) CR_GRP_CASE(3)
#undef CR_GRP_CASE
    if (nsplit) group_tail_epilogues(ctx, g, n, split);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// backward-weight
// ---------------------------------------------------------------------------
struct WgP {
    const void* dy;  // [M][Cout]  bf16 or f32
    const void* x;   // NHWC
    float* dw;       // [Cout][Kdim] f32, accumulated with atomics
    int N, Hin, Win, Cin, Hout, Wout, Cout, stride, pad, Kdim, M, cshift;
    int steps_per_split;    // 32-pixel steps handled by one split
    int tm, tn, xcd;        // tile counts (the grid is 1-D: tm * tn * splits blocks)
    unsigned x_bytes, dy_bytes;   // extents for the buffer-load descriptors
    float* dbias;           // optional [Cout]: += column sums of dy (bias gradient), accumulated by the first k-tile's blocks
    int dbg;                // tuning only (CR_S3_DBG): 1 = no MFMAs, 2 = no LDS fragment reads either, 4 = no atomics
    int cshift_w;           // log2(Wout) (k_conv_wgrad_s3_row)
    int cshift_hw;          // log2(Hout * Wout) or -1
    // deterministic mode (CR_DETERMINISTIC=1, f32 kernels): pixel split s writes ITS sum of every dW / bias element to
    // slab[s][...] with a plain store (one writer per element and split) and k_wgrad_reduce adds the splits in a fixed order
    float* slab;            // [splits][slab_stride] or NULL (atomics into dw)
    float* bslab;           // [splits][Cout] or NULL
    long slab_stride;
};

// out[i] += sum_s slab[s][i], s ascending: the fixed-order second pass of the deterministic weight gradient
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slab, int splits, long stride, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int s2 = 0; s2 < splits; ++s2) a += slab[(size_t)s2 * stride + i];
    out[i] += a;
}

static bool deterministic_on() {
    static const int v = getenv("CR_DETERMINISTIC") ? atoi(getenv("CR_DETERMINISTIC")) : 0;
    return v != 0;
}

// points p at slabs in the ctx workspace (at byte offset `off`, advanced) for `splits` pixel splits; false = does not fit
static bool wgrad_use_slabs(cr_ctx* ctx, WgP& p, int splits, size_t* off) {
    p.slab = nullptr; p.bslab = nullptr; p.slab_stride = 0;
    if (!deterministic_on() || !ctx->ws) return false;
    const size_t nw = ((size_t)p.Cout * p.Kdim + 3) & ~(size_t)3, nb = p.dbias ? (((size_t)p.Cout + 3) & ~(size_t)3) : 0;
    const size_t need = (size_t)splits * (nw + nb) * sizeof(float);
    if (*off + need > ctx->ws_bytes) return false;
    p.slab = (float*)((char*)ctx->ws + *off);
    p.slab_stride = (long)nw;
    if (nb) p.bslab = p.slab + (size_t)splits * nw;
    *off += need;
    return true;
}

static void wgrad_reduce_slabs(cr_ctx* ctx, const WgP& p, int splits) {
    const long n = (long)p.Cout * p.Kdim;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)cr_cdiv(n, 256)), dim3(256), 0, ctx->stream, p.slab, splits, p.slab_stride, n, p.dw);
    if (p.bslab)
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)cr_cdiv(p.Cout, 256)), dim3(256), 0, ctx->stream, p.bslab, splits,
                           (long)(((size_t)p.Cout + 3) & ~(size_t)3), (long)p.Cout, p.dbias);
}

// transposing LDS read: 16-lane group reads a 4(row) x 16(col) block of 16-bit elements and
// returns it column-major: lane i gets column i, rows 0..3 (cdna_hip_programming.md T10).
__device__ __forceinline__ s16x4 lds_tr16(const u16* ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(ptr));
}

// KG = wave groups per block: the groups walk interleaved 32*KU-pixel stages of the block's pixel range (4 waves per
// SIMD instead of 1 on the small-map layers, where the grid is only ~1 block per CU), sum their accumulators in an
// LDS tile and issue the atomics from there (full 256-B rows per wave-instruction).  Each group keeps its own
// operand buffers; the reduction tile re-uses them after the loop.
template <int TM, int KS, int KU, int KG>
__global__ __launch_bounds__(CONV_T * KG) void k_conv_wgrad(WgP p) {
    constexpr int TN = 128;
    constexpr int WM = (TM == 128) ? 2 : 1, WN = 4 / WM;
    constexpr int WTM = TM / WM, WTN = TN / WN;          // wave tile
    constexpr int TI = WTM / 16, TJ = WTN / 16;
    // LDS images of the 128-wide tiles: 144-element row pitch (row stride = 8 banks) and a 64-column rotation of the
    // rows with bit 3 set, so that the eight rows one 32-lane half of ds_read_b64_tr_b16 touches ({0-3, 8-11} / {4-7,
    // 12-15} of a 16-row slab) land on eight disjoint 8-bank windows (the 136-pitch image measured 33 % of its LDS
    // cycles as bank conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
    constexpr int PP = TM == 128 ? 144 : TM + 8, PQ = TN + 16;
    auto rotQ = [](int row, int col) { return (col + (((row >> 3) & 1) << 6)) & 127; };
    auto rotP = [&](int row, int col) { return TM == 128 ? rotQ(row, col) : col; };
    constexpr int CPR = TM / 8;                          // dy chunks per pixel row
    constexpr int NP = (32 * CPR + CONV_T - 1) / CONV_T; // dy chunks per thread per 32-pixel sub-step
    constexpr int PE = KU * 32 * PP, QE = KU * 32 * PQ;  // operand elements per group
    constexpr int TNP = TN + 4;                          // reduction tile pitch (floats)
    constexpr int OPB = KG * (PE + QE) * 2, REDB = (KG > 1) ? TM * TNP * 4 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[OPB > REDB ? OPB : REDB];
    const int grp = KG == 1 ? 0 : (int)(threadIdx.x >> 8);
    u16* sP = reinterpret_cast<u16*>(smem) + grp * (PE + QE);
    u16* sQ = sP + PE;

    const int tid = threadIdx.x & (CONV_T - 1), lane = tid & 63, wave = tid >> 6;
    // 1-D grid, channel tile fastest, split slowest: all tiles of one split (same pixel range) are neighbours
    const int bid = p.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int bx = bid % p.tm, by = (bid / p.tm) % p.tn, bz = bid / (p.tm * p.tn);
    const int c0 = bx * TM;                // output-channel tile
    const int q0 = by * TN;                // k (r,s,c) tile
    const int step0 = bz * p.steps_per_split;
    const int nsteps_total = (p.M + 31) >> 5;
    const int step1 = min(step0 + p.steps_per_split, nsteps_total);
    if (step0 >= step1) return;            // block-uniform

    // Q-gather bookkeeping: 32 pixels x 16 chunks = 512 chunks per sub-step, 2 per thread; the k-chunk (filter tap and
    // channel) is fixed per thread for the whole kernel, the two pixel rows walk forward and are tracked incrementally
    // as (n, ho, wo) -- no division in the loop.
    const int qc = tid & 15;                   // chunk within the 128-wide k tile
    const int qk = q0 + qc * 8;
    int qr = 0, qs = 0, qch = qk;
    const bool qvalid = qk < p.Kdim;
    if (KS > 1) {
        const int tap = qk >> p.cshift;
        qch = qk & (p.Cin - 1);
        qr = tap / KS;
        qs = tap - qr * KS;
    }
    const int first = step0 + grp * KU;        // first 32-pixel sub-step of this group
    int pn[2], pho[2], pwo[2];
    {
        const int hw = p.Hout * p.Wout;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = first * 32 + (tid >> 4) + 16 * i;
            pn[i] = m / hw;
            const int rem = m - pn[i] * hw;
            pho[i] = rem / p.Wout;
            pwo[i] = rem - pho[i] * p.Wout;
        }
    }
    int mrow = first * 32;
    const int mend = min(p.M, step1 * 32);                         // rows of later splits / past M contribute zeros
    uint4 rq[KU][2], rp[KU][NP];

    auto advance = [&](int pixels) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pwo[i] += pixels;
            while (pwo[i] >= p.Wout) { pwo[i] -= p.Wout; ++pho[i]; }
            while (pho[i] >= p.Hout) { pho[i] -= p.Hout; ++pn[i]; }
        }
        mrow += pixels;
    };
    // branch-free gathers like the forward kernel: raw buffer loads, masked elements read zero through an
    // out-of-range 32-bit offset
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    auto load_stage = [&]() {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hi = pho[i] * p.stride - p.pad + qr, wi = pwo[i] * p.stride - p.pad + qs;
            const bool ok = qvalid && pn[i] < p.N && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
            const unsigned off = (unsigned)(((pn[i] * p.Hin + hi) * p.Win + wi) * p.Cin + qch) * 2u;
            rq[u][i] = buf_load16(rx, ok ? off : OOB);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + CONV_T * i;
            const int prow = idx / CPR, pc = idx - prow * CPR;
            const bool ok = idx < 32 * CPR && mrow + prow < mend && c0 + pc * 8 < p.Cout;
            const unsigned off = (unsigned)((mrow + prow) * p.Cout + c0 + pc * 8) * 2u;
            rp[u][i] = buf_load16(rdy, ok ? off : OOB);
        }
        advance(32);
      }
      if (KG > 1) advance(32 * KU * (KG - 1));           // the other groups' stages
    };
    auto store_stage = [&]() {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<uint4*>(&sQ[u * 32 * PQ + ((tid >> 4) + 16 * i) * PQ + rotQ((tid >> 4) + 16 * i, qc * 8)]) = rq[u][i];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + CONV_T * i;
            const int prow = idx / CPR, pc = idx - prow * CPR;
            if (idx < 32 * CPR) *reinterpret_cast<uint4*>(&sP[u * 32 * PP + prow * PP + rotP(prow, pc * 8)]) = rp[u][i];
        }
      }
    };

    const int wm = (WM == 2) ? (wave >> 1) : 0, wn = (WM == 2) ? (wave & 1) : wave;
    const int moff = wm * WTM, noff = wn * WTN;
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing: 16-lane group g covers pixel rows 8g..8g+7; lane 4q+pp supplies row q, cols 4pp..
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    // dy rows at or past `mend` load zeros (and the x gather never reads past the batch), so stages past the end of the
    // range (k tail, idle groups) multiply zeros; every group runs the same number of stages (shared barriers)
    const int nstage = (step1 - step0 + KU * KG - 1) / (KU * KG);
    load_stage();
    store_stage();
    __syncthreads();
    // bias gradient = column sums of dy: the blocks of the first k-tile add them up from the dy tile they stage anyway
    const bool do_bias = p.dbias != nullptr && by == 0 && tid < TM;
    float bsum = 0.f;
    for (int st = 0; st < nstage; ++st) {
        if (st + 1 < nstage) load_stage();
        if (do_bias) {
#pragma unroll 8
            for (int r = 0; r < KU * 32; ++r) bsum += bf2f(sP[r * PP + rotP(r & 31, tid)]);
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            bf16x8 af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const u16* b0 = &sP[u * 32 * PP + (8 * g + tq) * PP + rotP(8 * g + tq, moff + i * 16 + 4 * tp)];
                const s16x4 lo = lds_tr16(b0), hi = lds_tr16(b0 + 4 * PP);
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const u16* b0 = &sQ[u * 32 * PQ + (8 * g + tq) * PQ + rotQ(8 * g + tq, noff + j * 16 + 4 * tp)];
                const s16x4 lo = lds_tr16(b0), hi = lds_tr16(b0 + 4 * PQ);
                bfr[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (st + 1 < nstage) {
            store_stage();
            __syncthreads();
        }
    }
    if (do_bias && c0 + tid < p.Cout) atomicAdd(&p.dbias[c0 + tid], bsum);
    // D: col (lane&15) = k index, row 4(lane>>4)+reg = channel
    if (KG == 1) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int kk = q0 + noff + j * 16 + li;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ch = c0 + moff + i * 16 + 4 * g + e;
                    if (kk < p.Kdim && ch < p.Cout) atomicAdd(&p.dw[(size_t)ch * p.Kdim + kk], acc[i][j][e]);
                }
            }
    } else {
        float* tile = reinterpret_cast<float*>(smem);     // operands are dead: the loop ended with a barrier
#pragma unroll
        for (int gg = 0; gg < KG; ++gg) {
            if (grp == gg) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float* d = &tile[(moff + i * 16 + 4 * g + e) * TNP + noff + j * 16 + li];
                            if (gg == 0) *d = acc[i][j][e]; else *d += acc[i][j][e];
                        }
            }
            __syncthreads();
        }
        for (int idx = threadIdx.x; idx < TM * TN; idx += CONV_T * KG) {
            const int chl = idx / TN, kl = idx - chl * TN;
            const int ch = c0 + chl, kk = q0 + kl;
            if (kk < p.Kdim && ch < p.Cout) atomicAdd(&p.dw[(size_t)ch * p.Kdim + kk], tile[chl * TNP + kl]);
        }
    }
}

// ---------------------------------------------------------------------------
// backward-weight in f32 (v_mfma_f32_16x16x4_f32): dW[ch][k] += sum_pixels dY[pixel][ch] * X_gather[pixel][k].
// Both operands arrive pixel-major ([pixel][channel] rows); the MFMA wants lane l to hold (row l&15, pixel l>>4), which
// is a plain ds_read_b32 at [pixel 4t + (l>>4)][col0 + (l&15)]: the two 16-lane halves of a 32-lane LDS group read two
// different pixel rows, so the row pitch is chosen = 16 (mod 32) banks and the read is conflict-free.  A sub-step is 16
// pixels (4 MFMA k-steps); KU = 2 sub-steps per stage.  MFMA-bound (1/16 of the bf16 rate): the f32 atomics of the
// split-over-pixels scheme are hidden behind 16x more matrix time than in the bf16 kernel.
// ---------------------------------------------------------------------------
template <int TM, int KS>
__device__ __forceinline__ void conv_wgrad_f32_body(const WgP& p, const int blk_id, const int blk_count) {
    constexpr int TN = 128, KU = 2, PS = 16;             // k tile, sub-steps per stage, pixels per sub-step
    constexpr int WM = (TM == 128) ? 2 : 1, WN = 4 / WM;
    constexpr int WTM = TM / WM, WTN = TN / WN;          // wave tile
    constexpr int TI = WTM / 16, TJ = WTN / 16;
    constexpr int PP = ((TM + 16) % 32 == 16) ? TM + 16 : TM + 32, PQ = TN + 16;     // row pitches (floats), = 16 mod 32
    constexpr int CPR = TM / 4;                          // dy chunks (4 floats) per pixel row
    constexpr int NP = (PS * CPR + CONV_T - 1) / CONV_T; // dy chunks per thread per sub-step
    __shared__ __attribute__((aligned(16))) float smem[KU * PS * (PP + PQ)];
    float* sP = smem;
    float* sQ = smem + KU * PS * PP;
    const float* __restrict__ xg = reinterpret_cast<const float*>(p.x);
    (void)xg;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = p.xcd ? xcd_swizzle(blk_id, blk_count) : blk_id;
    const int bx = bid % p.tm, by = (bid / p.tm) % p.tn, bz = bid / (p.tm * p.tn);
    const int c0 = bx * TM, q0 = by * TN;
    const int step0 = bz * p.steps_per_split;            // steps are PS pixels here
    const int nsteps_total = (p.M + PS - 1) / PS;
    const int step1 = min(step0 + p.steps_per_split, nsteps_total);
    if (step0 >= step1) return;                          // block-uniform

    // Q gather: 16 pixels x 32 chunks of 4 floats = 512 chunks per sub-step, 2 per thread; the k chunk (filter tap,
    // channel) is fixed per thread, the two pixel rows walk forward incrementally
    const int qc = tid & 31;
    const int qk = q0 + qc * 4;
    int qr = 0, qs = 0, qch = qk;
    const bool qvalid = qk < p.Kdim;
    if (KS > 1) {
        const int tap = qk >> p.cshift;
        qch = qk & (p.Cin - 1);
        qr = tap / KS;
        qs = tap - qr * KS;
    }
    int pn[2], pho[2], pwo[2];
    {
        const int hw = p.Hout * p.Wout;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = step0 * PS + (tid >> 5) + 8 * i;
            pn[i] = m / hw;
            const int rem = m - pn[i] * hw;
            pho[i] = rem / p.Wout;
            pwo[i] = rem - pho[i] * p.Wout;
        }
    }
    int mrow = step0 * PS;
    const int mend = min(p.M, step1 * PS);
    uint4 rq[KU][2], rp[KU][NP];
    auto advance = [&](int pixels) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pwo[i] += pixels;
            while (pwo[i] >= p.Wout) { pwo[i] -= p.Wout; ++pho[i]; }
            while (pho[i] >= p.Hout) { pho[i] -= p.Hout; ++pn[i]; }
        }
        mrow += pixels;
    };
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    auto load_stage = [&]() {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hi = pho[i] * p.stride - p.pad + qr, wi = pwo[i] * p.stride - p.pad + qs;
            const bool ok = qvalid && pn[i] < p.N && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
            const unsigned off = (unsigned)(((pn[i] * p.Hin + hi) * p.Win + wi) * p.Cin + qch) * 4u;
            rq[u][i] = buf_load16(rx, ok ? off : OOB);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + CONV_T * i;
            const int prow = idx / CPR, pc = idx - prow * CPR;
            const bool ok = idx < PS * CPR && mrow + prow < mend && c0 + pc * 4 < p.Cout;
            const unsigned off = (unsigned)((mrow + prow) * p.Cout + c0 + pc * 4) * 4u;
            rp[u][i] = buf_load16(rdy, ok ? off : OOB);
        }
        advance(PS);
      }
    };
    auto store_stage = [&]() {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<uint4*>(&sQ[(u * PS + (tid >> 5) + 8 * i) * PQ + qc * 4]) = rq[u][i];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + CONV_T * i;
            const int prow = idx / CPR, pc = idx - prow * CPR;
            if (idx < PS * CPR) *reinterpret_cast<uint4*>(&sP[(u * PS + prow) * PP + pc * 4]) = rp[u][i];
        }
      }
    };

    const int wm = (WM == 2) ? (wave >> 1) : 0, wn = (WM == 2) ? (wave & 1) : wave;
    const int moff = wm * WTM, noff = wn * WTN;
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, li = lane & 15;
    const int nstage = (step1 - step0 + KU - 1) / KU;
    load_stage();
    store_stage();
    __syncthreads();
    const bool do_bias = p.dbias != nullptr && by == 0 && tid < TM;
    float bsum = 0.f;
    for (int st = 0; st < nstage; ++st) {
        if (st + 1 < nstage) load_stage();
        if (do_bias) {
#pragma unroll 8
            for (int r = 0; r < KU * PS; ++r) bsum += sP[r * PP + tid];
        }
#pragma unroll
        for (int t = 0; t < KU * PS / 4; ++t) {          // MFMA k-steps of 4 pixels
            float af[TI], bq[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = sP[(4 * t + g) * PP + moff + i * 16 + li];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bq[j] = sQ[(4 * t + g) * PQ + noff + j * 16 + li];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bq[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (st + 1 < nstage) {
            store_stage();
            __syncthreads();
        }
    }
    float* const dwo = p.slab ? p.slab + (size_t)bz * p.slab_stride : p.dw;
    if (do_bias && c0 + tid < p.Cout) {
        if (p.bslab) p.bslab[(size_t)bz * (((size_t)p.Cout + 3) & ~(size_t)3) + c0 + tid] = bsum;
        else atomicAdd(&p.dbias[c0 + tid], bsum);
    }
    // D: col (lane&15) = k index, row 4(lane>>4)+reg = channel
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int kk = q0 + noff + j * 16 + li;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = c0 + moff + i * 16 + 4 * g + e;
                if (kk < p.Kdim && ch < p.Cout) {
                    if (p.slab) dwo[(size_t)ch * p.Kdim + kk] = acc[i][j][e];
                    else atomicAdd(&dwo[(size_t)ch * p.Kdim + kk], acc[i][j][e]);
                }
            }
        }
}

template <int TM, int KS>
__global__ __launch_bounds__(CONV_T) void k_conv_wgrad_f32(WgP p) {
    conv_wgrad_f32_body<TM, KS>(p, (int)blockIdx.x, (int)gridDim.x);
}

// grouped launch (see k_conv_igemm_dma_grp): the weight gradients of up to CR_MAX_GROUP convolutions of one geometry class in
// one grid; problems that share their weights (the RPN head over the five pyramid levels) name the same dW and their blocks
// add into it like the pixel splits of one problem do
struct WgGroup { WgP p[CR_MAX_GROUP]; int start[CR_MAX_GROUP], count[CR_MAX_GROUP]; int n; };

template <int KS>
__global__ __launch_bounds__(CONV_T) void k_conv_wgrad_f32_grp(WgGroup g) {
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.start[i + 1]) ++i;
    const int blk = (int)blockIdx.x - g.start[i];
    if (blk >= g.count[i]) return;
    conv_wgrad_f32_body<128, KS>(g.p[i], blk, g.count[i]);
}

// ---------------------------------------------------------------------------
// k_conv_wgrad_patch_f32: weight gradient of the stem convolutions (16 output channels, 512 x 512 maps: 7x7 on the 4-channel
// image and 3x3 on 16 channels, stride 1).  In the im2col formulation (k_conv_wgrad_f32) these layers read every input pixel
// once per filter tap -- 600 / 820 MB of L2 -> LDS traffic for 4.8 / 6.6 GFLOP, twice the matrix time -- and their 144 / 196 k
// columns fill the 128-column k tiles to 56 / 77 %.  Here a block stages a 16 x 16 pixel tile of dy and the (16 + KS - 1)^2
// input PATCH around it once, and every tap reads its shifted window from LDS:
//   D[co][col] += sum_pixels dy[pixel][co] * x[pixel + tap][ci],  v_mfma_f32_16x16x4_f32 with k = 4 consecutive pixels of a
//   tile row; A = dy (lane: row co = l & 15, pixel l >> 4), B = patch (lane: column l & 15, pixel l >> 4).
//   Cin = 16 (3x3): a column is an input channel, one accumulator tile per tap (9).
//   Cin = 4 (7x7):  a column is (tap-in-group q = col >> 2, channel col & 3): four taps per tile, 13 tiles for 49 taps.
// Two blocks per CU (the blocks start and finish together, so prologues / epilogues of more, shorter blocks only add up);
// block b walks tiles b, b + grid, ...: the next tile's loads are in flight under this tile's 144 / 208 MFMAs per wave and go
// to the other LDS stage afterwards (one barrier per tile).  The accumulators stay in registers; at the end the four waves
// are summed through LDS and added to dW in memory order with one set of atomics (and to dbias: column sums of the dy tiles).
// Measured (4 x 512 x 512): 56 us (3x3) / 70 us (7x7) against 144 / 146 us of the im2col kernel; the MFMAs alone are 35 / 51.
// ---------------------------------------------------------------------------
template <int KS, int CIN>
__global__ __launch_bounds__(256) void k_conv_wgrad_patch_f32(WgP p, int tiles_x, int tiles_per_img, int tiles_total) {
    constexpr int PAD = KS / 2, PW = 16 + KS - 1;                    // patch width (pixels)
    constexpr int TAPS = KS * KS;
    constexpr int NT = CIN == 16 ? TAPS : (TAPS + 3) / 4;            // accumulator tiles
    constexpr int C4 = CIN / 4, NXL = PW * PW * C4;                  // float4 loads of a patch
    constexpr int NXR = (NXL + 255) / 256;                           // ... per thread
    constexpr int SX = PW * PW * CIN;
    constexpr int STAGE = 256 * 16 + SX;                             // one tile: dy + patch; two stages (the next tile is
    constexpr int SMEM = 2 * STAGE > 4 * NT * 256 ? 2 * STAGE : 4 * NT * 256;   // stored while this one is multiplied)
    __shared__ float smem[SMEM];
    float* sDy = smem;                                               // [pixel][co]
    float* sX = smem + 256 * 16;                                     // [py][px][ci]
    float* sRed = smem;                                              // after the last tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int H = p.Hout, W = p.Wout;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                                // bias gradient: thread (q = tid >> 4, co = tid & 15)
    // this lane's B column -> (tap offset inside the tile's tap group, channel)
    const int bq = CIN == 16 ? 0 : (l15 >> 2), bc = CIN == 16 ? l15 : (l15 & 3);
    const float* ap0 = sDy + ((wave * 4) * 16 + g) * 16 + l15;
    const float* bp0[NT];                                            // (the 7x7's last tile has one tap: the other columns
#pragma unroll                                                       //  read a valid address and are never stored)
    for (int t = 0; t < NT; ++t) {
        const int tap = CIN == 16 ? t : min(t * 4 + bq, TAPS - 1);
        const int r = tap / KS, s2 = tap - r * KS;
        bp0[t] = sX + ((wave * 4 + r) * PW + g + s2) * CIN + bc;
    }
    // per-thread load slots (the same for every tile): dy pixel / chunk, patch pixel / chunk
    int xpy[NXR], xpx[NXR], xc4[NXR];
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
        const int i = tid + j * 256;
        const int pp = i / C4;
        xc4[j] = i - pp * C4; xpy[j] = pp / PW; xpx[j] = pp - xpy[j] * PW;
    }
    // branch-free loads (out-of-image patch pixels read zero through an out-of-range offset); a tile's loads stay in
    // registers while the previous tile is multiplied
    const __amdgpu_buffer_rsrc_t bx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    // the loads of tile i + 1 are issued when tile i starts and stored to the free LDS stage after its MFMAs (a second
    // register set fetching two tiles ahead was slower: 57.6 / 83 us against 56 / 70)
    u32x4 rdyA[4], rxA[NXR];
    auto load_tile = [&](int tile, u32x4 (&rdy)[4], u32x4 (&rx)[NXR]) {
        // unconditional (a tile past the end fetches nothing: out-of-range offsets), so that the number of loads in flight
        // is known at compile time and the waits before the LDS stores leave the newer set in flight
        const bool live = tile < tiles_total;
        const int n = tile / tiles_per_img, tr = tile - n * tiles_per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int y0 = ty * 16, x0 = tx * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + j * 256, pix = i >> 2, c4 = i & 3;
            rdy[j] = __builtin_amdgcn_raw_buffer_load_b128(bdy, live ? (unsigned)((((n * H + y0 + (pix >> 4)) * W + x0 + (pix & 15)) * 16 + c4 * 4) * 4) : OOB, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NXR; ++j) {
            const int yy = y0 + xpy[j] - PAD, xx = x0 + xpx[j] - PAD;
            const bool ok = live && tid + j * 256 < NXL && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            rx[j] = __builtin_amdgcn_raw_buffer_load_b128(bx, ok ? (unsigned)((((n * H + yy) * W + xx) * CIN + xc4[j] * 4) * 4) : OOB, 0, 0);
        }
    };
    auto store_tile = [&](int buf, const u32x4 (&rdy)[4], const u32x4 (&rx)[NXR]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(&sDy[buf * STAGE + (tid + j * 256) * 4]) = rdy[j];
#pragma unroll
        for (int j = 0; j < NXR; ++j)
            if (tid + j * 256 < NXL) *reinterpret_cast<u32x4*>(&sX[buf * STAGE + (tid + j * 256) * 4]) = rx[j];
    };
    // one tile out of LDS stage `buf`.  Wave w: tile rows 4w .. 4w+3; k-step (ry, kx) = 4 consecutive pixels of a row.  The
    // operands of step s + 1 are read from LDS before the MFMAs of step s are issued (the scheduler otherwise places every
    // read next to its use)
    auto compute = [&](int buf) {
        const int so = buf * STAGE;
        if (p.dbias != nullptr) {
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) sacc += sDy[so + ((tid >> 4) * 16 + j) * 16 + (tid & 15)];
            bsum += sacc;
        }
        const float* ap = ap0 + so;
        float fa[2], fb[2][NT];
        fa[0] = ap[0];
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[0][t] = bp0[t][so];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            if (st + 1 < 16) {
                const int ry = (st + 1) >> 2, kx = (st + 1) & 3;
                fa[(st + 1) & 1] = ap[(ry * 16 + kx * 4) * 16];
#pragma unroll
                for (int t = 0; t < NT; ++t) fb[(st + 1) & 1][t] = bp0[t][so + (ry * PW + kx * 4) * CIN];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st & 1], fb[st & 1][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int G = gridDim.x;
    int tile = blockIdx.x, buf = 0;                                  // < tiles_total (grid <= tiles)
    load_tile(tile, rdyA, rxA);
    store_tile(0, rdyA, rxA);
    __syncthreads();
    for (; tile < tiles_total; tile += G, buf ^= 1) {
        const bool more = tile + G < tiles_total;
        if (more) load_tile(tile + G, rdyA, rxA);                    // in flight under this tile's MFMAs
        compute(buf);
        if (more) store_tile(buf ^ 1, rdyA, rxA);
        __syncthreads();                                             // stage buf^1 is complete, stage buf is free
    }
    // D: row 4g + e = co, col l15 = column.  The four waves' tiles side by side in LDS, summed on the way to the atomics.
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) sRed[wave * (NT * 256) + t * 256 + (4 * g + e) * 16 + l15] = acc[t][e];
    __syncthreads();
    // dW in memory order (a wave adds 64 consecutive floats: whole L2 lines; per-tile order costs the 7x7 -- 784-B rows --
    // three times the line requests, and the blocks, which finish together, queue on them).  Each block starts elsewhere.
    constexpr int NW = 16 * TAPS * CIN, NCH = (NW + 255) / 256;
    for (int j = 0; j < NCH; ++j) {
        const int f = ((j + (int)blockIdx.x) % NCH) * 256 + tid;
        if (f < NW) {
            const int co = f / (TAPS * CIN), k = f - co * (TAPS * CIN);
            const int i = (k >> 4) * 256 + co * 16 + (k & 15);   // accumulator tile k / 16 (one tap x 16 ci, or four taps x 4 ci)
            const float v = (sRed[i] + sRed[NT * 256 + i]) + (sRed[2 * NT * 256 + i] + sRed[3 * NT * 256 + i]);
            if (p.slab) p.slab[(size_t)blockIdx.x * p.slab_stride + f] = v;
            else atomicAdd(&p.dw[f], v);
        }
    }
    if (p.dbias != nullptr) {
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) bsum += __shfl_xor(bsum, off, 64);   // the wave's four pixel groups
        // (the bias of these two layers does not exist in the model: BatchNorm follows; kept on atomics)
        if (lane < 16) atomicAdd(&p.dbias[lane], bsum);
    }
}

// true + launched if the layer takes the patch kernel
static bool try_launch_wgrad_patch_f32(cr_ctx* ctx, WgP& p, int ks, int* rc) {
    static const int on = env_int("CR_WG_PATCH", 1);
    const bool shape_ok = p.Cout == 16 && p.stride == 1 && p.pad == ks / 2 && (p.Hout & 15) == 0 && (p.Wout & 15) == 0 &&
                          p.Hin == p.Hout && p.Win == p.Wout && ((ks == 3 && p.Cin == 16) || (ks == 7 && p.Cin == 4));
    if (!on || !shape_ok) return false;
    const int tiles_x = p.Wout / 16, tiles_per_img = tiles_x * (p.Hout / 16), tiles_total = tiles_per_img * p.N;
    static const int per_cu = env_int("CR_WG_PATCH_BLOCKS", 2);       // few long blocks: the prologue / epilogue of all blocks coincide
    const dim3 grid((unsigned)std::min(tiles_total, 256 * per_cu));   // block b: tiles b, b + grid, ...
    size_t ws_off = 0;
    float* const db = p.dbias;
    if (deterministic_on() && db) return false;                       // (bias on this kernel stays on atomics: generic kernel instead)
    const bool slabs = wgrad_use_slabs(ctx, p, (int)grid.x, &ws_off);
    if (ks == 3) hipLaunchKernelGGL((k_conv_wgrad_patch_f32<3, 16>), grid, dim3(256), 0, ctx->stream, p, tiles_x, tiles_per_img, tiles_total);
    else hipLaunchKernelGGL((k_conv_wgrad_patch_f32<7, 4>), grid, dim3(256), 0, ctx->stream, p, tiles_x, tiles_per_img, tiles_total);
    if (slabs) wgrad_reduce_slabs(ctx, p, (int)grid.x);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cr_set_error("k_conv_wgrad_patch_f32 launch failed: %s", hipGetErrorString(e)); *rc = CR_EHIP; }
    else *rc = CR_OK;
    return true;
}

template <int KS>
static int launch_wgrad_f32_ks(cr_ctx* ctx, WgP& p) {
    constexpr int PS = 16;
    const int nsteps = (p.M + PS - 1) / PS;
    const int tn = (int)cr_cdiv(p.Kdim, 128);
    static const int tm_cap = env_int("CR_WG_F32_TM", 128);
    int TM = p.Cout >= 128 ? 128 : (p.Cout >= 64 ? 64 : (p.Cout >= 32 ? 32 : 16));
    if (TM > tm_cap) TM = tm_cap;
    // measured (scripts/wgrad_mid_tune.py): 64-channel tiles (twice the tiles, half the pixel splits and atomics) win on the
    // 64x64 maps with 128 channels (69.7 -> 61.6 us) and the 32x32 maps with 256 (66.7 -> 60.1), lose elsewhere
    if (TM == 128 && ((p.Cout == 128 && p.M <= 16384) || (p.Cout == 256 && p.M <= 4096))) TM = 64;
    // the RoI heads' FC layers (1024 outputs, <= 2048 rows; scripts/fc_bench.py): 64-row tiles, 63 -> 52 us (1024 x 1024),
    // 158 -> 137 (12544 x 1024, 512 rows); see the split rule below for the 1568-tile layer
    if (TM == 128 && KS == 1 && p.Cout >= 1024) TM = 64;
    const int tm = (int)cr_cdiv(p.Cout, TM);
    const int tiles = tm * tn;
    // Split the pixel range over blocks.  Measured on the 3x3 256->256 layers (scripts/wgrad_f32_tune.py): one block per CU
    // (one wave per SIMD) exposes the LDS / barrier latencies (92 TFLOP/s), two reach 110, three 114; a block count just
    // ABOVE a multiple of the 256 CUs runs at the speed of the next multiple (792 blocks: 92).  So: the largest split count
    // that keeps the grid <= 3 x 256 blocks, every block keeping >= 8 sub-steps (128 pixels) to hide its atomics behind.
    static const int force_splits = env_int("CR_WG_SPLITS_F32", 0);
    const int max_splits = nsteps / 8 > 1 ? nsteps / 8 : 1;
    int splits = 768 / tiles;
    // more tiles than the chip holds at once (12544 x 1024: 1568): the grid is a little over a whole number of rounds and
    // the last few blocks cost a round of their own; halving the blocks halves that tail (567 -> 480 us)
    if (tiles > 768 && nsteps >= 32) splits = 2;
    if (splits > max_splits) splits = max_splits;
    if (force_splits > 0) splits = force_splits;
    if (splits > nsteps) splits = nsteps;
    if (splits < 1) splits = 1;
    p.steps_per_split = (nsteps + splits - 1) / splits;
    p.steps_per_split = (p.steps_per_split + 1) & ~1;            // whole stages (KU = 2 sub-steps)
    splits = (nsteps + p.steps_per_split - 1) / p.steps_per_split;
    dim3 grid(tm * tn * splits);
    p.tm = tm; p.tn = tn; p.xcd = xcd_enabled();
    size_t ws_off = 0;
    const bool slabs = wgrad_use_slabs(ctx, p, splits, &ws_off);
    static const int lds_pad = env_int("CR_WG_F32_LDS_PAD", 0);     // extra dynamic LDS per block: caps the blocks per CU (tuning)
    if (TM == 128) hipLaunchKernelGGL((k_conv_wgrad_f32<128, KS>), grid, dim3(CONV_T), lds_pad, ctx->stream, p);
    else if (TM == 64) hipLaunchKernelGGL((k_conv_wgrad_f32<64, KS>), grid, dim3(CONV_T), 0, ctx->stream, p);
    else if (TM == 32) hipLaunchKernelGGL((k_conv_wgrad_f32<32, KS>), grid, dim3(CONV_T), 0, ctx->stream, p);
    else hipLaunchKernelGGL((k_conv_wgrad_f32<16, KS>), grid, dim3(CONV_T), 0, ctx->stream, p);
    if (slabs) wgrad_reduce_slabs(ctx, p, splits);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// backward-weight in split mode: k_conv_wgrad's operand scheme (pixel-major operands, transposing LDS reads, split over pixel
// ranges with f32 atomics) on f32 inputs: a step's chunks (32 pixels) are loaded as f32, split on their way into LDS (three
// bf16 images per operand) and every 16x16 tile pair takes six MFMAs per step.
// Schedule: the split makes the staging phase VALU-heavy (176 VALU + 12 ds_write_b128 per lane and step) next to an MFMA
// phase of 96 MFMAs, and two co-resident blocks run in lockstep, so phases of the same kind collide instead of overlapping
// (measured: staging 204 us + fragment reads 83 us + MFMAs 185 us = the kernel's 472 us).  So a block is TWO wave groups
// (8 waves, one of each group per SIMD) on interleaved steps of the block's pixel range, each with its own LDS images,
// running half a step apart: while group A issues MFMAs group B splits and stores its next step, then they swap -- one
// block-wide barrier per phase.  The groups' accumulators are summed in LDS before the atomics.
template <int TM, int KS>
__global__ __launch_bounds__(CONV_T * 2) void k_conv_wgrad_s3(WgP p) {
    constexpr int TN = 128;
    constexpr int WM = (TM == 128) ? 2 : 1, WN = 4 / WM;
    constexpr int WTM = TM / WM, WTN = TN / WN;          // wave tile
    constexpr int TI = WTM / 16, TJ = WTN / 16;
    constexpr int PP = TM == 128 ? 144 : TM + 8, PQ = TN + 16;        // as k_conv_wgrad (bank-conflict-free tr reads)
    auto rotQ = [](int row, int col) { return (col + (((row >> 3) & 1) << 6)) & 127; };
    auto rotP = [&](int row, int col) { return TM == 128 ? rotQ(row, col) : col; };
    constexpr int CPR = TM / 8;                          // dy chunks (8 channels) per pixel row
    constexpr int NP = (32 * CPR + CONV_T - 1) / CONV_T; // dy chunks per thread per 32-pixel step
    constexpr int PE = 32 * PP, QE = 32 * PQ;            // elements of one plane of one operand
    constexpr int GE = 3 * (PE + QE);                    // elements of one group's images
    constexpr int TNP = TN + 4;                          // reduction tile pitch (floats)
    static_assert(TM * TNP * 4 <= 2 * GE * 2, "reduction tile must fit in the operand images");
    __shared__ __attribute__((aligned(16))) u16 smem[2 * GE];
    const int grp = (int)(threadIdx.x >> 8);
    u16* sP = smem + grp * GE;                           // planes at sP + pl * PE
    u16* sQ = sP + 3 * PE;                               // planes at sQ + pl * QE

    const int tid = threadIdx.x & (CONV_T - 1), lane = tid & 63, wave = tid >> 6;
    const int bid = p.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int bx = bid % p.tm, by = (bid / p.tm) % p.tn, bz = bid / (p.tm * p.tn);
    const int c0 = bx * TM, q0 = by * TN;
    const int step0 = bz * p.steps_per_split;
    const int nsteps_total = (p.M + 31) >> 5;
    const int step1 = min(step0 + p.steps_per_split, nsteps_total);
    if (step0 >= step1) return;            // block-uniform

    const int qc = tid & 15;
    const int qk = q0 + qc * 8;
    int qr = 0, qs = 0, qch = qk;
    const bool qvalid = qk < p.Kdim;
    if (KS > 1) {
        const int tap = qk >> p.cshift;
        qch = qk & (p.Cin - 1);
        qr = tap / KS;
        qs = tap - qr * KS;
    }
    const int first = step0 + grp;             // this group's first step; it then takes every second one
    int pn[2], pho[2], pwo[2];
    {
        const int hw = p.Hout * p.Wout;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = first * 32 + (tid >> 4) + 16 * i;
            pn[i] = m / hw;
            const int rem = m - pn[i] * hw;
            pho[i] = rem / p.Wout;
            pwo[i] = rem - pho[i] * p.Wout;
        }
    }
    int mrow = first * 32;
    const int mend = min(p.M, step1 * 32);     // rows of later splits / past M contribute zeros
    u32x4 rq[2][2][2], rp[2][NP][2];               // two register sets: loads run two steps ahead of their split
    auto advance = [&](int pixels) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pwo[i] += pixels;
            while (pwo[i] >= p.Wout) { pwo[i] -= p.Wout; ++pho[i]; }
            while (pho[i] >= p.Hout) { pho[i] -= p.Hout; ++pn[i]; }
        }
        mrow += pixels;
    };
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    auto load_stage = [&](auto SET) {
        constexpr int rs = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hi = pho[i] * p.stride - p.pad + qr, wi = pwo[i] * p.stride - p.pad + qs;
            const bool ok = qvalid && mrow < mend && pn[i] < p.N && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
            const unsigned off = (unsigned)(((pn[i] * p.Hin + hi) * p.Win + wi) * p.Cin + qch) * 4u;
            rq[rs][i][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : OOB, 0, 0);
            rq[rs][i][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off + 16u : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + CONV_T * i;
            const int prow = idx / CPR, pc = idx - prow * CPR;
            const bool ok = idx < 32 * CPR && mrow + prow < mend && c0 + pc * 8 < p.Cout;
            const unsigned off = (unsigned)((mrow + prow) * p.Cout + c0 + pc * 8) * 4u;
            rp[rs][i][0] = __builtin_amdgcn_raw_buffer_load_b128(rdy, ok ? off : OOB, 0, 0);
            rp[rs][i][1] = __builtin_amdgcn_raw_buffer_load_b128(rdy, ok ? off + 16u : OOB, 0, 0);
        }
        advance(64);                            // the other group takes the step in between
    };
    auto store_stage = [&](auto SET) {
        constexpr int rs = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 H, M, L;
            split3(rq[rs][i][0], rq[rs][i][1], H, M, L);
            const int row = (tid >> 4) + 16 * i;
            u16* d = &sQ[row * PQ + rotQ(row, qc * 8)];
            *reinterpret_cast<u32x4*>(d) = H;
            *reinterpret_cast<u32x4*>(d + QE) = M;
            *reinterpret_cast<u32x4*>(d + 2 * QE) = L;
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + CONV_T * i;
            const int prow = idx / CPR, pc = idx - prow * CPR;
            if (idx < 32 * CPR) {
                u32x4 H, M, L;
                split3(rp[rs][i][0], rp[rs][i][1], H, M, L);
                u16* d = &sP[prow * PP + rotP(prow, pc * 8)];
                *reinterpret_cast<u32x4*>(d) = H;
                *reinterpret_cast<u32x4*>(d + PE) = M;
                *reinterpret_cast<u32x4*>(d + 2 * PE) = L;
            }
        }
    };

    const int wm = (WM == 2) ? (wave >> 1) : 0, wn = (WM == 2) ? (wave & 1) : wave;
    const int moff = wm * WTM, noff = wn * WTN;
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const bool do_bias = p.dbias != nullptr && by == 0 && tid < TM;
    float bsum = 0.f;
    auto compute = [&]() {
        if (do_bias) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const u16* s0 = &sP[r * PP + rotP(r, tid)];
                bsum += (bf2f(s0[0]) + bf2f(s0[PE])) + bf2f(s0[2 * PE]);
            }
        }
        // dy planes stay in registers; the x planes come one at a time: x_h meets dy_l, dy_m, dy_h; x_m meets dy_m, dy_h;
        // x_l meets dy_h (smallest terms first within each accumulator)
        bf16x8 af[3][TI];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const u16* b0 = &sP[pl * PE + (8 * g + tq) * PP + rotP(8 * g + tq, moff + i * 16 + 4 * tp)];
                const s16x4 lo = lds_tr16(b0), hi = lds_tr16(b0 + 4 * PP);
                af[pl][i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
        for (int pb = 0; pb < 3; ++pb) {
            bf16x8 bfr[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const u16* b0 = &sQ[pb * QE + (8 * g + tq) * PQ + rotQ(8 * g + tq, noff + j * 16 + 4 * tp)];
                const s16x4 lo = lds_tr16(b0), hi = lds_tr16(b0 + 4 * PQ);
                bfr[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            if (!(p.dbg & 1)) {
#pragma unroll
                for (int pa = 2 - pb; pa >= 0; --pa)
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[pa][i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // steps of this group: first, first + 2, ...; both groups run the same number of phases (a group without a step left
    // stages and multiplies zeros: its loads are masked by mrow >= mend).  Loads run TWO steps ahead (alternating register
    // sets): one MFMA phase (~0.8 us) does not cover a global round trip under load.
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nst = (step1 - step0 + 1) >> 1;
    load_stage(S0{});
    load_stage(S1{});
    store_stage(S0{});
    __syncthreads();
    // one iteration = steps i (register set 0 holds step i + 2 afterwards) and i + 1 (set 1 -> step i + 3)
    auto phase_pair = [&](int i, auto CUR, auto NXT) {
        // even phase: A multiplies step i, B splits and stores step i;  odd phase: A stores step i + 1, B multiplies step i
        if (grp == 0) { if (i + 2 < nst) load_stage(CUR); compute(); }
        else if (i > 0) store_stage(CUR);
        __syncthreads();
        if (grp == 0) { if (i + 1 < nst) store_stage(NXT); }
        else { if (i + 2 < nst) load_stage(CUR); compute(); }
        __syncthreads();
    };
    for (int i = 0; i < nst; i += 2) {
        phase_pair(i, S0{}, S1{});
        if (i + 1 < nst) phase_pair(i + 1, S1{}, S0{});
    }
    if (do_bias && c0 + tid < p.Cout) atomicAdd(&p.dbias[c0 + tid], bsum);
    if (p.dbg & 4) { if (acc[0][0][0] == 123.456f) p.dw[0] = 1.f; return; }
    // sum the two groups' accumulators in LDS (the operand images are dead: the loop ended with a barrier), then one set
    // of atomics per block, full 512-B rows per wave-instruction.  D: col (lane&15) = k index, row 4(lane>>4)+reg = channel
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
        if (grp == gg) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* d = &tile[(moff + i * 16 + 4 * g + e) * TNP + noff + j * 16 + li];
                        if (gg == 0) *d = acc[i][j][e]; else *d += acc[i][j][e];
                    }
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < TM * TN; idx += CONV_T * 2) {
        const int chl = idx / TN, kl = idx - chl * TN;
        const int ch = c0 + chl, kk = q0 + kl;
        if (kk < p.Kdim && ch < p.Cout) atomicAdd(&p.dw[(size_t)ch * p.Kdim + kk], tile[chl * TNP + kl]);
    }
}

// ---------------------------------------------------------------------------
// k_conv_wgrad_s3_row: split-mode weight gradient of a 3x3 / stride-1 / pad-1 convolution, one FILTER ROW per block.
// The weight gradient is bound by L2 -> LDS bytes (k_conv_wgrad_s3 at 128 x 128 tiles reads 2.4 GB per launch on the
// 4x128x128x256 layer: 0.65 ms in the step, 0.43 with warm caches).  The three horizontal taps (r, 0..2) of a filter row
// read the SAME input pixels shifted by one column, so a block that owns all three needs per 32-pixel step ONE window of
// 34 pixels x 128 input channels (not 3 x 32) and the 32 x 128 dy tile once (not three times): 33 KB per 3 x 128 x 128 x 32
// multiply-adds instead of 32 KB per 128 x 128 x 32 -- a third of the bytes per flop.
//   tile: 128 output channels x (3 taps x 128 input channels); 8 waves (2 x 4), wave tile 64 x 96 = 24 accumulator tiles;
//   LDS: two buffers x three bf16 planes x (32 dy rows + 36 window rows) x 144 = 117.5 KB, one block per CU; staging of step
//   i + 1 (global f32 -> split3 -> planes of the other buffer) is issued inside the MFMA stream of step i, one barrier per step;
//   tap s of pixel j is window row j + s (a row offset of the transposing read); the pixels whose horizontal neighbour falls
//   off the image row (w = 0 for s = 0, w = W - 1 for s = 2) are zeroed in the fragment by a per-step bit mask; vertical
//   padding and image boundaries are zero-filled when the window is loaded.
// Requires W a power of two >= 8, Cin % 128 == 0, Cout % 128 == 0, M % 32 == 0.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(CONV_T * 2) void k_conv_wgrad_s3_row(WgP p) {
    constexpr int TM = 128, PP = 144, PQ = 144, WR = 36;           // window rows held (34 used)
    constexpr int TI = 4, TJ = 6;                                   // wave tile 64 channels x 96 columns
    constexpr int PE = 32 * PP, QE = WR * PQ;
    constexpr int BUF = 3 * (PE + QE);
    extern __shared__ __attribute__((aligned(16))) u16 smem_row[];  // [2][BUF]
    auto rot = [](int row, int col) { return (col + (((row >> 3) & 1) << 6)) & 127; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = p.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int bx = bid % p.tm, by = (bid / p.tm) % p.tn, bz = bid / (p.tm * p.tn);
    const int c0 = bx * TM;
    const int nhalf = p.Cin >> 7;
    const int r = by / nhalf, chalf = by - r * nhalf;               // filter row, input-channel chunk of 128
    const int step0 = bz * p.steps_per_split;
    const int nsteps_total = p.M >> 5;
    const int step1 = min(step0 + p.steps_per_split, nsteps_total);
    if (step0 >= step1) return;            // block-uniform

    const int W = p.Wout, H = p.Hout, wsh = p.cshift_w;             // W = 1 << wsh
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int chunk = tid & 15, srow = tid >> 4;                    // staging: 16 chunks of 8 channels per row, 32 rows per pass
    const int hw = H * W;
    // window row wr of a step starting at pixel m is pixel m - 1 + wr, read r - 1 image rows away
    auto win_off = [&](int pix) -> unsigned {
        if (pix < 0 || pix >= p.M) return OOB;
        const int n = p.cshift_hw >= 0 ? (pix >> p.cshift_hw) : pix / hw, rem = pix - n * hw;
        const int h = (rem >> wsh) + r - 1, w = rem & (W - 1);
        if ((unsigned)h >= (unsigned)H) return OOB;
        return (unsigned)(((n * H + h) * W + w) * p.Cin + chalf * 128 + chunk * 8) * 4u;
    };
    u32x4 rq[2][2], rp[2], rqx[2];                                  // window rows srow / srow + 32 (tid < 32), dy row srow
    auto load_stage = [&](int step) {
        const int m = step << 5;
        const unsigned o0 = win_off(m - 1 + srow);
        rq[0][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, o0, 0, 0);
        rq[0][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, o0 == OOB ? OOB : o0 + 16u, 0, 0);
        if (tid < 32) {                                             // window rows 32, 33
            const unsigned o1 = win_off(m - 1 + 32 + srow);
            rqx[0] = __builtin_amdgcn_raw_buffer_load_b128(rx, o1, 0, 0);
            rqx[1] = __builtin_amdgcn_raw_buffer_load_b128(rx, o1 == OOB ? OOB : o1 + 16u, 0, 0);
        }
        const unsigned od = (unsigned)((m + srow) * p.Cout + c0 + chunk * 8) * 4u;
        rp[0] = __builtin_amdgcn_raw_buffer_load_b128(rdy, od, 0, 0);
        rp[1] = __builtin_amdgcn_raw_buffer_load_b128(rdy, od + 16u, 0, 0);
    };
    auto store_stage = [&](int buf) {
        u16* sP = smem_row + buf * BUF;
        u16* sQ = sP + 3 * PE;
        u32x4 Hh, Mm, Ll;
        split3(rp[0], rp[1], Hh, Mm, Ll);
        u16* d = &sP[srow * PP + rot(srow, chunk * 8)];
        *reinterpret_cast<u32x4*>(d) = Hh; *reinterpret_cast<u32x4*>(d + PE) = Mm; *reinterpret_cast<u32x4*>(d + 2 * PE) = Ll;
        split3(rq[0][0], rq[0][1], Hh, Mm, Ll);
        d = &sQ[srow * PQ + rot(srow, chunk * 8)];
        *reinterpret_cast<u32x4*>(d) = Hh; *reinterpret_cast<u32x4*>(d + QE) = Mm; *reinterpret_cast<u32x4*>(d + 2 * QE) = Ll;
        if (tid < 32) {
            split3(rqx[0], rqx[1], Hh, Mm, Ll);
            d = &sQ[(32 + srow) * PQ + rot(32 + srow, chunk * 8)];
            *reinterpret_cast<u32x4*>(d) = Hh; *reinterpret_cast<u32x4*>(d + QE) = Mm; *reinterpret_cast<u32x4*>(d + 2 * QE) = Ll;
        }
    };

    const int wm = wave >> 2, wn = wave & 3;
    const int moff = wm * 64, noff = wn * 96;
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const bool do_bias = p.dbias != nullptr && by == 0 && tid < TM;
    float bsum = 0.f;
    // per column tile: tap s and channel offset inside the tap
    int tap_s[TJ], tap_c[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) { const int col = noff + 16 * j; tap_s[j] = col >> 7; tap_c[j] = col & 127; }

    auto compute = [&](int buf, unsigned mask0, unsigned mask2) {
        const u16* sP = smem_row + buf * BUF;
        const u16* sQ = sP + 3 * PE;
        if (do_bias) {
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) {
                const u16* s0 = &sP[rr * PP + rot(rr, tid)];
                bsum += (bf2f(s0[0]) + bf2f(s0[PE])) + bf2f(s0[2 * PE]);
            }
        }
        // this lane's 8 pixels are 8g .. 8g+7: per-dword AND masks of the two edge taps
        const unsigned b0m = (mask0 >> (8 * g)) & 0xffu, b2m = (mask2 >> (8 * g)) & 0xffu;
        auto dy_plane = [&](int pl, bf16x8 (&a)[TI]) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const u16* b0 = &sP[pl * PE + (8 * g + tq) * PP + rot(8 * g + tq, moff + i * 16 + 4 * tp)];
                const s16x4 lo = lds_tr16(b0), hi = lds_tr16(b0 + 4 * PP);
                a[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        };
        auto x_frag = [&](int pb, int j) -> bf16x8 {
            const int row0 = 8 * g + tq + tap_s[j];                      // window row of pixel 8g + tq for this tap
            const u16* b0 = &sQ[pb * QE + row0 * PQ + rot(row0, tap_c[j] + 4 * tp)];
            const int row1 = row0 + 4;
            const u16* b1 = &sQ[pb * QE + row1 * PQ + rot(row1, tap_c[j] + 4 * tp)];
            const s16x4 lo = lds_tr16(b0), hi = lds_tr16(b1);
            u32x4 v = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            const unsigned bm = tap_s[j] == 0 ? b0m : (tap_s[j] == 2 ? b2m : 0xffu);
            if (bm != 0xffu) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[q] &= (((bm >> (2 * q)) & 1u) ? 0x0000ffffu : 0u) | (((bm >> (2 * q + 1)) & 1u) ? 0xffff0000u : 0u);
            }
            return __builtin_bit_cast(bf16x8, v);
        };
        // 96 accumulator registers leave room for two dy planes at a time: dy_l meets x_h first (and is dropped), then dy_m
        // and dy_h meet x_h (read again), x_m, and dy_h alone x_l -- smallest terms first within each accumulator
        {
            bf16x8 al[TI];
            dy_plane(2, al);
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const bf16x8 bq = x_frag(0, j);
#pragma unroll
                for (int i = 0; i < TI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bq, acc[i][j], 0, 0, 0);
            }
        }
        bf16x8 am[TI], ah[TI];
        dy_plane(1, am);
        dy_plane(0, ah);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const bf16x8 bq = x_frag(pb, j);
#pragma unroll
                for (int i = 0; i < TI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bq, acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bq, acc[i][j], 0, 0, 0);
            }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const bf16x8 bq = x_frag(2, j);
#pragma unroll
            for (int i = 0; i < TI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bq, acc[i][j], 0, 0, 0);
        }
    };

    // horizontal-edge masks of a step starting at pixel m (m % 32 == 0): bit j = pixel m + j has a left / right neighbour
    auto edge_masks = [&](int step, unsigned& m0, unsigned& m2) {
        if (W >= 32) {
            const int w0 = (step << 5) & (W - 1);
            m0 = w0 == 0 ? ~1u : ~0u;
            m2 = (w0 + 32 == W) ? ~(1u << 31) : ~0u;
        } else if (W == 16) { m0 = ~0x00010001u; m2 = ~0x80008000u; }
        else { m0 = ~0x01010101u; m2 = ~0x80808080u; }              // W == 8
    };

    load_stage(step0);
    store_stage(0);
    if (step0 + 1 < step1) load_stage(step0 + 1);
    __syncthreads();
    for (int st = step0; st < step1; ++st) {
        const int buf = (st - step0) & 1;
        unsigned m0, m2;
        edge_masks(st, m0, m2);
        if (st + 1 < step1) store_stage(buf ^ 1);                    // data of step st + 1 (loaded one step ago)
        if (st + 2 < step1) load_stage(st + 2);
        compute(buf, m0, m2);
        __syncthreads();
    }
    if (do_bias && c0 + tid < p.Cout) atomicAdd(&p.dbias[c0 + tid], bsum);
    // D: col (lane&15) = column, row 4(lane>>4)+reg = channel; column -> (tap s, input channel)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int kk = (r * 3 + tap_s[j]) * p.Cin + chalf * 128 + tap_c[j] + li;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = c0 + moff + i * 16 + 4 * g + e;
                atomicAdd(&p.dw[(size_t)ch * p.Kdim + kk], acc[i][j][e]);
            }
        }
}

static bool try_launch_wgrad_s3_row(cr_ctx* ctx, WgP& p, int* rc) {
    static const int on = env_int("CR_S3_WGRAD_ROW", 1);
    const int W = p.Wout;
    if (!on || p.stride != 1 || p.pad != 1 || (p.Cin & 127) || (p.Cout & 127) || W < 8 || (W & (W - 1)) || (p.M & 31) ||
        p.Hin != p.Hout || p.Win != p.Wout || p.M < 1024)
        return false;
    const int nsteps = p.M >> 5;
    const int tm = p.Cout / 128, tn = 3 * (p.Cin / 128);
    const int tiles = tm * tn;
    static const int force_splits = env_int("CR_WG_SPLITS_S3R", 0), target = env_int("CR_WG_S3R_BLOCKS", 256);
    int splits = target / tiles;
    // A row tile is three 128 x 128 tiles: the grid needs three times the pixel splits of k_conv_wgrad_s3 to fill the chip,
    // each with a 128 x 384 set of atomics.  Measured (scripts/conv_shapes_bench.py, fp32x3): it pays on the 4x128x128x256
    // layers (0.65 -> 0.47 ms in the step) and loses on the 64x64 and smaller maps (70 vs 43 us on 3x3 128->128), i.e. it needs
    // >= 64 steps per block to amortise them.
    static const int min_steps = env_int("CR_WG_S3R_MIN_STEPS", 64);
    if (splits < 1 || nsteps / splits < min_steps) return false;
    if (force_splits > 0) splits = force_splits;
    if (splits > nsteps) splits = nsteps;
    p.steps_per_split = (nsteps + splits - 1) / splits;
    splits = (nsteps + p.steps_per_split - 1) / p.steps_per_split;
    p.tm = tm; p.tn = tn; p.xcd = xcd_enabled();
    p.cshift_w = ilog2_exact(W);
    p.cshift_hw = ilog2_exact(p.Hout * W);            // -1: the image size is not a power of two (division per window row)
    constexpr size_t LDS = 2 * 3 * (32 * 144 + 36 * 144) * sizeof(u16);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t ea = hipFuncSetAttribute((const void*)k_conv_wgrad_s3_row, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (ea != hipSuccess) { cr_set_error("k_conv_wgrad_s3_row: LDS attribute: %s", hipGetErrorString(ea)); *rc = CR_EHIP; return true; }
        attr_done = true;
    }
    hipLaunchKernelGGL(k_conv_wgrad_s3_row, dim3((unsigned)(tiles * splits)), dim3(CONV_T * 2), LDS, ctx->stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cr_set_error("k_conv_wgrad_s3_row launch failed: %s", hipGetErrorString(e)); *rc = CR_EHIP; }
    else *rc = CR_OK;
    return true;
}

// true + launched when the layer takes the split-mode weight-gradient kernel
template <int KS>
static bool try_launch_wgrad_s3(cr_ctx* ctx, WgP& p, int* rc) {
    static const int on = env_int("CR_S3_WGRAD", 1);
    if (!on || p.Cout < 64 || (p.Cout & 7) || (p.Cin & 7) || p.M < 512) return false;
    const int nsteps = (p.M + 31) >> 5;
    const int tn = (int)cr_cdiv(p.Kdim, 128);
    const int TM = p.Cout >= 128 ? 128 : 64;
    const int tm = (int)cr_cdiv(p.Cout, TM);
    const int tiles = tm * tn;
    // 110 KB of LDS per block of 8 waves: one block per CU; every block keeps >= 8 steps (4 per group)
    static const int force_splits = env_int("CR_WG_SPLITS_S3", 0), target = env_int("CR_WG_S3_BLOCKS", 256);
    int splits = target / tiles;
    if (splits > nsteps / 8) splits = nsteps / 8;
    if (force_splits > 0) splits = force_splits;
    if (splits > nsteps) splits = nsteps;
    if (splits < 1) splits = 1;
    p.steps_per_split = (nsteps + splits - 1) / splits;
    p.steps_per_split = (p.steps_per_split + 1) & ~1;             // whole step pairs (one step per group)
    splits = (nsteps + p.steps_per_split - 1) / p.steps_per_split;
    dim3 grid(tm * tn * splits);
    p.tm = tm; p.tn = tn; p.xcd = xcd_enabled();
    static const int dbg = env_int("CR_S3_DBG", 0);
    p.dbg = dbg;
    if (TM == 128) hipLaunchKernelGGL((k_conv_wgrad_s3<128, KS>), grid, dim3(CONV_T * 2), 0, ctx->stream, p);
    else hipLaunchKernelGGL((k_conv_wgrad_s3<64, KS>), grid, dim3(CONV_T * 2), 0, ctx->stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cr_set_error("k_conv_wgrad_s3 launch failed: %s", hipGetErrorString(e)); *rc = CR_EHIP; }
    else *rc = CR_OK;
    return true;
}

template <int KS>
static int launch_wgrad_ks(cr_ctx* ctx, WgP& p) {
    const int nsteps = (p.M + 31) >> 5;
    const int tn = (int)cr_cdiv(p.Kdim, 128);
    int TM = p.Cout >= 128 ? 128 : (p.Cout >= 64 ? 64 : (p.Cout >= 32 ? 32 : 16));
    // wave groups per block (intra-block split over pixels): 0 = decide below
    static const int force_kg = env_int("CR_WG_KG", 0);
    int kg = force_kg;
    if (kg <= 0) {      // measured per layer shape (scripts/conv_shapes_bench.py, CR_WG_KG sweep)
        const int64_t dw_elems = (int64_t)p.Cout * p.Kdim;
        if (p.M >= 1024 && p.Cout >= 64 && dw_elems <= 160 * 1024) kg = 4;
        else if (p.M >= 1024 && p.M <= 4096 && p.Cout >= 128) kg = 2;
        else kg = 1;
    }
    if (kg == 4 && TM == 128) TM = 64;              // 16 waves per block: 128 registers per lane
    if (kg == 2 && TM != 128) kg = 1;
    const int tm = (int)cr_cdiv(p.Cout, TM);
    // Split the pixel range over blockIdx.z.  Every split adds one full set of f32 atomics over dW (1.3 TB/s on
    // MI355X), so the split count is bounded by ~16 pixel steps (512 pixels) of MFMA work per block and by the number
    // of blocks the chip holds at once (measured per layer shape with scripts/conv_shapes_bench.py, CR_WG_SPLITS sweep).
    const int tiles = tm * tn;
    const int block_cap = TM >= 64 ? 576 : 2048;
    int splits = nsteps / (16 * kg);
    if (splits > block_cap / tiles) splits = block_cap / tiles;
    if (kg == 1 && TM >= 64) {
        // large grids: exactly one resident wave of blocks (3 per CU = 768) when every block still gets >= 64 pixel steps
        // to hide its atomics behind; otherwise at least 40 steps per block (the atomics of every extra split cost as
        // much as ~2 steps of MFMA work on the 64x64 maps)
        const int s_full = 768 / tiles;
        if (s_full >= 1 && nsteps / s_full >= 64) splits = s_full;
        else if (splits > nsteps / 40) splits = nsteps / 40;
    }
    if (splits < 1) splits = 1;
    if (tiles * splits < 256) {     // far fewer blocks than CUs: trade steps per block (down to ~8) for more blocks
        int lo = nsteps < 4 ? nsteps : 4;
        int s2 = nsteps / 8 > lo ? nsteps / 8 : lo;
        if (s2 > 256 / tiles) s2 = 256 / tiles;
        if (s2 > splits) splits = s2;
    }
    static const int force_splits = env_int("CR_WG_SPLITS", 0);
    if (force_splits > 0) splits = force_splits;
    if (splits > nsteps) splits = nsteps;
    if (splits < 1) splits = 1;
    p.steps_per_split = (nsteps + splits - 1) / splits;
    splits = (nsteps + p.steps_per_split - 1) / p.steps_per_split;
    dim3 grid(tm * tn * splits);
    p.tm = tm; p.tn = tn; p.xcd = xcd_enabled();
    if (kg == 4) {
        if (TM == 64) hipLaunchKernelGGL((k_conv_wgrad<64, KS, 2, 4>), grid, dim3(CONV_T * 4), 0, ctx->stream, p);
        else if (TM == 32) hipLaunchKernelGGL((k_conv_wgrad<32, KS, 2, 4>), grid, dim3(CONV_T * 4), 0, ctx->stream, p);
        else hipLaunchKernelGGL((k_conv_wgrad<16, KS, 2, 4>), grid, dim3(CONV_T * 4), 0, ctx->stream, p);
    } else if (kg == 2) {
        hipLaunchKernelGGL((k_conv_wgrad<128, KS, 2, 2>), grid, dim3(CONV_T * 2), 0, ctx->stream, p);
    } else {
        if (TM == 128) hipLaunchKernelGGL((k_conv_wgrad<128, KS, 2, 1>), grid, dim3(CONV_T), 0, ctx->stream, p);
        else if (TM == 64) hipLaunchKernelGGL((k_conv_wgrad<64, KS, 2, 1>), grid, dim3(CONV_T), 0, ctx->stream, p);
        else if (TM == 32) hipLaunchKernelGGL((k_conv_wgrad<32, KS, 2, 1>), grid, dim3(CONV_T), 0, ctx->stream, p);
        else hipLaunchKernelGGL((k_conv_wgrad<16, KS, 2, 1>), grid, dim3(CONV_T), 0, ctx->stream, p);
    }
    CR_LAUNCH_CHECK();
    return CR_OK;
}

__global__ void k_fill_zero_f32(float* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// dW[k][r][s][c] (+)= sum dY * X ; dw is f32 [Cout][ks*ks*Cin]; `accumulate`=0 zeroes it first (with a kernel, not
// hipMemsetAsync: memset nodes inside captured HIP graphs were observed to leave garbage on ROCm 7.2).
static int conv2d_bwd_weight_impl(cr_ctx* ctx, const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W,
                                  int Cin, int Cout, int ks, int stride, int pad, int accumulate, int act_f32) {
    CR_CHECK_ARG(ctx && dy && x && dw, "cr_conv2d_bwd_weight: NULL pointer");
    int rc = conv_common_checks("cr_conv2d_bwd_weight", N, H, W, Cin, Cout, ks, stride, pad, act_f32);
    if (rc) return rc;
    const size_t es = act_f32 ? 4 : 2;
    WgP p;
    p.dbg = 0; p.cshift_w = 0; p.cshift_hw = -1; p.slab = nullptr; p.bslab = nullptr; p.slab_stride = 0;
    p.dy = dy; p.x = x; p.dw = dw; p.dbias = dbias;
    p.N = N; p.Hin = H; p.Win = W; p.Cin = Cin; p.Cout = Cout;
    p.Hout = (H + 2 * pad - ks) / stride + 1;
    p.Wout = (W + 2 * pad - ks) / stride + 1;
    p.stride = stride; p.pad = pad; p.Kdim = ks * ks * Cin; p.M = N * p.Hout * p.Wout;
    p.cshift = ks == 1 ? 0 : ilog2_exact(Cin);
    p.x_bytes = (unsigned)((size_t)N * H * W * Cin * es); p.dy_bytes = (unsigned)((size_t)p.M * Cout * es);
    if (!accumulate) {
        const int64_t nz = (int64_t)Cout * p.Kdim;
        hipLaunchKernelGGL(k_fill_zero_f32, dim3((unsigned)cr_cdiv(nz, 256)), dim3(256), 0, ctx->stream, dw, nz);
        CR_LAUNCH_CHECK();
    }
    if (act_f32 == 2 && ks == 3) {
        int rcr = CR_OK;
        if (try_launch_wgrad_s3_row(ctx, p, &rcr)) return rcr;
    }
    if (act_f32 == 2 && ks != 7) {
        int rc3 = CR_OK;
        if (ks == 1 ? try_launch_wgrad_s3<1>(ctx, p, &rc3) : try_launch_wgrad_s3<3>(ctx, p, &rc3)) return rc3;
    }
    if (act_f32 && ks != 1) {
        int rcp = CR_OK;
        if (try_launch_wgrad_patch_f32(ctx, p, ks, &rcp)) return rcp;
    }
    if (act_f32) {
        if (ks == 1) return launch_wgrad_f32_ks<1>(ctx, p);
        if (ks == 3) return launch_wgrad_f32_ks<3>(ctx, p);
        return launch_wgrad_f32_ks<7>(ctx, p);
    }
    if (ks == 1) return launch_wgrad_ks<1>(ctx, p);
    if (ks == 3) return launch_wgrad_ks<3>(ctx, p);
    return launch_wgrad_ks<7>(ctx, p);
}

extern "C" int cr_conv2d_bwd_weight(cr_ctx* ctx, const void* dy, const void* x, float* dw, int N, int H, int W,
                                    int Cin, int Cout, int ks, int stride, int pad, int accumulate, int act_f32) {
    return conv2d_bwd_weight_impl(ctx, dy, x, dw, nullptr, N, H, W, Cin, Cout, ks, stride, pad, accumulate, act_f32);
}

// same, and dbias[Cout] += sum over pixels of dy (the bias gradient of a conv with bias: detectron2 FPN / RPN head convs)
extern "C" int cr_conv2d_bwd_weight_bias(cr_ctx* ctx, const void* dy, const void* x, float* dw, float* dbias, int N, int H,
                                         int W, int Cin, int Cout, int ks, int stride, int pad, int accumulate, int act_f32) {
    CR_CHECK_ARG(dbias, "cr_conv2d_bwd_weight_bias: NULL dbias");
    return conv2d_bwd_weight_impl(ctx, dy, x, dw, dbias, N, H, W, Cin, Cout, ks, stride, pad, accumulate, act_f32);
}

// dW_i += dY_i^T X_i (f32 mode, k in {1, 3}, stride 1, Cout % 128 == 0) for n convolutions in one launch; dws may repeat a
// pointer (shared weights: the contributions are summed); dbiases (or entries) may be NULL.  ALWAYS accumulates.
// `batches` float32 weight-gradient GEMMs of one shape in ONE launch: dw[b] (O,K) += dy[b] (R,O)^T x[b] (R,K) (f32 atomics over
// the pixel splits), operands of batch b at base + b * stride elements -- the 16 products of a Winograd weight gradient
// (csrc/winograd.hip).  One launch lets a block keep a long pixel range (12 splits instead of 170 per product: the atomics of a
// 128 x 128 tile are paid 14 times less often).  dw must be zeroed (or hold what is to be added to).
__global__ __launch_bounds__(CONV_T) void k_wgrad_batched_f32(WgP p, long long sdy, long long sx, long long sdw, int count) {
    const long long b = blockIdx.y;
    p.dy = (const float*)p.dy + b * sdy;
    p.x = (const float*)p.x + b * sx;
    p.dw = p.dw + b * sdw;
    conv_wgrad_f32_body<128, 1>(p, (int)blockIdx.x, count);
}

extern "C" int cr_wgrad_batched_f32(cr_ctx* ctx, const float* dy, const float* x, float* dw, int R, int K, int O, int batches,
                                    int64_t stride_dy, int64_t stride_x, int64_t stride_dw) {
    CR_CHECK_ARG(ctx && dy && x && dw && R > 0 && batches >= 1 && batches <= 65535, "cr_wgrad_batched_f32: bad args");
    CR_CHECK_ARG(O % 128 == 0 && K % 32 == 0, "cr_wgrad_batched_f32: O %% 128 == 0, K %% 32 == 0");
    WgP p;
    p.dbg = 0; p.cshift_w = 0; p.cshift_hw = -1; p.slab = nullptr; p.bslab = nullptr; p.slab_stride = 0;
    p.dy = dy; p.x = x; p.dw = dw; p.dbias = nullptr;
    p.N = 1; p.Hin = 1; p.Win = R; p.Cin = K; p.Cout = O; p.Hout = 1; p.Wout = R;
    p.stride = 1; p.pad = 0; p.Kdim = K; p.M = R; p.cshift = 0;
    p.x_bytes = (unsigned)((size_t)R * K * 4); p.dy_bytes = (unsigned)((size_t)R * O * 4);
    const int nsteps = (R + 15) / 16;
    const int tm = O / 128, tn = (int)cr_cdiv(K, 128);
    const int tiles = tm * tn * batches;
    int splits = 768 / tiles;
    const int max_splits = nsteps / 8 > 1 ? nsteps / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.steps_per_split = (nsteps + splits - 1) / splits;
    p.steps_per_split = (p.steps_per_split + 1) & ~1;
    splits = (nsteps + p.steps_per_split - 1) / p.steps_per_split;
    p.tm = tm; p.tn = tn; p.xcd = xcd_enabled();
    const int count = tm * tn * splits;
    hipLaunchKernelGGL(k_wgrad_batched_f32, dim3((unsigned)count, (unsigned)batches), dim3(CONV_T), 0, ctx->stream, p,
                       (long long)stride_dy, (long long)stride_x, (long long)stride_dw, count);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_conv2d_bwd_weight_group(cr_ctx* ctx, int n, const void* const* dys, const void* const* xs, float* const* dws,
                                          float* const* dbiases, const int* Ns, const int* Hs, const int* Ws, int Cin, int Cout,
                                          int ks, int pad, int act_f32) {
    CR_CHECK_ARG(ctx && dys && xs && dws && Ns && Hs && Ws, "cr_conv2d_bwd_weight_group: NULL pointer");
    CR_CHECK_ARG(n >= 1 && n <= CR_MAX_GROUP, "cr_conv2d_bwd_weight_group: 1..%d problems", CR_MAX_GROUP);
    CR_CHECK_ARG((ks == 1 || ks == 3) && act_f32 == 1 && Cout % 128 == 0, "cr_conv2d_bwd_weight_group: fp32, k in {1,3}, Cout %% 128 == 0");
    WgGroup g;
    int counts[CR_MAX_GROUP];
    g.n = n;
    constexpr int PS = 16;
    int64_t steps_all = 0;
    for (int i = 0; i < n; ++i) steps_all += cr_cdiv((int64_t)Ns[i] * (Hs[i] + 2 * pad - ks + 1) * (Ws[i] + 2 * pad - ks + 1), PS);
    const int tm = Cout / 128, tn = (int)cr_cdiv((int64_t)ks * ks * Cin, 128), tiles = tm * tn;
    // one resident round of blocks over the whole group (3 per CU, the single-problem rule of launch_wgrad_f32_ks), every
    // block keeping >= 8 sub-steps; the same pixel count per block in every problem
    int splits_all = 768 / tiles > 1 ? 768 / tiles : 1;
    int sps = (int)cr_cdiv(steps_all, splits_all);
    if (sps < 8) sps = 8;
    sps = (sps + 1) & ~1;
    for (int i = 0; i < n; ++i) {
        CR_CHECK_ARG(dys[i] && xs[i] && dws[i], "cr_conv2d_bwd_weight_group: NULL tensor %d", i);
        int rc = conv_common_checks("cr_conv2d_bwd_weight_group", Ns[i], Hs[i], Ws[i], Cin, Cout, ks, 1, pad, act_f32);
        if (rc) return rc;
        WgP& p = g.p[i];
        p.dbg = 0; p.cshift_w = 0; p.cshift_hw = -1; p.slab = nullptr; p.bslab = nullptr; p.slab_stride = 0;
        p.dy = dys[i]; p.x = xs[i]; p.dw = dws[i]; p.dbias = dbiases ? dbiases[i] : nullptr;
        p.N = Ns[i]; p.Hin = Hs[i]; p.Win = Ws[i]; p.Cin = Cin; p.Cout = Cout;
        p.Hout = Hs[i] + 2 * pad - ks + 1; p.Wout = Ws[i] + 2 * pad - ks + 1;
        p.stride = 1; p.pad = pad; p.Kdim = ks * ks * Cin; p.M = p.N * p.Hout * p.Wout;
        p.cshift = ks == 1 ? 0 : ilog2_exact(Cin);
        p.x_bytes = (unsigned)((size_t)p.N * p.Hin * p.Win * Cin * 4); p.dy_bytes = (unsigned)((size_t)p.M * Cout * 4);
        const int nsteps = (p.M + PS - 1) / PS;
        p.steps_per_split = sps;
        p.tm = tm; p.tn = tn; p.xcd = xcd_enabled();
        counts[i] = tiles * (int)cr_cdiv(nsteps, sps);
    }
    // deterministic mode: every problem gets its own slabs (all of them, or none: shared parameters must not mix the two forms)
    size_t ws_off = 0;
    bool slabs = deterministic_on();
    for (int i = 0; i < n && slabs; ++i) slabs = wgrad_use_slabs(ctx, g.p[i], counts[i] / tiles, &ws_off);
    if (!slabs)
        for (int i = 0; i < n; ++i) { g.p[i].slab = nullptr; g.p[i].bslab = nullptr; g.p[i].slab_stride = 0; }
    const int total = group_starts(n, counts, g.start);
    for (int i = 0; i < n; ++i) g.count[i] = counts[i];
    for (int i = n; i < CR_MAX_GROUP; ++i) { g.start[i] = total; g.count[i] = 0; }
    const dim3 grid((unsigned)total), block(CONV_T);
    if (ks == 1) hipLaunchKernelGGL(k_conv_wgrad_f32_grp<1>, grid, block, 0, ctx->stream, g);
    else hipLaunchKernelGGL(k_conv_wgrad_f32_grp<3>, grid, block, 0, ctx->stream, g);
    if (slabs)
        for (int i = 0; i < n; ++i) wgrad_reduce_slabs(ctx, g.p[i], counts[i] / tiles);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// weight preparation: f32 master [Cout][ks*ks][Cin] (channels_last physical layout of a
// (Cout,Cin,ks,ks) parameter) -> bf16 same layout, and -> bf16 [Cin][ks*ks][Cout] for bwd-data.
// ---------------------------------------------------------------------------
__global__ void k_cast_f32_bf16(const float* __restrict__ src, u16* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        uint2 pk;
        pk.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
        pk.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
        *reinterpret_cast<uint2*>(dst + i) = pk;
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = f2bf(src[j]);
    }
}

// Inference-time BatchNorm folding: a frozen BatchNorm after a convolution is a per-output-channel affine map, so
// wf[co][:] = w[co][:] * s[co] (bf16) and bias[co] = beta[co] - mean[co] * s[co] with s = gamma / sqrt(var + eps) turn
// conv + BN (+ residual, ReLU) into ONE convolution with a bias epilogue: no statistics, no second pass over the
// activations.  One launch per layer (weights may have changed since the last call; the result is tiny).
template <typename T>
__global__ __launch_bounds__(256) void k_fold_bn(const float* __restrict__ w, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ mean,
                                                 const float* __restrict__ var, float eps, T* __restrict__ wf,
                                                 float* __restrict__ bias, int Cout, int K) {
    const int co = blockIdx.x;
    const float s = gamma[co] * rsqrtf(var[co] + eps);
    if (threadIdx.x == 0) bias[co] = beta[co] - mean[co] * s;
    const float* wr = w + (size_t)co * K;
    T* o = wf + (size_t)co * K;
    for (int i = threadIdx.x; i < K; i += 256) {
        if constexpr (sizeof(T) == 2) o[i] = f2bf(wr[i] * s); else o[i] = wr[i] * s;
    }
}

extern "C" int cr_fold_bn(cr_ctx* ctx, const float* w, const float* gamma, const float* beta, const float* mean,
                          const float* var, float eps, void* wf, float* bias, int Cout, int K, int act_f32) {
    CR_CHECK_ARG(ctx && w && gamma && beta && mean && var && wf && bias && Cout > 0 && K > 0, "cr_fold_bn: bad args");
    if (act_f32)
        hipLaunchKernelGGL(k_fold_bn<float>, dim3((unsigned)Cout), dim3(256), 0, ctx->stream, w, gamma, beta, mean, var, eps,
                           (float*)wf, bias, Cout, K);
    else
        hipLaunchKernelGGL(k_fold_bn<u16>, dim3((unsigned)Cout), dim3(256), 0, ctx->stream, w, gamma, beta, mean, var, eps,
                           (u16*)wf, bias, Cout, K);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_cast_f32_to_bf16(cr_ctx* ctx, const float* src, void* dst, int64_t n) {
    CR_CHECK_ARG(ctx && n >= 0, "cr_cast_f32_to_bf16: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(src && dst, "cr_cast_f32_to_bf16: NULL pointer");
    CR_CHECK_ARG((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 7) == 0, "cr_cast_f32_to_bf16: misaligned");
    hipLaunchKernelGGL(k_cast_f32_bf16, dim3((unsigned)cr_cdiv(cr_cdiv(n, 4), 256)), dim3(256), 0, ctx->stream, src,
                       (u16*)dst, n);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

template <typename T>
__global__ void k_weight_transpose(const float* __restrict__ w, T* __restrict__ wt, int Cout, int taps, int Cin) {
    // wt[c][t][k] = w[k][t][c]
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)Cout * taps * Cin;
    if (i >= n) return;
    const int k = (int)(i % Cout);
    const int t = (int)((i / Cout) % taps);
    const int c = (int)(i / ((int64_t)Cout * taps));
    const float v = w[((int64_t)k * taps + t) * Cin + c];
    if constexpr (sizeof(T) == 2) wt[i] = f2bf(v); else wt[i] = v;
}

extern "C" int cr_weight_transpose(cr_ctx* ctx, const float* w, void* wt, int Cout, int ks, int Cin, int act_f32) {
    CR_CHECK_ARG(ctx && w && wt && Cout > 0 && Cin > 0 && ks > 0, "cr_weight_transpose: bad args");
    const int64_t n = (int64_t)Cout * ks * ks * Cin;
    if (act_f32)
        hipLaunchKernelGGL(k_weight_transpose<float>, dim3((unsigned)cr_cdiv(n, 256)), dim3(256), 0, ctx->stream, w, (float*)wt,
                           Cout, ks * ks, Cin);
    else
        hipLaunchKernelGGL(k_weight_transpose<u16>, dim3((unsigned)cr_cdiv(n, 256)), dim3(256), 0, ctx->stream, w, (u16*)wt,
                           Cout, ks * ks, Cin);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// Multi-tensor weight preparation: ALL conv weights of the model (views into the optimizer's flat f32 parameter
// buffer) are recast to bf16 and re-laid for bwd-data by ONE launch per step instead of two tiny launches per layer.
// A block handles a 32 (cout) x 32 (cin) tile of one filter tap of one tensor (host-built tile table), writing the
// bf16 copy in place order and the transposed copy through an LDS tile (both sides coalesced).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_weights_prepare(const float* __restrict__ src_base, T* __restrict__ dst_base,
                                                         T* __restrict__ dstT_base, const cr_wdesc* __restrict__ descs,
                                                         const int4* __restrict__ tiles) {
    __shared__ T tile[32][33];
    const int4 tl = tiles[blockIdx.x];                   // (tensor, tap, cout0, cin0)
    const cr_wdesc d = descs[tl.x];
    const int r = tl.y, o0 = tl.z, c0 = tl.w;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const float* src = src_base + d.src_off;
    T* dst = dst_base ? dst_base + d.dst_off : nullptr;  // f32 mode: the master weights ARE the forward operand (no copy)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = o0 + ty + 8 * i, c = c0 + tx;
        T v = 0;
        if (o < d.Cout && c < d.Cin) {
            const int64_t e = ((int64_t)o * d.KK + r) * d.Cin + c;
            if constexpr (sizeof(T) == 2) v = f2bf(src[e]); else v = src[e];
            if (dst) dst[e] = v;
        }
        tile[ty + 8 * i][tx] = v;
    }
    if (!d.need_T) return;                               // block-uniform
    __syncthreads();
    T* dstT = dstT_base + d.dstT_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, o = o0 + tx;
        if (o < d.Cout && c < d.Cin) dstT[((int64_t)c * d.KK + r) * d.Cout + o] = tile[tx][ty + 8 * i];
    }
}

extern "C" int cr_weights_prepare(cr_ctx* ctx, const float* src_base, void* dst_base, void* dstT_base,
                                  const cr_wdesc* descs_dev, const int* tiles_dev, int ntiles, int act_f32) {
    CR_CHECK_ARG(ctx && ntiles >= 0, "cr_weights_prepare: bad args");
    if (ntiles == 0) return CR_OK;
    CR_CHECK_ARG(src_base && (dst_base || act_f32) && dstT_base && descs_dev && tiles_dev, "cr_weights_prepare: NULL pointer");
    if (act_f32)
        hipLaunchKernelGGL(k_weights_prepare<float>, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, src_base,
                           (float*)dst_base, (float*)dstT_base, descs_dev, (const int4*)tiles_dev);
    else
        hipLaunchKernelGGL(k_weights_prepare<u16>, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, src_base,
                           (u16*)dst_base, (u16*)dstT_base, descs_dev, (const int4*)tiles_dev);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// FC weights of the RoI heads.  The first FC consumes RoI features flattened in (h,w,c) order (NHWC) while the
// checkpoint keeps its columns in (c,h,w) order (roi_heads.py:2160-2204, cube_head.py:152-202 flatten NCHW):
//   cr_fc_weight_prepare   f32 (O, C, HW) -> bf16 (O, HW, C)   (HW = 1: plain cast), once per optimizer step
//   cr_fc_grad_accum       f32 (O, C, HW) += bf16 (O, HW, C)    the weight gradient lands in the flat gradient
// A block handles one output row and a tile of 64 input channels: both sides move whole contiguous runs.
// ---------------------------------------------------------------------------
#define FC_CT 64
template <typename T>
__global__ __launch_bounds__(256) void k_fc_weight_prepare(const float* __restrict__ src, T* __restrict__ dst, int C, int HW) {
    extern __shared__ float s_fc[];                       // [FC_CT][HW + 1]
    const int o = blockIdx.y, c0 = blockIdx.x * FC_CT;
    const int nc = min(FC_CT, C - c0), P1 = HW + 1;
    const float* in = src + ((size_t)o * C + c0) * HW;    // nc * HW contiguous floats
    for (int i = threadIdx.x; i < nc * HW; i += 256) s_fc[(i / HW) * P1 + (i % HW)] = in[i];
    __syncthreads();
    T* out = dst + (size_t)o * HW * C + c0;
    for (int i = threadIdx.x; i < nc * HW; i += 256) {
        const int hw = i / nc, c = i - hw * nc;
        if constexpr (sizeof(T) == 2) out[(size_t)hw * C + c] = f2bf(s_fc[c * P1 + hw]);
        else out[(size_t)hw * C + c] = s_fc[c * P1 + hw];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_fc_grad_accum(const T* __restrict__ g, float* __restrict__ acc, int C, int HW) {
    extern __shared__ float s_fc[];
    const int o = blockIdx.y, c0 = blockIdx.x * FC_CT;
    const int nc = min(FC_CT, C - c0), P1 = HW + 1;
    const T* in = g + (size_t)o * HW * C + c0;
    for (int i = threadIdx.x; i < nc * HW; i += 256) {
        const int hw = i / nc, c = i - hw * nc;
        s_fc[c * P1 + hw] = load1<T>(in, (size_t)hw * C + c);
    }
    __syncthreads();
    float* out = acc + ((size_t)o * C + c0) * HW;
    for (int i = threadIdx.x; i < nc * HW; i += 256) out[i] += s_fc[(i / HW) * P1 + (i % HW)];
}

__global__ void k_axpy_f32_f32(const float* __restrict__ g, float* __restrict__ acc, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(g + i);
        float4 a = *reinterpret_cast<float4*>(acc + i);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        *reinterpret_cast<float4*>(acc + i) = a;
    } else {
        for (int64_t j = i; j < n; ++j) acc[j] += g[j];
    }
}

__global__ void k_axpy_bf16_f32(const u16* __restrict__ g, float* __restrict__ acc, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const uint2 v = *reinterpret_cast<const uint2*>(g + i);
        float4 a = *reinterpret_cast<float4*>(acc + i);
        a.x += bf2f((u16)(v.x & 0xffff)); a.y += bf2f((u16)(v.x >> 16));
        a.z += bf2f((u16)(v.y & 0xffff)); a.w += bf2f((u16)(v.y >> 16));
        *reinterpret_cast<float4*>(acc + i) = a;
    } else {
        for (int64_t j = i; j < n; ++j) acc[j] += bf2f(g[j]);
    }
}

extern "C" int cr_fc_weight_prepare(cr_ctx* ctx, const float* w, void* wb, int O, int C, int HW, int act_f32) {
    CR_CHECK_ARG(ctx && w && wb && O > 0 && C > 0 && HW > 0 && HW <= 1024, "cr_fc_weight_prepare: bad args");
    if (HW == 1) {
        if (!act_f32) return cr_cast_f32_to_bf16(ctx, w, wb, (int64_t)O * C);
        CR_HIP(hipMemcpyAsync(wb, w, (size_t)O * C * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return CR_OK;
    }
    if (act_f32)
        hipLaunchKernelGGL(k_fc_weight_prepare<float>, dim3((unsigned)cr_cdiv(C, FC_CT), O), dim3(256),
                           FC_CT * (HW + 1) * sizeof(float), ctx->stream, w, (float*)wb, C, HW);
    else
        hipLaunchKernelGGL(k_fc_weight_prepare<u16>, dim3((unsigned)cr_cdiv(C, FC_CT), O), dim3(256),
                           FC_CT * (HW + 1) * sizeof(float), ctx->stream, w, (u16*)wb, C, HW);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_fc_grad_accum(cr_ctx* ctx, const void* g, float* acc, int O, int C, int HW, int act_f32) {
    CR_CHECK_ARG(ctx && g && acc && O > 0 && C > 0 && HW > 0 && HW <= 1024, "cr_fc_grad_accum: bad args");
    if (HW == 1) {
        const int64_t n = (int64_t)O * C;
        CR_CHECK_ARG((((uintptr_t)g) & 7) == 0 && (((uintptr_t)acc) & 15) == 0, "cr_fc_grad_accum: misaligned");
        if (act_f32)
            hipLaunchKernelGGL(k_axpy_f32_f32, dim3((unsigned)cr_cdiv(cr_cdiv(n, 4), 256)), dim3(256), 0, ctx->stream,
                               (const float*)g, acc, n);
        else
            hipLaunchKernelGGL(k_axpy_bf16_f32, dim3((unsigned)cr_cdiv(cr_cdiv(n, 4), 256)), dim3(256), 0, ctx->stream,
                               (const u16*)g, acc, n);
    } else if (act_f32) {
        hipLaunchKernelGGL(k_fc_grad_accum<float>, dim3((unsigned)cr_cdiv(C, FC_CT), O), dim3(256),
                           FC_CT * (HW + 1) * sizeof(float), ctx->stream, (const float*)g, acc, C, HW);
    } else {
        hipLaunchKernelGGL(k_fc_grad_accum<u16>, dim3((unsigned)cr_cdiv(C, FC_CT), O), dim3(256),
                           FC_CT * (HW + 1) * sizeof(float), ctx->stream, (const u16*)g, acc, C, HW);
    }
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// BatchNorm2d (training mode, per-GPU statistics: dla.py:17 BatchNorm = nn.BatchNorm2d)
// ---------------------------------------------------------------------------
// finalize: per-tile partials [nparts][2][C] -> mean / invstd (+ running stats update, momentum, unbiased var).
// One workgroup per channel; strided serial sums + a fixed-order LDS tree, in double: reproducible and accurate
// (no E[x^2]-E[x]^2 cancellation at float precision).
__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ stats, int nparts, int C, float count,
                                                     float eps, float momentum, float* __restrict__ mean_invstd,
                                                     float* __restrict__ running_mean, float* __restrict__ running_var) {
    __shared__ double ss[256], sq[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int r = t; r < nparts; r += 256) {
        s += (double)stats[(size_t)r * 2 * C + c];
        q += (double)stats[(size_t)r * 2 * C + C + c];
    }
    ss[t] = s; sq[t] = q;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) { ss[t] += ss[t + off]; sq[t] += sq[t + off]; }
        __syncthreads();
    }
    if (t == 0) {
        const double mean = ss[0] / (double)count;
        double var = sq[0] / (double)count - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_invstd[c] = (float)mean;
        mean_invstd[C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
}

// y = relu?( (x - mean) * invstd * gamma + beta (+ residual) ), 8 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void k_bn_apply(const T* __restrict__ x, const float* __restrict__ mean_invstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const T* __restrict__ res, T* __restrict__ y, int64_t M, int C,
                                                  int relu) {
    const int cg = C >> 3;
    const int64_t total = M * cg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cg) << 3;
        float xv[8], rv[8], o[8];
        load8<T>(x, (size_t)i * 8, xv);
        if (res) load8<T>(res, (size_t)i * 8, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            float t = (xv[e] - mean_invstd[c]) * mean_invstd[C + c] * gamma[c] + beta[c];
            if (res) t += rv[e];
            o[e] = relu ? fmaxf(t, 0.f) : t;
        }
        store8<T>(y, (size_t)i * 8, o);
    }
}

// finalize + apply in ONE launch for the layers with few statistics rows (the 64x64 and smaller maps: 30 of DLA34's 39
// BatchNorms): block (px chunk, channel slice of 32) first reduces the partial rows of ITS 32 channels (8 row lanes per
// channel, double accumulation, fixed order: every block of a slice computes the same bits), then normalises its pixels.
// The blocks of px chunk 0 publish mean / invstd for backward and update the running statistics.
template <typename T>
__global__ __launch_bounds__(256) void k_bn_apply_fused(const T* __restrict__ x, const float* __restrict__ stats, int nparts,
                                                        float count, float eps, float momentum, float* __restrict__ mean_invstd,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const T* __restrict__ res, T* __restrict__ y, int64_t M, int C, int relu,
                                                        int px_per_block) {
    __shared__ double sd[2][8][32];
    __shared__ float sc[2][32];                          // mean, invstd of this slice
    const int t = threadIdx.x, ch = t & 31, rl = t >> 5;
    const int cbase = blockIdx.y * 32, c = cbase + ch;
    {
        double s = 0.0, q = 0.0;
        for (int r = rl; r < nparts; r += 8) {
            s += (double)stats[(size_t)r * 2 * C + c];
            q += (double)stats[(size_t)r * 2 * C + C + c];
        }
        sd[0][rl][ch] = s; sd[1][rl][ch] = q;
    }
    __syncthreads();
    if (t < 32) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) { s += sd[0][r][t]; q += sd[1][r][t]; }
        const double mean = s / (double)count;
        double var = q / (double)count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mean, isf = (float)(1.0 / sqrt(var + (double)eps));
        sc[0][t] = mf;
        sc[1][t] = isf;
        if (blockIdx.x == 0) {
            mean_invstd[cbase + t] = mf;
            mean_invstd[C + cbase + t] = isf;
            if (running_mean) {
                const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
                running_mean[cbase + t] = (float)((1.0 - momentum) * running_mean[cbase + t] + momentum * mean);
                running_var[cbase + t] = (float)((1.0 - momentum) * running_var[cbase + t] + momentum * unbiased);
            }
        }
    }
    __syncthreads();
    const int cgl = t & 3;                               // 8-channel group inside the slice
    float mu[8], is8[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mu[e] = sc[0][cgl * 8 + e]; is8[e] = sc[1][cgl * 8 + e];
        ga[e] = gamma[cbase + cgl * 8 + e]; be[e] = beta[cbase + cgl * 8 + e];
    }
    const int64_t m0 = (int64_t)blockIdx.x * px_per_block, m1 = m0 + px_per_block < M ? m0 + px_per_block : M;
    for (int64_t m = m0 + (t >> 2); m < m1; m += 64) {
        const size_t i8 = (size_t)m * C + cbase + cgl * 8;
        float xv[8], rv[8], o[8];
        load8<T>(x, i8, xv);
        if (res) load8<T>(res, i8, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (xv[e] - mu[e]) * is8[e] * ga[e] + be[e];        // as k_bn_apply
            if (res) v += rv[e];
            o[e] = relu ? fmaxf(v, 0.f) : v;
        }
        store8<T>(y, i8, o);
    }
}

static int bn_fuse_rows() {
    static const int v = env_int("CR_BN_FUSE_ROWS", 256);
    return v;
}

extern "C" int cr_bn_fwd(cr_ctx* ctx, const void* x, const float* stats, int nparts, const float* gamma,
                         const float* beta, const void* residual, void* y, int64_t M, int C, int relu, float eps,
                         float momentum, float* mean_invstd, float* running_mean, float* running_var, int act_f32) {
    CR_CHECK_ARG(ctx && x && stats && gamma && beta && y && mean_invstd, "cr_bn_fwd: NULL pointer");
    CR_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0 && nparts > 0, "cr_bn_fwd: bad dims M=%lld C=%d", (long long)M, C);
    if (C % 32 == 0 && nparts <= bn_fuse_rows()) {
        // ~1024 blocks; a block's statistics pass reads nparts x 64 floats
        int64_t chunks = 1024 / (C / 32);
        if (chunks < 1) chunks = 1;
        int64_t ppb = cr_cdiv(M, chunks);
        ppb = (ppb + 63) / 64 * 64;
        const dim3 grid((unsigned)cr_cdiv(M, ppb), (unsigned)(C / 32));
        if (act_f32)
            hipLaunchKernelGGL(k_bn_apply_fused<float>, grid, dim3(256), 0, ctx->stream, (const float*)x, stats, nparts, (float)M,
                               eps, momentum, mean_invstd, running_mean, running_var, gamma, beta, (const float*)residual,
                               (float*)y, M, C, relu, (int)ppb);
        else
            hipLaunchKernelGGL(k_bn_apply_fused<u16>, grid, dim3(256), 0, ctx->stream, (const u16*)x, stats, nparts, (float)M,
                               eps, momentum, mean_invstd, running_mean, running_var, gamma, beta, (const u16*)residual,
                               (u16*)y, M, C, relu, (int)ppb);
        CR_LAUNCH_CHECK();
        return CR_OK;
    }
    hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)C), dim3(256), 0, ctx->stream, stats, nparts, C, (float)M, eps,
                       momentum, mean_invstd, running_mean, running_var);
    CR_LAUNCH_CHECK();
    const int64_t total = M * (C >> 3);
    const unsigned grid = (unsigned)(cr_cdiv(total, 256) < 4096 ? cr_cdiv(total, 256) : 4096);
    if (act_f32)
        hipLaunchKernelGGL(k_bn_apply<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)x, mean_invstd, gamma, beta,
                           (const float*)residual, (float*)y, M, C, relu);
    else
        hipLaunchKernelGGL(k_bn_apply<u16>, dim3(grid), dim3(256), 0, ctx->stream, (const u16*)x, mean_invstd, gamma, beta,
                           (const u16*)residual, (u16*)y, M, C, relu);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// backward reduce: g = dy * (relu ? out > 0 : 1);  partial[block][0][c] = sum g ; partial[block][1][c] = sum g*xhat
#define BNB_MAXBLOCKS 1024
template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const T* __restrict__ dy, const T* __restrict__ out,
                                                       const T* __restrict__ x, const float* __restrict__ mean_invstd,
                                                       float* __restrict__ partial, int64_t M, int C, int relu) {
    extern __shared__ float s_acc[];        // [rows_per_block][2][C]  (= 4096 floats for every supported C)
    const int cg = C >> 3;
    const int mycg = threadIdx.x % cg, myrow = threadIdx.x / cg;
    const int rows_per_block = blockDim.x / cg;
    const int c0 = mycg << 3;
    float a[8], b[8], mu[8], is[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; b[e] = 0.f; mu[e] = mean_invstd[c0 + e]; is[e] = mean_invstd[C + c0 + e]; }
    for (int64_t m = (int64_t)blockIdx.x * rows_per_block + myrow; m < M; m += (int64_t)gridDim.x * rows_per_block) {
        const size_t i8 = (size_t)(m * cg + mycg) * 8;
        float dv[8], xv[8], ov[8];
        load8<T>(dy, i8, dv);
        load8<T>(x, i8, xv);
        if (relu) load8<T>(out, i8, ov);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gq = (!relu || ov[e] > 0.f) ? dv[e] : 0.f;
            const float xh = (xv[e] - mu[e]) * is[e];
            a[e] += gq;
            b[e] += gq * xh;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s_acc[(myrow * 2 + 0) * C + c0 + e] = a[e]; s_acc[(myrow * 2 + 1) * C + c0 + e] = b[e]; }
    __syncthreads();
    float* dst = partial + (size_t)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        const int which = i / C, c = i - which * C;
        float s = 0.f;
        for (int r = 0; r < rows_per_block; ++r) s += s_acc[(r * 2 + which) * C + c];     // fixed order
        dst[i] = s;
    }
}

// partial [nparts][2][C] -> sums [2][C]; dgamma/dbeta accumulated.  One workgroup per channel, fixed-order tree.
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ partial, int nparts, int C,
                                                         float* __restrict__ sums, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta) {
    __shared__ double ss[256], sq[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int r = t; r < nparts; r += 256) {
        s += (double)partial[(size_t)r * 2 * C + c];
        q += (double)partial[(size_t)r * 2 * C + C + c];
    }
    ss[t] = s; sq[t] = q;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) { ss[t] += ss[t + off]; sq[t] += sq[t + off]; }
        __syncthreads();
    }
    if (t == 0) {
        sums[c] = (float)ss[0];
        sums[C + c] = (float)sq[0];
        dbeta[c] += (float)ss[0];
        dgamma[c] += (float)sq[0];
    }
}

// dx = gamma*invstd*(g - sum_g/M - xhat*sum_gx/M); also writes g (the masked grad) for the residual branch
template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const T* __restrict__ dy, const T* __restrict__ out,
                                                      const T* __restrict__ x, const float* __restrict__ mean_invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ sums,
                                                      T* __restrict__ dx, T* __restrict__ dres, int64_t M, int C,
                                                      int relu) {
    const int cg = C >> 3;
    const float invM = 1.f / (float)M;
    const int64_t total = M * cg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cg) << 3;
        const size_t i8 = (size_t)i * 8;
        float dv[8], xv[8], ov[8], vd[8], vg[8];
        load8<T>(dy, i8, dv);
        load8<T>(x, i8, xv);
        if (relu) load8<T>(out, i8, ov);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            const float gq = (!relu || ov[e] > 0.f) ? dv[e] : 0.f;
            const float is = mean_invstd[C + c];
            const float xh = (xv[e] - mean_invstd[c]) * is;
            vd[e] = gamma[c] * is * (gq - sums[c] * invM - xh * sums[C + c] * invM);
            vg[e] = gq;
        }
        store8<T>(dx, i8, vd);
        if (dres) store8<T>(dres, i8, vg);
    }
}

// finalize + apply of the backward in one launch (same scheme as k_bn_apply_fused): the blocks of px chunk 0 accumulate
// dgamma / dbeta
template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_apply_fused(const T* __restrict__ dy, const T* __restrict__ out,
                                                            const T* __restrict__ x, const float* __restrict__ mean_invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ partial,
                                                            int nparts, T* __restrict__ dx, T* __restrict__ dres,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t M,
                                                            int C, int relu, int px_per_block) {
    __shared__ double sd[2][8][32];
    __shared__ float sc[4][32];                          // mean, invstd * gamma, sum_g / M, invstd * sum_gx / M
    const int t = threadIdx.x, ch = t & 31, rl = t >> 5;
    const int cbase = blockIdx.y * 32, c = cbase + ch;
    {
        double s = 0.0, q = 0.0;
        for (int r = rl; r < nparts; r += 8) {
            s += (double)partial[(size_t)r * 2 * C + c];
            q += (double)partial[(size_t)r * 2 * C + C + c];
        }
        sd[0][rl][ch] = s; sd[1][rl][ch] = q;
    }
    __syncthreads();
    if (t < 32) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) { s += sd[0][r][t]; q += sd[1][r][t]; }
        const float sg = (float)s, sgx = (float)q, invM = 1.f / (float)M;
        const float is = mean_invstd[C + cbase + t];
        sc[0][t] = mean_invstd[cbase + t];
        sc[1][t] = is;
        sc[2][t] = sg * invM;
        sc[3][t] = sgx * invM;
        if (blockIdx.x == 0) { dbeta[cbase + t] += sg; dgamma[cbase + t] += sgx; }
    }
    __syncthreads();
    const int cgl = t & 3;
    float mu[8], is8[8], ga[8], k1[8], k2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int cc = cgl * 8 + e;
        mu[e] = sc[0][cc]; is8[e] = sc[1][cc]; ga[e] = gamma[cbase + cc]; k1[e] = sc[2][cc]; k2[e] = sc[3][cc];
    }
    const int64_t m0 = (int64_t)blockIdx.x * px_per_block, m1 = m0 + px_per_block < M ? m0 + px_per_block : M;
    for (int64_t m = m0 + (t >> 2); m < m1; m += 64) {
        const size_t i8 = (size_t)m * C + cbase + cgl * 8;
        float dv[8], xv[8], ov[8], vd[8], vg[8];
        load8<T>(dy, i8, dv);
        load8<T>(x, i8, xv);
        if (relu) load8<T>(out, i8, ov);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gq = (!relu || ov[e] > 0.f) ? dv[e] : 0.f;
            const float xh = (xv[e] - mu[e]) * is8[e];
            vd[e] = ga[e] * is8[e] * (gq - k1[e] - xh * k2[e]);
            vg[e] = gq;
        }
        store8<T>(dx, i8, vd);
        if (dres) store8<T>(dres, i8, vg);
    }
}

// sums: f32 workspace of (CR_BN_BWD_WS_ROWS) x 2 x C floats (partials + the reduced [2][C] in the last row).
// dgamma/dbeta are ACCUMULATED (+=).  No atomics: bitwise reproducible.
extern "C" int cr_bn_bwd(cr_ctx* ctx, const void* dy, const void* out, const void* x, const float* mean_invstd,
                         const float* gamma, float* sums, void* dx, void* dres, float* dgamma, float* dbeta,
                         int64_t M, int C, int relu, int act_f32) {
    CR_CHECK_ARG(ctx && dy && x && mean_invstd && gamma && sums && dx && dgamma && dbeta, "cr_bn_bwd: NULL pointer");
    CR_CHECK_ARG(!relu || out, "cr_bn_bwd: relu needs the forward output");
    CR_CHECK_ARG(M > 0 && C % 8 == 0 && C <= 2048 && 256 % (C >> 3) == 0, "cr_bn_bwd: unsupported C=%d", C);
    const int rows_per_block = 256 / (C >> 3);
    int64_t nb = cr_cdiv(M, (int64_t)rows_per_block * 8);
    if (nb > BNB_MAXBLOCKS) nb = BNB_MAXBLOCKS;
    if (nb < 1) nb = 1;
    float* reduced = sums + (size_t)BNB_MAXBLOCKS * 2 * C;
    const size_t shm = sizeof(float) * rows_per_block * 2 * C;
    if (act_f32)
        hipLaunchKernelGGL(k_bn_bwd_reduce<float>, dim3((unsigned)nb), dim3(256), shm, ctx->stream, (const float*)dy,
                           (const float*)out, (const float*)x, mean_invstd, sums, M, C, relu);
    else
        hipLaunchKernelGGL(k_bn_bwd_reduce<u16>, dim3((unsigned)nb), dim3(256), shm, ctx->stream, (const u16*)dy,
                           (const u16*)out, (const u16*)x, mean_invstd, sums, M, C, relu);
    CR_LAUNCH_CHECK();
    if (C % 32 == 0 && nb <= bn_fuse_rows()) {
        int64_t chunks = 1024 / (C / 32);
        if (chunks < 1) chunks = 1;
        int64_t ppb = cr_cdiv(M, chunks);
        ppb = (ppb + 63) / 64 * 64;
        const dim3 grid((unsigned)cr_cdiv(M, ppb), (unsigned)(C / 32));
        if (act_f32)
            hipLaunchKernelGGL(k_bn_bwd_apply_fused<float>, grid, dim3(256), 0, ctx->stream, (const float*)dy, (const float*)out,
                               (const float*)x, mean_invstd, gamma, sums, (int)nb, (float*)dx, (float*)dres, dgamma, dbeta, M, C,
                               relu, (int)ppb);
        else
            hipLaunchKernelGGL(k_bn_bwd_apply_fused<u16>, grid, dim3(256), 0, ctx->stream, (const u16*)dy, (const u16*)out,
                               (const u16*)x, mean_invstd, gamma, sums, (int)nb, (u16*)dx, (u16*)dres, dgamma, dbeta, M, C, relu,
                               (int)ppb);
        CR_LAUNCH_CHECK();
        return CR_OK;
    }
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((unsigned)C), dim3(256), 0, ctx->stream, sums, (int)nb, C, reduced, dgamma,
                       dbeta);
    CR_LAUNCH_CHECK();
    const int64_t total = M * (C >> 3);
    const unsigned grid = (unsigned)(cr_cdiv(total, 256) < 4096 ? cr_cdiv(total, 256) : 4096);
    if (act_f32)
        hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)dy, (const float*)out,
                           (const float*)x, mean_invstd, gamma, reduced, (float*)dx, (float*)dres, M, C, relu);
    else
        hipLaunchKernelGGL(k_bn_bwd_apply<u16>, dim3(grid), dim3(256), 0, ctx->stream, (const u16*)dy, (const u16*)out,
                           (const u16*)x, mean_invstd, gamma, reduced, (u16*)dx, (u16*)dres, M, C, relu);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// bias gradient: out[c] += sum_m x[m][c], x bf16 or f32 [M][C]; two fixed-order stages (no atomics, no memset)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_colsum_partial(const T* __restrict__ x, int64_t M, int C, int rows_per_block,
                                                        float* __restrict__ partial) {
    // thread -> column c = tid % C' ... simple mapping: each thread owns columns c = tid, tid+256, ...
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int64_t r = r0; r < r1; ++r) {
            if (sizeof(T) == 2) s += bf2f(((const u16*)x)[r * C + c]);
            else s += ((const float*)x)[r * C + c];
        }
        partial[(size_t)blockIdx.x * C + c] = s;
    }
}

__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ partial, int nparts, int C,
                                                      float* __restrict__ out) {
    __shared__ double ss[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s = 0.0;
    for (int r = t; r < nparts; r += 256) s += (double)partial[(size_t)r * C + c];
    ss[t] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) ss[t] += ss[t + off];
        __syncthreads();
    }
    if (t == 0) out[c] += (float)ss[0];
}

// ws: f32 workspace of 1024*C floats.  out is ACCUMULATED.
extern "C" int cr_colsum_accum(cr_ctx* ctx, const void* x, int is_f32, int64_t M, int C, float* ws, float* out) {
    CR_CHECK_ARG(ctx && x && ws && out && M > 0 && C > 0, "cr_colsum_accum: bad args");
    int64_t rows = cr_cdiv(M, 1024);
    if (rows < 8) rows = 8;
    const int nb = (int)cr_cdiv(M, rows);
    if (is_f32)
        hipLaunchKernelGGL((k_colsum_partial<float>), dim3(nb), dim3(256), 0, ctx->stream, (const float*)x, M, C, (int)rows, ws);
    else
        hipLaunchKernelGGL((k_colsum_partial<u16>), dim3(nb), dim3(256), 0, ctx->stream, (const u16*)x, M, C, (int)rows, ws);
    CR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_colsum_final, dim3(C), dim3(256), 0, ctx->stream, ws, nb, C, out);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// pooling / resampling / elementwise, NHWC (bf16 or f32 storage), 8 channels per thread
// ---------------------------------------------------------------------------
#define CR_DISPATCH_T(act_f32, KERNEL, grid, block, shm, stream, ...)                                   \
    do {                                                                                                \
        if (act_f32) hipLaunchKernelGGL((KERNEL<float>), grid, block, shm, stream, __VA_ARGS__);        \
        else hipLaunchKernelGGL((KERNEL<u16>), grid, block, shm, stream, __VA_ARGS__);                  \
    } while (0)

// ReLU backward: g = y > 0 ? dy : 0 (y = the ReLU's OUTPUT), 4 elements per thread; the tail by the last thread
template <typename T>
__global__ __launch_bounds__(256) void k_relu_bwd(const void* __restrict__ yv, const void* __restrict__ dyv, void* __restrict__ gv, int64_t n) {
    const T* y = (const T*)yv; const T* dy = (const T*)dyv; T* g = (T*)gv;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        if constexpr (sizeof(T) == 4) {
            const float4 a = *reinterpret_cast<const float4*>(y + i), d = *reinterpret_cast<const float4*>(dy + i);
            *reinterpret_cast<float4*>(g + i) = make_float4(a.x > 0.f ? d.x : 0.f, a.y > 0.f ? d.y : 0.f, a.z > 0.f ? d.z : 0.f, a.w > 0.f ? d.w : 0.f);
        } else {
            const uint2 a = *reinterpret_cast<const uint2*>(y + i), d = *reinterpret_cast<const uint2*>(dy + i);
            uint2 o;        // bf16: positive <=> sign bit clear and not zero
            auto pos = [](unsigned h) { return (h & 0x8000u) == 0 && (h & 0x7fffu) != 0; };
            o.x = (pos(a.x & 0xffffu) ? (d.x & 0xffffu) : 0u) | (pos(a.x >> 16) ? (d.x & 0xffff0000u) : 0u);
            o.y = (pos(a.y & 0xffffu) ? (d.y & 0xffffu) : 0u) | (pos(a.y >> 16) ? (d.y & 0xffff0000u) : 0u);
            *reinterpret_cast<uint2*>(g + i) = o;
        }
    } else {
        for (int64_t j = i; j < n; ++j) {
            if constexpr (sizeof(T) == 4) g[j] = y[j] > 0.f ? dy[j] : 0.f;
            else g[j] = ((y[j] & 0x8000u) == 0 && (y[j] & 0x7fffu) != 0) ? dy[j] : (u16)0;
        }
    }
}

extern "C" int cr_relu_bwd(cr_ctx* ctx, const void* y, const void* dy, void* g, int64_t n, int act_f32) {
    CR_CHECK_ARG(ctx && n >= 0, "cr_relu_bwd: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(y && dy && g && ((((uintptr_t)y) | ((uintptr_t)dy) | ((uintptr_t)g)) & 15) == 0, "cr_relu_bwd: NULL or misaligned pointer");
    CR_DISPATCH_T(act_f32, k_relu_bwd, dim3((unsigned)cr_cdiv(cr_cdiv(n, 4), 256)), dim3(256), 0, ctx->stream, y, dy, g, n);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// window = 2: MaxPool2d(2,2) (dla.py:208); window = 1: max_pool2d(k=1,s=2) = subsample (dla.py:474)
template <typename T>
__global__ void k_pool_fwd(const void* __restrict__ xv, void* __restrict__ yv, int N, int H, int W, int C, int window) {
    const T* __restrict__ x = (const T*)xv;
    T* __restrict__ y = (T*)yv;
    const int Ho = H / 2, Wo = W / 2, cg = C >> 3;
    const int64_t total = (int64_t)N * Ho * Wo * cg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg);
    const int wo = (int)((i / cg) % Wo);
    const int ho = (int)((i / ((int64_t)cg * Wo)) % Ho);
    const int n = (int)(i / ((int64_t)cg * Wo * Ho));
    const size_t b = (((size_t)(n * H + 2 * ho) * W + 2 * wo) * C) + c * 8;
    float m[8], t[8];
    load8<T>(x, b, m);
    if (window == 2) {
        const size_t offs[3] = {(size_t)C, (size_t)W * C, (size_t)W * C + C};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            load8<T>(x, b + offs[q], t);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = (t[e] > m[e] || t[e] != t[e]) ? t[e] : m[e];
        }
    }
    store8<T>(y, (size_t)i * 8, m);
}

// routes dy to the FIRST maximal element of each window in scan order (PyTorch's tie rule)
template <typename T>
__global__ void k_pool_bwd(const void* __restrict__ xv, const void* __restrict__ dyv, void* __restrict__ dxv, int N, int H,
                           int W, int C, int window, const void* __restrict__ accv) {
    const T* __restrict__ x = (const T*)xv;
    const T* __restrict__ dy = (const T*)dyv;
    const T* __restrict__ acc = (const T*)accv;          // optional: dx = routed dy + acc (gradient fan-in, hipops._GradSlot)
    T* __restrict__ dx = (T*)dxv;
    const int Ho = H / 2, Wo = W / 2, cg = C >> 3;
    const int64_t total = (int64_t)N * Ho * Wo * cg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg);
    const int wo = (int)((i / cg) % Wo);
    const int ho = (int)((i / ((int64_t)cg * Wo)) % Ho);
    const int n = (int)(i / ((int64_t)cg * Wo * Ho));
    const size_t base = (((size_t)(n * H + 2 * ho) * W + 2 * wo) * C) + c * 8;
    const size_t offs[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    float g[8];
    load8<T>(dy, (size_t)i * 8, g);
    const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (window == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = q == 0 ? g[e] : z[e];
            if (acc) {
                float a8[8];
                load8<T>(acc, base + offs[q], a8);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += a8[e];
            }
            store8<T>(dx, base + offs[q], o);
        }
        return;
    }
    float v[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) load8<T>(x, base + offs[q], v[q]);
    int am[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int a = 0;
        float m = v[0][e];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (v[q][e] > m || v[q][e] != v[q][e]) { m = v[q][e]; a = q; }
        am[e] = a;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = am[e] == q ? g[e] : 0.f;
        if (acc) {
            float a8[8];
            load8<T>(acc, base + offs[q], a8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += a8[e];
        }
        store8<T>(dx, base + offs[q], o);
    }
}

extern "C" int cr_pool2x_fwd(cr_ctx* ctx, const void* x, void* y, int N, int H, int W, int C, int window, int act_f32) {
    CR_CHECK_ARG(ctx && x && y, "cr_pool2x_fwd: NULL pointer");
    CR_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && (window == 1 || window == 2), "cr_pool2x_fwd: bad dims");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    if (total == 0) return CR_OK;
    CR_DISPATCH_T(act_f32, k_pool_fwd, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, x, y, N, H, W, C, window);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_pool2x_bwd(cr_ctx* ctx, const void* x, const void* dy, void* dx, int N, int H, int W, int C,
                             int window, int act_f32, const void* accumulate) {
    CR_CHECK_ARG(ctx && x && dy && dx, "cr_pool2x_bwd: NULL pointer");
    CR_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && (window == 1 || window == 2), "cr_pool2x_bwd: bad dims");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    if (total == 0) return CR_OK;
    CR_DISPATCH_T(act_f32, k_pool_bwd, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, x, dy, dx, N, H, W, C,
                  window, accumulate);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the torchvision ResNet stem (resnet.py:33,49 of the reference
// takes it from torchvision.models.resnet34 [third-party]).  NaN propagates, ties go to the first element in scan
// order (PyTorch).  Backward is a gather: an input pixel belongs to at most 4 windows; it receives a window's
// gradient when it is that window's (first) arg-max -- no atomics.
template <typename T>
__device__ __forceinline__ void pool3_window(const T* __restrict__ x, int n, int ho, int wo, int H, int W, int C, int c8,
                                             float* m, int* am) {
    const int h0 = max(2 * ho - 1, 0), w0 = max(2 * wo - 1, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; am[e] = h0 * W + w0; }
    for (int r = 0; r < 3; ++r) {
        const int h = 2 * ho - 1 + r;
        if ((unsigned)h >= (unsigned)H) continue;
        for (int q = 0; q < 3; ++q) {
            const int w = 2 * wo - 1 + q;
            if ((unsigned)w >= (unsigned)W) continue;
            float t[8];
            load8<T>(x, (((size_t)(n * H + h) * W + w) * C) + c8, t);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (t[e] > m[e] || t[e] != t[e]) { m[e] = t[e]; am[e] = h * W + w; }     // ATen's update rule
        }
    }
}

template <typename T>
__global__ void k_pool3s2_fwd(const void* __restrict__ xv, void* __restrict__ yv, int N, int H, int W, int C, int Ho, int Wo) {
    const int cg = C >> 3;
    const int64_t total = (int64_t)N * Ho * Wo * cg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg);
    const int wo = (int)((i / cg) % Wo);
    const int ho = (int)((i / ((int64_t)cg * Wo)) % Ho);
    const int n = (int)(i / ((int64_t)cg * Wo * Ho));
    float m[8];
    int am[8];
    pool3_window<T>((const T*)xv, n, ho, wo, H, W, C, c * 8, m, am);
    store8<T>((T*)yv, (size_t)i * 8, m);
}

template <typename T>
__global__ void k_pool3s2_bwd(const void* __restrict__ xv, const void* __restrict__ dyv, void* __restrict__ dxv, int N, int H,
                              int W, int C, int Ho, int Wo) {
    const int cg = C >> 3;
    const int64_t total = (int64_t)N * H * W * cg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg);
    const int w = (int)((i / cg) % W);
    const int h = (int)((i / ((int64_t)cg * W)) % H);
    const int n = (int)(i / ((int64_t)cg * W * H));
    float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int me = h * W + w;
    // windows (ho, wo) with 2*ho-1 <= h <= 2*ho+1
    for (int ho = (h >> 1); ho <= ((h + 1) >> 1); ++ho) {
        if (ho >= Ho) continue;
        for (int wo = (w >> 1); wo <= ((w + 1) >> 1); ++wo) {
            if (wo >= Wo) continue;
            float m[8], d[8];
            int am[8];
            pool3_window<T>((const T*)xv, n, ho, wo, H, W, C, c * 8, m, am);
            load8<T>((const T*)dyv, (((size_t)(n * Ho + ho) * Wo + wo) * C) + c * 8, d);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (am[e] == me) g[e] += d[e];
        }
    }
    store8<T>((T*)dxv, (size_t)i * 8, g);
}

extern "C" int cr_maxpool3x3s2_fwd(cr_ctx* ctx, const void* x, void* y, int N, int H, int W, int C, int act_f32) {
    CR_CHECK_ARG(ctx && x && y, "cr_maxpool3x3s2_fwd: NULL pointer");
    CR_CHECK_ARG(N > 0 && H > 0 && W > 0 && C % 8 == 0, "cr_maxpool3x3s2_fwd: bad dims");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
    CR_DISPATCH_T(act_f32, k_pool3s2_fwd, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, x, y, N, H, W, C, Ho, Wo);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_maxpool3x3s2_bwd(cr_ctx* ctx, const void* x, const void* dy, void* dx, int N, int H, int W, int C, int act_f32) {
    CR_CHECK_ARG(ctx && x && dy && dx, "cr_maxpool3x3s2_bwd: NULL pointer");
    CR_CHECK_ARG(N > 0 && H > 0 && W > 0 && C % 8 == 0, "cr_maxpool3x3s2_bwd: bad dims");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * H * W * (C / 8);
    CR_DISPATCH_T(act_f32, k_pool3s2_bwd, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, x, dy, dx, N, H, W, C,
                  Ho, Wo);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// FPN top-down: y[n,h,w,:] = lat[n,h,w,:] + top[n,h/2,w/2,:]   (nearest 2x upsample + sum)
template <typename T>
__global__ void k_upsample_add(const void* __restrict__ latv, const void* __restrict__ topv, void* __restrict__ yv, int N,
                               int H, int W, int C) {
    const int cg = C >> 3;
    const int64_t total = (int64_t)N * H * W * cg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg);
    const int w = (int)((i / cg) % W);
    const int h = (int)((i / ((int64_t)cg * W)) % H);
    const int n = (int)(i / ((int64_t)cg * W * H));
    float a[8], b[8];
    load8<T>((const T*)latv, (size_t)i * 8, a);
    load8<T>((const T*)topv, (((size_t)(n * (H / 2) + h / 2) * (W / 2) + w / 2) * C) + c * 8, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    store8<T>((T*)yv, (size_t)i * 8, a);
}

// its backward w.r.t. `top`: dtop[n,h2,w2,:] = sum of the 2x2 block of dy
template <typename T>
__global__ void k_sum2x2(const void* __restrict__ dyv, void* __restrict__ dtopv, int N, int H, int W, int C) {
    const T* __restrict__ dy = (const T*)dyv;
    const int Ho = H / 2, Wo = W / 2, cg = C >> 3;
    const int64_t total = (int64_t)N * Ho * Wo * cg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg);
    const int wo = (int)((i / cg) % Wo);
    const int ho = (int)((i / ((int64_t)cg * Wo)) % Ho);
    const int n = (int)(i / ((int64_t)cg * Wo * Ho));
    const size_t b = (((size_t)(n * H + 2 * ho) * W + 2 * wo) * C) + c * 8;
    float s[8], t[8];
    load8<T>(dy, b, s);
    const size_t offs[3] = {(size_t)C, (size_t)W * C, (size_t)W * C + C};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        load8<T>(dy, b + offs[q], t);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += t[e];
    }
    store8<T>((T*)dtopv, (size_t)i * 8, s);
}

extern "C" int cr_upsample2x_add(cr_ctx* ctx, const void* lat, const void* top, void* y, int N, int H, int W, int C,
                                 int act_f32) {
    CR_CHECK_ARG(ctx && lat && top && y, "cr_upsample2x_add: NULL pointer");
    CR_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "cr_upsample2x_add: bad dims");
    const int64_t total = (int64_t)N * H * W * (C / 8);
    if (total == 0) return CR_OK;
    CR_DISPATCH_T(act_f32, k_upsample_add, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, lat, top, y, N, H, W, C);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_sum2x2(cr_ctx* ctx, const void* dy, void* dtop, int N, int H, int W, int C, int act_f32) {
    CR_CHECK_ARG(ctx && dy && dtop, "cr_sum2x2: NULL pointer");
    CR_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "cr_sum2x2: bad dims");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    if (total == 0) return CR_OK;
    CR_DISPATCH_T(act_f32, k_sum2x2, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, dy, dtop, N, H, W, C);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// preprocess_image (detectron2 GeneralizedRCNN, Base.yaml:32-33): (x - mean)/std on a stacked
// uint8 (N,3,H,W) batch -> NHWC with channels padded with zeros to one 16-B chunk per pixel: 3 -> 8 (bf16) or 3 -> 4 (f32:
// the stem convolution then multiplies 7*7*4 instead of 7*7*8 of k, half the f32 MFMA work of the padded form)
template <typename T>
__global__ void k_preprocess(const unsigned char* __restrict__ img, void* __restrict__ yv, int N, int H, int W, float m0,
                             float m1, float m2, float s0, float s1, float s2) {
    const int64_t total = (int64_t)N * H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t hw = (int64_t)H * W;
    const int64_t n = i / hw, pix = i - n * hw;
    const unsigned char* b = img + n * 3 * hw + pix;
    const float f[8] = {((float)b[0] - m0) / s0, ((float)b[hw] - m1) / s1, ((float)b[2 * hw] - m2) / s2, 0, 0, 0, 0, 0};
    if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>((float*)yv + (size_t)i * 4) = make_float4(f[0], f[1], f[2], 0.f);
    else store8<T>((T*)yv, (size_t)i * 8, f);
}

extern "C" int cr_preprocess(cr_ctx* ctx, const unsigned char* img, void* y, int N, int H, int W, const float* mean3,
                             const float* std3, int act_f32) {
    CR_CHECK_ARG(ctx && img && y && mean3 && std3, "cr_preprocess: NULL pointer (mean3/std3 are HOST pointers)");
    const int64_t total = (int64_t)N * H * W;
    if (total == 0) return CR_OK;
    CR_DISPATCH_T(act_f32, k_preprocess, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, img, y, N, H, W,
                  mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// Linear layers of the RoI heads (FastRCNNConvFCHead 12544 -> 1024 -> 1024, the box predictors, CubeHead's shared FCs and
// its fused 13K predictor: cubercnn/modeling/roi_heads/cube_head.py:75,113-149,161-168; roi_heads.py:2160-2204) on the
// SAME hand-written implicit-GEMM kernels: a linear layer over R rows is a 1x1 convolution over a (1,1,R,K) map, so
// forward / backward-data run k_conv_igemm(_dma) and the weight gradient k_conv_wgrad(_f32) -- no library GEMM.
//   x (R,K), w (O,K), wt (K,O) in the activations' type (bf16 or f32 by act_f32); bias / dw / dbias f32.
// ---------------------------------------------------------------------------
extern "C" int cr_linear_fwd(cr_ctx* ctx, const void* x, const void* w, const float* bias, void* y, int R, int K, int O,
                             int relu, int out_f32, int act_f32, const void* w_split) {
    CR_CHECK_ARG(R >= 0, "cr_linear_fwd: bad row count");
    if (R == 0) return CR_OK;
    return cr_conv2d_fwd(ctx, x, w, y, 1, 1, R, K, O, 1, 1, 0, bias, nullptr, relu, nullptr, out_f32, act_f32, w_split);
}

extern "C" int cr_linear_bwd_data(cr_ctx* ctx, const void* dy, const void* wt, void* dx, int R, int K, int O, int act_f32,
                                  const void* wt_split) {
    CR_CHECK_ARG(R >= 0, "cr_linear_bwd_data: bad row count");
    if (R == 0) return CR_OK;
    return cr_conv2d_bwd_data(ctx, dy, wt, dx, 1, 1, R, K, O, 1, 1, 0, act_f32, wt_split, nullptr);
}

// dw (O,K) f32 (+)= dy^T x ; dbias (O) += column sums of dy when given (from the dy tiles the kernel stages anyway)
extern "C" int cr_linear_bwd_weight(cr_ctx* ctx, const void* dy, const void* x, float* dw, float* dbias, int R, int K, int O,
                                    int accumulate, int act_f32) {
    CR_CHECK_ARG(R > 0, "cr_linear_bwd_weight: bad row count");
    return conv2d_bwd_weight_impl(ctx, dy, x, dw, dbias, 1, 1, R, K, O, 1, 1, 0, accumulate, act_f32);
}

// dst (cols, rows) = src (rows, cols)^T, 2- or 4-byte elements: the (K,O) operand of cr_linear_bwd_data from a prepared
// (O,K) weight.  32x32 tiles through LDS, both sides coalesced.
template <typename T>
__global__ __launch_bounds__(256) void k_transpose2d(const T* __restrict__ src, T* __restrict__ dst, int rows, int cols) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * i][tx] = src[(size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][ty + 8 * i];
    }
}

extern "C" int cr_transpose2d(cr_ctx* ctx, const void* src, void* dst, int rows, int cols, int act_f32) {
    CR_CHECK_ARG(ctx && src && dst && rows > 0 && cols > 0, "cr_transpose2d: bad args");
    const dim3 grid((unsigned)cr_cdiv(cols, 32), (unsigned)cr_cdiv(rows, 32));
    if (act_f32) hipLaunchKernelGGL(k_transpose2d<float>, grid, dim3(256), 0, ctx->stream, (const float*)src, (float*)dst, rows, cols);
    else hipLaunchKernelGGL(k_transpose2d<u16>, grid, dim3(256), 0, ctx->stream, (const u16*)src, (u16*)dst, rows, cols);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
