// Row-wise top-k (values sorted descending + indices) for the four selections of the train step: the RPN's IoU-weighted
// anchor sampling and the RoI sampling (keys = (IoU + eps) / Exp(1): cubercnn/modeling/proposal_generator/rpn.py:275-328,
// roi_heads.py:2737-2771 of the reference), detectron2's per-level pre-NMS top-k and post-NMS top-k
// (find_top_rpn_proposals [third-party]).  Replaces torch.topk, whose multi-block path is five to eight launches plus
// memset nodes (which do not belong in a captured HIP graph on ROCm 7.2): here two launches, no memset, no atomics on
// global memory, deterministic (ties: lower index first).
//
//   k_topk_select   grid (rows, NB): block b of a row holds its chunk (<= 32768 values = 128 KB) in LDS as
//                   order-preserving 32-bit keys, finds the chunk's k-th largest key EXACTLY by a radix select (12 + 10 + 10
//                   bits) on LDS histograms, and writes its k candidates (key, index) -- all keys above the threshold in any order,
//                   then the ties in index order.
//   k_topk_merge    grid (rows): the NB x k candidates of a row as 64-bit words (key << 32 | ~index) in LDS, bitonic sort
//                   (descending: larger key first, then smaller index), the first k are the result.
// NaN follows torch.topk (a positive NaN is the largest value).  k <= 2048, NB * k <= 16384.
#include "cr_common.h"

#define TK_T 1024
#define TK_CH 32768

__device__ __forceinline__ unsigned tk_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // ascending uint order == ascending float order
}
__device__ __forceinline__ float tk_val(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(TK_T) void k_topk_select(const float* __restrict__ x, int64_t n, int k, int nb,
                                                      unsigned long long* __restrict__ cand) {
    extern __shared__ unsigned sk[];                             // chunk keys [m]
    __shared__ unsigned hist[4096];
    __shared__ unsigned s_prefix, s_need, s_cnt, s_scan[TK_T / 64], s_base;
    const int row = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int64_t chunk = (n + nb - 1) / nb;
    const int64_t c0 = (int64_t)b * chunk;
    const int m = (int)max((int64_t)0, min(chunk, n - c0));
    const float* xr = x + (size_t)row * n + c0;
    for (int i = t; i < m; i += TK_T) sk[i] = tk_key(xr[i]);
    unsigned long long* out = cand + ((size_t)row * nb + b) * k;
    const int kk = min(k, m);                                    // candidates this chunk can supply
    // slots past kk: smallest possible word (sorts last)
    for (int i = kk + t; i < k; i += TK_T) out[i] = 0ull;
    if (m == 0) return;                                          // block-uniform
    __syncthreads();
    // ---- exact kk-th largest key of the chunk: radix select, digits of 12 + 10 + 10 bits (most significant first; 12 bits
    // = sign, exponent and 3 mantissa bits spread typical data over enough bins that the LDS atomics rarely collide)
    unsigned prefix = 0, need = (unsigned)kk;                    // `need` of the elements matching `prefix` are still to be chosen
    const int lane0 = t & 63, wave0 = t >> 6;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const int bits = pass == 0 ? 12 : 10;
        const int shift = pass == 0 ? 20 : (pass == 1 ? 10 : 0);
        const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + bits));
        const unsigned dmask = (1u << bits) - 1u;
        const int nbins = 1 << bits, per = nbins / TK_T;          // bins per thread: 4 or 1
        for (int i = t; i < nbins; i += TK_T) hist[i] = 0;
        __syncthreads();
        for (int i = t; i < m; i += TK_T) {
            const unsigned key = sk[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & dmask], 1u);
        }
        __syncthreads();
        // thread t owns the bins [top - per*t - (per-1), top - per*t] (top = nbins - 1): counts from the largest digit down
        unsigned mine = 0;
        for (int q = 0; q < per; ++q) mine += hist[nbins - 1 - (per * t + q)];
        // exclusive prefix over the threads (wave scan + wave totals)
        unsigned inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(inc, off, 64);
            if (lane0 >= off) inc += o;
        }
        if (lane0 == 63) s_scan[wave0] = inc;
        __syncthreads();
        unsigned wbase = 0;
        for (int w = 0; w < wave0; ++w) wbase += s_scan[w];
        const unsigned before = wbase + inc - mine;               // elements in larger digits than this thread's bins
        if (before < need && need <= before + mine) {             // exactly one thread
            unsigned acc = before;
            for (int q = 0; q < per; ++q) {
                const int bin = nbins - 1 - (per * t + q);
                if (acc + hist[bin] >= need) { s_prefix = prefix | ((unsigned)bin << shift); s_need = need - acc; break; }
                acc += hist[bin];
            }
        }
        __syncthreads();
        prefix = s_prefix; need = s_need;
        __syncthreads();
    }
    const unsigned T = prefix;                                   // the kk-th largest key; `need` ties (== T) are taken, lowest index first
    if (t == 0) { s_cnt = 0; s_base = 0; }
    __syncthreads();
    // keys above T: any order (the merge sorts)
    for (int i = t; i < m; i += TK_T) {
        const unsigned key = sk[i];
        if (key > T) {
            const unsigned pos = atomicAdd(&s_cnt, 1u);
            out[pos] = ((unsigned long long)key << 32) | (unsigned long long)(~(unsigned)(c0 + i));
        }
    }
    __syncthreads();
    const unsigned n_gt = s_cnt;                                  // == kk - need
    // ties in index order: block-wide running rank
    const int lane = t & 63, wave = t >> 6;
    for (int i0 = 0; i0 < m; i0 += TK_T) {
        const int i = i0 + t;
        const bool tie = i < m && sk[i] == T;
        const unsigned long long bal = __ballot(tie);
        const unsigned before = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_scan[wave] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned wbase = 0, total = 0;
        for (int w = 0; w < TK_T / 64; ++w) { if (w < wave) wbase += s_scan[w]; total += s_scan[w]; }
        const unsigned rank = s_base + wbase + before;
        if (tie && rank < need)
            out[n_gt + rank] = ((unsigned long long)T << 32) | (unsigned long long)(~(unsigned)(c0 + i));
        __syncthreads();
        if (t == 0) s_base += total;
        __syncthreads();
        if (s_base >= need) break;                               // block-uniform
    }
}

__global__ __launch_bounds__(TK_T) void k_topk_merge(const unsigned long long* __restrict__ cand, int ncand, int k, int npad,
                                                     float* __restrict__ vals, int64_t* __restrict__ idx) {
    extern __shared__ unsigned long long sw[];                   // [npad], npad = power of two >= ncand
    const int row = blockIdx.x, t = threadIdx.x;
    const unsigned long long* c = cand + (size_t)row * ncand;
    for (int i = t; i < npad; i += TK_T) sw[i] = i < ncand ? c[i] : 0ull;
    __syncthreads();
    for (int len = 2; len <= npad; len <<= 1) {
        for (int j = len >> 1; j > 0; j >>= 1) {
            for (int i = t; i < npad; i += TK_T) {
                const int p = i ^ j;
                if (p > i) {
                    const bool desc = (i & len) == 0;            // descending overall
                    const unsigned long long a = sw[i], bq = sw[p];
                    if (desc ? a < bq : a > bq) { sw[i] = bq; sw[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = t; i < k; i += TK_T) {
        const unsigned long long w = sw[i];
        vals[(size_t)row * k + i] = tk_val((unsigned)(w >> 32));
        idx[(size_t)row * k + i] = (int64_t)(~(unsigned)(w & 0xffffffffull));
    }
}

// ws: rows * nb * k 64-bit words, nb = cr_topk_blocks(n, k)
extern "C" int cr_topk_blocks(int64_t n, int k) {
    int64_t nb = cr_cdiv(n, TK_CH);
    if (nb < 1) nb = 1;
    return (int)nb;
}

extern "C" int cr_topk(cr_ctx* ctx, const float* x, int rows, int64_t n, int k, void* ws, float* vals, int64_t* idx) {
    CR_CHECK_ARG(ctx && rows >= 0 && n >= 1 && k >= 1, "cr_topk: bad args");
    if (rows == 0) return CR_OK;
    CR_CHECK_ARG(x && ws && vals && idx, "cr_topk: NULL pointer");
    CR_CHECK_ARG(k <= 2048 && (int64_t)k <= n && n < (int64_t)0x7fffffff, "cr_topk: k=%d must be <= 2048 and <= n=%lld", k, (long long)n);
    const int nb = cr_topk_blocks(n, k);
    const int ncand = nb * k;
    CR_CHECK_ARG(ncand <= 16384, "cr_topk: %d chunks x k=%d candidates exceed the merge's 16384", nb, k);
    int npad = 1;
    while (npad < ncand) npad <<= 1;
    const int64_t chunk = cr_cdiv(n, nb);
    static bool attr_done = false;
    if (!attr_done) {
        CR_HIP(hipFuncSetAttribute((const void*)k_topk_select, hipFuncAttributeMaxDynamicSharedMemorySize, TK_CH * 4));
        CR_HIP(hipFuncSetAttribute((const void*)k_topk_merge, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
        attr_done = true;
    }
    hipLaunchKernelGGL(k_topk_select, dim3((unsigned)rows, (unsigned)nb), dim3(TK_T), (size_t)chunk * 4, ctx->stream, x, n, k, nb,
                       (unsigned long long*)ws);
    CR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_topk_merge, dim3((unsigned)rows), dim3(TK_T), (size_t)npad * 8, ctx->stream,
                       (const unsigned long long*)ws, ncand, k, npad, vals, idx);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
