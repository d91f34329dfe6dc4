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
    if (v != v) return 0xffffffffu;                              // every NaN (either sign) is the largest value, like torch.topk
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // ascending uint order == ascending float order
}
__device__ __forceinline__ float tk_val(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(TK_T) void k_topk_select(const float* __restrict__ x, int64_t n, int k, int nb,
                                                      unsigned long long* __restrict__ cand) {
    extern __shared__ unsigned sk[];                             // chunk keys [m]
    __shared__ unsigned hist[4096];
    __shared__ unsigned s_prefix, s_need, s_nties, s_cnt, s_scan[TK_T / 64], s_base;
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
        // the digit of the wave's first active lane is counted once for all lanes that share it (padding / suppressed
        // scores are thousands of equal keys: 64 lanes on one LDS address serialise), the others add one each
        for (int i0 = 0; i0 < m; i0 += TK_T) {
            const int i = i0 + t;
            const unsigned key = i < m ? sk[i] : 0u;
            const bool act = i < m && (key & mask) == prefix;
            const unsigned d = (key >> shift) & dmask;
            const unsigned long long am = __ballot(act);
            if (am == 0ull) continue;                                // wave-uniform
            const int first = __ffsll((long long)am) - 1;
            const unsigned d0 = (unsigned)__shfl((int)d, first, 64);
            const unsigned long long same = __ballot(act && d == d0);
            if (lane0 == first) atomicAdd(&hist[d0], (unsigned)__popcll(same));
            else if (act && d != d0) atomicAdd(&hist[d], 1u);
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
                if (acc + hist[bin] >= need) { s_prefix = prefix | ((unsigned)bin << shift); s_need = need - acc; s_nties = hist[bin]; break; }
                acc += hist[bin];
            }
        }
        __syncthreads();
        prefix = s_prefix; need = s_need;
        __syncthreads();
    }
    const unsigned T = prefix;                                   // the kk-th largest key; `need` ties (== T) are taken, lowest index first
    // after the last pass s_nties = number of keys equal to T: when all of them are needed (the usual case: distinct keys,
    // one tie) they go out with the keys above T and the index-ordered tie loop (three barriers per 1024 keys) is skipped
    const bool all_ties = s_nties == need;
    if (t == 0) { s_cnt = 0; s_base = 0; }
    __syncthreads();
    // keys above T: any order (the merge sorts)
    for (int i0 = 0; i0 < m; i0 += TK_T) {
        const int i = i0 + t;
        const unsigned key = i < m ? sk[i] : 0u;
        const bool gt = i < m && (key > T || (all_ties && key == T));
        const unsigned long long bal = __ballot(gt);
        if (bal == 0ull) continue;                                   // wave-uniform
        const int first = __ffsll((long long)bal) - 1;
        unsigned base = 0;
        if (lane0 == first) base = atomicAdd(&s_cnt, (unsigned)__popcll(bal));      // one LDS atomic per wave
        base = (unsigned)__shfl((int)base, first, 64);
        if (gt) {
            const unsigned pos = base + (unsigned)__popcll(bal & ((1ull << lane0) - 1ull));
            out[pos] = ((unsigned long long)key << 32) | (unsigned long long)(~(unsigned)(c0 + i));
        }
    }
    __syncthreads();
    if (all_ties) return;                                        // block-uniform
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

// E = npad / 1024 words per thread (element g = t * E + e) stay in registers: compare-exchange distances below E are
// register swaps, below 64 E lane shuffles inside the wave, and only the rest (10 of the 66 steps at npad = 2048) go through
// LDS with barriers -- the LDS-only version spent ~0.8 us per step on barriers and dependent LDS round trips.
template <int E>
__global__ __launch_bounds__(TK_T) void k_topk_merge(const unsigned long long* __restrict__ cand, int ncand, int k, int npad,
                                                     float* __restrict__ vals, int64_t* __restrict__ idx) {
    extern __shared__ unsigned long long sw[];                   // [npad] (cross-wave steps), word g at (g % E) * 1024 + g / E
    const int row = blockIdx.x, t = threadIdx.x;
    const unsigned long long* c = cand + (size_t)row * ncand;
    unsigned long long v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int g = t * E + e;
        v[e] = g < ncand ? c[g] : 0ull;
    }
    // g's partner is g ^ j; the pair keeps (larger, smaller) in index order where (g & len) == 0 (descending overall)
    for (int len = 2; len <= npad; len <<= 1) {
        for (int j = len >> 1; j >= E; j >>= 1) {
            if (j >= 64 * E) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) sw[e * TK_T + t] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int g = t * E + e, pg = g ^ j;
                    const unsigned long long pv = sw[(pg % E) * TK_T + pg / E];
                    const bool take_max = ((g & len) == 0) == ((g & j) == 0);
                    v[e] = take_max ? (v[e] > pv ? v[e] : pv) : (v[e] < pv ? v[e] : pv);
                }
            } else {
                const int lx = j / E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int g = t * E + e;
                    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(v[e] & 0xffffffffull), lx, 64);
                    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v[e] >> 32), lx, 64);
                    const unsigned long long pv = ((unsigned long long)hi << 32) | lo;
                    const bool take_max = ((g & len) == 0) == ((g & j) == 0);
                    v[e] = take_max ? (v[e] > pv ? v[e] : pv) : (v[e] < pv ? v[e] : pv);
                }
            }
        }
        // distances below E: inside the thread (compile-time register indices)
#pragma unroll
        for (int jj = E >> 1; jj > 0; jj >>= 1) {
            if (jj < len) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & jj) == 0) {
                        const int g = t * E + e;
                        const unsigned long long a = v[e], bq = v[e | jj];
                        const bool desc = (g & len) == 0;
                        if (desc ? a < bq : a > bq) { v[e] = bq; v[e | jj] = a; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int g = t * E + e;
        if (g < k) {
            vals[(size_t)row * k + g] = tk_val((unsigned)(v[e] >> 32));
            idx[(size_t)row * k + g] = (int64_t)(~(unsigned)(v[e] & 0xffffffffull));
        }
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
        CR_HIP(hipFuncSetAttribute((const void*)k_topk_merge<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
        CR_HIP(hipFuncSetAttribute((const void*)k_topk_merge<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
        attr_done = true;
    }
    hipLaunchKernelGGL(k_topk_select, dim3((unsigned)rows, (unsigned)nb), dim3(TK_T), (size_t)chunk * 4, ctx->stream, x, n, k, nb,
                       (unsigned long long*)ws);
    CR_LAUNCH_CHECK();
    const int E = npad <= TK_T ? 1 : npad / TK_T;
    const size_t lds = (size_t)E * TK_T * 8;                      // [E][1024] words
#define CR_TK_MERGE(E_) hipLaunchKernelGGL(k_topk_merge<E_>, dim3((unsigned)rows), dim3(TK_T), lds, ctx->stream, \
                                           (const unsigned long long*)ws, ncand, k, npad, vals, idx)
    if (E == 1) CR_TK_MERGE(1); else if (E == 2) CR_TK_MERGE(2); else if (E == 4) CR_TK_MERGE(4);
    else if (E == 8) CR_TK_MERGE(8); else CR_TK_MERGE(16);
#undef CR_TK_MERGE
    CR_LAUNCH_CHECK();
    return CR_OK;
}
