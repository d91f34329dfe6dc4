// Detection-side kernels: ROIAlign (aligned, adaptive sampling) forward/backward over the FPN
// pyramid with the level assignment fused in, grouped NMS, and RPN anchor decoding.
//
// Reference call sites (paths into the reference tree); the arithmetic itself lives in
// third-party code that is absent from the reference and is restated from its published
// definition (parity unpinned, pinned here by a torch restatement in oracle/):
//   ROIPooler(ROIAlignV2 7x7, sampling_ratio 0)  cubercnn/modeling/roi_heads/roi_heads.py:2075-2080,2178,2273
//        = torchvision roi_align(aligned=True) + detectron2 assign_boxes_to_levels
//   nms / batched_nms                              cubercnn/modeling/roi_heads/fast_rcnn.py:105; detectron2 RPN
#include "cr_common.h"
#include <math.h>
#include <stdlib.h>

#include "cr_elem.h"

#define MAX_LEVELS 5
struct Pyramid {
    const void* feat[MAX_LEVELS];  // NHWC bf16 or f32 (template parameter of the kernels)
    float* grad[MAX_LEVELS];       // NHWC f32 (backward)
    int H[MAX_LEVELS], W[MAX_LEVELS];
    float scale[MAX_LEVELS];
    int nlev, C, min_level;        // min_level = log2(stride of level 0)
};

// detectron2 assign_boxes_to_levels: floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to the pyramid
__device__ __forceinline__ int roi_level(const float* b, const Pyramid& py) {
    const float area = (b[2] - b[0]) * (b[3] - b[1]);
    float lv = floorf(4.0f + log2f(sqrtf(area) / 224.0f + 1e-8f));
    const float lo = (float)py.min_level, hi = (float)(py.min_level + py.nlev - 1);
    lv = fminf(fmaxf(lv, lo), hi);
    return (int)lv - py.min_level;
}

struct Samp { int yl, yh, xl, xh; float w1, w2, w3, w4; bool ok; };
__device__ __forceinline__ Samp bilinear(float y, float x, int H, int W) {
    Samp s;
    s.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) && (y == y) && (x == x);   // NaN boxes sample nothing
    if (!s.ok) { s.yl = s.yh = s.xl = s.xh = 0; s.w1 = s.w2 = s.w3 = s.w4 = 0.f; return s; }
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    s.yl = (int)y; s.xl = (int)x;
    if (s.yl >= H - 1) { s.yh = s.yl = H - 1; y = (float)s.yl; } else s.yh = s.yl + 1;
    if (s.xl >= W - 1) { s.xh = s.xl = W - 1; x = (float)s.xl; } else s.xh = s.xl + 1;
    const float ly = y - s.yl, lx = x - s.xl, hy = 1.f - ly, hx = 1.f - lx;
    s.w1 = hy * hx; s.w2 = hy * lx; s.w3 = ly * hx; s.w4 = ly * lx;
    return s;
}

// one thread = (roi, ph, pw, 8 channels).  rois (R,5) = [batch, x1,y1,x2,y2].  out (R,PH,PW,C) in the storage type T
template <bool BWD, typename T>
__global__ __launch_bounds__(256) void k_roi_align(Pyramid py, const float* __restrict__ rois, int R, int PH, int PW,
                                                   T* __restrict__ out, const T* __restrict__ dout) {
    const int cg = py.C >> 3;
    const int64_t total = (int64_t)R * PH * PW * cg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg);
    const int pw = (int)((i / cg) % PW);
    const int ph = (int)((i / ((int64_t)cg * PW)) % PH);
    const int r = (int)(i / ((int64_t)cg * PW * PH));
    const float* rb = rois + (size_t)r * 5;
    const int n = (int)rb[0];
    const int lv = roi_level(rb + 1, py);
    const int H = py.H[lv], W = py.W[lv];
    const float sc = py.scale[lv];
    const float x1 = rb[1] * sc - 0.5f, y1 = rb[2] * sc - 0.5f;
    const float rw = (rb[3] - rb[1]) * sc, rh = (rb[4] - rb[2]) * sc;     // aligned: no 1-px floor
    const float bw = rw / (float)PW, bh = rh / (float)PH;
    const int gh = (int)ceilf(rh / (float)PH), gw = (int)ceilf(rw / (float)PW);
    const float cnt = fmaxf((float)(gh * gw), 1.f);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float g8[8];
    if (BWD) {
        load8<T>(dout, (size_t)i * 8, g8);
#pragma unroll
        for (int e = 0; e < 8; ++e) g8[e] /= cnt;
    }
    const size_t img = (size_t)n * H * W;
    for (int iy = 0; iy < gh; ++iy) {
        const float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
            const float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
            const Samp s = bilinear(y, x, H, W);
            if (!s.ok) continue;
            const size_t o1 = (img + (size_t)s.yl * W + s.xl) * py.C + c * 8, o2 = (img + (size_t)s.yl * W + s.xh) * py.C + c * 8;
            const size_t o3 = (img + (size_t)s.yh * W + s.xl) * py.C + c * 8, o4 = (img + (size_t)s.yh * W + s.xh) * py.C + c * 8;
            if (!BWD) {
                const T* f = (const T*)py.feat[lv];
                float a1[8], a2[8], a3[8], a4[8];
                load8<T>(f, o1, a1); load8<T>(f, o2, a2); load8<T>(f, o3, a3); load8<T>(f, o4, a4);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += s.w1 * a1[e] + s.w2 * a2[e] + s.w3 * a3[e] + s.w4 * a4[e];
            } else {
                float* gq = py.grad[lv];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    atomicAdd(gq + o1 + e, g8[e] * s.w1);
                    atomicAdd(gq + o2 + e, g8[e] * s.w2);
                    atomicAdd(gq + o3 + e, g8[e] * s.w3);
                    atomicAdd(gq + o4 + e, g8[e] * s.w4);
                }
            }
        }
    }
    if (!BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] /= cnt;
        store8<T>(out, (size_t)i * 8, acc);
    }
}

// backward: one thread = (roi, ph, pw, ONE channel), channel fastest, so every atomic wave-instruction adds
// 64 consecutive floats (256 contiguous bytes: the full-rate shape of MI355X_MICROARCH.md "Global float atomics").
template <typename T>
__global__ __launch_bounds__(256) void k_roi_align_bwd(Pyramid py, const float* __restrict__ rois, int R, int PH,
                                                       int PW, const T* __restrict__ dout) {
    const int C = py.C;
    const int64_t total = (int64_t)R * PH * PW * C;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int pw = (int)((i / C) % PW);
    const int ph = (int)((i / ((int64_t)C * PW)) % PH);
    const int r = (int)(i / ((int64_t)C * PW * PH));
    const float* rb = rois + (size_t)r * 5;
    const int n = (int)rb[0];
    const int lv = roi_level(rb + 1, py);
    const int H = py.H[lv], W = py.W[lv];
    const float sc = py.scale[lv];
    const float x1 = rb[1] * sc - 0.5f, y1 = rb[2] * sc - 0.5f;
    const float rw = (rb[3] - rb[1]) * sc, rh = (rb[4] - rb[2]) * sc;
    const float bw = rw / (float)PW, bh = rh / (float)PH;
    const int gh = (int)ceilf(rh / (float)PH), gw = (int)ceilf(rw / (float)PW);
    const float cnt = fmaxf((float)(gh * gw), 1.f);
    const float g = load1<T>(dout, (size_t)i) / cnt;
    if (g == 0.f) return;        // masked-out (padding) RoIs and dead channels add nothing: skip their 16 atomics
    float* gq = py.grad[lv] + (size_t)n * H * W * C + c;
    for (int iy = 0; iy < gh; ++iy) {
        const float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
            const float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
            const Samp s = bilinear(y, x, H, W);
            if (!s.ok) continue;
            atomicAdd(gq + ((size_t)s.yl * W + s.xl) * C, g * s.w1);
            atomicAdd(gq + ((size_t)s.yl * W + s.xh) * C, g * s.w2);
            atomicAdd(gq + ((size_t)s.yh * W + s.xl) * C, g * s.w3);
            atomicAdd(gq + ((size_t)s.yh * W + s.xh) * C, g * s.w4);
        }
    }
}

// Separable backward.  A sample's bilinear weight factors into a row part and a column part, so the gradient of one
// RoI w.r.t. its footprint of feature pixels is  Ay . G . Ax^T  per channel, with
//   Ay[y][ph] = sum over the gh samples of bin-row ph of their weight on feature row y, / gh      (Ax likewise),
//   G[ph][pw] = d(out)[r][ph][pw][c].
// One block per RoI builds Ay/Ax in LDS once; a thread owns one channel, keeps G in registers and issues ONE atomic per
// footprint pixel instead of 4 per sample (3-4x fewer atomics for the 2x2..4x4 sampling grids of FPN RoIs), still
// 64 consecutive floats per wave-instruction.
struct Lin1 { int lo, hi; float wlo, whi; bool ok; };
__device__ __forceinline__ Lin1 lin1(float y, int H) {
    Lin1 s;
    s.ok = (y >= -1.0f) && (y <= (float)H);             // false for NaN
    if (y <= 0.f) y = 0.f;
    s.lo = s.ok ? (int)y : 0;
    if (s.lo >= H - 1) { s.hi = s.lo = H - 1; y = (float)s.lo; } else s.hi = s.lo + 1;
    const float l = y - (float)s.lo;
    s.wlo = 1.f - l; s.whi = l;
    return s;
}

// XG = 8: the block index is (RoI, channel group blockIdx % 8) and a block covers C / 8 channels.  Workgroups are dealt
// round-robin to the 8 XCDs, so group x runs on XCD x and every 128-B line of the gradient maps (32 f32 channels of a pixel)
// is only ever added to from ONE XCD: the atomics stay in that XCD's L2 instead of the line migrating between the L2s of
// all the XCDs whose RoIs overlap there.  (Performance only: the L2s are coherent, any other dispatch order gives the same
// sums.)  The block's threads split the footprint rows: thread = (row phase t / CB, channel t % CB).
template <int P, typename T, int XG>
__global__ __launch_bounds__(256) void k_roi_align_bwd_sep(Pyramid py, const float* __restrict__ rois, int R,
                                                           const T* __restrict__ dout, int maxH) {
    extern __shared__ float sm[];                        // Ay [maxH][P] | Ax [maxW][P]
    __shared__ int s_lo[2], s_hi[2];
    float* Ay = sm;
    float* Ax = sm + (size_t)maxH * P;
    const int r = blockIdx.x / XG, t = threadIdx.x, C = py.C;
    const int CB = C / XG, cbase = (blockIdx.x % XG) * CB;   // this block's channels
    const int nph = XG == 1 ? 1 : (int)blockDim.x / CB;      // row phases (XG = 1: a thread walks every footprint row)
    const float* rb = rois + (size_t)r * 5;
    const int n = (int)rb[0];
    const int lv = roi_level(rb + 1, py);
    const int H = py.H[lv], W = py.W[lv];
    const float sc = py.scale[lv];
    const float x1 = rb[1] * sc - 0.5f, y1 = rb[2] * sc - 0.5f;
    const float rw = (rb[3] - rb[1]) * sc, rh = (rb[4] - rb[2]) * sc;
    const float bw = rw / (float)P, bh = rh / (float)P;
    const int gh = min((int)ceilf(rh / (float)P), 4096), gw = min((int)ceilf(rw / (float)P), 4096);
    if (t < 2) { s_lo[t] = 0x7fffffff; s_hi[t] = -1; }
    __syncthreads();
    // pass 1: footprint extent per axis (threads 0..P-1 rows, P..2P-1 columns)
    const bool builder = t < 2 * P;
    const int ax = t / P, p = t % P;                     // ax 0 = y, 1 = x
    const int gN = ax ? gw : gh, L = ax ? W : H;
    const float o1 = ax ? x1 : y1, bsz = ax ? bw : bh;
    if (builder) {
        int lo = 0x7fffffff, hi = -1;
        for (int i = 0; i < gN; ++i) {
            const Lin1 s = lin1(o1 + p * bsz + (i + 0.5f) * bsz / (float)gN, L);
            if (s.ok) { lo = min(lo, s.lo); hi = max(hi, s.hi); }
        }
        if (hi >= 0) { atomicMin(&s_lo[ax], lo); atomicMax(&s_hi[ax], hi); }
    }
    __syncthreads();
    const int y0 = s_lo[0], Py = s_hi[0] - y0 + 1, x0 = s_lo[1], Px = s_hi[1] - x0 + 1;
    if (s_hi[0] < 0 || s_hi[1] < 0 || n < 0) return;     // no valid sample at all (block-uniform)
    for (int i = t; i < Py * P; i += (int)blockDim.x) Ay[i] = 0.f;
    for (int i = t; i < Px * P; i += (int)blockDim.x) Ax[i] = 0.f;
    __syncthreads();
    if (builder) {                                       // pass 2: column p of Ay / Ax is owned by one thread
        float* A = ax ? Ax : Ay;
        const int base = ax ? x0 : y0;
        const float inv = 1.f / (float)gN;
        for (int i = 0; i < gN; ++i) {
            const Lin1 s = lin1(o1 + p * bsz + (i + 0.5f) * bsz / (float)gN, L);
            if (!s.ok) continue;
            A[(s.lo - base) * P + p] += s.wlo * inv;
            A[(s.hi - base) * P + p] += s.whi * inv;
        }
    }
    __syncthreads();
    for (int c = cbase + (XG == 1 ? t : t % CB); c < cbase + CB; c += (XG == 1 ? 256 : CB)) {
        const int ph0 = XG == 1 ? 0 : t / CB;
        float G[P][P];
        bool any = false;
#pragma unroll
        for (int ph = 0; ph < P; ++ph)
#pragma unroll
            for (int pw = 0; pw < P; ++pw) {
                G[ph][pw] = load1<T>(dout, (((size_t)r * P + ph) * P + pw) * C + c);
                any |= G[ph][pw] != 0.f;
            }
        if (!any) continue;                              // masked (padding) RoIs carry zero gradient
        float* gq = py.grad[lv] + (((size_t)n * H + y0) * W + x0) * C + c;
        for (int yy = ph0; yy < Py; yy += nph) {
            float tr[P];
#pragma unroll
            for (int pw = 0; pw < P; ++pw) tr[pw] = 0.f;
#pragma unroll
            for (int ph = 0; ph < P; ++ph) {
                const float a = Ay[yy * P + ph];         // LDS broadcast: uniform
                if (a != 0.f) {
#pragma unroll
                    for (int pw = 0; pw < P; ++pw) tr[pw] += a * G[ph][pw];
                }
            }
            float* grow = gq + (size_t)yy * W * C;
            for (int xx = 0; xx < Px; ++xx) {
                float v = 0.f;
#pragma unroll
                for (int pw = 0; pw < P; ++pw) v += Ax[xx * P + pw] * tr[pw];
                if (v != 0.f) atomicAdd(grow + (size_t)xx * C, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tile-owner backward (no global atomics, bit-reproducible).  The gradient maps are cut into 16 x 16 pixel tiles; a block
// owns (tile, 64 channels), walks the RoIs whose footprint touches its tile IN RoI ORDER, accumulates their  Ay . G . Ax^T
// contributions in registers (lane = channel, wave w = tile rows 4w .. 4w+3: 64 accumulators per lane) and stores every
// pixel of its tile once -- 256 contiguous bytes per wave store, zeros where no RoI reaches, so the maps need no zero
// fill either.  k_roi_align_bwd_sep adds 2048 RoIs x ~256 footprint pixels x 256 channels through f32 atomics, which run
// at ~1.3 TB/s of added bytes (MI355X_MICROARCH.md "Global float atomics"): 422 us per train step; here dY is read about
// once per tile it touches (~4x, from L2) and the maps are written once.
//   k_roi_bbox        one thread per RoI: level, image, footprint extent (same sample walk as the separable kernel)
//   k_roi_bwd_tiles   per block: ordered scan of the RoI boxes (chunks of 256, wave ballots -> LDS queue), then batches
//                     of 16 RoIs: their Ay / Ax columns restricted to the tile are built in LDS by 14 threads per RoI,
//                     every wave then runs the two small contractions per RoI with LDS-broadcast weights.
// ---------------------------------------------------------------------------------------------------------------
#define RT_TILE 16
#define RT_BATCH 16
struct RoiExt { int lv, n, y0, y1, x0, x1, pad0, pad1; };      // footprint rows y0..y1 / columns x0..x1 (y1 < y0: none)

template <int P>
__global__ __launch_bounds__(256) void k_roi_bbox(Pyramid py, const float* __restrict__ rois, int R, int N,
                                                  RoiExt* __restrict__ ext) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float* rb = rois + (size_t)r * 5;
    RoiExt e;
    e.n = (int)rb[0];
    e.lv = roi_level(rb + 1, py);
    e.pad0 = e.pad1 = 0;
    const int H = py.H[e.lv], W = py.W[e.lv];
    const float sc = py.scale[e.lv];
    const float x1 = rb[1] * sc - 0.5f, y1 = rb[2] * sc - 0.5f;
    const float rw = (rb[3] - rb[1]) * sc, rh = (rb[4] - rb[2]) * sc;
    const float bw = rw / (float)P, bh = rh / (float)P;
    const int gh = min((int)ceilf(rh / (float)P), 4096), gw = min((int)ceilf(rw / (float)P), 4096);
    int lo[2] = {0x7fffffff, 0x7fffffff}, hi[2] = {-1, -1};
#pragma unroll
    for (int ax = 0; ax < 2; ++ax) {
        const int gN = ax ? gw : gh, L = ax ? W : H;
        const float o1 = ax ? x1 : y1, bsz = ax ? bw : bh;
        // the sample positions grow with (p, i): the extent is [lo of the first valid sample, hi of the last valid one]
        const int tot = P * gN;
        for (int k = 0; k < tot; ++k) {
            const Lin1 s = lin1(o1 + (k / gN) * bsz + ((k % gN) + 0.5f) * bsz / (float)gN, L);
            if (s.ok) { lo[ax] = s.lo; hi[ax] = s.hi; break; }
        }
        for (int k = tot - 1; k >= 0 && hi[ax] >= 0; --k) {
            const Lin1 s = lin1(o1 + (k / gN) * bsz + ((k % gN) + 0.5f) * bsz / (float)gN, L);
            if (s.ok) { hi[ax] = max(hi[ax], s.hi); lo[ax] = min(lo[ax], s.lo); break; }
        }
    }
    const bool none = hi[0] < 0 || hi[1] < 0 || e.n < 0 || e.n >= N;
    // one pixel of slack per side: the weights themselves are rebuilt exactly per tile (zero where no sample reaches), the
    // extent only selects tiles, so a float rounding of a sample position across an integer cannot lose a contribution
    e.y0 = none ? 0 : max(lo[0] - 1, 0); e.y1 = none ? -1 : min(hi[0] + 1, H - 1);
    e.x0 = none ? 0 : max(lo[1] - 1, 0); e.x1 = none ? -1 : min(hi[1] + 1, W - 1);
    ext[r] = e;
}

struct TileMap { int base[MAX_LEVELS + 1], ty[MAX_LEVELS], tx[MAX_LEVELS]; int N; };

#define RT_QCAP 2048
#define RT_RING 4                      // G slots per block: the dY tiles of up to three RoIs are in flight under the current one
typedef __attribute__((address_space(3))) void* rt_lds_ptr_t;
// one LDS-DMA: 64 lanes x 16 B from the resource's base + voff[lane] (zeros past its extent) to dst + 16 lane.  A plain
// function (not inside the kernel template): the address-space cast and the target builtin are re-checked per instantiation
// there and fail silently in the host pass (see conv.hip).
__device__ __forceinline__ void rt_dma16(__amdgpu_buffer_rsrc_t r, void* dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (rt_lds_ptr_t)dst, 16, voff, 0, 0, 0);
}

// G comes out of the ring through inline-asm LDS reads: for a compiler-visible read of an array that an LDS-DMA writes,
// hipcc drains EVERY outstanding DMA (s_waitcnt vmcnt(0)) first, which would serialise the ring; completion of slot j is
// established by the counted vmcnt + barrier at the top of the RoI loop instead.
template <int OFF> __device__ __forceinline__ float rt_lds_f32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ unsigned rt_lds_u32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ unsigned rt_lds_addr(const void* p) { return (unsigned)(size_t)(rt_lds_ptr_t)p; }
template <typename T, int A, int P> __device__ __forceinline__ void rt_read_row(unsigned addr, float (&g)[P]) {
    // the P bins of bin-row A of one RoI's 64-channel slice: [A * P + b][64] elements of T, this lane's channel
    constexpr int ES = (int)sizeof(T);
    if (ES == 4) {
        g[0] = rt_lds_f32<(A * P + 0) * 64 * 4>(addr); g[1] = rt_lds_f32<(A * P + 1) * 64 * 4>(addr);
        g[2] = rt_lds_f32<(A * P + 2) * 64 * 4>(addr); g[3] = rt_lds_f32<(A * P + 3) * 64 * 4>(addr);
        g[4] = rt_lds_f32<(A * P + 4) * 64 * 4>(addr); g[5] = rt_lds_f32<(A * P + 5) * 64 * 4>(addr);
        g[6] = rt_lds_f32<(A * P + 6) * 64 * 4>(addr);
        // the loaded registers are in/out operands of the wait: no use of them may be scheduled above it (the compiler does
        // not know that an asm ds_read's destination is still in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6])
                     :: "memory");
    } else {
        // bf16: the 32-bit word holding this lane's channel (even lanes the low half, odd lanes the high half)
        const unsigned a4 = addr & ~3u;
        unsigned u[P];
        u[0] = rt_lds_u32<(A * P + 0) * 64 * 2>(a4); u[1] = rt_lds_u32<(A * P + 1) * 64 * 2>(a4);
        u[2] = rt_lds_u32<(A * P + 2) * 64 * 2>(a4); u[3] = rt_lds_u32<(A * P + 3) * 64 * 2>(a4);
        u[4] = rt_lds_u32<(A * P + 4) * 64 * 2>(a4); u[5] = rt_lds_u32<(A * P + 5) * 64 * 2>(a4);
        u[6] = rt_lds_u32<(A * P + 6) * 64 * 2>(a4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6])
                     :: "memory");
        const bool hi = (addr & 2u) != 0;
#pragma unroll
        for (int b = 0; b < P; ++b) g[b] = __uint_as_float(hi ? (u[b] & 0xffff0000u) : (u[b] << 16));
    }
}

template <int P, typename T>
__global__ __launch_bounds__(256, 2) void k_roi_bwd_tiles(Pyramid py, TileMap tm, const float* __restrict__ rois, int R,
                                                          const T* __restrict__ dout, const RoiExt* __restrict__ ext) {
    static_assert(P == 7, "built for 7 x 7 pooling");
    constexpr int RB = 64 * (int)sizeof(T);              // bytes of one bin's 64-channel row of dY
    constexpr int RPD = 1024 / RB;                        // rows per LDS-DMA instruction (1 KiB each)
    constexpr int NDMA = (P * P + RPD - 1) / RPD;         // DMA instructions per RoI: 13 (f32) / 7 (bf16)
    constexpr int SLOT = NDMA * 1024;                     // bytes of one ring slot
    __shared__ __attribute__((aligned(1024))) unsigned char s_G[RT_RING * SLOT];
    __shared__ int s_queue[RT_QCAP];
    __shared__ int s_wcnt[4];
    __shared__ __attribute__((aligned(16))) float s_A[RT_BATCH][2][P][RT_TILE];   // [RoI of the batch][y | x][bin][tile pixel]
    __shared__ int s_meta[RT_BATCH][4];                                            // tile-relative rows r0..r1, (bin, quarter) mask of x
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, C = py.C;
    const int cgs = C >> 6;
    const int tile = blockIdx.x / cgs, cg = blockIdx.x % cgs, c = cg * 64 + lane;
    int lv = 0;
    while (lv + 1 < py.nlev && tile >= tm.base[lv + 1]) ++lv;
    const int rel = tile - tm.base[lv];
    const int H = py.H[lv], W = py.W[lv];
    const int n = rel / (tm.ty[lv] * tm.tx[lv]);
    const int ty0 = ((rel / tm.tx[lv]) % tm.ty[lv]) * RT_TILE, tx0 = (rel % tm.tx[lv]) * RT_TILE;
    const float sc = py.scale[lv];

    float acc[4][RT_TILE];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RT_TILE; ++j) acc[i][j] = 0.f;

    const unsigned g_base = rt_lds_addr(s_G) + (unsigned)(lane * (int)sizeof(T));     // this lane's channel inside a ring slot
    // this wave's share of the DMA instructions of a RoI: m = w, w + 4, ... (4 for the first NDMA % 4 waves, else 3 / ...)
    const int my_dma = (NDMA - w + 3) / 4;
    // per-lane source offset inside a RoI's (P*P, C) block for DMA instruction m: row m * RPD + lane / (64 / RPD), 16 B per lane
    constexpr int LPR = 64 / RPD;                         // lanes per row
    const unsigned lane_row = (unsigned)(lane / LPR), lane_col = (unsigned)((lane % LPR) * 16);
    auto issue = [&](int rq, int slot) {
        // the 64-channel slice of RoI rq's dY -> ring slot (rows past P*P read zeros through the resource's range check)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(dout + (size_t)rq * (P * P) * (size_t)C), 0, (int)(P * P * C * sizeof(T)), 0x00020000);
        for (int m = w; m < NDMA; m += 4) {
            const unsigned row = (unsigned)(m * RPD) + lane_row;
            const unsigned voff = row < (unsigned)(P * P) ? row * (unsigned)(C * sizeof(T)) + (unsigned)(cg * RB) + lane_col
                                                          : 0x7fffffffu;
            rt_dma16(rs, s_G + slot * SLOT + m * 1024, voff);
        }
    };

    // ---- the batches of RT_BATCH queued RoIs s_queue[0 .. qn)
    auto run_queue = [&](int qn) {
        for (int q0 = 0; q0 < qn; q0 += RT_BATCH) {
            const int nb = min(RT_BATCH, qn - q0);
            // the first RT_RING - 1 dY tiles of the batch start moving before the weights are built
#pragma unroll
            for (int k = 0; k < RT_RING - 1; ++k)
                if (k < nb) issue(__builtin_amdgcn_readfirstlane(s_queue[q0 + k]), k);
            for (int u = t; u < nb * 2 * P; u += 256) {
                // thread (RoI j of the batch, axis, bin p): column p of Ay / Ax restricted to the tile's 16 rows / columns
                const int j = u / (2 * P), ax = (u / P) & 1, p = u % P;
                const float* rb = rois + (size_t)s_queue[q0 + j] * 5;
                const float o1 = (ax ? rb[1] : rb[2]) * sc - 0.5f;
                const float len = ((ax ? rb[3] : rb[4]) - (ax ? rb[1] : rb[2])) * sc;
                const float bsz = len / (float)P;
                const int gN = min((int)ceilf(len / (float)P), 4096), L = ax ? W : H, base = ax ? tx0 : ty0;
                float* A = s_A[j][ax][p];
#pragma unroll
                for (int i = 0; i < RT_TILE; i += 4) *reinterpret_cast<float4*>(A + i) = make_float4(0.f, 0.f, 0.f, 0.f);
                const float inv = 1.f / (float)gN;
                for (int i = 0; i < gN; ++i) {
                    const Lin1 s = lin1(o1 + p * bsz + (i + 0.5f) * bsz / (float)gN, L);
                    if (!s.ok) continue;
                    // (the two adds of a sample stay in the separable kernel's order: lo first, then hi)
                    const int a = s.lo - base, b = s.hi - base;
                    if (a >= 0 && a < RT_TILE) A[a] += s.wlo * inv;
                    if (b >= 0 && b < RT_TILE) A[b] += s.whi * inv;
                }
            }
            if (t < nb) {
                // rows of the tile RoI t of the batch reaches
                const RoiExt e = ext[s_queue[q0 + t]];
                s_meta[t][0] = max(e.y0 - ty0, 0);
                s_meta[t][1] = min(e.y1 - ty0, RT_TILE - 1);
                s_meta[t][2] = 0;
            }
            __syncthreads();
            if (t < nb * P) {
                // which x bins have any weight on which QUARTER (4 columns) of the tile: bit 4 p + q of s_meta[j][2].  A bin is
                // 2-4 feature pixels wide, so it reaches one or two of the four quarters: the column contraction below only
                // runs the (bin, quarter) pairs that are set
                const int j = t / P, p = t % P;
                int m = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(&s_A[j][1][p][4 * q]);
                    if (v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f) m |= 1 << (4 * p + q);
                }
                if (m) atomicOr(&s_meta[j][2], m);
            }
            // (the barrier of iteration 0 below publishes the masks)
            for (int j = 0; j < nb; ++j) {
                // ---- slot j % RT_RING is complete once every wave's share of RoI j has landed: a wave's DMAs complete in
                // issue order, so leaving the shares of the (up to two) younger RoIs outstanding is a counted vmcnt
                const int younger = min(nb - 1 - j, RT_RING - 2);
                if (younger >= 2) {
                    if (my_dma == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (my_dma == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else if (my_dma == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else if (younger == 1) {
                    if (my_dma == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (my_dma == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else if (my_dma == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();           // slot j ready for everyone; everyone is done reading slot (j - 1) % RT_RING
                if (j + RT_RING - 1 < nb)
                    issue(__builtin_amdgcn_readfirstlane(s_queue[q0 + j + RT_RING - 1]), (j + RT_RING - 1) % RT_RING);
                const int xmask = s_meta[j][2];
                if (s_meta[j][1] < 4 * w || s_meta[j][0] > 4 * w + 3 || xmask == 0) continue;   // wave-uniform: none of this wave's rows
                const unsigned gaddr = g_base + (unsigned)((j % RT_RING) * SLOT);
                float tr[4][P];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int b = 0; b < P; ++b) tr[rr][b] = 0.f;
#define RT_TROW(A)                                                                                                  \
                {                                                                                                   \
                    const float4 ay = *reinterpret_cast<const float4*>(&s_A[j][0][A][4 * w]);     /* LDS broadcast */ \
                    const float ayv[4] = {ay.x, ay.y, ay.z, ay.w};                                                  \
                    if (ay.x != 0.f || ay.y != 0.f || ay.z != 0.f || ay.w != 0.f) {                /* wave-uniform */ \
                        float g[P];                                                                                 \
                        rt_read_row<T, A, P>(gaddr, g);                                                             \
                        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr)                                            \
                            _Pragma("unroll") for (int b = 0; b < P; ++b) tr[rr][b] += ayv[rr] * g[b];              \
                    }                                                                                               \
                }
                RT_TROW(0) RT_TROW(1) RT_TROW(2) RT_TROW(3) RT_TROW(4) RT_TROW(5) RT_TROW(6)
#undef RT_TROW
#pragma unroll
                for (int b = 0; b < P; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (!((xmask >> (4 * b + q)) & 1)) continue;               // wave-uniform
                        const float4 v = *reinterpret_cast<const float4*>(&s_A[j][1][b][4 * q]);   // LDS broadcast
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const float tv = tr[rr][b];
                            acc[rr][4 * q + 0] += v.x * tv; acc[rr][4 * q + 1] += v.y * tv;
                            acc[rr][4 * q + 2] += v.z * tv; acc[rr][4 * q + 3] += v.w * tv;
                        }
                    }
            }
            __syncthreads();               // s_A, s_meta and the ring are free for the next batch
        }
    };

    // ---- ordered scan of the RoI extents: super-chunks of 8 x 256 RoIs with all eight loads of a thread in flight at once,
    // ordered compaction (wave ballots) into the LDS queue, drained after every super-chunk
    int qn = 0;
#pragma unroll 1
    for (int rbase = 0; rbase < R; rbase += 8 * 256) {
        RoiExt e8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = rbase + k * 256 + t;
            if (r < R) e8[k] = ext[r]; else { e8[k].lv = -1; e8[k].n = -1; e8[k].y0 = e8[k].x0 = 0; e8[k].y1 = e8[k].x1 = -1; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (rbase + k * 256 >= R) break;                                       // block-uniform
            const RoiExt& e = e8[k];
            const bool hit = e.lv == lv && e.n == n && e.y1 >= ty0 && e.y0 < ty0 + RT_TILE && e.x1 >= tx0 && e.x0 < tx0 + RT_TILE;
            const unsigned long long bal = __ballot(hit);
            __syncthreads();                                                       // s_wcnt free again (previous round's readers done)
            if (lane == 0) s_wcnt[w] = __popcll(bal);
            __syncthreads();
            int off = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { off += i < w ? s_wcnt[i] : 0; tot += s_wcnt[i]; }
            if (hit) s_queue[qn + off + __popcll(bal & ((1ull << lane) - 1ull))] = rbase + k * 256 + t;
            qn += tot;
        }
        __syncthreads();
        run_queue(qn);                     // (the queue holds at most the 2048 RoIs of one super-chunk)
        qn = 0;
    }
    // ---- every pixel of the tile is written once (64 consecutive channels per wave store)
    float* g = py.grad[lv];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int y = ty0 + 4 * w + rr;
        if (y >= H) continue;
#pragma unroll
        for (int i = 0; i < RT_TILE; ++i) {
            const int x = tx0 + i;
            if (x < W) g[(((size_t)n * H + y) * W + x) * C + c] = acc[rr][i];
        }
    }
}

static int fill_pyramid(Pyramid& py, const void* const* feats, float* const* grads, const int* Hs, const int* Ws,
                        const float* scales, int nlev, int C) {
    CR_CHECK_ARG(nlev >= 1 && nlev <= MAX_LEVELS, "roi_align: 1..%d levels", MAX_LEVELS);
    CR_CHECK_ARG(C % 8 == 0, "roi_align: C %% 8");
    py.nlev = nlev; py.C = C;
    for (int l = 0; l < nlev; ++l) {
        py.feat[l] = feats ? feats[l] : nullptr;
        py.grad[l] = grads ? grads[l] : nullptr;
        py.H[l] = Hs[l]; py.W[l] = Ws[l]; py.scale[l] = scales[l];
    }
    py.min_level = (int)lroundf(-log2f(scales[0]));
    return CR_OK;
}

// feats/Hs/Ws/scales are HOST arrays of length nlev (device pointers inside feats)
extern "C" int cr_roi_align_fwd(cr_ctx* ctx, const void* const* feats, const int* Hs, const int* Ws,
                                const float* scales, int nlev, int C, const float* rois, int64_t R, int PH, int PW,
                                void* out, int act_f32) {
    CR_CHECK_ARG(ctx && feats && Hs && Ws && scales, "cr_roi_align_fwd: NULL pointer");
    if (R == 0) return CR_OK;
    CR_CHECK_ARG(rois && out && PH > 0 && PW > 0, "cr_roi_align_fwd: bad args");
    Pyramid py;
    int rc = fill_pyramid(py, feats, nullptr, Hs, Ws, scales, nlev, C);
    if (rc) return rc;
    const int64_t total = R * PH * PW * (C / 8);
    if (act_f32)
        hipLaunchKernelGGL((k_roi_align<false, float>), dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, py,
                           rois, (int)R, PH, PW, (float*)out, (const float*)nullptr);
    else
        hipLaunchKernelGGL((k_roi_align<false, u16>), dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, py,
                           rois, (int)R, PH, PW, (u16*)out, (const u16*)nullptr);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// grads: HOST array of nlev device pointers to f32 NHWC maps (accumulated with atomics; zero them first)
extern "C" int cr_roi_align_bwd(cr_ctx* ctx, float* const* grads, const int* Hs, const int* Ws, const float* scales,
                                int nlev, int C, const float* rois, int64_t R, int PH, int PW, const void* dout,
                                int act_f32) {
    CR_CHECK_ARG(ctx && grads && Hs && Ws && scales, "cr_roi_align_bwd: NULL pointer");
    if (R == 0) return CR_OK;
    CR_CHECK_ARG(rois && dout && PH > 0 && PW > 0, "cr_roi_align_bwd: bad args");
    Pyramid py;
    int rc = fill_pyramid(py, nullptr, grads, Hs, Ws, scales, nlev, C);
    if (rc) return rc;
    if (PH == 7 && PW == 7 && !getenv("CR_ROI_BWD_PLAIN")) {
        int maxH = 0, maxW = 0;
        for (int l = 0; l < nlev; ++l) { maxH = Hs[l] > maxH ? Hs[l] : maxH; maxW = Ws[l] > maxW ? Ws[l] : maxW; }
        const size_t lds = (size_t)(maxH + maxW) * 7 * sizeof(float);
        if (lds <= 60 * 1024) {
            static const int xg_on = getenv("CR_ROI_BWD_XCD") ? atoi(getenv("CR_ROI_BWD_XCD")) : 1;
            static const int xg_threads = getenv("CR_ROI_BWD_T") ? atoi(getenv("CR_ROI_BWD_T")) : 128;   // 4 row phases x 32 channels
            if (xg_on && C % 256 == 0 && R * 8 < 0x7fffffff) {       // C / 8 channels per block, a multiple of a 128-B line
                const int nt = (C / 8) * (xg_threads / (C / 8) > 0 ? xg_threads / (C / 8) : 1);
                if (act_f32)
                    hipLaunchKernelGGL((k_roi_align_bwd_sep<7, float, 8>), dim3((unsigned)R * 8), dim3(nt), lds, ctx->stream, py,
                                       rois, (int)R, (const float*)dout, maxH);
                else
                    hipLaunchKernelGGL((k_roi_align_bwd_sep<7, u16, 8>), dim3((unsigned)R * 8), dim3(nt), lds, ctx->stream, py,
                                       rois, (int)R, (const u16*)dout, maxH);
            } else if (act_f32)
                hipLaunchKernelGGL((k_roi_align_bwd_sep<7, float, 1>), dim3((unsigned)R), dim3(256), lds, ctx->stream, py, rois,
                                   (int)R, (const float*)dout, maxH);
            else
                hipLaunchKernelGGL((k_roi_align_bwd_sep<7, u16, 1>), dim3((unsigned)R), dim3(256), lds, ctx->stream, py, rois,
                                   (int)R, (const u16*)dout, maxH);
            CR_LAUNCH_CHECK();
            return CR_OK;
        }
    }
    const int64_t total = R * PH * PW * (int64_t)C;
    CR_CHECK_ARG(cr_cdiv(total, 256) < 0x7fffffff, "cr_roi_align_bwd: too many RoIs");
    if (act_f32)
        hipLaunchKernelGGL(k_roi_align_bwd<float>, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, py, rois,
                           (int)R, PH, PW, (const float*)dout);
    else
        hipLaunchKernelGGL(k_roi_align_bwd<u16>, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream, py, rois,
                           (int)R, PH, PW, (const u16*)dout);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// Deterministic variant of cr_roi_align_bwd: the maps are OVERWRITTEN (every pixel of every level is stored once, no zero
// fill needed) and no global atomic is used -> bit-reproducible.  N = images in the batch.  7 x 7 pooling, C % 64 == 0.
extern "C" int cr_roi_align_bwd_set(cr_ctx* ctx, float* const* grads, const int* Hs, const int* Ws, const float* scales,
                                    int nlev, int C, int N, const float* rois, int64_t R, int PH, int PW, const void* dout,
                                    int act_f32) {
    CR_CHECK_ARG(ctx && grads && Hs && Ws && scales, "cr_roi_align_bwd_set: NULL pointer");
    CR_CHECK_ARG(PH == 7 && PW == 7 && C % 64 == 0 && N >= 1, "cr_roi_align_bwd_set: built for 7x7 pooling, C %% 64 == 0");
    CR_CHECK_ARG(R >= 0 && R <= (int64_t)(ctx->ws_bytes / sizeof(RoiExt)) && (R == 0 || (rois && dout)),
                 "cr_roi_align_bwd_set: bad RoI arguments");
    Pyramid py;
    int rc = fill_pyramid(py, nullptr, grads, Hs, Ws, scales, nlev, C);
    if (rc) return rc;
    TileMap tm;
    tm.N = N;
    int total = 0;
    for (int l = 0; l < nlev; ++l) {
        tm.base[l] = total;
        tm.ty[l] = (Hs[l] + RT_TILE - 1) / RT_TILE;
        tm.tx[l] = (Ws[l] + RT_TILE - 1) / RT_TILE;
        total += N * tm.ty[l] * tm.tx[l];
    }
    for (int l = nlev; l <= MAX_LEVELS; ++l) tm.base[l] = total;
    RoiExt* ext = (RoiExt*)ctx->ws;
    if (R > 0)
        hipLaunchKernelGGL(k_roi_bbox<7>, dim3((unsigned)cr_cdiv(R, 256)), dim3(256), 0, ctx->stream, py, rois, (int)R, N, ext);
    const unsigned nblk = (unsigned)total * (unsigned)(C / 64);
    if (act_f32)
        hipLaunchKernelGGL((k_roi_bwd_tiles<7, float>), dim3(nblk), dim3(256), 0, ctx->stream, py, tm, rois, (int)R,
                           (const float*)dout, ext);
    else
        hipLaunchKernelGGL((k_roi_bwd_tiles<7, u16>), dim3(nblk), dim3(256), 0, ctx->stream, py, tm, rois, (int)R,
                           (const u16*)dout, ext);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// grouped NMS.  boxes [G][maxn][4] sorted by descending score inside each group, counts [G].
// keep [G][maxn] uint8.  Standard bitmask formulation: pass 1 builds the suppression matrix in
// parallel, pass 2 is one wave per group walking the rows in order (rows prefetched).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_nms_mask(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                 int maxn, float thresh, unsigned long long* __restrict__ mask,
                                                 const int* __restrict__ cls = nullptr) {
    const int g = blockIdx.z, n = counts[g];
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    if (row0 >= n || col0 >= n) return;
    if (col0 + 63 < row0) return;                       // only j > i matters
    __shared__ float sb[64 * 4];
    __shared__ int sc[64];
    const float* gb = boxes + (size_t)g * maxn * 4;
    const int* gc = cls ? cls + (size_t)g * maxn : nullptr;     // class of every box: only same-class pairs suppress
    const int t = threadIdx.x;
    if (col0 + t < n) {
        sb[t * 4 + 0] = gb[(col0 + t) * 4 + 0]; sb[t * 4 + 1] = gb[(col0 + t) * 4 + 1];
        sb[t * 4 + 2] = gb[(col0 + t) * 4 + 2]; sb[t * 4 + 3] = gb[(col0 + t) * 4 + 3];
        sc[t] = gc ? gc[col0 + t] : 0;
    }
    __syncthreads();
    const int i = row0 + t;
    if (i >= n) return;
    const float ax1 = gb[i * 4], ay1 = gb[i * 4 + 1], ax2 = gb[i * 4 + 2], ay2 = gb[i * 4 + 3];
    const float aa = (ax2 - ax1) * (ay2 - ay1);
    const int ci = gc ? gc[i] : 0;
    unsigned long long bits = 0;
    const int jn = min(64, n - col0);
    for (int j = 0; j < jn; ++j) {
        if (col0 + j <= i || sc[j] != ci) continue;
        const float bx1 = sb[j * 4], by1 = sb[j * 4 + 1], bx2 = sb[j * 4 + 2], by2 = sb[j * 4 + 3];
        const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
        const float inter = w * h, ba = (bx2 - bx1) * (by2 - by1);
        if (inter / (aa + ba - inter) > thresh) bits |= 1ULL << j;
    }
    const int words = (maxn + 63) / 64;
    mask[((size_t)g * maxn + i) * words + blockIdx.x] = bits;
}

__global__ __launch_bounds__(64) void k_nms_scan(const int* __restrict__ counts, int maxn,
                                                 const unsigned long long* __restrict__ mask,
                                                 unsigned char* __restrict__ keep) {
    extern __shared__ unsigned long long removed[];     // words
    const int g = blockIdx.x, n = counts[g], t = threadIdx.x;
    const int words = (maxn + 63) / 64, nw = (n + 63) / 64;
    for (int w = t; w < words; w += 64) removed[w] = 0;
    __syncthreads();
    const unsigned long long* gm = mask + (size_t)g * maxn * words;
    unsigned char* gk = keep + (size_t)g * maxn;
    for (int i = 0; i < n; ++i) {
        const bool dead = (removed[i >> 6] >> (i & 63)) & 1ULL;     // uniform
        if (t == 0) gk[i] = dead ? 0 : 1;
        if (!dead) {
            // words left of the diagonal block were never written: start at i>>6
            for (int w = (i >> 6) + t; w < nw; w += 64) removed[w] |= gm[(size_t)i * words + w];
        }
        __syncthreads();
    }
    for (int i = n + t; i < maxn; i += 64) gk[i] = 0;
}

// Fast walk for maxn <= 2048 (words <= 32): 256 threads = 8 row-slots x 32 words.  Per 64-box chunk the rows are
// prefetched into registers while wave 0 resolves the diagonal block with register-only readlane steps, so no
// global-memory latency sits on the serial chain (the plain kernel above pays one load per box).
__global__ __launch_bounds__(256) void k_nms_scan32(const int* __restrict__ counts, int maxn,
                                                    const unsigned long long* __restrict__ mask,
                                                    unsigned char* __restrict__ keep) {
    __shared__ unsigned long long removed[32];
    __shared__ unsigned long long diag[64];
    __shared__ unsigned long long s_alive;
    const int g = blockIdx.x, n = counts[g], t = threadIdx.x;
    const int words = (maxn + 63) / 64, nch = (n + 63) / 64;
    const int w = t & 31, bs = t >> 5;
    if (t < 32) removed[t] = 0;
    const unsigned long long* gm = mask + (size_t)g * maxn * words;
    unsigned char* gk = keep + (size_t)g * maxn;
    unsigned long long cur[8], nxt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int row = bs + 8 * k;
        cur[k] = (w < words && row < n) ? gm[(size_t)row * words + w] : 0ULL;
    }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        if (w == c) {
#pragma unroll
            for (int k = 0; k < 8; ++k) diag[bs + 8 * k] = cur[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {                       // prefetch the next chunk's rows (words right of it only)
            const int row = (c + 1) * 64 + bs + 8 * k;
            nxt[k] = (c + 1 < nch && w > c && w < words && row < n) ? gm[(size_t)row * words + w] : 0ULL;
        }
        __syncthreads();
        if (t < 64) {
            const unsigned long long d = diag[t];
            const int dlo = (int)(unsigned)d, dhi = (int)(unsigned)(d >> 32);
            unsigned long long rem = removed[c];
#pragma unroll
            for (int b = 0; b < 64; ++b) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane(dlo, b), hi = (unsigned)__builtin_amdgcn_readlane(dhi, b);
                const unsigned long long row = ((unsigned long long)hi << 32) | lo;
                if (!((rem >> b) & 1ULL)) rem |= row;
            }
            const int left = n - c * 64;
            const unsigned long long valid = left >= 64 ? ~0ULL : ((1ULL << left) - 1ULL);
            const unsigned long long alive = ~rem & valid;
            if (c * 64 + t < maxn) gk[c * 64 + t] = (unsigned char)((alive >> t) & 1ULL);
            if (t == 0) s_alive = alive;
        }
        __syncthreads();
        const unsigned long long alive = s_alive;
        unsigned long long acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if ((alive >> (bs + 8 * k)) & 1ULL) acc |= cur[k];
        if (w > c && acc) atomicOr(&removed[w], acc);
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = nxt[k];
        __syncthreads();
    }
    for (int i = nch * 64 + t; i < maxn; i += 256) gk[i] = 0;
}

__global__ void k_fill_zero_u64(unsigned long long* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0ULL;
}

// mask_ws: workspace of G*maxn*ceil(maxn/64) u64.
static int nms_grouped_impl(cr_ctx* ctx, const float* boxes, const int* counts, int G, int maxn, float thresh, void* mask_ws,
                            unsigned char* keep, const int* cls);
extern "C" int cr_nms_grouped(cr_ctx* ctx, const float* boxes, const int* counts, int G, int maxn, float thresh,
                              void* mask_ws, unsigned char* keep) {
    return nms_grouped_impl(ctx, boxes, counts, G, maxn, thresh, mask_ws, keep, nullptr);
}
// the same with a class per box (cls (G,maxn) int32): a box is only suppressed by a kept box of its own class -- torchvision
// batched_nms over the detections of an image (fast_rcnn.py:105) without regrouping them by class
extern "C" int cr_nms_grouped_cls(cr_ctx* ctx, const float* boxes, const int32_t* cls, const int* counts, int G, int maxn,
                                  float thresh, void* mask_ws, unsigned char* keep) {
    CR_CHECK_ARG(cls != nullptr, "cr_nms_grouped_cls: cls is NULL");
    return nms_grouped_impl(ctx, boxes, counts, G, maxn, thresh, mask_ws, keep, cls);
}
static int nms_grouped_impl(cr_ctx* ctx, const float* boxes, const int* counts, int G, int maxn, float thresh, void* mask_ws,
                            unsigned char* keep, const int* cls) {
    CR_CHECK_ARG(ctx, "cr_nms_grouped: ctx is NULL");
    if (G == 0 || maxn == 0) return CR_OK;
    CR_CHECK_ARG(boxes && counts && mask_ws && keep && G > 0 && maxn > 0 && maxn <= 65536, "cr_nms_grouped: bad args");
    const int words = (maxn + 63) / 64;
    // rows below the diagonal block are skipped by the kernel -> clear the matrix first
    {   // a kernel, not hipMemsetAsync: memset nodes misbehave inside captured HIP graphs (ROCm 7.2)
        const int64_t nz = (int64_t)G * maxn * words;
        hipLaunchKernelGGL(k_fill_zero_u64, dim3((unsigned)cr_cdiv(nz, 256)), dim3(256), 0, ctx->stream,
                           (unsigned long long*)mask_ws, nz);
        CR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_nms_mask, dim3(words, words, G), dim3(64), 0, ctx->stream, boxes, counts, maxn, thresh,
                       (unsigned long long*)mask_ws, cls);
    CR_LAUNCH_CHECK();
    if (words <= 32)
        hipLaunchKernelGGL(k_nms_scan32, dim3(G), dim3(256), 0, ctx->stream, counts, maxn,
                           (const unsigned long long*)mask_ws, keep);
    else
        hipLaunchKernelGGL(k_nms_scan, dim3(G), dim3(64), words * 8, ctx->stream, counts, maxn,
                           (const unsigned long long*)mask_ws, keep);
    CR_LAUNCH_CHECK();
    return CR_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Test-time filter of the box head on padded proposals, without a host round trip (FastRCNNOutputs.inference ->
// fast_rcnn_inference_single_image, cubercnn/modeling/roi_heads/fast_rcnn.py:57-116 of the reference):
//   k_det_scores   softmax over the K+1 logits of every proposal; entry (row, class) of the masked score matrix S (B, P*K) is the
//                  class probability if it exceeds the threshold and the row is a real, finite prediction, else -inf; the
//                  number of candidates of every image is counted
//   (cr_topk)      the Kc best candidates of every image in descending order
//   k_det_gather   their boxes: Box2BoxTransform.apply_deltas of the (row, class) deltas, clipped to the image
//   (cr_nms_grouped_cls)  class-wise NMS over the sorted candidates
//   k_det_pick     the first `topk` survivors of every image, with the full score row of their proposal
// If an image has more candidates than Kc AND fewer than `topk` survivors among the first Kc, a box beyond the first Kc could
// still be a detection: the image is flagged (overflow) and the caller repeats it on the exact, unbounded path.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_det_scores(const float* __restrict__ logits, int ldl, const float* __restrict__ deltas,
                                                    int ldd, const float* __restrict__ prop, const float* __restrict__ objectness,
                                                    int B, int P, int K, int nreg, float thresh, float* __restrict__ S,
                                                    int* __restrict__ ncand) {
    // one wave per proposal row
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= B * P) return;
    const int b = r / P;
    const float* lg = logits + (size_t)r * ldl;
    float mx = -INFINITY;
    bool fin = true;
    for (int c = lane; c <= K; c += 64) { const float v = lg[c]; mx = fmaxf(mx, v); fin &= isfinite(v); }
    for (int c = lane; c < nreg * 4; c += 64) fin &= isfinite(deltas[(size_t)r * ldd + c]);
    if (lane < 4) fin &= isfinite(prop[(size_t)r * 4 + lane]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    fin = __all(fin) && (objectness == nullptr || isfinite(objectness[r]));
    float sum = 0.f;
    for (int c = lane; c <= K; c += 64) sum += expf(lg[c] - mx);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    int mine = 0;
    for (int c = lane; c < K; c += 64) {
        const float pr = expf(lg[c] - mx) / sum;
        const bool ok = fin && pr > thresh;
        S[(size_t)r * K + c] = ok ? pr : -INFINITY;
        mine += ok ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if (lane == 0 && mine) atomicAdd(&ncand[b], mine);
}

__global__ __launch_bounds__(256) void k_det_gather(const float* __restrict__ val, const int64_t* __restrict__ idx,
                                                    const float* __restrict__ deltas, int ldd, const float* __restrict__ prop,
                                                    const float* __restrict__ img_hw, int B, int P, int K, int nreg, int Kc, float wx,
                                                    float wy, float ww, float wh, float scale_clamp, float* __restrict__ boxes,
                                                    int* __restrict__ cls, int* __restrict__ row, int* __restrict__ counts) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * Kc) return;
    const int b = t / Kc;
    const float v = val[t];
    const bool ok = v > -INFINITY;
    const int64_t id = ok ? idx[t] : 0;
    const int rr = (int)(id / K), c = (int)(id - (int64_t)rr * K);
    const int r = b * P + rr;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        const float* d = deltas + (size_t)r * ldd + (nreg == 1 ? 0 : c * 4);
        const float* pb = prop + (size_t)r * 4;
        const float w = pb[2] - pb[0], h = pb[3] - pb[1], cx = pb[0] + 0.5f * w, cy = pb[1] + 0.5f * h;
        const float dx = d[0] / wx, dy = d[1] / wy, dw = fminf(d[2] / ww, scale_clamp), dh = fminf(d[3] / wh, scale_clamp);
        const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
        const float H = img_hw[b * 2], W = img_hw[b * 2 + 1];
        bx.x = fminf(fmaxf(pcx - 0.5f * pw, 0.f), W); bx.y = fminf(fmaxf(pcy - 0.5f * ph, 0.f), H);       // Boxes.clip
        bx.z = fminf(fmaxf(pcx + 0.5f * pw, 0.f), W); bx.w = fminf(fmaxf(pcy + 0.5f * ph, 0.f), H);
    }
    reinterpret_cast<float4*>(boxes)[t] = bx;
    cls[t] = ok ? c : -1;
    row[t] = ok ? rr : -1;
    // the candidates are sorted: the count of an image is the position of its first empty slot
    const float nxt = (t % Kc) + 1 < Kc ? val[t + 1] : -INFINITY;
    if (ok && !(nxt > -INFINITY)) counts[b] = (t % Kc) + 1;
    if (!ok && (t % Kc) == 0) counts[b] = 0;
}

#define DET_MAX_TOPK 512
__global__ __launch_bounds__(256) void k_det_pick(const unsigned char* __restrict__ keep, const int* __restrict__ counts,
                                                  const int* __restrict__ ncand, const float* __restrict__ val,
                                                  const float* __restrict__ boxes, const int* __restrict__ cls,
                                                  const int* __restrict__ row, const float* __restrict__ logits, int ldl, int P,
                                                  int K, int Kc, int topk, float* __restrict__ out_boxes,
                                                  float* __restrict__ out_scores, int64_t* __restrict__ out_cls,
                                                  int64_t* __restrict__ out_row, float* __restrict__ out_full,
                                                  int* __restrict__ out_count) {
    // one workgroup per image: ranks of the survivors by a block-wide prefix sum over chunks of 256 candidates
    __shared__ int s_wave[4];
    __shared__ int s_base;
    __shared__ int s_row[DET_MAX_TOPK];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = counts[b];
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + t;
        const bool k = i < n && keep[(size_t)b * Kc + i];
        const unsigned long long bal = __ballot(k);
        const int before = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        const int rank = off + before;
        if (k && rank < topk) {
            const size_t src = (size_t)b * Kc + i, dst = (size_t)b * topk + rank;
            reinterpret_cast<float4*>(out_boxes)[dst] = reinterpret_cast<const float4*>(boxes)[src];
            out_scores[dst] = val[src];
            out_cls[dst] = cls[src];
            out_row[dst] = row[src];
            s_row[rank] = row[src];
        }
        __syncthreads();
        if (t == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
        if (s_base >= topk) break;                            // block-uniform
    }
    const int kept = min(s_base, topk);
    if (t == 0) {
        out_count[b * 2] = kept;
        out_count[b * 2 + 1] = (ncand[b] > Kc && kept < topk) ? 1 : 0;       // overflow: repeat on the exact path
    }
    // scores_full of every detection: the softmax row of its proposal without the background column (fast_rcnn.py:96,110)
    for (int d = wave; d < topk; d += 4) {
        float* of = out_full + ((size_t)b * topk + d) * K;
        if (d >= kept) {
            for (int c = lane; c < K; c += 64) of[c] = 0.f;
            if (lane == 0) {
                const size_t dst = (size_t)b * topk + d;
                reinterpret_cast<float4*>(out_boxes)[dst] = make_float4(0.f, 0.f, 0.f, 0.f);
                out_scores[dst] = 0.f; out_cls[dst] = 0; out_row[dst] = 0;
            }
            continue;
        }
        const int rr = s_row[d];
        const float* lg = logits + ((size_t)b * P + rr) * ldl;
        float mx = -INFINITY;
        for (int c = lane; c <= K; c += 64) mx = fmaxf(mx, lg[c]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float sum = 0.f;
        for (int c = lane; c <= K; c += 64) sum += expf(lg[c] - mx);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        for (int c = lane; c < K; c += 64) of[c] = expf(lg[c] - mx) / sum;
    }
}

extern "C" int cr_det_scores(cr_ctx* ctx, const float* logits, int ldl, const float* deltas, int ldd, const float* prop_boxes,
                             const float* objectness, int B, int P, int K, int nreg, float thresh, float* S, int32_t* ncand) {
    CR_CHECK_ARG(ctx && logits && deltas && prop_boxes && S && ncand, "cr_det_scores: NULL pointer");
    CR_CHECK_ARG(B > 0 && P > 0 && K > 0 && ldl >= K + 1 && (nreg == 1 || nreg == K) && ldd >= nreg * 4, "cr_det_scores: bad sizes");
    hipLaunchKernelGGL(k_fill_zero_u64, dim3(1), dim3(256), 0, ctx->stream, (unsigned long long*)ncand, (int64_t)((B + 1) / 2));
    hipLaunchKernelGGL(k_det_scores, dim3((unsigned)cr_cdiv((int64_t)B * P, 4)), dim3(256), 0, ctx->stream, logits, ldl, deltas, ldd,
                       prop_boxes, objectness, B, P, K, nreg, thresh, S, ncand);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// val / idx (B,Kc): cr_topk of S viewed as (B, P*K), descending.  boxes (B,Kc,4), cls / row (B,Kc) int32 (-1 = empty slot),
// counts (B) int32 candidates per image among the Kc.
extern "C" int cr_det_gather(cr_ctx* ctx, const float* val, const int64_t* idx, const float* deltas, int ldd,
                             const float* prop_boxes, const float* img_hw, int B, int P, int K, int nreg, int Kc,
                             const float* weights4, float scale_clamp, float* boxes, int32_t* cls, int32_t* row, int32_t* counts) {
    CR_CHECK_ARG(ctx && val && idx && deltas && prop_boxes && img_hw && weights4 && boxes && cls && row && counts,
                 "cr_det_gather: NULL pointer");
    CR_CHECK_ARG(B > 0 && P > 0 && K > 0 && Kc > 0 && (nreg == 1 || nreg == K), "cr_det_gather: bad sizes");
    hipLaunchKernelGGL(k_det_gather, dim3((unsigned)cr_cdiv((int64_t)B * Kc, 256)), dim3(256), 0, ctx->stream, val, idx, deltas, ldd,
                       prop_boxes, img_hw, B, P, K, nreg, Kc, weights4[0], weights4[1], weights4[2], weights4[3], scale_clamp, boxes,
                       cls, row, counts);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// keep (B,Kc) from cr_nms_grouped_cls.  out_* (B,topk,...): boxes f32 x4, scores f32, classes / rows int64, scores_full (B,topk,K);
// out_count (B,2) int32 = [detections, overflow flag].  Empty slots are zero.
extern "C" int cr_det_pick(cr_ctx* ctx, const unsigned char* keep, const int32_t* counts, const int32_t* ncand, const float* val,
                           const float* boxes, const int32_t* cls, const int32_t* row, const float* logits, int ldl, int B, int P,
                           int K, int Kc, int topk, float* out_boxes, float* out_scores, int64_t* out_cls, int64_t* out_row,
                           float* out_full, int32_t* out_count) {
    CR_CHECK_ARG(ctx && keep && counts && ncand && val && boxes && cls && row && logits && out_boxes && out_scores && out_cls &&
                 out_row && out_full && out_count, "cr_det_pick: NULL pointer");
    CR_CHECK_ARG(B > 0 && P > 0 && K > 0 && Kc > 0 && topk > 0 && topk <= DET_MAX_TOPK && ldl >= K + 1, "cr_det_pick: bad sizes");
    hipLaunchKernelGGL(k_det_pick, dim3((unsigned)B), dim3(256), 0, ctx->stream, keep, counts, ncand, val, boxes, cls, row, logits, ldl,
                       P, K, Kc, topk, out_boxes, out_scores, out_cls, out_row, out_full, out_count);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
