// Geometry kernels of the 1000-cube proposal-and-scoring method (HBM-bound, no
// MFMA): K17 project+score+argmax, K18 proposal sampler, K21 RANSAC plane, and
// the 8-corner helper.  gfx950 only; wave = 64.
//
// This translation unit is compiled with -ffp-contract=off: the float32
// operation order below is the contract shared with oracle/geometry.py, so the
// scores (and therefore the argmax) agree bit-for-bit with the oracle.
//
// Reference semantics restated here (paths into the reference tree):
//   corners      cubercnn/util/math_util.py:142-245
//   projection   ProposalNetwork/utils/spaces.py:224-245
//   2D boxes     ProposalNetwork/utils/conversions.py:25-48
//   scores       ProposalNetwork/scoring/scorefunction.py:47-85,144-160
//   argmax       cubercnn/modeling/roi_heads/roi_heads.py:492-505
//   propose      ProposalNetwork/proposals/proposals.py:338-424
//   RANSAC       ProposalNetwork/utils/plane.py:79-134
#include "cr_common.h"
#include <math.h>
#include <stdlib.h>

#define GEO_T 256          // threads per workgroup (4 waves)
#define GEO_W (GEO_T / 64)

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float nan_f() { return __builtin_nanf(""); }

// np.minimum / np.maximum: NaN-propagating
__device__ __forceinline__ float nmin(float a, float b) { return (a != a || b != b) ? nan_f() : fminf(a, b); }
__device__ __forceinline__ float nmax(float a, float b) { return (a != a || b != b) ? nan_f() : fmaxf(a, b); }

// torch.clamp / Tensor.clamp_: NaN stays NaN
__device__ __forceinline__ float clamp_keep_nan(float x, float lo, float hi) {
    return (x != x) ? x : fminf(fmaxf(x, lo), hi);
}

// deterministic float32 exp (Cephes scheme), bit-identical to oracle/geometry.py:exp_f32
__device__ __forceinline__ float cr_exp_f32(float x) {
    if (x != x) return x;
    if (x < -87.0f) return 0.0f;
    float xc = fminf(x, 88.0f);
    float k = rintf(xc * 1.4426950408889634f);
    float r = xc - k * 0.693359375f;
    r = r - k * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r; p = p + 1.3981999507e-3f;
    p = p * r; p = p + 8.3334519073e-3f;
    p = p * r; p = p + 4.1665795894e-2f;
    p = p * r; p = p + 1.6666665459e-1f;
    p = p * r; p = p + 5.0000001201e-1f;
    float r2 = r * r;
    p = p * r2;
    p = p + r;
    p = p + 1.0f;
    return ldexpf(p, (int)k);
}

// corner v of a cuboid in camera space, vertex order of math_util.py:198-207
// (x <- l, y <- h, z <- w).  c[0..14] = cx,cy,cz,w,h,l,R row-major.
__device__ __forceinline__ void cube_corner3d(const float* c, const int v, float& X, float& Y, float& Z) {
    const float hw = c[3] / 2.0f, hh = c[4] / 2.0f, hl = c[5] / 2.0f;
    const float vx = ((v & 3) == 1 || (v & 3) == 2) ? hl : -hl;   // +l/2 for {1,2,5,6}
    const float vy = (v & 2) ? hh : -hh;                           // +h/2 for {2,3,6,7}
    const float vz = (v & 4) ? hw : -hw;                           // +w/2 for {4,5,6,7}
    float a;
    a = c[6] * vx;  a = a + c[7] * vy;  a = a + c[8] * vz;  X = a + c[0];
    a = c[9] * vx;  a = a + c[10] * vy; a = a + c[11] * vz; Y = a + c[1];
    a = c[12] * vx; a = a + c[13] * vy; a = a + c[14] * vz; Z = a + c[2];
}

__device__ __forceinline__ void cube_corners3d(const float* c, float* X, float* Y, float* Z) {
#pragma unroll
    for (int v = 0; v < 8; ++v) cube_corner3d(c, v, X[v], Y[v], Z[v]);
}

struct Clamp { float lo0, hi0, lo1, hi1; };

// K @ X (homogeneous image point)
__device__ __forceinline__ void project_h(const float X, const float Y, const float Z, const float* K,
                                          float& p0, float& p1, float& p2) {
    p0 = (K[0] * X + K[1] * Y) + K[2] * Z;
    p1 = (K[3] * X + K[4] * Y) + K[5] * Z;
    p2 = (K[6] * X + K[7] * Y) + K[8] * Z;
}

// K @ X, perspective divide (no guard on p2 <= 0), clamp.  spaces.py:233-243
__device__ __forceinline__ void project1(const float X, const float Y, const float Z, const float* K, const Clamp cl,
                                         float& u, float& v) {
    float p0, p1, p2;
    project_h(X, Y, Z, K, p0, p1, p2);
    u = clamp_keep_nan(p0 / p2, cl.lo0, cl.hi0);
    v = clamp_keep_nan(p1 / p2, cl.lo1, cl.hi1);
}

__device__ __forceinline__ void project8(const float* X, const float* Y, const float* Z, const float* K,
                                         const Clamp cl, float* u, float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) project1(X[i], Y[i], Z[i], K, cl, u[i], v[i]);
}

__device__ __forceinline__ void minmax8(const float* a, float& mn, float& mx) {
    float lo = a[0], hi = a[0];
    bool nan = a[0] != a[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        nan |= a[i] != a[i];
        lo = fminf(lo, a[i]);
        hi = fmaxf(hi, a[i]);
    }
    mn = nan ? nan_f() : lo;      // torch.min/max over a dim propagate NaN
    mx = nan ? nan_f() : hi;
}

// ---- block reductions (256 threads = 4 waves) -----------------------------
// max with torch.max semantics: NaN if any NaN.  scratch: >= 2*GEO_W floats.
template <bool LDS_ONLY = false>
__device__ __forceinline__ float block_max_nanprop(float val, bool valid, float* scratch) {
    float m = valid && val == val ? val : -INFINITY;
    int anynan = valid && (val != val);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_down(m, off, 64));
        anynan |= __shfl_down(anynan, off, 64);
    }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
    if (l == 0) { scratch[w] = m; scratch[GEO_W + w] = anynan ? 1.0f : 0.0f; }
    if (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
    float r = scratch[0];
    float n = scratch[GEO_W];
#pragma unroll
    for (int i = 1; i < GEO_W; ++i) { r = fmaxf(r, scratch[i]); n += scratch[GEO_W + i]; }
    return n > 0.0f ? nan_f() : r;
}

__device__ __forceinline__ double block_sum_f64(double val, double* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = val;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int i = 1; i < GEO_W; ++i) r += scratch[i];
    return r;
}

// np.argmax ordering: NaN beats everything, then larger value, then smaller index
__device__ __forceinline__ bool arg_better(float va, int ia, float vb, int ib) {
    const bool na = va != va, nb = vb != vb;
    if (ia < 0) return false;
    if (ib < 0) return true;
    if (na != nb) return na;
    if (na) return ia < ib;
    if (va != vb) return va > vb;
    return ia < ib;
}

// ---------------------------------------------------------------------------
// K17: one workgroup per object; every cube is read once from HBM (staged
// through LDS with coalesced 16-B loads), all per-cube intermediates stay on
// chip across the two per-object reductions (max ratio difference, max
// chamfer), outputs are written once.
//
// Two kernels share the per-cube arithmetic below:
//   k_project_score       every score with the reference's float32 / float64 operation sequence (bit-exact planes)
//   k_project_score_fast  planes from reciprocal / native-exp / float32-chamfer arithmetic (1e-4 of the exact planes),
//                         argmax + best score re-evaluated with the exact sequence on the few cubes whose fast scores
//                         lie within an error interval of a per-object maximum -- the argmax stays bit-exact
// ---------------------------------------------------------------------------
struct ObjConst {
    float K[9];
    float r0, r1, r2, r3, a1, gt_ratio;       // box of the IoU term, its area, aspect ratio of the annotated box
    float mu0, mu1, mu2, sg0, sg1, sg2;
};

struct ScoreArgs {
    const float* cubes; int P; const float* Kmat; int k_per_object; Clamp cl;
    const float* ref_boxes; const float* prior_mu; const float* prior_sigma; const float* rect_pts;
    float* out_corners; float* out_boxes; float* out_iou; float* out_dim; float* out_corner; float* out_combined;
    int64_t* out_argmax; float* out_best; const float* iou_boxes;
    unsigned long long* stats;     // fast kernel only, may be NULL: [0] += objects sent through the exact sequence, [1] += candidates
};

__device__ __forceinline__ void load_obj(const ScoreArgs& a, int obj, ObjConst& o) {
    const float* kp = a.Kmat + (a.k_per_object ? (size_t)obj * 9 : 0);
#pragma unroll
    for (int i = 0; i < 9; ++i) o.K[i] = kp[i];
    // the box of the IoU term: the object's reference box, or a separate one (the MABO / pseudo-GT branches score IoU
    // against the PROJECTED ground-truth cube and the aspect ratio against the annotated box: roi_heads.py:459-460,530-537)
    const float* ib = a.iou_boxes ? a.iou_boxes : a.ref_boxes;
    o.r0 = ib[obj * 4 + 0]; o.r1 = ib[obj * 4 + 1]; o.r2 = ib[obj * 4 + 2]; o.r3 = ib[obj * 4 + 3];
    o.mu0 = a.prior_mu[obj * 3 + 0]; o.mu1 = a.prior_mu[obj * 3 + 1]; o.mu2 = a.prior_mu[obj * 3 + 2];
    o.sg0 = a.prior_sigma[obj * 3 + 0]; o.sg1 = a.prior_sigma[obj * 3 + 1]; o.sg2 = a.prior_sigma[obj * 3 + 2];
    o.a1 = (o.r2 - o.r0) * (o.r3 - o.r1);
    o.gt_ratio = (a.ref_boxes[obj * 4 + 2] - a.ref_boxes[obj * 4 + 0]) / (a.ref_boxes[obj * 4 + 3] - a.ref_boxes[obj * 4 + 1]);
}

// exact: 2D box, IoU (detectron2 pairwise_iou definition), size prior (scorefunction.py:151-152; dims are (w,h,l) =
// cu[3..5]), aspect-ratio difference
__device__ __forceinline__ void cube_scores_exact(const float* cu, const ObjConst& o,
                                                  const float b0, const float b1, const float b2, const float b3,
                                                  float& iou_out, float& gauss, float& diff) {
    const float a2 = (b2 - b0) * (b3 - b1);
    float w = nmin(o.r2, b2) - nmax(o.r0, b0);
    float h = nmin(o.r3, b3) - nmax(o.r1, b1);
    w = (w != w) ? w : fmaxf(w, 0.0f);
    h = (h != h) ? h : fmaxf(h, 0.0f);
    const float inter = w * h;
    const float iou = inter / ((o.a1 + a2) - inter);
    iou_out = inter > 0.0f ? iou : 0.0f;
    const float z0 = (cu[3] - o.mu0) / o.sg0, z1 = (cu[4] - o.mu1) / o.sg1, z2 = (cu[5] - o.mu2) / o.sg2;
    const float e0 = cr_exp_f32(-0.5f * (z0 * z0));
    const float e1 = cr_exp_f32(-0.5f * (z1 * z1));
    const float e2 = cr_exp_f32(-0.5f * (z2 * z2));
    gauss = ((e0 + e1) + e2) / 3.0f;
    const float pr = (b2 - b0) / (b3 - b1);
    diff = fabsf(o.gt_ratio - pr);
}

// exact: modified chamfer (scorefunction.py:51-56), float64 like scipy's cKDTree
__device__ __forceinline__ float cube_chamfer_exact(const float* u, const float* v, const float* s_rect) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double rx = (double)s_rect[q * 2], ry = (double)s_rect[q * 2 + 1];
        double best = INFINITY;
        bool nan = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double dx = rx - (double)u[i], dy = ry - (double)v[i];
            const double d2 = dx * dx + dy * dy;
            nan |= d2 != d2;
            best = fmin(best, d2);
        }
        const double d = nan ? (double)nan_f() : sqrt(best);
        acc = (q == 0) ? d : acc + d;
    }
    return (float)(acc / 4.0);
}

struct ScoreLds {
    float* s_cubes;      // GEO_T * 15
    double* s_red64;     // GEO_W * 4
    float* s_red;        // GEO_W * 2
    int* s_redi;         // GEO_W
    float* s_rect;       // 8
    float* s_v;          // >= 4 planes of CPT * GEO_T: per-cube intermediates that live across the two per-object
                         // reductions ([quantity][cube], conflict-free) -- not CPT-sized register arrays: with the
                         // cube loop rolled the kernel needs ~half the VGPRs and twice the waves are resident per SIMD
};

// coalesced staging of cnt cubes (15 floats each) into LDS
__device__ __forceinline__ void stage_cubes(const float* src, int cnt, float* s_cubes) {
    const int tid = threadIdx.x;
    const int nfl = cnt * 15;
    if ((((uintptr_t)src) & 15) == 0) {
        const int n4 = nfl >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(s_cubes);
        for (int i = tid; i < n4; i += GEO_T) d4[i] = s4[i];
        for (int i = (n4 << 2) + tid; i < nfl; i += GEO_T) s_cubes[i] = src[i];
    } else {
        for (int i = tid; i < nfl; i += GEO_T) s_cubes[i] = src[i];
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// The 64 cubes of a wave own 4 KB of contiguous corner output (64 B per lane).  Stored lane by lane that is four store
// instructions that each touch a quarter of 32 cache lines (39 -> 32.5 us for the fast kernel with whole lines instead).
// Transposed through LDS in four rounds of 16 cubes: the 16 lanes of round k park their four float4 in four 256-byte
// pieces `tw + piece * piece_stride` (the wave's still unused score slots of the chunk it is working on), then lane l
// picks float4 l of that 1 KB and the wave stores 1 KB contiguous.  Inline asm: the compiler would put
// `s_waitcnt vmcnt(0)` in front of LDS writes it can see while an LDS-DMA is in flight; one wave's LDS operations
// complete in order, so no wait between the writes and the read.  All 64 lanes of the wave must be active.
__device__ __forceinline__ void store_corners_transposed(const float* u, const float* v, float* wave_out, float* tw,
                                                         int piece_stride) {
    const int lane = threadIdx.x & 63;
    const unsigned wr_a = (unsigned)(size_t)(lds_ptr_t)(tw + ((lane & 15) >> 2) * piece_stride + (lane & 3) * 16);
    const unsigned rd_a = (unsigned)(size_t)(lds_ptr_t)(tw + (lane >> 4) * piece_stride + (lane & 15) * 4);
    f32x4_t* const og = reinterpret_cast<f32x4_t*>(wave_out) + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((lane >> 4) == k)
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\t"
                         "ds_write_b128 %0, %3 offset:32\n\tds_write_b128 %0, %4 offset:48"
                         :: "v"(wr_a), "v"((f32x4_t){u[0], v[0], u[1], v[1]}), "v"((f32x4_t){u[2], v[2], u[3], v[3]}),
                            "v"((f32x4_t){u[4], v[4], u[5], v[5]}), "v"((f32x4_t){u[6], v[6], u[7], v[7]}) : "memory");
        f32x4_t t;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(rd_a) : "memory");
        og[k * 64] = t;
    }
}

#define VAT(arr, c) arr[(c) * GEO_T]

// (defined with the fast kernel below)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cubes_rsrc(const float* cb, int P);
__device__ __forceinline__ void stage_cubes_dma(__amdgpu_buffer_rsrc_t r, int c, float* s_cubes);
__device__ __forceinline__ void lds_read_cube_asm(const float* s_cubes, int tid, float* cu);
__device__ __forceinline__ void wait_dma_under_stores(int stores);

// the whole object with the exact sequence (body of k_project_score; the fast kernel's fallback).  PIPE (k_project_score
// only; L.s_cubes is 16 KB then): objects with a rectangle stream their cubes like the fast kernel -- every wave copies its
// 64 cubes of the next chunk by LDS-DMA while it evaluates the current ones from registers, waits for its own copies only,
// and never for the acknowledgement of its plane stores; no workgroup barrier in the chunk loop.
template <int CPT, bool PIPE = false>
__device__ __forceinline__ void score_object_exact(const ScoreArgs& a, const ScoreLds& L, const int obj) {
    const int tid = threadIdx.x;
    const int P = a.P;
    const float* cb = a.cubes + (size_t)obj * P * 15;
    ObjConst o;
    load_obj(a, obj, o);
    float* const s_rect = L.s_rect;

    // no rectangle at all, or a NaN row for this object (empty mask, cr_mask_rects): the no-contour fallback below
    const bool have_rect = a.rect_pts != nullptr && a.rect_pts[obj * 8] == a.rect_pts[obj * 8];
    __syncthreads();
    if (have_rect && tid < 8) s_rect[tid] = a.rect_pts[obj * 8 + tid];

    float* const v_iou = L.s_v + 0 * CPT * GEO_T + tid;
    float* const v_gauss = L.s_v + 1 * CPT * GEO_T + tid;
    float* const v_diff = L.s_v + 2 * CPT * GEO_T + tid;
    float* const v_s = L.s_v + 3 * CPT * GEO_T + tid;
    double sum_mnx = 0, sum_mxx = 0, sum_mny = 0, sum_mxy = 0;

    const bool piped = PIPE && have_rect && (P & 3) == 0 && ((uintptr_t)a.cubes & 15) == 0;     // block-uniform
    if (piped) {
        const __amdgpu_buffer_rsrc_t rc = cubes_rsrc(cb, P);
        stage_cubes_dma(rc, 0, L.s_cubes);
        __syncthreads();                                    // s_rect
        float rect_r[8];                                    // (an LDS read inside the loop would wait for the copies in flight)
#pragma unroll
        for (int i = 0; i < 8; ++i) rect_r[i] = s_rect[i];
        const int nstores = (a.out_corners ? 4 : 0) + (a.out_boxes ? 1 : 0);
#pragma unroll 1
        for (int c = 0; c < CPT; ++c) {
            const int base = c * GEO_T;
            if (base >= P) break;
            const int cnt = min(GEO_T, P - base);
            float cu[15];
            wait_dma_under_stores(c == 0 ? 0 : nstores);
            lds_read_cube_asm(L.s_cubes + (tid >> 6) * 1024, tid & 63, cu);
            if (base + GEO_T < P) stage_cubes_dma(rc, c + 1, L.s_cubes);
            asm volatile("" ::: "memory");
            if (tid < cnt) {
                float X[8], Y[8], Z[8], u[8], v[8];
                cube_corners3d(cu, X, Y, Z);
                project8(X, Y, Z, o.K, a.cl, u, v);
                float b0, b1, b2, b3, iou, gauss, diff;
                minmax8(u, b0, b2);
                minmax8(v, b1, b3);
                cube_scores_exact(cu, o, b0, b1, b2, b3, iou, gauss, diff);
                const size_t gi = (size_t)obj * P + base + tid;
                if (a.out_corners) {
                    float4* oc = reinterpret_cast<float4*>(a.out_corners + gi * 16);
                    oc[0] = make_float4(u[0], v[0], u[1], v[1]);
                    oc[1] = make_float4(u[2], v[2], u[3], v[3]);
                    oc[2] = make_float4(u[4], v[4], u[5], v[5]);
                    oc[3] = make_float4(u[6], v[6], u[7], v[7]);
                }
                if (a.out_boxes) reinterpret_cast<float4*>(a.out_boxes)[gi] = make_float4(b0, b1, b2, b3);
                VAT(v_iou, c) = iou;
                VAT(v_gauss, c) = gauss;
                VAT(v_diff, c) = diff;
                VAT(v_s, c) = cube_chamfer_exact(u, v, rect_r);
            }
        }
        __syncthreads();
    } else
    // ---------------- pass A: corners, boxes, iou, gauss, ratio diff (+ chamfer if rect given)
    for (int pass = 0; pass < (have_rect ? 1 : 2); ++pass) {
        if (pass == 1) {
            // no-contour fallback rect (scorefunction.py:69-75): mean over proposals of min/max u,v
            const double t0 = block_sum_f64(sum_mnx, L.s_red64);
            const double t1 = block_sum_f64(sum_mxx, L.s_red64);
            const double t2 = block_sum_f64(sum_mny, L.s_red64);
            const double t3 = block_sum_f64(sum_mxy, L.s_red64);
            if (tid == 0) {
                const float mnx = (float)(t0 / P), mxx = (float)(t1 / P);
                const float mny = (float)(t2 / P), mxy = (float)(t3 / P);
                s_rect[0] = mnx; s_rect[1] = mny; s_rect[2] = mxx; s_rect[3] = mny;
                s_rect[4] = mxx; s_rect[5] = mxy; s_rect[6] = mnx; s_rect[7] = mxy;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < CPT; ++c) {
            const int base = c * GEO_T;
            if (base < P) {                          // block-uniform
            const int cnt = min(GEO_T, P - base);
            stage_cubes(cb + (size_t)base * 15, cnt, L.s_cubes);
            __syncthreads();
            const int p = base + tid;
            const bool valid = tid < cnt;
            if (valid) {
                float cu[15];
#pragma unroll
                for (int j = 0; j < 15; ++j) cu[j] = L.s_cubes[tid * 15 + j];
                float X[8], Y[8], Z[8], u[8], v[8];
                cube_corners3d(cu, X, Y, Z);
                project8(X, Y, Z, o.K, a.cl, u, v);
                if (pass == 0) {
                    float b0, b1, b2, b3, iou, gauss, diff;
                    minmax8(u, b0, b2);
                    minmax8(v, b1, b3);
                    cube_scores_exact(cu, o, b0, b1, b2, b3, iou, gauss, diff);
                    const size_t gi = (size_t)obj * P + p;
                    if (a.out_corners) {
                        float4* oc = reinterpret_cast<float4*>(a.out_corners + gi * 16);
                        oc[0] = make_float4(u[0], v[0], u[1], v[1]);
                        oc[1] = make_float4(u[2], v[2], u[3], v[3]);
                        oc[2] = make_float4(u[4], v[4], u[5], v[5]);
                        oc[3] = make_float4(u[6], v[6], u[7], v[7]);
                    }
                    if (a.out_boxes) reinterpret_cast<float4*>(a.out_boxes)[gi] = make_float4(b0, b1, b2, b3);
                    VAT(v_iou, c) = iou;
                    VAT(v_gauss, c) = gauss;
                    VAT(v_diff, c) = diff;
                    sum_mnx += b0; sum_mxx += b2; sum_mny += b1; sum_mxy += b3;
                }
                if (have_rect || pass == 1) VAT(v_s, c) = cube_chamfer_exact(u, v, s_rect);
            }
            __syncthreads();
            }
        }
    }

    // ---------------- per-object normalisers
    float lmaxd = -INFINITY, lmaxs = -INFINITY;
    bool nand = false, nans = false;
#pragma unroll 1
    for (int c = 0; c < CPT; ++c) {
        if (c * GEO_T + tid < P) {
            nand |= VAT(v_diff, c) != VAT(v_diff, c);
            nans |= VAT(v_s, c) != VAT(v_s, c);
            lmaxd = fmaxf(lmaxd, VAT(v_diff, c));
            lmaxs = fmaxf(lmaxs, VAT(v_s, c));
        }
    }
    const bool anyv = tid < P;
    const float maxdiff = block_max_nanprop(nand ? nan_f() : lmaxd, anyv, L.s_red);
    const float maxs = block_max_nanprop(nans ? nan_f() : lmaxs, anyv, L.s_red);

    // ---------------- pass B: final scores + argmax
    float bestv = 0.0f;
    int besti = -1;
#pragma unroll 1
    for (int c = 0; c < CPT; ++c) {
        const int p = c * GEO_T + tid;
        if (p < P) {
            const float dim = (1.0f - VAT(v_diff, c) / maxdiff) * VAT(v_gauss, c);
            const float cor = 1.0f - VAT(v_s, c) / maxs;
            const float comb = (VAT(v_iou, c) * dim) * cor;
            const size_t gi = (size_t)obj * P + p;
            if (a.out_iou) a.out_iou[gi] = VAT(v_iou, c);
            if (a.out_dim) a.out_dim[gi] = dim;
            if (a.out_corner) a.out_corner[gi] = cor;
            if (a.out_combined) a.out_combined[gi] = comb;
            if (arg_better(comb, p, bestv, besti)) { bestv = comb; besti = p; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(bestv, off, 64);
        const int oi = __shfl_down(besti, off, 64);
        if (arg_better(ov, oi, bestv, besti)) { bestv = ov; besti = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { L.s_red[tid >> 6] = bestv; L.s_redi[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        float bv = L.s_red[0];
        int bi = L.s_redi[0];
        for (int i = 1; i < GEO_W; ++i)
            if (arg_better(L.s_red[i], L.s_redi[i], bv, bi)) { bv = L.s_red[i]; bi = L.s_redi[i]; }
        a.out_argmax[obj] = bi < 0 ? 0 : bi;
        if (a.out_best) a.out_best[obj] = bi < 0 ? 0.0f : bv;
    }
}

template <int CPT>
__global__ __launch_bounds__(GEO_T) void k_project_score(const ScoreArgs a) {
    __shared__ __attribute__((aligned(16))) float s_cubes[GEO_T * 16];
    __shared__ double s_red64[GEO_W * 4];
    __shared__ float s_red[GEO_W * 2];
    __shared__ int s_redi[GEO_W];
    __shared__ float s_rect[8];
    __shared__ float s_v[4 * CPT * GEO_T];
    const ScoreLds L = {s_cubes, s_red64, s_red, s_redi, s_rect, s_v};
    score_object_exact<CPT, true>(a, L, blockIdx.x);
}

// ---- fast variant ------------------------------------------------------------
// Error model of the fast planes.  Corners X,Y,Z and the homogeneous p0,p1,p2 use the exact kernel's operation
// sequence (bit-identical); u = p0 * rcp(p2) instead of the IEEE quotient is off by <= 2 ulp of |u| <= `delta` =
// 2^-22 * max |clamp bound| pixels.  Everything downstream is bounded from that: a box edge by delta, a chamfer
// distance by 2 delta (+ float32 rounding of the distance itself), the aspect ratio by 2 delta (1 + ratio) / height.
// The candidate intervals below use FOUR times those bounds; tests/test_gpu_geometry.py measures the actual
// differences.  Anything non-finite anywhere (a cube behind the camera plane hitting p2 = 0, a degenerate box, a zero
// sigma ...) sends the whole object through the exact sequence: NaN / inf ordering is the reference's business.
#define GEO_CAND 256
#define GEO_STAGGER 8            // x 64 cycles between the first fetches of a workgroup's waves (4 ... 12 measure the same)
typedef float v2f __attribute__((ext_vector_type(2)));

// one LDS-DMA: 64 lanes x 16 B from base + voff[lane] (0 past the descriptor's extent) to dst + 16 lane.  A plain function:
// the target builtin inside an instantiation-dependent call is re-checked per template instantiation.
__device__ __forceinline__ void dma16_f(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cubes_rsrc(const float* cb, int P) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)cb, 0, P * 60, 0x00020000);
}
// asynchronous copy of this WAVE's 64 cubes of chunk c (3 840 B = 240 16-byte items; P % 4 == 0) into the wave's own
// 4 KB of s_cubes: no workgroup barrier in the staging loop, a wave waits for its own copies only (vmcnt).  Unconditional
// (branches would let the scheduler sink the copies below the arithmetic they are meant to hide under): the 16 items past
// the wave's cubes are the next wave's first bytes or, past the object's last cube, the zeros the buffer descriptor
// returns out of range.
__device__ __forceinline__ void stage_cubes_dma(__amdgpu_buffer_rsrc_t r, int c, float* s_cubes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned src = (unsigned)(c * GEO_T + wave * 64) * 60u;
#pragma unroll
    for (int k = 0; k < 4; ++k) dma16_f(r, s_cubes + wave * 1024 + k * 256, src + (unsigned)(k * 64 + lane) * 16u);
}

// one candidate cube evaluated by 8 consecutive lanes (lane i = corner i) with the exact operation sequence of
// score_object_exact; only minima / maxima / NaN flags cross lanes (exact in any order), so every lane of the group ends
// with the same bits the one-thread-per-cube sequence produces -- at an eighth of its dependent instruction chain
__device__ __forceinline__ void cand_exact8(const float* cu, const int i, const ObjConst& o, const Clamp cl,
                                            const float* s_rect, float& iou, float& gauss, float& diff, float& sc) {
    float X, Y, Z, u, v;
    cube_corner3d(cu, i, X, Y, Z);
    project1(X, Y, Z, o.K, cl, u, v);
    float ulo = u, uhi = u, vlo = v, vhi = v;
    int nu = u != u, nv = v != v;
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        ulo = fminf(ulo, __shfl_xor(ulo, m, 64)); uhi = fmaxf(uhi, __shfl_xor(uhi, m, 64));
        vlo = fminf(vlo, __shfl_xor(vlo, m, 64)); vhi = fmaxf(vhi, __shfl_xor(vhi, m, 64));
        nu |= __shfl_xor(nu, m, 64); nv |= __shfl_xor(nv, m, 64);
    }
    const float b0 = nu ? nan_f() : ulo, b2 = nu ? nan_f() : uhi;      // minmax8
    const float b1 = nv ? nan_f() : vlo, b3 = nv ? nan_f() : vhi;
    cube_scores_exact(cu, o, b0, b1, b2, b3, iou, gauss, diff);
    double acc = 0.0;                                                   // cube_chamfer_exact
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double rx = (double)s_rect[q * 2], ry = (double)s_rect[q * 2 + 1];
        const double dx = rx - (double)u, dy = ry - (double)v;
        const double d2 = dx * dx + dy * dy;
        int nan = d2 != d2;
        double best = fmin((double)INFINITY, d2);
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            best = fmin(best, __shfl_xor(best, m, 64));
            nan |= __shfl_xor(nan, m, 64);
        }
        const double d = nan ? (double)nan_f() : sqrt(best);
        acc = (q == 0) ? d : acc + d;
    }
    sc = (float)(acc / 4.0);
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global store
__device__ __forceinline__ void barrier_lds_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// `scratch` must not be in use by an earlier reduction that slower waves may still be reading (no leading barrier: a
// wave publishes its partial as soon as it is done with its own cubes; the one barrier waits for the slowest wave)
__device__ __forceinline__ float block_max_plain(float v, float* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    barrier_lds_only();
    float r = scratch[0];
#pragma unroll
    for (int i = 1; i < GEO_W; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// maxima of three values at once (same contract)
__device__ __forceinline__ void block_max3_plain(float& a, float& b, float& c, float* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a = fmaxf(a, __shfl_xor(a, off, 64)); b = fmaxf(b, __shfl_xor(b, off, 64)); c = fmaxf(c, __shfl_xor(c, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        scratch[threadIdx.x >> 6] = a; scratch[GEO_W + (threadIdx.x >> 6)] = b; scratch[2 * GEO_W + (threadIdx.x >> 6)] = c;
    }
    barrier_lds_only();
    a = scratch[0]; b = scratch[GEO_W]; c = scratch[2 * GEO_W];
#pragma unroll
    for (int i = 1; i < GEO_W; ++i) {
        a = fmaxf(a, scratch[i]); b = fmaxf(b, scratch[GEO_W + i]); c = fmaxf(c, scratch[2 * GEO_W + i]);
    }
}

// The staging loop's LDS reads and barriers are inline asm.  (1) The compiler puts `s_waitcnt vmcnt(0)` in front of every
// LDS read it can see while an LDS-DMA is outstanding, and __syncthreads() carries one too; vmcnt counts loads AND stores
// in issue order, so either would hold a wave until the plane stores of the chunk before have been acknowledged by L2 --
// with all 1 024 workgroups resident and in step, the chip would alternate between computing and storing.  The loop waits
// only for `vmcnt(number of stores issued after the DMA)`: the copies are older than the stores, so they have landed.
// (2) asm reads are tied to the lgkmcnt wait through "+v" operands (the scheduler may not hoist their uses above it).
__device__ __forceinline__ void lds_read_cube_asm(const float* s_cubes, int tid, float* cu) {
    const unsigned addr = (unsigned)(size_t)(lds_ptr_t)s_cubes + (unsigned)tid * 60u;
    asm volatile("ds_read_b32 %0, %15\n\tds_read_b32 %1, %15 offset:4\n\tds_read_b32 %2, %15 offset:8\n\t"
                 "ds_read_b32 %3, %15 offset:12\n\tds_read_b32 %4, %15 offset:16\n\tds_read_b32 %5, %15 offset:20\n\t"
                 "ds_read_b32 %6, %15 offset:24\n\tds_read_b32 %7, %15 offset:28\n\tds_read_b32 %8, %15 offset:32\n\t"
                 "ds_read_b32 %9, %15 offset:36\n\tds_read_b32 %10, %15 offset:40\n\tds_read_b32 %11, %15 offset:44\n\t"
                 "ds_read_b32 %12, %15 offset:48\n\tds_read_b32 %13, %15 offset:52\n\tds_read_b32 %14, %15 offset:56\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(cu[0]), "=&v"(cu[1]), "=&v"(cu[2]), "=&v"(cu[3]), "=&v"(cu[4]), "=&v"(cu[5]), "=&v"(cu[6]),
                   "=&v"(cu[7]), "=&v"(cu[8]), "=&v"(cu[9]), "=&v"(cu[10]), "=&v"(cu[11]), "=&v"(cu[12]), "=&v"(cu[13]),
                   "=&v"(cu[14])
                 : "v"(addr) : "memory");
}
__device__ __forceinline__ void wait_dma_under_stores(int stores) {       // block-uniform
    if (stores == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (stores == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (stores == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int CPT>
__global__ __launch_bounds__(GEO_T, (CPT <= 4 ? 4 : 1)) void k_project_score_fast(const ScoreArgs a, const float delta) {
    __shared__ __attribute__((aligned(16))) float s_cubes[GEO_T * 16];
    __shared__ double s_red64[GEO_W * 4];
    __shared__ float s_red[GEO_W * 2];
    __shared__ int s_redi[GEO_W];
    __shared__ float s_rect[8];
    __shared__ float s_v[5 * CPT * GEO_T];
    __shared__ int s_cand[GEO_CAND];
    __shared__ int s_cnt;
    const ScoreLds L = {s_cubes, s_red64, s_red, s_redi, s_rect, s_v};

    const int obj = blockIdx.x;
    const int tid = threadIdx.x;
    const int P = a.P;
    const float* cb = a.cubes + (size_t)obj * P * 15;
    // chunk c + 1 streams into the ONE staging buffer while chunk c is evaluated from registers: the buffer is free as soon
    // as every thread has read its cube (LDS-DMA needs 16-byte aligned chunks: P % 4 == 0).  The first copy is the first
    // thing the kernel does: the per-object constants load under it.
    const bool dma = (P & 3) == 0 && ((uintptr_t)a.cubes & 15) == 0;
    const __amdgpu_buffer_rsrc_t rc = cubes_rsrc(cb, P);
    // (the per-object constants are requested BEFORE anything with side effects: loads the compiler can prove unclobbered
    // are scalar loads into SGPRs; behind an `s_sleep` or an asm statement they become vector loads and ~30 VGPRs)
    const bool have_rect = a.rect_pts != nullptr && a.rect_pts[obj * 8] == a.rect_pts[obj * 8];
    ObjConst o;
    load_obj(a, obj, o);
    // the four waves ask for their first cubes GEO_STAGGER x 64 cycles apart: all 1 024 workgroups start together, and
    // 16 MB requested in the same microsecond arrive together 4 us later -- staggered, wave 0 computes while 1..3 still wait
    // (33.2 -> 32.5 us; staggering the four workgroups of a CU on top of it measured no better)
    for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) __builtin_amdgcn_s_sleep(GEO_STAGGER);
    if (dma) stage_cubes_dma(rc, 0, s_cubes);
    // without a rectangle the fallback one is a float64 mean of the EXACT boxes of all cubes: the exact sequence (its
    // first barrier waits for the copy above)
    bool exact_object = !have_rect;
    if (!exact_object) {
    if (tid < 8) s_rect[tid] = a.rect_pts[obj * 8 + tid];
    if (tid == 0) s_cnt = 0;
    const float is0 = 1.0f / o.sg0, is1 = 1.0f / o.sg1, is2 = 1.0f / o.sg2;
    const float d4 = 4.0f * delta;

    float* const v_iou = s_v + 0 * CPT * GEO_T + tid;
    float* const v_gauss = s_v + 1 * CPT * GEO_T + tid;
    float* const v_diff = s_v + 2 * CPT * GEO_T + tid;
    float* const v_s = s_v + 3 * CPT * GEO_T + tid;
    float* const v_err = s_v + 4 * CPT * GEO_T + tid;     // error bound of the ratio difference

    // poison: stays 0 while every intermediate is finite, NaN otherwise (0 * inf = NaN)
    float poison = 0.0f * (((o.a1 + o.gt_ratio) + (is0 + is1)) + is2);
    float lmaxd = 0.0f, lmaxlo = -INFINITY, lmaxs = 0.0f;
    __syncthreads();
    float rx[4], ry[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { rx[q] = s_rect[q * 2]; ry[q] = s_rect[q * 2 + 1]; poison = fmaf(rx[q] + ry[q], 0.0f, poison); }

    const int nstores = (a.out_corners ? 4 : 0) + (a.out_boxes ? 1 : 0);      // per thread and chunk, after the next DMA
#pragma unroll 1
    for (int c = 0; c < CPT; ++c) {
        const int base = c * GEO_T;
        if (base >= P) break;                     // block-uniform
        const int cnt = min(GEO_T, P - base);
        float cu[15];
        if (dma) {
            wait_dma_under_stores(c == 0 ? 0 : nstores);
            lds_read_cube_asm(s_cubes + (tid >> 6) * 1024, tid & 63, cu);      // (rows past cnt: zero bytes, never used)
            if (base + GEO_T < P) stage_cubes_dma(rc, c + 1, s_cubes);         // the wave's own region: no barrier
            asm volatile("" ::: "memory");                // the plane stores below stay younger than the copies
        } else {
            stage_cubes(cb + (size_t)base * 15, cnt, s_cubes);
            __syncthreads();
            if (tid < cnt) {
#pragma unroll
                for (int j = 0; j < 15; ++j) cu[j] = s_cubes[tid * 15 + j];
            }
            __syncthreads();
        }
        if (tid < cnt) {
            // corners and homogeneous image points in PAIRS (0,1) (2,3) (4,5) (6,7) on the packed float32 pipe
            // (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per lane and instruction, unfused, in the order of
            // cube_corner3d / project_h -- the same bits at half the issue slots)
            float u[8], v[8];
            {
                const float hw = cu[3] / 2.0f, hh = cu[4] / 2.0f, hl = cu[5] / 2.0f;
                v2f pz = {0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v2f vx = (j & 1) ? (v2f){hl, -hl} : (v2f){-hl, hl};     // +l/2 for corners {1,2,5,6}
                    const float vy = (j & 1) ? hh : -hh;                          // +h/2 for {2,3,6,7}
                    const float vz = (j & 2) ? hw : -hw;                          // +w/2 for {4,5,6,7}
                    v2f t, X, Y, Z;
                    t = cu[6] * vx;  t = t + cu[7] * vy;  t = t + cu[8] * vz;  X = t + cu[0];
                    t = cu[9] * vx;  t = t + cu[10] * vy; t = t + cu[11] * vz; Y = t + cu[1];
                    t = cu[12] * vx; t = t + cu[13] * vy; t = t + cu[14] * vz; Z = t + cu[2];
                    const v2f p0 = (o.K[0] * X + o.K[1] * Y) + o.K[2] * Z;
                    const v2f p1 = (o.K[3] * X + o.K[4] * Y) + o.K[5] * Z;
                    const v2f p2 = (o.K[6] * X + o.K[7] * Y) + o.K[8] * Z;
                    const v2f r = {__builtin_amdgcn_rcpf(p2.x), __builtin_amdgcn_rcpf(p2.y)};
                    const v2f ur = p0 * r, vr = p1 * r;
                    pz = __builtin_elementwise_fma(ur, (v2f){0.0f, 0.0f}, pz);
                    pz = __builtin_elementwise_fma(vr, (v2f){0.0f, 0.0f}, pz);
                    u[2 * j] = __builtin_amdgcn_fmed3f(ur.x, a.cl.lo0, a.cl.hi0);
                    u[2 * j + 1] = __builtin_amdgcn_fmed3f(ur.y, a.cl.lo0, a.cl.hi0);
                    v[2 * j] = __builtin_amdgcn_fmed3f(vr.x, a.cl.lo1, a.cl.hi1);
                    v[2 * j + 1] = __builtin_amdgcn_fmed3f(vr.y, a.cl.lo1, a.cl.hi1);
                }
                poison = poison + (pz.x + pz.y);
            }
            const float b0 = fminf(fminf(fminf(u[0], u[1]), fminf(u[2], u[3])), fminf(fminf(u[4], u[5]), fminf(u[6], u[7])));
            const float b2 = fmaxf(fmaxf(fmaxf(u[0], u[1]), fmaxf(u[2], u[3])), fmaxf(fmaxf(u[4], u[5]), fmaxf(u[6], u[7])));
            const float b1 = fminf(fminf(fminf(v[0], v[1]), fminf(v[2], v[3])), fminf(fminf(v[4], v[5]), fminf(v[6], v[7])));
            const float b3 = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
            const size_t gi = (size_t)obj * P + base + tid;
            if (a.out_corners && (tid | 63) < cnt) {              // whole wave active
                store_corners_transposed(u, v, a.out_corners + ((size_t)obj * P + base + (tid & ~63)) * 16,
                                         s_v + c * GEO_T + (tid & ~63), CPT * GEO_T);
            } else if (a.out_corners) {
                float4* oc = reinterpret_cast<float4*>(a.out_corners + gi * 16);
                oc[0] = make_float4(u[0], v[0], u[1], v[1]);
                oc[1] = make_float4(u[2], v[2], u[3], v[3]);
                oc[2] = make_float4(u[4], v[4], u[5], v[5]);
                oc[3] = make_float4(u[6], v[6], u[7], v[7]);
            }
            if (a.out_boxes) reinterpret_cast<float4*>(a.out_boxes)[gi] = make_float4(b0, b1, b2, b3);
            const float bw = b2 - b0, bh = b3 - b1;
            const float w = fmaxf(fminf(o.r2, b2) - fmaxf(o.r0, b0), 0.0f);
            const float h = fmaxf(fminf(o.r3, b3) - fmaxf(o.r1, b1), 0.0f);
            const float inter = w * h;
            const float iou = inter > 0.0f ? inter * __builtin_amdgcn_rcpf((o.a1 + bw * bh) - inter) : 0.0f;
            // a box thinner than a pixel that overlaps the reference box: its IoU is only good to delta / width, which the
            // 4e-3 interval of the combined score does not cover -> the exact sequence for this object (rare: off-screen
            // cubes have no overlap, cubes in view project to several pixels)
            if (fminf(bw, bh) < 1.0f && inter > 0.0f) poison = nan_f();
            const float z0 = (cu[3] - o.mu0) * is0, z1 = (cu[4] - o.mu1) * is1, z2 = (cu[5] - o.mu2) * is2;
            const float kE = -0.5f * 1.4426950408889634f;
            const float gauss = ((__builtin_amdgcn_exp2f(kE * (z0 * z0)) + __builtin_amdgcn_exp2f(kE * (z1 * z1)))
                                 + __builtin_amdgcn_exp2f(kE * (z2 * z2))) * 0.333333343f;
            const float rh = __builtin_amdgcn_rcpf(bh);
            const float pr = bw * rh;
            const float diff = fabsf(o.gt_ratio - pr);
            const float err = fmaf(2.0f * d4 * rh, 1.0f + pr, 4e-6f * (pr + fabsf(o.gt_ratio)));
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float best = INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v2f dx = rx[q] - (v2f){u[2 * j], u[2 * j + 1]}, dy = ry[q] - (v2f){v[2 * j], v[2 * j + 1]};
                    const v2f d2 = __builtin_elementwise_fma(dy, dy, dx * dx);
                    best = fminf(fminf(best, d2.x), d2.y);
                }
                acc += __builtin_amdgcn_sqrtf(best);
            }
            const float sc = acc * 0.25f;
            poison = fmaf(((iou + gauss) + (diff + err)) + sc, 0.0f, poison);
            VAT(v_iou, c) = iou;
            VAT(v_gauss, c) = gauss;
            VAT(v_diff, c) = diff;
            VAT(v_s, c) = sc;
            VAT(v_err, c) = err;
            lmaxd = fmaxf(lmaxd, diff);
            lmaxlo = fmaxf(lmaxlo, diff - err);
            lmaxs = fmaxf(lmaxs, sc);
        }
    }

    // ---------------- per-object normalisers of the fast planes
    float maxdiff = lmaxd, lo_d = lmaxlo, maxs = lmaxs;
    block_max3_plain(maxdiff, lo_d, maxs, reinterpret_cast<float*>(s_red64));
    const float rmd = __builtin_amdgcn_rcpf(maxdiff), rms = __builtin_amdgcn_rcpf(maxs);
    poison = fmaf(rmd + rms, 0.0f, poison);          // a zero normaliser: the exact path decides what x / 0 means

    // ---------------- maximum of the fast combined score
    float lmaxc = 0.0f;
#pragma unroll 1
    for (int c = 0; c < CPT; ++c) {
        if (c * GEO_T + tid < P) {
            const float dim = (1.0f - VAT(v_diff, c) * rmd) * VAT(v_gauss, c);
            const float cor = 1.0f - VAT(v_s, c) * rms;
            lmaxc = fmaxf(lmaxc, (VAT(v_iou, c) * dim) * cor);
        }
    }
    const float maxc = block_max_plain(lmaxc, s_red);

    // ---------------- candidates: cubes whose fast value could be the exact maximum of (ratio difference | chamfer |
    // combined score).  bit 16: ratio, 17: chamfer, 18: combined
    const float th_s = maxs - (2.0f * d4 + 4e-6f * maxs);
    const float th_c = maxc - (4e-3f * maxc + 1e-7f);
#pragma unroll 1
    for (int c = 0; c < CPT; ++c) {
        const int p = c * GEO_T + tid;
        if (p < P) {
            const float dim = (1.0f - VAT(v_diff, c) * rmd) * VAT(v_gauss, c);
            const float cor = 1.0f - VAT(v_s, c) * rms;
            const float comb = (VAT(v_iou, c) * dim) * cor;
            const int f = ((VAT(v_diff, c) + VAT(v_err, c) >= lo_d) ? 1 << 16 : 0) | ((VAT(v_s, c) >= th_s) ? 1 << 17 : 0)
                        | ((comb >= th_c) ? 1 << 18 : 0);
            if (f) {
                const int j = atomicAdd(&s_cnt, 1);
                if (j < GEO_CAND) s_cand[j] = p | f;
            }
        }
    }
    if ((poison != poison) || !(maxc > 1e-6f)) atomicAdd(&s_cnt, GEO_CAND + 1);      // -> the exact sequence
    barrier_lds_only();
    const int ncand = s_cnt;
    const bool bad = false;
    exact_object = bad || ncand > GEO_CAND;
    if (!exact_object) {
    if (a.stats && tid == 0) atomicAdd(a.stats + 1, (unsigned long long)ncand);
    // ---------------- exact re-evaluation of the candidates: 8 lanes per candidate, 32 candidates per round (the
    // objects resident on one CU start in different waves); results parked in the staging buffer, free by now
    const int slot = (tid + 64 * ((blockIdx.x >> 8) & 3)) & (GEO_T - 1);
    if (ncand <= 8) {
        // the usual case (3 candidates on average): ONE wave re-evaluates them, 8 lanes each, and reduces across its eight
        // groups with shuffles -- no LDS round trip and none of the seven workgroup barriers of the general path below; the
        // other three waves go straight to the plane stores
        if (slot < 64) {
            const int j = slot >> 3;
            const bool act = j < ncand;
            float e_iou = 0.0f, e_gauss = 0.0f, e_diff = 0.0f, e_s = 0.0f;
            int cp = 0, cf = 0;
            if (act) {
                cp = s_cand[j] & 0xffff;
                cf = s_cand[j] >> 16;
                float cu[15];
#pragma unroll
                for (int k = 0; k < 15; ++k) cu[k] = cb[(size_t)cp * 15 + k];
                cand_exact8(cu, slot & 7, o, a.cl, s_rect, e_iou, e_gauss, e_diff, e_s);
            }
            // torch.max semantics over the flagged candidates (block_max_nanprop): NaN if any NaN
            const bool vd = act && (cf & 1), vs = act && (cf & 2);
            float md = vd && e_diff == e_diff ? e_diff : -INFINITY, ms = vs && e_s == e_s ? e_s : -INFINITY;
            int nd = vd && e_diff != e_diff, ns = vs && e_s != e_s;
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) {
                md = fmaxf(md, __shfl_xor(md, m, 64)); ms = fmaxf(ms, __shfl_xor(ms, m, 64));
                nd |= __shfl_xor(nd, m, 64); ns |= __shfl_xor(ns, m, 64);
            }
            const float xmaxdiff = nd ? nan_f() : md, xmaxs = ns ? nan_f() : ms;
            float bestv = 0.0f;
            int besti = -1;
            if (act && (cf & 4)) {
                const float dim = (1.0f - e_diff / xmaxdiff) * e_gauss;
                const float cor = 1.0f - e_s / xmaxs;
                bestv = (e_iou * dim) * cor;
                besti = cp;
            }
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) {
                const float ov = __shfl_xor(bestv, m, 64);
                const int oi = __shfl_xor(besti, m, 64);
                if (arg_better(ov, oi, bestv, besti)) { bestv = ov; besti = oi; }
            }
            if (slot == 0) {
                a.out_argmax[obj] = besti < 0 ? 0 : besti;
                if (a.out_best) a.out_best[obj] = besti < 0 ? 0.0f : bestv;
            }
        }
    } else {
#pragma unroll 1
    for (int j0 = 0; j0 < ncand; j0 += GEO_T / 8) {
        const int j = j0 + (slot >> 3);
        if (j < ncand) {                          // whole groups of 8 lanes
            const int cp = s_cand[j] & 0xffff;
            float cu[15];
#pragma unroll
            for (int k = 0; k < 15; ++k) cu[k] = cb[(size_t)cp * 15 + k];
            float e_iou, e_gauss, e_diff, e_s;
            cand_exact8(cu, slot & 7, o, a.cl, s_rect, e_iou, e_gauss, e_diff, e_s);
            if ((slot & 7) == 0) {
                s_cubes[0 * GEO_CAND + j] = e_iou; s_cubes[1 * GEO_CAND + j] = e_gauss;
                s_cubes[2 * GEO_CAND + j] = e_diff; s_cubes[3 * GEO_CAND + j] = e_s;
            }
        }
    }
    barrier_lds_only();
    const bool act = tid < ncand;
    const int cp = act ? s_cand[tid] & 0xffff : 0, cf = act ? s_cand[tid] >> 16 : 0;
    const float e_iou = s_cubes[0 * GEO_CAND + tid], e_gauss = s_cubes[1 * GEO_CAND + tid];
    const float e_diff = s_cubes[2 * GEO_CAND + tid], e_s = s_cubes[3 * GEO_CAND + tid];
    const float xmaxdiff = block_max_nanprop<true>(e_diff, act && (cf & 1), s_red);
    const float xmaxs = block_max_nanprop<true>(e_s, act && (cf & 2), s_red);
    float bestv = 0.0f;
    int besti = -1;
    if (act && (cf & 4)) {
        const float dim = (1.0f - e_diff / xmaxdiff) * e_gauss;
        const float cor = 1.0f - e_s / xmaxs;
        bestv = (e_iou * dim) * cor;
        besti = cp;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(bestv, off, 64);
        const int oi = __shfl_down(besti, off, 64);
        if (arg_better(ov, oi, bestv, besti)) { bestv = ov; besti = oi; }
    }
    barrier_lds_only();
    if ((tid & 63) == 0) { s_red[tid >> 6] = bestv; s_redi[tid >> 6] = besti; }
    barrier_lds_only();
    if (tid == 0) {
        float bv = s_red[0];
        int bi = s_redi[0];
        for (int i = 1; i < GEO_W; ++i)
            if (arg_better(s_red[i], s_redi[i], bv, bi)) { bv = s_red[i]; bi = s_redi[i]; }
        a.out_argmax[obj] = bi < 0 ? 0 : bi;
        if (a.out_best) a.out_best[obj] = bi < 0 ? 0.0f : bv;
    }
    }
    // ---------------- the four score planes, last: nothing waits for these stores
    if (a.out_iou || a.out_dim || a.out_corner || a.out_combined) {
#pragma unroll 1
        for (int c = 0; c < CPT; ++c) {
            const int p = c * GEO_T + tid;
            if (p < P) {
                const float iou = VAT(v_iou, c);
                const float dim = (1.0f - VAT(v_diff, c) * rmd) * VAT(v_gauss, c);
                const float cor = 1.0f - VAT(v_s, c) * rms;
                const size_t gi = (size_t)obj * P + p;
                if (a.out_iou) a.out_iou[gi] = iou;
                if (a.out_dim) a.out_dim[gi] = dim;
                if (a.out_corner) a.out_corner[gi] = cor;
                if (a.out_combined) a.out_combined[gi] = (iou * dim) * cor;
            }
        }
    }
    }
    }
    if (exact_object) {               // block-uniform
        if (a.stats && tid == 0) atomicAdd(a.stats, 1ull);
        score_object_exact<CPT>(a, L, obj);
    }
}

static int project_score_launch(cr_ctx* ctx, const float* cubes, int64_t N, int64_t P,
                                const float* K, int k_per_object, float im_w, float im_h,
                                const float* ref_boxes, const float* prior_mu, const float* prior_sigma,
                                const float* rect_pts, float* out_corners, float* out_boxes,
                                float* out_iou, float* out_dim, float* out_corner, float* out_combined,
                                int64_t* out_argmax, float* out_best, const float* iou_boxes, bool fast, int64_t* stats,
                                const char* who) {
    CR_CHECK_ARG(ctx != nullptr, "%s: ctx is NULL", who);
    CR_CHECK_ARG(N >= 0 && P >= 0, "%s: negative N/P", who);
    if (N == 0) return CR_OK;
    CR_CHECK_ARG(P >= 1 && P <= 4096, "%s: P=%lld outside [1,4096]", who, (long long)P);
    CR_CHECK_ARG(N <= 0x7fffffff, "%s: N too large", who);
    CR_CHECK_ARG(cubes && K && ref_boxes && prior_mu && prior_sigma && out_argmax, "%s: NULL required pointer", who);
    ScoreArgs a;
    a.cubes = cubes; a.P = (int)P; a.Kmat = K; a.k_per_object = k_per_object;
    // python int() truncation toward zero, spaces.py:241-242
    a.cl.lo0 = (float)(int)(-(double)im_w / 2 + 1);
    a.cl.hi0 = (float)(int)((double)im_w - 1 + (double)im_w);
    a.cl.lo1 = (float)(int)(-(double)im_h / 2 + 1);
    a.cl.hi1 = (float)(int)((double)im_h - 1 + (double)im_h);
    a.ref_boxes = ref_boxes; a.prior_mu = prior_mu; a.prior_sigma = prior_sigma; a.rect_pts = rect_pts;
    a.out_corners = out_corners; a.out_boxes = out_boxes; a.out_iou = out_iou; a.out_dim = out_dim;
    a.out_corner = out_corner; a.out_combined = out_combined; a.out_argmax = out_argmax; a.out_best = out_best;
    a.iou_boxes = iou_boxes;
    a.stats = reinterpret_cast<unsigned long long*>(stats);
    dim3 grid((unsigned)N), block(GEO_T);
    if (fast) {
        const float mag = fmaxf(fmaxf(fabsf(a.cl.lo0), fabsf(a.cl.hi0)), fmaxf(fabsf(a.cl.lo1), fabsf(a.cl.hi1)));
        const float delta = mag * (1.0f / 4194304.0f);
        if (P <= 4 * GEO_T) hipLaunchKernelGGL(k_project_score_fast<4>, grid, block, 0, ctx->stream, a, delta);
        else hipLaunchKernelGGL(k_project_score_fast<16>, grid, block, 0, ctx->stream, a, delta);
    } else {
        if (P <= 4 * GEO_T) hipLaunchKernelGGL(k_project_score<4>, grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL(k_project_score<16>, grid, block, 0, ctx->stream, a);
    }
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_cubes_project_score(cr_ctx* ctx, const float* cubes, int64_t N, int64_t P,
                                      const float* K, int k_per_object, float im_w, float im_h,
                                      const float* ref_boxes, const float* prior_mu, const float* prior_sigma,
                                      const float* rect_pts, float* out_corners, float* out_boxes,
                                      float* out_iou, float* out_dim, float* out_corner, float* out_combined,
                                      int64_t* out_argmax, float* out_best, const float* iou_boxes) {
    return project_score_launch(ctx, cubes, N, P, K, k_per_object, im_w, im_h, ref_boxes, prior_mu, prior_sigma, rect_pts,
                                out_corners, out_boxes, out_iou, out_dim, out_corner, out_combined, out_argmax, out_best,
                                iou_boxes, false, nullptr, "cr_cubes_project_score");
}

extern "C" int cr_cubes_project_score_fast(cr_ctx* ctx, const float* cubes, int64_t N, int64_t P,
                                           const float* K, int k_per_object, float im_w, float im_h,
                                           const float* ref_boxes, const float* prior_mu, const float* prior_sigma,
                                           const float* rect_pts, float* out_corners, float* out_boxes,
                                           float* out_iou, float* out_dim, float* out_corner, float* out_combined,
                                           int64_t* out_argmax, float* out_best, const float* iou_boxes, int64_t* stats) {
    return project_score_launch(ctx, cubes, N, P, K, k_per_object, im_w, im_h, ref_boxes, prior_mu, prior_sigma, rect_pts,
                                out_corners, out_boxes, out_iou, out_dim, out_corner, out_combined, out_argmax, out_best,
                                iou_boxes, true, stats, "cr_cubes_project_score_fast");
}

// ---------------------------------------------------------------------------
// get_cuboid_verts_faces (verts only)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(GEO_T) void k_cuboid_corners(const float* __restrict__ box6,
                                                           const float* __restrict__ R, int64_t n,
                                                           float* __restrict__ verts) {
    const int64_t i = (int64_t)blockIdx.x * GEO_T + threadIdx.x;
    if (i >= n) return;
    float c[15];
#pragma unroll
    for (int j = 0; j < 6; ++j) c[j] = box6[i * 6 + j];
#pragma unroll
    for (int j = 0; j < 9; ++j) c[6 + j] = R[i * 9 + j];
    float X[8], Y[8], Z[8];
    cube_corners3d(c, X, Y, Z);
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        verts[i * 24 + v * 3 + 0] = X[v];
        verts[i * 24 + v * 3 + 1] = Y[v];
        verts[i * 24 + v * 3 + 2] = Z[v];
    }
}

extern "C" int cr_cuboid_corners(cr_ctx* ctx, const float* box6, const float* R, int64_t n, float* verts) {
    CR_CHECK_ARG(ctx != nullptr, "cr_cuboid_corners: ctx is NULL");
    CR_CHECK_ARG(n >= 0, "cr_cuboid_corners: negative n");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(box6 && R && verts, "cr_cuboid_corners: NULL pointer");
    hipLaunchKernelGGL(k_cuboid_corners, dim3((unsigned)cr_cdiv(n, GEO_T)), dim3(GEO_T), 0, ctx->stream, box6, R, n,
                       verts);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// K18: proposal sampler (one workgroup per object, P <= 1024)
// ---------------------------------------------------------------------------
#define PROP_MAXP 1024
#define PROP_SLOTS (PROP_MAXP / GEO_T)

// in-LDS bitonic sort of 1024 floats (ascending; padded with +inf)
__device__ __forceinline__ void bitonic_sort_1024(float* a) {
    for (int k = 2; k <= PROP_MAXP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < PROP_MAXP / 2; t += GEO_T) {
                const int i = ((t / j) * 2 * j) + (t % j);
                const int l = i + j;
                const bool up = ((i & k) == 0);
                const float x = a[i], y = a[l];
                if ((x > y) == up) { a[i] = y; a[l] = x; }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(GEO_T) void k_propose(
    const float* __restrict__ boxes, const float* __restrict__ depth, int H, int W,
    const float* __restrict__ prior_mu, const float* __restrict__ prior_sigma, const float* __restrict__ Kmat,
    int N, int P, const float* __restrict__ dim_normals, int rounds, const float* __restrict__ ctr_normals,
    const int32_t* __restrict__ yaw_idx, const float* __restrict__ normal, float* __restrict__ out_cubes,
    int32_t* __restrict__ out_exhausted, const int32_t* __restrict__ img_idx) {
    __shared__ float s_sort[PROP_MAXP];
    __shared__ double s_red64[GEO_W];
    __shared__ float s_tab[36 * 9];
    __shared__ float s_stat[6];       // median x,y,z ; std x,y,z

    const int obj = blockIdx.x, tid = threadIdx.x;
    const float b0 = boxes[obj * 4 + 0], b1 = boxes[obj * 4 + 1], b2 = boxes[obj * 4 + 2], b3 = boxes[obj * 4 + 3];
    if (img_idx) {                         // objects of several images in one launch: per-image depth, K and normal
        const int im = img_idx[obj];
        depth += (size_t)im * H * W;
        Kmat += im * 9;
        normal += im * 3;
    }
    const float K00 = Kmat[0], K02 = Kmat[2], K12 = Kmat[5];

    // 36-yaw table from the ground normal: utils.py:112-146, proposals.py:404-405
    if (tid < 36) {
        const float n0 = normal[0], n1 = normal[1], n2 = normal[2];
        float p0, p1, p2;
        if (n0 == 0.0f) { p0 = 0.0f; p1 = n2; p2 = -n1; }
        else {
            float mag = sqrtf((n1 * n1 + n0 * n0) + 0.0f);
            mag = fmaxf(mag, 1e-8f);
            p0 = n1 / mag; p1 = -n0 / mag; p2 = 0.0f / mag;
        }
        // torch.linspace(0, pi, 36): symmetric fill
        const float endv = 3.14159274101257324f, step = endv / 35.0f;
        const float th = tid < 18 ? 0.0f + step * (float)tid : endv - step * (float)(35 - tid);
        const float ct = cosf(th), st = sinf(th);
        const float k0 = n1 * p2 - n2 * p1, k1 = n2 * p0 - n0 * p2, k2 = n0 * p1 - n1 * p0;
        const float kd = (n0 * p0 + n1 * p1) + n2 * p2;
        const float omc = 1.0f - ct;
        const float x0 = (p0 * ct + k0 * st) + (n0 * kd) * omc;
        const float x1 = (p1 * ct + k1 * st) + (n1 * kd) * omc;
        const float x2 = (p2 * ct + k2 * st) + (n2 * kd) * omc;
        const float y0 = n1 * x2 - n2 * x1, y1 = n2 * x0 - n0 * x2, y2 = n0 * x1 - n1 * x0;
        float* t = s_tab + tid * 9;            // columns (x, n, y)
        t[0] = x0; t[1] = n0; t[2] = y0;
        t[3] = x1; t[4] = n1; t[5] = y1;
        t[6] = x2; t[7] = n2; t[8] = y2;
    }

    const float wd = b2 - b0, ht = b3 - b1;
    const float x_lo = b0 + wd / 4.0f, x_hi = b2 - wd / 4.0f;
    const float y_lo = b1 + ht / 4.0f, y_hi = b3 - ht / 4.0f;
    const float x_sp = (x_hi - x_lo) / (float)(P - 1), y_sp = (y_hi - y_lo) / (float)(P - 1);

    const float mu[3] = {prior_mu[obj * 3], prior_mu[obj * 3 + 1], prior_mu[obj * 3 + 2]};
    const float sg[3] = {prior_sigma[obj * 3], prior_sigma[obj * 3 + 1] * 1.1f, prior_sigma[obj * 3 + 2]};
    const float hi[3] = {mu[0] + 2.0f * prior_sigma[obj * 3], mu[1] + 2.2f * prior_sigma[obj * 3 + 1],
                         mu[2] + 2.0f * prior_sigma[obj * 3 + 2]};

    float vx[PROP_SLOTS], vy[PROP_SLOTS], vz[PROP_SLOTS], dims[PROP_SLOTS][3];
    int exhausted = 0;
#pragma unroll
    for (int s = 0; s < PROP_SLOTS; ++s) {
        const int p = s * GEO_T + tid;
        vx[s] = vy[s] = vz[s] = INFINITY;
        if (p < P) {
            // vectorized_linspace(...).long(): trunc toward zero (utils.py:170-177, proposals.py:360-363)
            const long xg = (long)truncf((float)p * x_sp + x_lo);
            const long yg = (long)truncf((float)p * y_sp + y_lo);
            long xi = xg < 0 ? xg + W : xg, yi = yg < 0 ? yg + H : yg;      // python negative-index wrap
            xi = min(max(xi, 0L), (long)W - 1);
            yi = min(max(yi, 0L), (long)H - 1);
            const float d = depth[yi * W + xi];
            const float ox = (float)xg - K02, oy = (float)yg - K12, a = K00;
            const float ang_x = atan2f(ox, a);
            const float dxc = sqrtf(ox * ox + a * a);
            const float ang_d = atan2f(oy, dxc);
            const float y = d * sinf(ang_d);
            const float dx = sqrtf(d * d - y * y);
            const float x = dx * sinf(ang_x);
            const float zt = sqrtf(dx * dx - x * x);
            // truncated normals for w,h,l with `rounds` pre-drawn rejection rounds
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float v = mu[k] + sg[k] * dim_normals[(((size_t)0 * 3 + k) * N + obj) * P + p];
                for (int r = 1; r < rounds && (v < 0.05f || v > hi[k]); ++r)
                    v = mu[k] + sg[k] * dim_normals[(((size_t)r * 3 + k) * N + obj) * P + p];
                if (v < 0.05f || v > hi[k]) exhausted++;
                dims[s][k] = v;
            }
            vx[s] = x; vy[s] = y; vz[s] = zt + dims[s][2] / 2.0f;
        }
    }
    if (exhausted) atomicAdd(out_exhausted, exhausted);

    // median (lower) + unbiased std over the P samples of x, y, z
    for (int k = 0; k < 3; ++k) {
        __syncthreads();
        double lsum = 0.0;
#pragma unroll
        for (int s = 0; s < PROP_SLOTS; ++s) {
            const int p = s * GEO_T + tid;
            const float v = k == 0 ? vx[s] : (k == 1 ? vy[s] : vz[s]);
            s_sort[p] = p < P ? v : INFINITY;
            if (p < P) lsum += (double)v;
        }
        const double mean = block_sum_f64(lsum, s_red64) / (double)P;
        double lsq = 0.0;
#pragma unroll
        for (int s = 0; s < PROP_SLOTS; ++s) {
            const int p = s * GEO_T + tid;
            if (p < P) {
                const double dv = (double)(k == 0 ? vx[s] : (k == 1 ? vy[s] : vz[s])) - mean;
                lsq += dv * dv;
            }
        }
        const double var = block_sum_f64(lsq, s_red64) / (double)(P - 1);
        bitonic_sort_1024(s_sort);
        if (tid == 0) {
            s_stat[k] = s_sort[(P - 1) / 2];
            s_stat[3 + k] = (float)sqrt(var);
        }
    }
    __syncthreads();
    const float mx = 1.15f * s_stat[0] + 0.0f, sx = s_stat[3] * 1.2f;
    const float my = 1.1f * s_stat[1] + 0.0f, sy = s_stat[4] * 0.8f;
    const float mz = 0.85f * s_stat[2] + 0.35f, sz = s_stat[5] * 1.2f;
#pragma unroll
    for (int s = 0; s < PROP_SLOTS; ++s) {
        const int p = s * GEO_T + tid;
        if (p < P) {
            const size_t gi = (size_t)obj * P + p;
            float* o = out_cubes + gi * 15;
            o[0] = mx + sx * ctr_normals[((size_t)0 * N + obj) * P + p];
            o[1] = my + sy * ctr_normals[((size_t)1 * N + obj) * P + p];
            o[2] = mz + sz * ctr_normals[((size_t)2 * N + obj) * P + p];
            o[3] = dims[s][0]; o[4] = dims[s][1]; o[5] = dims[s][2];
            int yi = yaw_idx[gi];
            yi = min(max(yi, 0), 35);
            const float* t = s_tab + yi * 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) o[6 + j] = t[j];
        }
    }
}

extern "C" int cr_propose(cr_ctx* ctx, const float* boxes, int64_t N, const float* depth, int H, int W,
                          const float* prior_mu, const float* prior_sigma, const float* K, int64_t P,
                          const float* dim_normals, int rounds, const float* ctr_normals,
                          const int32_t* yaw_idx, const float* normal, float* out_cubes,
                          int32_t* out_exhausted) {
    CR_CHECK_ARG(ctx != nullptr, "cr_propose: ctx is NULL");
    CR_CHECK_ARG(N >= 0, "cr_propose: negative N");
    if (N == 0) return CR_OK;
    CR_CHECK_ARG(P >= 2 && P <= PROP_MAXP, "cr_propose: P=%lld outside [2,%d]", (long long)P, PROP_MAXP);
    CR_CHECK_ARG(rounds >= 1, "cr_propose: rounds must be >= 1");
    CR_CHECK_ARG(H > 0 && W > 0, "cr_propose: bad depth shape");
    CR_CHECK_ARG(boxes && depth && prior_mu && prior_sigma && K && dim_normals && ctr_normals && yaw_idx &&
                     normal && out_cubes && out_exhausted,
                 "cr_propose: NULL pointer");
    CR_HIP(hipMemsetAsync(out_exhausted, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(k_propose, dim3((unsigned)N), dim3(GEO_T), 0, ctx->stream, boxes, depth, H, W, prior_mu,
                       prior_sigma, K, (int)N, (int)P, dim_normals, rounds, ctr_normals, yaw_idx, normal, out_cubes,
                       out_exhausted, (const int32_t*)nullptr);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_propose_batched(cr_ctx* ctx, const float* boxes, const int32_t* img_idx, int64_t N, const float* depth,
                                  int B, int H, int W, const float* prior_mu, const float* prior_sigma, const float* K,
                                  int64_t P, const float* dim_normals, int rounds, const float* ctr_normals,
                                  const int32_t* yaw_idx, const float* normals, float* out_cubes, int32_t* out_exhausted) {
    CR_CHECK_ARG(ctx != nullptr, "cr_propose_batched: ctx is NULL");
    CR_CHECK_ARG(N >= 0 && B >= 1, "cr_propose_batched: bad N / B");
    if (N == 0) return CR_OK;
    CR_CHECK_ARG(P >= 2 && P <= PROP_MAXP, "cr_propose_batched: P=%lld outside [2,%d]", (long long)P, PROP_MAXP);
    CR_CHECK_ARG(rounds >= 1 && H > 0 && W > 0, "cr_propose_batched: bad rounds / depth shape");
    CR_CHECK_ARG(boxes && img_idx && depth && prior_mu && prior_sigma && K && dim_normals && ctr_normals && yaw_idx &&
                     normals && out_cubes && out_exhausted,
                 "cr_propose_batched: NULL pointer");
    CR_HIP(hipMemsetAsync(out_exhausted, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(k_propose, dim3((unsigned)N), dim3(GEO_T), 0, ctx->stream, boxes, depth, H, W, prior_mu,
                       prior_sigma, K, (int)N, (int)P, dim_normals, rounds, ctr_normals, yaw_idx, normals, out_cubes,
                       out_exhausted, img_idx);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// ---------------------------------------------------------------------------
// K21: parallel RANSAC plane.  One workgroup per RANSAC_TPB candidate triples counts
// their inliers over all points (points stay L2-resident: Q*12 B ~ 126 KB at 512^2).
// ---------------------------------------------------------------------------
#define RANSAC_TPB 8      // candidate planes per workgroup: every point read scores 8 hypotheses
// compacted (optional): the eligible points of every image packed to the front of its (Q, 3) slab by k_ransac_compact, and
// how many there are -- the count loop then runs over them alone (a ground mask covers ~40 % of the strided pixels)
__global__ __launch_bounds__(GEO_T) void k_ransac_count(const float* __restrict__ pts, int Q,
                                                         const int32_t* __restrict__ triples, int T, float thresh,
                                                         float* __restrict__ eqs, int32_t* __restrict__ counts,
                                                         const unsigned char* __restrict__ eligible,
                                                         const float* __restrict__ compacted = nullptr,
                                                         const int32_t* __restrict__ n_compacted = nullptr) {
    __shared__ int s_cnt[GEO_W][RANSAC_TPB];
    __shared__ float s_pl[RANSAC_TPB][5];   // cx, cy, cz, k, den
    const int t0 = blockIdx.x * RANSAC_TPB, tid = threadIdx.x;
    pts += (size_t)blockIdx.y * Q * 3;      // blockIdx.y = image of a batched fit (cr_ransac_plane_batched)
    triples += (size_t)blockIdx.y * T * 3;
    eqs += (size_t)blockIdx.y * T * 4;
    counts += (size_t)blockIdx.y * T;
    if (eligible) eligible += (size_t)blockIdx.y * Q;
    if (tid < RANSAC_TPB && t0 + tid < T) {
        const int t = t0 + tid;
        const int i0 = triples[t * 3], i1 = triples[t * 3 + 1], i2 = triples[t * 3 + 2];
        const float ax = pts[i1 * 3] - pts[i0 * 3], ay = pts[i1 * 3 + 1] - pts[i0 * 3 + 1], az = pts[i1 * 3 + 2] - pts[i0 * 3 + 2];
        const float bx = pts[i2 * 3] - pts[i0 * 3], by = pts[i2 * 3 + 1] - pts[i0 * 3 + 1], bz = pts[i2 * 3 + 2] - pts[i0 * 3 + 2];
        float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        const float nrm = sqrtf((cx * cx + cy * cy) + cz * cz);
        cx = cx / nrm; cy = cy / nrm; cz = cz / nrm;
        s_pl[tid][0] = cx; s_pl[tid][1] = cy; s_pl[tid][2] = cz;
        s_pl[tid][3] = -((cx * pts[i1 * 3] + cy * pts[i1 * 3 + 1]) + cz * pts[i1 * 3 + 2]);
        s_pl[tid][4] = sqrtf((cx * cx + cy * cy) + cz * cz);
    }
    __syncthreads();
    const int nh = min(RANSAC_TPB, T - t0);
    float pl[RANSAC_TPB][5];
    int cnt[RANSAC_TPB];
#pragma unroll
    for (int h = 0; h < RANSAC_TPB; ++h) {
        cnt[h] = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) pl[h][j] = s_pl[h < nh ? h : 0][j];
    }
    const float* lp = pts;
    int Ql = Q;
    if (compacted) { lp = compacted + (size_t)blockIdx.y * Q * 3; Ql = n_compacted[blockIdx.y]; eligible = nullptr; }
    for (int q = tid; q < Ql; q += GEO_T) {
        const float px = lp[q * 3], py = lp[q * 3 + 1], pz = lp[q * 3 + 2];
        const bool ok = !eligible || eligible[q];
#pragma unroll
        for (int h = 0; h < RANSAC_TPB; ++h) {
            // the reference tests |num / den| <= thresh.  Division is monotonic and thresh is a float, so the outcome is
            // decided by |num| against thresh * den except within a few ulp of equality; only there is the division done.
            const float a = fabsf(((pl[h][0] * px + pl[h][1] * py) + pl[h][2] * pz) + pl[h][3]);
            const float b = thresh * pl[h][4];
            bool in = a < b;
            if (fabsf(a - b) <= b * 4.8e-7f) in = a / pl[h][4] <= thresh;
            cnt[h] += (in && ok) ? 1 : 0;
        }
    }
#pragma unroll
    for (int h = 0; h < RANSAC_TPB; ++h) {
        int c = cnt[h];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if ((tid & 63) == 0) s_cnt[tid >> 6][h] = c;
    }
    __syncthreads();
    if (tid < nh) {
        int c = 0;
        for (int i = 0; i < GEO_W; ++i) c += s_cnt[i][tid];
        const int t = t0 + tid;
        counts[t] = c;
        eqs[t * 4] = s_pl[tid][0]; eqs[t * 4 + 1] = s_pl[tid][1]; eqs[t * 4 + 2] = s_pl[tid][2]; eqs[t * 4 + 3] = s_pl[tid][3];
    }
}

// ordered compaction of the eligible points of image blockIdx.x (ballots + running offset; order is irrelevant to the counts
// but kept anyway): out (B, Q, 3), n_out (B)
__global__ __launch_bounds__(GEO_T) void k_ransac_compact(const float* __restrict__ pts, const unsigned char* __restrict__ eligible,
                                                           int Q, float* __restrict__ out, int32_t* __restrict__ n_out) {
    __shared__ int s_w[GEO_W];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    pts += (size_t)b * Q * 3; eligible += (size_t)b * Q; out += (size_t)b * Q * 3;
    int base = 0;
    for (int q0 = 0; q0 < Q; q0 += GEO_T) {
        const int q = q0 + tid;
        const bool ok = q < Q && eligible[q];
        const unsigned long long bal = __ballot(ok);
        __syncthreads();
        if (lane == 0) s_w[w] = __popcll(bal);
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int i = 0; i < GEO_W; ++i) { off += i < w ? s_w[i] : 0; tot += s_w[i]; }
        if (ok) {
            const int j = off + __popcll(bal & ((1ull << lane) - 1ull));
            out[j * 3] = pts[q * 3]; out[j * 3 + 1] = pts[q * 3 + 1]; out[j * 3 + 2] = pts[q * 3 + 2];
        }
        base += tot;
    }
    if (tid == 0) n_out[b] = base;
}

__global__ __launch_bounds__(GEO_T) void k_ransac_pick(const float* __restrict__ eqs,
                                                        const int32_t* __restrict__ counts, int T,
                                                        float* __restrict__ out_neg_eq, int32_t* __restrict__ out_best) {
    __shared__ int s_c[GEO_W], s_i[GEO_W];
    const int tid = threadIdx.x;
    eqs += (size_t)blockIdx.x * T * 4;
    counts += (size_t)blockIdx.x * T;
    out_neg_eq += blockIdx.x * 4;
    out_best += blockIdx.x * 2;
    int bc = -1, bi = 0x7fffffff;
    for (int t = tid; t < T; t += GEO_T) {
        const int c = counts[t];
        if (c > bc || (c == bc && t < bi)) { bc = c; bi = t; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int oc = __shfl_down(bc, off, 64), oi = __shfl_down(bi, off, 64);
        if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    if ((tid & 63) == 0) { s_c[tid >> 6] = bc; s_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < GEO_W; ++i)
            if (s_c[i] > bc || (s_c[i] == bc && s_i[i] < bi)) { bc = s_c[i]; bi = s_i[i]; }
        // torch.argmax = first maximal index; reference returns -equation (plane.py:134)
        for (int j = 0; j < 4; ++j) out_neg_eq[j] = -eqs[bi * 4 + j];
        out_best[0] = bi;
        out_best[1] = bc;
    }
}

extern "C" int cr_ransac_plane(cr_ctx* ctx, const float* pts, int64_t Q, const int32_t* triples, int64_t T,
                               float thresh, float* out_neg_eq, int32_t* out_counts, int32_t* out_best) {
    CR_CHECK_ARG(ctx != nullptr, "cr_ransac_plane: ctx is NULL");
    CR_CHECK_ARG(Q >= 3 && T >= 1, "cr_ransac_plane: need Q>=3 points and T>=1 triples");
    CR_CHECK_ARG(Q <= 0x7fffffff / 3 && T <= (int64_t)(ctx->ws_bytes / 16), "cr_ransac_plane: too large");
    CR_CHECK_ARG(pts && triples && out_neg_eq && out_counts && out_best, "cr_ransac_plane: NULL pointer");
    float* eqs = (float*)ctx->ws;
    hipLaunchKernelGGL(k_ransac_count, dim3((unsigned)cr_cdiv(T, RANSAC_TPB)), dim3(GEO_T), 0, ctx->stream, pts, (int)Q, triples,
                       (int)T, thresh, eqs, out_counts, (const unsigned char*)nullptr);
    CR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ransac_pick, dim3(1), dim3(GEO_T), 0, ctx->stream, eqs, out_counts, (int)T, out_neg_eq,
                       out_best);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_ransac_plane_batched(cr_ctx* ctx, const float* pts, const unsigned char* eligible, int B, int64_t Q,
                                       const int32_t* triples, int64_t T, float thresh, float* out_neg_eq,
                                       int32_t* out_counts, int32_t* out_best) {
    CR_CHECK_ARG(ctx != nullptr, "cr_ransac_plane_batched: ctx is NULL");
    CR_CHECK_ARG(B >= 0 && B <= 65535 && Q >= 3 && T >= 1, "cr_ransac_plane_batched: need B<=65535, Q>=3 points, T>=1 triples");
    if (B == 0) return CR_OK;
    CR_CHECK_ARG(Q <= 0x7fffffff / 3 && (int64_t)B * T <= (int64_t)(ctx->ws_bytes / 16), "cr_ransac_plane_batched: too large");
    CR_CHECK_ARG(pts && triples && out_neg_eq && out_counts && out_best, "cr_ransac_plane_batched: NULL pointer");
    float* eqs = (float*)ctx->ws;
    // workspace: plane equations (B, T, 4) | compacted points (B, Q, 3) | their counts (B)
    const size_t off_c = ((size_t)B * T * 16 + 255) & ~(size_t)255, need = off_c + (size_t)B * Q * 12 + (size_t)B * 4;
    if (eligible && need <= ctx->ws_bytes) {
        float* cp = (float*)((char*)ctx->ws + off_c);
        int32_t* cn = (int32_t*)(cp + (size_t)B * Q * 3);
        hipLaunchKernelGGL(k_ransac_compact, dim3((unsigned)B), dim3(GEO_T), 0, ctx->stream, pts, eligible, (int)Q, cp, cn);
        hipLaunchKernelGGL(k_ransac_count, dim3((unsigned)cr_cdiv(T, RANSAC_TPB), (unsigned)B), dim3(GEO_T), 0, ctx->stream, pts,
                           (int)Q, triples, (int)T, thresh, eqs, out_counts, eligible, (const float*)cp, (const int32_t*)cn);
    } else
    hipLaunchKernelGGL(k_ransac_count, dim3((unsigned)cr_cdiv(T, RANSAC_TPB), (unsigned)B), dim3(GEO_T), 0, ctx->stream, pts,
                       (int)Q, triples, (int)T, thresh, eqs, out_counts, eligible);
    CR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ransac_pick, dim3((unsigned)B), dim3(GEO_T), 0, ctx->stream, eqs, out_counts, (int)T, out_neg_eq,
                       out_best);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
