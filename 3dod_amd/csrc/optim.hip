// Optimizer step of tools/train_net.py:233-266 + cubercnn/solver/build.py:50-56 fused on device:
// a non-finite scan over the (already all-reduced) flat gradient and an SGD-momentum update that
// is skipped on-device when the flag is set -- no host round trip in the step.
#include "cr_common.h"

__global__ __launch_bounds__(256) void k_nonfinite(const float* __restrict__ g, int64_t n, int* __restrict__ flag) {
    int bad = 0;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
        } else {
            for (int64_t j = i; j < n; ++j) bad |= !isfinite(g[j]);
        }
    }
    if (__any(bad)) { if ((threadIdx.x & 63) == 0) atomicOr(flag, 1); }
}

// flag is OR-ed (caller zeroes it once per step)
extern "C" int cr_nonfinite_flag(cr_ctx* ctx, const float* g, int64_t n, int* flag) {
    CR_CHECK_ARG(ctx && flag && n >= 0, "cr_nonfinite_flag: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(g && (((uintptr_t)g) & 15) == 0, "cr_nonfinite_flag: NULL or misaligned gradient");
    int64_t nb = cr_cdiv(cr_cdiv(n, 4), 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_nonfinite, dim3((unsigned)nb), dim3(256), 0, ctx->stream, g, n, flag);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// torch.optim.SGD(momentum, dampening 0): g' = g*gscale + wd*p ; m = mom*m + g' ; p -= lr*m   (nesterov: p -= lr*(g' + mom*m))
// lr = lr_base * (*lr_scale): the schedule's factor is read on the device, so a captured launch (HIP graph) follows the
// warm-up / multi-step schedule without being re-captured
template <bool NESTEROV>
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             int64_t n, float lr_base, const float* __restrict__ lr_scale, float mom, float wd,
                                             float gscale, const int* __restrict__ skip) {
    if (skip && *skip) return;
    const float lr = lr_scale ? lr_base * *lr_scale : lr_base;
    auto upd = [&](float& pv, float gv, float& mv) {
        const float gg = gv * gscale + wd * pv;
        mv = mom * mv + gg;
        pv -= lr * (NESTEROV ? gg + mom * mv : mv);
    };
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            float4 pv = *reinterpret_cast<float4*>(p + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            float4 mv = *reinterpret_cast<float4*>(m + i);
            upd(pv.x, gv.x, mv.x); upd(pv.y, gv.y, mv.y); upd(pv.z, gv.z, mv.z); upd(pv.w, gv.w, mv.w);
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
        } else {
            for (int64_t j = i; j < n; ++j) {
                float pv = p[j], mv = m[j];
                upd(pv, g[j], mv);
                m[j] = mv;
                p[j] = pv;
            }
        }
    }
}

static int sgd_launch(cr_ctx* ctx, float* p, const float* g, float* m, int64_t n, float lr, const float* lr_scale_dev,
                      float momentum, float weight_decay, float grad_scale, const int* skip_flag, bool nesterov, const char* who) {
    CR_CHECK_ARG(ctx && n >= 0, "%s: bad args", who);
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(p && g && m, "%s: NULL pointer", who);
    CR_CHECK_ARG(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m)) & 15) == 0, "%s: misaligned buffers", who);
    int64_t nb = cr_cdiv(cr_cdiv(n, 4), 256);
    if (nb > 4096) nb = 4096;
    if (nesterov)
        hipLaunchKernelGGL(k_sgd<true>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, p, g, m, n, lr, lr_scale_dev, momentum,
                           weight_decay, grad_scale, skip_flag);
    else
        hipLaunchKernelGGL(k_sgd<false>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, p, g, m, n, lr, lr_scale_dev, momentum,
                           weight_decay, grad_scale, skip_flag);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_sgd_step(cr_ctx* ctx, float* p, const float* g, float* m, int64_t n, float lr, const float* lr_scale_dev,
                           float momentum, float weight_decay, float grad_scale, const int* skip_flag) {
    return sgd_launch(ctx, p, g, m, n, lr, lr_scale_dev, momentum, weight_decay, grad_scale, skip_flag, false, "cr_sgd_step");
}

extern "C" int cr_sgd_step_nesterov(cr_ctx* ctx, float* p, const float* g, float* m, int64_t n, float lr, const float* lr_scale_dev,
                                    float momentum, float weight_decay, float grad_scale, const int* skip_flag) {
    return sgd_launch(ctx, p, g, m, n, lr, lr_scale_dev, momentum, weight_decay, grad_scale, skip_flag, true, "cr_sgd_step_nesterov");
}

// SOLVER.CLIP_GRADIENTS (detectron2 maybe_add_gradient_clipping, called from cubercnn/solver/build.py:68), applied to the flat
// gradient in place before the update.  `gscale` (1 / world size after the all-reduce) is folded in, so the update that follows
// runs with grad_scale 1.
//   value:  g = clamp(g * gscale, -clip, clip)                                   (torch.nn.utils.clip_grad_value_)
//   norm:   per PARAMETER (detectron2 clips each parameter on its own): norm = || g * gscale ||_type,
//           g = g * gscale * min(1, max_norm / (norm + 1e-6))                    (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void k_clip_value(float* __restrict__ g, int64_t n, float clip, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        g[i] = fminf(fmaxf(g[i] * gscale, -clip), clip);
}

#define CLIP_SPLIT 16
// partial[param][split]: sum |g|^type over the split's slice (type 2: squares; other finite types: powf) or its maximum (inf)
__global__ __launch_bounds__(256) void k_clip_norm_partial(const float* __restrict__ g, const int64_t* __restrict__ starts,
                                                           const int64_t* __restrict__ counts, float type, float gscale,
                                                           float* __restrict__ partial) {
    __shared__ float s[4];
    const int prm = blockIdx.x, sp = blockIdx.y;
    const int64_t n = counts[prm], per = (n + CLIP_SPLIT - 1) / CLIP_SPLIT;
    const int64_t a = starts[prm] + sp * per, b = starts[prm] + min(n, (sp + 1) * per);
    const bool inf = isinf(type);
    float acc = 0.f;
    for (int64_t i = a + threadIdx.x; i < b; i += 256) {
        const float v = fabsf(g[i] * gscale);
        acc = inf ? fmaxf(acc, v) : acc + (type == 2.f ? v * v : (type == 1.f ? v : powf(v, type)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_down(acc, off, 64);
        acc = inf ? fmaxf(acc, o) : acc + o;
    }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[prm * CLIP_SPLIT + sp] = inf ? fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])) : ((s[0] + s[1]) + (s[2] + s[3]));
}

__global__ __launch_bounds__(256) void k_clip_norm_apply(float* __restrict__ g, const int64_t* __restrict__ starts,
                                                         const int64_t* __restrict__ counts, float type, float max_norm,
                                                         float gscale, const float* __restrict__ partial) {
    const int prm = blockIdx.x, sp = blockIdx.y;
    const bool inf = isinf(type);
    float tot = 0.f;
    for (int k = 0; k < CLIP_SPLIT; ++k) {                      // fixed order: every block of a parameter gets the same bits
        const float v = partial[prm * CLIP_SPLIT + k];
        tot = inf ? fmaxf(tot, v) : tot + v;
    }
    const float norm = inf ? tot : (type == 2.f ? sqrtf(tot) : (type == 1.f ? tot : powf(tot, 1.f / type)));
    const float coef = fminf(max_norm / (norm + 1e-6f), 1.f) * gscale;
    const int64_t n = counts[prm], per = (n + CLIP_SPLIT - 1) / CLIP_SPLIT;
    const int64_t a = starts[prm] + sp * per, b = starts[prm] + min(n, (sp + 1) * per);
    for (int64_t i = a + threadIdx.x; i < b; i += 256) g[i] *= coef;
}

extern "C" int cr_grad_clip_value(cr_ctx* ctx, float* g, int64_t n, float clip_value, float grad_scale) {
    CR_CHECK_ARG(ctx && n >= 0 && clip_value >= 0.f, "cr_grad_clip_value: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(g != nullptr, "cr_grad_clip_value: NULL gradient");
    int64_t nb = cr_cdiv(n, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_clip_value, dim3((unsigned)nb), dim3(256), 0, ctx->stream, g, n, clip_value, grad_scale);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// starts / counts (nparam) int64 on the device: offset and element count of every parameter inside g; partial: nparam * 16 floats
extern "C" int cr_grad_clip_norm(cr_ctx* ctx, float* g, const int64_t* starts, const int64_t* counts, int nparam, float max_norm,
                                 float norm_type, float grad_scale, float* partial) {
    CR_CHECK_ARG(ctx && nparam >= 0 && max_norm >= 0.f && norm_type > 0.f, "cr_grad_clip_norm: bad args");
    if (nparam == 0) return CR_OK;
    CR_CHECK_ARG(g && starts && counts && partial, "cr_grad_clip_norm: NULL pointer");
    CR_CHECK_ARG(nparam <= 65535 * 16, "cr_grad_clip_norm: too many parameters");
    const dim3 grid((unsigned)nparam, CLIP_SPLIT);
    hipLaunchKernelGGL(k_clip_norm_partial, grid, dim3(256), 0, ctx->stream, g, starts, counts, norm_type, grad_scale, partial);
    hipLaunchKernelGGL(k_clip_norm_apply, grid, dim3(256), 0, ctx->stream, g, starts, counts, norm_type, max_norm, grad_scale, partial);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// torch.optim.Adam / AdamW (cubercnn/solver/build.py:57-64 selects them with eps = 1e-2, optionally amsgrad), one launch per
// hyper-parameter segment like k_sgd:
//   Adam   g' = g * gscale + wd * p                       AdamW   p *= 1 - lr * wd ;  g' = g * gscale
//   m = b1 m + (1 - b1) g' ;  v = b2 v + (1 - b2) g'^2 ;  amsgrad: vmax = max(vmax, v), used in place of v
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// t is the number of updates actually applied: it lives on the device (`step`, advanced by k_adam_tick unless the step is
// skipped), so a skipped step leaves the optimizer state untouched exactly like the host-side `continue` of train_net.py:246.
__global__ void k_adam_tick(float* __restrict__ step, const int* __restrict__ skip) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && !(skip && *skip)) *step += 1.f;
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, float* __restrict__ vmax, int64_t n, float lr_base,
                                              const float* __restrict__ lr_scale, float b1, float b2, float eps, float wd,
                                              float gscale, int decoupled, const float* __restrict__ step,
                                              const int* __restrict__ skip) {
    if (skip && *skip) return;
    const float lr = lr_scale ? lr_base * *lr_scale : lr_base;
    const float t = *step;
    const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pv = p[i];
        float gv = g[i] * gscale;
        if (decoupled) pv *= 1.f - lr * wd; else gv += wd * pv;
        const float mv = b1 * m[i] + (1.f - b1) * gv;
        float vv = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        if (vmax) { vv = fmaxf(vmax[i], vv); vmax[i] = vv; }
        p[i] = pv - step_size * (mv / (sqrtf(vv) / bc2s + eps));
    }
}

// step: device float, the number of applied updates (advance it once per optimizer step with cr_adam_tick BEFORE the segments)
extern "C" int cr_adam_tick(cr_ctx* ctx, float* step, const int* skip_flag) {
    CR_CHECK_ARG(ctx && step, "cr_adam_tick: NULL pointer");
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(64), 0, ctx->stream, step, skip_flag);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

extern "C" int cr_adam_step(cr_ctx* ctx, float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                            int64_t n, float lr, const float* lr_scale_dev, float beta1, float beta2, float eps,
                            float weight_decay, float grad_scale, int decoupled, const float* step, const int* skip_flag) {
    CR_CHECK_ARG(ctx && n >= 0, "cr_adam_step: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(p && g && exp_avg && exp_avg_sq && step, "cr_adam_step: NULL pointer");
    int64_t nb = cr_cdiv(n, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)nb), dim3(256), 0, ctx->stream, p, g, exp_avg, exp_avg_sq, max_exp_avg_sq, n, lr,
                       lr_scale_dev, beta1, beta2, eps, weight_decay, grad_scale, decoupled, step, skip_flag);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// Multi-segment copy / accumulate: segment d moves n floats from src to dst (mode 0: dst = src, 1: dst += src); one thread
// per float, the segment found by binary search over the running item count.  Used to stack the predictor weights of a
// head into one GEMM operand (and to route the stacked gradient back into the flat gradient) with ONE launch instead of
// one cat / add per parameter (cube_head.py:113-149 of the reference: five nn.Linear predictors on the same input).
struct cr_segdesc { const float* src; float* dst; int64_t n, item0; };
__global__ __launch_bounds__(256) void k_multi_seg(const cr_segdesc* __restrict__ descs, int ndesc, int64_t total, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].item0 <= i) lo = mid; else hi = mid - 1;
    }
    const cr_segdesc d = descs[lo];
    const int64_t j = i - d.item0;
    if (mode) d.dst[j] += d.src[j]; else d.dst[j] = d.src[j];
}

extern "C" int cr_multi_seg(cr_ctx* ctx, const void* descs_dev, int ndesc, int64_t total, int accumulate) {
    CR_CHECK_ARG(ctx && ndesc >= 0 && total >= 0, "cr_multi_seg: bad args");
    if (ndesc == 0 || total == 0) return CR_OK;
    CR_CHECK_ARG(descs_dev, "cr_multi_seg: NULL descriptor table");
    hipLaunchKernelGGL(k_multi_seg, dim3((unsigned)cr_cdiv(total, 256)), dim3(256), 0, ctx->stream,
                       (const cr_segdesc*)descs_dev, ndesc, total, accumulate ? 1 : 0);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// Loss-divergence guard of tools/train_net.py:202-220 on the device, one thread: vals (n) are this step's loss terms summed
// over the ranks (scale = 1 / world size).  red[i] = vals[i] * scale, total = sum red; the step "diverges" when total is not
// finite or exceeds tolerance x the rolling mean (first step: the mean starts at 2 x total); the mean is updated only on good
// steps (gamma).  flag = diverging (stabilize) or 0; cr_nonfinite_flag ORs the gradient scan into it afterwards.
__global__ void k_loss_guard(const float* __restrict__ vals, int n, float scale, float* __restrict__ red, float* __restrict__ total,
                             float* __restrict__ recent, int stabilize, float tol, float gamma, int* __restrict__ flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float t = 0.f;
    for (int i = 0; i < n; ++i) {
        const float v = vals[i] * scale;
        if (red) red[i] = v;
        t += v;
    }
    *total = t;
    const float r0 = *recent;
    const float rec = (r0 != r0) ? t * 2.0f : r0;
    bool div = (t > rec * tol) || !isfinite(t);
    if (!stabilize) div = false;
    *recent = div ? rec : rec * (1.f - gamma) + t * gamma;
    *flag = div ? 1 : 0;
}

extern "C" int cr_loss_guard(cr_ctx* ctx, const float* vals, int n, float scale, float* red, float* total, float* recent,
                             int stabilize, float tolerance, float gamma, int* flag) {
    CR_CHECK_ARG(ctx && vals && total && recent && flag && n >= 0, "cr_loss_guard: bad args");
    hipLaunchKernelGGL(k_loss_guard, dim3(1), dim3(64), 0, ctx->stream, vals, n, scale, red, total, recent, stabilize, tolerance,
                       gamma, flag);
    CR_LAUNCH_CHECK();
    return CR_OK;
}

// iterations_explode += flag != 0, iterations_success += flag == 0 (train_net.py:259-266), after the update
__global__ void k_step_counters(const int* __restrict__ flag, float* __restrict__ explode, float* __restrict__ success) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (*flag) *explode += 1.f; else *success += 1.f;
}

extern "C" int cr_step_counters(cr_ctx* ctx, const int* flag, float* explode, float* success) {
    CR_CHECK_ARG(ctx && flag && explode && success, "cr_step_counters: NULL pointer");
    hipLaunchKernelGGL(k_step_counters, dim3(1), dim3(64), 0, ctx->stream, flag, explode, success);
    CR_LAUNCH_CHECK();
    return CR_OK;
}
