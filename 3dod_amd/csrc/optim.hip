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

// torch.optim.SGD(momentum, dampening 0, no nesterov): g' = g*gscale + wd*p ; m = mom*m + g' ; p -= lr*m
// lr = lr_base * (*lr_scale): the schedule's factor is read on the device, so a captured launch (HIP graph) follows the
// warm-up / multi-step schedule without being re-captured
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             int64_t n, float lr_base, const float* __restrict__ lr_scale, float mom, float wd,
                                             float gscale, const int* __restrict__ skip) {
    if (skip && *skip) return;
    const float lr = lr_scale ? lr_base * *lr_scale : lr_base;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            float4 pv = *reinterpret_cast<float4*>(p + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            float4 mv = *reinterpret_cast<float4*>(m + i);
            mv.x = mom * mv.x + (gv.x * gscale + wd * pv.x); pv.x -= lr * mv.x;
            mv.y = mom * mv.y + (gv.y * gscale + wd * pv.y); pv.y -= lr * mv.y;
            mv.z = mom * mv.z + (gv.z * gscale + wd * pv.z); pv.z -= lr * mv.z;
            mv.w = mom * mv.w + (gv.w * gscale + wd * pv.w); pv.w -= lr * mv.w;
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
        } else {
            for (int64_t j = i; j < n; ++j) {
                const float mm = mom * m[j] + (g[j] * gscale + wd * p[j]);
                m[j] = mm;
                p[j] -= lr * mm;
            }
        }
    }
}

extern "C" int cr_sgd_step(cr_ctx* ctx, float* p, const float* g, float* m, int64_t n, float lr, const float* lr_scale_dev,
                           float momentum, float weight_decay, float grad_scale, const int* skip_flag) {
    CR_CHECK_ARG(ctx && n >= 0, "cr_sgd_step: bad args");
    if (n == 0) return CR_OK;
    CR_CHECK_ARG(p && g && m, "cr_sgd_step: NULL pointer");
    CR_CHECK_ARG(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m)) & 15) == 0, "cr_sgd_step: misaligned buffers");
    int64_t nb = cr_cdiv(cr_cdiv(n, 4), 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_sgd, dim3((unsigned)nb), dim3(256), 0, ctx->stream, p, g, m, n, lr, lr_scale_dev, momentum,
                       weight_decay, grad_scale, skip_flag);
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
