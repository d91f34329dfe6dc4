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
