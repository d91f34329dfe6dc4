// Storage-type helpers shared by the activation kernels.  Activations live in HBM either as bf16 (u16 bit patterns;
// the fast mode) or as f32 (the reference's precision: tools/train_net.py of the reference trains without autocast).
// Kernels are templates on the storage type T in {u16, float}; arithmetic is always f32.
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned short u16;

__device__ __forceinline__ float bf2f(u16 b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ u16 f2bf(float f) {
    __bf16 b = (__bf16)f;                      // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(u16, b);
}

__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
    f[0] = bf2f((u16)(v.x & 0xffff)); f[1] = bf2f((u16)(v.x >> 16));
    f[2] = bf2f((u16)(v.y & 0xffff)); f[3] = bf2f((u16)(v.y >> 16));
    f[4] = bf2f((u16)(v.z & 0xffff)); f[5] = bf2f((u16)(v.z >> 16));
    f[6] = bf2f((u16)(v.w & 0xffff)); f[7] = bf2f((u16)(v.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 o;
    o.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
    o.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
    o.z = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
    o.w = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
    return o;
}

// 8 consecutive channels at element index `e` (a multiple of 8): one 16-B access for bf16, two for f32
template <typename T> __device__ __forceinline__ void load8(const T* __restrict__ p, size_t e, float* f);
template <> __device__ __forceinline__ void load8<u16>(const u16* __restrict__ p, size_t e, float* f) {
    unpack8(*reinterpret_cast<const uint4*>(p + e), f);
}
template <> __device__ __forceinline__ void load8<float>(const float* __restrict__ p, size_t e, float* f) {
    const float4 a = *reinterpret_cast<const float4*>(p + e), b = *reinterpret_cast<const float4*>(p + e + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename T> __device__ __forceinline__ void store8(T* __restrict__ p, size_t e, const float* f);
template <> __device__ __forceinline__ void store8<u16>(u16* __restrict__ p, size_t e, const float* f) {
    *reinterpret_cast<uint4*>(p + e) = pack8(f);
}
template <> __device__ __forceinline__ void store8<float>(float* __restrict__ p, size_t e, const float* f) {
    *reinterpret_cast<float4*>(p + e) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(p + e + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
template <typename T> __device__ __forceinline__ float load1(const T* __restrict__ p, size_t e);
template <> __device__ __forceinline__ float load1<u16>(const u16* __restrict__ p, size_t e) { return bf2f(p[e]); }
template <> __device__ __forceinline__ float load1<float>(const float* __restrict__ p, size_t e) { return p[e]; }
