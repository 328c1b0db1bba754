// Shared helpers for the gfx950 kernels of libctunet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "ctunet_hip.h"

#define CTU_ABI_VERSION 5

typedef float f32x4 __attribute__((ext_vector_type(4)));

void ctu_set_error(const char* fmt, ...);

#define CTU_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ctu_set_error(__VA_ARGS__);        \
            return CTU_EINVAL;                 \
        }                                      \
    } while (0)

#define CTU_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            ctu_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return CTU_ELAUNCH;                                                    \
        }                                                                          \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// lazy-BN input transform on 4 consecutive channels: 2 packed FMAs + 4 max (the clamp bound is -inf without ReLU,
// so there is no branch; __builtin_elementwise_max avoids the extra canonicalising max of fmaxf)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 xform4(float4 v, float4 sc, float4 sh, int relu) {
    const float lo = relu ? 0.f : -__builtin_inff();
    const f32x2 l2 = {lo, lo};
    const f32x2 a = __builtin_elementwise_max(__builtin_elementwise_fma(f32x2{v.x, v.y}, f32x2{sc.x, sc.y}, f32x2{sh.x, sh.y}), l2);
    const f32x2 b = __builtin_elementwise_max(__builtin_elementwise_fma(f32x2{v.z, v.w}, f32x2{sc.z, sc.w}, f32x2{sh.z, sh.w}), l2);
    return make_float4(a.x, a.y, b.x, b.y);
}

// ---- activation storage types of the reduced-precision path (ctu_lp_* entry points): 16-bit channels-last tensors,
// fp32 arithmetic.  ld4 / st4 move 4 consecutive channels as one 16-byte (fp32) or 8-byte (bf16 / fp16) access; rnd()
// rounds a value to the storage precision (the value a consumer will read back), RNE (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef f16_t f16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <class T> struct Vec;
template <> struct Vec<bf16_t> { typedef bf16x4 v4; typedef bf16x8 v8; };
template <> struct Vec<f16_t> { typedef f16x4 v4; typedef f16x8 v8; };

template <class T> __device__ __forceinline__ float4 ld4(const T* p) {
    const f32x4 f = __builtin_convertvector(*reinterpret_cast<const typename Vec<T>::v4*>(p), f32x4);
    return make_float4(f[0], f[1], f[2], f[3]);
}
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <class T> __device__ __forceinline__ void st4(T* p, float4 v) {
    *reinterpret_cast<typename Vec<T>::v4*>(p) = __builtin_convertvector(f32x4{v.x, v.y, v.z, v.w}, typename Vec<T>::v4);
}
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <class T> __device__ __forceinline__ float rnd(float v) { return (float)(T)v; }
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <class T> __device__ __forceinline__ float4 rnd4(float4 v) {
    const f32x4 f = __builtin_convertvector(__builtin_convertvector(f32x4{v.x, v.y, v.z, v.w}, typename Vec<T>::v4), f32x4);
    return make_float4(f[0], f[1], f[2], f[3]);
}
template <> __device__ __forceinline__ float4 rnd4<float>(float4 v) { return v; }

// dtype codes of the ctu_lp_* entry points (include/ctunet_hip.h)
#define CTU_DISPATCH_LP(dtype, ...)                                                  \
    do {                                                                             \
        if ((dtype) == CTU_BF16) { typedef bf16_t T; __VA_ARGS__; }                  \
        else if ((dtype) == CTU_F16) { typedef f16_t T; __VA_ARGS__; }               \
        else { ctu_set_error("unsupported dtype %d (1 = bf16, 2 = fp16)", (int)(dtype)); return CTU_EINVAL; } \
    } while (0)

// slab reductions: RPARTS thread groups of 64 stride over the partial slabs, then a fixed-order sum
constexpr int RPARTS = 16;
__device__ __forceinline__ float red_total(const float (*red)[64], int e) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < RPARTS; ++q) t += red[q][e];
    return t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
