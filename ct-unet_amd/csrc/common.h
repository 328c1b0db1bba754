// Shared helpers for the gfx950 kernels of libctunet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "ctunet_hip.h"

#define CTU_ABI_VERSION 7

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

// Wave priority by phase (MI355X_MICROARCH "Two waves per SIMD", item 2): two blocks share a CU, so a SIMD holds one wave
// streaming MFMAs and one staging the next box; at equal priority the OLDER wave wins every issue slot, and a younger
// stager advances one instruction per MFMA.  Staging phases raise their priority, MFMA loops run at 0.
#ifndef CTU_PRIO
#define CTU_PRIO 1
#endif
// the persistent forward / data-gradient kernels pin their fragment reads one tap group ahead of the MFMAs that use them
#ifndef CTU_PIN_FWD
#define CTU_PIN_FWD 1
#endif
// weight-gradient kernels: VALU instructions of the staging transform pinned under each MFMA of a K-step (0: left to the
// scheduler -- measured: 3 per MFMA made the lazy-BatchNorm step 3.37 ms against 3.29 unpinned)
#ifndef CTU_PIN_VALU
#define CTU_PIN_VALU 0
#endif
#ifndef CTU_XF_SCALAR
#define CTU_XF_SCALAR 0
#endif
#ifndef CTU_PRIO_STAGE
#define CTU_PRIO_STAGE 1
#endif
#if CTU_PRIO
#define CTU_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define CTU_SETPRIO(n) do { } while (0)
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// lazy-BN input transform on 4 consecutive channels: 2 packed FMAs + 4 max (the clamp bound is -inf without ReLU,
// so there is no branch; __builtin_elementwise_max avoids the extra canonicalising max of fmaxf)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 xform4(float4 v, float4 sc, float4 sh, int relu) {
    const float lo = relu ? 0.f : -__builtin_inff();
#if CTU_XF_SCALAR    /* dev A/B: four v_fma_f32 instead of two v_pk_fma_f32 (needs -fno-slp-vectorize to stay scalar) */
    return make_float4(__builtin_fmaxf(__builtin_fmaf(v.x, sc.x, sh.x), lo), __builtin_fmaxf(__builtin_fmaf(v.y, sc.y, sh.y), lo),
                       __builtin_fmaxf(__builtin_fmaf(v.z, sc.z, sh.z), lo), __builtin_fmaxf(__builtin_fmaf(v.w, sc.w, sh.w), lo));
#endif
    const f32x2 l2 = {lo, lo};
    const f32x2 a = __builtin_elementwise_max(__builtin_elementwise_fma(f32x2{v.x, v.y}, f32x2{sc.x, sc.y}, f32x2{sh.x, sh.y}), l2);
    const f32x2 b = __builtin_elementwise_max(__builtin_elementwise_fma(f32x2{v.z, v.w}, f32x2{sc.z, sc.w}, f32x2{sh.z, sh.w}), l2);
    return make_float4(a.x, a.y, b.x, b.y);
}

// lazy BatchNorm+ReLU backward on 4 consecutive channels: the gradient w.r.t. the raw conv output y from the gradient g
// w.r.t. the activated output,  gy = [y sc + sh > 0] k0 g + (A y + B)  with bn_bwd_finalize_kernel's per-channel
// k0 = gamma invstd, A = -k0 k2 invstd, B = -k0 (k1 - k2 mean invstd)  (k1 = dbeta / n, k2 = dgamma / n) -- the same
// value as bn_relu_bwd_apply_kernel's  k0 (gz - k1 - (y - mean) invstd k2)
__device__ __forceinline__ float4 bn_bwd_lazy4(float4 g, float4 y, float4 sc, float4 sh, float4 k0, float4 A, float4 B) {
    float4 r;
    r.x = fmaf(k0.x, (fmaf(y.x, sc.x, sh.x) > 0.f) ? g.x : 0.f, fmaf(A.x, y.x, B.x));
    r.y = fmaf(k0.y, (fmaf(y.y, sc.y, sh.y) > 0.f) ? g.y : 0.f, fmaf(A.y, y.y, B.y));
    r.z = fmaf(k0.z, (fmaf(y.z, sc.z, sh.z) > 0.f) ? g.z : 0.f, fmaf(A.z, y.z, B.z));
    r.w = fmaf(k0.w, (fmaf(y.w, sc.w, sh.w) > 0.f) ? g.w : 0.f, fmaf(A.w, y.w, B.w));
    return r;
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
