// Shared helpers for the gfx950 kernels of libctunet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "ctunet_hip.h"

#define CTU_ABI_VERSION 3

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
