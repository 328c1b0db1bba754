// Inference tail and sample-schema helpers for gfx950: hard segmentation, hard-Dice counts,
// one-hot targets.  Index / counting work on NCDHW maps: HBM-bound streaming, results are exact integers.
//
// Replaces: hard_segm_from_tensor (torch.argmax over classes, as float)         ctunet/utilities.py:103-124
//           dice_coeff (monai compute_meandice on one_hot(argmax(pred)))         ctunet/utilities.py:53-59
//           one_hot(label.long(), C).movedim(-1, 1).float() of the datasets      ctunet/pytorch/datasets.py:107-110,212-217
#include "common.h"

namespace {

constexpr int SB = 256;
constexpr int SEG_MAXC = 8;
constexpr int DICE_BX = 256;     // blocks per batch item

// first maximum wins (torch.argmax); NaN never wins against a number, like a strict > scan
__device__ __forceinline__ int argmax_c(const float* p, int C, int64_t V, int64_t v) {
    float best = p[v];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
        const float x = p[(int64_t)c * V + v];
        if (x > best) { best = x; bi = c; }
    }
    return bi;
}

__global__ void hard_segm_kernel(const float* __restrict__ prob, int C, int64_t V, int N, float* __restrict__ seg) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * V) return;
    const int64_t n = idx / V, v = idx % V;
    seg[idx] = (float)argmax_c(prob + n * C * V, C, V, v);
}

__global__ void one_hot_kernel(const float* __restrict__ label, int C, int64_t V, int N, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * V) return;
    const int64_t n = idx / V, v = idx % V;
    const long long k = (long long)label[idx];               // .long(): truncation toward zero
    for (int c = 0; c < C; ++c) out[(n * C + c) * V + v] = (k == c) ? 1.f : 0.f;
}

// per (n, c): [ sum_v hard_c * target_c , sum_v hard_c , sum_v target_c ] in double (exact for 0/1 targets)
__global__ __launch_bounds__(SB) void hard_dice_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                               int C, int64_t V, double* __restrict__ partials) {
    const int n = blockIdx.y;
    const float* pp = pred + (int64_t)n * C * V;
    const float* tp = target + (int64_t)n * C * V;
    double acc[SEG_MAXC][3];
#pragma unroll
    for (int c = 0; c < SEG_MAXC; ++c) { acc[c][0] = 0.0; acc[c][1] = 0.0; acc[c][2] = 0.0; }
    for (int64_t v = (int64_t)blockIdx.x * SB + threadIdx.x; v < V; v += (int64_t)gridDim.x * SB) {
        const int k = argmax_c(pp, C, V, v);
#pragma unroll
        for (int c = 0; c < SEG_MAXC; ++c)
            if (c < C) {
                const double t = (double)tp[(int64_t)c * V + v];
                acc[c][2] += t;
                if (k == c) { acc[c][0] += t; acc[c][1] += 1.0; }
            }
    }
    __shared__ double red[SB / 64][SEG_MAXC * 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < SEG_MAXC; ++c)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = acc[c][j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) red[wave][c * 3 + j] = s;
        }
    __syncthreads();
    if (threadIdx.x < SEG_MAXC * 3) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SB / 64; ++w) s += red[w][threadIdx.x];
        partials[((size_t)n * gridDim.x + blockIdx.x) * (SEG_MAXC * 3) + threadIdx.x] = s;
    }
}

__global__ void hard_dice_final_kernel(const double* __restrict__ partials, int nb, int C, double* __restrict__ counts) {
    const int n = blockIdx.x, i = threadIdx.x;                 // i = c * 3 + j
    if (i >= C * 3) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += partials[((size_t)n * nb + b) * (SEG_MAXC * 3) + i];     // fixed order
    counts[(size_t)n * C * 3 + i] = s;
}

}  // namespace

extern "C" int ctu_hard_segm(const float* prob, int N, int C, int64_t nvox_per_item, float* seg, void* stream) {
    CTU_REQUIRE(prob && seg, "hard_segm: null pointer");
    CTU_REQUIRE(N > 0 && C >= 1 && nvox_per_item > 0, "hard_segm: bad shape N=%d C=%d", N, C);
    const int64_t total = (int64_t)N * nvox_per_item;
    hard_segm_kernel<<<(unsigned)ceil_div64(total, SB), SB, 0, (hipStream_t)stream>>>(prob, C, nvox_per_item, N, seg);
    CTU_CHECK_LAUNCH("hard_segm");
    return CTU_OK;
}

extern "C" int ctu_one_hot(const float* label, int N, int C, int64_t nvox_per_item, float* out, void* stream) {
    CTU_REQUIRE(label && out, "one_hot: null pointer");
    CTU_REQUIRE(N > 0 && C >= 1 && nvox_per_item > 0, "one_hot: bad shape N=%d C=%d", N, C);
    const int64_t total = (int64_t)N * nvox_per_item;
    one_hot_kernel<<<(unsigned)ceil_div64(total, SB), SB, 0, (hipStream_t)stream>>>(label, C, nvox_per_item, N, out);
    CTU_CHECK_LAUNCH("one_hot");
    return CTU_OK;
}

extern "C" size_t ctu_hard_dice_ws_doubles(int N) { return (size_t)N * DICE_BX * SEG_MAXC * 3; }

extern "C" int ctu_hard_dice_counts(const float* pred, const float* target, int N, int C, int64_t nvox_per_item,
                                    double* counts, double* ws, void* stream) {
    CTU_REQUIRE(pred && target && counts && ws, "hard_dice_counts: null pointer");
    CTU_REQUIRE(N > 0 && C >= 1 && C <= SEG_MAXC && nvox_per_item > 0, "hard_dice_counts: N=%d C=%d (C <= %d)", N, C, SEG_MAXC);
    hipStream_t st = (hipStream_t)stream;
    hard_dice_partial_kernel<<<dim3(DICE_BX, N), SB, 0, st>>>(pred, target, C, nvox_per_item, ws);
    CTU_CHECK_LAUNCH("hard_dice_partial");
    hard_dice_final_kernel<<<N, 64, 0, st>>>(ws, DICE_BX, C, counts);
    CTU_CHECK_LAUNCH("hard_dice_final");
    return CTU_OK;
}
