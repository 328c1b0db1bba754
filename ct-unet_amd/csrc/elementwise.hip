// HBM-bound kernels of the U-Net path for gfx950: layout conversion, BatchNorm3d statistics
// finalisation and backward, MaxPool3d(2,2) forward/backward, per-channel sums, fused Adam.
// All activation accesses are 16-byte (float4) channels-last vectors; reductions are
// two-stage (per-block partials, then a small finalize kernel) -- no float atomics, so
// results are bitwise reproducible run to run.
#include "common.h"
#include "bn_tail.h"

static thread_local char g_err[512] = "";

void ctu_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ctu_last_error(void) { return g_err; }
extern "C" int ctu_abi_version(void) { return CTU_ABI_VERSION; }
extern "C" const char* ctu_arch(void) { return "gfx950"; }

namespace {

constexpr int EW_BLOCK = 256;
constexpr int FIN_BLOCK = 256;    // threads of the per-channel statistics finalisation
constexpr int MAX_RED_BLOCKS = 1024;

// ------------------------------------------------------------------ layout
// NCDHW -> NDHWC: one thread per (voxel, channel-quad); reads are coalesced along w per channel.
template <class T>
__global__ void ncdhw_to_ndhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int64_t V,
                                      int64_t total_vox, int cp, int cs) {
    const int nq = cp >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_vox * nq) return;
    // voxel fastest so that each channel plane is read contiguously by consecutive lanes
    const int64_t vox = idx % total_vox;
    const int qd = (int)(idx / total_vox);
    const int64_t n = vox / V, v = vox % V;
    float4 o;
    float* op = &o.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = qd * 4 + j;
        op[j] = (c < C) ? src[(n * C + c) * V + v] : 0.f;
    }
    st4<T>(dst + vox * cs + qd * 4, o);
}

template <class T>
__global__ void ndhwc_to_ncdhw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C, int64_t V,
                                      int64_t total_vox, int cs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_vox * C) return;
    const int64_t vox = idx % total_vox;
    const int c = (int)(idx / total_vox);
    const int64_t n = vox / V, v = vox % V;
    dst[(n * C + c) * V + v] = (float)src[vox * cs + c];
}

// Sum of (a, b) over the NT threads of a block, in every thread: wave butterflies, one LDS exchange, a fixed-order sum
// over the waves (deterministic; one barrier instead of a log2(NT)-step LDS tree -- these kernels are all latency).
template <int NT>
__device__ __forceinline__ void block_sum2(double& a, double& b) {
    __shared__ double sm[2 * (NT / 64)];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { sm[(threadIdx.x >> 6) * 2] = a; sm[(threadIdx.x >> 6) * 2 + 1] = b; }
    __syncthreads();
    a = 0.0; b = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { a += sm[w * 2]; b += sm[w * 2 + 1]; }
}

// ------------------------------------------------------------------ BN finalize
__global__ void bn_finalize_kernel(const float* __restrict__ stats, int nblocks, int C, int cp, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                   int n_updates, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   long long* __restrict__ nbt) {
    const int c = blockIdx.x;
    if (nbt && c == 0 && threadIdx.x == 0 && n_updates > 0) nbt[0] += n_updates;     // num_batches_tracked
    // the channel's parameters are fetched BEFORE the row reduction (this kernel is one dependent chain of memory round
    // trips on the step's critical path: rows -> block sum -> parameters -> stores; the parameter trip now overlaps the rows)
    float g_ = 0.f, b_ = 0.f, rm0 = 0.f, rv0 = 0.f;
    if (threadIdx.x == 0 && c < C) {
        g_ = gamma[c]; b_ = beta[c];
        if (rmean && n_updates > 0) { rm0 = rmean[c]; rv0 = rvar[c]; }
    }
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int b = threadIdx.x; b < nblocks; b += FIN_BLOCK) {
            s1 += (double)stats[(size_t)b * 2 * cp + c];
            s2 += (double)stats[(size_t)b * 2 * cp + cp + c];
        }
    }
    block_sum2<FIN_BLOCK>(s1, s2);
    if (threadIdx.x == 0) {
        if (c < C) {
            const double mean = s1 / count;
            double var = s2 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = g_ * invstd;
            scale[c] = sc;
            shift[c] = b_ - (float)mean * sc;
            mean_out[c] = (float)mean;
            invstd_out[c] = invstd;
            if (rmean && n_updates > 0) {
                const float unb = (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
                float rm = rm0, rv = rv0;
                for (int u = 0; u < n_updates; ++u) {
                    rm = (1.f - momentum) * rm + momentum * (float)mean;
                    rv = (1.f - momentum) * rv + momentum * unb;
                }
                rmean[c] = rm; rvar[c] = rv;
            }
        } else {
            scale[c] = 0.f; shift[c] = 0.f; mean_out[c] = 0.f; invstd_out[c] = 0.f;
        }
    }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                      float eps, int C, int cp, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cp) return;
    if (c < C) {
        const float sc = gamma[c] / sqrtf(rvar[c] + eps);
        scale[c] = sc;
        shift[c] = beta[c] - rmean[c] * sc;
    } else {
        scale[c] = 0.f; shift[c] = 0.f;
    }
}

// ------------------------------------------------------------------ BN+ReLU backward
// thread = (voxel stripe, channel quad); channel quad fixed per thread so its constants sit in registers.
template <class T>
__global__ void bn_relu_bwd_reduce_kernel(const T* __restrict__ y, int y_cs, const T* __restrict__ ga,
                                          int g_cs, int cp, const float* __restrict__ scale,
                                          const float* __restrict__ shift, const float* __restrict__ mean,
                                          const float* __restrict__ invstd, int64_t nvox,
                                          float* __restrict__ partials, ctu_bn_bwd_tail tail) {
    const int nq = cp >> 2;
    const int tpv = EW_BLOCK / nq;                  // voxels per block pass (nq divides 256 or not: extra threads idle)
    const int qd = threadIdx.x % nq, vl = threadIdx.x / nq;
    __shared__ float red[EW_BLOCK * 8];
    float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
    if (vl < tpv) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + qd * 4);
        const float4 sh = *reinterpret_cast<const float4*>(shift + qd * 4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + qd * 4);
        const float4 is = *reinterpret_cast<const float4*>(invstd + qd * 4);
        for (int64_t v = (int64_t)blockIdx.x * tpv + vl; v < nvox; v += (int64_t)gridDim.x * tpv) {
            const float4 yy = ld4<T>(y + v * y_cs + qd * 4);
            const float4 gg = ld4<T>(ga + v * g_cs + qd * 4);
            float gz;
            gz = (fmaf(yy.x, sc.x, sh.x) > 0.f) ? gg.x : 0.f; a1.x += gz; a2.x += gz * (yy.x - mu.x) * is.x;
            gz = (fmaf(yy.y, sc.y, sh.y) > 0.f) ? gg.y : 0.f; a1.y += gz; a2.y += gz * (yy.y - mu.y) * is.y;
            gz = (fmaf(yy.z, sc.z, sh.z) > 0.f) ? gg.z : 0.f; a1.z += gz; a2.z += gz * (yy.z - mu.z) * is.z;
            gz = (fmaf(yy.w, sc.w, sh.w) > 0.f) ? gg.w : 0.f; a1.w += gz; a2.w += gz * (yy.w - mu.w) * is.w;
        }
    }
    float* r = &red[threadIdx.x * 8];
    r[0] = a1.x; r[1] = a1.y; r[2] = a1.z; r[3] = a1.w; r[4] = a2.x; r[5] = a2.y; r[6] = a2.z; r[7] = a2.w;
    __syncthreads();
    // thread t < cp sums channel t over the voxel lanes (fixed order -> deterministic)
    if (threadIdx.x < cp) {
        const int c = threadIdx.x, q = c >> 2, j = c & 3;
        float s1 = 0.f, s2 = 0.f;
        for (int l = 0; l < tpv; ++l) {
            s1 += red[(l * nq + q) * 8 + j];
            s2 += red[(l * nq + q) * 8 + 4 + j];
        }
        st_row(tail.counter != nullptr, partials + (size_t)blockIdx.x * 2 * cp + c, s1);
        st_row(tail.counter != nullptr, partials + (size_t)blockIdx.x * 2 * cp + cp + c, s2);
    }
    if (tail.counter) bn_bwd_tail(tail, partials, gridDim.x, cp, gridDim.x);
}

// coef[0..cp) = k0 = gamma*invstd ; coef[cp..2cp) = k1 = dbeta/n ; coef[2cp..3cp) = k2 = dgamma/n ; and the same backward
// as one affine map of the raw output (bn_bwd_lazy4, common.h): coef[3cp..4cp) = A = -k0 k2 invstd,
// coef[4cp..5cp) = B = -k0 (k1 - k2 mean invstd)
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nb, int C, int cp, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef, const float* __restrict__ mean,
                                       float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                       long long* __restrict__ nbt) {
    const int c = blockIdx.x;
    if (nbt && rmean && c == 0 && threadIdx.x == 0) nbt[0] += 1;                      // the replayed update counts too
    float g_ = 0.f, is_ = 0.f, mu_ = 0.f, rm0 = 0.f, rv0 = 0.f;                      // (prefetched: see bn_finalize_kernel)
    if (threadIdx.x == 0 && c < C) {
        g_ = gamma[c]; is_ = invstd[c]; mu_ = mean[c];
        if (rmean) { rm0 = rmean[c]; rv0 = rvar[c]; }
    }
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int b = threadIdx.x; b < nb; b += EW_BLOCK) {
            s1 += (double)partials[(size_t)b * 2 * cp + c];
            s2 += (double)partials[(size_t)b * 2 * cp + cp + c];
        }
    }
    block_sum2<EW_BLOCK>(s1, s2);
    if (threadIdx.x == 0) {
        if (c < C) {
            dbeta[c] = (float)s1;
            dgamma[c] = (float)s2;
            const float k0 = g_ * is_, k1 = (float)(s1 / count), k2 = (float)(s2 / count);
            coef[c] = k0;
            coef[cp + c] = k1;
            coef[2 * cp + c] = k2;
            coef[3 * cp + c] = -k0 * k2 * is_;
            coef[4 * cp + c] = -k0 * (k1 - k2 * mu_ * is_);
            if (rmean) {
                // the running-stat update torch.utils.checkpoint's recompute repeats in backward (models.py:232-255):
                // same batch statistics as the forward update; the biased variance is recovered from invstd
                const double istd = (double)is_;
                double var = 1.0 / (istd * istd) - (double)eps;
                if (var < 0.0) var = 0.0;
                const float unb = (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
                rmean[c] = (1.f - momentum) * rm0 + momentum * mu_;
                rvar[c] = (1.f - momentum) * rv0 + momentum * unb;
            }
        } else {
            coef[c] = 0.f; coef[cp + c] = 0.f; coef[2 * cp + c] = 0.f; coef[3 * cp + c] = 0.f; coef[4 * cp + c] = 0.f;
        }
    }
}

template <class T>
__global__ void bn_relu_bwd_apply_kernel(const T* __restrict__ y, int y_cs, T* __restrict__ ga, int g_cs,
                                         int cp, const float* __restrict__ scale, const float* __restrict__ shift,
                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                         const float* __restrict__ coef, int64_t nvox) {
    const int nq = cp >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvox * nq) return;
    const int qd = (int)(idx % nq);
    const int64_t v = idx / nq;
    const float4 sc = *reinterpret_cast<const float4*>(scale + qd * 4);
    const float4 sh = *reinterpret_cast<const float4*>(shift + qd * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + qd * 4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + qd * 4);
    const float4 k0 = *reinterpret_cast<const float4*>(coef + qd * 4);
    const float4 k1 = *reinterpret_cast<const float4*>(coef + cp + qd * 4);
    const float4 k2 = *reinterpret_cast<const float4*>(coef + 2 * cp + qd * 4);
    const float4 yy = ld4<T>(y + v * y_cs + qd * 4);
    float4 gg = ld4<T>(ga + v * g_cs + qd * 4);
    float gz;
    gz = (fmaf(yy.x, sc.x, sh.x) > 0.f) ? gg.x : 0.f; gg.x = k0.x * (gz - k1.x - (yy.x - mu.x) * is.x * k2.x);
    gz = (fmaf(yy.y, sc.y, sh.y) > 0.f) ? gg.y : 0.f; gg.y = k0.y * (gz - k1.y - (yy.y - mu.y) * is.y * k2.y);
    gz = (fmaf(yy.z, sc.z, sh.z) > 0.f) ? gg.z : 0.f; gg.z = k0.z * (gz - k1.z - (yy.z - mu.z) * is.z * k2.z);
    gz = (fmaf(yy.w, sc.w, sh.w) > 0.f) ? gg.w : 0.f; gg.w = k0.w * (gz - k1.w - (yy.w - mu.w) * is.w * k2.w);
    st4<T>(ga + v * g_cs + qd * 4, gg);
}

// ------------------------------------------------------------------ MaxPool3d(2,2)
template <class T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ in, int in_cs, int cp, const float* __restrict__ scale,
                                    const float* __restrict__ shift, int relu, T* __restrict__ out, int out_cs,
                                    int N, int D, int H, int W) {
    const int nq = cp >> 2;
    const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Do * Ho * Wo * nq;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int qd = (int)(idx % nq);
    int64_t o = idx / nq;
    const int wo = (int)(o % Wo); o /= Wo;
    const int ho = (int)(o % Ho); o /= Ho;
    const int dd = (int)(o % Do);
    const int n = (int)(o / Do);
    const bool xf = scale != nullptr;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xf) {
        sc = *reinterpret_cast<const float4*>(scale + qd * 4);
        sh = *reinterpret_cast<const float4*>(shift + qd * 4);
    }
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int d = dd * 2 + (t >> 2), h = ho * 2 + ((t >> 1) & 1), w = wo * 2 + (t & 1);
        const size_t vox = (((size_t)n * D + d) * H + h) * W + w;
        float4 v = ld4<T>(in + vox * in_cs + qd * 4);
        if (xf) v = xform4(v, sc, sh, relu);
        best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y);
        best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
    }
    const size_t ovox = (((size_t)n * Do + dd) * Ho + ho) * Wo + wo;
    st4<T>(out + ovox * out_cs + qd * 4, best);
}

// ---- additive skip connections (UNet(cat=False), models.py:250-251): out = act(a) + act(b), act = lazy BN + ReLU
template <class T>
__global__ void skip_add_kernel(const T* __restrict__ a, int a_cs, const float* __restrict__ a_scale,
                                const float* __restrict__ a_shift, int a_relu, const T* __restrict__ b, int b_cs,
                                const float* __restrict__ b_scale, const float* __restrict__ b_shift, int b_relu,
                                T* __restrict__ out, int out_cs, int cp, int64_t nvox) {
    const int nq = cp >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvox * nq) return;
    const int qd = (int)(idx % nq);
    const int64_t v = idx / nq;
    float4 x = ld4<T>(a + v * a_cs + qd * 4);
    if (a_scale)
        x = xform4(x, *reinterpret_cast<const float4*>(a_scale + qd * 4), *reinterpret_cast<const float4*>(a_shift + qd * 4), a_relu);
    if (b) {
        float4 y = ld4<T>(b + v * b_cs + qd * 4);
        if (b_scale)
            y = xform4(y, *reinterpret_cast<const float4*>(b_scale + qd * 4), *reinterpret_cast<const float4*>(b_shift + qd * 4), b_relu);
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    }
    st4<T>(out + v * out_cs + qd * 4, x);
}

// RED: the pooled tensor is relu(BN(raw conv output)) and gin is the gradient w.r.t. that activated tensor, complete once
// this kernel has added its share -- so the BatchNorm-backward reduction of that layer (sum gz, sum gz * xhat per channel,
// gz = gradient masked by the ReLU) rides along: one block-partial row per block, same layout as bn_relu_bwd_reduce.
template <bool RED, class T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ in, int in_cs, int cp, const float* __restrict__ scale,
                                    const float* __restrict__ shift, int relu, const T* __restrict__ gout,
                                    int gout_cs, T* __restrict__ gin, int gin_cs, int accumulate, int N, int D,
                                    int H, int W, const float* __restrict__ mean, const float* __restrict__ invstd,
                                    float* __restrict__ partials, ctu_bn_bwd_tail tail) {
    const int nq = cp >> 2;
    const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Do * Ho * Wo * nq;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
    if (idx < total) {
    const int qd = (int)(idx % nq);
    int64_t o = idx / nq;
    const int wo = (int)(o % Wo); o /= Wo;
    const int ho = (int)(o % Ho); o /= Ho;
    const int dd = (int)(o % Do);
    const int n = (int)(o / Do);
    const bool xf = scale != nullptr;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xf) {
        sc = *reinterpret_cast<const float4*>(scale + qd * 4);
        sh = *reinterpret_cast<const float4*>(shift + qd * 4);
    }
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
    if constexpr (RED) {
        mu = *reinterpret_cast<const float4*>(mean + qd * 4);
        is = *reinterpret_cast<const float4*>(invstd + qd * 4);
    }
    float4 raw[8];
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int d = dd * 2 + (t >> 2), h = ho * 2 + ((t >> 1) & 1), w = wo * 2 + (t & 1);
        const size_t vox = (((size_t)n * D + d) * H + h) * W + w;
        float4 v = ld4<T>(in + vox * in_cs + qd * 4);
        raw[t] = v;
        if (xf) v = xform4(v, sc, sh, relu);
        // first maximum in (d,h,w) scan order wins, as ATen's max_pool3d does (strict >)
        if (v.x > best.x) { best.x = v.x; bi[0] = t; }
        if (v.y > best.y) { best.y = v.y; bi[1] = t; }
        if (v.z > best.z) { best.z = v.z; bi[2] = t; }
        if (v.w > best.w) { best.w = v.w; bi[3] = t; }
    }
    const size_t ovox = (((size_t)n * Do + dd) * Ho + ho) * Wo + wo;
    const float4 g = ld4<T>(gout + ovox * gout_cs + qd * 4);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int d = dd * 2 + (t >> 2), h = ho * 2 + ((t >> 1) & 1), w = wo * 2 + (t & 1);
        const size_t vox = (((size_t)n * D + d) * H + h) * W + w;
        T* gp = gin + vox * gin_cs + qd * 4;
        float4 r = accumulate ? ld4<T>(gp) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (bi[0] == t) r.x += g.x;
        if (bi[1] == t) r.y += g.y;
        if (bi[2] == t) r.z += g.z;
        if (bi[3] == t) r.w += g.w;
        r = rnd4<T>(r);                      // the reduction sees what the BatchNorm-backward apply pass will read back
        st4<T>(gp, r);
        if constexpr (RED) {
            const float4 y = raw[t];
            float gz;
            gz = (fmaf(y.x, sc.x, sh.x) > 0.f) ? r.x : 0.f; a1.x += gz; a2.x += gz * (y.x - mu.x) * is.x;
            gz = (fmaf(y.y, sc.y, sh.y) > 0.f) ? r.y : 0.f; a1.y += gz; a2.y += gz * (y.y - mu.y) * is.y;
            gz = (fmaf(y.z, sc.z, sh.z) > 0.f) ? r.z : 0.f; a1.z += gz; a2.z += gz * (y.z - mu.z) * is.z;
            gz = (fmaf(y.w, sc.w, sh.w) > 0.f) ? r.w : 0.f; a1.w += gz; a2.w += gz * (y.w - mu.w) * is.w;
        }
    }
    }
    if constexpr (RED) {
        // thread t < cp sums channel t over the block's voxel lanes in a fixed order (threadIdx % nq is the channel quad)
        __shared__ float red[EW_BLOCK * 8];
        float* r = &red[threadIdx.x * 8];
        r[0] = a1.x; r[1] = a1.y; r[2] = a1.z; r[3] = a1.w; r[4] = a2.x; r[5] = a2.y; r[6] = a2.z; r[7] = a2.w;
        __syncthreads();
        if (threadIdx.x < cp) {
            const int c = threadIdx.x, q = c >> 2, j = c & 3;
            float s1 = 0.f, s2 = 0.f;
            for (int l = 0; l < EW_BLOCK / nq; ++l) {
                s1 += red[(l * nq + q) * 8 + j];
                s2 += red[(l * nq + q) * 8 + 4 + j];
            }
            st_row(tail.counter != nullptr, partials + (size_t)blockIdx.x * 2 * cp + c, s1);
            st_row(tail.counter != nullptr, partials + (size_t)blockIdx.x * 2 * cp + cp + c, s2);
        }
        if (tail.counter) bn_bwd_tail(tail, partials, gridDim.x, cp, gridDim.x);
    }
}

// ------------------------------------------------------------------ channel sums
template <class T>
__global__ void channel_sum_partial_kernel(const T* __restrict__ x, int cs, int cp, int64_t nvox,
                                           float* __restrict__ partials) {
    const int nq = cp >> 2;
    const int tpv = EW_BLOCK / nq;
    const int qd = threadIdx.x % nq, vl = threadIdx.x / nq;
    __shared__ float red[EW_BLOCK * 4];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vl < tpv)
        for (int64_t v = (int64_t)blockIdx.x * tpv + vl; v < nvox; v += (int64_t)gridDim.x * tpv) {
            const float4 t = ld4<T>(x + v * cs + qd * 4);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
    float* r = &red[threadIdx.x * 4];
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
    __syncthreads();
    if (threadIdx.x < cp) {
        const int c = threadIdx.x, q = c >> 2, j = c & 3;
        float s = 0.f;
        for (int l = 0; l < tpv; ++l) s += red[(l * nq + q) * 4 + j];
        partials[(size_t)blockIdx.x * cp + c] = s;
    }
}

// one block per channel, fixed-order tree -> deterministic
__global__ void channel_sum_final_kernel(const float* __restrict__ partials, int nb, int cp, float* __restrict__ out,
                                         int C) {
    const int c = blockIdx.x;
    __shared__ double r[EW_BLOCK];
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) s += (double)partials[(size_t)b * cp + c];
    r[threadIdx.x] = s;
    __syncthreads();
    for (int o = EW_BLOCK / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) r[threadIdx.x] += r[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = (float)r[0];
}

// ------------------------------------------------------------------ Adam(amsgrad)
// Pointer table passed BY VALUE in the kernel arguments (<= 64 tensors, 3 KB): no device-side table to keep
// in sync, and the launch is graph-capturable.  The step counter lives on the device so a replayed graph
// still advances the bias corrections.
constexpr int ADAM_MAXT = 64;
struct AdamTable {
    float* p[ADAM_MAXT];
    const float* g[ADAM_MAXT];
    float* m[ADAM_MAXT];
    float* v[ADAM_MAXT];
    float* vm[ADAM_MAXT];
    int64_t n[ADAM_MAXT];
};

// skip: optional device flag (float[1]); non-zero = this step's gradients overflowed (ctu_scale_tensors found a non-finite
// value while un-scaling fp16 gradients): the whole update is skipped -- parameters, moments AND the step counter stay
// as they are, what torch.amp.GradScaler.step does.  Works inside a replayed graph (the flag is read on the device).
__global__ void adam_step_inc_kernel(float* step, const float* __restrict__ skip) {
    if (skip && skip[0] != 0.f) return;
    step[0] += 1.f;
}

__global__ void adam_amsgrad_kernel(AdamTable tb, const float* __restrict__ step, float lr, float beta1, float beta2,
                                    float omb1, float omb2, float eps, float wd, int decoupled, const float* __restrict__ skip) {
    if (skip && skip[0] != 0.f) return;
    const int t = blockIdx.y;
    const int64_t n = tb.n[t];
    float* __restrict__ p = tb.p[t];
    const float* __restrict__ g = tb.g[t];
    float* __restrict__ m = tb.m[t];
    float* __restrict__ v = tb.v[t];
    float* __restrict__ vm = tb.vm[t];
    // bias corrections in double, like the Python scalars of torch.optim.Adam
    const double st = (double)step[0];
    const float bc1 = (float)(1.0 - pow((double)beta1, st));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, st));
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gr = g[i];
        float pv = p[i];
        if (wd != 0.f) {
            if (decoupled) pv *= 1.f - lr * wd;     // AdamW
            else gr = fmaf(wd, pv, gr);            // Adam (L2)
        }
        // torch.optim.Adam single-tensor form: lerp for exp_avg, mul/addcmul for exp_avg_sq
        const float mi = m[i] + (gr - m[i]) * omb1;          // omb = 1 - beta formed in double on the host, as torch does
        const float vi = v[i] * beta2 + omb2 * gr * gr;
        const float vmx = fmaxf(vm[i], vi);
        m[i] = mi; v[i] = vi; vm[i] = vmx;
        const float denom = sqrtf(vmx) / bc2_sqrt + eps;
        p[i] = pv - step_size * (mi / denom);
    }
}

struct ScaleTable {
    float* p[ADAM_MAXT];
    int64_t n[ADAM_MAXT];
};

// nonfinite: optional device flag (float[1]) set to 1 when any scaled value is inf / NaN (the fp16 backward overflowed)
__global__ void scale_tensors_kernel(ScaleTable tb, float s, float* __restrict__ nonfinite) {
    const int t = blockIdx.y;
    float* __restrict__ p = tb.p[t];
    const int64_t n = tb.n[t];
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = p[i] * s;
        p[i] = v;
        bad |= !(fabsf(v) <= 3.4028234e38f);            // inf or NaN
    }
    if (nonfinite && bad) nonfinite[0] = 1.f;          // (every writer stores the same value)
}

}  // namespace

// =================================================================== C ABI
// The activation-touching entry points exist twice: the fp32 ones (float*) and ctu_lp_* (dtype code + void*, 16-bit
// storage); both are thin wrappers over one template per operation.
namespace {

template <class T>
int ncdhw_to_ndhwc_impl(const float* src, T* dst, int N, int C, int D, int H, int W, int cp, int cs, void* stream) {
    CTU_REQUIRE(src && dst && N > 0 && C > 0, "ncdhw_to_ndhwc: null/empty");
    CTU_REQUIRE(cp % 8 == 0 && cp >= C && cs >= cp && cs % 4 == 0, "ncdhw_to_ndhwc: cp=%d cs=%d C=%d", cp, cs, C);
    const int64_t V = (int64_t)D * H * W, tv = V * N;
    const int64_t total = tv * (cp >> 2);
    ncdhw_to_ndhwc_kernel<T><<<(unsigned)ceil_div64(total, EW_BLOCK), EW_BLOCK, 0, (hipStream_t)stream>>>(src, dst, C, V, tv, cp, cs);
    CTU_CHECK_LAUNCH("ncdhw_to_ndhwc");
    return CTU_OK;
}

template <class T>
int ndhwc_to_ncdhw_impl(const T* src, float* dst, int N, int C, int D, int H, int W, int cs, void* stream) {
    CTU_REQUIRE(src && dst && N > 0 && C > 0 && cs >= C, "ndhwc_to_ncdhw: bad argument");
    const int64_t V = (int64_t)D * H * W, tv = V * N;
    ndhwc_to_ncdhw_kernel<T><<<(unsigned)ceil_div64(tv * C, EW_BLOCK), EW_BLOCK, 0, (hipStream_t)stream>>>(src, dst, C, V, tv, cs);
    CTU_CHECK_LAUNCH("ndhwc_to_ncdhw");
    return CTU_OK;
}

template <class T>
int bn_relu_bwd_reduce_impl(const T* y, int y_cs, const T* ga, int g_cs, int cp, const float* scale, const float* shift,
                            const float* mean, const float* invstd, int64_t nvox, float* partials, const ctu_bn_bwd_tail* tail,
                            void* stream) {
    CTU_REQUIRE(y && ga && scale && shift && mean && invstd && partials, "bn_relu_bwd_reduce: null pointer");
    CTU_REQUIRE(!tail || (tail->counter && tail->gamma && tail->invstd && tail->dgamma && tail->dbeta && tail->coef &&
                          tail->C > 0 && tail->C <= cp && tail->count > 0 && (!tail->running_mean || (tail->mean && tail->running_var))),
                "bn_relu_bwd_reduce: incomplete BatchNorm tail");
    CTU_REQUIRE(cp % 8 == 0 && cp > 0 && cp <= 256 && y_cs % 4 == 0 && g_cs % 4 == 0, "bn_relu_bwd_reduce: cp=%d", cp);
    bn_relu_bwd_reduce_kernel<T><<<ctu_bn_bwd_num_blocks(nvox), EW_BLOCK, 0, (hipStream_t)stream>>>(
        y, y_cs, ga, g_cs, cp, scale, shift, mean, invstd, nvox, partials, bwd_tail_or_off(tail));
    CTU_CHECK_LAUNCH("bn_relu_bwd_reduce");
    return CTU_OK;
}

template <class T>
int bn_relu_bwd_apply_impl(const T* y, int y_cs, T* ga, int g_cs, int cp, const float* scale, const float* shift,
                           const float* mean, const float* invstd, const float* coef, int64_t nvox, void* stream) {
    CTU_REQUIRE(y && ga && scale && shift && mean && invstd && coef, "bn_relu_bwd_apply: null pointer");
    CTU_REQUIRE(cp % 8 == 0 && cp > 0, "bn_relu_bwd_apply: cp=%d", cp);
    const int64_t total = nvox * (cp >> 2);
    bn_relu_bwd_apply_kernel<T><<<(unsigned)ceil_div64(total, EW_BLOCK), EW_BLOCK, 0, (hipStream_t)stream>>>(
        y, y_cs, ga, g_cs, cp, scale, shift, mean, invstd, coef, nvox);
    CTU_CHECK_LAUNCH("bn_relu_bwd_apply");
    return CTU_OK;
}

template <class T>
int maxpool2_fwd_impl(const T* in, int in_cs, int cp, const float* in_scale, const float* in_shift, int in_relu, T* out,
                      int out_cs, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(in && out, "maxpool2_fwd: null pointer");
    CTU_REQUIRE(cp % 8 == 0 && cp > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool2_fwd: D,H,W must be even");
    const int64_t total = (int64_t)N * (D / 2) * (H / 2) * (W / 2) * (cp >> 2);
    maxpool2_fwd_kernel<T><<<(unsigned)ceil_div64(total, EW_BLOCK), EW_BLOCK, 0, (hipStream_t)stream>>>(
        in, in_cs, cp, in_scale, in_shift, in_relu, out, out_cs, N, D, H, W);
    CTU_CHECK_LAUNCH("maxpool2_fwd");
    return CTU_OK;
}

template <class T>
int maxpool2_bwd_impl(const T* in, int in_cs, int cp, const float* in_scale, const float* in_shift, int in_relu, const T* gout,
                      int gout_cs, T* gin, int gin_cs, int accumulate, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(in && gout && gin, "maxpool2_bwd: null pointer");
    CTU_REQUIRE(cp % 8 == 0 && cp > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool2_bwd: D,H,W must be even");
    const int64_t total = (int64_t)N * (D / 2) * (H / 2) * (W / 2) * (cp >> 2);
    maxpool2_bwd_kernel<false, T><<<(unsigned)ceil_div64(total, EW_BLOCK), EW_BLOCK, 0, (hipStream_t)stream>>>(
        in, in_cs, cp, in_scale, in_shift, in_relu, gout, gout_cs, gin, gin_cs, accumulate, N, D, H, W, nullptr, nullptr, nullptr,
        bwd_tail_or_off(nullptr));
    CTU_CHECK_LAUNCH("maxpool2_bwd");
    return CTU_OK;
}

template <class T>
int maxpool2_bwd_bn_impl(const T* in, int in_cs, int cp, const float* in_scale, const float* in_shift, const float* mean,
                         const float* invstd, const T* gout, int gout_cs, T* gin, int gin_cs, int accumulate, int N, int D,
                         int H, int W, float* partials, const ctu_bn_bwd_tail* tail, void* stream) {
    CTU_REQUIRE(in && gout && gin && in_scale && in_shift && mean && invstd && partials, "maxpool2_bwd_bn: null pointer");
    CTU_REQUIRE(!tail || (tail->counter && tail->gamma && tail->invstd && tail->dgamma && tail->dbeta && tail->coef &&
                          tail->C > 0 && tail->C <= cp && tail->count > 0 && (!tail->running_mean || (tail->mean && tail->running_var))),
                "maxpool2_bwd_bn: incomplete BatchNorm tail");
    CTU_REQUIRE(cp % 8 == 0 && cp > 0 && cp <= EW_BLOCK && EW_BLOCK % (cp >> 2) == 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0,
                "maxpool2_bwd_bn: cp=%d must be a multiple of 8 whose quads divide the block; D,H,W even", cp);
    const int nb = ctu_maxpool2_bwd_bn_num_blocks(N, D, H, W, cp);
    maxpool2_bwd_kernel<true, T><<<nb, EW_BLOCK, 0, (hipStream_t)stream>>>(
        in, in_cs, cp, in_scale, in_shift, 1, gout, gout_cs, gin, gin_cs, accumulate, N, D, H, W, mean, invstd, partials,
        bwd_tail_or_off(tail));
    CTU_CHECK_LAUNCH("maxpool2_bwd_bn");
    return CTU_OK;
}

template <class T>
int skip_add_impl(const T* a, int a_cs, const float* a_scale, const float* a_shift, int a_relu, const T* b, int b_cs,
                  const float* b_scale, const float* b_shift, int b_relu, T* out, int out_cs, int cp, int64_t nvox, void* stream) {
    CTU_REQUIRE(a && out, "skip_add: null pointer");
    CTU_REQUIRE(cp % 4 == 0 && cp > 0 && a_cs % 4 == 0 && out_cs % 4 == 0 && (!b || b_cs % 4 == 0), "skip_add: channel counts / strides must be multiples of 4");
    CTU_REQUIRE((a_scale == nullptr) == (a_shift == nullptr) && (b_scale == nullptr) == (b_shift == nullptr), "skip_add: scale/shift come in pairs");
    if (nvox <= 0) return CTU_OK;
    const int64_t total = nvox * (cp >> 2);
    skip_add_kernel<T><<<(unsigned)ceil_div64(total, EW_BLOCK), EW_BLOCK, 0, (hipStream_t)stream>>>(
        a, a_cs, a_scale, a_shift, a_relu, b, b_cs, b_scale, b_shift, b_relu, out, out_cs, cp, nvox);
    CTU_CHECK_LAUNCH("skip_add");
    return CTU_OK;
}

template <class T>
int channel_sum_impl(const T* x, int cs, int cp, int64_t nvox, float* partials, float* out, int C, void* stream) {
    CTU_REQUIRE(x && partials && out, "channel_sum: null pointer");
    CTU_REQUIRE(cp % 8 == 0 && cp > 0 && cp <= 256 && C <= cp && cs % 4 == 0, "channel_sum: cp=%d", cp);
    const int nb = ctu_channel_sum_num_blocks(nvox);
    channel_sum_partial_kernel<T><<<nb, EW_BLOCK, 0, (hipStream_t)stream>>>(x, cs, cp, nvox, partials);
    CTU_CHECK_LAUNCH("channel_sum_partial");
    channel_sum_final_kernel<<<C, EW_BLOCK, 0, (hipStream_t)stream>>>(partials, nb, cp, out, C);
    CTU_CHECK_LAUNCH("channel_sum_final");
    return CTU_OK;
}

}  // namespace

extern "C" int ctu_ncdhw_to_ndhwc(const float* src, float* dst, int N, int C, int D, int H, int W, int cp, int cs,
                                  void* stream) {
    return ncdhw_to_ndhwc_impl<float>(src, dst, N, C, D, H, W, cp, cs, stream);
}
extern "C" int ctu_lp_ncdhw_to_ndhwc(int dtype, const float* src, void* dst, int N, int C, int D, int H, int W, int cp, int cs,
                                     void* stream) {
    CTU_DISPATCH_LP(dtype, return ncdhw_to_ndhwc_impl<T>(src, (T*)dst, N, C, D, H, W, cp, cs, stream));
}

extern "C" int ctu_ndhwc_to_ncdhw(const float* src, float* dst, int N, int C, int D, int H, int W, int cs,
                                  void* stream) {
    return ndhwc_to_ncdhw_impl<float>(src, dst, N, C, D, H, W, cs, stream);
}
extern "C" int ctu_lp_ndhwc_to_ncdhw(int dtype, const void* src, float* dst, int N, int C, int D, int H, int W, int cs,
                                     void* stream) {
    CTU_DISPATCH_LP(dtype, return ndhwc_to_ncdhw_impl<T>((const T*)src, dst, N, C, D, H, W, cs, stream));
}

extern "C" int ctu_bn_finalize(const float* stats, int nblocks, int C, int cp, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               int n_updates, float* scale, float* shift, float* mean_out, float* invstd_out,
                               long long* num_batches_tracked, void* stream) {
    CTU_REQUIRE(stats && gamma && beta && scale && shift && mean_out && invstd_out, "bn_finalize: null pointer");
    CTU_REQUIRE(C > 0 && cp >= C && cp % 8 == 0 && nblocks > 0 && count > 0, "bn_finalize: bad sizes");
    bn_finalize_kernel<<<cp, FIN_BLOCK, 0, (hipStream_t)stream>>>(stats, nblocks, C, cp, count, gamma, beta,
                                                                running_mean, running_var, momentum, eps, n_updates,
                                                                scale, shift, mean_out, invstd_out, num_batches_tracked);
    CTU_CHECK_LAUNCH("bn_finalize");
    return CTU_OK;
}

extern "C" int ctu_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, int cp, float* scale, float* shift,
                                  void* stream) {
    CTU_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, "bn_eval_affine: null pointer");
    CTU_REQUIRE(C > 0 && cp >= C && cp % 8 == 0, "bn_eval_affine: bad sizes");
    bn_eval_affine_kernel<<<ceil_div(cp, 64), 64, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var,
                                                                           eps, C, cp, scale, shift);
    CTU_CHECK_LAUNCH("bn_eval_affine");
    return CTU_OK;
}

extern "C" int ctu_bn_bwd_num_blocks(int64_t nvox) {
    int64_t nb = ceil_div64(nvox, 64);
    if (nb > MAX_RED_BLOCKS) nb = MAX_RED_BLOCKS;
    if (nb < 1) nb = 1;
    return (int)nb;
}

extern "C" int ctu_bn_relu_bwd_reduce(const float* y, int y_cs, const float* ga, int g_cs, int cp, const float* scale,
                                      const float* shift, const float* mean, const float* invstd, int64_t nvox,
                                      float* partials, const ctu_bn_bwd_tail* tail, void* stream) {
    return bn_relu_bwd_reduce_impl<float>(y, y_cs, ga, g_cs, cp, scale, shift, mean, invstd, nvox, partials, tail, stream);
}
extern "C" int ctu_lp_bn_relu_bwd_reduce(int dtype, const void* y, int y_cs, const void* ga, int g_cs, int cp, const float* scale,
                                         const float* shift, const float* mean, const float* invstd, int64_t nvox,
                                         float* partials, const ctu_bn_bwd_tail* tail, void* stream) {
    CTU_DISPATCH_LP(dtype, return bn_relu_bwd_reduce_impl<T>((const T*)y, y_cs, (const T*)ga, g_cs, cp, scale, shift, mean, invstd,
                                                             nvox, partials, tail, stream));
}

extern "C" int ctu_bn_bwd_finalize(const float* partials, int nb, int C, int cp, double count, const float* gamma,
                                   const float* invstd, float* dgamma, float* dbeta, float* coef, const float* mean,
                                   float* running_mean, float* running_var, float momentum, float eps,
                                   long long* num_batches_tracked, void* stream) {
    CTU_REQUIRE(partials && gamma && invstd && dgamma && dbeta && coef && mean, "bn_bwd_finalize: null pointer");
    CTU_REQUIRE((running_mean == nullptr) == (running_var == nullptr) && (!running_mean || mean),
                "bn_bwd_finalize: the running-stat replay needs mean, running_mean and running_var together");
    bn_bwd_finalize_kernel<<<cp, EW_BLOCK, 0, (hipStream_t)stream>>>(partials, nb, C, cp, count, gamma, invstd, dgamma,
                                                                    dbeta, coef, mean, running_mean, running_var,
                                                                    momentum, eps, num_batches_tracked);
    CTU_CHECK_LAUNCH("bn_bwd_finalize");
    return CTU_OK;
}

extern "C" int ctu_bn_relu_bwd_apply(const float* y, int y_cs, float* ga, int g_cs, int cp, const float* scale,
                                     const float* shift, const float* mean, const float* invstd, const float* coef,
                                     int64_t nvox, void* stream) {
    return bn_relu_bwd_apply_impl<float>(y, y_cs, ga, g_cs, cp, scale, shift, mean, invstd, coef, nvox, stream);
}
extern "C" int ctu_lp_bn_relu_bwd_apply(int dtype, const void* y, int y_cs, void* ga, int g_cs, int cp, const float* scale,
                                        const float* shift, const float* mean, const float* invstd, const float* coef,
                                        int64_t nvox, void* stream) {
    CTU_DISPATCH_LP(dtype, return bn_relu_bwd_apply_impl<T>((const T*)y, y_cs, (T*)ga, g_cs, cp, scale, shift, mean, invstd, coef,
                                                            nvox, stream));
}

extern "C" int ctu_maxpool2_fwd(const float* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                                int in_relu, float* out, int out_cs, int N, int D, int H, int W, void* stream) {
    return maxpool2_fwd_impl<float>(in, in_cs, cp, in_scale, in_shift, in_relu, out, out_cs, N, D, H, W, stream);
}
extern "C" int ctu_lp_maxpool2_fwd(int dtype, const void* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                                   int in_relu, void* out, int out_cs, int N, int D, int H, int W, void* stream) {
    CTU_DISPATCH_LP(dtype, return maxpool2_fwd_impl<T>((const T*)in, in_cs, cp, in_scale, in_shift, in_relu, (T*)out, out_cs, N, D,
                                                       H, W, stream));
}

extern "C" int ctu_maxpool2_bwd(const float* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                                int in_relu, const float* gout, int gout_cs, float* gin, int gin_cs, int accumulate,
                                int N, int D, int H, int W, void* stream) {
    return maxpool2_bwd_impl<float>(in, in_cs, cp, in_scale, in_shift, in_relu, gout, gout_cs, gin, gin_cs, accumulate, N, D, H, W,
                                    stream);
}
extern "C" int ctu_lp_maxpool2_bwd(int dtype, const void* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                                   int in_relu, const void* gout, int gout_cs, void* gin, int gin_cs, int accumulate,
                                   int N, int D, int H, int W, void* stream) {
    CTU_DISPATCH_LP(dtype, return maxpool2_bwd_impl<T>((const T*)in, in_cs, cp, in_scale, in_shift, in_relu, (const T*)gout, gout_cs,
                                                       (T*)gin, gin_cs, accumulate, N, D, H, W, stream));
}

extern "C" int ctu_maxpool2_bwd_bn_num_blocks(int N, int D, int H, int W, int cp) {
    // 0 = not available for this channel count (the in-block reduction needs threadIdx % (cp/4) to be the channel quad)
    if (cp <= 0 || cp % 8 != 0 || cp > EW_BLOCK || EW_BLOCK % (cp >> 2) != 0) return 0;
    return (int)ceil_div64((int64_t)N * (D / 2) * (H / 2) * (W / 2) * (cp >> 2), EW_BLOCK);
}

extern "C" int ctu_maxpool2_bwd_bn(const float* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                                   const float* mean, const float* invstd, const float* gout, int gout_cs, float* gin,
                                   int gin_cs, int accumulate, int N, int D, int H, int W, float* partials, const ctu_bn_bwd_tail* tail, void* stream) {
    return maxpool2_bwd_bn_impl<float>(in, in_cs, cp, in_scale, in_shift, mean, invstd, gout, gout_cs, gin, gin_cs, accumulate, N, D,
                                       H, W, partials, tail, stream);
}
extern "C" int ctu_lp_maxpool2_bwd_bn(int dtype, const void* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                                      const float* mean, const float* invstd, const void* gout, int gout_cs, void* gin,
                                      int gin_cs, int accumulate, int N, int D, int H, int W, float* partials, const ctu_bn_bwd_tail* tail, void* stream) {
    CTU_DISPATCH_LP(dtype, return maxpool2_bwd_bn_impl<T>((const T*)in, in_cs, cp, in_scale, in_shift, mean, invstd, (const T*)gout,
                                                          gout_cs, (T*)gin, gin_cs, accumulate, N, D, H, W, partials, tail, stream));
}

extern "C" int ctu_skip_add(const float* a, int a_cs, const float* a_scale, const float* a_shift, int a_relu,
                            const float* b, int b_cs, const float* b_scale, const float* b_shift, int b_relu,
                            float* out, int out_cs, int cp, int64_t nvox, void* stream) {
    return skip_add_impl<float>(a, a_cs, a_scale, a_shift, a_relu, b, b_cs, b_scale, b_shift, b_relu, out, out_cs, cp, nvox, stream);
}
extern "C" int ctu_lp_skip_add(int dtype, const void* a, int a_cs, const float* a_scale, const float* a_shift, int a_relu,
                               const void* b, int b_cs, const float* b_scale, const float* b_shift, int b_relu,
                               void* out, int out_cs, int cp, int64_t nvox, void* stream) {
    CTU_DISPATCH_LP(dtype, return skip_add_impl<T>((const T*)a, a_cs, a_scale, a_shift, a_relu, (const T*)b, b_cs, b_scale, b_shift,
                                                   b_relu, (T*)out, out_cs, cp, nvox, stream));
}

extern "C" int ctu_channel_sum_num_blocks(int64_t nvox) {
    int64_t nb = ceil_div64(nvox, 256);
    if (nb > MAX_RED_BLOCKS) nb = MAX_RED_BLOCKS;
    if (nb < 1) nb = 1;
    return (int)nb;
}

extern "C" int ctu_channel_sum(const float* x, int cs, int cp, int64_t nvox, float* partials, float* out, int C,
                               void* stream) {
    return channel_sum_impl<float>(x, cs, cp, nvox, partials, out, C, stream);
}
extern "C" int ctu_lp_channel_sum(int dtype, const void* x, int cs, int cp, int64_t nvox, float* partials, float* out, int C,
                                  void* stream) {
    CTU_DISPATCH_LP(dtype, return channel_sum_impl<T>((const T*)x, cs, cp, nvox, partials, out, C, stream));
}

extern "C" int ctu_adam_amsgrad(void* const* ptrs, const int64_t* sizes, int n, float* step, double lr, double beta1,
                                double beta2, double eps, double weight_decay, int decoupled, const float* skip_flag, void* stream) {
    CTU_REQUIRE(ptrs && sizes && n > 0 && step, "adam_amsgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    adam_step_inc_kernel<<<1, 1, 0, st>>>(step, skip_flag);
    CTU_CHECK_LAUNCH("adam_step_inc");
    for (int t0 = 0; t0 < n; t0 += ADAM_MAXT) {
        const int nt = (n - t0) < ADAM_MAXT ? (n - t0) : ADAM_MAXT;
        AdamTable tb;
        int64_t mx = 1;
        for (int t = 0; t < nt; ++t) {
            void* const* q = ptrs + (size_t)(t0 + t) * 5;
            CTU_REQUIRE(q[0] && q[1] && q[2] && q[3] && q[4], "adam_amsgrad: null tensor pointer at %d", t0 + t);
            tb.p[t] = (float*)q[0]; tb.g[t] = (const float*)q[1]; tb.m[t] = (float*)q[2]; tb.v[t] = (float*)q[3];
            tb.vm[t] = (float*)q[4]; tb.n[t] = sizes[t0 + t];
            if (tb.n[t] > mx) mx = tb.n[t];
        }
        int gx = (int)ceil_div64(mx, EW_BLOCK * 4);
        if (gx > 128) gx = 128;
        if (gx < 1) gx = 1;
        adam_amsgrad_kernel<<<dim3(gx, nt), EW_BLOCK, 0, st>>>(tb, step, (float)lr, (float)beta1, (float)beta2,
                                                              (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                                                              (float)weight_decay, decoupled, skip_flag);
        CTU_CHECK_LAUNCH("adam_amsgrad");
    }
    return CTU_OK;
}

extern "C" int ctu_scale_tensors(void* const* ptrs, const int64_t* sizes, int n, float sc, float* nonfinite_flag, void* stream) {
    CTU_REQUIRE(ptrs && sizes && n > 0, "scale_tensors: bad argument");
    for (int t0 = 0; t0 < n; t0 += ADAM_MAXT) {
        const int nt = (n - t0) < ADAM_MAXT ? (n - t0) : ADAM_MAXT;
        ScaleTable tb;
        int64_t mx = 1;
        for (int t = 0; t < nt; ++t) {
            CTU_REQUIRE(ptrs[t0 + t], "scale_tensors: null tensor pointer at %d", t0 + t);
            tb.p[t] = (float*)ptrs[t0 + t];
            tb.n[t] = sizes[t0 + t];
            if (tb.n[t] > mx) mx = tb.n[t];
        }
        int gx = (int)ceil_div64(mx, EW_BLOCK * 4);
        if (gx > 2048) gx = 2048;                                   // (dx of a 256^3 two-channel input is 134 MB)
        if (gx < 1) gx = 1;
        scale_tensors_kernel<<<dim3(gx, nt), EW_BLOCK, 0, (hipStream_t)stream>>>(tb, sc, nonfinite_flag);
        CTU_CHECK_LAUNCH("scale_tensors");
    }
    return CTU_OK;
}
