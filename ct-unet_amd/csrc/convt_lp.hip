// Reduced-precision (bf16 / fp16 storage, fp32 accumulate) ConvTranspose3d(C, C, kernel 2, stride 2) + bias for gfx950:
// the up-sampling step of every decoder block (ctunet/pytorch/models.py:37,427-429), forward, data gradient and
// weight gradient, on v_mfma_f32_16x16x32_{bf16,f16}.  Every output voxel receives exactly one tap, so the three are plain
// GEMMs whose voxel-side operand is read straight from global memory in fragment order (a lane's 8 consecutive channels
// of one voxel are one 16-byte load) -- no LDS staging for forward / data gradient; the weight gradient (K = voxels)
// reads channels-last LDS images through the hardware transpose (ds_read_b64_tr_b16).  HBM-bound (8x output stream).
#include "common.h"

namespace {

template <class T> struct MfmaT;
template <> struct MfmaT<bf16_t> {
    static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaT<f16_t> {
    static __device__ __forceinline__ f32x4 run(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// ---- packing.  mode 0 (forward): wp[tap][ks][n16][lane][8], A[row o][k r] = w[ci(r)][o][tap], r = 32 ks + 8 (l>>4) + j
//                mode 1 (data gradient): wp[ks][n16][lane][8], rows = padded input positions r, k = pair p = 4 ks + (l>>4)
//                -> (tap, 8-channel chunk of o) = (p / nch, p % nch), A[row r][k] = w[ci(r)][o][tap]
template <class T>
__global__ void lp_pack_convt_w_kernel(const float* __restrict__ w, T* __restrict__ wp, int Ci, int Co, const int32_t* __restrict__ cinv,
                                       int rin_p, int nout_p, int mode) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = idx & 7, lane = (idx >> 3) & 63, m = lane & 15, kg = lane >> 4;
    float v = 0.f;
    if (mode == 0) {
        const int ksn = (rin_p + 31) >> 5, n16 = (nout_p + 15) >> 4;
        if (idx >= 8 * ksn * n16 * 512) return;
        int r = idx >> 9;
        const int nt = r % n16; r /= n16;
        const int ks = r % ksn;
        const int tap = r / ksn;
        const int rp = ks * 32 + kg * 8 + j, o = nt * 16 + m;
        const int ci = (rp < rin_p) ? (cinv ? cinv[rp] : (rp < Ci ? rp : -1)) : -1;
        if (ci >= 0 && o < Co) v = w[((size_t)ci * Co + o) * 8 + tap];
    } else {
        // reduction side = o (rin_p = padded Co), output side = padded input positions (nout_p)
        const int nch = rin_p >> 3, ksn = 2 * nch, n16 = (nout_p + 15) >> 4;
        if (idx >= ksn * n16 * 512) return;
        int r = idx >> 9;
        const int nt = r % n16;
        const int ks = r / n16;
        const int pr = 4 * ks + kg, tap = pr / nch, ch = pr % nch;
        const int o = ch * 8 + j, rp = nt * 16 + m;
        const int ci = (rp < nout_p) ? (cinv ? cinv[rp] : (rp < Ci ? rp : -1)) : -1;
        if (ci >= 0 && o < Co) v = w[((size_t)ci * Co + o) * 8 + tap];
    }
    wp[idx] = (T)v;
}

struct LpCtP {
    const void* in;
    const void* wp;
    void* out;
    const float* scale;
    const float* shift;
    const float* bias;
    int in_cs, rin_p, relu, out_cs, nout_p, nbias;
    int N, D, H, W;         // COARSE grid
    int64_t nvox;
};

// forward: one wave = 16 coarse voxels x all taps x NT-tile groups; KSN = K-steps (ceil(rin_p / 32)) kept in registers
template <class T, int KSN>
__global__ __launch_bounds__(256) void lp_convt_fwd_kernel(LpCtP p) {
    typedef typename Vec<T>::v8 v8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kg = lane >> 4;
    const int64_t v = ((int64_t)blockIdx.x * 4 + wave) * 16 + m;
    const bool vok = v < p.nvox;
    const int64_t vc = vok ? v : 0;
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* wp = reinterpret_cast<const T*>(p.wp);
    v8 b[KSN];
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
        const int c = ks * 32 + kg * 8;
        uint4 raw = make_uint4(0u, 0u, 0u, 0u);
        if (vok && c < p.rin_p) raw = *reinterpret_cast<const uint4*>(in + vc * p.in_cs + c);
        if (p.scale && vok && c < p.rin_p) {
            const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&raw), f32x8);
            f32x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = fmaf(f[j], p.scale[c + j], p.shift[c + j]);
                o[j] = p.relu ? fmaxf(a, 0.f) : a;
            }
            *reinterpret_cast<v8*>(&raw) = __builtin_convertvector(o, v8);
        }
        b[ks] = *reinterpret_cast<v8*>(&raw);
    }
    const int wq = (int)(vc % p.W);
    int64_t t = vc / p.W;
    const int hq = (int)(t % p.H); t /= p.H;
    const int dq = (int)(t % p.D);
    const int n = (int)(t / p.D);
    const int n16 = (p.nout_p + 15) >> 4;
    T* out = reinterpret_cast<T*>(p.out);
    for (int nt = 0; nt < n16; ++nt) {
        const int cb = nt * 16 + 4 * kg;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            bv.x = cb + 0 < p.nbias ? p.bias[cb + 0] : 0.f; bv.y = cb + 1 < p.nbias ? p.bias[cb + 1] : 0.f;
            bv.z = cb + 2 < p.nbias ? p.bias[cb + 2] : 0.f; bv.w = cb + 3 < p.nbias ? p.bias[cb + 3] : 0.f;
        }
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) {
                const v8 a = *reinterpret_cast<const v8*>(wp + ((size_t)((tap * KSN + ks) * n16 + nt) * 64 + lane) * 8);
                acc = MfmaT<T>::run(a, b[ks], acc);
            }
            if (vok && cb < p.nout_p) {
                const size_t fv = (((size_t)n * 2 * p.D + 2 * dq + (tap >> 2)) * 2 * p.H + 2 * hq + ((tap >> 1) & 1)) * 2 * p.W + 2 * wq + (tap & 1);
                st4<T>(out + fv * p.out_cs + cb, make_float4(acc[0] + bv.x, acc[1] + bv.y, acc[2] + bv.z, acc[3] + bv.w));
            }
        }
    }
}

// data gradient: gin[v][r] = sum_{tap, o} gout[2v + tap][o] w[r][o][tap]; K-steps = 2 * (rout_p / 8)
template <class T>
__global__ __launch_bounds__(256) void lp_convt_bwd_data_kernel(LpCtP p) {
    typedef typename Vec<T>::v8 v8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kg = lane >> 4;
    const int64_t v = ((int64_t)blockIdx.x * 4 + wave) * 16 + m;
    const bool vok = v < p.nvox;
    const int64_t vc = vok ? v : 0;
    const T* g = reinterpret_cast<const T*>(p.in);          // fine-grid gradient, p.rin_p = padded Co
    const T* wp = reinterpret_cast<const T*>(p.wp);
    const int wq = (int)(vc % p.W);
    int64_t t = vc / p.W;
    const int hq = (int)(t % p.H); t /= p.H;
    const int dq = (int)(t % p.D);
    const int n = (int)(t / p.D);
    const int nch = p.rin_p >> 3, ksn = 2 * nch, n16 = (p.nout_p + 15) >> 4;
    T* gin = reinterpret_cast<T*>(p.out);
    for (int nt0 = 0; nt0 < n16; nt0 += 4) {                 // 4 output tiles share one pass over the gradient
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < ksn; ++ks) {
            const int pr = 4 * ks + kg, tap = pr / nch, ch = pr % nch;
            const size_t fv = (((size_t)n * 2 * p.D + 2 * dq + (tap >> 2)) * 2 * p.H + 2 * hq + ((tap >> 1) & 1)) * 2 * p.W + 2 * wq + (tap & 1);
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (vok) raw = *reinterpret_cast<const uint4*>(g + fv * p.in_cs + ch * 8);
            const v8 b = *reinterpret_cast<v8*>(&raw);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nt = min(nt0 + q, n16 - 1);
                const v8 a = *reinterpret_cast<const v8*>(wp + ((size_t)(ks * n16 + nt) * 64 + lane) * 8);
                acc[q] = MfmaT<T>::run(a, b, acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cb = (nt0 + q) * 16 + 4 * kg;
            if (vok && nt0 + q < n16 && cb < p.nout_p)
                st4<T>(gin + vc * p.out_cs + cb, make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]));
        }
    }
}

// weight gradient: dW[tap][ci 16][co 16] += X^T[ci][v] G_tap[v][co] over K-steps of 32 coarse voxels.
struct LpCtWgP {
    const void* x;
    const void* g;
    const float* scale;
    const float* shift;
    float* ws;
    int x_cs, cin_p, relu, g_cs, cout_p;
    int N, D, H, W;
    int64_t nvox;
};

constexpr int CTW_S = 32;                 // LDS bytes per voxel (16 channels)

template <class T>
__global__ __launch_bounds__(256) void lp_convt_wgrad_kernel(LpCtWgP p, int chunks_per_block) {
    typedef typename Vec<T>::v8 v8;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    __shared__ __attribute__((aligned(16))) unsigned char sX[128 * CTW_S];          // 4 K-steps of 32 voxels
    __shared__ __attribute__((aligned(16))) unsigned char sG[8 * 128 * CTW_S];      // [tap][voxel]
    __shared__ float sXf[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, i = lane & 15, q = i >> 2, pc = i & 3;
    const int nco = (p.cout_p + 15) >> 4;
    const int cit = blockIdx.y / nco, cot = blockIdx.y % nco;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* gr = reinterpret_cast<const T*>(p.g);
    const bool xf = p.scale != nullptr;
    if (tid < 32) {
        const int c = cit * 16 + (tid & 15);
        float v = (tid < 16) ? 1.f : 0.f;
        if (xf) v = (c < p.cin_p) ? ((tid < 16) ? p.scale[c] : p.shift[c]) : 0.f;
        sXf[tid] = v;
    }
    const int nchx = min(2, (p.cin_p - cit * 16) >> 3), nchg = min(2, (p.cout_p - cot * 16) >> 3);
    int ra[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) ra[r] = (wave * 32 + 8 * g4 + 4 * r + q) * CTW_S + 8 * pc;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t chunk = (int64_t)blockIdx.x * chunks_per_block;
    const int64_t nchunks = (p.nvox + 127) / 128;
    const int64_t chunk_end = min(nchunks, chunk + chunks_per_block);
    for (; chunk < chunk_end; ++chunk) {
        const int64_t v0 = chunk * 128;
        __syncthreads();
        {   // X: 128 voxels x 2 chunks = 256 items, one per thread
            const int vl = tid >> 1, c = tid & 1;
            const int64_t v = v0 + vl;
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            const bool ok = v < p.nvox && c < nchx;
            if (ok) raw = *reinterpret_cast<const uint4*>(x + v * p.x_cs + cit * 16 + c * 8);
            if (ok && xf) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&raw), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], sXf[c * 8 + j], sXf[16 + c * 8 + j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&raw) = __builtin_convertvector(o, v8);
            }
            *reinterpret_cast<uint4*>(sX + vl * CTW_S + c * 16) = raw;
        }
        // G: 8 taps x 128 voxels x 2 chunks = 2048 items, 8 per thread (all loads first)
        uint4 rg[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int it = tid + u * 256, c = it & 1, vl = (it >> 1) & 127, tap = it >> 8;
            const int64_t v = v0 + vl;
            rg[u] = make_uint4(0u, 0u, 0u, 0u);
            if (v < p.nvox && c < nchg) {
                const int wq = (int)(v % p.W);
                int64_t t = v / p.W;
                const int hq = (int)(t % p.H); t /= p.H;
                const int dq = (int)(t % p.D);
                const int n = (int)(t / p.D);
                const size_t fv = (((size_t)n * 2 * p.D + 2 * dq + (tap >> 2)) * 2 * p.H + 2 * hq + ((tap >> 1) & 1)) * 2 * p.W + 2 * wq + (tap & 1);
                rg[u] = *reinterpret_cast<const uint4*>(gr + fv * p.g_cs + cot * 16 + c * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int it = tid + u * 256, c = it & 1, vl = (it >> 1) & 127, tap = it >> 8;
            *reinterpret_cast<uint4*>(sG + (tap * 128 + vl) * CTW_S + c * 16) = rg[u];
        }
        __syncthreads();
        const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + ra[0]));
        const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + ra[1]));
        const s16x8 aa = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + t * 128 * CTW_S + ra[0]));
            const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + t * 128 * CTW_S + ra[1]));
            const s16x8 bb = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
            acc[t] = MfmaT<T>::run(*reinterpret_cast<const v8*>(&aa), *reinterpret_cast<const v8*>(&bb), acc[t]);
        }
    }
    // cross-wave sum -> one slab [8][16 ci][16 co] per block
    float* sS = reinterpret_cast<float*>(sG);
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* e = &sS[t * 256 + (4 * g4 + r) * 16 + i];
                    *e = (wv == 0) ? acc[t][r] : (*e + acc[t][r]);
                }
        }
    }
    __syncthreads();
    float* dst = p.ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2048;
    for (int e = tid; e < 2048; e += 256) dst[e] = sS[e];
}

// dw[ci][co][tap] (torch layout [Ci][Co][2][2][2]); imap: logical ci -> padded position.  16 lanes per output element
// split the gx slabs, fixed-order butterfly over them (deterministic).
__global__ __launch_bounds__(256) void lp_convt_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Ci, int Co,
                                                                   const int32_t* __restrict__ imap, int cout_p, int gx) {
    const int idx = blockIdx.x * 16 + (threadIdx.x >> 4), part = threadIdx.x & 15;
    const bool ok = idx < Ci * Co * 8;
    const int id = ok ? idx : 0;
    const int tap = id & 7, co = (id >> 3) % Co, ci = (id >> 3) / Co;
    const int pos = imap ? imap[ci] : ci;
    const int nco = (cout_p + 15) >> 4;
    const int pair = (pos >> 4) * nco + (co >> 4);
    const float* base = ws + (size_t)pair * gx * 2048 + tap * 256 + (pos & 15) * 16 + (co & 15);
    float s = 0.f;
    for (int k = part; k < gx; k += 16) s += base[(size_t)k * 2048];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (ok && part == 0) dw[idx] = s;
}

void ctw_grid(int64_t nvox, int pairs, int* gx, int* cpb) {
    const int64_t nchunks = (nvox + 127) / 128;
    int64_t g = 1024 / pairs;
    if (g < 16) g = 16;
    if (g > nchunks) g = nchunks;
    *cpb = (int)((nchunks + g - 1) / g);
    *gx = (int)((nchunks + *cpb - 1) / *cpb);
}

}  // namespace

// =================================================================== C ABI
extern "C" size_t ctu_lp_convt_packed_elems(int rin_p, int nout_p, int mode) {
    if (rin_p <= 0 || rin_p % 8 || nout_p <= 0 || nout_p % 8) return 0;
    const size_t n16 = (nout_p + 15) >> 4;
    return mode == 0 ? (size_t)8 * ((rin_p + 31) >> 5) * n16 * 512 : (size_t)2 * (rin_p >> 3) * n16 * 512;
}

extern "C" int ctu_lp_pack_convt_weight(int dtype, const float* w, void* wp, int Ci, int Co, const int32_t* cinv, int rin_p,
                                        int nout_p, int mode, void* stream) {
    CTU_REQUIRE(w && wp, "lp_pack_convt_weight: null pointer");
    CTU_REQUIRE(rin_p > 0 && rin_p % 8 == 0 && nout_p > 0 && nout_p % 8 == 0 && (mode == 0 || mode == 1),
                "lp_pack_convt_weight: rin_p=%d nout_p=%d mode=%d", rin_p, nout_p, mode);
    const size_t total = ctu_lp_convt_packed_elems(rin_p, nout_p, mode);
    const unsigned grid = (unsigned)ceil_div64((int64_t)total, 256);
    CTU_DISPATCH_LP(dtype, lp_pack_convt_w_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(w, (T*)wp, Ci, Co, cinv, rin_p, nout_p, mode));
    CTU_CHECK_LAUNCH("lp_pack_convt_weight");
    return CTU_OK;
}

extern "C" int ctu_lp_convt2_fwd(int dtype, const void* in, int in_cs, int rin_p, const float* in_scale, const float* in_shift,
                                 int in_relu, const void* wp, const float* bias, int nbias, void* out, int out_cs, int nout_p,
                                 int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(in && wp && out, "lp_convt2_fwd: null pointer");
    CTU_REQUIRE(rin_p > 0 && rin_p % 8 == 0 && rin_p <= 256 && nout_p > 0 && nout_p % 8 == 0, "lp_convt2_fwd: rin_p=%d nout_p=%d", rin_p, nout_p);
    CTU_REQUIRE(in_cs >= rin_p && in_cs % 8 == 0 && out_cs >= nout_p && out_cs % 4 == 0 && ((uintptr_t)in & 15) == 0 &&
                ((uintptr_t)out & 7) == 0, "lp_convt2_fwd: strides / alignment");
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lp_convt2_fwd: scale/shift come in pairs");
    LpCtP p{};
    p.in = in; p.wp = wp; p.out = out; p.scale = in_scale; p.shift = in_shift; p.bias = bias;
    p.in_cs = in_cs; p.rin_p = rin_p; p.relu = in_relu; p.out_cs = out_cs; p.nout_p = nout_p; p.nbias = bias ? nbias : 0;
    p.N = N; p.D = D; p.H = H; p.W = W; p.nvox = (int64_t)N * D * H * W;
    const unsigned grid = (unsigned)ceil_div64(p.nvox, 64);
    const int ksn = (rin_p + 31) >> 5;
    hipStream_t st = (hipStream_t)stream;
    CTU_DISPATCH_LP(dtype, {
        switch (ksn) {
            case 1: lp_convt_fwd_kernel<T, 1><<<grid, 256, 0, st>>>(p); break;
            case 2: lp_convt_fwd_kernel<T, 2><<<grid, 256, 0, st>>>(p); break;
            case 3: lp_convt_fwd_kernel<T, 3><<<grid, 256, 0, st>>>(p); break;
            case 4: lp_convt_fwd_kernel<T, 4><<<grid, 256, 0, st>>>(p); break;
            case 5: lp_convt_fwd_kernel<T, 5><<<grid, 256, 0, st>>>(p); break;
            case 6: lp_convt_fwd_kernel<T, 6><<<grid, 256, 0, st>>>(p); break;
            case 7: lp_convt_fwd_kernel<T, 7><<<grid, 256, 0, st>>>(p); break;
            default: lp_convt_fwd_kernel<T, 8><<<grid, 256, 0, st>>>(p); break;
        }
    });
    CTU_CHECK_LAUNCH("lp_convt2_fwd");
    return CTU_OK;
}

extern "C" int ctu_lp_convt2_bwd_data(int dtype, const void* gout, int g_cs, int rout_p, const void* wp, void* gin, int gin_cs,
                                      int nin_p, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(gout && wp && gin, "lp_convt2_bwd_data: null pointer");
    CTU_REQUIRE(rout_p > 0 && rout_p % 8 == 0 && nin_p > 0 && nin_p % 8 == 0, "lp_convt2_bwd_data: rout_p=%d nin_p=%d", rout_p, nin_p);
    CTU_REQUIRE(g_cs >= rout_p && g_cs % 8 == 0 && gin_cs >= nin_p && gin_cs % 4 == 0 && ((uintptr_t)gout & 15) == 0 &&
                ((uintptr_t)gin & 7) == 0, "lp_convt2_bwd_data: strides / alignment");
    LpCtP p{};
    p.in = gout; p.wp = wp; p.out = gin; p.in_cs = g_cs; p.rin_p = rout_p; p.out_cs = gin_cs; p.nout_p = nin_p;
    p.N = N; p.D = D; p.H = H; p.W = W; p.nvox = (int64_t)N * D * H * W;
    const unsigned grid = (unsigned)ceil_div64(p.nvox, 64);
    CTU_DISPATCH_LP(dtype, lp_convt_bwd_data_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(p));
    CTU_CHECK_LAUNCH("lp_convt2_bwd_data");
    return CTU_OK;
}

extern "C" size_t ctu_lp_convt2_wgrad_ws_floats(int N, int D, int H, int W, int cin_p, int cout_p) {
    const int pairs = ((cin_p + 15) >> 4) * ((cout_p + 15) >> 4);
    int gx, cpb;
    ctw_grid((int64_t)N * D * H * W, pairs, &gx, &cpb);
    return (size_t)gx * pairs * 2048;
}

extern "C" int ctu_lp_convt2_wgrad(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                   int in_relu, const void* gout, int g_cs, int cout_p, float* dw, int Ci, int Co,
                                   const int32_t* imap, float* ws, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(in && gout && dw && ws, "lp_convt2_wgrad: null pointer");
    CTU_REQUIRE(cin_p > 0 && cin_p % 8 == 0 && cout_p > 0 && cout_p % 8 == 0 && Ci <= cin_p && Co <= cout_p,
                "lp_convt2_wgrad: cin_p=%d cout_p=%d", cin_p, cout_p);
    CTU_REQUIRE(in_cs >= cin_p && in_cs % 8 == 0 && g_cs >= cout_p && g_cs % 8 == 0 && ((uintptr_t)in & 15) == 0 &&
                ((uintptr_t)gout & 15) == 0, "lp_convt2_wgrad: strides / alignment");
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lp_convt2_wgrad: scale/shift come in pairs");
    LpCtWgP p{};
    p.x = in; p.g = gout; p.scale = in_scale; p.shift = in_shift; p.ws = ws;
    p.x_cs = in_cs; p.cin_p = cin_p; p.relu = in_relu; p.g_cs = g_cs; p.cout_p = cout_p;
    p.N = N; p.D = D; p.H = H; p.W = W; p.nvox = (int64_t)N * D * H * W;
    const int pairs = ((cin_p + 15) >> 4) * ((cout_p + 15) >> 4);
    int gx, cpb;
    ctw_grid(p.nvox, pairs, &gx, &cpb);
    hipStream_t st = (hipStream_t)stream;
    CTU_DISPATCH_LP(dtype, lp_convt_wgrad_kernel<T><<<dim3(gx, pairs), 256, 0, st>>>(p, cpb));
    CTU_CHECK_LAUNCH("lp_convt2_wgrad");
    lp_convt_wgrad_reduce_kernel<<<ceil_div(Ci * Co * 8, 16), 256, 0, st>>>(ws, dw, Ci, Co, imap, cout_p, gx);
    CTU_CHECK_LAUNCH("lp_convt2_wgrad reduce");
    return CTU_OK;
}
