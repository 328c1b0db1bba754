// First encoder convolution (C_in = 1 or 2, k = 3, C_out <= 8) for gfx950.
//
// nn.Conv3d(input_channels, i_size, 3, 1, 1, bias=False) at ctunet/pytorch/models.py:26 with
// input_channels in {1, 2} (models.py:175, 272-296).  K = 27 * C_in is far too short for an implicit-GEMM tile
// (the generic kernel pads C_in to 8: 8x the MFMAs, plus a padded channels-last copy of the input), and the layer is
// HBM-bound (12-21 FLOP/B), so:
//   forward ......... direct VALU convolution, reads the caller's NCDHW planes in place, writes channels-last + BN partials
//   input gradient .. direct VALU convolution of the 8-channel gradient with the flipped weights, writes NCDHW planes
//   weight gradient . MFMA with M = (tap, c_in) (27*C_in rows, padded to 32/64), N = c_out, K = voxels
// All three are persistent over 4x4x32 / 4x4x16 voxel boxes with the box staged in LDS.
#include <type_traits>
#include "common.h"
#include "bn_tail.h"

namespace {

struct FirstP {
    const float* x;        // NCDHW input planes [N][CIN][D][H][W]
    const float* w;        // torch weight [Co][CIN][27]
    const float* bias;
    const void* g;         // channels-last gradient / output side tensor (8 padded channels), element type T
    void* out;             // fwd: channels-last output (T); bwd-data: NCDHW dx (float)
    float* stats;
    float* ws;
    int g_cs, out_cs, Co, nbias;
    int N, D, H, W;
    int tiles_d, tiles_h, tiles_w, ntiles;
    // lazy BatchNorm + ReLU backward in first_wgrad_kernel<.., LZ = true> (as conv3d_wgrad_k3s_kernel's, conv3d.hip): g is the
    // gradient w.r.t. the ACTIVATED output, lz_y the raw output (geometry and stride of g), lz_coef bn_bwd_finalize's
    // [5][8] rows; the raw-output gradient goes to lz_out (geometry of g) for first_bwd_data_kernel
    const float* lz_y;
    float* lz_out;
    const float* lz_scale;
    const float* lz_shift;
    const float* lz_coef;
};

constexpr int TD = 4, TH = 4;

__device__ __forceinline__ void box_origin(const FirstP& p, int t, int TW, int& n, int& d0, int& h0, int& w0) {
    const int tx = t % p.tiles_w; t /= p.tiles_w;
    const int ty = t % p.tiles_h; t /= p.tiles_h;
    const int tz = t % p.tiles_d; t /= p.tiles_d;
    n = t; d0 = tz * TD; h0 = ty * TH; w0 = tx * TW;
}

// ------------------------------------------------------------------ forward: MFMA, weights in registers
// rows = (w-shift s, c_out) [16], columns = 16 voxel pairs (w = 2m, 2m+1), K = (kd, kh, kw' in 0..3, ci):
//   out[(d,h,2m+s), co] = sum x[(d+kd, h+kh, 2m+kw') - 1][ci] * W[co][ci][kd,kh,kw'-s]   (0 <= kw'-s <= 2)
// K = 36*CIN -> 9*CIN MFMAs per 32 voxels.  The per-lane weight fragments (9*CIN floats) and LDS offsets are
// computed once per block; each MFMA needs a single ds_read_b32.  Box 4x4x32, persistent, register prefetch.
template <int CIN, class T>
__global__ __launch_bounds__(256) void first_fwd_kernel(FirstP p, int tiles_per_block, ctu_bn_tail tail) {
    constexpr int TW = 32, HD = TD + 2, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW;
    constexpr int NIT = (HV * CIN + 255) / 256;
    constexpr int NS = 9 * CIN;                                     // K-steps of 4
    __shared__ float sX[HV * CIN];                                  // [ci][halo voxel]
    __shared__ float sRed[4 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    // weight fragment of this lane as the ROW operand: row = m = (s, co), k = 4*st + kq = (tap', ci)
    float wreg[NS];
    int aoff[NS];
    {
        const int s_ = m >> 3, co = m & 7;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int k = 4 * st + kq, tapp = k / CIN, ci = k % CIN;       // tap' = (kd*3+kh)*4 + kw'
            const int kwp = tapp & 3, r = tapp >> 2, kw = kwp - s_;
            wreg[st] = (co < p.Co && kw >= 0 && kw <= 2) ? p.w[((size_t)co * CIN + ci) * 27 + r * 3 + kw] : 0.f;
            aoff[st] = ci * HV + ((r / 3) * HH + r % 3) * HW + kwp;
        }
    }
    int abase[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) abase[mt] = (wave * HH + mt) * HW + 2 * m;       // td = wave, th = mt, pair m
    const int64_t plane = (int64_t)p.D * p.H * p.W;
    const int cq = (kq & 1) * 4, sh = kq >> 1;                       // this lane's 4 channels / shift in the epilogue
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
        bv.x = cq + 0 < p.nbias ? p.bias[cq + 0] : 0.f; bv.y = cq + 1 < p.nbias ? p.bias[cq + 1] : 0.f;
        bv.z = cq + 2 < p.nbias ? p.bias[cq + 2] : 0.f; bv.w = cq + 3 < p.nbias ? p.bias[cq + 3] : 0.f;
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(p.ntiles, tile + tiles_per_block);
    float vx[NIT];
    auto load = [&](int t) {
        int n, d0, h0, w0;
        box_origin(p, t, TW, n, d0, h0, w0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256;
            const int ci = e / HV, v = e % HV;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - 1, gh = h0 + ph - 1, gw = w0 + pw - 1;
            float val = 0.f;
            if (e < HV * CIN && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W)
                val = p.x[((int64_t)n * CIN + ci) * plane + ((int64_t)gd * p.H + gh) * p.W + gw];
            vx[it] = val;
        }
    };
    if (tile < tile_end) load(tile);
    while (tile < tile_end) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256;
            if (e < HV * CIN) sX[e] = vx[it];
        }
        __syncthreads();
        if (tile + 1 < tile_end) load(tile + 1);
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NS; ++st) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[st], sX[abase[mt] + aoff[st]], acc[mt], 0, 0, 0);
        }
        int n, d0, h0, w0;
        box_origin(p, tile, TW, n, d0, h0, w0);
        const int gd = d0 + wave, gw = w0 + 2 * m + sh;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int gh = h0 + mt;
            if (gd < p.D && gh < p.H && gw < p.W) {
                const float4 o = rnd4<T>(make_float4(acc[mt][0] + bv.x, acc[mt][1] + bv.y, acc[mt][2] + bv.z, acc[mt][3] + bv.w));
                st4<T>(reinterpret_cast<T*>(p.out) + ((((size_t)n * p.D + gd) * p.H + gh) * p.W + gw) * p.out_cs + cq, o);
                s1[0] += o.x; s1[1] += o.y; s1[2] += o.z; s1[3] += o.w;
                s2[0] += o.x * o.x; s2[1] += o.y * o.y; s2[2] += o.z * o.z; s2[3] += o.w * o.w;
            }
        }
        ++tile;
    }
    if (p.stats) {                                   // one BN partial row [2][8] per block
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a1 = s1[r], a2 = s2[r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
            a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32);          // the two shifts hold the same channels
            if (m == 0 && kq < 2) { sRed[wave * 16 + cq + r] = a1; sRed[wave * 16 + 8 + cq + r] = a2; }
        }
        __syncthreads();
        if (tid < 16) st_row(tail.counter != nullptr, p.stats + (size_t)blockIdx.x * 16 + tid, (sRed[tid] + sRed[16 + tid]) + (sRed[32 + tid] + sRed[48 + tid]));
        if (tail.counter) bn_fwd_tail(tail, p.stats, gridDim.x, 8, gridDim.x);
    }
}

// ------------------------------------------------------------------ input gradient: dx[v][ci] = sum g[v - tap + 1][co] W[co][ci][tap]
template <int CIN, class T>
__global__ __launch_bounds__(256) void first_bwd_data_kernel(FirstP p, int tiles_per_block) {
    constexpr int TW = 32, HD = TD + 2, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW, GS = 12;
    constexpr int NIT = (HV * 2 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float sG[HV * GS];          // haloed gradient box, 8 channels + 4 pad per voxel
    __shared__ __attribute__((aligned(16))) float sW[27 * CIN * 8];     // [tap (flipped)][ci][co], read as broadcasts
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * CIN * 8; i += 256) {
        const int co = i & 7, r = i >> 3, ci = r % CIN, tap = r / CIN;
        sW[i] = co < p.Co ? p.w[((size_t)co * CIN + ci) * 27 + (26 - tap)] : 0.f;
    }
    const int tw = tid & 15, th = (tid >> 4) & 3, td = tid >> 6;
    const int pbase = (td * HH + th) * HW + tw;
    const int half = tid & 1;
    const int64_t plane = (int64_t)p.D * p.H * p.W;
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(p.ntiles, tile + tiles_per_block);
    float4 vg[NIT];
    auto load = [&](int t) {
        int n, d0, h0, w0;
        box_origin(p, t, TW, n, d0, h0, w0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256, v = e >> 1;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - 1, gh = h0 + ph - 1, gw = w0 + pw - 1;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < HV * 2 && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W)
                val = ld4<T>(reinterpret_cast<const T*>(p.g) + ((((size_t)n * p.D + gd) * p.H + gh) * p.W + gw) * p.g_cs + half * 4);
            vg[it] = val;
        }
    };
    if (tile < tile_end) load(tile);
    while (tile < tile_end) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256;
            if (e < HV * 2) *reinterpret_cast<float4*>(&sG[(e >> 1) * GS + half * 4]) = vg[it];
        }
        __syncthreads();
        if (tile + 1 < tile_end) load(tile + 1);
        // 8 independent partial sums (one per gradient channel) per output and voxel: no long dependent FMA chain
        float q0[CIN][8], q1[CIN][8];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int c = 0; c < 8; ++c) { q0[ci][c] = 0.f; q1[ci][c] = 0.f; }
#pragma unroll 1
        for (int r9 = 0; r9 < 9; ++r9)                       // (kd, kh) rows stay rolled: full unrolling hoists every LDS load and spills
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int tap = r9 * 3 + kw;
            const int off = (((r9 / 3) * HH + r9 % 3) * HW + kw) * GS;
            float g0[8], g1[8];
            *reinterpret_cast<float4*>(&g0[0]) = *reinterpret_cast<const float4*>(&sG[pbase * GS + off]);
            *reinterpret_cast<float4*>(&g0[4]) = *reinterpret_cast<const float4*>(&sG[pbase * GS + off + 4]);
            *reinterpret_cast<float4*>(&g1[0]) = *reinterpret_cast<const float4*>(&sG[(pbase + 16) * GS + off]);
            *reinterpret_cast<float4*>(&g1[4]) = *reinterpret_cast<const float4*>(&sG[(pbase + 16) * GS + off + 4]);
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                float wv[8];
                *reinterpret_cast<float4*>(&wv[0]) = *reinterpret_cast<const float4*>(&sW[(tap * CIN + ci) * 8]);
                *reinterpret_cast<float4*>(&wv[4]) = *reinterpret_cast<const float4*>(&sW[(tap * CIN + ci) * 8 + 4]);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    q0[ci][c] = fmaf(g0[c], wv[c], q0[ci][c]);
                    q1[ci][c] = fmaf(g1[c], wv[c], q1[ci][c]);
                }
            }
        }
        float a0[CIN], a1[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            a0[ci] = ((q0[ci][0] + q0[ci][1]) + (q0[ci][2] + q0[ci][3])) + ((q0[ci][4] + q0[ci][5]) + (q0[ci][6] + q0[ci][7]));
            a1[ci] = ((q1[ci][0] + q1[ci][1]) + (q1[ci][2] + q1[ci][3])) + ((q1[ci][4] + q1[ci][5]) + (q1[ci][6] + q1[ci][7]));
        }
        int n, d0, h0, w0;
        box_origin(p, tile, TW, n, d0, h0, w0);
        const int gd = d0 + td, gh = h0 + th;
        if (gd < p.D && gh < p.H) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                float* dst = reinterpret_cast<float*>(p.out) + ((int64_t)n * CIN + ci) * plane + ((int64_t)gd * p.H + gh) * p.W;
                if (w0 + tw < p.W) dst[w0 + tw] = a0[ci];
                if (w0 + tw + 16 < p.W) dst[w0 + tw + 16] = a1[ci];
            }
        }
        ++tile;
    }
}

// ------------------------------------------------------------------ weight gradient: M = (tap, ci), N = co, K = voxels
template <int CIN, class T, bool LZ = false>
__global__ __launch_bounds__(256) void first_wgrad_kernel(FirstP p, int tiles_per_block) {
    constexpr int TW = 32, HD = TD + 2, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW, NV = TD * TH * TW;
    constexpr int MTN = (27 * CIN + 15) / 16;                  // M tiles
    constexpr int NIT = (HV * CIN + 255) / 256;
    constexpr int GIT = NV * 2 / 256;
    __shared__ float sX[HV * CIN];                             // [ci][halo voxel]
    __shared__ __attribute__((aligned(16))) float sG[NV * 8];
    __shared__ float sRed[4 * MTN * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int64_t plane = (int64_t)p.D * p.H * p.W;
    int rowoff[MTN];
    bool rowok[MTN];
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) {
        const int r = mt * 16 + i, tap = r / CIN, ci = r % CIN;
        rowok[mt] = r < 27 * CIN;
        const int tp = rowok[mt] ? tap : 0;
        rowoff[mt] = ci * HV + ((tp / 9) * HH + (tp / 3) % 3) * HW + tp % 3;
    }
    f32x4 acc[MTN];
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(p.ntiles, tile + tiles_per_block);
    const int half = tid & 1;
    float vx[NIT];
    float4 vg[GIT];
    float4 vy[LZ ? GIT : 1];
    float4 l_sc = make_float4(0.f, 0.f, 0.f, 0.f), l_sh = l_sc, l_k0 = l_sc, l_A = l_sc, l_B = l_sc;
    if constexpr (LZ) {
        l_sc = *reinterpret_cast<const float4*>(p.lz_scale + half * 4);
        l_sh = *reinterpret_cast<const float4*>(p.lz_shift + half * 4);
        l_k0 = *reinterpret_cast<const float4*>(p.lz_coef + half * 4);
        l_A = *reinterpret_cast<const float4*>(p.lz_coef + 24 + half * 4);
        l_B = *reinterpret_cast<const float4*>(p.lz_coef + 32 + half * 4);
    }
    auto load = [&](int t) {
        int n, d0, h0, w0;
        box_origin(p, t, TW, n, d0, h0, w0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256;
            const int ci = e / HV, v = e % HV;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - 1, gh = h0 + ph - 1, gw = w0 + pw - 1;
            float val = 0.f;
            if (e < HV * CIN && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W)
                val = p.x[((int64_t)n * CIN + ci) * plane + ((int64_t)gd * p.H + gh) * p.W + gw];
            vx[it] = val;
        }
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int e = tid + it * 256, v = e >> 1;
            const int tw_ = v % TW, th_ = (v / TW) % TH, td_ = v / (TW * TH);
            const int gd = d0 + td_, gh = h0 + th_, gw = w0 + tw_;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gd < p.D && gh < p.H && gw < p.W) {
                const size_t idx = ((((size_t)n * p.D + gd) * p.H + gh) * p.W + gw) * p.g_cs + half * 4;
                val = ld4<T>(reinterpret_cast<const T*>(p.g) + idx);
                if constexpr (LZ) vy[it] = *reinterpret_cast<const float4*>(p.lz_y + idx);
            }
            vg[it] = val;
        }
    };
    if (tile < tile_end) load(tile);
    while (tile < tile_end) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256;
            if (e < HV * CIN) sX[e] = vx[it];
        }
        if constexpr (LZ) {
            // the registers hold box `tile`: form its raw-output gradient (out-of-volume items stay 0) and write it out
            int n, d0, h0, w0;
            box_origin(p, tile, TW, n, d0, h0, w0);
#pragma unroll
            for (int it = 0; it < GIT; ++it) {
                const int e = tid + it * 256, v = e >> 1;
                const int tw_ = v % TW, th_ = (v / TW) % TH, td_ = v / (TW * TH);
                const int gd = d0 + td_, gh = h0 + th_, gw = w0 + tw_;
                if (gd < p.D && gh < p.H && gw < p.W) {
                    vg[it] = bn_bwd_lazy4(vg[it], vy[it], l_sc, l_sh, l_k0, l_A, l_B);
                    *reinterpret_cast<float4*>(p.lz_out + ((((size_t)n * p.D + gd) * p.H + gh) * p.W + gw) * p.g_cs + half * 4) = vg[it];
                }
            }
        }
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int e = tid + it * 256;
            *reinterpret_cast<float4*>(&sG[(e >> 1) * 8 + half * 4]) = vg[it];
        }
        __syncthreads();
        if (tile + 1 < tile_end) load(tile + 1);
#pragma unroll 4
        for (int ks = 0; ks < NV / 16; ++ks) {
            const int th_ = ks / (TW / 4), tw_ = (ks % (TW / 4)) * 4 + kq;             // td = wave
            const float b = (i < 8) ? sG[((wave * TH + th_) * TW + tw_) * 8 + i] : 0.f;
            const int pb = (wave * HH + th_) * HW + tw_;
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) {
                const float a = rowok[mt] ? sX[pb + rowoff[mt]] : 0.f;
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[mt], 0, 0, 0);
            }
        }
        ++tile;
    }
    // cross-wave reduction -> one slab [MTN][16 rows][16 co] per block
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sRed[(wave * MTN + mt) * 256 + (kq * 4 + r) * 16 + i] = acc[mt][r];
    __syncthreads();
    float* dst = p.ws + (size_t)blockIdx.x * (MTN * 256);
    for (int e = tid; e < MTN * 256; e += 256)
        dst[e] = (sRed[e] + sRed[MTN * 256 + e]) + (sRed[2 * MTN * 256 + e] + sRed[3 * MTN * 256 + e]);
}

template <int CIN>
__global__ __launch_bounds__(1024) void first_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Co,
                                                                 int gx) {
    constexpr int MTN = (27 * CIN + 15) / 16;
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;          // 1024 threads: 16 slab groups x 64 elements
    const int el = blockIdx.x * 64 + e;
    float s = 0.f;
    if (el < MTN * 256) {
        // 4 independent chains per thread (coalesced 256-byte rows of 4 slabs in flight), combined in a fixed order
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = part;
        for (; k + 48 < gx; k += 64) {
            s0 += ws[(size_t)k * (MTN * 256) + el];
            s1 += ws[(size_t)(k + 16) * (MTN * 256) + el];
            s2 += ws[(size_t)(k + 32) * (MTN * 256) + el];
            s3 += ws[(size_t)(k + 48) * (MTN * 256) + el];
        }
        for (; k < gx; k += 16) s0 += ws[(size_t)k * (MTN * 256) + el];
        s = (s0 + s1) + (s2 + s3);
    }
    red[part][e] = s;
    __syncthreads();
    if (part != 0 || el >= MTN * 256) return;
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][e];
    const int r = el >> 4, co = el & 15, tap = r / CIN, ci = r % CIN;
    if (r < 27 * CIN && co < Co) dw[((size_t)co * CIN + ci) * 27 + tap] = tot;
}

void grid_for(int ntiles, int per_cu, int* gx, int* tpb) {
    int g = 256 * per_cu;
    if (g > ntiles) g = ntiles;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

int fill(FirstP& p, int N, int D, int H, int W, int TW) {
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.tiles_d = ceil_div(D, TD); p.tiles_h = ceil_div(H, TH); p.tiles_w = ceil_div(W, TW);
    p.ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
    return p.ntiles;
}

}  // namespace

// =================================================================== C ABI
extern "C" int ctu_conv3d_first_supported(int k, int cin, int nout_p, int W) {
    return (k == 3 && (cin == 1 || cin == 2) && nout_p == 8 && W >= 16) ? 1 : 0;
}

extern "C" int ctu_conv3d_first_num_blocks(int N, int D, int H, int W) {
    FirstP p;
    int gx, tpb;
    grid_for(fill(p, N, D, H, W, 32), 4, &gx, &tpb);
    return gx;
}

namespace {

template <class T>
int first_fwd_impl(const float* x, int cin, const float* w, const float* bias, int nbias, T* out, int out_cs, int Co, float* stats,
                   int N, int D, int H, int W, const ctu_bn_tail* tail, void* stream) {
    CTU_REQUIRE(!tail || (stats && tail->counter && tail->gamma && tail->beta && tail->scale && tail->shift && tail->mean &&
                          tail->invstd && tail->C > 0 && tail->C <= 8 && tail->count > 0),
                "conv3d_first_fwd: incomplete BatchNorm tail");
    CTU_REQUIRE(x && w && out, "conv3d_first_fwd: null pointer");
    CTU_REQUIRE((cin == 1 || cin == 2) && Co >= 1 && Co <= 8, "conv3d_first_fwd: cin=%d Co=%d unsupported", cin, Co);
    CTU_REQUIRE(out_cs >= 8 && out_cs % 4 == 0 && ((uintptr_t)out & (4 * sizeof(T) - 1)) == 0, "conv3d_first_fwd: output slice alignment");
    FirstP p{};
    p.x = x; p.w = w; p.bias = bias; p.nbias = bias ? nbias : 0; p.out = out; p.out_cs = out_cs; p.Co = Co; p.stats = stats;
    int gx, tpb;
    grid_for(fill(p, N, D, H, W, 32), 4, &gx, &tpb);
    if (cin == 1) first_fwd_kernel<1, T><<<gx, 256, 0, (hipStream_t)stream>>>(p, tpb, tail_or_off(tail));
    else first_fwd_kernel<2, T><<<gx, 256, 0, (hipStream_t)stream>>>(p, tpb, tail_or_off(tail));
    CTU_CHECK_LAUNCH("conv3d_first_fwd");
    return CTU_OK;
}

template <class T>
int first_bwd_data_impl(const T* g, int g_cs, const float* w, int cin, int Co, float* dx, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(g && w && dx, "conv3d_first_bwd_data: null pointer");
    CTU_REQUIRE((cin == 1 || cin == 2) && Co >= 1 && Co <= 8 && g_cs >= 8 && g_cs % 4 == 0 && ((uintptr_t)g & (4 * sizeof(T) - 1)) == 0,
                "conv3d_first_bwd_data: bad argument");
    FirstP p{};
    p.g = g; p.g_cs = g_cs; p.w = w; p.out = dx; p.Co = Co;
    int gx, tpb;
    grid_for(fill(p, N, D, H, W, 32), 3, &gx, &tpb);
    if (cin == 1) first_bwd_data_kernel<1, T><<<gx, 256, 0, (hipStream_t)stream>>>(p, tpb);
    else first_bwd_data_kernel<2, T><<<gx, 256, 0, (hipStream_t)stream>>>(p, tpb);
    CTU_CHECK_LAUNCH("conv3d_first_bwd_data");
    return CTU_OK;
}

struct FirstLazy { const float* y; const float* scale; const float* shift; const float* coef; float* out; };

template <class T>
int first_wgrad_impl(const float* x, int cin, const T* g, int g_cs, float* dw, int Co, float* ws, int N, int D, int H, int W,
                     const FirstLazy* lz, void* stream) {
    CTU_REQUIRE(x && g && dw && ws, "conv3d_first_wgrad: null pointer");
    CTU_REQUIRE((cin == 1 || cin == 2) && Co >= 1 && Co <= 8 && g_cs >= 8 && g_cs % 4 == 0 && ((uintptr_t)g & (4 * sizeof(T) - 1)) == 0,
                "conv3d_first_wgrad: bad argument");
    FirstP p{};
    p.x = x; p.g = g; p.g_cs = g_cs; p.ws = ws; p.Co = Co;
    if (lz) { p.lz_y = lz->y; p.lz_out = lz->out; p.lz_scale = lz->scale; p.lz_shift = lz->shift; p.lz_coef = lz->coef; }
    int gx, tpb;
    grid_for(fill(p, N, D, H, W, 32), 5, &gx, &tpb);
    hipStream_t st = (hipStream_t)stream;
    if (cin == 1) {
        if constexpr (std::is_same<T, float>::value) {
            if (lz) first_wgrad_kernel<1, T, true><<<gx, 256, 0, st>>>(p, tpb);
            else first_wgrad_kernel<1, T><<<gx, 256, 0, st>>>(p, tpb);
        } else first_wgrad_kernel<1, T><<<gx, 256, 0, st>>>(p, tpb);
        first_wgrad_reduce_kernel<1><<<ceil_div(2 * 256, 64), 1024, 0, st>>>(ws, dw, Co, gx);
    } else {
        if constexpr (std::is_same<T, float>::value) {
            if (lz) first_wgrad_kernel<2, T, true><<<gx, 256, 0, st>>>(p, tpb);
            else first_wgrad_kernel<2, T><<<gx, 256, 0, st>>>(p, tpb);
        } else first_wgrad_kernel<2, T><<<gx, 256, 0, st>>>(p, tpb);
        first_wgrad_reduce_kernel<2><<<ceil_div(4 * 256, 64), 1024, 0, st>>>(ws, dw, Co, gx);
    }
    CTU_CHECK_LAUNCH("conv3d_first_wgrad");
    return CTU_OK;
}

}  // namespace

extern "C" int ctu_conv3d_first_fwd(const float* x, int cin, const float* w, const float* bias, int nbias, float* out,
                                    int out_cs, int Co, float* stats, int N, int D, int H, int W, const ctu_bn_tail* tail, void* stream) {
    return first_fwd_impl<float>(x, cin, w, bias, nbias, out, out_cs, Co, stats, N, D, H, W, tail, stream);
}
extern "C" int ctu_lp_conv3d_first_fwd(int dtype, const float* x, int cin, const float* w, const float* bias, int nbias, void* out,
                                       int out_cs, int Co, float* stats, int N, int D, int H, int W, const ctu_bn_tail* tail, void* stream) {
    CTU_DISPATCH_LP(dtype, return first_fwd_impl<T>(x, cin, w, bias, nbias, (T*)out, out_cs, Co, stats, N, D, H, W, tail, stream));
}

extern "C" int ctu_conv3d_first_bwd_data(const float* g, int g_cs, const float* w, int cin, int Co, float* dx, int N, int D,
                                         int H, int W, void* stream) {
    return first_bwd_data_impl<float>(g, g_cs, w, cin, Co, dx, N, D, H, W, stream);
}
extern "C" int ctu_lp_conv3d_first_bwd_data(int dtype, const void* g, int g_cs, const float* w, int cin, int Co, float* dx, int N,
                                            int D, int H, int W, void* stream) {
    CTU_DISPATCH_LP(dtype, return first_bwd_data_impl<T>((const T*)g, g_cs, w, cin, Co, dx, N, D, H, W, stream));
}

extern "C" size_t ctu_conv3d_first_wgrad_ws_floats(int N, int D, int H, int W, int cin) {
    FirstP p;
    int gx, tpb;
    grid_for(fill(p, N, D, H, W, 32), 5, &gx, &tpb);
    return (size_t)gx * ((27 * cin + 15) / 16) * 256;
}

extern "C" int ctu_conv3d_first_wgrad(const float* x, int cin, const float* g, int g_cs, float* dw, int Co, float* ws, int N,
                                      int D, int H, int W, void* stream) {
    return first_wgrad_impl<float>(x, cin, g, g_cs, dw, Co, ws, N, D, H, W, nullptr, stream);
}
extern "C" int ctu_conv3d_first_wgrad_bn(const float* x, int cin, const float* ga, int g_cs, const float* y, const float* bn_scale,
                                         const float* bn_shift, const float* coef, float* gy_out, float* dw, int Co, float* ws,
                                         int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(y && bn_scale && bn_shift && coef && gy_out && gy_out != ga, "conv3d_first_wgrad_bn: null pointer / gy_out aliases ga");
    CTU_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)gy_out & 15) == 0, "conv3d_first_wgrad_bn: 16-byte alignment");
    const FirstLazy lz = {y, bn_scale, bn_shift, coef, gy_out};
    return first_wgrad_impl<float>(x, cin, ga, g_cs, dw, Co, ws, N, D, H, W, &lz, stream);
}
extern "C" int ctu_lp_conv3d_first_wgrad(int dtype, const float* x, int cin, const void* g, int g_cs, float* dw, int Co, float* ws,
                                         int N, int D, int H, int W, void* stream) {
    CTU_DISPATCH_LP(dtype, return first_wgrad_impl<T>(x, cin, (const T*)g, g_cs, dw, Co, ws, N, D, H, W, nullptr, stream));
}
