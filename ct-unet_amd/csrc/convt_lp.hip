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

// forward: one wave = 16 CTV coarse voxels (CTV column tiles of 16: 4 on large grids, 1 where that would leave CUs idle) x all
// taps x all out tiles; the voxel-side fragments of all
// KSN K-steps (ceil(rin_p / 32)) stay in registers, every weight fragment is loaded once per wave and used by 4 MFMAs
template <class T, int KSN, int CTV>
__global__ __launch_bounds__(256) void lp_convt_fwd_kernel(LpCtP p) {
    typedef typename Vec<T>::v8 v8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kg = lane >> 4;
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* wp = reinterpret_cast<const T*>(p.wp);
    T* out = reinterpret_cast<T*>(p.out);
    const int64_t v0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * CTV);
    v8 b[CTV][KSN];
    size_t fbase[CTV];                   // fine-grid voxel (2d, 2h, 2w) of this lane's coarse voxel in column tile ct
    bool vok[CTV];
    // all voxel-side loads first (branch-free: a load behind a branch is waited for before the next one is issued)
    uint4 rawb[CTV][KSN];
#pragma unroll
    for (int ct = 0; ct < CTV; ++ct) {
        const int64_t v = v0 + ct * 16 + m;
        vok[ct] = v < p.nvox;
        const int64_t vc = vok[ct] ? v : 0;
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
            const int c = ks * 32 + kg * 8;
            rawb[ct][ks] = *reinterpret_cast<const uint4*>(in + vc * p.in_cs + (c < p.rin_p ? c : 0));
        }
    }
    float scv[KSN][8], shv[KSN][8];
    if (p.scale) {
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
            const int c = ks * 32 + kg * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                scv[ks][j] = c < p.rin_p ? p.scale[c + j] : 0.f;
                shv[ks][j] = c < p.rin_p ? p.shift[c + j] : 0.f;
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CTV; ++ct) {
        const int64_t v = v0 + ct * 16 + m;
        const int64_t vc = vok[ct] ? v : 0;
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
            const int c = ks * 32 + kg * 8;
            const bool live = vok[ct] && c < p.rin_p;
            uint4 raw = rawb[ct][ks];
            if (p.scale) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&raw), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], scv[ks][j], shv[ks][j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&raw) = __builtin_convertvector(o, v8);
            }
            if (!live) raw = make_uint4(0u, 0u, 0u, 0u);
            b[ct][ks] = *reinterpret_cast<v8*>(&raw);
        }
        const int wq = (int)(vc % p.W);
        int64_t t = vc / p.W;
        const int hq = (int)(t % p.H); t /= p.H;
        const int dq = (int)(t % p.D);
        const int n = (int)(t / p.D);
        fbase[ct] = (((size_t)n * 2 * p.D + 2 * dq) * 2 * p.H + 2 * hq) * 2 * p.W + 2 * wq;
    }
    const int n16 = (p.nout_p + 15) >> 4;
    const size_t fH = (size_t)2 * p.W, fD = fH * 2 * p.H;
    for (int nt = blockIdx.y; nt < n16; nt += gridDim.y) {           // (small grids: one output tile per blockIdx.y)
        const int cb = nt * 16 + 4 * kg;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            bv.x = cb + 0 < p.nbias ? p.bias[cb + 0] : 0.f; bv.y = cb + 1 < p.nbias ? p.bias[cb + 1] : 0.f;
            bv.z = cb + 2 < p.nbias ? p.bias[cb + 2] : 0.f; bv.w = cb + 3 < p.nbias ? p.bias[cb + 3] : 0.f;
        }
        // weight fragments of a GROUP of taps are loaded together (<= 16 fragments in flight), then their MFMAs and stores
        constexpr int TG = KSN <= 2 ? 8 : (KSN <= 4 ? 4 : 2);
#pragma unroll
        for (int tg = 0; tg < 8; tg += TG) {
            v8 aw[TG][KSN];
#pragma unroll
            for (int u = 0; u < TG; ++u)
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks)
                    aw[u][ks] = *reinterpret_cast<const v8*>(wp + ((size_t)(((tg + u) * KSN + ks) * n16 + nt) * 64 + lane) * 8);
#pragma unroll
            for (int u = 0; u < TG; ++u) {
                const int tap = tg + u;
                f32x4 acc[CTV];
#pragma unroll
                for (int ct = 0; ct < CTV; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
                    for (int ct = 0; ct < CTV; ++ct) acc[ct] = MfmaT<T>::run(aw[u][ks], b[ct][ks], acc[ct]);
                const size_t toff = (tap >> 2) * fD + ((tap >> 1) & 1) * fH + (tap & 1);
#pragma unroll
                for (int ct = 0; ct < CTV; ++ct)
                    if (vok[ct] && cb < p.nout_p)
                        st4<T>(out + (fbase[ct] + toff) * p.out_cs + cb,
                               make_float4(acc[ct][0] + bv.x, acc[ct][1] + bv.y, acc[ct][2] + bv.z, acc[ct][3] + bv.w));
            }
        }
    }
}

// data gradient: gin[v][r] = sum_{tap, o} gout[2v + tap][o] w[r][o][tap]; K-steps = 2 * (rout_p / 8); one wave = 64 coarse
// voxels, 2 output tiles per pass over the gradient
template <class T, int CTV>
__global__ __launch_bounds__(256) void lp_convt_bwd_data_kernel(LpCtP p) {
    typedef typename Vec<T>::v8 v8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kg = lane >> 4;
    const T* g = reinterpret_cast<const T*>(p.in);          // fine-grid gradient, p.rin_p = padded Co
    const T* wp = reinterpret_cast<const T*>(p.wp);
    T* gin = reinterpret_cast<T*>(p.out);
    const int64_t v0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * CTV);
    size_t fbase[CTV];
    int64_t vcs[CTV];
    bool vok[CTV];
#pragma unroll
    for (int ct = 0; ct < CTV; ++ct) {
        const int64_t v = v0 + ct * 16 + m;
        vok[ct] = v < p.nvox;
        const int64_t vc = vok[ct] ? v : 0;
        vcs[ct] = vc;
        const int wq = (int)(vc % p.W);
        int64_t t = vc / p.W;
        const int hq = (int)(t % p.H); t /= p.H;
        const int dq = (int)(t % p.D);
        const int n = (int)(t / p.D);
        fbase[ct] = (((size_t)n * 2 * p.D + 2 * dq) * 2 * p.H + 2 * hq) * 2 * p.W + 2 * wq;
    }
    const size_t fH = (size_t)2 * p.W, fD = fH * 2 * p.H;
    const int nch = p.rin_p >> 3, ksn = 2 * nch, n16 = (p.nout_p + 15) >> 4;
    // (small grids: the output-tile passes are spread over blockIdx.y -- 64 blocks x 4 passes of 32 dependent K-steps was one
    //  54 us latency chain at 16^3 x 128 channels)
    for (int nt0 = 2 * blockIdx.y; nt0 < n16; nt0 += 2 * gridDim.y) {
        f32x4 acc[CTV][2];
#pragma unroll
        for (int ct = 0; ct < CTV; ++ct) { acc[ct][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[ct][1] = acc[ct][0]; }
        // K-step ks + 1's gradient and weight fragments are in flight while K-step ks multiplies
        auto load_ks = [&](int ks, uint4 (&rb)[CTV], v8 (&ra)[2]) {
            const int pr = 4 * ks + kg, tap = pr / nch, ch = pr % nch;
            const size_t toff = (tap >> 2) * fD + ((tap >> 1) & 1) * fH + (tap & 1);
#pragma unroll
            for (int ct = 0; ct < CTV; ++ct)
                rb[ct] = *reinterpret_cast<const uint4*>(g + (fbase[ct] + toff) * p.in_cs + ch * 8);     // vc is clamped: valid
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int nt = min(nt0 + q, n16 - 1);
                ra[q] = *reinterpret_cast<const v8*>(wp + ((size_t)(ks * n16 + nt) * 64 + lane) * 8);
            }
        };
        uint4 rb0[CTV], rb1[CTV];
        v8 ra0[2], ra1[2];
        load_ks(0, rb0, ra0);
        for (int ks = 0; ks < ksn; ks += 2) {
            load_ks(min(ks + 1, ksn - 1), rb1, ra1);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int ct = 0; ct < CTV; ++ct) {
                    uint4 raw = vok[ct] ? rb0[ct] : make_uint4(0u, 0u, 0u, 0u);
                    acc[ct][q] = MfmaT<T>::run(ra0[q], *reinterpret_cast<v8*>(&raw), acc[ct][q]);
                }
            load_ks(min(ks + 2, ksn - 1), rb0, ra0);
            if (ks + 1 < ksn) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int ct = 0; ct < CTV; ++ct) {
                        uint4 raw = vok[ct] ? rb1[ct] : make_uint4(0u, 0u, 0u, 0u);
                        acc[ct][q] = MfmaT<T>::run(ra1[q], *reinterpret_cast<v8*>(&raw), acc[ct][q]);
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cb = (nt0 + q) * 16 + 4 * kg;
#pragma unroll
            for (int ct = 0; ct < CTV; ++ct)
                if (vok[ct] && nt0 + q < n16 && cb < p.nout_p)
                    st4<T>(gin + vcs[ct] * p.out_cs + cb, make_float4(acc[ct][q][0], acc[ct][q][1], acc[ct][q][2], acc[ct][q][3]));
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

// Block (x = range of 128-voxel chunks, y = group of MT ci tiles x NTL co tiles): the gradient (8x the input's bytes) is
// read once per co-tile GROUP instead of once per (ci tile, co tile) pair.
template <class T, int MT, int NTL>
__global__ __launch_bounds__(256) void lp_convt_wgrad_kernel(LpCtWgP p, int chunks_per_block) {
    typedef typename Vec<T>::v8 v8;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    constexpr int SX = MT * 32, SG = NTL * 32;                      // LDS bytes per voxel of the X / G images
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sXf = reinterpret_cast<float*>(smem);                    // [2][MT * 16]
    unsigned char* sX = smem + 512;                                 // [128 voxels][MT * 16 ch]
    unsigned char* sG = sX + 128 * SX;                              // [8 taps][128 voxels][NTL * 16 ch]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, i = lane & 15, q = i >> 2, pc = i & 3;
    const int ncog = ((p.cout_p + 15) >> 4) / NTL + ((((p.cout_p + 15) >> 4) % NTL) ? 1 : 0);
    const int cig = blockIdx.y / ncog, cog = blockIdx.y % ncog;
    const int ci0 = cig * MT * 16, co0 = cog * NTL * 16;            // first channel of this block's groups
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* gr = reinterpret_cast<const T*>(p.g);
    const bool xf = p.scale != nullptr;
    if (tid < 2 * MT * 16) {
        const int which = tid / (MT * 16), c = ci0 + tid % (MT * 16);
        float v = which ? 0.f : 1.f;
        if (xf) v = (c < p.cin_p) ? (which ? p.shift[c] : p.scale[c]) : 0.f;
        sXf[tid] = v;
    }
    const int nchx = min(MT * 2, (p.cin_p - ci0) >> 3), nchg = min(NTL * 2, (p.cout_p - co0) >> 3);
    int rxa[2], rga[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int vl = wave * 32 + 8 * g4 + 4 * r + q;
        rxa[r] = vl * SX + 8 * pc;
        rga[r] = vl * SG + 8 * pc;
    }
    f32x4 acc[8][MT][NTL];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int a_ = 0; a_ < MT; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < NTL; ++b_) acc[t][a_][b_] = f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t chunk = (int64_t)blockIdx.x * chunks_per_block;
    const int64_t nchunks = (p.nvox + 127) / 128;
    const int64_t chunk_end = min(nchunks, chunk + chunks_per_block);
    // software pipeline over the 128-voxel chunks: the next chunk's loads are in flight while this one multiplies
    uint4 rxv[MT], rgv[8 * NTL];
    auto load_chunk = [&](int64_t ch) {
        const int64_t v0 = ch * 128;
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            const int it = tid + u * 256, c = it % (2 * MT), vl = it / (2 * MT);
            const int64_t v = v0 + vl;
            const int64_t vv = v < p.nvox ? v : p.nvox - 1;
            rxv[u] = *reinterpret_cast<const uint4*>(x + vv * p.x_cs + ci0 + (c < nchx ? c : 0) * 8);
            if (!(v < p.nvox && c < nchx)) rxv[u] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8 * NTL; ++u) {
            const int it = tid + u * 256, c = it % (2 * NTL), vl = (it / (2 * NTL)) & 127, tap = it / (2 * NTL * 128);
            const int64_t v = v0 + vl;
            const int64_t vv = v < p.nvox ? v : p.nvox - 1;
            const int wq = (int)(vv % p.W);
            int64_t t = vv / p.W;
            const int hq = (int)(t % p.H); t /= p.H;
            const int dq = (int)(t % p.D);
            const int n = (int)(t / p.D);
            const size_t fv = (((size_t)n * 2 * p.D + 2 * dq + (tap >> 2)) * 2 * p.H + 2 * hq + ((tap >> 1) & 1)) * 2 * p.W + 2 * wq + (tap & 1);
            rgv[u] = *reinterpret_cast<const uint4*>(gr + fv * p.g_cs + co0 + (c < nchg ? c : 0) * 8);
            if (!(v < p.nvox && c < nchg)) rgv[u] = make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if (chunk < chunk_end) load_chunk(chunk);
    for (; chunk < chunk_end; ++chunk) {
        const int64_t v0 = chunk * 128;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            const int it = tid + u * 256, c = it % (2 * MT), vl = it / (2 * MT);
            uint4 raw = rxv[u];
            if (xf && v0 + vl < p.nvox && c < nchx) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&raw), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], sXf[c * 8 + j], sXf[MT * 16 + c * 8 + j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&raw) = __builtin_convertvector(o, v8);
            }
            *reinterpret_cast<uint4*>(sX + vl * SX + c * 16) = raw;
        }
#pragma unroll
        for (int u = 0; u < 8 * NTL; ++u) {
            const int it = tid + u * 256, c = it % (2 * NTL), vl = (it / (2 * NTL)) & 127, tap = it / (2 * NTL * 128);
            *reinterpret_cast<uint4*>(sG + (tap * 128 + vl) * SG + c * 16) = rgv[u];
        }
        __syncthreads();
        if (chunk + 1 < chunk_end) load_chunk(chunk + 1);
        // all transposed fragment reads of the chunk first, then the MFMAs
        v8 afr[MT], bfr[8][NTL];
#pragma unroll
        for (int a_ = 0; a_ < MT; ++a_) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + rxa[0] + a_ * 32));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + rxa[1] + a_ * 32));
            const s16x8 aa = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            afr[a_] = *reinterpret_cast<const v8*>(&aa);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int b_ = 0; b_ < NTL; ++b_) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + t * 128 * SG + rga[0] + b_ * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + t * 128 * SG + rga[1] + b_ * 32));
                const s16x8 bb = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[t][b_] = *reinterpret_cast<const v8*>(&bb);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int b_ = 0; b_ < NTL; ++b_)
#pragma unroll
                for (int a_ = 0; a_ < MT; ++a_)
                    acc[t][a_][b_] = MfmaT<T>::run(afr[a_], bfr[t][b_], acc[t][a_][b_]);
    }
    // cross-wave sum -> slabs [pair][8][16 ci][16 co] of this block (pair = (ci tile, co tile))
    float* sS = reinterpret_cast<float*>(sG);
    const int nco = (p.cout_p + 15) >> 4, nci = (p.cin_p + 15) >> 4;
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int a_ = 0; a_ < MT; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < NTL; ++b_)
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* e = &sS[((a_ * NTL + b_) * 8 + t) * 256 + (4 * g4 + r) * 16 + i];
                            *e = (wv == 0) ? acc[t][a_][b_][r] : (*e + acc[t][a_][b_][r]);
                        }
        }
    }
    __syncthreads();
    for (int a_ = 0; a_ < MT; ++a_)
        for (int b_ = 0; b_ < NTL; ++b_) {
            const int cit = cig * MT + a_, cot = cog * NTL + b_;
            if (cit >= nci || cot >= nco) continue;
            float* dst = p.ws + ((size_t)(cit * nco + cot) * gridDim.x + blockIdx.x) * 2048;
            for (int e = tid; e < 2048; e += 256) dst[e] = sS[(a_ * NTL + b_) * 2048 + e];
        }
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
    const int ksn = (rin_p + 31) >> 5;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = p.nvox > 65536;                  // 64 voxels per wave only where that still fills the chip
    const unsigned gx = (unsigned)ceil_div64(p.nvox, wide ? 256 : 64);
    // deep levels: 64 blocks x 8 output tiles x 8 taps was one serial chain per wave -- one output tile per blockIdx.y there
    const dim3 grid(gx, gx < 256 ? (unsigned)((nout_p + 15) >> 4) : 1u);
#define CTU_CT_FWD(K)                                                                   \
    case K:                                                                             \
        if (wide) lp_convt_fwd_kernel<T, K, 4><<<grid, 256, 0, st>>>(p);                \
        else lp_convt_fwd_kernel<T, K, 1><<<grid, 256, 0, st>>>(p);                     \
        break;
    CTU_DISPATCH_LP(dtype, {
        switch (ksn) {
            CTU_CT_FWD(1) CTU_CT_FWD(2) CTU_CT_FWD(3) CTU_CT_FWD(4) CTU_CT_FWD(5) CTU_CT_FWD(6) CTU_CT_FWD(7)
            default:
                if (wide) lp_convt_fwd_kernel<T, 8, 4><<<grid, 256, 0, st>>>(p);
                else lp_convt_fwd_kernel<T, 8, 1><<<grid, 256, 0, st>>>(p);
        }
    });
#undef CTU_CT_FWD
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
    const bool wide = p.nvox > 65536;
    const unsigned gx = (unsigned)ceil_div64(p.nvox, wide ? 256 : 64);
    const int passes = (((nin_p + 15) >> 4) + 1) >> 1;              // pairs of 16-wide output tiles
    const dim3 grid(gx, (gx < 256 && passes > 1) ? passes : 1);
    CTU_DISPATCH_LP(dtype, {
        if (wide) lp_convt_bwd_data_kernel<T, 4><<<grid, 256, 0, (hipStream_t)stream>>>(p);
        else lp_convt_bwd_data_kernel<T, 1><<<grid, 256, 0, (hipStream_t)stream>>>(p);
    });
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
    const int nci = (cin_p + 15) >> 4, nco = (cout_p + 15) >> 4, pairs = nci * nco;
    int gx, cpb;
    ctw_grid(p.nvox, pairs, &gx, &cpb);
    hipStream_t st = (hipStream_t)stream;
    // tile groups per block: 2 x 2 while the accumulators (8 taps x MT x NTL) and the gradient image fit
#ifndef LP_CTW_MT4
#define LP_CTW_MT4 1
#endif
    // 4 x 1 groups where there are at least four input-channel tiles: the gradient (8x the input's bytes) is then read ONCE per
    // output-channel tile instead of once per pair of input-channel tiles, and a block's LDS image is 48 KB instead of 74
    const bool mt4 = LP_CTW_MT4 && nci >= 4;
    const int mt = mt4 ? 4 : (nci >= 2 ? 2 : 1), ntl = mt4 ? 1 : (nco >= 2 ? 2 : 1);
    const dim3 grid(gx, ceil_div(nci, mt) * ceil_div(nco, ntl));
    const size_t lds = 512 + 128 * (size_t)mt * 32 + 8 * 128 * (size_t)ntl * 32;
    CTU_DISPATCH_LP(dtype, {
        if (mt == 2 && ntl == 2) {
            // the dynamic-LDS limit is raised only for launches that need more than the default 64 KB, and only to what they need
    static size_t raised = 64 * 1024;
    if (lds > raised) {
        CTU_REQUIRE(hipFuncSetAttribute((const void*)lp_convt_wgrad_kernel<T, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                    "lp_convt2_wgrad: cannot raise the dynamic LDS limit");
        raised = lds;
    }
            lp_convt_wgrad_kernel<T, 2, 2><<<grid, 256, lds, st>>>(p, cpb);
        } else if (mt == 4) lp_convt_wgrad_kernel<T, 4, 1><<<grid, 256, lds, st>>>(p, cpb);
        else if (mt == 2) lp_convt_wgrad_kernel<T, 2, 1><<<grid, 256, lds, st>>>(p, cpb);
        else if (ntl == 2) lp_convt_wgrad_kernel<T, 1, 2><<<grid, 256, lds, st>>>(p, cpb);
        else lp_convt_wgrad_kernel<T, 1, 1><<<grid, 256, lds, st>>>(p, cpb);
    });
    CTU_CHECK_LAUNCH("lp_convt2_wgrad");
    lp_convt_wgrad_reduce_kernel<<<ceil_div(Ci * Co * 8, 16), 256, 0, st>>>(ws, dw, Ci, Co, imap, cout_p, gx);
    CTU_CHECK_LAUNCH("lp_convt2_wgrad reduce");
    return CTU_OK;
}
