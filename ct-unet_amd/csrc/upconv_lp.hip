// Reduced-precision (bf16 / fp16 storage, fp32 accumulate) fused decoder up-convolution for gfx950:
//   ConvTranspose3d(C, C, k=2, s=2, bias) -> Conv3d(C, C_out <= 8, k=3, p=1, no bias)      (ctunet/pytorch/models.py:37-38)
// as ONE kernel per direction on the coarse grid, on v_mfma_f32_16x16x32_{bf16,f16} -- the 16-bit twins of upconv_fused.hip
// (same algebra: per output parity p the pair is a 2x2x2 convolution of the coarse input with composite weights W_eff[p][d],
// the bias differs only on the volume faces).  The composite weights are built in fp32 by ctu_upconv_fused_pack and rounded
// ONCE into 16-bit MFMA fragments here; activations and gradients are 16-bit in HBM and LDS, accumulation is fp32.
//
// With 16-bit operands the fused pair is HBM-bound (level 0 of UNet() at 128^3: 16 MB in, 33 MB out, 0.27 GF of matrix work
// per microsecond of HBM time), so the kernels are built around bytes: one block per CU walks a contiguous range of 4x4x16
// coarse boxes, the next box's loads are in flight in registers under the current box's MFMAs, every fragment read is one
// ds_read_b128 at a compile-time offset from a per-lane base (no offset tables: a K-step = the 32 channels of ONE tap, the
// lane's K-quarter picks the 8-channel chunk), the composite weights of a 32-channel stage live in LDS for the whole block.
//
// forward  (8 padded output channels): columns = 16 coarse voxels along w, rows = (w-parity, c_out) -- the PW tile of
//   upconv_fused.hip -- 4 (p_d, p_h) parities x 2 x 2 x 3 taps = 48 MFMAs per column tile and stage; the 6 halo rows of one
//   (d, w) tap offset are read once and feed the 3 h offsets of the 4 column tiles (54 + 48 fragment reads per 192 MFMAs).
// data gradient: the adjoint is a STRIDE-2 convolution of the fine-grid gradient with a 4x4x4 kernel,
//   dX[j] = sum_{f in {-1,0,1,2}^3} g[2j + f] V[f],  V[f] = W_eff[p(f)][d(f)]^T  (per axis f = -1,0,1,2 <-> (p,d) = (1,1),(0,1),
//   (1,0),(0,0)); a K-step = the 4 w offsets x 8 channels of one (f_d, f_h) row: 16 K-steps, rows = input channels.
#include <type_traits>
#include "common.h"
#include "bn_tail.h"

namespace {

template <class T> struct UMfma;
template <> struct UMfma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct UMfma<f16_t> {
    static __device__ __forceinline__ f32x4 run(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int UL_S = 80;                  // LDS bytes per coarse halo voxel of a 32-channel stage (64 data + 16 pad)
constexpr int UL_TD = 4, UL_TH = 4, UL_BW = 16;
constexpr int UL_HD = 6, UL_HH = 6, UL_HW = 18, UL_HV = UL_HD * UL_HH * UL_HW;

// ------------------------------------------------------------------------------------------------ weight fragments
// forward: wf[stage][par4 = pz*2+py][tap12 = (dz*2+dy)*3+dxx][lane][8]: element j of lane l = A[row l&15 = (p_w, co)][k = channel
// 8 (l>>4) + j of the stage], gathered from ctu_upconv_fused_pack's fp32 PW packing wpw[chunk][par4][tap12][kq][n][jj]
template <class T>
__global__ void lp_upconv_pack_fwd_kernel(const float* __restrict__ wpw, T* __restrict__ wf, int nstage) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nstage * 48 * 512) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, m = lane & 15, kg = lane >> 4;
    const int r = idx >> 9, pt = r % 48, s = r / 48;
    const int c = s * 4 + kg;
    wf[idx] = (T)wpw[((size_t)c * 48 + pt) * 128 + (j >> 1) * 32 + m * 2 + (j & 1)];
}

// data gradient: wb[ks = fd*4+fh][n16][lane][8]: element j of lane l = V[f = (fd, fh, l>>4)][co = j][ci = n16*16 + (l&15)]
//   = W_eff[par(f)][tap(f)][ci][co], from the fp32 standard packing wp[chunk][par 8][tap 8][1][kq][n][jj] (nout_p = 8)
template <class T>
__global__ void lp_upconv_pack_bwd_kernel(const float* __restrict__ wp, T* __restrict__ wb, int cin_p) {
    const int n16 = cin_p >> 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 16 * n16 * 512) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, m = lane & 15, fw = lane >> 4;
    const int r = idx >> 9, nt = r % n16, ks = r / n16, fd = ks >> 2, fh = ks & 3;
    // per axis: f index 0..3 (offset -1..2) -> parity bit 1 - (f & 1), sub-cube tap bit 1 - (f >> 1)
    const int par = ((1 - (fd & 1)) << 2) | ((1 - (fh & 1)) << 1) | (1 - (fw & 1));
    const int tap = ((1 - (fd >> 1)) << 2) | ((1 - (fh >> 1)) << 1) | (1 - (fw >> 1));
    const int ci = nt * 16 + m, c = ci >> 3, r8 = ci & 7;
    wb[idx] = (T)wp[((size_t)(c * 8 + par) * 8 + tap) * 128 + (r8 >> 1) * 32 + j * 2 + (r8 & 1)];
}

// ------------------------------------------------------------------------------------------------ forward
// the 48 (parity, tap) pairs of a stage in the order the forward kernel walks them: tap offset nzx = nz * 3 + nx (d and w
// offsets in the halo), then the h offset ny, then the (p_d, p_h) parities with pz + dz = nz, py + dy = ny
struct UlStep { int nzx, ny, par4, tap; };
__host__ __device__ constexpr UlStep ul_step(int t) {
    int k = 0;
    for (int nzx = 0; nzx < 9; ++nzx)
        for (int ny = 0; ny < 3; ++ny)
            for (int pz = 0; pz < 2; ++pz)
                for (int py = 0; py < 2; ++py) {
                    const int nz = nzx / 3, nx = nzx % 3, dz = nz - pz, dy = ny - py;
                    if (dz < 0 || dz > 1 || dy < 0 || dy > 1) continue;
                    if (k == t) return UlStep{nzx, ny, pz * 2 + py, (dz * 2 + dy) * 3 + nx};
                    ++k;
                }
    return UlStep{0, 0, 0, 0};
}

template <int I, int N, class F>
__device__ __forceinline__ void ul_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ul_static_for<I + 1, N>(f);
    }
}

struct UlFwdP {
    const void* in;           // coarse [N, D, H, W, in_cs], raw (lazy BatchNorm + ReLU through scale / shift)
    const void* wf;           // lp_upconv_pack_fwd_kernel's fragments
    void* out;                // fine [N, 2D, 2H, 2W, out_cs], 8 padded channels, raw
    const float* scale;
    const float* shift;
    const float* beff;        // [27][8] fp32 border-class biases (ctu_upconv_fused_pack)
    float* stats;             // [gridDim.x][2][8]
    int in_cs, cin_p, relu, out_cs;
    int N, D, H, W, tiles_d, tiles_h, tiles_w;
};

constexpr int UL_FWD_NI = (UL_HV * 4 + 255) / 256;          // 16-byte staging items per thread and stage
constexpr size_t UL_FWD_LDS = 512 + (size_t)UL_HV * UL_S + 48 * 1024;

template <class T>
__global__ __launch_bounds__(256, 1) void lp_upconv_fwd_kernel(UlFwdP p, int ntiles, int tiles_per_block) {
    typedef typename Vec<T>::v8 v8;
    constexpr int HH = UL_HH, HW = UL_HW, HV = UL_HV, S = UL_S, NI = UL_FWD_NI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sXf = reinterpret_cast<float*>(smem);                    // [2][32] scale / shift of the stage
    float* sRed = reinterpret_cast<float*>(smem + 256);             // [4 waves][2][8]
    unsigned char* sIn = smem + 512;
    unsigned char* sW = sIn + (size_t)HV * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kg = lane >> 4;
    const T* in = reinterpret_cast<const T*>(p.in);
    T* out = reinterpret_cast<T*>(p.out);
    const int ns = p.cin_p >> 5;
    const bool xf = p.scale != nullptr;

    // per-thread staging items: byte offset relative to the halo origin (+ 64 bytes per stage), LDS destination
    unsigned ioff[NI];
    unsigned live = 0;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + u * 256;
        const bool lv = i < HV * 4;
        const int v = lv ? (i >> 2) : 0, c = i & 3;
        const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
        ioff[u] = (unsigned)((((pd * p.H + ph) * p.W + pw) * p.in_cs + c * 8) * (int)sizeof(T));
        live |= lv ? (1u << u) : 0u;
    }
    uint4 raw[NI];
    unsigned okb = 0;
    auto box_of = [&](int t, int& n, int& d0, int& h0, int& w0) {
        const int tx = t % p.tiles_w; t /= p.tiles_w;
        const int ty = t % p.tiles_h; t /= p.tiles_h;
        const int tz = t % p.tiles_d;
        n = t / p.tiles_d; d0 = tz * UL_TD; h0 = ty * UL_TH; w0 = tx * UL_BW;
    };
    auto load_box = [&](int t, int st) {
        int n, d0, h0, w0;
        box_of(t, n, d0, h0, w0);
        if (d0 >= 1 && d0 + UL_TD + 1 <= p.D && h0 >= 1 && h0 + UL_TH + 1 <= p.H && w0 >= 1 && w0 + UL_BW + 1 <= p.W) {     // uniform
            const char* base = reinterpret_cast<const char*>(in + ((((size_t)n * p.D + d0 - 1) * p.H + h0 - 1) * p.W + w0 - 1) * p.in_cs + st * 32);
#pragma unroll
            for (int u = 0; u < NI; ++u) raw[u] = *reinterpret_cast<const uint4*>(base + (((live >> u) & 1u) ? ioff[u] : 0u));
            okb = live;
            return;
        }
        okb = 0;
#pragma unroll
        for (int u = 0; u < NI; ++u) {                              // border / ragged box: clamped, branch-free
            const int i = tid + u * 256;
            const int v = min(i >> 2, HV - 1), c = i & 3;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - 1, gh = h0 + ph - 1, gw = w0 + pw - 1;
            const bool ok = ((live >> u) & 1u) && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            const int cd = min(max(gd, 0), p.D - 1), chh = min(max(gh, 0), p.H - 1), cw = min(max(gw, 0), p.W - 1);
            raw[u] = *reinterpret_cast<const uint4*>(in + ((((size_t)n * p.D + cd) * p.H + chh) * p.W + cw) * p.in_cs + st * 32 + c * 8);
            okb |= ok ? (1u << u) : 0u;
        }
    };
    auto load_weights = [&](int st) {                               // 48 fragments of 1 KB -> sW (12 pieces per thread)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // (uint4 arrays end up as stack objects)
        const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.wf) + (size_t)st * 48 * 512);
        u32x4 w8[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) w8[u] = src[tid + u * 256];
#pragma unroll
        for (int u = 0; u < 12; ++u) *reinterpret_cast<u32x4*>(sW + (size_t)(tid + u * 256) * 16) = w8[u];
    };

    f32x4 acc[4][4];                                                // [par4][column tile = row th]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[q][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const float4 b0 = *reinterpret_cast<const float4*>(p.beff + (kg & 1) * 4);          // class 0: interior fine voxels
    const int lbase = m * S + kg * 16;                              // this lane's (voxel, chunk) inside a halo row

    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(ntiles, tile + tiles_per_block);
    if (tile >= tile_end) return;
    int st = 0;
    load_box(tile, 0);
    if (ns == 1) load_weights(0);
    while (true) {
        __syncthreads();                                            // the previous step's readers are done with sIn / sW
        if (xf && tid < 64) {
            const int c = st * 32 + (tid & 31);
            sXf[tid] = (tid < 32) ? p.scale[c] : p.shift[c];
        }
        if (ns > 1) load_weights(st);
        if (xf) __syncthreads();                                    // sXf visible
        // every item of a thread is the same 8-channel chunk (256 % 4 == 0): its BatchNorm vectors go to registers once per
        // stage instead of 16 LDS reads per item
        float xs[8], xh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { xs[j] = xf ? sXf[(tid & 3) * 8 + j] : 1.f; xh[j] = xf ? sXf[32 + (tid & 3) * 8 + j] : 0.f; }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            if (!((live >> u) & 1u)) continue;
            uint4 r = raw[u];
            const int i = tid + u * 256;
            if (!((okb >> u) & 1u)) r = make_uint4(0u, 0u, 0u, 0u);
            else if (xf) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&r), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], xs[j], xh[j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&r) = __builtin_convertvector(o, v8);
            }
            *reinterpret_cast<uint4*>(sIn + (i >> 2) * S + (i & 3) * 16) = r;
        }
        __syncthreads();
        // ---- prefetch the next (box, stage) while this one computes
        int ntile = tile, nst = st + 1;
        if (nst == ns) { nst = 0; ntile = tile + 1; }
        const bool has_next = ntile < tile_end;
        if (has_next) load_box(ntile, nst);
        // ---- MFMAs: for each (d, w) tap offset (nz, nx) the 6 halo rows feed the 3 h offsets ny of the 4 column tiles; a
        // (parity, tap) pair belongs to (nz, ny) = (pz + dz, py + dy), its w tap is nx.  The 48 pairs form one flat sequence
        // (ul_step): pair t's weight fragment is read two pairs ahead, a tap offset's 6 rows one offset ahead, the order
        // read / 4 MFMAs pinned (one wave per SIMD: nothing else hides an LDS latency)
        {
            const unsigned char* bIn = sIn + wave * HH * HW * S + lbase;
            const unsigned char* bW = sW + lane * 16;
            auto read_rows = [&](int nzx, v8 (&rw)[6]) {
                const int nz = nzx / 3, nx = nzx % 3;
#pragma unroll
                for (int r = 0; r < 6; ++r) rw[r] = *reinterpret_cast<const v8*>(bIn + ((nz * HH + r) * HW + nx) * S);
            };
            auto read_a = [&](int t) -> v8 {
                const UlStep q = ul_step(t);                        // (t is a compile-time constant at every call)
                return *reinterpret_cast<const v8*>(bW + (q.par4 * 12 + q.tap) * 1024);
            };
            v8 rows[2][6], ar[3];
            read_rows(0, rows[0]);
            ar[0] = read_a(0);
            ar[1] = read_a(1);
            __builtin_amdgcn_sched_barrier(0);
            ul_static_for<0, 48>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr UlStep q = ul_step(t);
                constexpr bool first = t == 0 || ul_step(t > 0 ? t - 1 : 0).nzx != q.nzx;      // first pair of its tap offset
                constexpr bool rd_a = t + 2 < 48, rd_r = first && q.nzx + 1 < 9;
                if constexpr (rd_a) ar[(t + 2) % 3] = read_a(t + 2);
                if constexpr (rd_r) read_rows(q.nzx + 1, rows[(q.nzx + 1) & 1]);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[q.par4][ct] = UMfma<T>::run(ar[t % 3], rows[q.nzx & 1][ct + q.ny], acc[q.par4][ct]);
                if constexpr (rd_a && rd_r) __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
                else if constexpr (rd_r) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                else if constexpr (rd_a) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            });
        }
        if (st == ns - 1) {
            // ---- epilogue of this coarse box: its 8 x 256 fine voxels, border-class bias, BatchNorm sums of the ROUNDED values
            int n, d0, h0, w0;
            box_of(tile, n, d0, h0, w0);
            const int Df = 2 * p.D, Hf = 2 * p.H, Wf = 2 * p.W;
            const int cd = d0 + wave, cw = w0 + m;
            const bool touches = d0 == 0 || h0 == 0 || w0 == 0 || d0 + UL_TD >= p.D || h0 + UL_TH >= p.H || w0 + UL_BW >= p.W;   // uniform
            const int fx = 2 * cw + (kg >> 1);
#pragma unroll
            for (int par4 = 0; par4 < 4; ++par4) {
                const int fz = 2 * cd + (par4 >> 1);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int ch = h0 + ct, fy = 2 * ch + (par4 & 1);
                    if (cd < p.D && ch < p.H && cw < p.W) {
                        float4 bv = b0;
                        if (touches) {
                            const int cz = fz == 0 ? 1 : (fz == Df - 1 ? 2 : 0), cy = fy == 0 ? 1 : (fy == Hf - 1 ? 2 : 0),
                                      cx = fx == 0 ? 1 : (fx == Wf - 1 ? 2 : 0);
                            bv = *reinterpret_cast<const float4*>(p.beff + (size_t)((cz * 3 + cy) * 3 + cx) * 8 + (kg & 1) * 4);
                        }
                        const float4 o = rnd4<T>(make_float4(acc[par4][ct][0] + bv.x, acc[par4][ct][1] + bv.y,
                                                             acc[par4][ct][2] + bv.z, acc[par4][ct][3] + bv.w));
                        const size_t fv = (((size_t)n * Df + fz) * Hf + fy) * Wf + fx;
                        st4<T>(out + fv * p.out_cs + (kg & 1) * 4, o);
                        s1[0] += o.x; s1[1] += o.y; s1[2] += o.z; s1[3] += o.w;
                        s2[0] += o.x * o.x; s2[1] += o.y * o.y; s2[2] += o.z * o.z; s2[3] += o.w * o.w;
                    }
                    acc[par4][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (!has_next) break;
        tile = ntile; st = nst;
    }
    // ---- one BatchNorm partial row per block: reduce over the 16 voxel lanes, the two w-parity halves, the 4 waves
    if (p.stats) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a1 = s1[r], a2 = s2[r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
            a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32);
            if (m == 0 && kg < 2) {
                sRed[(wave * 2 + 0) * 8 + kg * 4 + r] = a1;
                sRed[(wave * 2 + 1) * 8 + kg * 4 + r] = a2;
            }
        }
        __syncthreads();
        if (tid < 16) {
            const int which = tid >> 3, c = tid & 7;
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) s += sRed[(wv * 2 + which) * 8 + c];
            p.stats[(size_t)blockIdx.x * 16 + which * 8 + c] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------ data gradient
struct UlBwdP {
    const void* g;            // fine [N, 2D, 2H, 2W, g_cs], 8 padded channels: raw-output gradient of the fused op
    const void* wb;           // lp_upconv_pack_bwd_kernel's fragments
    void* out;                // coarse [N, D, H, W, out_cs]
    int g_cs, out_cs, cin_p;
    int N, D, H, W, tiles_d, tiles_h, tiles_w;
};

constexpr int UL_FD = 2 * UL_TD + 2, UL_FH = 2 * UL_TH + 2, UL_FW = 2 * UL_BW + 2, UL_FV = UL_FD * UL_FH * UL_FW;     // fine halo box
constexpr int UL_BWD_NI = (UL_FV + 255) / 256;

template <class T, int NT>
__global__ __launch_bounds__(256, 1) void lp_upconv_bwd_data_kernel(UlBwdP p, int ntiles, int tiles_per_block) {
    typedef typename Vec<T>::v8 v8;
    constexpr int FH = UL_FH, FW = UL_FW, FV = UL_FV, NI = UL_BWD_NI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sG = smem;                                       // fine halo box, 16 bytes per voxel
    unsigned char* sW = smem + (size_t)FV * 16;                     // [16 K-steps][NT][1 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kg = lane >> 4;
    const T* g = reinterpret_cast<const T*>(p.g);
    T* out = reinterpret_cast<T*>(p.out);
    const int n16 = p.cin_p >> 4, nt0 = blockIdx.y * NT;
    const int Df = 2 * p.D, Hf = 2 * p.H, Wf = 2 * p.W;
    {   // this block's weight fragments, once
        const T* wb = reinterpret_cast<const T*>(p.wb);
        for (int i = tid; i < 16 * NT * 64; i += 256) {
            const int ln = i & 63, fr = i >> 6, nt = fr % NT, ks = fr / NT;
            const int tile = min(nt0 + nt, n16 - 1);
            *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = *reinterpret_cast<const uint4*>(wb + ((size_t)(ks * n16 + tile) * 64 + ln) * 8);
        }
    }
    unsigned ioff[NI];
    unsigned live = 0;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + u * 256;
        const bool lv = i < FV;
        const int v = lv ? i : 0;
        const int pw = v % FW, t2 = v / FW, ph = t2 % FH, pd = t2 / FH;
        ioff[u] = (unsigned)((((pd * Hf + ph) * Wf + pw) * p.g_cs) * (int)sizeof(T));
        live |= lv ? (1u << u) : 0u;
    }
    uint4 raw[NI];
    unsigned okb = 0;
    auto box_of = [&](int t, int& n, int& d0, int& h0, int& w0) {
        const int tx = t % p.tiles_w; t /= p.tiles_w;
        const int ty = t % p.tiles_h; t /= p.tiles_h;
        const int tz = t % p.tiles_d;
        n = t / p.tiles_d; d0 = tz * UL_TD; h0 = ty * UL_TH; w0 = tx * UL_BW;
    };
    auto load_box = [&](int t) {
        int n, d0, h0, w0;
        box_of(t, n, d0, h0, w0);
        const int z0 = 2 * d0 - 1, y0 = 2 * h0 - 1, x0 = 2 * w0 - 1;          // fine halo origin
        if (z0 >= 0 && z0 + UL_FD <= Df && y0 >= 0 && y0 + FH <= Hf && x0 >= 0 && x0 + FW <= Wf) {       // uniform
            const char* base = reinterpret_cast<const char*>(g + ((((size_t)n * Df + z0) * Hf + y0) * Wf + x0) * p.g_cs);
#pragma unroll
            for (int u = 0; u < NI; ++u) raw[u] = *reinterpret_cast<const uint4*>(base + (((live >> u) & 1u) ? ioff[u] : 0u));
            okb = live;
            return;
        }
        okb = 0;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int v = min(tid + u * 256, FV - 1);
            const int pw = v % FW, t2 = v / FW, ph = t2 % FH, pd = t2 / FH;
            const int gz = z0 + pd, gy = y0 + ph, gx = x0 + pw;
            const bool ok = ((live >> u) & 1u) && (unsigned)gz < (unsigned)Df && (unsigned)gy < (unsigned)Hf && (unsigned)gx < (unsigned)Wf;
            const int cz = min(max(gz, 0), Df - 1), cy = min(max(gy, 0), Hf - 1), cx = min(max(gx, 0), Wf - 1);
            raw[u] = *reinterpret_cast<const uint4*>(g + ((((size_t)n * Df + cz) * Hf + cy) * Wf + cx) * p.g_cs);
            okb |= ok ? (1u << u) : 0u;
        }
    };
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(ntiles, tile + tiles_per_block);
    if (tile < tile_end) load_box(tile);
    // lane (coarse voxel m of the row, K-quarter kg = w offset): fine voxel 2 m + kg of the halo row, 16 bytes per voxel
    const int lbase = (2 * wave * FH * FW + 2 * m + kg) * 16;
    for (; tile < tile_end; ++tile) {
        __syncthreads();                                            // weights visible (first pass) / previous box's readers done
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            if (!((live >> u) & 1u)) continue;
            uint4 r = raw[u];
            if (!((okb >> u) & 1u)) r = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(sG + (size_t)(tid + u * 256) * 16) = r;
        }
        __syncthreads();
        if (tile + 1 < tile_end) load_box(tile + 1);
        f32x4 acc[4][NT];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const unsigned char* bG = sG + lbase;
            const unsigned char* bW = sW + lane * 16;
            auto read_b = [&](int ks, v8 (&bb)[4]) {
                const int fd = ks >> 2, fh = ks & 3;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) bb[ct] = *reinterpret_cast<const v8*>(bG + ((fd * FH + 2 * ct + fh) * FW) * 16);
            };
            v8 bb[2][4];
            read_b(0, bb[0]);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks + 1 < 16) read_b(ks + 1, bb[(ks + 1) & 1]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const v8 a = *reinterpret_cast<const v8*>(bW + (ks * NT + nt) * 1024);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[ct][nt] = UMfma<T>::run(a, bb[ks & 1][ct], acc[ct][nt]);
                }
            }
        }
        int n, d0, h0, w0;
        box_of(tile, n, d0, h0, w0);
        const int cd = d0 + wave, cw = w0 + m;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ci = (nt0 + nt) * 16 + 4 * kg;
            if (ci >= p.cin_p) continue;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int ch = h0 + ct;
                if (cd < p.D && ch < p.H && cw < p.W)
                    st4<T>(out + ((((size_t)n * p.D + cd) * p.H + ch) * p.W + cw) * p.out_cs + ci,
                           make_float4(acc[ct][nt][0], acc[ct][nt][1], acc[ct][nt][2], acc[ct][nt][3]));
            }
        }
    }
}

void ul_grid(int ntiles, int* gx, int* tpb) {                      // one persistent block per CU
    int g = 256;
    if (g > ntiles) g = ntiles;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

template <class K>
int ul_raise_lds(K kernel, size_t lds, size_t* raised, const char* what) {
    if (lds > *raised) {
        CTU_REQUIRE(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                    "%s: cannot raise the dynamic LDS limit", what);
        *raised = lds;
    }
    return CTU_OK;
}

}  // namespace

// =================================================================== C ABI
// 16-bit fused up-convolution: 8 padded output channels, input channels a multiple of 32 (the decoder's top level of every
// shipped net: the concat of two 16-channel tensors), coarse volume at least one 16-wide box.  Other geometries take the
// unfused 16-bit ConvTranspose3d + Conv3d kernels.
extern "C" int ctu_lp_upconv_fused_supported(int k, int D, int H, int W, int cin_p, int nout_p) {
    return (k == 3 && W >= 16 && D >= 1 && H >= 1 && cin_p >= 32 && cin_p % 32 == 0 && cin_p <= 128 && nout_p == 8) ? 1 : 0;
}

extern "C" size_t ctu_lp_upconv_fused_packed_elems(int cin_p) {       // forward fragments, then data-gradient fragments
    return (size_t)(cin_p >> 5) * 48 * 512 + (size_t)16 * (cin_p >> 4) * 512;
}

extern "C" int ctu_lp_upconv_fused_num_blocks(int N, int D, int H, int W) {
    int gx, tpb;
    ul_grid(N * ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 16), &gx, &tpb);
    return gx;
}

// wp32: ctu_upconv_fused_pack's fp32 packing for (cin_p, nout_p = 8); wp16: ctu_lp_upconv_fused_packed_elems(cin_p) elements
extern "C" int ctu_lp_upconv_fused_pack(int dtype, const float* wp32, int cin_p, void* wp16, void* stream) {
    CTU_REQUIRE(wp32 && wp16 && cin_p >= 32 && cin_p % 32 == 0, "lp_upconv_fused_pack: bad argument (cin_p=%d)", cin_p);
    hipStream_t st = (hipStream_t)stream;
    const int nstage = cin_p >> 5;
    const size_t std_floats = (size_t)(cin_p / 8) * 8 * 8 * 128;      // standard packing precedes the PW packing (nout_p = 8)
    const int nf = nstage * 48 * 512, nb = 16 * (cin_p >> 4) * 512;
    CTU_DISPATCH_LP(dtype, {
        lp_upconv_pack_fwd_kernel<T><<<ceil_div(nf, 256), 256, 0, st>>>(wp32 + std_floats, (T*)wp16, nstage);
        lp_upconv_pack_bwd_kernel<T><<<ceil_div(nb, 256), 256, 0, st>>>(wp32, (T*)wp16 + nf, cin_p);
    });
    CTU_CHECK_LAUNCH("lp_upconv_fused_pack");
    return CTU_OK;
}

extern "C" int ctu_lp_upconv_fused_fwd(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                       int in_relu, const void* wp16, const float* beff, void* out, int out_cs, float* stats,
                                       int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(in && wp16 && beff && out, "lp_upconv_fused_fwd: null pointer");
    CTU_REQUIRE(ctu_lp_upconv_fused_supported(3, D, H, W, cin_p, 8), "lp_upconv_fused_fwd: unsupported geometry (W=%d cin_p=%d)", W, cin_p);
    CTU_REQUIRE(in_cs >= cin_p && in_cs % 8 == 0 && out_cs >= 8 && out_cs % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 7) == 0 &&
                ((uintptr_t)wp16 & 15) == 0 && ((uintptr_t)beff & 15) == 0, "lp_upconv_fused_fwd: strides / alignment");
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lp_upconv_fused_fwd: scale/shift come in pairs");
    CTU_REQUIRE((int64_t)(6 * H + 6) * W * in_cs * 2 < (int64_t)1 << 31, "lp_upconv_fused_fwd: volume too large for 32-bit offsets");
    UlFwdP p{};
    p.in = in; p.wf = wp16; p.out = out; p.scale = in_scale; p.shift = in_shift; p.beff = beff; p.stats = stats;
    p.in_cs = in_cs; p.cin_p = cin_p; p.relu = in_relu; p.out_cs = out_cs;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, 4); p.tiles_w = ceil_div(W, 16);
    const int ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
    int gx, tpb;
    ul_grid(ntiles, &gx, &tpb);
    hipStream_t st = (hipStream_t)stream;
    static size_t raised[2] = {64 * 1024, 64 * 1024};
    int rc = CTU_OK;
    CTU_DISPATCH_LP(dtype, {
        rc = ul_raise_lds(lp_upconv_fwd_kernel<T>, UL_FWD_LDS, &raised[dtype == CTU_BF16 ? 0 : 1], "lp_upconv_fused_fwd");
        if (rc != CTU_OK) return rc;
        lp_upconv_fwd_kernel<T><<<gx, 256, UL_FWD_LDS, st>>>(p, ntiles, tpb);
    });
    CTU_CHECK_LAUNCH("lp_upconv_fused_fwd");
    return CTU_OK;
}

extern "C" int ctu_lp_upconv_fused_bwd_data(int dtype, const void* gout, int g_cs, const void* wp16, void* gin, int gin_cs, int cin_p,
                                            int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(gout && wp16 && gin, "lp_upconv_fused_bwd_data: null pointer");
    CTU_REQUIRE(ctu_lp_upconv_fused_supported(3, D, H, W, cin_p, 8), "lp_upconv_fused_bwd_data: unsupported geometry");
    CTU_REQUIRE(g_cs >= 8 && g_cs % 8 == 0 && gin_cs >= cin_p && gin_cs % 4 == 0 && ((uintptr_t)gout & 15) == 0 && ((uintptr_t)gin & 7) == 0,
                "lp_upconv_fused_bwd_data: strides / alignment");
    CTU_REQUIRE((int64_t)(10 * 2 * H + 10) * 2 * W * g_cs * 2 < (int64_t)1 << 31, "lp_upconv_fused_bwd_data: volume too large for 32-bit offsets");
    UlBwdP p{};
    p.g = gout; p.out = gin; p.g_cs = g_cs; p.out_cs = gin_cs; p.cin_p = cin_p;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, 4); p.tiles_w = ceil_div(W, 16);
    const int ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
    const int n16 = cin_p >> 4;
    const int NT = n16 >= 4 ? 4 : 2;                                // cin_p is a multiple of 32: at least two tiles
    int gx, tpb;
    ul_grid(ntiles, &gx, &tpb);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)UL_FV * 16 + (size_t)16 * NT * 1024;
    static size_t raised[4] = {64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024};
    const dim3 grid(gx, ceil_div(n16, NT));
    int rc = CTU_OK;
    CTU_DISPATCH_LP(dtype, {
        p.wb = (const T*)wp16 + (size_t)(cin_p >> 5) * 48 * 512;
        if (NT == 4) {
            rc = ul_raise_lds(lp_upconv_bwd_data_kernel<T, 4>, lds, &raised[(dtype == CTU_BF16 ? 0 : 1) * 2 + 1], "lp_upconv_fused_bwd_data");
            if (rc != CTU_OK) return rc;
            lp_upconv_bwd_data_kernel<T, 4><<<grid, 256, lds, st>>>(p, ntiles, tpb);
        } else {
            rc = ul_raise_lds(lp_upconv_bwd_data_kernel<T, 2>, lds, &raised[(dtype == CTU_BF16 ? 0 : 1) * 2], "lp_upconv_fused_bwd_data");
            if (rc != CTU_OK) return rc;
            lp_upconv_bwd_data_kernel<T, 2><<<grid, 256, lds, st>>>(p, ntiles, tpb);
        }
    });
    CTU_CHECK_LAUNCH("lp_upconv_fused_bwd_data");
    return CTU_OK;
}
