// Reduced-precision (bf16 / fp16 storage, fp32 accumulate) 3D convolution for gfx950: BASELINE configs 4 and 5
// ("bf16, 192^3 patches", "fp16 MFMA conv path, 256^3 patches").  nn.Conv3d(k = 3 | 5, stride 1, padding (k-1)/2) of
// ctunet/pytorch/models.py:26,29,38,41,71,76,403,407,430,434,482-488 -- forward, data gradient (flipped / transposed
// packing) and weight gradient -- on v_mfma_f32_16x16x32_{bf16,f16}.
//
// With 16-bit activations every 128^3 / 64^3 stage of the net is HBM-bound (the bf16 MFMA ridge is ~400
// FLOP/B), so these kernels are built for bytes, not for MFMA issue rate: one voxel box per block, several blocks per
// CU (latency hidden by occupancy instead of hand pipelining), 16-byte channels-last accesses, the lazy BatchNorm + ReLU
// applied in fp32 while staging, BatchNorm partial sums taken from the ROUNDED outputs (the values consumers read back).
//
// forward / data gradient:  D[co][voxel] += A[co][k] * B[k][voxel],  k = (tap, 8-channel chunk) pairs, 4 pairs per MFMA.
//   B fragment of a lane = 8 consecutive channels of one (voxel + tap) = ONE ds_read_b128 of the haloed LDS box;
//   A fragment = packed weights in fragment order, read from global (L1 / L2 resident) through a register ring.
// weight gradient:  D[ci][co] += X^T[ci][voxel] * G[voxel][co], K = 32 voxels per MFMA; both operands are channels-last
//   LDS images read with ds_read_b64_tr_b16 (the hardware transpose), one accumulator per tap, one slab per block,
//   deterministic slab reduction (no float atomics).
#include "common.h"
#include "bn_tail.h"

namespace {

constexpr int LP_SC = 32;                 // input channels per stage (4 chunks of 8)

__host__ __device__ inline int lp_nstage(int rin_p) { return (rin_p + LP_SC - 1) / LP_SC; }
__host__ __device__ inline int lp_stage_nch(int rin_p, int st) {
    const int r = rin_p - st * LP_SC;
    return (r >= LP_SC ? LP_SC : r) >> 3;
}
__host__ __device__ inline int lp_ksteps(int taps, int nch) { return (taps * nch + 3) >> 2; }
__host__ __device__ inline int lp_total_ksteps(int taps, int rin_p) {
    const int ns = lp_nstage(rin_p);
    return (ns - 1) * taps + lp_ksteps(taps, lp_stage_nch(rin_p, ns - 1));       // full stages: taps * 4 / 4
}

template <class T> struct Mfma;
template <> struct Mfma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma<f16_t> {
    static __device__ __forceinline__ f32x4 run(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// ------------------------------------------------------------------------------------------------ weight packing
// wp[kstep (global over stages)][16-wide out tile][lane][8]: element j of lane l = A[row l&15][k = 8 (l>>4) + j] of that
// K-step: pair p = 4 s + (l >> 4) -> (tap, chunk) = (p / nch, p % nch) of the stage (nch chunks), zero for the padding pairs.
template <class T, int KS>
__device__ __forceinline__ void lp_pack_conv_w_elem(int idx, const float* __restrict__ w, T* __restrict__ wp, int Co, int Ci,
                                                    const int32_t* __restrict__ cinv, int rin_p, int nout_p, int mode) {
    constexpr int TAPS = KS * KS * KS;
    const int n16 = (nout_p + 15) >> 4;
    const int total = lp_total_ksteps(TAPS, rin_p) * n16 * 512;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    const int nt = (idx >> 9) % n16, kg = (idx >> 9) / n16;
    const int ns = lp_nstage(rin_p);
    int st = kg / TAPS;
    if (st > ns - 1) st = ns - 1;
    const int s = kg - st * TAPS, nch = lp_stage_nch(rin_p, st);
    const int pr = 4 * s + (lane >> 4), tap = pr / nch, ch = pr % nch;
    const int rp = st * LP_SC + ch * 8 + j, np = nt * 16 + (lane & 15);
    float v = 0.f;
    if (tap < TAPS) {
        if (mode == 0) {
            const int ci = cinv ? cinv[rp] : (rp < Ci ? rp : -1);
            if (ci >= 0 && np < Co) v = w[((size_t)np * Ci + ci) * TAPS + tap];
        } else {
            const int ci = (np < nout_p) ? (cinv ? cinv[np] : (np < Ci ? np : -1)) : -1;
            if (ci >= 0 && rp < Co) v = w[((size_t)rp * Ci + ci) * TAPS + (TAPS - 1 - tap)];
        }
    }
    wp[idx] = (T)v;
}

template <class T, int KS>
__global__ void lp_pack_conv_w_kernel(const float* __restrict__ w, T* __restrict__ wp, int Co, int Ci, const int32_t* __restrict__ cinv,
                                      int rin_p, int nout_p, int mode) {
    lp_pack_conv_w_elem<T, KS>(blockIdx.x * blockDim.x + threadIdx.x, w, wp, Co, Ci, cinv, rin_p, nout_p, mode);
}

// "pair" layout (k = 3, 8 padded channels on both sides, volumes at least 32 wide; lp_conv_fwd_pair_kernel): a K-step = the 4 w
// offsets kw' of one (kd, kh) row x 8 input channels, the 16 rows = (w-shift s, c_out): wp[ks = kd*3+kh][lane][8], element j of
// lane l = W[co = l&7][ci = j][kd, kh, kw' - s] for s = (l>>3)&1, kw' = l>>4, 0 <= kw' - s <= 2 (else 0).  mode 1: the data
// gradient's flipped / transposed weights in the same layout.
constexpr int LP_PAIR_ELEMS = 9 * 512;
template <class T>
__device__ __forceinline__ void lp_pack_conv_pair_elem(int idx, const float* __restrict__ w, T* __restrict__ wp, int Co, int Ci,
                                                       const int32_t* __restrict__ cinv, int mode) {
    if (idx >= LP_PAIR_ELEMS) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, row = lane & 15, s = row >> 3, r8 = row & 7, kwp = lane >> 4;
    const int ks = idx >> 9, kd = ks / 3, kh = ks % 3, kw = kwp - s;
    float v = 0.f;
    if (kw >= 0 && kw <= 2) {
        if (mode == 0) {
            const int ci = cinv ? cinv[j] : (j < Ci ? j : -1);
            if (ci >= 0 && r8 < Co) v = w[((size_t)r8 * Ci + ci) * 27 + (kd * 3 + kh) * 3 + kw];
        } else {
            const int ci = cinv ? cinv[r8] : (r8 < Ci ? r8 : -1);
            if (ci >= 0 && j < Co) v = w[((size_t)j * Ci + ci) * 27 + 26 - ((kd * 3 + kh) * 3 + kw)];
        }
    }
    wp[idx] = (T)v;
}
template <class T>
__global__ void lp_pack_conv_pair_kernel(const float* __restrict__ w, T* __restrict__ wp, int Co, int Ci, const int32_t* __restrict__ cinv, int mode) {
    lp_pack_conv_pair_elem<T>(blockIdx.x * blockDim.x + threadIdx.x, w, wp, Co, Ci, cinv, mode);
}

// ConvTranspose3d(C, C, 2, 2) packing (layouts: convt_lp.hip): mode 0 wp[tap][ks][n16][lane][8], mode 1 wp[ks][n16][lane][8]
template <class T>
__device__ __forceinline__ void lp_pack_convt_w_elem(int idx, const float* __restrict__ w, T* __restrict__ wp, int Ci, int Co,
                                                     const int32_t* __restrict__ cinv, int rin_p, int nout_p, int mode) {
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

// every 16-bit weight copy of a network in ONE launch: the job table travels in the kernel arguments (blockIdx.y = job)
constexpr int LP_PACK_MAXJ = 48;
struct LpPackTable {
    ctu_pack_job j[LP_PACK_MAXJ];
};

template <class T>
__global__ void lp_pack_batch_kernel(LpPackTable tb) {
    const ctu_pack_job& jb = tb.j[blockIdx.y];
    T* wp = reinterpret_cast<T*>(jb.wp);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x;; idx += gridDim.x * blockDim.x) {
        int total;
        if (jb.kind == 0) total = jb.layout == 1 ? LP_PAIR_ELEMS : lp_total_ksteps(jb.k * jb.k * jb.k, jb.rin_p) * ((jb.nout_p + 15) >> 4) * 512;
        else total = (jb.mode == 0 ? 8 * ((jb.rin_p + 31) >> 5) : 2 * (jb.rin_p >> 3)) * ((jb.nout_p + 15) >> 4) * 512;
        if (idx >= total) break;
        if (jb.kind == 0) {
            if (jb.layout == 1) lp_pack_conv_pair_elem<T>(idx, jb.w, wp, jb.Co, jb.Ci, jb.cinv, jb.mode);
            else if (jb.k == 3) lp_pack_conv_w_elem<T, 3>(idx, jb.w, wp, jb.Co, jb.Ci, jb.cinv, jb.rin_p, jb.nout_p, jb.mode);
            else lp_pack_conv_w_elem<T, 5>(idx, jb.w, wp, jb.Co, jb.Ci, jb.cinv, jb.rin_p, jb.nout_p, jb.mode);
        } else {
            lp_pack_convt_w_elem<T>(idx, jb.w, wp, jb.Ci, jb.Co, jb.cinv, jb.rin_p, jb.nout_p, jb.mode);
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward / data gradient
struct LpConvP {
    const void* in;
    const void* wp;
    void* out;
    const float* scale;
    const float* shift;
    const float* bias;
    float* stats;
    int in_cs, rin_p, relu, out_cs, nout_p, nbias;
    int N, D, H, W, tiles_d, tiles_h, tiles_w;
    int S;            // LDS bytes per halo voxel
    ctu_bn_tail tail; // counter != NULL: the last block of the launch finalizes the BatchNorm (bn_tail.h)
};

// Box TD = 4 x TH x BW voxels, one plane (td) per wave, CT = TH * BW / 16 column tiles of 16 voxels per wave.  The box grows
// when a voxel is small (few input channels): 4x8x32 for one 8-channel chunk, 4x8x16 for two, 4x4x16 otherwise -- a block
// then moves 16-32 KB instead of 4 KB and the per-block latency chain (global -> LDS -> barrier -> MFMA) amortises.
template <class T, int KS, int NT, int TH, int BW, bool WG>
__global__ __launch_bounds__(256) void lp_conv_fwd_kernel(LpConvP p) {
    typedef typename Vec<T>::v8 v8;
    constexpr int TAPS = KS * KS * KS, PK = (KS - 1) / 2;
    constexpr int TD = 4, CT = TH * BW / 16;
    constexpr int HD = TD + 2 * PK, HH = TH + 2 * PK, HW = BW + 2 * PK, HV = HD * HH * HW;
    constexpr int UB = 12;                                          // staging loads in flight per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* sK = reinterpret_cast<int*>(smem);                         // [TAPS * 4 padded to 512] byte offsets of the (tap, chunk) pairs
    float* sXf = reinterpret_cast<float*>(smem + 2048);             // [2][32] scale / shift of the stage
    float* sRed = reinterpret_cast<float*>(smem + 2048 + 256);      // [4 waves][NT][2][16]
    unsigned char* sIn = smem + 2048 + 256 + 4 * NT * 32 * 4;       // haloed input box of the stage
    // The weight fragments of GK K-steps at a time sit in LDS behind the box (GK * NT KB); a K-step then waits for LDS, not for
    // L2.  k = 3: GK = the whole stage (<= 27 K-steps).  k = 5 (up to 125 K-steps per stage): groups of 25; the next group is
    // fetched into registers under the current group's MFMAs and written behind a barrier.  (Reading them from global through
    // a register ring inside the K loop waited vmcnt(0) every K-step: 32 -> 8 at 192^3 took 5.5 ms against 0.5 ms for 8 -> 32.)
    // WG = false (k = 5 with one 8-channel chunk, or two out tiles per block: 8-16 MFMAs per K-step and up to 4 blocks per CU
    // hide the L2 latency): the fragments come straight from global through a register ring PF K-steps deep, no LDS copy.
    // Measured at 192^3 / 96^3 (bf16, k = 5): 32->8 5537 -> 3136 us, 64->16 1580 -> 793 with groups; 8->32 519 vs 1249,
    // 8->8 435 vs 519 without.
    constexpr int GK = KS == 3 ? 27 : 25;
    constexpr int NWR = (WG && KS != 3) ? (GK * NT * 64 + 255) / 256 : 1;      // 16-byte pieces of one weight group per thread (k = 5)
    constexpr int PF = 4;                                           // weight-fragment ring depth (K-steps ahead), WG = false
    unsigned char* sW = sIn + (size_t)HV * p.S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kg = lane >> 4;
    const int S = p.S;
    int t = blockIdx.x;
    const int tx = t % p.tiles_w; t /= p.tiles_w;
    const int ty = t % p.tiles_h; t /= p.tiles_h;
    const int tz = t % p.tiles_d;
    const int n = t / p.tiles_d;
    const int d0 = tz * TD, h0 = ty * TH, w0 = tx * BW;
    const int n16 = (p.nout_p + 15) >> 4, nt0 = blockIdx.y * NT;
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* wp = reinterpret_cast<const T*>(p.wp);
    // this lane's voxel in each of the wave's column tiles: (th, tw) inside the box; halo byte offset of the top-left tap
    auto vox_of = [&](int ct, int& th, int& tw) {
        if (BW == 32) { th = ct >> 1; tw = (ct & 1) * 16 + m; }
        else if (BW == 16) { th = ct; tw = m; }
        else { th = ct * 2 + (m >> 3); tw = m & 7; }
    };
    f32x4 acc[CT][NT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int th0, tw0;
    vox_of(0, th0, tw0);
    const int hb0 = ((wave * HH + th0) * HW + tw0) * S;             // column tile ct sits ct_step further
    constexpr int CT_ROWSTEP = (BW == 32) ? 0 : ((BW == 16) ? 1 : 2);
    const int ns = lp_nstage(p.rin_p);
    const bool xf = p.scale != nullptr;
    for (int st = 0; st < ns; ++st) {
        const int nch = lp_stage_nch(p.rin_p, st);
        const int nchp = nch == 3 ? 4 : nch, sh = nchp == 4 ? 2 : (nchp == 2 ? 1 : 0);
        const int ks = lp_ksteps(TAPS, nch);
        __syncthreads();                                            // the previous stage's reads are done
        for (int i = tid; i < ks * 4; i += 256) {
            const int tap = i / nch, ch = i % nch;
            int off = 0;                                            // padding pairs: zero weights, any valid address
            if (tap < TAPS) off = (((tap / (KS * KS)) * HH + (tap / KS) % KS) * HW + tap % KS) * S + ch * 16;
            sK[i] = off;
        }
        if (xf && tid < 64) {
            const int c = st * LP_SC + (tid & 31);
            sXf[tid] = (c < p.rin_p) ? ((tid < 32) ? p.scale[c] : p.shift[c]) : 0.f;
        }
        // weight fragments of K-step group g of this stage -> registers (independent of LDS; L2-resident, clamped index)
        const int kbase = st * TAPS;
        const int ngroups = KS == 3 ? 1 : (ks + GK - 1) / GK;      // (k = 3: one group, known at compile time)
        uint4 wreg[NWR];
        auto load_wgroup = [&](int g) {
            const int s0 = g * GK, pieces = min(GK, ks - s0) * NT * 64;
#pragma unroll
            for (int u = 0; u < NWR; ++u) {
                const int i = min(tid + u * 256, pieces - 1);
                const int ln = i & 63, fr = i >> 6, nt = fr % NT, s_ = fr / NT;
                const int tile = min(nt0 + nt, n16 - 1);
                wreg[u] = *reinterpret_cast<const uint4*>(wp + ((size_t)((kbase + s0 + s_) * n16 + tile) * 64 + ln) * 8);
            }
        };
        auto store_wgroup = [&](int g) {
            const int pieces = min(GK, ks - g * GK) * NT * 64;
#pragma unroll
            for (int u = 0; u < NWR; ++u) {
                const int i = tid + u * 256;
                if (i < pieces) *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = wreg[u];
            }
        };
        v8 ring[PF][NT];
        if constexpr (WG && KS == 3) {
            // the whole stage (ks * NT fragments of 1 KB) in batches of 8 pieces per thread: all of them in flight at once
            // would cost 56 registers for two out tiles and an occupancy step of the small-volume launches
            const int pieces = ks * NT * 64;
            for (int i0 = tid; i0 < pieces; i0 += 256 * 8) {
                uint4 w8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = min(i0 + u * 256, pieces - 1);
                    const int ln = i & 63, fr = i >> 6, nt = fr % NT, s_ = fr / NT;
                    const int tile = min(nt0 + nt, n16 - 1);
                    w8[u] = *reinterpret_cast<const uint4*>(wp + ((size_t)((kbase + s_) * n16 + tile) * 64 + ln) * 8);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * 256;
                    if (i < pieces) *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = w8[u];
                }
            }
        } else if constexpr (WG) {
            load_wgroup(0);
            store_wgroup(0);                                        // (sW is free: the previous stage ended behind the barrier above)
        } else {
#pragma unroll
            for (int u = 0; u < PF; ++u)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int tile = min(nt0 + nt, n16 - 1), s_ = min(u, ks - 1);
                    ring[u][nt] = *reinterpret_cast<const v8*>(wp + ((size_t)((kbase + s_) * n16 + tile) * 64 + lane) * 8);
                }
        }
        if (xf) __syncthreads();                                    // sXf visible
        // ---- stage the haloed box: 16-byte items (halo voxel, chunk), up to UB loads in flight per thread
        // (every item of a thread is the same chunk -- 256 is a multiple of the chunks per voxel -- so its BatchNorm vectors go
        //  to registers once per stage instead of 16 LDS reads per item)
        float xs[8], xh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { xs[j] = xf ? sXf[(tid & (nchp - 1)) * 8 + j] : 1.f; xh[j] = xf ? sXf[32 + (tid & (nchp - 1)) * 8 + j] : 0.f; }
        const int items = HV << sh;
        for (int i0 = tid; i0 < items; i0 += 256 * UB) {
            uint4 raw[UB];
            int dst[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * 256;
                const int v = i >> sh, c = i & (nchp - 1);
                const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
                const int gd = d0 + pd - PK, gh = h0 + ph - PK, gw = w0 + pw - PK;
                const bool live = i < items && c < nch;
                const bool ok = live && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
                // bit 30: outside the volume (the ACTIVATED input is zero padded); bits 28-29: chunk within the stage
                dst[u] = live ? ((v * S + c * 16) | (c << 26) | (ok ? 0 : 0x40000000)) : -1;
                // branch-free: every item loads (a clamped, always valid address); the out-of-volume ones are zeroed below.
                // A load behind a per-item branch is waited for before the next one is issued (one HBM round trip each).
                const int cd = min(max(gd, 0), p.D - 1), chh = min(max(gh, 0), p.H - 1), cw = min(max(gw, 0), p.W - 1);
                const int cc = live ? c : 0;
                raw[u] = *reinterpret_cast<const uint4*>(in + ((((size_t)n * p.D + cd) * p.H + chh) * p.W + cw) * p.in_cs +
                                                         st * LP_SC + cc * 8);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (dst[u] < 0) continue;
                uint4 r = raw[u];
                if (dst[u] & 0x40000000) r = make_uint4(0u, 0u, 0u, 0u);
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
                *reinterpret_cast<uint4*>(sIn + (dst[u] & 0x03ffffff)) = r;
            }
        }
        __syncthreads();
        // ---- K loop, one weight group at a time
        // A compiler-scheduled "read, wait, MFMA" chain pays one LDS latency (~100 cycles) per 16-cycle MFMA.  Instead: the
        // fragments of half a K-step (CT / 2 column tiles) are read as one batch while the other half's MFMAs run.
        constexpr int HB = CT / 2;
        auto read_b = [&](int koff, int half, v8 (&bb)[HB]) {
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int ct = half * HB + j;
                const int cto = (BW == 32) ? ((ct >> 1) * HW + (ct & 1) * 16) * S : ct * CT_ROWSTEP * HW * S;
                bb[j] = *reinterpret_cast<const v8*>(sIn + koff + cto);
            }
        };
        if constexpr (WG)
        for (int g = 0; g < ngroups; ++g) {
            if constexpr (KS != 3) { if (g + 1 < ngroups) load_wgroup(g + 1); }     // lands under this group's MFMAs
            const int s0 = g * GK, s1e = min(ks, s0 + GK);           // K-steps [s0, s1e) of the stage; their weights: sW[s - s0]
            v8 b0[HB], b1[HB], a[NT];
            int koff = sK[4 * s0 + kg] + hb0;
            int koff_n = sK[4 * min(s0 + 1, ks - 1) + kg] + hb0;    // the offset table is read TWO K-steps ahead: its (in-order)
#pragma unroll                                                      // LDS read is then the oldest in flight when it is needed
            for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const v8*>(sW + ((size_t)nt * 64 + lane) * 16);
            read_b(koff, 0, b0);
            for (int s = s0; s < s1e; ++s) {
                const int koff_nn = sK[4 * min(s + 2, ks - 1) + kg] + hb0;
                read_b(koff, 1, b1);                                // second half of K-step s: lands under the first half's MFMAs
                const int sn = min(s + 1, s1e - 1) - s0;
                v8 an[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) an[nt] = *reinterpret_cast<const v8*>(sW + ((size_t)(sn * NT + nt) * 64 + lane) * 16);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HB; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[j][nt] = Mfma<T>::run(a[nt], b0[j], acc[j][nt]);
                __builtin_amdgcn_sched_barrier(0);
                read_b(koff_n, 0, b0);                              // first half of K-step s + 1: under the second half's MFMAs
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HB; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[HB + j][nt] = Mfma<T>::run(a[nt], b1[j], acc[HB + j][nt]);
                __builtin_amdgcn_sched_barrier(0);
                koff = koff_n;
                koff_n = koff_nn;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) a[nt] = an[nt];
            }
            if constexpr (KS != 3) {
                if (g + 1 < ngroups) {
                    __syncthreads();                                // every wave is done with this group's weights
                    store_wgroup(g + 1);
                    __syncthreads();
                }
            }
        }
        else
        for (int s0 = 0; s0 < ks; s0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int s = s0 + u;
                if (s < ks) {
                    const int koff = sK[4 * s + kg] + hb0;
                    v8 a[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) a[nt] = ring[u][nt];
                    if (s + PF < ks) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const int tile = min(nt0 + nt, n16 - 1);
                            ring[u][nt] = *reinterpret_cast<const v8*>(wp + ((size_t)((kbase + s + PF) * n16 + tile) * 64 + lane) * 8);
                        }
                    }
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        // column tile ct: BW 32 -> (row ct / 2, half ct & 1); BW 16 -> row ct; BW 8 -> rows 2 ct, 2 ct + 1
                        const int cto = (BW == 32) ? ((ct >> 1) * HW + (ct & 1) * 16) * S : ct * CT_ROWSTEP * HW * S;
                        const v8 b = *reinterpret_cast<const v8*>(sIn + koff + cto);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[ct][nt] = Mfma<T>::run(a[nt], b, acc[ct][nt]);
                    }
                }
            }
        }
    }
    // ---- epilogue: lane holds out channels 4 kg .. 4 kg + 3 of tile nt for its voxel of column tile ct
    T* out = reinterpret_cast<T*>(p.out);
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    const int gd = d0 + wave;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cb = (nt0 + nt) * 16 + 4 * kg;
        if (cb >= p.nout_p) continue;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            bv.x = cb + 0 < p.nbias ? p.bias[cb + 0] : 0.f; bv.y = cb + 1 < p.nbias ? p.bias[cb + 1] : 0.f;
            bv.z = cb + 2 < p.nbias ? p.bias[cb + 2] : 0.f; bv.w = cb + 3 < p.nbias ? p.bias[cb + 3] : 0.f;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            int th, tw;
            vox_of(ct, th, tw);
            const int gh = h0 + th, gw = w0 + tw;
            if (gd < p.D && gh < p.H && gw < p.W) {
                const float4 o = rnd4<T>(make_float4(acc[ct][nt][0] + bv.x, acc[ct][nt][1] + bv.y, acc[ct][nt][2] + bv.z,
                                                     acc[ct][nt][3] + bv.w));
                st4<T>(out + ((((size_t)n * p.D + gd) * p.H + gh) * p.W + gw) * p.out_cs + cb, o);
                s1[nt][0] += o.x; s1[nt][1] += o.y; s1[nt][2] += o.z; s1[nt][3] += o.w;
                s2[nt][0] += o.x * o.x; s2[nt][1] += o.y * o.y; s2[nt][2] += o.z * o.z; s2[nt][3] += o.w * o.w;
            }
        }
    }
    if (p.stats) {                                                  // one BatchNorm partial row [2][nout_p] per spatial block
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a1 = s1[nt][r], a2 = s2[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
                if (m == 0) {
                    sRed[((wave * NT + nt) * 2 + 0) * 16 + 4 * kg + r] = a1;
                    sRed[((wave * NT + nt) * 2 + 1) * 16 + 4 * kg + r] = a2;
                }
            }
        __syncthreads();
        if (tid < NT * 32) {
            const int nt = tid >> 5, which = (tid >> 4) & 1, c = tid & 15;
            const int ch = (nt0 + nt) * 16 + c;
            if (ch < p.nout_p) {
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) s += sRed[((wv * NT + nt) * 2 + which) * 16 + c];
                st_row(p.tail.counter != nullptr, p.stats + (size_t)blockIdx.x * 2 * p.nout_p + which * p.nout_p + ch, s);
            }
        }
        if (p.tail.counter) bn_fwd_tail(p.tail, p.stats, gridDim.x, p.nout_p, gridDim.x * gridDim.y);
    }
}

// ---- deep-level k = 3 variant (<= 16^3 voxels, 4x4x8 boxes): these launches have 16-128 blocks on 256 CUs, so a block's own
// latency chain is the kernel's duration; in the box kernel above every 32-channel stage pays two exposed global round trips
// (weights, input box) before its 54 MFMAs.  Here the NEXT stage's weights, input items and BatchNorm vectors are fetched into
// registers while this stage's K loop runs: a stage costs LDS writes, three barriers and the K loop
// (128 -> 128 at 8^3: 4 stages).  Same packed weights, same stats rows (one per box) as lp_conv_fwd_kernel<.., 4, 8, ..>.
template <class T, int NT>
__global__ __launch_bounds__(256) void lp_conv_fwd_small_kernel(LpConvP p) {
    typedef typename Vec<T>::v8 v8;
    constexpr int TAPS = 27, PK = 1, TD = 4, TH = 4, BW = 8, CT = 2;
    constexpr int HD = TD + 2, HH = TH + 2, HW = BW + 2, HV = HD * HH * HW;
    constexpr int NI = (HV * 4 + 255) / 256;                         // input items (halo voxel, chunk slot) per thread
    constexpr int NW = (TAPS * NT * 64 + 255) / 256;                 // 16-byte weight pieces of a stage per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* sK = reinterpret_cast<int*>(smem);
    float* sXf = reinterpret_cast<float*>(smem + 2048);
    float* sRed = reinterpret_cast<float*>(smem + 2048 + 256);
    unsigned char* sIn = smem + 2048 + 256 + 4 * NT * 32 * 4;
    unsigned char* sW = sIn + (size_t)HV * p.S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kg = lane >> 4;
    const int S = p.S;
    int t = blockIdx.x;
    const int tx = t % p.tiles_w; t /= p.tiles_w;
    const int ty = t % p.tiles_h; t /= p.tiles_h;
    const int tz = t % p.tiles_d;
    const int n = t / p.tiles_d;
    const int d0 = tz * TD, h0 = ty * TH, w0 = tx * BW;
    const int n16 = (p.nout_p + 15) >> 4, nt0 = blockIdx.y * NT;
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* wp = reinterpret_cast<const T*>(p.wp);
    const int ns = lp_nstage(p.rin_p);
    const bool xf = p.scale != nullptr;
    // per-thread input items: (halo voxel v, chunk slot c of 4); clamped global element offset, validity, LDS destination
    size_t goff[NI];
    int dst[NI];
    unsigned inb = 0;                                               // bit u: the voxel of item u lies inside the volume
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + u * 256, v = min(i >> 2, HV - 1), c = i & 3;
        const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
        const int gd = d0 + pd - PK, gh = h0 + ph - PK, gw = w0 + pw - PK;
        const bool ok = i < HV * 4 && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
        const int cd = min(max(gd, 0), p.D - 1), chh = min(max(gh, 0), p.H - 1), cw = min(max(gw, 0), p.W - 1);
        goff[u] = ((((size_t)n * p.D + cd) * p.H + chh) * p.W + cw) * p.in_cs;
        dst[u] = i < HV * 4 ? v * S + c * 16 : -1;
        inb |= ok ? (1u << u) : 0u;
    }
    uint4 araw[NI], wq[NW];
    float xfv = 0.f;
    auto load_stage = [&](int st) {
        const int nch = lp_stage_nch(p.rin_p, st), pieces = lp_ksteps(TAPS, nch) * NT * 64, kbase = st * TAPS;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = min(tid + u * 256, pieces - 1);
            const int ln = i & 63, fr = i >> 6, nt = fr % NT, s_ = fr / NT;
            const int tile = min(nt0 + nt, n16 - 1);
            wq[u] = *reinterpret_cast<const uint4*>(wp + ((size_t)((kbase + s_) * n16 + tile) * 64 + ln) * 8);
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int c = (tid + u * 256) & 3;
            araw[u] = *reinterpret_cast<const uint4*>(in + goff[u] + st * LP_SC + (c < nch ? c : 0) * 8);
        }
        if (xf && tid < 64) {
            const int c = st * LP_SC + (tid & 31);
            xfv = (c < p.rin_p) ? ((tid < 32) ? p.scale[c] : p.shift[c]) : 0.f;
        }
    };
    f32x4 acc[CT][NT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's voxel of column tile ct: rows th = 2 ct + (m >> 3), column tw = m & 7
    const int hb0 = ((wave * HH + (m >> 3)) * HW + (m & 7)) * S;
    load_stage(0);
    for (int st = 0; st < ns; ++st) {
        const int nch = lp_stage_nch(p.rin_p, st), ks = lp_ksteps(TAPS, nch);
        __syncthreads();                                            // the previous stage's K loop is done with sK, sW, sIn
        for (int i = tid; i < ks * 4; i += 256) {
            const int tap = i / nch, ch = i % nch;
            sK[i] = tap < TAPS ? (((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3) * S + ch * 16 : 0;
        }
        if (tid < 64) sXf[tid] = xfv;
        {
            const int pieces = ks * NT * 64;
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                const int i = tid + u * 256;
                if (i < pieces) *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = wq[u];
            }
        }
        __syncthreads();                                            // sXf visible
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int c = (tid + u * 256) & 3;
            if (dst[u] < 0 || c >= nch) continue;
            uint4 r = araw[u];
            if (!((inb >> u) & 1u)) r = make_uint4(0u, 0u, 0u, 0u);     // the ACTIVATED input is zero padded
            else if (xf) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&r), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], sXf[c * 8 + j], sXf[32 + c * 8 + j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&r) = __builtin_convertvector(o, v8);
            }
            *reinterpret_cast<uint4*>(sIn + dst[u]) = r;
        }
        __syncthreads();
        if (st + 1 < ns) load_stage(st + 1);                        // in flight under this stage's K loop
        // ---- K loop, U K-steps per batch: every fragment of the batch (2 U input + U NT weight reads, U offset-table
        // reads for the NEXT batch) is in flight before its 2 U NT MFMAs -- with two column tiles per wave a K-step alone
        // carries too few MFMAs to hide an LDS round trip (one K-step at a time: 27 dependent round trips per stage)
        {
            constexpr int U = 3;
            int ko[U];
#pragma unroll
            for (int u = 0; u < U; ++u) ko[u] = sK[4 * min(u, ks - 1) + kg] + hb0;
            for (int s0 = 0; s0 < ks; s0 += U) {
                v8 bb[U][2], aa[U][NT];
                int kn[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int su = min(s0 + u, ks - 1);
                    bb[u][0] = *reinterpret_cast<const v8*>(sIn + ko[u]);
                    bb[u][1] = *reinterpret_cast<const v8*>(sIn + ko[u] + 2 * HW * S);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) aa[u][nt] = *reinterpret_cast<const v8*>(sW + ((size_t)(su * NT + nt) * 64 + lane) * 16);
                    kn[u] = sK[4 * min(s0 + U + u, ks - 1) + kg] + hb0;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (s0 + u < ks) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            acc[0][nt] = Mfma<T>::run(aa[u][nt], bb[u][0], acc[0][nt]);
                            acc[1][nt] = Mfma<T>::run(aa[u][nt], bb[u][1], acc[1][nt]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < U; ++u) ko[u] = kn[u];
            }
        }
    }
    // ---- epilogue: lane holds out channels 4 kg .. 4 kg + 3 of tile nt for its voxel of column tile ct
    T* out = reinterpret_cast<T*>(p.out);
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    const int gd = d0 + wave;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cb = (nt0 + nt) * 16 + 4 * kg;
        if (cb >= p.nout_p) continue;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            bv.x = cb + 0 < p.nbias ? p.bias[cb + 0] : 0.f; bv.y = cb + 1 < p.nbias ? p.bias[cb + 1] : 0.f;
            bv.z = cb + 2 < p.nbias ? p.bias[cb + 2] : 0.f; bv.w = cb + 3 < p.nbias ? p.bias[cb + 3] : 0.f;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int gh = h0 + ct * 2 + (m >> 3), gw = w0 + (m & 7);
            if (gd < p.D && gh < p.H && gw < p.W) {
                const float4 o = rnd4<T>(make_float4(acc[ct][nt][0] + bv.x, acc[ct][nt][1] + bv.y, acc[ct][nt][2] + bv.z,
                                                     acc[ct][nt][3] + bv.w));
                st4<T>(out + ((((size_t)n * p.D + gd) * p.H + gh) * p.W + gw) * p.out_cs + cb, o);
                s1[nt][0] += o.x; s1[nt][1] += o.y; s1[nt][2] += o.z; s1[nt][3] += o.w;
                s2[nt][0] += o.x * o.x; s2[nt][1] += o.y * o.y; s2[nt][2] += o.z * o.z; s2[nt][3] += o.w * o.w;
            }
        }
    }
    if (p.stats) {                                                  // one BatchNorm partial row [2][nout_p] per spatial block
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a1 = s1[nt][r], a2 = s2[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
                if (m == 0) {
                    sRed[((wave * NT + nt) * 2 + 0) * 16 + 4 * kg + r] = a1;
                    sRed[((wave * NT + nt) * 2 + 1) * 16 + 4 * kg + r] = a2;
                }
            }
        __syncthreads();
        if (tid < NT * 32) {
            const int nt = tid >> 5, which = (tid >> 4) & 1, c = tid & 15;
            const int ch = (nt0 + nt) * 16 + c;
            if (ch < p.nout_p) {
                float sx = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) sx += sRed[((wv * NT + nt) * 2 + which) * 16 + c];
                st_row(p.tail.counter != nullptr, p.stats + (size_t)blockIdx.x * 2 * p.nout_p + which * p.nout_p + ch, sx);
            }
        }
        if (p.tail.counter) bn_fwd_tail(p.tail, p.stats, gridDim.x, p.nout_p, gridDim.x * gridDim.y);
    }
}

// ---- persistent single-stage k = 3 variant (rin_p <= 32: every 128^3 / 64^3 layer of the shipped nets).  With 16-byte voxels
// the box kernel above is VALU-bound on index arithmetic (SQ counters: ~2000 vector instructions per wave for 112 MFMAs), so:
//   * a block walks a contiguous range of boxes; tap-offset table, BatchNorm vectors and ALL weight fragments are staged once;
//   * per-thread item offsets are computed once per block -- an interior box costs one 32-bit add per load;
//   * the next box's loads are in flight (in registers) under this box's MFMAs and epilogue;
//   * BatchNorm partial sums stay in registers across the boxes: ONE stats row per block.
#ifdef CTU_LP_STAMP
// diagnostic build only (scripts/diag_stamp_lp.hip): per-phase cycle sums of wave 0, written to a buffer of their own
__device__ unsigned long long* g_lp_stamp_out = nullptr;
#define LPSTAMP(var)                                                                 \
    do {                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");  \
        __builtin_amdgcn_sched_barrier(0);                                           \
    } while (0)
#else
#define LPSTAMP(var) do { } while (0)
#endif

template <class T, int NT, int TH, int BW>
__global__ __launch_bounds__(256) void lp_conv_fwd_p1_kernel(LpConvP p, int ntiles, int tiles_per_block) {
    typedef typename Vec<T>::v8 v8;
    constexpr int KS = 3, TAPS = 27, PK = 1;
    constexpr int TD = 4, CT = TH * BW / 16, HB = CT / 2;
    constexpr int HD = TD + 2, HH = TH + 2, HW = BW + 2, HV = HD * HH * HW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* sK = reinterpret_cast<int*>(smem);
    float* sXf = reinterpret_cast<float*>(smem + 2048);
    float* sRed = reinterpret_cast<float*>(smem + 2048 + 256);
    unsigned char* sIn = smem + 2048 + 256 + 4 * NT * 32 * 4;
    unsigned char* sW = sIn + (size_t)HV * p.S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kg = lane >> 4;
    const int S = p.S;
    const int n16 = (p.nout_p + 15) >> 4, nt0 = blockIdx.y * NT;
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* wp = reinterpret_cast<const T*>(p.wp);
    T* out = reinterpret_cast<T*>(p.out);
    const int nch = p.rin_p >> 3, nchp = nch == 3 ? 4 : nch, sh = nchp == 4 ? 2 : (nchp == 2 ? 1 : 0);
    const int ks = lp_ksteps(TAPS, nch);
    const bool xf = p.scale != nullptr;
    // ---- once per block: tables and weights
    for (int i = tid; i < ks * 4; i += 256) {
        const int tap = i / nch, ch = i % nch;
        int off = 0;
        if (tap < TAPS) off = (((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3) * S + ch * 16;
        sK[i] = off;
    }
    if (tid < 64) {
        const int c = tid & 31;
        sXf[tid] = (xf && c < p.rin_p) ? ((tid < 32) ? p.scale[c] : p.shift[c]) : 0.f;
    }
    {
        const int pieces = ks * NT * 64;
        for (int i0 = tid; i0 < pieces; i0 += 256 * 8) {
            uint4 w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * 256, pieces - 1);
                const int ln = i & 63, fr = i >> 6, nt = fr % NT, sx = fr / NT;
                const int tile = min(nt0 + nt, n16 - 1);
                w8[u] = *reinterpret_cast<const uint4*>(wp + ((size_t)(sx * n16 + tile) * 64 + ln) * 8);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                if (i < pieces) *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = w8[u];
            }
        }
    }
    // ---- per-thread staging items: byte offset relative to the halo origin, LDS destination, chunk
    constexpr int MAXCH = (BW == 32) ? 1 : ((TH == 8) ? 2 : 4);     // chunks per voxel this box is chosen for (lp_box)
    constexpr int NI = (HV * MAXCH + 255) / 256;
    const int items = HV << sh;
    unsigned ioff[NI];
    int dst[NI];
    unsigned live = 0;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + u * 256;
        const int v = min(i >> sh, HV - 1), c = i & (nchp - 1);
        const bool lv = i < items && c < nch;
        const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
        ioff[u] = (unsigned)((((pd * p.H + ph) * p.W + pw) * p.in_cs + (lv ? c : 0) * 8) * (int)sizeof(T));
        dst[u] = (v * S + c * 16) | (c << 26);
        live |= lv ? (1u << u) : 0u;
    }
    auto vox_of = [&](int ct, int& th, int& tw) {
        if (BW == 32) { th = ct >> 1; tw = (ct & 1) * 16 + m; }
        else if (BW == 16) { th = ct; tw = m; }
        else { th = ct * 2 + (m >> 3); tw = m & 7; }
    };
    int th0, tw0;
    vox_of(0, th0, tw0);
    const int hb0 = ((wave * HH + th0) * HW + tw0) * S;
    constexpr int CT_ROWSTEP = (BW == 32) ? 0 : ((BW == 16) ? 1 : 2);
    // output byte offsets: row pitch, voxel pitch, and this lane's offset inside the box for column tile 0 / out tile nt0
    const int orow_b = p.W * p.out_cs * (int)sizeof(T), vox_b = p.out_cs * (int)sizeof(T);
    const unsigned eoff0 = (unsigned)(th0 * orow_b + tw0 * vox_b + (nt0 * 16 + 4 * kg) * (int)sizeof(T));
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(ntiles, tile + tiles_per_block);
    uint4 raw[NI];
    unsigned okb = 0;
    auto box_of = [&](int t, int& n, int& d0, int& h0, int& w0) {
        const int tx = t % p.tiles_w; t /= p.tiles_w;
        const int ty = t % p.tiles_h; t /= p.tiles_h;
        const int tz = t % p.tiles_d;
        n = t / p.tiles_d; d0 = tz * TD; h0 = ty * TH; w0 = tx * BW;
    };
    auto load_box = [&](int t) {
        int n, d0, h0, w0;
        box_of(t, n, d0, h0, w0);
        if (d0 >= 1 && d0 + TD + 1 <= p.D && h0 >= 1 && h0 + TH + 1 <= p.H && w0 >= 1 && w0 + BW + 1 <= p.W) {   // uniform
            const char* base = reinterpret_cast<const char*>(in + ((((size_t)n * p.D + d0 - 1) * p.H + h0 - 1) * p.W + w0 - 1) * p.in_cs);
#pragma unroll
            for (int u = 0; u < NI; ++u) raw[u] = *reinterpret_cast<const uint4*>(base + ioff[u]);
            okb = live;
            return;
        }
        okb = 0;
#pragma unroll
        for (int u = 0; u < NI; ++u) {                              // border / ragged box: clamped, branch-free
            const int i = tid + u * 256;
            const int v = min(i >> sh, HV - 1), c = i & (nchp - 1);
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - PK, gh = h0 + ph - PK, gw = w0 + pw - PK;
            const bool ok = ((live >> u) & 1u) && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            const int cd = min(max(gd, 0), p.D - 1), chh = min(max(gh, 0), p.H - 1), cw = min(max(gw, 0), p.W - 1);
            raw[u] = *reinterpret_cast<const uint4*>(in + ((((size_t)n * p.D + cd) * p.H + chh) * p.W + cw) * p.in_cs + (((live >> u) & 1u) ? c : 0) * 8);
            okb |= ok ? (1u << u) : 0u;
        }
    };
    if (tile < tile_end) load_box(tile);
#ifdef CTU_LP_STAMP
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, ph_[5] = {0, 0, 0, 0, 0}, nst = 0;
    LPSTAMP(q0);
#endif
    // every item of a thread is the same 8-channel chunk (256 is a multiple of the chunks per voxel): its BatchNorm vectors
    // sit in registers for the whole block instead of 16 LDS reads per item
    float xs[8], xh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = (tid & (nchp - 1)) * 8 + j;
        xs[j] = (xf && c < p.rin_p) ? p.scale[c] : 0.f;
        xh[j] = (xf && c < p.rin_p) ? p.shift[c] : 0.f;
    }
    for (; tile < tile_end; ++tile) {
        __syncthreads();                                            // tables visible (first pass) / previous box's readers done
        LPSTAMP(q1);
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            if (!((live >> u) & 1u)) continue;
            uint4 r = raw[u];
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
            *reinterpret_cast<uint4*>(sIn + (dst[u] & 0x03ffffff)) = r;
        }
        __syncthreads();
        LPSTAMP(q2);
        if (tile + 1 < tile_end) load_box(tile + 1);
        LPSTAMP(q3);
        // ---- K loop (see lp_conv_fwd_kernel)
        f32x4 acc[CT][NT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            auto read_b = [&](int koff, int half, v8 (&bb)[HB]) {
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const int ct = half * HB + j;
                    const int cto = (BW == 32) ? ((ct >> 1) * HW + (ct & 1) * 16) * S : ct * CT_ROWSTEP * HW * S;
                    bb[j] = *reinterpret_cast<const v8*>(sIn + koff + cto);
                }
            };
            v8 b0[HB], b1[HB], a[NT];
            int koff = sK[kg] + hb0;
            int koff_n = sK[4 * min(1, ks - 1) + kg] + hb0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const v8*>(sW + ((size_t)nt * 64 + lane) * 16);
            read_b(koff, 0, b0);
            for (int s = 0; s < ks; ++s) {
                const int koff_nn = sK[4 * min(s + 2, ks - 1) + kg] + hb0;
                read_b(koff, 1, b1);
                const int sn = min(s + 1, ks - 1);
                v8 an[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) an[nt] = *reinterpret_cast<const v8*>(sW + ((size_t)(sn * NT + nt) * 64 + lane) * 16);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HB; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[j][nt] = Mfma<T>::run(a[nt], b0[j], acc[j][nt]);
                __builtin_amdgcn_sched_barrier(0);
                read_b(koff_n, 0, b0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HB; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[HB + j][nt] = Mfma<T>::run(a[nt], b1[j], acc[HB + j][nt]);
                __builtin_amdgcn_sched_barrier(0);
                koff = koff_n;
                koff_n = koff_nn;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) a[nt] = an[nt];
            }
        }
        LPSTAMP(q4);
        // ---- epilogue of this box
        int n, d0, h0, w0;
        box_of(tile, n, d0, h0, w0);
        const bool full = d0 + TD <= p.D && h0 + TH <= p.H && w0 + BW <= p.W;        // uniform
        char* obase = reinterpret_cast<char*>(out + ((((size_t)n * p.D + d0 + wave) * p.H + h0) * p.W + w0) * p.out_cs);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cb = (nt0 + nt) * 16 + 4 * kg;
            if (cb >= p.nout_p) continue;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) {
                bv.x = cb + 0 < p.nbias ? p.bias[cb + 0] : 0.f; bv.y = cb + 1 < p.nbias ? p.bias[cb + 1] : 0.f;
                bv.z = cb + 2 < p.nbias ? p.bias[cb + 2] : 0.f; bv.w = cb + 3 < p.nbias ? p.bias[cb + 3] : 0.f;
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                int th, tw;
                vox_of(ct, th, tw);
                if (full || (d0 + wave < p.D && h0 + th < p.H && w0 + tw < p.W)) {
                    const float4 o = rnd4<T>(make_float4(acc[ct][nt][0] + bv.x, acc[ct][nt][1] + bv.y, acc[ct][nt][2] + bv.z,
                                                         acc[ct][nt][3] + bv.w));
                    // byte offset inside the box: precomputed per lane (eoff0) + a per-tile multiple of the row pitch
                    const unsigned eo = eoff0 + (unsigned)((BW == 32 ? (ct >> 1) : (BW == 16 ? ct : 2 * ct)) * orow_b) +
                                        (unsigned)((BW == 32 ? (ct & 1) * 16 : 0) * vox_b) + (unsigned)(nt * 16 * (int)sizeof(T));
                    st4<T>(reinterpret_cast<T*>(obase + eo), o);
                    s1[nt][0] += o.x; s1[nt][1] += o.y; s1[nt][2] += o.z; s1[nt][3] += o.w;
                    s2[nt][0] += o.x * o.x; s2[nt][1] += o.y * o.y; s2[nt][2] += o.z * o.z; s2[nt][3] += o.w * o.w;
                }
            }
        }
#ifdef CTU_LP_STAMP
        LPSTAMP(q5);
        ph_[0] += q1 - q0; ph_[1] += q2 - q1; ph_[2] += q3 - q2; ph_[3] += q4 - q3; ph_[4] += q5 - q4;
        q0 = q5; ++nst;
#endif
    }
#ifdef CTU_LP_STAMP
    if (g_lp_stamp_out && tid == 0) {
        unsigned long long* o_ = g_lp_stamp_out + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 6;
        for (int k_ = 0; k_ < 5; ++k_) o_[k_] = ph_[k_];
        o_[5] = nst;
    }
#endif
    if (p.stats) {                                                  // ONE BatchNorm partial row [2][nout_p] per block
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a1 = s1[nt][r], a2 = s2[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
                if (m == 0) {
                    sRed[((wave * NT + nt) * 2 + 0) * 16 + 4 * kg + r] = a1;
                    sRed[((wave * NT + nt) * 2 + 1) * 16 + 4 * kg + r] = a2;
                }
            }
        __syncthreads();
        if (tid < NT * 32) {
            const int nt = tid >> 5, which = (tid >> 4) & 1, c = tid & 15;
            const int ch = (nt0 + nt) * 16 + c;
            if (ch < p.nout_p) {
                float sx = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) sx += sRed[((wv * NT + nt) * 2 + which) * 16 + c];
                st_row(p.tail.counter != nullptr, p.stats + (size_t)blockIdx.x * 2 * p.nout_p + which * p.nout_p + ch, sx);
            }
        }
        if (p.tail.counter) bn_fwd_tail(p.tail, p.stats, gridDim.x, p.nout_p, gridDim.x * gridDim.y);
    }
}

// ---- "pair" variant: k = 3 layers with 8 padded channels on BOTH sides at volumes at least 32 wide (every 128^3 conv of the
// shipped nets but the first; forward and data gradient).  A 16-byte voxel leaves no room for per-voxel work, so:
//   * rows = (w-shift s, c_out), columns = 16 voxel PAIRS of a 32-voxel row, a K-step = the 4 w offsets of one (kd, kh) row x 8
//     channels: all 16 MFMA rows and all 64 epilogue lanes carry real outputs (the plain tile pads 8 channels to 16: half the
//     lanes of its stores idle), 9 K-steps per pair tile instead of 7 per 16 voxels;
//   * a lane's fragment address = its pair's first voxel + kw' (16 bytes apart: consecutive lanes read overlapping, conflict-
//     free 16-byte slots) + a compile-time (plane, row) offset -- no offset table, no index arithmetic in the K loop;
//   * the 10 halo rows of a kd plane are read once and feed the 3 kh taps of the 8 output rows (30 + 9 fragment reads per 72 MFMAs);
//   * the BatchNorm vectors of the 8 input channels sit in registers; box 4 x 8 x 32, three blocks per CU, persistent.
constexpr int LPP_TH = 8, LPP_BW = 32, LPP_HH = LPP_TH + 2, LPP_HW = LPP_BW + 2, LPP_HV = 6 * LPP_HH * LPP_HW;
constexpr int LPP_NI = (LPP_HV + 255) / 256;
constexpr size_t LPP_LDS = 256 + (size_t)LPP_HV * 16 + 9 * 1024;

// OUT32 (the first layer's data gradient, ctu_lp_conv3d_first_bwd_data_pair): the first p.nbias (<= 4) output channels are
// written as float32 PLANES (the network input's NCDHW layout) instead of the 16-bit channels-last tensor; no statistics.
template <class T, bool OUT32 = false>
__global__ __launch_bounds__(256, 3) void lp_conv_fwd_pair_kernel(LpConvP p, int ntiles, int tiles_per_block) {
    typedef typename Vec<T>::v8 v8;
    constexpr int TH = LPP_TH, BW = LPP_BW, HH = LPP_HH, HW = LPP_HW, HV = LPP_HV, NI = LPP_NI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sRed = reinterpret_cast<float*>(smem);                   // [4 waves][2][8]
    unsigned char* sIn = smem + 256;
    unsigned char* sW = sIn + (size_t)HV * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kg = lane >> 4;
    const T* in = reinterpret_cast<const T*>(p.in);
    T* out = reinterpret_cast<T*>(p.out);
    const bool xf = p.scale != nullptr;
    for (int i = tid; i < 9 * 64; i += 256)
        *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = reinterpret_cast<const uint4*>(p.wp)[i];
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = xf ? p.scale[j] : 1.f; sh[j] = xf ? p.shift[j] : 0.f; }
    unsigned ioff[NI];
    unsigned live = 0;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + u * 256;
        const bool lv = i < HV;
        const int v = lv ? i : 0;
        const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
        ioff[u] = (unsigned)((((pd * p.H + ph) * p.W + pw) * p.in_cs) * (int)sizeof(T));
        live |= lv ? (1u << u) : 0u;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 raw[NI];
    unsigned okb = 0;
    auto box_of = [&](int t, int& n, int& d0, int& h0, int& w0) {
        const int tx = t % p.tiles_w; t /= p.tiles_w;
        const int ty = t % p.tiles_h; t /= p.tiles_h;
        const int tz = t % p.tiles_d;
        n = t / p.tiles_d; d0 = tz * 4; h0 = ty * TH; w0 = tx * BW;
    };
    auto load_box = [&](int t) {
        int n, d0, h0, w0;
        box_of(t, n, d0, h0, w0);
        if (d0 >= 1 && d0 + 5 <= p.D && h0 >= 1 && h0 + TH + 1 <= p.H && w0 >= 1 && w0 + BW + 1 <= p.W) {       // uniform
            const char* base = reinterpret_cast<const char*>(in + ((((size_t)n * p.D + d0 - 1) * p.H + h0 - 1) * p.W + w0 - 1) * p.in_cs);
#pragma unroll
            for (int u = 0; u < NI; ++u) raw[u] = *reinterpret_cast<const u32x4*>(base + (((live >> u) & 1u) ? ioff[u] : 0u));
            okb = live;
            return;
        }
        okb = 0;
#pragma unroll
        for (int u = 0; u < NI; ++u) {                              // border / ragged box: clamped, branch-free
            const int v = min(tid + u * 256, HV - 1);
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - 1, gh = h0 + ph - 1, gw = w0 + pw - 1;
            const bool ok = ((live >> u) & 1u) && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            const int cd = min(max(gd, 0), p.D - 1), chh = min(max(gh, 0), p.H - 1), cw = min(max(gw, 0), p.W - 1);
            raw[u] = *reinterpret_cast<const u32x4*>(in + ((((size_t)n * p.D + cd) * p.H + chh) * p.W + cw) * p.in_cs);
            okb |= ok ? (1u << u) : 0u;
        }
    };
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(ntiles, tile + tiles_per_block);
    if (tile < tile_end) load_box(tile);
    // lane (pair m, K-quarter kg = w offset): halo voxel 2 m + kg of the row
    const unsigned char* bIn = sIn + (wave * HH * HW + 2 * m + kg) * 16;
    const unsigned char* bW = sW + lane * 16;
    const int orow = p.W * p.out_cs;
    for (; tile < tile_end; ++tile) {
        __syncthreads();                                            // weights visible (first pass) / previous box's readers done
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            if (!((live >> u) & 1u)) continue;
            u32x4 r = raw[u];
            if (!((okb >> u) & 1u)) r = u32x4{0u, 0u, 0u, 0u};
            else if (xf) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&r), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], sc[j], sh[j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&r) = __builtin_convertvector(o, v8);
            }
            *reinterpret_cast<u32x4*>(sIn + (size_t)(tid + u * 256) * 16) = r;
        }
        __syncthreads();
        if (tile + 1 < tile_end) load_box(tile + 1);
        f32x4 acc[TH];
#pragma unroll
        for (int ct = 0; ct < TH; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            v8 rows[HH];
#pragma unroll
            for (int r = 0; r < HH; ++r) rows[r] = *reinterpret_cast<const v8*>(bIn + ((kd * HH + r) * HW) * 16);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const v8 a = *reinterpret_cast<const v8*>(bW + (kd * 3 + kh) * 1024);
#pragma unroll
                for (int ct = 0; ct < TH; ++ct) acc[ct] = Mfma<T>::run(a, rows[ct + kh], acc[ct]);
            }
        }
        // ---- epilogue: lane = (pair m, s = kg >> 1, channel quad kg & 1) of each of the wave's 8 rows
        int n, d0, h0, w0;
        box_of(tile, n, d0, h0, w0);
        const int gd = d0 + wave, gw = w0 + 2 * m + (kg >> 1);
        const bool full = d0 + 4 <= p.D && h0 + TH <= p.H && w0 + BW <= p.W;          // uniform
        if constexpr (OUT32) {
            if ((kg & 1) == 0) {                                    // channels 0..3 of both w-shifts
                float* o32 = reinterpret_cast<float*>(p.out) + (((size_t)n * p.nbias * p.D + gd) * p.H + h0) * p.W + gw;
                const size_t plane = (size_t)p.D * p.H * p.W;
#pragma unroll
                for (int ct = 0; ct < TH; ++ct)
                    if (full || (gd < p.D && h0 + ct < p.H && gw < p.W)) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < p.nbias) o32[c * plane + (size_t)ct * p.W] = acc[ct][c];
                    }
            }
            continue;
        }
        T* obase = out + ((((size_t)n * p.D + gd) * p.H + h0) * p.W + gw) * p.out_cs + (kg & 1) * 4;
#pragma unroll
        for (int ct = 0; ct < TH; ++ct) {
            if (full || (gd < p.D && h0 + ct < p.H && gw < p.W)) {
                const float4 o = rnd4<T>(make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]));
                st4<T>(obase + (size_t)ct * orow, o);
                s1[0] += o.x; s1[1] += o.y; s1[2] += o.z; s1[3] += o.w;
                s2[0] += o.x * o.x; s2[1] += o.y * o.y; s2[2] += o.z * o.z; s2[3] += o.w * o.w;
            }
        }
    }
    if (p.stats) {                                                  // ONE BatchNorm partial row [2][8] per block
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a1 = s1[r], a2 = s2[r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
            a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32);    // the two w-shifts hold the same channels
            if (m == 0 && kg < 2) {
                sRed[(wave * 2 + 0) * 8 + kg * 4 + r] = a1;
                sRed[(wave * 2 + 1) * 8 + kg * 4 + r] = a2;
            }
        }
        __syncthreads();
        if (tid < 16) {
            const int which = tid >> 3, c = tid & 7;
            float sx = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) sx += sRed[(wv * 2 + which) * 8 + c];
            st_row(p.tail.counter != nullptr, p.stats + (size_t)blockIdx.x * 16 + which * 8 + c, sx);
        }
        if (p.tail.counter) bn_fwd_tail(p.tail, p.stats, gridDim.x, 8, gridDim.x);
    }
}

inline bool lp_use_pair(int k, int rin_p, int nout_p, int W) { return k == 3 && rin_p == 8 && nout_p == 8 && W >= 32; }
inline int lp_pair_ntiles(int N, int D, int H, int W) { return N * ceil_div(D, 4) * ceil_div(H, LPP_TH) * ceil_div(W, LPP_BW); }
inline void lp_pair_grid(int ntiles, int* gx, int* tpb) {
    int g = 768;                                                    // 3 blocks per CU
    if (g > ntiles) g = ntiles;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

struct LpBox { int th, bw; };

// box of a forward / data-gradient launch: bigger boxes for thin voxels (see lp_conv_fwd_kernel)
LpBox lp_box(int W, int rin_p, int64_t nvox) {
    const int nch = (rin_p >= LP_SC ? LP_SC : rin_p) >> 3;
    // the deep levels (<= 16^3 voxels): 4x4x8 boxes -- twice the blocks of the 4x8x8 / 4x4x16 boxes (a 16^3 layer with 64
    // output channels: 128 blocks instead of 64) and half the per-block latency chain; these launches never fill the chip
    if (nvox <= 4096) return {4, 8};
    if (W < 16) return {8, 8};
    // (a 4x8x32 box for single-chunk layers was tried: 16 column tiles per wave cost 277 registers = one wave per SIMD, and
    //  the stamps showed every phase serialised in that one wave: 17.6 k cycles per box against 1.8 k of MFMA work)
    if (nch <= 2) return {8, 16};
    return {4, 16};
}

int lp_fill(LpConvP& p, int N, int D, int H, int W, int rin_p) {
    const LpBox bx = lp_box(W, rin_p, (int64_t)N * D * H * W);
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, bx.th); p.tiles_w = ceil_div(W, bx.bw);
    return N * p.tiles_d * p.tiles_h * p.tiles_w;
}

// persistent single-stage kernel: k = 3, one stage, enough boxes to give every CU a few
bool lp_use_persist(int k, int rin_p, int ntiles) { return k == 3 && rin_p <= LP_SC && ntiles >= 1024; }

void lp_persist_grid(int ntiles, int* gx, int* tpb) {
    int g = 512;                                                    // 2 blocks per CU
    if (g > ntiles) g = ntiles;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

int lp_voxel_stride(int rin_p) { return rin_p >= 16 ? (rin_p >= LP_SC ? LP_SC : rin_p) * 2 + 16 : 16; }

template <class T, int KS, int NT, int TH, int BW, bool WG>
int lp_conv_launch_box_wg(LpConvP& p, int ntiles, hipStream_t st) {
    constexpr int PK = (KS - 1) / 2;
    const int hv = (4 + 2 * PK) * (TH + 2 * PK) * (BW + 2 * PK);
    const int nch_max = (p.rin_p >= LP_SC ? LP_SC : p.rin_p) >> 3;
    const int ks_max = lp_ksteps(KS * KS * KS, nch_max);
    // one weight group (GK K-steps) behind the box; none when the fragments come through the register ring
    const size_t wlds = WG ? (size_t)(KS == 3 ? ks_max : (ks_max < 25 ? ks_max : 25)) * NT * 1024 : 0;
    const size_t lds = 2048 + 256 + 4 * NT * 32 * 4 + (size_t)hv * p.S + wlds;
    CTU_REQUIRE(lds <= 160 * 1024, "lp_conv3d_fwd: LDS box of %zu bytes", lds);
    const int n16 = (p.nout_p + 15) >> 4;
    const dim3 grid(ntiles, ceil_div(n16, NT));
    // the dynamic-LDS limit is raised only for launches that need more than the default 64 KB, and only to what they need
    static size_t raised = 64 * 1024;
    if (lds > raised) {
        CTU_REQUIRE(hipFuncSetAttribute((const void*)lp_conv_fwd_kernel<T, KS, NT, TH, BW, WG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                    "lp_conv3d_fwd: cannot raise the dynamic LDS limit");
        raised = lds;
    }
    lp_conv_fwd_kernel<T, KS, NT, TH, BW, WG><<<grid, 256, lds, st>>>(p);
    CTU_CHECK_LAUNCH("lp_conv3d_fwd");
    return CTU_OK;
}

// k = 3: the stage's weights always sit in LDS.  k = 5: weight groups through LDS when a K-step carries few MFMAs and the box
// already limits the block to one or two per CU (one out tile per block, >= 2 chunks per voxel); the register ring otherwise.
template <class T, int KS, int NT, int TH, int BW>
int lp_conv_launch_box(LpConvP& p, int ntiles, hipStream_t st) {
    if constexpr (KS == 3) return lp_conv_launch_box_wg<T, KS, NT, TH, BW, true>(p, ntiles, st);
    else {
        const int nch_max = (p.rin_p >= LP_SC ? LP_SC : p.rin_p) >> 3;
        if (NT == 1 && nch_max >= 2) return lp_conv_launch_box_wg<T, KS, NT, TH, BW, true>(p, ntiles, st);
        return lp_conv_launch_box_wg<T, KS, NT, TH, BW, false>(p, ntiles, st);
    }
}

template <class T, int NT, int TH, int BW>
int lp_conv_launch_persist(LpConvP& p, int ntiles, hipStream_t st) {
    const int hv = 6 * (TH + 2) * (BW + 2);
    const size_t lds = 2048 + 256 + 4 * NT * 32 * 4 + (size_t)hv * p.S + (size_t)lp_ksteps(27, p.rin_p >> 3) * NT * 1024;
    CTU_REQUIRE(lds <= 160 * 1024, "lp_conv3d_fwd: LDS box of %zu bytes", lds);
    int gx, tpb;
    lp_persist_grid(ntiles, &gx, &tpb);
    const int n16 = (p.nout_p + 15) >> 4;
    static size_t raised = 64 * 1024;
    if (lds > raised) {
        CTU_REQUIRE(hipFuncSetAttribute((const void*)lp_conv_fwd_p1_kernel<T, NT, TH, BW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) == hipSuccess, "lp_conv3d_fwd: cannot raise the dynamic LDS limit");
        raised = lds;
    }
    lp_conv_fwd_p1_kernel<T, NT, TH, BW><<<dim3(gx, ceil_div(n16, NT)), 256, lds, st>>>(p, ntiles, tpb);
    CTU_CHECK_LAUNCH("lp_conv3d_fwd (persistent)");
    return CTU_OK;
}

template <class T, int KS, int NT>
int lp_conv_launch(LpConvP& p, int ntiles, hipStream_t st) {
    const LpBox bx = lp_box(p.W, p.rin_p, (int64_t)p.N * p.D * p.H * p.W);
    if constexpr (KS == 3) {
        if (lp_use_persist(3, p.rin_p, ntiles)) {
            if (bx.bw == 32) return lp_conv_launch_persist<T, NT, 8, 32>(p, ntiles, st);
            if (bx.th == 8 && bx.bw == 16) return lp_conv_launch_persist<T, NT, 8, 16>(p, ntiles, st);
            if (bx.bw == 16) return lp_conv_launch_persist<T, NT, 4, 16>(p, ntiles, st);
        }
    }
    if (bx.bw == 32) return lp_conv_launch_box<T, KS, NT, 8, 32>(p, ntiles, st);
    if (bx.bw == 8 && bx.th == 4) {
        if constexpr (KS == 3) {
            const size_t lds = 2048 + 256 + 4 * NT * 32 * 4 + (size_t)(6 * 6 * 10) * p.S + (size_t)27 * NT * 1024;
            const int n16 = (p.nout_p + 15) >> 4;
            static size_t raised = 64 * 1024;
            if (lds > raised) {
                CTU_REQUIRE(hipFuncSetAttribute((const void*)lp_conv_fwd_small_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)lds) == hipSuccess, "lp_conv3d_fwd: cannot raise the dynamic LDS limit");
                raised = lds;
            }
            lp_conv_fwd_small_kernel<T, NT><<<dim3(ntiles, ceil_div(n16, NT)), 256, lds, st>>>(p);
            CTU_CHECK_LAUNCH("lp_conv3d_fwd (deep level)");
            return CTU_OK;
        }
        return lp_conv_launch_box<T, KS, NT, 4, 8>(p, ntiles, st);
    }
    if (bx.bw == 8) return lp_conv_launch_box<T, KS, NT, 8, 8>(p, ntiles, st);
    if (bx.th == 8) return lp_conv_launch_box<T, KS, NT, 8, 16>(p, ntiles, st);
    return lp_conv_launch_box<T, KS, NT, 4, 16>(p, ntiles, st);
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct LpWgP {
    const void* x;
    const void* g;
    const float* scale;
    const float* shift;
    float* ws;
    int x_cs, cin_p, relu, g_cs, cout_p;
    int N, D, H, W, tiles_d, tiles_h, tiles_w, ntiles;
    // lazy BatchNorm + ReLU backward (lp_conv_wgrad_kernel<.., LZ = true>; conv3d.hip's conv3d_wgrad_k3s_kernel<.., LZ> for 16-bit
    // tensors): g is the gradient w.r.t. the ACTIVATED output, lz_y the raw output (geometry and stride of g), lz_coef
    // bn_bwd_finalize's [5][lz_cp] rows; the raw-output gradient is rounded to the storage type and written to lz_out
    const void* lz_y;
    void* lz_out;
    const float* lz_scale;
    const float* lz_shift;
    const float* lz_coef;
    int lz_cp;
};

constexpr int WG_SX = 32;      // LDS bytes per voxel of the 16-channel images

// (w-shift, channel) tiles for 8-channel sides (SM / SN = 2), as conv3d_wgrad_k3s_kernel / _k5s_kernel of the fp32 path: the
// second 8-channel half of the 16-channel LDS image holds the NEIGHBOUR voxel's channels -- X image half 1 = x(v + e_w),
// G image half 1 = g(u - e_w) -- so the unchanged 16 x 16 MFMA tile carries rows (s, ci), columns (s', co) and entry
// [(s, ci), (s', co)] of the accumulator of w offset t0 is dW[kw = t0 + s + s'].  Per (kd, kh) row QN offsets t0 instead of KS
// taps.  An entry with s' = 1 misses the term of the volume's last w column, which multiplies x at w = W - 1 + kw - pad: zero
// padding only for kw >= pad + 1, so smaller kw always come from s' = 0 entries.
__host__ __device__ constexpr int lp_wg_qn(int KS, int SM, int SN) {
    return KS == 3 ? ((SM == 2 && SN == 2) ? 1 : ((SM == 1 && SN == 1) ? 3 : 2))
                   : ((SM == 2 && SN == 2) ? 2 : ((SM == 1 && SN == 1) ? 5 : (SN == 2 ? 4 : 3)));
}
__host__ __device__ constexpr int lp_wg_t0(int KS, int SM, int SN, int q) {
    return (SM == 1 && SN == 1) ? q : (KS == 3 ? q : (SM == 2 && SN == 2 ? 2 * q : (SN == 2 ? q : (q == 2 ? 3 : 2 * q))));
}
// which (q, s, s') entry supplies tap kw (exactly one per kw): returns kw or -1
__host__ __device__ constexpr int lp_wg_entry_kw(int KS, int SM, int SN, int q, int sm, int sn) {
    const int pad = (KS - 1) / 2, kw = lp_wg_t0(KS, SM, SN, q) + sm + sn;
    if (kw >= KS) return -1;
    if (sn == 1 && kw < pad + 1) return -1;
    if (SM == 1 && SN == 1) return kw;
    if (KS == 3) {
        if (SM == 2 && SN == 2) return (sm == 0 && sn == 1) ? -1 : kw;              // t0 = 0: kw 0, 1 (s = 1), 2
        if (SN == 2) return kw;                                                     // t0 = 0: kw 0; t0 = 1: kw 1, 2
        return (q == 1 && sm == 0) ? -1 : kw;                                       // t0 = 0: kw 0, 1; t0 = 1: kw 2
    }
    if (SM == 2 && SN == 2) return (sm == 0 && sn == 1) ? -1 : kw;                  // t0 = 0: kw 0, 1; t0 = 2: kw 2, 3 (s = 1), 4
    if (SN == 2) return (sn == 0 && q == 3) ? -1 : kw;                              // t0 = q: kw 0, 1, 2 from s' = 0; 3, 4 from s' = 1
    return (q == 2 && sm == 0) ? -1 : kw;                                           // t0 = 0, 2, 3: kw (0, 1), (2, 3), 4
}

// Block (x, y = (ci tile, co tile), z = kd plane for k = 5): accumulates dW[tap][16 ci][16 co] over a contiguous range of
// boxes (4 x TH x BW voxels, 16 K-steps of 32 voxels, 4 per wave) and writes ONE slab.
// UP = 2: the weight gradient of the FUSED decoder up-convolution with 8 padded output channels (upconv_lp.hip; the 16-bit
// twin of conv3d_wgrad_k3s_kernel<1, 1, 2>): x = COARSE activations, g = the fine-grid gradient (channel stride 8) read at one
// (p_d, p_h) output parity per block (blockIdx.y = par4 * n_ci + ci tile); the gradient image's 16 columns are (w-parity,
// c_out) -- the fine voxels 2w, 2w + 1 are 32 contiguous bytes -- and the 12 taps are (dz, dy, dxx) of the parity's sub-cube of
// the coarse halo: slabs [12][16 ci][16 = (p_w, co)] in upconv_wgrad_reduce_pw_kernel's layout.
// LZ: lazy BatchNorm + ReLU backward -- g is the gradient w.r.t. the ACTIVATED output; the raw-output gradient
//   gy = [y sc + sh > 0] k0 g + (A y + B)   (fp32 arithmetic, rounded once to the storage type)
// is formed when the gradient box is written to LDS, used for dW, and stored to p.lz_out for the data-gradient kernel, so the
// separate in-place pass over the layer disappears.  Volumes that are multiples of the box only (host check).
template <class T, int KS, int BW, int SM, int SN, int UP = 0, bool LZ = false>
__global__ __launch_bounds__(256) void lp_conv_wgrad_kernel(LpWgP p, int tiles_per_block) {
    typedef typename Vec<T>::v8 v8;
    static_assert(!UP || (KS == 3 && SM == 1 && SN == 1), "fused up-convolution: full 16 x 16 tile, k = 3 halo");
    static_assert(!LZ || KS == 3, "lazy BatchNorm backward: k = 3 layers");
    constexpr int PK = (KS - 1) / 2;
    constexpr int QN = UP ? 3 : lp_wg_qn(KS, SM, SN), ROWS = UP ? 4 : ((KS == 3) ? 9 : KS);
    constexpr int NTAP = ROWS * QN;                                 // accumulators per block (k = 5: one kd plane)
    constexpr int RPK = 32 / BW, TD = 4, TH = 4 * RPK;
    constexpr int HD = (KS == 3) ? TD + 2 : TD, HH = TH + 2 * PK, HW = BW + 2 * PK, HV = HD * HH * HW, NV = TD * TH * BW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sXf = reinterpret_cast<float*>(smem);                    // [2][16]
    unsigned char* sX = smem + 128;                                 // haloed input image, 16 channels
    unsigned char* sG = sX + (size_t)HV * WG_SX;                    // gradient image, 16 channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15, q = i >> 2, pc = i & 3;
    const int nco = (p.cout_p + 15) >> 4, nci_up = p.cin_p >> 4;
    const int cit = UP ? blockIdx.y % nci_up : blockIdx.y / nco, cot = UP ? 0 : blockIdx.y % nco;
    const int upz = UP ? (blockIdx.y / nci_up) >> 1 : 0, upy = UP ? (blockIdx.y / nci_up) & 1 : 0;      // UP: this block's (p_d, p_h)
    const int gH = UP ? 2 * p.H : p.H, gW = UP ? 2 * p.W : p.W;     // the gradient grid's row strides
    const int kd = (KS == 3) ? 0 : blockIdx.z;                      // k = 5: this block's kd plane (halo rows d0 + td + kd - PK)
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* gr = reinterpret_cast<const T*>(p.g);
    const bool xf = p.scale != nullptr;
    if (tid < 32) {
        const int c = cit * 16 + (tid & 15);
        float v = (tid < 16) ? 1.f : 0.f;
        if (xf) v = (c < p.cin_p) ? ((tid < 16) ? p.scale[c] : p.shift[c]) : 0.f;
        sXf[tid] = v;
    }
    // transposed-read addresses: K-step voxel kk = 8 g + 4 r + q (r = 0, 1), columns 4 pc .. 4 pc + 3
    int xa[2], ga[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int kk = 8 * g + 4 * r + q, rr = kk / BW, tw = kk % BW;
        xa[r] = ((wave * HH + rr) * HW + tw) * WG_SX + 8 * pc;      // + K-step row offset + tap offset
        ga[r] = ((wave * TH + rr) * BW + tw) * WG_SX + 8 * pc;
    }
    f32x4 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // live 8-channel halves of the two images (a shifted half is always live)
    const int nchx = SM == 2 ? 2 : min(2, (p.cin_p - cit * 16) >> 3), nchg = (SN == 2 || UP) ? 2 : min(2, (p.cout_p - cot * 16) >> 3);
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(p.ntiles, tile + tiles_per_block);
    // software pipeline: the NEXT box's global loads are in flight (in registers) while this box's taps run
    constexpr int NX = (HV * 2 + 255) / 256, NG = NV * 2 / 256;
    uint4 rx[NX], rg[NG];
    uint4 ry[LZ ? NG : 1];                                          // LZ: the raw outputs beside the gradient items
    unsigned okx = 0, okg = 0;                                      // bit it: X / G item it lies inside the volume
    // per-thread item offsets (bytes, relative to the box's halo origin / first voxel), computed ONCE per block: an interior
    // box then costs one 32-bit add per load instead of ~50 VALU instructions of index arithmetic (the 8-channel layers
    // are VALU-bound otherwise: 16 bytes per voxel leave no room for per-voxel address math)
    unsigned xoff[NX], xlive = 0, glive = 0;
    int goff[NG];                                                   // (signed: the shifted half of the first column sits at w0 - 1)
#pragma unroll
    for (int u = 0; u < NX; ++u) {
        const int it = tid + u * 256, v = it >> 1, c = it & 1;
        const int vv = it < HV * 2 ? v : 0;
        const int pw = vv % HW, t2 = vv / HW, ph = t2 % HH, pd = t2 / HH;
        // SM = 2: half 1 = channels 0..7 of the voxel one step further in w (the last halo column's is never read)
        const bool live = it < HV * 2 && c < nchx && (SM == 1 || c == 0 || pw + 1 < HW);
        const int sw = (SM == 2 && c == 1 && live) ? 1 : 0, cc = (SM == 2 || !live) ? 0 : c;
        xoff[u] = (unsigned)((((pd * p.H + ph) * p.W + pw + sw) * p.x_cs + cit * 16 + cc * 8) * (int)sizeof(T));
        xlive |= live ? (1u << u) : 0u;
    }
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int it = tid + u * 256, v = it >> 1, c = it & 1;
        const bool live = c < nchg;
        const int tw = v % BW, t2 = v / BW, th = t2 % TH, td = t2 / TH;
        // SN = 2: half 1 = channels 0..7 of the voxel one step BACK in w
        const int sw = (SN == 2 && c == 1) ? -1 : 0, cc = (SN == 2 || !live) ? 0 : c;
        if (UP) goff[u] = (((2 * td * gH + 2 * th) * gW + 2 * tw + c) * p.g_cs) * (int)sizeof(T);      // half c = fine voxel 2 tw + c
        else goff[u] = (((td * p.H + th) * p.W + tw + sw) * p.g_cs + cot * 16 + cc * 8) * (int)sizeof(T);
        glive |= live ? (1u << u) : 0u;
    }
    const ptrdiff_t lz_ydelta = LZ ? reinterpret_cast<const char*>(p.lz_y) - reinterpret_cast<const char*>(p.g) : 0;
    // LZ: every gradient item of a thread carries the same 8 channels (item parity = thread parity; both halves of a shifted /
    // w-parity image hold channels 0..7): its BatchNorm-backward vectors sit in registers
    float lsc[LZ ? 8 : 1], lsh[LZ ? 8 : 1], lk0[LZ ? 8 : 1], lA[LZ ? 8 : 1], lB[LZ ? 8 : 1];
    if constexpr (LZ) {
        const int cb = (SN == 2 || UP) ? 0 : cot * 16 + (tid & 1) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb + j;
            const bool in = c < p.lz_cp;
            lsc[j] = in ? p.lz_scale[c] : 0.f; lsh[j] = in ? p.lz_shift[c] : 0.f;
            lk0[j] = in ? p.lz_coef[c] : 0.f; lA[j] = in ? p.lz_coef[3 * p.lz_cp + c] : 0.f; lB[j] = in ? p.lz_coef[4 * p.lz_cp + c] : 0.f;
        }
    }
    auto load_box = [&](int tl) {
        int t = tl;
        const int tx = t % p.tiles_w; t /= p.tiles_w;
        const int ty = t % p.tiles_h; t /= p.tiles_h;
        const int tz = t % p.tiles_d;
        const int n = t / p.tiles_d;
        const int d0 = tz * TD, h0 = ty * TH, w0 = tx * BW;
        const int dlo = d0 + ((KS == 3) ? -1 : kd - PK), hlo = h0 - PK, wlo = w0 - PK;
        // uniform: the whole haloed box (and the gradient box) lies inside the volume
        if (dlo >= 0 && dlo + HD <= p.D && hlo >= 0 && hlo + HH <= p.H && wlo >= 0 && wlo + HW <= p.W) {
            const char* xb = reinterpret_cast<const char*>(x + ((((size_t)n * p.D + dlo) * p.H + hlo) * p.W + wlo) * p.x_cs);
            const char* gb = UP ? reinterpret_cast<const char*>(gr + ((((size_t)n * 2 * p.D + 2 * d0 + upz) * gH + 2 * h0 + upy) * gW + 2 * w0) * p.g_cs)
                                : reinterpret_cast<const char*>(gr + ((((size_t)n * p.D + d0) * p.H + h0) * p.W + w0) * p.g_cs);
#pragma unroll
            for (int u = 0; u < NX; ++u) rx[u] = *reinterpret_cast<const uint4*>(xb + xoff[u]);
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const uint4 r = *reinterpret_cast<const uint4*>(gb + (ptrdiff_t)goff[u]);
                rg[u] = ((glive >> u) & 1u) ? r : make_uint4(0u, 0u, 0u, 0u);
                if constexpr (LZ) ry[u] = *reinterpret_cast<const uint4*>(gb + lz_ydelta + (ptrdiff_t)goff[u]);
            }
            okx = xlive; okg = glive;
            return;
        }
        okx = 0; okg = 0;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int it = tid + u * 256, v = it >> 1, c = it & 1;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd + ((KS == 3) ? -1 : kd - PK), gh = h0 + ph - PK, gw = w0 + pw - PK + ((SM == 2 && c == 1) ? 1 : 0);
            const bool ok = it < HV * 2 && c < nchx && (SM == 1 || c == 0 || pw + 1 < HW) && (unsigned)gd < (unsigned)p.D &&
                            (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            // branch-free loads from clamped addresses (see lp_conv_fwd_kernel); zeroed at the LDS write
            const int cd = min(max(gd, 0), p.D - 1), chh = min(max(gh, 0), p.H - 1), cw = min(max(gw, 0), p.W - 1);
            rx[u] = *reinterpret_cast<const uint4*>(x + ((((size_t)n * p.D + cd) * p.H + chh) * p.W + cw) * p.x_cs + cit * 16 +
                                                    ((SM == 1 && c < nchx) ? c : 0) * 8);
            okx |= ok ? (1u << u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int it = tid + u * 256, v = it >> 1, c = it & 1;
            const int tw = v % BW, t2 = v / BW, th = t2 % TH, td = t2 / TH;
            const int gd = d0 + td, gh = h0 + th, gw = w0 + tw - ((SN == 2 && c == 1) ? 1 : 0);
            const bool ok = c < nchg && gd < p.D && gh < p.H && gw >= 0 && gw < p.W;
            const int cd = min(gd, p.D - 1), chh = min(gh, p.H - 1), cw = min(max(gw, 0), p.W - 1);
            const size_t ge = UP ? ((((size_t)n * 2 * p.D + 2 * cd + upz) * gH + 2 * chh + upy) * gW + 2 * cw + c) * p.g_cs
                                 : ((((size_t)n * p.D + cd) * p.H + chh) * p.W + cw) * p.g_cs + cot * 16 + ((SN == 1 && c < nchg) ? c : 0) * 8;
            const uint4 r = *reinterpret_cast<const uint4*>(gr + ge);
            rg[u] = ok ? r : make_uint4(0u, 0u, 0u, 0u);
            if constexpr (LZ) ry[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.lz_y) + ge);
            okg |= ok ? (1u << u) : 0u;
        }
    };
    if (tile < tile_end) load_box(tile);
    // every X item of a thread is the same 8-channel half (item parity = thread parity; a shifted half holds channels 0..7
    // again): its BatchNorm vectors go to registers once instead of 16 LDS reads per item
    float wxs[8], wxh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cit * 16 + (SM == 2 ? 0 : (tid & 1) * 8) + j;
        wxs[j] = xf ? (c < p.cin_p ? p.scale[c] : 0.f) : 1.f;
        wxh[j] = xf ? (c < p.cin_p ? p.shift[c] : 0.f) : 0.f;
    }
    for (; tile < tile_end; ++tile) {
        __syncthreads();                                            // the previous box's readers are done
        // ---- registers -> LDS images: X with the lazy BatchNorm + ReLU (zero outside the volume), G as it is
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int it = tid + u * 256;
            if (it >= HV * 2) continue;
            uint4 r = rx[u];
            if (!((okx >> u) & 1u)) r = make_uint4(0u, 0u, 0u, 0u);
            else if (xf) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&r), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], wxs[j], wxh[j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&r) = __builtin_convertvector(o, v8);
            }
            *reinterpret_cast<uint4*>(sX + (it >> 1) * WG_SX + (it & 1) * 16) = r;
        }
        if constexpr (LZ) {
            // the registers hold box `tile`: its raw-output gradient (inside the volume; zeros stay zeros), also written out --
            // an in-volume item's address is the plain box formula (the clamped loads only differ outside the volume)
            int t = tile;
            const int tx = t % p.tiles_w; t /= p.tiles_w;
            const int ty = t % p.tiles_h; t /= p.tiles_h;
            const int tz = t % p.tiles_d;
            const int n = t / p.tiles_d;
            const int d0 = tz * TD, h0 = ty * TH, w0 = tx * BW;
            T* lzo = reinterpret_cast<T*>(p.lz_out);
            char* ob = UP ? reinterpret_cast<char*>(lzo + ((((size_t)n * 2 * p.D + 2 * d0 + upz) * gH + 2 * h0 + upy) * gW + 2 * w0) * p.g_cs)
                          : reinterpret_cast<char*>(lzo + ((((size_t)n * p.D + d0) * p.H + h0) * p.W + w0) * p.g_cs);
            const bool st_own = SN == 2 ? (tid & 1) == 0 : true;   // a shifted half is the neighbour voxel's value: its own box stores it
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                if (!((okg >> u) & 1u)) continue;
                const f32x8 gq = __builtin_convertvector(*reinterpret_cast<v8*>(&rg[u]), f32x8);
                const f32x8 yq = __builtin_convertvector(*reinterpret_cast<v8*>(&ry[u]), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    o[j] = fmaf(lk0[j], (fmaf(yq[j], lsc[j], lsh[j]) > 0.f) ? gq[j] : 0.f, fmaf(lA[j], yq[j], lB[j]));
                *reinterpret_cast<v8*>(&rg[u]) = __builtin_convertvector(o, v8);
                if (st_own) *reinterpret_cast<uint4*>(ob + (ptrdiff_t)goff[u]) = rg[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int it = tid + u * 256;
            *reinterpret_cast<uint4*>(sG + (it >> 1) * WG_SX + (it & 1) * 16) = rg[u];
        }
        __syncthreads();
        if (tile + 1 < tile_end) load_box(tile + 1);
        // ---- 4 K-steps per wave (td = wave; rows ks * RPK .. of the box)
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {
            const int xo = ks * RPK * HW * WG_SX, go = ks * RPK * BW * WG_SX;
            v8 b;
            {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + ga[0] + go));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + ga[1] + go));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 bb = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                b = *reinterpret_cast<const v8*>(&bb);
            }
            // taps in groups (k = 3: the 9 taps of one kd plane, k = 5: the 5 of one kh row): the NEXT group's transposed
            // fragment reads are issued before this group's MFMAs, so a tap no longer pays its own LDS latency
            constexpr int GT = UP ? 6 : ((KS == 3) ? (QN == 3 ? 9 : 3 * QN) : KS), NGRP = NTAP / GT;
            static_assert(NTAP % GT == 0, "tap groups");
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            auto read_grp = [&](int grp, s16x8 (&fr)[GT]) {
#pragma unroll
                for (int j = 0; j < GT; ++j) {
                    const int t = grp * GT + j, row = t / QN, kw = UP ? t % 3 : lp_wg_t0(KS, SM, SN, t % QN);     // (kw = the w offset t0)
                    // UP: tap (dz, dy, dxx) of the sub-cube that starts at (p_d, p_h, 0): halo rows upz + dz, upy + dy
                    const int kdd = UP ? upz + (row >> 1) : ((KS == 3) ? row / 3 : 0), kh = UP ? upy + (row & 1) : ((KS == 3) ? row % 3 : row);
                    const int to = ((kdd * HH + kh) * HW + kw) * WG_SX;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + xa[0] + xo + to));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + xa[1] + xo + to));
                    fr[j] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            };
            s16x8 f0[GT], f1[GT];
            read_grp(0, f0);
#pragma unroll
            for (int grp = 0; grp < NGRP; ++grp) {
                s16x8 (&cur)[GT] = (grp & 1) ? f1 : f0;
                s16x8 (&nxt)[GT] = (grp & 1) ? f0 : f1;
                if (grp + 1 < NGRP) read_grp(grp + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < GT; ++j)
                    acc[grp * GT + j] = Mfma<T>::run(*reinterpret_cast<const v8*>(&cur[j]), b, acc[grp * GT + j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- cross-wave sum in LDS (one wave after the other: the slab is 27 KB, the images' space is reused), one slab per block
    float* sS = reinterpret_cast<float*>(smem + 128);
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < NTAP; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* e = &sS[t * 256 + (4 * g + r) * 16 + i];        // row = ci 4 g + r, col = co i
                    *e = (wv == 0) ? acc[t][r] : (*e + acc[t][r]);
                }
        }
    }
    __syncthreads();
    float* dst = p.ws + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NTAP * 256);
    for (int e = tid; e < NTAP * 256; e += 256) dst[e] = sS[e];
}

// ---- 8 -> 8 padded channels, k = 3, volumes that are multiples of the 4 x 8 x 32 box: the "pair" weight gradient (the twin of
// lp_conv_fwd_pair_kernel).  Same (w-shift, channel) tile as lp_conv_wgrad_kernel<T, 3, 32, 2, 2> -- rows (s, ci), columns
// (s', co), accumulator (kd, kh) entry = dW[kw = s + s'] -- and the same slabs ([9][16][16] per block, lp_wgrad_reduce_kernel
// <3, 2, 2>), but
//   * the LDS images hold each voxel ONCE (16 bytes): the transposing fragment read takes one address per lane, so the lanes of
//     a shifted half simply read the neighbour voxel (X: v + e_w, G: u - e_w; the gradient rows carry the column left of the
//     box).  Half the global-load and LDS-write instructions of the 32-byte-per-voxel images;
//   * the box is 4 x 8 x 32 (halo 2.0x instead of 2.4x), one d plane per wave: the 8 gradient fragments of the plane stay in
//     registers and every input halo row is read once and used by the 3 kh taps that meet it: 76 fragment reads per 72 MFMAs
//     instead of 160;
//   * 512 persistent blocks (2 per CU, 50 KB of LDS each), the next box's loads in flight while this one's taps run.
// LZ as in lp_conv_wgrad_kernel: the raw-output gradient is formed at the LDS write and stored to p.lz_out.
constexpr int W8_HH = 10, W8_HW = 34, W8_HV = 6 * W8_HH * W8_HW, W8_GW = 33, W8_GV = 32 * W8_GW;
constexpr int W8_LDS = 128 + (W8_HV + W8_GV) * 16;

template <class T, bool LZ>
__global__ __launch_bounds__(256, 2) void lp_wgrad8_kernel(LpWgP p, int tiles_per_block) {
    typedef typename Vec<T>::v8 v8;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));     // (u32x4 arrays end up as stack objects)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem + 128;                                 // [6][10][34] voxels x 16 B
    unsigned char* sG = sX + (size_t)W8_HV * 16;                    // [32 rows][33] voxels x 16 B (column 0 = w0 - 1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15, q = i >> 2, pc = i & 3;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* gr = reinterpret_cast<const T*>(p.g);
    const bool xf = p.scale != nullptr;
    // transposed-read addresses: K-step voxel kk = 8 g + 4 r + q, lane slot pc = (shift, channel quad)
    int xa[2], ga[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int kk = 8 * g + 4 * r + q;
        xa[r] = (kk + (pc >> 1)) * 16 + 8 * (pc & 1);
        ga[r] = (kk + 1 - (pc >> 1)) * 16 + 8 * (pc & 1);
    }
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging items: 8 input items (halo voxel it = tid + 256 u), 4 gradient items (box voxel (d = u, h = tid >> 5, w = tid & 31))
    // and, for the first 32 threads, the gradient row's left neighbour.  Offsets relative to the halo origin / box origin, and
    // which faces of the halo an item sits on (bits d-, d+, h-, h+, w-, w+): a box that touches the volume's border zeroes
    // exactly the items on those faces
    int xoff[8];
    unsigned xface[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int it = tid + u * 256, v = it < W8_HV ? it : 0;
        const int pw = v % W8_HW, t2 = v / W8_HW, ph = t2 % W8_HH, pd = t2 / W8_HH;
        xoff[u] = ((pd * p.H + ph) * p.W + pw) * p.x_cs;
        xface[u] = (pd == 0 ? 1u : 0u) | (pd == 5 ? 2u : 0u) | (ph == 0 ? 4u : 0u) | (ph == W8_HH - 1 ? 8u : 0u) |
                   (pw == 0 ? 16u : 0u) | (pw == W8_HW - 1 ? 32u : 0u) | (it < W8_HV ? 0u : 64u);
    }
    const int gh = tid >> 5, gw = tid & 31;
    const int goff0 = (gh * p.W + gw) * p.g_cs, gstep = p.H * p.W * p.g_cs;                 // item u: goff0 + u gstep
    const int gloff = (((tid >> 3) * p.H + (tid & 7)) * p.W - 1) * p.g_cs;                  // tid < 32: row (d = tid >> 3, h = tid & 7)
    const long long xlim = ((long long)p.N * p.D * p.H * p.W - 1) * p.x_cs;
    const ptrdiff_t lz_ydelta = LZ ? reinterpret_cast<const char*>(p.lz_y) - reinterpret_cast<const char*>(p.g) : 0;
    float lsc[LZ ? 8 : 1], lsh[LZ ? 8 : 1], lk0[LZ ? 8 : 1], lA[LZ ? 8 : 1], lB[LZ ? 8 : 1];
    if constexpr (LZ) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = j < p.lz_cp;
            lsc[j] = in ? p.lz_scale[j] : 0.f; lsh[j] = in ? p.lz_shift[j] : 0.f;
            lk0[j] = in ? p.lz_coef[j] : 0.f; lA[j] = in ? p.lz_coef[3 * p.lz_cp + j] : 0.f; lB[j] = in ? p.lz_coef[4 * p.lz_cp + j] : 0.f;
        }
    }
    float wxs[8], wxh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wxs[j] = xf ? p.scale[j] : 1.f;
        wxh[j] = xf ? p.shift[j] : 0.f;
    }
    u32x4 rx[8], rg[4], rgl, ry[LZ ? 4 : 1], ryl;
    unsigned okx = 0;                                               // bit u: input item u lies inside the volume
    bool okl = false;                                               // the left neighbour column lies inside the volume
    rgl = u32x4{0u, 0u, 0u, 0u}; ryl = rgl;
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(p.ntiles, tile + tiles_per_block);
    auto box_origin = [&](int tl, int& n, int& d0, int& h0, int& w0) {
        int t = tl;
        const int tx = t % p.tiles_w; t /= p.tiles_w;
        const int ty = t % p.tiles_h; t /= p.tiles_h;
        const int tz = t % p.tiles_d;
        n = t / p.tiles_d;
        d0 = tz * 4; h0 = ty * 8; w0 = tx * 32;
    };
    auto load_box = [&](int tl) {
        int n, d0, h0, w0;
        box_origin(tl, n, d0, h0, w0);
        const unsigned bm = (d0 == 0 ? 1u : 0u) | (d0 + 4 == p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) | (h0 + 8 == p.H ? 8u : 0u) |
                            (w0 == 0 ? 16u : 0u) | (w0 + 32 == p.W ? 32u : 0u) | 64u;
        const long long xb = ((((long long)n * p.D + d0 - 1) * p.H + h0 - 1) * p.W + w0 - 1) * p.x_cs;
        okx = 0;
        if (bm == 64u) {                                            // (uniform) interior box
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u == 7 && tid + 7 * 256 >= W8_HV) continue;
                rx[u] = *reinterpret_cast<const u32x4*>(x + xb + xoff[u]);
                okx |= 1u << u;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                long long e = xb + xoff[u];
                e = e < 0 ? 0 : (e > xlim ? xlim : e);              // any readable address: the item is zeroed at the LDS write
                rx[u] = *reinterpret_cast<const u32x4*>(x + e);
                okx |= (xface[u] & bm) ? 0u : (1u << u);
            }
        }
        const long long gb = ((((long long)n * p.D + d0) * p.H + h0) * p.W + w0) * p.g_cs;
        const char* gp = reinterpret_cast<const char*>(gr + gb);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rg[u] = *reinterpret_cast<const u32x4*>(gp + (ptrdiff_t)(goff0 + u * gstep) * (int)sizeof(T));
            if constexpr (LZ) ry[u] = *reinterpret_cast<const u32x4*>(gp + lz_ydelta + (ptrdiff_t)(goff0 + u * gstep) * (int)sizeof(T));
        }
        okl = w0 > 0;
        if (tid < 32 && okl) {
            rgl = *reinterpret_cast<const u32x4*>(gp + (ptrdiff_t)gloff * (int)sizeof(T));
            if constexpr (LZ) ryl = *reinterpret_cast<const u32x4*>(gp + lz_ydelta + (ptrdiff_t)gloff * (int)sizeof(T));
        }
    };
    auto lazy_bn = [&](u32x4& gq_, const u32x4& yq_) {
        const f32x8 gq = __builtin_convertvector(*reinterpret_cast<v8*>(&gq_), f32x8);
        const f32x8 yq = __builtin_convertvector(*reinterpret_cast<const v8*>(&yq_), f32x8);
        f32x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j] = fmaf(lk0[j], (fmaf(yq[j], lsc[j], lsh[j]) > 0.f) ? gq[j] : 0.f, fmaf(lA[j], yq[j], lB[j]));
        *reinterpret_cast<v8*>(&gq_) = __builtin_convertvector(o, v8);
    };
    auto rd_tr = [&](const unsigned char* base, const int (&a)[2]) -> v8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a[0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a[1]));
        const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return *reinterpret_cast<const v8*>(&f);
    };
    if (tile < tile_end) load_box(tile);
    for (; tile < tile_end; ++tile) {
        __syncthreads();                                            // the previous box's readers are done
        // ---- registers -> LDS: X with the lazy BatchNorm + ReLU of its producer (zero outside the volume), G as it is
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int it = tid + u * 256;
            if (u == 7 && it >= W8_HV) continue;
            u32x4 r = rx[u];
            if (!((okx >> u) & 1u)) r = u32x4{0u, 0u, 0u, 0u};
            else if (xf) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&r), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], wxs[j], wxh[j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&r) = __builtin_convertvector(o, v8);
            }
            *reinterpret_cast<u32x4*>(sX + it * 16) = r;
        }
        if constexpr (LZ) {
            int n, d0, h0, w0;
            box_origin(tile, n, d0, h0, w0);
            char* ob = reinterpret_cast<char*>(reinterpret_cast<T*>(p.lz_out) + ((((long long)n * p.D + d0) * p.H + h0) * p.W + w0) * p.g_cs);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                lazy_bn(rg[u], ry[u]);
                *reinterpret_cast<u32x4*>(ob + (ptrdiff_t)(goff0 + u * gstep) * (int)sizeof(T)) = rg[u];
            }
            if (tid < 32 && okl) lazy_bn(rgl, ryl);                 // (the neighbour box stores its own column)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4*>(sG + ((u * 8 + gh) * W8_GW + gw + 1) * 16) = rg[u];
        if (tid < 32) *reinterpret_cast<u32x4*>(sG + tid * W8_GW * 16) = okl ? rgl : u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
        if (tile + 1 < tile_end) load_box(tile + 1);
        // ---- this wave's d plane: 8 gradient fragments in registers, 30 input halo rows x the kh taps that meet them
        v8 gf[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) gf[h] = rd_tr(sG + (wave * 8 + h) * (W8_GW * 16), ga);
#pragma unroll
        for (int hp = 0; hp < W8_HH; ++hp) {
            v8 xr[3];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) xr[kd] = rd_tr(sX + ((wave + kd) * W8_HH + hp) * (W8_HW * 16), xa);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int h = hp - kh;
                    if (h >= 0 && h < 8) acc[kd * 3 + kh] = Mfma<T>::run(xr[kd], gf[h], acc[kd * 3 + kh]);
                }
        }
    }
    // ---- cross-wave sum in LDS, one slab per block (layout of lp_conv_wgrad_kernel<T, 3, 32, 2, 2>)
    float* sS = reinterpret_cast<float*>(smem + 128);
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* e = &sS[t * 256 + (4 * g + r) * 16 + i];        // row = (s, ci) 4 g + r, col = (s', co) i
                    *e = (wv == 0) ? acc[t][r] : (*e + acc[t][r]);
                }
        }
    }
    __syncthreads();
    float* dst = p.ws + (size_t)blockIdx.x * (9 * 256);
    for (int e = tid; e < 9 * 256; e += 256) dst[e] = sS[e];
}

// ---- 16-channel tiles on both sides, k = 3, volumes that are multiples of the 4 x 8 x 32 box: lp_wgrad8_kernel's structure for
// the plain 16 x 16 tile (27 accumulators, one (ci tile, co tile) pair per blockIdx.y).  One d plane per wave: its 8 gradient
// fragments stay in registers, every input halo row is read once per (kd, kw) -- 9 fragments -- and used by the 3 kh taps that
// meet it (196 fragment reads per 216 MFMAs; lp_conv_wgrad_kernel: 448); box 4 x 8 x 32 (halo 2.0x), one block per CU with the
// next box's loads in flight in registers; face-code border handling; the last box along w may be partial (any W >= 32 that is a
// multiple of 8).  16 -> 16 at 128^3: 139 -> 71 us, 64 -> 16 at 64^3: 107 -> 47 us.
constexpr int W16_HV = 6 * 10 * 34, W16_NX = (W16_HV * 2 + 255) / 256, W16_NG = 8;
constexpr int W16_LDS = 128 + (W16_HV + 1024) * 32;

template <class T, bool LZ>
__global__ __launch_bounds__(256, 1) void lp_wgrad16_kernel(LpWgP p, int tiles_per_block) {
    typedef typename Vec<T>::v8 v8;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem + 128;                                 // [6][10][34] voxels x 32 B (16 channels)
    unsigned char* sG = sX + (size_t)W16_HV * 32;                   // [32 rows][32] voxels x 32 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15, q = i >> 2, pc = i & 3;
    const int nco = (p.cout_p + 15) >> 4, cit = blockIdx.y / nco, cot = blockIdx.y % nco, half = tid & 1;
    // live 8-channel halves of the two tiles (a tile's second half may lie beyond the padded channel count)
    const bool xlive = cit * 16 + half * 8 < p.cin_p, glive = cot * 16 + half * 8 < p.cout_p;
    const T* x = reinterpret_cast<const T*>(p.x) + cit * 16 + (xlive ? half * 8 : 0);
    const T* gr = reinterpret_cast<const T*>(p.g) + cot * 16 + (glive ? half * 8 : 0);
    const bool xf = p.scale != nullptr;
    int xa[2], ga[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int kk = 8 * g + 4 * r + q;
        xa[r] = kk * 32 + 8 * pc;
        ga[r] = kk * 32 + 8 * pc;
    }
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging items: input item it = tid + 256 u -> (halo voxel it >> 1, half it & 1 = tid & 1); gradient item -> box voxel
    // (tid >> 1) + 128 u (row = voxel >> 5), same half
    int xoff[W16_NX];
    unsigned xface[W16_NX];
#pragma unroll
    for (int u = 0; u < W16_NX; ++u) {
        const int it = tid + u * 256, v = it < W16_HV * 2 ? (it >> 1) : 0;
        const int pw = v % 34, t2 = v / 34, ph = t2 % 10, pd = t2 / 10;
        xoff[u] = ((pd * p.H + ph) * p.W + pw) * p.x_cs;
        // (bits 8..: the halo column -- the volume's last box along w may be partial, its items are checked against the row end)
        xface[u] = (pd == 0 ? 1u : 0u) | (pd == 5 ? 2u : 0u) | (ph == 0 ? 4u : 0u) | (ph == 9 ? 8u : 0u) |
                   (pw == 0 ? 16u : 0u) | ((it < W16_HV * 2 && xlive) ? 0u : 64u) | ((unsigned)pw << 8);
    }
    const int gv0 = tid >> 1;                                                               // item u: voxel gv0 + 128 u = row 4 u + (gv0 >> 5)
    const int goff0 = (((gv0 >> 5) & 7) * p.W + (gv0 & 31)) * p.g_cs;                       // rows 4 u + (0..3): th = (4 u + r) & 7, td = u >> 1
    const long long xlim = ((long long)p.N * p.D * p.H * p.W - 1) * p.x_cs;
    float wxs[8], wxh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cit * 16 + half * 8 + j;
        wxs[j] = (xf && xlive) ? p.scale[c] : 1.f;
        wxh[j] = (xf && xlive) ? p.shift[c] : 0.f;
    }
    // LZ (lazy BatchNorm + ReLU backward, as in lp_conv_wgrad_kernel): g is the gradient w.r.t. the ACTIVATED output; the raw-
    // output gradient is formed at the LDS write from the raw output y (same geometry) and stored to p.lz_out
    const ptrdiff_t lz_ydelta = LZ ? reinterpret_cast<const char*>(p.lz_y) - reinterpret_cast<const char*>(p.g) : 0;
    float lsc[LZ ? 8 : 1], lsh[LZ ? 8 : 1], lk0[LZ ? 8 : 1], lA[LZ ? 8 : 1], lB[LZ ? 8 : 1];
    if constexpr (LZ) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cot * 16 + half * 8 + j;
            const bool in = c < p.lz_cp;
            lsc[j] = in ? p.lz_scale[c] : 0.f; lsh[j] = in ? p.lz_shift[c] : 0.f;
            lk0[j] = in ? p.lz_coef[c] : 0.f; lA[j] = in ? p.lz_coef[3 * p.lz_cp + c] : 0.f; lB[j] = in ? p.lz_coef[4 * p.lz_cp + c] : 0.f;
        }
    }
    u32x4 rx[W16_NX], rg[W16_NG], ry[LZ ? W16_NG : 1];
    unsigned okx = 0;
    bool gok = false;                                               // the loaded box's gradient items lie inside the row
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(p.ntiles, tile + tiles_per_block);
    auto load_box = [&](int tl) {
        int t = tl;
        const int tx = t % p.tiles_w; t /= p.tiles_w;
        const int ty = t % p.tiles_h; t /= p.tiles_h;
        const int tz = t % p.tiles_d;
        const int n = t / p.tiles_d;
        const int d0 = tz * 4, h0 = ty * 8, w0 = tx * 32;
        const unsigned bm = (d0 == 0 ? 1u : 0u) | (d0 + 4 == p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) | (h0 + 8 == p.H ? 8u : 0u) |
                            (w0 == 0 ? 16u : 0u) | 64u;
        const unsigned wmax = (unsigned)(p.W - w0);                 // halo columns 0 .. wmax lie inside the row (33 = all)
        const long long xb = ((((long long)n * p.D + d0 - 1) * p.H + h0 - 1) * p.W + w0 - 1) * p.x_cs;
        okx = 0;
#pragma unroll
        for (int u = 0; u < W16_NX; ++u) {
            long long e = xb + xoff[u];
            e = e < 0 ? 0 : (e > xlim ? xlim : e);                  // any readable address: an outside item is zeroed at the LDS write
            rx[u] = *reinterpret_cast<const u32x4*>(x + e);
            okx |= ((xface[u] & 0x7fu & bm) || (xface[u] >> 8) > wmax) ? 0u : (1u << u);
        }
        gok = glive && (unsigned)(gv0 & 31) < wmax;                 // this thread's box column lies inside the row
        // (a column beyond the row end reads the row's last voxel instead -- readable -- and is zeroed)
        const T* gb = gr + ((((long long)n * p.D + d0) * p.H + h0) * p.W + w0) * p.g_cs - (gok || !glive ? 0 : ((gv0 & 31) - (int)wmax + 1) * p.g_cs);
#pragma unroll
        for (int u = 0; u < W16_NG; ++u) {
            // row 4 u + r (r = gv0 >> 5): td = u >> 1, th = (u & 1) * 4 + r
            const T* src = gb + goff0 + (((u >> 1) * p.H + (u & 1) * 4) * p.W) * p.g_cs;
            const u32x4 r = *reinterpret_cast<const u32x4*>(src);
            rg[u] = gok ? r : u32x4{0u, 0u, 0u, 0u};
            if constexpr (LZ) ry[u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src) + lz_ydelta);
        }
    };
    auto rd_tr = [&](const unsigned char* base, const int (&a)[2]) -> v8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a[0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a[1]));
        const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return *reinterpret_cast<const v8*>(&f);
    };
    if (tile < tile_end) load_box(tile);
    for (; tile < tile_end; ++tile) {
        __syncthreads();                                            // the previous box's readers are done
#pragma unroll
        for (int u = 0; u < W16_NX; ++u) {
            const int it = tid + u * 256;
            if (u == W16_NX - 1 && it >= W16_HV * 2) continue;
            u32x4 r = rx[u];
            if (!((okx >> u) & 1u)) r = u32x4{0u, 0u, 0u, 0u};
            else if (xf) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&r), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], wxs[j], wxh[j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&r) = __builtin_convertvector(o, v8);
            }
            *reinterpret_cast<u32x4*>(sX + it * 16) = r;
        }
        if constexpr (LZ) {
            if (gok) {                                              // (the registers hold box `tile`: gok is its flag)
                int t = tile;
                const int tx = t % p.tiles_w; t /= p.tiles_w;
                const int ty = t % p.tiles_h; t /= p.tiles_h;
                const int tz = t % p.tiles_d;
                const int n = t / p.tiles_d;
                T* ob = reinterpret_cast<T*>(p.lz_out) + cot * 16 + half * 8 +
                        ((((long long)n * p.D + tz * 4) * p.H + ty * 8) * p.W + tx * 32) * p.g_cs;
#pragma unroll
                for (int u = 0; u < W16_NG; ++u) {
                    const f32x8 gq = __builtin_convertvector(*reinterpret_cast<v8*>(&rg[u]), f32x8);
                    const f32x8 yq = __builtin_convertvector(*reinterpret_cast<v8*>(&ry[u]), f32x8);
                    f32x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        o[j] = fmaf(lk0[j], (fmaf(yq[j], lsc[j], lsh[j]) > 0.f) ? gq[j] : 0.f, fmaf(lA[j], yq[j], lB[j]));
                    *reinterpret_cast<v8*>(&rg[u]) = __builtin_convertvector(o, v8);
                    if (cit == 0) *reinterpret_cast<u32x4*>(ob + goff0 + (((u >> 1) * p.H + (u & 1) * 4) * p.W) * p.g_cs) = rg[u];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < W16_NG; ++u) *reinterpret_cast<u32x4*>(sG + (tid + u * 256) * 16) = rg[u];
        __syncthreads();
        if (tile + 1 < tile_end) load_box(tile + 1);
        // ---- this wave's d plane: 8 gradient fragments in registers; per halo row 9 input fragments (kd, kw) x the kh taps
        v8 gf[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) gf[h] = rd_tr(sG + (wave * 8 + h) * 1024, ga);
#pragma unroll
        for (int hp = 0; hp < 10; ++hp) {
            v8 xr[3][3];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) xr[kd][kw] = rd_tr(sX + (((wave + kd) * 10 + hp) * 34 + kw) * 32, xa);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int h = hp - kh;
                if (h < 0 || h > 7) continue;
#pragma unroll
                for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc[(kd * 3 + kh) * 3 + kw] = Mfma<T>::run(xr[kd][kw], gf[h], acc[(kd * 3 + kh) * 3 + kw]);
            }
        }
    }
    // ---- cross-wave sum in LDS, one slab per block (layout of lp_conv_wgrad_kernel<T, 3, .., 1, 1>)
    float* sS = reinterpret_cast<float*>(smem + 128);
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < 27; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* e = &sS[t * 256 + (4 * g + r) * 16 + i];        // row = ci 4 g + r, col = co i
                    *e = (wv == 0) ? acc[t][r] : (*e + acc[t][r]);
                }
        }
    }
    __syncthreads();
    float* dst = p.ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (27 * 256);
    for (int e = tid; e < 27 * 256; e += 256) dst[e] = sS[e];
}

inline bool lp_wg16_ok(int D, int H, int W, int k, int cin_p, int cout_p) {
    return k == 3 && cin_p >= 16 && cout_p >= 16 && D % 4 == 0 && H % 8 == 0 && W >= 32 && W % 8 == 0;     // (last w box may be partial)
}
inline void lp_wg16_grid(int ntiles, int pairs, int* gx, int* tpb) {
    int g = 256 / pairs;
    if (g < 8) g = 8;
    if (g > ntiles) g = ntiles;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

// the geometries lp_wgrad8_kernel takes
inline bool lp_wg8_ok(int D, int H, int W, int k, int cin_p, int cout_p) {
    return k == 3 && cin_p == 8 && cout_p == 8 && D % 4 == 0 && H % 8 == 0 && W % 32 == 0;
}
inline void lp_wg8_grid(int ntiles, int* gx, int* tpb) {
    int g = ntiles < 512 ? ntiles : 512;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

// dw[co][ci][tap] (torch layout) = sum over the gx slabs of (pair, plane); 16 slab groups x 64 elements per block
template <int KS, int SM, int SN>
__global__ __launch_bounds__(1024) void lp_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Co, int Ci,
                                                              const int32_t* __restrict__ cinv, int cin_p, int cout_p, int gx) {
    constexpr int TAPS = KS * KS * KS, QN = lp_wg_qn(KS, SM, SN), ROWS = (KS == 3) ? 9 : KS, NTAP = ROWS * QN;
    constexpr int NPL = (KS == 3) ? 1 : KS;
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int nco = (cout_p + 15) >> 4, nci = (cin_p + 15) >> 4;
    const int per_pair = NTAP * 256;
    const int el = blockIdx.x * 64 + e;                             // over [plane][pair][tap][ci 16][co 16]
    const int total = NPL * nci * nco * per_pair;
    float s = 0.f;
    if (el < total) {
        const int grp = el / per_pair, within = el % per_pair;      // grp = plane * (nci * nco) + pair
        const float* base = ws + (size_t)grp * gx * per_pair + within;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = part;
        for (; k + 48 < gx; k += 64) {
            s0 += base[(size_t)k * per_pair];
            s1 += base[(size_t)(k + 16) * per_pair];
            s2 += base[(size_t)(k + 32) * per_pair];
            s3 += base[(size_t)(k + 48) * per_pair];
        }
        for (; k < gx; k += 16) s0 += base[(size_t)k * per_pair];
        s = (s0 + s1) + (s2 + s3);
    }
    red[part][e] = s;
    __syncthreads();
    if (part != 0 || el >= total) return;
    float tot = 0.f;
#pragma unroll
    for (int qd = 0; qd < 16; ++qd) tot += red[qd][e];
    const int grp = el / per_pair, within = el % per_pair;
    const int plane = grp / (nci * nco), pair = grp % (nci * nco);
    const int t = within >> 8, row = (within >> 4) & 15, col = within & 15;
    // (shift, channel) tiles: row = (s, ci), col = (s', co); entry (q, s, s') of tap row r is tap kw = t0(q) + s + s' (or unused)
    const int sm = SM == 2 ? row >> 3 : 0, cil = SM == 2 ? row & 7 : row, sn = SN == 2 ? col >> 3 : 0, col_ = SN == 2 ? col & 7 : col;
    const int kw = lp_wg_entry_kw(KS, SM, SN, t % QN, sm, sn), r = t / QN;
    const int cpos = (pair / nco) * 16 + cil, co = (pair % nco) * 16 + col_;
    const int ci = (cpos < cin_p) ? (cinv ? cinv[cpos] : (cpos < Ci ? cpos : -1)) : -1;
    if (kw >= 0 && ci >= 0 && co < Co) dw[((size_t)co * Ci + ci) * TAPS + (plane * ROWS + r) * KS + kw] = tot;
}

int lp_wg_box_w(int W) { return W >= 32 ? 32 : (W >= 16 ? 16 : 8); }

int lp_wg_fill(LpWgP& p, int N, int D, int H, int W) {
    const int bw = lp_wg_box_w(W), th = 4 * (32 / bw);
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, th); p.tiles_w = ceil_div(W, bw);
    p.ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
    return p.ntiles;
}

// persistent blocks per (pair, plane): about 768 blocks in all (3 per CU), at least 1 box each -- and at least LP_WG_MINBOX
// boxes each while that still leaves one block per CU: a block's fixed costs (the first box's exposed loads, the cross-wave
// reduction and its slab, which the reduce kernel reads back) are those of several boxes' MFMAs
#ifndef LP_WG_MINBOX
#define LP_WG_MINBOX 2
#endif
#ifndef LP_WG16
#define LP_WG16 1              // 16-channel tiles, k = 3, box-multiple volumes: lp_wgrad16_kernel (0: the generic kernel)
#endif
#ifndef LP_UPWG4
#define LP_UPWG4 1             // fused up-convolution weight gradient: lp_upwg4_kernel on box-multiple volumes (0: the UP = 2 mode)
#endif
#ifndef LP_WG8
#define LP_WG8 1               // 8 -> 8 padded channels, k = 3, box-multiple volumes: lp_wgrad8_kernel (0: the generic kernel)
#endif
void lp_wg_grid(int ntiles, int groups, int* gx, int* tpb) {
    int g = 768 / groups;
    if (g < 16) g = 16;
    if (g * LP_WG_MINBOX > ntiles && ntiles / LP_WG_MINBOX * groups >= 256) g = ntiles / LP_WG_MINBOX;
    if (g > ntiles) g = ntiles;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

template <class T, int KS, int BW, int SM, int SN, bool LZ = false>
int lp_wgrad_launch(LpWgP& p, int gx, int tpb, int pairs, hipStream_t st) {
    constexpr int PK = (KS - 1) / 2, RPK = 32 / BW, TH = 4 * RPK;
    constexpr int HD = (KS == 3) ? 6 : 4, HV = HD * (TH + 2 * PK) * (BW + 2 * PK), NV = 4 * TH * BW;
    constexpr int NTAP = ((KS == 3) ? 9 : KS) * lp_wg_qn(KS, SM, SN);
    size_t lds = 128 + (size_t)(HV + NV) * WG_SX;
    if (lds < 128 + (size_t)NTAP * 1024) lds = 128 + (size_t)NTAP * 1024;
    // the dynamic-LDS limit is raised only for launches that need more than the default 64 KB, and only to what they need
    static size_t raised = 64 * 1024;
    if (lds > raised) {
        CTU_REQUIRE(hipFuncSetAttribute((const void*)lp_conv_wgrad_kernel<T, KS, BW, SM, SN, 0, LZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                    "lp_conv3d_wgrad: cannot raise the dynamic LDS limit");
        raised = lds;
    }
    lp_conv_wgrad_kernel<T, KS, BW, SM, SN, 0, LZ><<<dim3(gx, pairs, KS == 3 ? 1 : KS), 256, lds, st>>>(p, tpb);
    CTU_CHECK_LAUNCH("lp_conv3d_wgrad");
    return CTU_OK;
}

// fused up-convolution (UP = 2): COARSE boxes 4 x TH x BW, blockIdx.y = (p_d, p_h) parity x input-channel tile
template <class T, int BW, bool LZ = false>
int lp_upwg_launch(LpWgP& p, int gx, int tpb, hipStream_t st) {
    constexpr int RPK = 32 / BW, TH = 4 * RPK, HV = 6 * (TH + 2) * (BW + 2), NV = 4 * TH * BW;
    size_t lds = 128 + (size_t)(HV + NV) * WG_SX;
    if (lds < 128 + (size_t)12 * 1024) lds = 128 + (size_t)12 * 1024;
    static size_t raised = 64 * 1024;
    if (lds > raised) {
        CTU_REQUIRE(hipFuncSetAttribute((const void*)lp_conv_wgrad_kernel<T, 3, BW, 1, 1, 2, LZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                    "lp_upconv_fused_wgrad: cannot raise the dynamic LDS limit");
        raised = lds;
    }
    lp_conv_wgrad_kernel<T, 3, BW, 1, 1, 2, LZ><<<dim3(gx, 4 * (p.cin_p >> 4)), 256, lds, st>>>(p, tpb);
    CTU_CHECK_LAUNCH("lp_upconv_fused_wgrad");
    return CTU_OK;
}

// ---- fused up-convolution weight gradient, all four (p_d, p_h) parities and 32 input channels per block (coarse volumes that
// are multiples of the 4 x 4 x 32 box, no lazy BatchNorm backward).  lp_conv_wgrad_kernel<.., UP = 2> gives every (parity, input-
// channel tile) its own blocks: the coarse halo is staged 4 x (cin_p / 16) times and each block reads every other fine row of
// the gradient (half of every cache line) -- 110 us for the 64^3 -> 128^3 level of UNet() against 50 MB of operands.  Here a
// block stages the coarse halo (32 channels, two 16-channel planes) and the WHOLE fine gradient box once; wave = parity; the
// wave's 16 gradient fragments stay in registers and each of the 25 halo rows a parity touches is read once (3 w offsets x 2
// channel tiles) and used by the (dz, dy) taps that meet it: 332 fragment reads per 384 MFMAs; every wave owns its 24
// accumulators, so there is no cross-wave reduction: the slabs ([12][16][16] per (parity, ci tile), lp_upwg_reduce_kernel's
// layout) are written straight from the registers.  One block per CU (144 KB of LDS), next box's loads in flight in registers.
constexpr int U4_HV = 6 * 6 * 34, U4_NX = (U4_HV * 4 + 255) / 256, U4_NG = 16;
constexpr int U4_LDS = 128 + 2 * U4_HV * 32 + 64 * 1024;

template <class T>
__global__ __launch_bounds__(256, 1) void lp_upwg4_kernel(LpWgP p, int tiles_per_block) {
    typedef typename Vec<T>::v8 v8;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem + 128;                                 // [2 channel tiles][6][6][34] voxels x 32 B
    unsigned char* sG = sX + (size_t)2 * U4_HV * 32;                // [8][8] fine rows x 64 fine voxels x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15, q = i >> 2, pc = i & 3;
    const int pd = wave >> 1, ph = wave & 1;                        // this wave's output parity
    const int cib = blockIdx.y * 32;                                // first input channel of the block
    const T* x = reinterpret_cast<const T*>(p.x) + cib + (tid & 3) * 8;         // (this thread's 8-channel quarter)
    const T* gr = reinterpret_cast<const T*>(p.g);
    const bool xf = p.scale != nullptr;
    const int gH = 2 * p.H, gW = 2 * p.W;
    int xa[2], ga[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int kk = 8 * g + 4 * r + q;
        xa[r] = ((pd * 6 + ph) * 34 + kk) * 32 + 8 * pc;
        ga[r] = (pd * 8 + ph) * 1024 + kk * 32 + 8 * pc;
    }
    f32x4 acc[2][12];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 12; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging items: input item it = tid + 256 u -> (halo voxel it >> 2, 8-channel quarter it & 3 = tid & 3); gradient item ->
    // fine voxel (fd = u >> 1, fh = (u & 1) * 4 + (tid >> 6), fw = tid & 63)
    int xoff[U4_NX];
    unsigned xface[U4_NX];
#pragma unroll
    for (int u = 0; u < U4_NX; ++u) {
        const int it = tid + u * 256, v = it < U4_HV * 4 ? (it >> 2) : 0;
        const int pw = v % 34, t2 = v / 34, phh = t2 % 6, pdd = t2 / 6;
        xoff[u] = ((pdd * p.H + phh) * p.W + pw) * p.x_cs;
        xface[u] = (pdd == 0 ? 1u : 0u) | (pdd == 5 ? 2u : 0u) | (phh == 0 ? 4u : 0u) | (phh == 5 ? 8u : 0u) |
                   (pw == 0 ? 16u : 0u) | (pw == 33 ? 32u : 0u) | (it < U4_HV * 4 ? 0u : 64u);
    }
    const int goff0 = ((tid >> 6) * gW + (tid & 63)) * 8;                                   // + (fd gH + (u & 1) 4) gW 8
    const long long xlim = ((long long)p.N * p.D * p.H * p.W - 1) * p.x_cs;
    float wxs[8], wxh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cib + (tid & 3) * 8 + j;
        wxs[j] = xf ? p.scale[c] : 1.f;
        wxh[j] = xf ? p.shift[c] : 0.f;
    }
    u32x4 rx[U4_NX], rg[U4_NG];
    unsigned okx = 0;
    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(p.ntiles, tile + tiles_per_block);
    auto load_box = [&](int tl) {
        int t = tl;
        const int tx = t % p.tiles_w; t /= p.tiles_w;
        const int ty = t % p.tiles_h; t /= p.tiles_h;
        const int tz = t % p.tiles_d;
        const int n = t / p.tiles_d;
        const int d0 = tz * 4, h0 = ty * 4, w0 = tx * 32;
        const unsigned bm = (d0 == 0 ? 1u : 0u) | (d0 + 4 == p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) | (h0 + 4 == p.H ? 8u : 0u) |
                            (w0 == 0 ? 16u : 0u) | (w0 + 32 == p.W ? 32u : 0u) | 64u;
        const long long xb = ((((long long)n * p.D + d0 - 1) * p.H + h0 - 1) * p.W + w0 - 1) * p.x_cs;
        okx = 0;
        if (bm == 64u) {                                            // (uniform) interior box
#pragma unroll
            for (int u = 0; u < U4_NX; ++u) {
                if (u == U4_NX - 1 && tid + u * 256 >= U4_HV * 4) continue;
                rx[u] = *reinterpret_cast<const u32x4*>(x + xb + xoff[u]);
                okx |= 1u << u;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U4_NX; ++u) {
                long long e = xb + xoff[u];
                e = e < 0 ? 0 : (e > xlim ? xlim : e);              // any readable address: the item is zeroed at the LDS write
                rx[u] = *reinterpret_cast<const u32x4*>(x + e);
                okx |= (xface[u] & bm) ? 0u : (1u << u);
            }
        }
        const T* gb = gr + ((((long long)n * 2 * p.D + 2 * d0) * gH + 2 * h0) * gW + 2 * w0) * 8;
#pragma unroll
        for (int u = 0; u < U4_NG; ++u)
            rg[u] = *reinterpret_cast<const u32x4*>(gb + goff0 + (((u >> 1) * gH + (u & 1) * 4) * gW) * 8);
    };
    auto rd_tr = [&](const unsigned char* base, const int (&a)[2]) -> v8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a[0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a[1]));
        const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return *reinterpret_cast<const v8*>(&f);
    };
    if (tile < tile_end) load_box(tile);
    for (; tile < tile_end; ++tile) {
        __syncthreads();                                            // the previous box's readers are done
#pragma unroll
        for (int u = 0; u < U4_NX; ++u) {
            const int it = tid + u * 256;
            if (u == U4_NX - 1 && it >= U4_HV * 4) continue;
            u32x4 r = rx[u];
            if (!((okx >> u) & 1u)) r = u32x4{0u, 0u, 0u, 0u};
            else if (xf) {
                const f32x8 f = __builtin_convertvector(*reinterpret_cast<v8*>(&r), f32x8);
                f32x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaf(f[j], wxs[j], wxh[j]);
                    o[j] = p.relu ? fmaxf(a, 0.f) : a;
                }
                *reinterpret_cast<v8*>(&r) = __builtin_convertvector(o, v8);
            }
            // plane = channel tile (quarter >> 1), 32 B per voxel
            *reinterpret_cast<u32x4*>(sX + ((tid & 3) >> 1) * (U4_HV * 32) + (it >> 2) * 32 + (tid & 1) * 16) = r;
        }
#pragma unroll
        for (int u = 0; u < U4_NG; ++u)
            *reinterpret_cast<u32x4*>(sG + (((u >> 1) * 8 + (u & 1) * 4 + (tid >> 6)) * 64 + (tid & 63)) * 16) = rg[u];
        __syncthreads();
        if (tile + 1 < tile_end) load_box(tile + 1);
        // ---- this wave's parity: 16 gradient fragments (coarse rows (td, th)), 25 halo rows x 3 w offsets x 2 channel tiles
        v8 gf[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) gf[r] = rd_tr(sG + ((2 * (r >> 2)) * 8 + 2 * (r & 3)) * 1024, ga);
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                v8 xr[2][3];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int dxx = 0; dxx < 3; ++dxx)
                        xr[c][dxx] = rd_tr(sX + c * (U4_HV * 32) + ((a * 6 + b) * 34 + dxx) * 32, xa);
#pragma unroll
                for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const int td = a - dz, th = b - dy;
                        if (td < 0 || td > 3 || th < 0 || th > 3) continue;
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int dxx = 0; dxx < 3; ++dxx)
                                acc[c][(dz * 2 + dy) * 3 + dxx] = Mfma<T>::run(xr[c][dxx], gf[td * 4 + th], acc[c][(dz * 2 + dy) * 3 + dxx]);
                    }
            }
    }
    // ---- slabs: [(parity, ci tile)][block][12][16 ci][16 = (p_w, co)], straight from the registers
    const int n_ci_g = p.cin_p >> 4;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float* dst = p.ws + ((size_t)(wave * n_ci_g + blockIdx.y * 2 + c) * gridDim.x + blockIdx.x) * (12 * 256);
#pragma unroll
        for (int t = 0; t < 12; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[t * 256 + (4 * g + r) * 16 + i] = acc[c][t][r];
    }
}

inline bool lp_upwg4_ok(int D, int H, int W, int cin_p) { return cin_p % 32 == 0 && D % 4 == 0 && H % 4 == 0 && W % 32 == 0; }
inline void lp_upwg4_grid(int ntiles, int* gx, int* tpb) {
    int g = ntiles < 256 ? ntiles : 256;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

// dW_eff[(pz,py,px)][(dz,dy,dxx-px)][cin_p][8] from the slabs [12 taps][16 ci][16 = (px, co)] (= conv3d.hip's
// upconv_wgrad_reduce_pw_kernel for the 16-bit kernel's slabs)
__global__ __launch_bounds__(1024) void lp_upwg_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dweff, int cin_p, int n_ci_g,
                                                             int gx) {
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int el = blockIdx.x * 64 + e;                             // element of the [12][16][16] slab
    const int yb = blockIdx.y;                                      // par4 * n_ci_g + cig
    float s = 0.f;
    const float* src = ws + (size_t)yb * gx * (12 * 256) + el;
    for (int k = part; k < gx; k += 16) s += src[(size_t)k * (12 * 256)];
    red[part][e] = s;
    __syncthreads();
    if (part != 0) return;
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][e];
    const int t = el >> 8, i = (el >> 4) & 15, j = el & 15;
    const int cig = yb % n_ci_g, par4 = yb / n_ci_g;
    const int px = j >> 3, co = j & 7, jx = t % 3 - px, rp = cig * 16 + i;
    if ((jx == 0 || jx == 1) && rp < cin_p)
        dweff[((size_t)((par4 * 2 + px) * 8 + (t / 3) * 2 + jx) * cin_p + rp) * 8 + co] = tot;
}

// 8-channel sides of volumes at least 16 wide take the (w-shift, channel) tiles
inline int lp_wg_sm(int W, int cin_p) { return (W >= 16 && cin_p == 8) ? 2 : 1; }
inline int lp_wg_sn(int W, int cout_p) { return (W >= 16 && cout_p == 8) ? 2 : 1; }

template <class T, int KS, int BW, bool LZ = false>
int lp_wgrad_launch_s(LpWgP& p, int gx, int tpb, int pairs, hipStream_t st) {
    if constexpr (BW >= 16) {
        const int sm = lp_wg_sm(p.W, p.cin_p), sn = lp_wg_sn(p.W, p.cout_p);
        if (sm == 2 && sn == 2) return lp_wgrad_launch<T, KS, BW, 2, 2, LZ>(p, gx, tpb, pairs, st);
        if (sm == 2) return lp_wgrad_launch<T, KS, BW, 2, 1, LZ>(p, gx, tpb, pairs, st);
        if (sn == 2) return lp_wgrad_launch<T, KS, BW, 1, 2, LZ>(p, gx, tpb, pairs, st);
    }
    return lp_wgrad_launch<T, KS, BW, 1, 1, LZ>(p, gx, tpb, pairs, st);
}

template <class T, int KS>
int lp_wgrad_dispatch(LpWgP& p, int gx, int tpb, int pairs, hipStream_t st) {
    const int bw = lp_wg_box_w(p.W);
    if constexpr (KS == 3) {
        if (p.lz_y) {                                               // lazy BatchNorm backward (16-wide boxes and wider)
            if (bw == 32) return lp_wgrad_launch_s<T, 3, 32, true>(p, gx, tpb, pairs, st);
            return lp_wgrad_launch_s<T, 3, 16, true>(p, gx, tpb, pairs, st);
        }
    }
    if (bw == 32) return lp_wgrad_launch_s<T, KS, 32>(p, gx, tpb, pairs, st);
    if (bw == 16) return lp_wgrad_launch_s<T, KS, 16>(p, gx, tpb, pairs, st);
    return lp_wgrad_launch_s<T, KS, 8>(p, gx, tpb, pairs, st);
}

// lazy BatchNorm backward inside the 16-bit weight-gradient kernel: k = 3, volumes at least 16 wide that are multiples of the box
inline bool lp_wg_lazy_ok(int D, int H, int W, int k) {
    if (k != 3 || W < 16) return false;
    const int bw = lp_wg_box_w(W), th = 4 * (32 / bw);
    return D % 4 == 0 && H % th == 0 && W % bw == 0;
}

}  // namespace

// =================================================================== C ABI
// packed-weight layouts of the 16-bit forward / data-gradient kernels: 0 = [K-step][16-wide out tile][lane][8]; 1 ("pair",
// k = 3 with 8 padded channels on both sides at volumes at least 32 wide) = lp_conv_fwd_pair_kernel's (w-shift, channel) rows.
// ctu_lp_conv3d_layout: the layout the forward kernel wants for a geometry -- pack, num_blocks and forward must agree on it.
extern "C" int ctu_lp_conv3d_layout(int k, int rin_p, int nout_p, int W) { return lp_use_pair(k, rin_p, nout_p, W) ? 1 : 0; }

extern "C" size_t ctu_lp_conv3d_packed_elems(int k, int rin_p, int nout_p) {
    if ((k != 3 && k != 5) || rin_p <= 0 || rin_p % 8 || nout_p <= 0 || nout_p % 8) return 0;
    const size_t n = (size_t)lp_total_ksteps(k * k * k, rin_p) * ((nout_p + 15) >> 4) * 512;
    return (k == 3 && rin_p == 8 && nout_p == 8 && n < (size_t)LP_PAIR_ELEMS) ? (size_t)LP_PAIR_ELEMS : n;      // either layout fits
}

extern "C" int ctu_lp_conv3d_num_blocks(int N, int D, int H, int W, int k, int rin_p, int nout_p, int layout) {
    if (layout == 1) {
        int gx, tpb;
        lp_pair_grid(lp_pair_ntiles(N, D, H, W), &gx, &tpb);
        return gx;
    }
    (void)nout_p;
    LpConvP p;
    const int ntiles = lp_fill(p, N, D, H, W, rin_p);
    if (lp_use_persist(k, rin_p, ntiles) && lp_box(W, rin_p, (int64_t)N * D * H * W).bw >= 16) {
        int gx, tpb;
        lp_persist_grid(ntiles, &gx, &tpb);
        return gx;
    }
    return ntiles;
}

extern "C" int ctu_lp_pack_conv3d_weight(int dtype, const float* w, void* wp, int Co, int Ci, int k, const int32_t* cinv,
                                         int rin_p, int nout_p, int mode, int layout, void* stream) {
    CTU_REQUIRE(w && wp, "lp_pack_conv3d_weight: null pointer");
    CTU_REQUIRE(layout == 0 || (layout == 1 && k == 3 && rin_p == 8 && nout_p == 8), "lp_pack_conv3d_weight: layout %d needs k = 3 and 8 padded channels on both sides", layout);
    if (layout == 1) {
        CTU_REQUIRE(mode == 0 || mode == 1, "lp_pack_conv3d_weight: mode=%d", mode);
        CTU_DISPATCH_LP(dtype, lp_pack_conv_pair_kernel<T><<<ceil_div(LP_PAIR_ELEMS, 256), 256, 0, (hipStream_t)stream>>>(w, (T*)wp, Co, Ci, cinv, mode));
        CTU_CHECK_LAUNCH("lp_pack_conv3d_weight(pair)");
        return CTU_OK;
    }
    CTU_REQUIRE((k == 3 || k == 5) && rin_p > 0 && rin_p % 8 == 0 && nout_p > 0 && nout_p % 8 == 0 && (mode == 0 || mode == 1),
                "lp_pack_conv3d_weight: k=%d rin_p=%d nout_p=%d mode=%d", k, rin_p, nout_p, mode);
    const size_t total = ctu_lp_conv3d_packed_elems(k, rin_p, nout_p);
    const unsigned grid = (unsigned)ceil_div64((int64_t)total, 256);
    hipStream_t st = (hipStream_t)stream;
    CTU_DISPATCH_LP(dtype, {
        if (k == 3) lp_pack_conv_w_kernel<T, 3><<<grid, 256, 0, st>>>(w, (T*)wp, Co, Ci, cinv, rin_p, nout_p, mode);
        else lp_pack_conv_w_kernel<T, 5><<<grid, 256, 0, st>>>(w, (T*)wp, Co, Ci, cinv, rin_p, nout_p, mode);
    });
    CTU_CHECK_LAUNCH("lp_pack_conv3d_weight");
    return CTU_OK;
}

extern "C" int ctu_lp_conv3d_fwd(int dtype, const void* in, int in_cs, int rin_p, const float* in_scale, const float* in_shift,
                                 int in_relu, const void* wp, const float* bias, int nbias, void* out, int out_cs, int nout_p,
                                 float* stats, int N, int D, int H, int W, int k, int layout, const ctu_bn_tail* tail, void* stream) {
    CTU_REQUIRE(in && wp && out, "lp_conv3d_fwd: null pointer");
    CTU_REQUIRE(layout == 0 || (layout == 1 && lp_use_pair(k, rin_p, nout_p, W) && !bias), "lp_conv3d_fwd: layout %d not valid for this geometry", layout);
    CTU_REQUIRE((k == 3 || k == 5) && rin_p > 0 && rin_p % 8 == 0 && nout_p > 0 && nout_p % 8 == 0,
                "lp_conv3d_fwd: k=%d rin_p=%d nout_p=%d", k, rin_p, nout_p);
    CTU_REQUIRE(in_cs >= rin_p && in_cs % 8 == 0 && out_cs >= nout_p && out_cs % 4 == 0 && ((uintptr_t)in & 15) == 0 &&
                ((uintptr_t)out & 7) == 0 && ((uintptr_t)wp & 15) == 0, "lp_conv3d_fwd: strides / alignment (in_cs=%d out_cs=%d)", in_cs, out_cs);
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lp_conv3d_fwd: scale/shift come in pairs");
    CTU_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "lp_conv3d_fwd: empty volume");
    LpConvP p{};
    p.in = in; p.wp = wp; p.out = out; p.scale = in_scale; p.shift = in_shift; p.bias = bias; p.stats = stats;
    CTU_REQUIRE(!tail || (stats && tail->counter && tail->gamma && tail->beta && tail->scale && tail->shift && tail->mean &&
                          tail->invstd && tail->C > 0 && tail->C <= nout_p && tail->count > 0),
                "lp_conv3d_fwd: incomplete BatchNorm tail");
    p.tail = tail_or_off(tail);
    p.in_cs = in_cs; p.rin_p = rin_p; p.relu = in_relu; p.out_cs = out_cs; p.nout_p = nout_p; p.nbias = bias ? nbias : 0;
    p.S = lp_voxel_stride(rin_p);
    hipStream_t st = (hipStream_t)stream;
    if (layout == 1) {
        p.N = N; p.D = D; p.H = H; p.W = W;
        p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, LPP_TH); p.tiles_w = ceil_div(W, LPP_BW);
        const int nt_ = lp_pair_ntiles(N, D, H, W);
        int gx, tpb;
        lp_pair_grid(nt_, &gx, &tpb);
        CTU_REQUIRE((int64_t)(6 * H + 6) * W * in_cs * 2 < (int64_t)1 << 31, "lp_conv3d_fwd: volume too large for 32-bit offsets");
        CTU_DISPATCH_LP(dtype, lp_conv_fwd_pair_kernel<T><<<gx, 256, LPP_LDS, st>>>(p, nt_, tpb));
        CTU_CHECK_LAUNCH("lp_conv3d_fwd (pair)");
        return CTU_OK;
    }
    const int ntiles = lp_fill(p, N, D, H, W, rin_p);
    // two out tiles per block share one staged box -- unless that leaves CUs idle (small volumes: one tile per block)
    const bool two = nout_p > 16 && (int64_t)ntiles * ceil_div((nout_p + 15) >> 4, 2) >= 256;
    int rc = CTU_OK;
    CTU_DISPATCH_LP(dtype, {
        if (k == 3) rc = two ? lp_conv_launch<T, 3, 2>(p, ntiles, st) : lp_conv_launch<T, 3, 1>(p, ntiles, st);
        else rc = two ? lp_conv_launch<T, 5, 2>(p, ntiles, st) : lp_conv_launch<T, 5, 1>(p, ntiles, st);
    });
    return rc;
}

// The first layer's data gradient on the matrix pipe: g (8 padded channels, 16-bit, raw-output gradient) -> dx float32
// [N][cin][D][H][W], cin <= 4, through lp_conv_fwd_pair_kernel<T, OUT32> (an 8 -> 8 layer whose outputs beyond cin are
// never stored).  wp = ctu_lp_pack_conv3d_weight(w [Co][cin][27], mode 1, layout 1) with rin_p = nout_p = 8.  The VALU kernel
// (ctu_lp_conv3d_first_bwd_data) spends 27 x 8 x cin FMAs per voxel on conversions and LDS reads: 78 us at 128^3, 0.70 ms at
// 256^3 with two input channels; this launch costs what an 8 -> 8 forward costs.
extern "C" int ctu_lp_conv3d_first_bwd_data_pair_supported(int cin, int W) { return cin >= 1 && cin <= 4 && W >= 32; }

extern "C" int ctu_lp_conv3d_first_bwd_data_pair(int dtype, const void* g, int g_cs, const void* wp, int cin, float* dx,
                                                 int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(g && wp && dx, "lp_conv3d_first_bwd_data_pair: null pointer");
    CTU_REQUIRE(ctu_lp_conv3d_first_bwd_data_pair_supported(cin, W), "lp_conv3d_first_bwd_data_pair: cin=%d W=%d", cin, W);
    CTU_REQUIRE(g_cs >= 8 && g_cs % 8 == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)wp & 15) == 0 && ((uintptr_t)dx & 3) == 0,
                "lp_conv3d_first_bwd_data_pair: strides / alignment (g_cs=%d)", g_cs);
    CTU_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "lp_conv3d_first_bwd_data_pair: empty volume");
    CTU_REQUIRE((int64_t)(6 * H + 6) * W * g_cs * 2 < (int64_t)1 << 31, "lp_conv3d_first_bwd_data_pair: volume too large for 32-bit offsets");
    LpConvP p{};
    p.in = g; p.wp = wp; p.out = dx;
    p.tail = tail_or_off(nullptr);
    p.in_cs = g_cs; p.rin_p = 8; p.relu = 0; p.out_cs = 0; p.nout_p = 8; p.nbias = cin;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, LPP_TH); p.tiles_w = ceil_div(W, LPP_BW);
    const int nt_ = lp_pair_ntiles(N, D, H, W);
    int gx, tpb;
    lp_pair_grid(nt_, &gx, &tpb);
    hipStream_t st = (hipStream_t)stream;
    CTU_DISPATCH_LP(dtype, (lp_conv_fwd_pair_kernel<T, true><<<gx, 256, LPP_LDS, st>>>(p, nt_, tpb)));
    CTU_CHECK_LAUNCH("lp_conv3d_first_bwd_data_pair");
    return CTU_OK;
}

// kernel symbols the launches above pick for a geometry (measurement only: bench.py's per-kernel roofline leg and
// scripts/stage_table.py tag their event pairs with them)
extern "C" const char* ctu_lp_conv3d_fwd_kernel_name(int N, int D, int H, int W, int k, int rin_p, int nout_p, int layout) {
    (void)nout_p;
    if (layout == 1) return "lp_conv_fwd_pair_kernel";
    LpConvP p;
    const int ntiles = lp_fill(p, N, D, H, W, rin_p);
    const LpBox bx = lp_box(W, rin_p, (int64_t)N * D * H * W);
    if (k == 3 && lp_use_persist(3, rin_p, ntiles) && bx.bw >= 16) return "lp_conv_fwd_p1_kernel";
    if (k == 3 && bx.bw == 8 && bx.th == 4) return "lp_conv_fwd_small_kernel";
    return "lp_conv_fwd_kernel";
}
extern "C" const char* ctu_lp_conv3d_wgrad_kernel_name(int D, int H, int W, int k, int cin_p, int cout_p) {
    if (LP_WG8 && lp_wg8_ok(D, H, W, k, cin_p, cout_p)) return "lp_wgrad8_kernel";
    return (LP_WG16 && lp_wg16_ok(D, H, W, k, cin_p, cout_p)) ? "lp_wgrad16_kernel" : "lp_conv_wgrad_kernel";
}

extern "C" size_t ctu_lp_conv3d_wgrad_ws_floats(int N, int D, int H, int W, int k, int cin_p, int cout_p) {
    if ((k != 3 && k != 5) || cin_p <= 0 || cout_p <= 0) return 0;
    LpWgP p;
    const int ntiles = lp_wg_fill(p, N, D, H, W);
    const int pairs = ((cin_p + 15) >> 4) * ((cout_p + 15) >> 4), planes = k == 3 ? 1 : k;
    int gx, tpb;
    lp_wg_grid(ntiles, pairs * planes, &gx, &tpb);
    const int ntap = (k == 3 ? 9 : k) * lp_wg_qn(k, lp_wg_sm(W, cin_p), lp_wg_sn(W, cout_p));
    size_t n = (size_t)gx * pairs * planes * ntap * 256;
    if (lp_wg8_ok(D, H, W, k, cin_p, cout_p) && n < (size_t)512 * 9 * 256) n = (size_t)512 * 9 * 256;     // lp_wgrad8_kernel's slabs
    if (lp_wg16_ok(D, H, W, k, cin_p, cout_p)) {                                                         // lp_wgrad16_kernel's
        int gx16, tpb16;
        lp_wg16_grid(N * (D / 4) * (H / 8) * ceil_div(W, 32), pairs, &gx16, &tpb16);
        const size_t n16 = (size_t)gx16 * pairs * 27 * 256;
        if (n < n16) n = n16;
    }
    return n;
}

struct LpLazy { const void* y; const float* scale; const float* shift; const float* coef; void* out; int cp; };

static int lp_conv3d_wgrad_impl(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                int in_relu, const void* gout, int g_cs, int cout_p, float* dw, int Co, int Ci,
                                const int32_t* cinv, float* ws, int N, int D, int H, int W, int k, const LpLazy* lz, void* stream);

extern "C" int ctu_lp_conv3d_wgrad(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                   int in_relu, const void* gout, int g_cs, int cout_p, float* dw, int Co, int Ci,
                                   const int32_t* cinv, float* ws, int N, int D, int H, int W, int k, void* stream) {
    return lp_conv3d_wgrad_impl(dtype, in, in_cs, cin_p, in_scale, in_shift, in_relu, gout, g_cs, cout_p, dw, Co, Ci, cinv, ws, N, D, H, W,
                                k, nullptr, stream);
}

extern "C" int ctu_lp_conv3d_wgrad_bn_supported(int N, int D, int H, int W, int k, int cin_p, int cout_p) {
    return (N > 0 && cin_p > 0 && cout_p > 0 && lp_wg_lazy_ok(D, H, W, k)) ? 1 : 0;
}

// ctu_conv3d_wgrad_bn for 16-bit tensors: ga = gradient w.r.t. the ACTIVATED output, y = the raw output (geometry of ga), coef =
// ctu_bn_bwd_finalize's [5][cout_p] rows; gy_out (geometry of ga, must not alias it) receives the raw-output gradient.
extern "C" int ctu_lp_conv3d_wgrad_bn(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                      int in_relu, const void* ga, int g_cs, int cout_p, const void* y, const float* bn_scale,
                                      const float* bn_shift, const float* coef, void* gy_out, float* dw, int Co, int Ci,
                                      const int32_t* cinv, float* ws, int N, int D, int H, int W, int k, void* stream) {
    CTU_REQUIRE(y && bn_scale && bn_shift && coef && gy_out && gy_out != ga, "lp_conv3d_wgrad_bn: null pointer / gy_out aliases ga");
    CTU_REQUIRE(lp_wg_lazy_ok(D, H, W, k), "lp_conv3d_wgrad_bn: geometry not supported (ask ctu_lp_conv3d_wgrad_bn_supported)");
    CTU_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)gy_out & 15) == 0, "lp_conv3d_wgrad_bn: 16-byte alignment");
    const LpLazy lz = {y, bn_scale, bn_shift, coef, gy_out, cout_p};
    return lp_conv3d_wgrad_impl(dtype, in, in_cs, cin_p, in_scale, in_shift, in_relu, ga, g_cs, cout_p, dw, Co, Ci, cinv, ws, N, D, H, W,
                                k, &lz, stream);
}

static int lp_conv3d_wgrad_impl(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                int in_relu, const void* gout, int g_cs, int cout_p, float* dw, int Co, int Ci,
                                const int32_t* cinv, float* ws, int N, int D, int H, int W, int k, const LpLazy* lz, void* stream) {
    CTU_REQUIRE(in && gout && dw && ws, "lp_conv3d_wgrad: null pointer");
    CTU_REQUIRE((k == 3 || k == 5) && cin_p > 0 && cin_p % 8 == 0 && cout_p > 0 && cout_p % 8 == 0,
                "lp_conv3d_wgrad: k=%d cin_p=%d cout_p=%d", k, cin_p, cout_p);
    CTU_REQUIRE(in_cs >= cin_p && in_cs % 8 == 0 && g_cs >= cout_p && g_cs % 8 == 0 && ((uintptr_t)in & 15) == 0 &&
                ((uintptr_t)gout & 15) == 0, "lp_conv3d_wgrad: strides / alignment (in_cs=%d g_cs=%d)", in_cs, g_cs);
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lp_conv3d_wgrad: scale/shift come in pairs");
    LpWgP p{};
    p.x = in; p.g = gout; p.scale = in_scale; p.shift = in_shift; p.ws = ws;
    p.x_cs = in_cs; p.cin_p = cin_p; p.relu = in_relu; p.g_cs = g_cs; p.cout_p = cout_p;
    if (lz) { p.lz_y = lz->y; p.lz_out = lz->out; p.lz_scale = lz->scale; p.lz_shift = lz->shift; p.lz_coef = lz->coef; p.lz_cp = lz->cp; }
    hipStream_t st = (hipStream_t)stream;
    if (LP_WG8 && lp_wg8_ok(D, H, W, k, cin_p, cout_p)) {
        p.N = N; p.D = D; p.H = H; p.W = W;
        p.tiles_d = D / 4; p.tiles_h = H / 8; p.tiles_w = W / 32;
        p.ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
        int gx8, tpb8;
        lp_wg8_grid(p.ntiles, &gx8, &tpb8);
        CTU_DISPATCH_LP(dtype, {
            if (lz) lp_wgrad8_kernel<T, true><<<gx8, 256, W8_LDS, st>>>(p, tpb8);
            else lp_wgrad8_kernel<T, false><<<gx8, 256, W8_LDS, st>>>(p, tpb8);
        });
        CTU_CHECK_LAUNCH("lp_conv3d_wgrad (8 -> 8)");
        lp_wgrad_reduce_kernel<3, 2, 2><<<dim3(ceil_div(9 * 256, 64)), 1024, 0, st>>>(ws, dw, Co, Ci, cinv, cin_p, cout_p, gx8);
        CTU_CHECK_LAUNCH("lp_conv3d_wgrad reduce");
        return CTU_OK;
    }
    if (LP_WG16 && lp_wg16_ok(D, H, W, k, cin_p, cout_p)) {
        p.N = N; p.D = D; p.H = H; p.W = W;
        p.tiles_d = D / 4; p.tiles_h = H / 8; p.tiles_w = ceil_div(W, 32);
        p.ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
        const int pairs16 = ((cin_p + 15) >> 4) * ((cout_p + 15) >> 4);
        int gx16, tpb16;
        lp_wg16_grid(p.ntiles, pairs16, &gx16, &tpb16);
        static bool raised = false;
        if (!raised) {
            CTU_REQUIRE(hipFuncSetAttribute((const void*)lp_wgrad16_kernel<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, W16_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)lp_wgrad16_kernel<f16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, W16_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)lp_wgrad16_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, W16_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)lp_wgrad16_kernel<f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, W16_LDS) == hipSuccess,
                        "lp_conv3d_wgrad: cannot raise the dynamic LDS limit");
            raised = true;
        }
        // (lazy BatchNorm backward: only the output-channel tile's FIRST input-channel tile forms and stores the raw-output
        //  gradient -- every (ci tile, co tile) block of a co tile would write the same values)
        CTU_DISPATCH_LP(dtype, {
            if (lz) lp_wgrad16_kernel<T, true><<<dim3(gx16, pairs16), 256, W16_LDS, st>>>(p, tpb16);
            else lp_wgrad16_kernel<T, false><<<dim3(gx16, pairs16), 256, W16_LDS, st>>>(p, tpb16);
        });
        CTU_CHECK_LAUNCH("lp_conv3d_wgrad (16-channel tiles)");
        lp_wgrad_reduce_kernel<3, 1, 1><<<dim3(ceil_div(pairs16 * 27 * 256, 64)), 1024, 0, st>>>(ws, dw, Co, Ci, cinv, cin_p, cout_p, gx16);
        CTU_CHECK_LAUNCH("lp_conv3d_wgrad reduce");
        return CTU_OK;
    }
    const int ntiles = lp_wg_fill(p, N, D, H, W);
    const int nci = (cin_p + 15) >> 4, nco = (cout_p + 15) >> 4, pairs = nci * nco, planes = k == 3 ? 1 : k;
    int gx, tpb;
    lp_wg_grid(ntiles, pairs * planes, &gx, &tpb);
    int rc = CTU_OK;
    CTU_DISPATCH_LP(dtype, {
        if (k == 3) rc = lp_wgrad_dispatch<T, 3>(p, gx, tpb, pairs, st);
        else rc = lp_wgrad_dispatch<T, 5>(p, gx, tpb, pairs, st);
    });
    if (rc != CTU_OK) return rc;
    const int sm = lp_wg_box_w(W) >= 16 ? lp_wg_sm(W, cin_p) : 1, sn = lp_wg_box_w(W) >= 16 ? lp_wg_sn(W, cout_p) : 1;
    const int total = planes * pairs * (k == 3 ? 9 : k) * lp_wg_qn(k, sm, sn) * 256;
    const dim3 rg(ceil_div(total, 64));
#define CTU_LP_WG_REDUCE(K_, SM_, SN_) lp_wgrad_reduce_kernel<K_, SM_, SN_><<<rg, 1024, 0, st>>>(ws, dw, Co, Ci, cinv, cin_p, cout_p, gx)
    if (k == 3) {
        if (sm == 2 && sn == 2) CTU_LP_WG_REDUCE(3, 2, 2); else if (sm == 2) CTU_LP_WG_REDUCE(3, 2, 1);
        else if (sn == 2) CTU_LP_WG_REDUCE(3, 1, 2); else CTU_LP_WG_REDUCE(3, 1, 1);
    } else {
        if (sm == 2 && sn == 2) CTU_LP_WG_REDUCE(5, 2, 2); else if (sm == 2) CTU_LP_WG_REDUCE(5, 2, 1);
        else if (sn == 2) CTU_LP_WG_REDUCE(5, 1, 2); else CTU_LP_WG_REDUCE(5, 1, 1);
    }
#undef CTU_LP_WG_REDUCE
    CTU_CHECK_LAUNCH("lp_conv3d_wgrad reduce");
    return CTU_OK;
}

// ---- 16-bit fused decoder up-convolution (upconv_lp.hip): weight gradient w.r.t. the composite weights.  in = COARSE
// activations (lazy transform), gout = fine-grid gradient of the fused op's raw output (8 padded channels, channel stride 8),
// dweff fp32 [8][8][cin_p][8] for ctu_lp_upconv_fused_project.  N, D, H, W: COARSE dims; cin_p a multiple of 16.
static void lp_upwg_geom(int N, int D, int H, int W, int cin_p, int* gx, int* tpb) {
    LpWgP p;
    const int ntiles = lp_wg_fill(p, N, D, H, W);
    lp_wg_grid(ntiles, 4 * (cin_p >> 4), gx, tpb);
}
extern "C" size_t ctu_lp_upconv_fused_wgrad_ws_floats(int N, int D, int H, int W, int cin_p) {
    int gx, tpb;
    lp_upwg_geom(N, D, H, W, cin_p, &gx, &tpb);
    if (lp_upwg4_ok(D, H, W, cin_p) && gx < 256) gx = 256;        // lp_upwg4_kernel's slabs
    return (size_t)gx * 4 * (cin_p >> 4) * 12 * 256;
}
static int lp_upconv_fused_wgrad_impl(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                      int in_relu, const void* gout, int g_cs, float* dweff, float* ws, int N, int D, int H,
                                      int W, const LpLazy* lz, void* stream);
extern "C" int ctu_lp_upconv_fused_wgrad(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                         int in_relu, const void* gout, int g_cs, float* dweff, float* ws, int N, int D, int H,
                                         int W, void* stream) {
    return lp_upconv_fused_wgrad_impl(dtype, in, in_cs, cin_p, in_scale, in_shift, in_relu, gout, g_cs, dweff, ws, N, D, H, W, nullptr, stream);
}
extern "C" int ctu_lp_upconv_fused_wgrad_bn_supported(int N, int D, int H, int W, int cin_p) {
    return (N > 0 && cin_p % 16 == 0 && lp_wg_lazy_ok(D, H, W, 3)) ? 1 : 0;
}
// ... with the BatchNorm + ReLU backward of the fused op's output folded in (ctu_upconv_fused_wgrad_bn for 16-bit tensors)
extern "C" int ctu_lp_upconv_fused_wgrad_bn(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                            int in_relu, const void* ga, int g_cs, const void* y, const float* bn_scale,
                                            const float* bn_shift, const float* coef, void* gy_out, float* dweff, float* ws,
                                            int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(y && bn_scale && bn_shift && coef && gy_out && gy_out != ga, "lp_upconv_fused_wgrad_bn: null pointer / gy_out aliases ga");
    CTU_REQUIRE(ctu_lp_upconv_fused_wgrad_bn_supported(N, D, H, W, cin_p), "lp_upconv_fused_wgrad_bn: geometry not supported");
    const LpLazy lz = {y, bn_scale, bn_shift, coef, gy_out, 8};
    return lp_upconv_fused_wgrad_impl(dtype, in, in_cs, cin_p, in_scale, in_shift, in_relu, ga, g_cs, dweff, ws, N, D, H, W, &lz, stream);
}
static int lp_upconv_fused_wgrad_impl(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                      int in_relu, const void* gout, int g_cs, float* dweff, float* ws, int N, int D, int H,
                                      int W, const LpLazy* lz, void* stream) {
    CTU_REQUIRE(in && gout && dweff && ws, "lp_upconv_fused_wgrad: null pointer");
    CTU_REQUIRE(cin_p >= 16 && cin_p % 16 == 0 && in_cs >= cin_p && in_cs % 8 == 0 && g_cs == 8 && ((uintptr_t)in & 15) == 0 &&
                ((uintptr_t)gout & 15) == 0, "lp_upconv_fused_wgrad: cin_p=%d in_cs=%d g_cs=%d (gradient stride must be 8)", cin_p, in_cs, g_cs);
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lp_upconv_fused_wgrad: scale/shift come in pairs");
    LpWgP p{};
    p.x = in; p.g = gout; p.scale = in_scale; p.shift = in_shift; p.ws = ws;
    p.x_cs = in_cs; p.cin_p = cin_p; p.relu = in_relu; p.g_cs = g_cs; p.cout_p = 16;
    if (lz) { p.lz_y = lz->y; p.lz_out = lz->out; p.lz_scale = lz->scale; p.lz_shift = lz->shift; p.lz_coef = lz->coef; p.lz_cp = lz->cp; }
    hipStream_t st = (hipStream_t)stream;
    if (LP_UPWG4 && !lz && lp_upwg4_ok(D, H, W, cin_p)) {
        p.N = N; p.D = D; p.H = H; p.W = W;
        p.tiles_d = D / 4; p.tiles_h = H / 4; p.tiles_w = W / 32;
        p.ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
        int gx4, tpb4;
        lp_upwg4_grid(p.ntiles, &gx4, &tpb4);
        static bool raised = false;
        if (!raised) {
            CTU_REQUIRE(hipFuncSetAttribute((const void*)lp_upwg4_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, U4_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)lp_upwg4_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, U4_LDS) == hipSuccess,
                        "lp_upconv_fused_wgrad: cannot raise the dynamic LDS limit");
            raised = true;
        }
        CTU_DISPATCH_LP(dtype, (lp_upwg4_kernel<T><<<dim3(gx4, cin_p >> 5), 256, U4_LDS, st>>>(p, tpb4)));
        CTU_CHECK_LAUNCH("lp_upconv_fused_wgrad (4 parities)");
        lp_upwg_reduce_kernel<<<dim3(12 * 256 / 64, 4 * (cin_p >> 4)), 1024, 0, st>>>(ws, dweff, cin_p, cin_p >> 4, gx4);
        CTU_CHECK_LAUNCH("lp_upconv_fused_wgrad reduce");
        return CTU_OK;
    }
    lp_wg_fill(p, N, D, H, W);
    int gx, tpb;
    lp_upwg_geom(N, D, H, W, cin_p, &gx, &tpb);
    const int bw = lp_wg_box_w(W);
    int rc = CTU_OK;
    CTU_DISPATCH_LP(dtype, {
        if (lz) rc = bw == 32 ? lp_upwg_launch<T, 32, true>(p, gx, tpb, st) : lp_upwg_launch<T, 16, true>(p, gx, tpb, st);
        else if (bw == 32) rc = lp_upwg_launch<T, 32>(p, gx, tpb, st);
        else if (bw == 16) rc = lp_upwg_launch<T, 16>(p, gx, tpb, st);
        else rc = lp_upwg_launch<T, 8>(p, gx, tpb, st);
    });
    if (rc != CTU_OK) return rc;
    lp_upwg_reduce_kernel<<<dim3(12 * 256 / 64, 4 * (cin_p >> 4)), 1024, 0, st>>>(ws, dweff, cin_p, cin_p >> 4, gx);
    CTU_CHECK_LAUNCH("lp_upconv_fused_wgrad reduce");
    return CTU_OK;
}

// jobs: HOST array of ctu_pack_job (kind 0 = Conv3d [Co,Ci,k,k,k], kind 1 = ConvTranspose3d [Ci,Co,2,2,2]; `layout` unused);
// wp of each job: 16-bit buffer of ctu_lp_conv3d_packed_elems / ctu_lp_convt_packed_elems elements.
extern "C" int ctu_lp_pack_batch(int dtype, const ctu_pack_job* jobs, int n, void* stream) {
    CTU_REQUIRE(jobs && n > 0, "lp_pack_batch: no jobs");
    hipStream_t st = (hipStream_t)stream;
    for (int j0 = 0; j0 < n; j0 += LP_PACK_MAXJ) {
        const int nj = (n - j0) < LP_PACK_MAXJ ? (n - j0) : LP_PACK_MAXJ;
        LpPackTable tb;
        size_t mx = 1;
        for (int j = 0; j < nj; ++j) {
            const ctu_pack_job& jb = jobs[j0 + j];
            CTU_REQUIRE(jb.w && jb.wp && (jb.kind == 0 || jb.kind == 1) && (jb.mode == 0 || jb.mode == 1) && jb.rin_p > 0 &&
                        jb.rin_p % 8 == 0 && jb.nout_p > 0 && jb.nout_p % 8 == 0 && (jb.kind == 1 || jb.k == 3 || jb.k == 5),
                        "lp_pack_batch: bad job %d", j0 + j);
            tb.j[j] = jb;
            CTU_REQUIRE(jb.kind == 1 || jb.layout == 0 || (jb.layout == 1 && jb.k == 3 && jb.rin_p == 8 && jb.nout_p == 8), "lp_pack_batch: bad layout in job %d", j0 + j);
            const size_t tot = jb.kind == 0 ? ctu_lp_conv3d_packed_elems(jb.k, jb.rin_p, jb.nout_p)
                                            : (size_t)(jb.mode == 0 ? 8 * ((jb.rin_p + 31) >> 5) : 2 * (jb.rin_p >> 3)) * ((jb.nout_p + 15) >> 4) * 512;
            if (tot > mx) mx = tot;
        }
        int gx = (int)ceil_div64((int64_t)mx, 256 * 4);
        if (gx > 256) gx = 256;
        CTU_DISPATCH_LP(dtype, lp_pack_batch_kernel<T><<<dim3(gx, nj), 256, 0, st>>>(tb));
        CTU_CHECK_LAUNCH("lp_pack_batch");
    }
    return CTU_OK;
}
