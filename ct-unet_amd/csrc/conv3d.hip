// 3D convolution (k=3 / k=5, stride 1, "same" zero padding) for gfx950 as implicit GEMM on
// v_mfma_f32_16x16x4_f32: forward / data-gradient kernel, weight-gradient kernel, packing.
//
// Replaces nn.Conv3d at ctunet/pytorch/models.py:26,29,38,41,71,76 (k3, no bias) and
// :403,407,430,434,482-488 (k5, bias), plus their autograd backward.
//
// Data layout (DESIGN.md "HBM layout"): activations are channels-last fp32 with a channel
// stride, channel counts padded to 8.  GEMM view: M = output voxels, N = output channels,
// K = taps x input channels.  A block owns a TDxTHxTW box of output voxels (16 M-tiles of 16
// voxels, 4 waves x MT tiles), stages the haloed input box for ONE 8-channel chunk into LDS
// (12 floats per voxel: 8 + 4 pad -> conflict-free ds_read_b64 of 16 consecutive voxels) with
// the lazy-BN transform applied on the way in, stages that chunk's packed weights, and walks
// the taps with compile-time LDS offsets.  One ds_read_b64 per lane feeds two MFMAs: lane
// (m, kq) holds channels {2kq, 2kq+1}; MFMA #0 contracts channels {0,2,4,6}, #1 {1,3,5,7}.
#include "common.h"
#include "pack.h"
#include "bn_tail.h"

#ifndef CTU_K3S_OCC
#define CTU_K3S_OCC 2
#endif
#ifndef CTU_K3S_ALL
#define CTU_K3S_ALL 1
#endif
#ifndef CTU_WG_OCC
#define CTU_WG_OCC 2
#endif
#ifndef CTU_WG_BLOCKS
#define CTU_WG_BLOCKS 512
#endif

namespace {

typedef float gv2f __attribute__((ext_vector_type(2)));

struct ConvP {
    const float* in;
    const float* in_scale;
    const float* in_shift;
    const float* wp;
    const float* bias;
    float* out;
    float* stats;
    int in_cs, rin_p, in_relu, out_cs, nout_p, nbias;
    int N, D, H, W;
    int tiles_d, tiles_h, tiles_w;
    int n16;                 // number of 16-wide output-channel tiles in the packed weights
    ctu_bn_tail tail;        // counter != NULL: the last block of the launch finalizes the BatchNorm (bn_tail.h)
};


__host__ __device__ inline void pick_tile(int W, int* td, int* th, int* tw) {
    if (W >= 16) { *td = 4; *th = 4; *tw = 16; }
    else if (W >= 8) { *td = 4; *th = 8; *tw = 8; }
    else { *td = 4; *th = 4; *tw = 4; }
}



// CCH = 8-channel chunks staged per barrier pair (4 on the small deep-level volumes, where the per-stage latency,
// not the matrix pipe, sets the time).
template <int KS, int NT, int TD, int TH, int TW, int CCH = 1>
__global__ __launch_bounds__(256) void conv3d_fwd_kernel(ConvP p) {
    constexpr int PAD = (KS - 1) / 2;
    constexpr int HD = TD + KS - 1, HH = TH + KS - 1, HW = TW + KS - 1;
    constexpr int HV = HD * HH * HW;
    constexpr int NVOX = TD * TH * TW;
    constexpr int NMT = NVOX / 16;
    static_assert(NMT % 4 == 0, "block tile must hold a multiple of 4 M-tiles");
    constexpr int MT = NMT / 4;
    constexpr int STAPS = Taps<KS>::STAPS, NSTAGE = Taps<KS>::NSTAGE;
    constexpr int WFL = STAPS * NT * 128;
    constexpr int VSL = 8 * CCH + 4;                  // LDS floats per halo voxel
    constexpr int AQ = 2 * CCH;                       // float4 per halo voxel
    constexpr int AITEMS = HV * AQ;
    constexpr int AITER = (AITEMS + 255) / 256;
    static_assert(CCH == 1 || NSTAGE == 1, "multi-chunk staging needs all taps' weights in LDS at once");

    __shared__ __attribute__((aligned(16))) float sA[HV * VSL];
    __shared__ __attribute__((aligned(16))) float sW[CCH * WFL];
    __shared__ float sRed[4 * NT * 16 * 2];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;

    int bx = blockIdx.x;
    const int tx = bx % p.tiles_w; bx /= p.tiles_w;
    const int ty = bx % p.tiles_h; bx /= p.tiles_h;
    const int tz = bx % p.tiles_d; bx /= p.tiles_d;
    const int n_img = bx;
    const int d0 = tz * TD, h0 = ty * TH, w0 = tx * TW;
    const int by = blockIdx.y;

    // per-lane LDS base (floats) of each of this wave's M-tiles
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int vt = (wave * MT + mt) * 16 + m;
        const int tw = vt % TW, th = (vt / TW) % TH, td = vt / (TW * TH);
        abase[mt] = ((td * HH + th) * HW + tw) * VSL + kq * 2;
    }
    const int bbase = kq * 32 + m * 2;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunk = p.rin_p >> 3;
    const int half = tid % AQ;                        // this thread's channel quad inside a stage (256 % AQ == 0)
    const bool has_xf = p.in_scale != nullptr;
    const int n16 = p.n16;

    if constexpr (CCH > 1) {
        // Small deep-level volumes: one block per CU and one wave per SIMD, so nothing hides a latency unless the kernel
        // does it itself.  (a) the NEXT stage's halo and weights are fetched into registers (all loads issued together,
        // branch-free: out-of-volume items read voxel 0 and are zeroed at the LDS write) while this stage's taps run;
        // (b) the (chunk, tap) fragment pairs go through a ring of R registers refilled R steps ahead, the order
        // MFMA / 2 reads / MFMA pinned for the scheduler.
        static_assert(MT == 1 && NT == 1 && NSTAGE == 1, "multi-chunk staging: one accumulator, all taps in one stage");
        constexpr int WITER = (CCH * WFL + 1023) / 1024;
        f32x4 vals[AITER], wv[WITER];                 // (ext vectors: float4 arrays end up as LDS-promoted stack objects)
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned aoff[AITER], amask = 0;              // element offset of each halo item, in-volume bits
#pragma unroll
        for (int it = 0; it < AITER; ++it) {
            const int i = tid + it * 256;
            const int v = i / AQ;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - PAD, gh = h0 + ph - PAD, gw = w0 + pw - PAD;
            const bool ok = i < AITEMS && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            aoff[it] = ok ? (unsigned)(((gd * p.H + gh) * p.W + gw) * p.in_cs + half * 4) : 0u;
            amask |= ok ? (1u << it) : 0u;
        }
        const float* in_img = p.in + (size_t)n_img * p.D * p.H * p.W * p.in_cs;
        const float* w_blk = p.wp + (size_t)by * 128;
        auto fetch = [&](int c) {
#pragma unroll
            for (int it = 0; it < AITER; ++it) vals[it] = *reinterpret_cast<const f32x4*>(in_img + aoff[it] + c * 8);
#pragma unroll
            for (int wi = 0; wi < WITER; ++wi) {
                int i = tid * 4 + wi * 1024;
                if (i >= CCH * WFL) i = 0;             // (the last pass is ragged: re-read item 0, not stored)
                const int g = i / WFL, j = i % WFL;
                wv[wi] = *reinterpret_cast<const f32x4*>(w_blk + ((size_t)(c + g) * STAPS + j / 128) * n16 * 128 + j % 128);
            }
            if (has_xf) {
                sc = *reinterpret_cast<const float4*>(p.in_scale + c * 8 + half * 4);
                sh = *reinterpret_cast<const float4*>(p.in_shift + c * 8 + half * 4);
            }
        };
        fetch(0);
        for (int c = 0; c < nchunk; c += CCH) {
            __syncthreads();   // previous stage's readers are done with sA / sW
#pragma unroll
            for (int it = 0; it < AITER; ++it) {
                const int i = tid + it * 256;
                float4 v = make_float4(vals[it][0], vals[it][1], vals[it][2], vals[it][3]);
                if (has_xf) v = xform4(v, sc, sh, p.in_relu);
                const bool ok = (amask >> it) & 1u;
                if (i < AITEMS)
                    *reinterpret_cast<float4*>(&sA[(i / AQ) * VSL + half * 4]) =
                        make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
#pragma unroll
            for (int wi = 0; wi < WITER; ++wi) {
                const int i = tid * 4 + wi * 1024;
                if (i < CCH * WFL) *reinterpret_cast<f32x4*>(&sW[i]) = wv[wi];
            }
            __syncthreads();
            if (c + CCH < nchunk) fetch(c + CCH);
            __builtin_amdgcn_sched_barrier(0);         // (the loads are issued HERE, in front of the taps)
            constexpr int R = 8, NJ = CCH * STAPS;
            gv2f ar[R], br[R];
            auto rd = [&](int j, gv2f& a, gv2f& b) {
                const int g = j / STAPS, ts = j % STAPS;
                const int kd = ts / (KS * KS), kh = (ts / KS) % KS, kw = ts % KS;
                a = *reinterpret_cast<const gv2f*>(&sA[abase[0] + ((kd * HH + kh) * HW + kw) * VSL + g * 8]);
                b = *reinterpret_cast<const gv2f*>(&sW[g * WFL + ts * 128 + bbase]);
            };
#pragma unroll
            for (int j = 0; j < R; ++j) rd(j, ar[j], br[j]);
            // the whole ring is in flight before the first MFMA (fake use: keeps the fill reads from trickling in)
            static_assert(R == 8, "fake-use list");
            asm volatile("" : "+v"(ar[0]), "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3]), "+v"(ar[4]), "+v"(ar[5]), "+v"(ar[6]),
                         "+v"(ar[7]), "+v"(br[0]), "+v"(br[1]), "+v"(br[2]), "+v"(br[3]), "+v"(br[4]), "+v"(br[5]),
                         "+v"(br[6]), "+v"(br[7]));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const gv2f a = ar[j % R], b = br[j % R];
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[0][0], 0, 0, 0);
                if (j + R < NJ) rd(j + R, ar[j % R], br[j % R]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (j + R < NJ) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
    } else
    for (int c = 0; c < nchunk; c += CCH) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_xf) {
            sc = *reinterpret_cast<const float4*>(p.in_scale + c * 8 + half * 4);
            sh = *reinterpret_cast<const float4*>(p.in_shift + c * 8 + half * 4);
        }
        __syncthreads();   // previous chunk's readers are done with sA / sW
        // ---- stage the haloed input box, 8 channels of chunk c
        float4 vals[AITER];
#pragma unroll
        for (int it = 0; it < AITER; ++it) {
            const int i = tid + it * 256;
            const int v = i / AQ;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - PAD, gh = h0 + ph - PAD, gw = w0 + pw - PAD;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < AITEMS && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H &&
                (unsigned)gw < (unsigned)p.W) {
                const size_t vox = (((size_t)n_img * p.D + gd) * p.H + gh) * p.W + gw;
                val = *reinterpret_cast<const float4*>(p.in + vox * p.in_cs + c * 8 + half * 4);
                if (has_xf) val = xform4(val, sc, sh, p.in_relu);
            }
            vals[it] = val;
        }
#pragma unroll
        for (int it = 0; it < AITER; ++it) {
            const int i = tid + it * 256;
            if (i < AITEMS) *reinterpret_cast<float4*>(&sA[(i / AQ) * VSL + half * 4]) = vals[it];
        }
        for (int s = 0; s < NSTAGE; ++s) {
            if (s > 0) __syncthreads();
            // packed weights: [chunk][stage][tap][n16 tile][128]; this block takes tiles by*NT .. by*NT+NT-1
            // (loads in batches of 8, branch-free: a load/store pair per iteration exposes one L2 round trip each)
            constexpr int WB = 8;
            for (int i0 = tid * 4; i0 < CCH * WFL; i0 += WB * 1024) {
                f32x4 wv[WB];
#pragma unroll
                for (int q = 0; q < WB; ++q) {
                    const int i = (i0 + q * 1024 < CCH * WFL) ? i0 + q * 1024 : i0;
                    const int g = i / WFL, j = i % WFL;
                    const int ts = j / (NT * 128), r = j % (NT * 128);
                    const bool ok = by * NT * 128 + r < n16 * 128;
                    wv[q] = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)(c + g) * NSTAGE + s) * STAPS * n16 * 128 +
                                                            (size_t)ts * n16 * 128 + (ok ? (size_t)by * NT * 128 + r : (size_t)0));
                }
#pragma unroll
                for (int q = 0; q < WB; ++q) {
                    const int i = i0 + q * 1024;
                    if (i < CCH * WFL) {
                        const bool ok = by * NT * 128 + (i % WFL) % (NT * 128) < n16 * 128;
                        *reinterpret_cast<f32x4*>(&sW[i]) = ok ? wv[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int g = 0; g < CCH; ++g)
#pragma unroll
            for (int ts = 0; ts < STAPS; ++ts) {
                // tap (kd,kh,kw): with NSTAGE > 1 the stage index is kd
                const int kd = (NSTAGE == 1) ? ts / (KS * KS) : s;
                const int kh = (NSTAGE == 1) ? (ts / KS) % KS : ts / KS;
                const int kw = ts % KS;
                const int toff = ((kd * HH + kh) * HW + kw) * VSL + g * 8;
                float2 b[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[nt] = *reinterpret_cast<const float2*>(&sW[g * WFL + (ts * NT + nt) * 128 + bbase]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float2 a = *reinterpret_cast<const float2*>(&sA[abase[mt] + toff]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[nt].x, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[nt].y, acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: bias, store, BatchNorm partial statistics
    const int q = kq;
    float s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (by * NT + nt) * 16 + m;
        const bool cok = co < p.nout_p;
        const float bv = (p.bias && co < p.nbias) ? p.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int vt = (wave * MT + mt) * 16 + q * 4 + r;
                const int tw = vt % TW, th = (vt / TW) % TH, td = vt / (TW * TH);
                const int gd = d0 + td, gh = h0 + th, gw = w0 + tw;
                if (cok && gd < p.D && gh < p.H && gw < p.W) {
                    const float v = acc[mt][nt][r] + bv;
                    const size_t vox = (((size_t)n_img * p.D + gd) * p.H + gh) * p.W + gw;
                    p.out[vox * p.out_cs + co] = v;
                    s1[nt] += v;
                    s2[nt] += v * v;
                }
            }
        }
    }
    if (p.stats) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float a = s1[nt], b = s2[nt];
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
            if (q == 0) {
                sRed[((wave * NT + nt) * 16 + m) * 2 + 0] = a;
                sRed[((wave * NT + nt) * 16 + m) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < NT * 16) {
            const int co = by * NT * 16 + tid;
            if (co < p.nout_p) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    a += sRed[((w * NT) * 16 + tid) * 2 + 0];
                    b += sRed[((w * NT) * 16 + tid) * 2 + 1];
                }
                float* row = p.stats + (size_t)blockIdx.x * 2 * p.nout_p;
                st_row(p.tail.counter != nullptr, row + co, a);
                st_row(p.tail.counter != nullptr, row + p.nout_p + co, b);
            }
        }
        if (p.tail.counter) bn_fwd_tail(p.tail, p.stats, gridDim.x, p.nout_p, gridDim.x * gridDim.y);
    }
}


// ------------------------------------------------------------------ persistent k=3 kernels (large layers)
// Same GEMM view and LDS images as conv3d_fwd_kernel, restructured for the large layers:
//  * a block walks a CONTIGUOUS range of voxel boxes; per-thread halo offsets are computed once per block,
//  * the next (box, chunk) stage's input and weights are prefetched into registers while the current
//    stage's taps run on the matrix cores; one LDS image, two barriers per stage,
//  * layers with a single 8-channel chunk (C_in <= 8) stage their weights once per block,
//  * MFMA operands are swapped (weights = row operand): a lane ends with 4 consecutive output channels of
//    one voxel -> one 16-byte store per lane per tile,
//  * BatchNorm partial sums live in registers across all boxes of the block: ONE stats row per block.
// PAIR = false: N = 16*NT output channels, box 4x4x16.
// PAIR = true : C_out = 8, N = (w-shift s, co), M = voxel pairs, 36 taps, box 4x4x32 (NT must be 1):
//   out[(d,h,2m+s), co] = sum in[(d+kd, h+kh, 2m+kw') - pad][ci] * Wp[kd,kh,kw'][ci][(s,co)],
//   Wp[..kw'][ci][(s,co)] = W[..kw'-s][ci][co] for 0 <= kw'-s <= 2, else 0  -> 1.5x fewer MFMAs than padding N.
typedef float v2f __attribute__((ext_vector_type(2)));

#ifdef CTU_STAMP
// diagnostic build only (scripts/diag_stamp.hip): per-phase cycle sums of wave 0, written to a buffer of their own
__device__ unsigned long long* g_stamp_out = nullptr;
#define STAMP(var)                                                                   \
    do {                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");  \
        __builtin_amdgcn_sched_barrier(0);                                           \
    } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

// second launch bound = waves per SIMD the register allocation must allow: 3 resident blocks for the NT = 1 box
// (45 KB of LDS each), 2 for the pair layout and NT = 2 (LDS- / accumulator-bound)
#ifndef CTU_FWD_OCC1
#define CTU_FWD_OCC1 3
#endif
template <int NT, bool PAIR>
__global__ __launch_bounds__(256, (NT == 1 && !PAIR) ? CTU_FWD_OCC1 : 2) void conv3d_fwd_k3_persist(ConvP p, int ntiles, int tiles_per_block) {
    constexpr int MT = 4;
    constexpr int TD = 4, TH = 4, TW = PAIR ? 32 : 16;
    constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW;
    constexpr int NWAVE = 16 / MT, NTHR = 64 * NWAVE;
    constexpr int NTAP = PAIR ? 36 : 27, KWN = PAIR ? 4 : 3;
    constexpr int WFL = NTAP * NT * 128;
    constexpr int AITEMS = HV * 2, AITER = (AITEMS + NTHR - 1) / NTHR;
    constexpr int WITER = (WFL / 4 + NTHR - 1) / NTHR;
    static_assert(!PAIR || NT == 1, "pair layout has a single N tile");

    // LDS tiles are indexed in float2 units so that every fragment read is a provably 8-byte-aligned ds_read_b64
    // (the layouts are conflict-free for b64; the b32/read2_b32 forms the compiler falls back to are not).
    // Voxel stride: 12 floats (6 float2) for 1-voxel lane stride, 10 floats (5 float2) for the pair layout's 2-voxel stride.
    constexpr int VSK = PAIR ? 10 : 12, VS2 = VSK / 2;
    __shared__ __attribute__((aligned(16))) v2f sA2[HV * VS2];
    __shared__ __attribute__((aligned(16))) v2f sW2[WFL / 2];
    // fragment reads go through volatile pointers: each is ONE ds_read_b64 that the compiler can neither split into the
    // b32 / read2_b32 forms (half the bandwidth, bank-conflicting on these layouts) nor fuse into read2_b64
    typedef const volatile __attribute__((address_space(3))) v2f* lds_v2f_ptr;
    lds_v2f_ptr vA = (lds_v2f_ptr)sA2;
    lds_v2f_ptr vW = (lds_v2f_ptr)sW2;
    __shared__ float sRed[NWAVE * NT * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int by = blockIdx.y;
    const int n16 = p.n16;
    const int nchunk = p.rin_p >> 3;
    const bool hoist_w = nchunk == 1;
    const int half = tid & 1;
    const bool has_xf = p.in_scale != nullptr;

    // this wave owns box plane td = wave; its 4 M-tiles are the rows th = 0..3.  Per-lane base of row 0, tap (0,0,0):
    const int rowbase = ((wave * HH) * HW + (PAIR ? 2 * m : m)) * VS2 + kq;                 // float2 units
    const int bbase = kq * 16 + m;                                                          // float2 units

    // per-thread staging items: BYTE offsets relative to the box's halo origin (non-negative: the loads take the
    // scalar-base + 32-bit-offset form, no VALU address arithmetic) and 6-bit halo-face codes, 5 per word
    unsigned hoff[AITER];
    constexpr int FPW = 5, NFW = (AITER + FPW - 1) / FPW;
    unsigned fw[NFW];
#pragma unroll
    for (int q = 0; q < NFW; ++q) fw[q] = 0;
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
        const int i = tid + it * NTHR;
        const int v = (i < AITEMS) ? (i >> 1) : 0;
        const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
        hoff[it] = (unsigned)(((pd * p.H + ph) * p.W + pw) * p.in_cs + half * 4) * 4u;
        const unsigned face = (pd == 0 ? 1u : 0u) | (pd == HD - 1 ? 2u : 0u) | (ph == 0 ? 4u : 0u) | (ph == HH - 1 ? 8u : 0u) |
                              (pw == 0 ? 16u : 0u) | (pw == HW - 1 ? 32u : 0u);
        fw[it / FPW] |= face << (6 * (it % FPW));
    }
    // stands in for the out-of-volume items of a full border box: the box origin, always inside the volume
    const unsigned safe_off = (unsigned)(((p.H + 1) * p.W + 1) * p.in_cs + half * 4) * 4u;

    f32x4 acc[MT][NT];
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    }

    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(ntiles, tile + tiles_per_block);
    if (tile >= tile_end) return;
    int c = 0;

    float4 va[AITER], vw[WITER];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned vmask = 0;

    // box coordinates advance incrementally (no div/mod per stage)
    struct Box { int tx, ty, tz, n; };
    auto box_of = [&](int t) {
        Box b;
        b.tx = t % p.tiles_w; t /= p.tiles_w;
        b.ty = t % p.tiles_h; t /= p.tiles_h;
        b.tz = t % p.tiles_d; b.n = t / p.tiles_d;
        return b;
    };
    auto box_next = [&](Box b) {
        if (++b.tx == p.tiles_w) { b.tx = 0; if (++b.ty == p.tiles_h) { b.ty = 0; if (++b.tz == p.tiles_d) { b.tz = 0; ++b.n; } } }
        return b;
    };
    bool a_interior = false;                       // uniform: the staged box's halo lies inside the volume
    auto load_a = [&](Box b, int cc) {
        const int d0 = b.tz * TD, h0 = b.ty * TH, w0 = b.tx * TW;
        if (has_xf) {
            sc = *reinterpret_cast<const float4*>(p.in_scale + cc * 8 + half * 4);
            sh = *reinterpret_cast<const float4*>(p.in_shift + cc * 8 + half * 4);
        }
        const long long org = (((long long)b.n * p.D + d0) * p.H + h0) * p.W + w0 - ((long long)p.H * p.W + p.W + 1);
        const char* base = reinterpret_cast<const char*>(p.in + org * p.in_cs + cc * 8);      // halo origin of the box
        a_interior = d0 >= 1 && h0 >= 1 && w0 >= 1 && d0 + TD < p.D && h0 + TH < p.H && w0 + TW < p.W;
        if (a_interior) {                          // the common case: no per-item predicates
#pragma unroll
            for (int it = 0; it < AITER; ++it) {
                if ((it + 1) * NTHR <= AITEMS || tid + it * NTHR < AITEMS)
                    va[it] = *reinterpret_cast<const float4*>(base + hoff[it]);
                else
                    va[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
        vmask = 0;
        if (d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W) {
            // full border box: the only out-of-volume items sit on halo faces that coincide with volume faces; they
            // load the box origin instead (branch-free) and are zeroed when written to LDS
            const unsigned bface = (d0 == 0 ? 1u : 0u) | (d0 + TD == p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) |
                                   (h0 + TH == p.H ? 8u : 0u) | (w0 == 0 ? 16u : 0u) | (w0 + TW == p.W ? 32u : 0u);
#pragma unroll
            for (int it = 0; it < AITER; ++it) {
                const bool ok = (fw[it / FPW] & (bface << (6 * (it % FPW)))) == 0u;
                va[it] = *reinterpret_cast<const float4*>(base + (ok ? hoff[it] : safe_off));
                vmask |= ok ? (1u << it) : 0u;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < AITER; ++it) {       // ragged box (volume not a multiple of the box): per-item bounds
            const int i = tid + it * NTHR, v = i >> 1;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - 1, gh = h0 + ph - 1, gw = w0 + pw - 1;
            const bool ok = i < AITEMS && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                val = *reinterpret_cast<const float4*>(base + hoff[it]);
                vmask |= 1u << it;
            }
            va[it] = val;
        }
    };
    const bool w_full = (by + 1) * NT <= n16;      // uniform: every N tile of this block exists
    auto load_w = [&](int cc) {
        const float* wsrc = p.wp + (size_t)cc * NTAP * n16 * 128 + (size_t)by * NT * 128;
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int i = (tid + it * NTHR) * 4;
            const int ts = i / (NT * 128), r = i % (NT * 128);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (((it + 1) * NTHR * 4 <= WFL || i < WFL) && (w_full || by * NT * 128 + r < n16 * 128))
                v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(wsrc) + (unsigned)(ts * n16 * 128 + r) * 4u);
            vw[it] = v;
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int i = (tid + it * NTHR) * 4;
            if (i < WFL) *reinterpret_cast<float4*>(&sW2[i >> 1]) = vw[it];
        }
    };

    Box box = box_of(tile);
    load_a(box, 0);
    load_w(0);
    if (hoist_w) store_w();
#ifdef CTU_STAMP
    unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0, tq5 = 0, ph[5] = {0, 0, 0, 0, 0}, nst = 0;
    STAMP(tq0);
#endif

    while (true) {
        CTU_SETPRIO(CTU_PRIO_STAGE);           // staging phases outrank the other block's MFMA stream
        __syncthreads();                       // the previous stage's readers are done with sA / sW
        STAMP(tq1);
        {
            auto put = [&](int it, float4 val) {
                const int i = tid + it * NTHR;
#ifdef CTU_ABL_NOLDSW                                 // timing-only build: the transform runs, the LDS stores do not
                asm volatile("" :: "v"(val.x), "v"(val.y), "v"(val.z), "v"(val.w));
                return;
#endif
                if ((it + 1) * NTHR <= AITEMS || i < AITEMS) {
                    if (PAIR) {                       // 40-byte voxel stride: two 8-byte stores
                        sA2[(i >> 1) * VS2 + half * 2] = v2f{val.x, val.y};
                        sA2[(i >> 1) * VS2 + half * 2 + 1] = v2f{val.z, val.w};
                    } else {
                        *reinterpret_cast<float4*>(&sA2[(i >> 1) * VS2 + half * 2]) = val;
                    }
                }
            };
            if (a_interior) {                          // (the flag still describes the box now in va[])
#pragma unroll
                for (int it = 0; it < AITER; ++it) put(it, has_xf ? xform4(va[it], sc, sh, p.in_relu) : va[it]);
            } else if (has_xf) {                       // padding voxels are zeros AFTER the BatchNorm/ReLU transform
#pragma unroll
                for (int it = 0; it < AITER; ++it) {
                    const float4 t = xform4(va[it], sc, sh, p.in_relu);
                    const bool ok = (vmask >> it) & 1u;
                    put(it, make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f));
                }
            } else {
#pragma unroll
                for (int it = 0; it < AITER; ++it) {
                    const bool ok = (vmask >> it) & 1u;
                    put(it, make_float4(ok ? va[it].x : 0.f, ok ? va[it].y : 0.f, ok ? va[it].z : 0.f, ok ? va[it].w : 0.f));
                }
            }
        }
        if (!hoist_w) store_w();
        __syncthreads();
        STAMP(tq2);
        // ---- prefetch the next stage while this one computes
        int ntile = tile, nc = c + 1;
        Box nbox = box;
        if (nc == nchunk) { nc = 0; ntile = tile + 1; nbox = box_next(box); }
        const bool has_next = ntile < tile_end;
        if (has_next) {
#ifndef CTU_ABL_NOLOAD
            load_a(nbox, nc);
#else       // timing-only build: the staged values stay what the first fetch brought (opaque, so nothing is hoisted)
#pragma unroll
            for (int it = 0; it < AITER; ++it) asm volatile("" : "+v"(va[it].x), "+v"(va[it].y), "+v"(va[it].z), "+v"(va[it].w));
#endif
            if (!hoist_w) load_w(nc);
        }
        STAMP(tq3);
        CTU_SETPRIO(0);
        // ---- taps on the matrix cores: rows = output channels, columns = voxels (or voxel pairs).
        // Group (kd, kw): the 6 input rows th' = 0..5 feed the 3 kh taps of all 4 M-tiles (row th + kh), so each group
        // needs 6 + 3*NT fragment reads for 12*NT*2 MFMAs; the next group's fragments are read before this group's MFMAs.
        {
            constexpr int NG = 3 * KWN;
            v2f ar[2][6], br[2][3][NT];
            auto load_group = [&](int gq, v2f (&aa)[6], v2f (&bb)[3][NT]) {
                const int kd = gq / KWN, kw = gq % KWN;
#pragma unroll
                for (int r = 0; r < 6; ++r) aa[r] = vA[rowbase + ((kd * HH + r) * HW + kw) * VS2];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bb[kh][nt] = vW[(((kd * 3 + kh) * KWN + kw) * NT + nt) * 64 + bbase];
            };
#ifdef CTU_ABL_NOMFMA
            constexpr int NGX = 1;                     // timing-only build: one tap group instead of NG
#else
            constexpr int NGX = NG;
#endif
            load_group(0, ar[0], br[0]);
#if CTU_PIN_FWD
            __builtin_amdgcn_sched_barrier(0);         // (the first group's reads stay out of the pinned sequence below)
#endif
#pragma unroll
            for (int gq = 0; gq < NGX; ++gq) {
                if (gq + 1 < NGX) load_group(gq + 1, ar[(gq + 1) & 1], br[(gq + 1) & 1]);
                v2f (&aa)[6] = ar[gq & 1];
                v2f (&bb)[3][NT] = br[gq & 1];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[kh][nt].x, aa[mt + kh].x, acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[kh][nt].y, aa[mt + kh].y, acc[mt][nt], 0, 0, 0);
                }
#if CTU_PIN_FWD
                // pin the order: the next group's fragment reads go out one per MFMA under THIS group's MFMAs (left to
                // itself the scheduler sinks every read to just above its first use and drains lgkmcnt(0) there: one
                // exposed LDS latency per 8 MFMAs, which only a second wave in ITS MFMA phase can cover)
#pragma unroll
                for (int q = 0; q < 24 * NT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (gq + 1 < NGX && q < 6 + 3 * NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#endif
            }
        }
        STAMP(tq4);
        CTU_SETPRIO(CTU_PRIO_STAGE);
        if (c == nchunk - 1) {
            // ---- epilogue of this box: bias, one float4 store per lane and (mt, nt), BN partial sums
            const int n_img = box.n, d0 = box.tz * TD, h0 = box.ty * TH, w0 = box.tx * TW;
            const int gw = PAIR ? (w0 + 2 * m + (kq >> 1)) : (w0 + m);
            const bool box_full = d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W;      // uniform
            float* obase = p.out + ((((size_t)n_img * p.D + d0 + wave) * p.H + h0) * p.W + gw) * p.out_cs;
            const int orow = p.W * p.out_cs;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = PAIR ? (kq & 1) * 4 : (by * NT + nt) * 16 + kq * 4;
                const bool cok = co < p.nout_p;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias) {
                    bv.x = co + 0 < p.nbias ? p.bias[co + 0] : 0.f; bv.y = co + 1 < p.nbias ? p.bias[co + 1] : 0.f;
                    bv.z = co + 2 < p.nbias ? p.bias[co + 2] : 0.f; bv.w = co + 3 < p.nbias ? p.bias[co + 3] : 0.f;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    // this wave's M-tile mt is box plane td = wave, row th = mt
                    if (cok && (box_full || (d0 + wave < p.D && h0 + mt < p.H && gw < p.W))) {
                        float4 o;
                        o.x = acc[mt][nt][0] + bv.x; o.y = acc[mt][nt][1] + bv.y;
                        o.z = acc[mt][nt][2] + bv.z; o.w = acc[mt][nt][3] + bv.w;
#ifndef CTU_ABL_NOSTORE
                        *reinterpret_cast<float4*>(obase + mt * orow + co) = o;
#endif
                        s1[nt][0] += o.x; s1[nt][1] += o.y; s1[nt][2] += o.z; s1[nt][3] += o.w;
                        s2[nt][0] += o.x * o.x; s2[nt][1] += o.y * o.y; s2[nt][2] += o.z * o.z; s2[nt][3] += o.w * o.w;
                    }
                    acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
#ifdef CTU_STAMP
        STAMP(tq5);
        ph[0] += tq1 - tq0; ph[1] += tq2 - tq1; ph[2] += tq3 - tq2; ph[3] += tq4 - tq3; ph[4] += tq5 - tq4;
        tq0 = tq5; ++nst;
#endif
        if (!has_next) break;
        tile = ntile; c = nc; box = nbox;
    }
#ifdef CTU_STAMP
    if (g_stamp_out && tid == 0) {
        unsigned long long* o = g_stamp_out + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 6;
        for (int k = 0; k < 5; ++k) o[k] = ph[k];
        o[5] = nst;
    }
#endif
    // ---- one BatchNorm partial row per block: reduce over the 16 voxel lanes, (pair: the two shifts,) the 4 waves
    if (p.stats) {
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a1 = s1[nt][r], a2 = s2[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
                if (PAIR) { a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32); }
                const int ch = PAIR ? (kq & 1) * 4 + r : nt * 16 + kq * 4 + r;      // channel inside this block's N range
                if (m == 0 && (!PAIR || kq < 2)) {
                    sRed[(wave * NT * 16 + ch) * 2 + 0] = a1;
                    sRed[(wave * NT * 16 + ch) * 2 + 1] = a2;
                }
            }
        __syncthreads();
        const int nch = PAIR ? 8 : NT * 16;
        if (tid < nch) {
            const int co = (PAIR ? 0 : by * NT * 16) + tid;
            if (co < p.nout_p) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) {
                    a1 += sRed[(w * NT * 16 + tid) * 2 + 0];
                    a2 += sRed[(w * NT * 16 + tid) * 2 + 1];
                }
                float* row = p.stats + (size_t)blockIdx.x * 2 * p.nout_p;
                st_row(p.tail.counter != nullptr, row + co, a1);
                st_row(p.tail.counter != nullptr, row + p.nout_p + co, a2);
            }
        }
        if (p.tail.counter) bn_fwd_tail(p.tail, p.stats, gridDim.x, p.nout_p, gridDim.x * gridDim.y);
    }
}


// ------------------------------------------------------------------ persistent k = 5 kernels (the legacy nets: recAE_v2_fixed,
// UNet4_2IC; models.py:393-538).  Same structure as conv3d_fwd_k3_persist (contiguous box range per block, register prefetch
// of the next stage's input, swapped operands, one stats row per block) with what 125 taps change:
//  * halo 2: the LDS image of a 4x4x16 box is 8x8x20 voxels (61 KB), of the pair layout's 4x4x32 box 8x8x36 (92 KB); a whole
//    chunk of weights (64 / 77 KB) does not fit beside it, so the weights stream through LDS ONE kd PLANE at a time (25 or 30
//    taps, double-buffered, prefetched into registers during the previous plane's MFMAs): one extra barrier per plane;
//  * one block (4 waves) per CU: a stage carries 1000-1200 MFMAs per wave (32-38 k cycles), so the exposed staging of a
//    single resident wave per SIMD is a few percent (the k = 3 kernels need 2-3 waves per SIMD to hide theirs);
//  * tap group = (kd, kw): the 8 input rows th' = 0..7 feed the 5 kh taps of the 4 M-tiles -- 8 + 5*NT fragment reads for
//    40*NT MFMAs;
//  * PAIR (C_out = 8): columns = (w-shift s, co) as for k = 3, 6 w-offsets -> 150 tile-taps instead of 250.
template <int NT, bool PAIR>
__global__ __launch_bounds__(256, 1) void conv3d_fwd_k5_persist(ConvP p, int ntiles, int tiles_per_block) {
    constexpr int KS = 5, PAD = 2;
    constexpr int MT = 4;
    constexpr int TD = 4, TH = 4, TW = PAIR ? 32 : 16;
    constexpr int HD = TD + 2 * PAD, HH = TH + 2 * PAD, HW = TW + 2 * PAD, HV = HD * HH * HW;
    constexpr int NTHR = 256;
    constexpr int KWN = PAIR ? KS + 1 : KS, PTAPS = KS * KWN, NTAP = KS * PTAPS;
    constexpr int WPL = PTAPS * NT * 128;                                 // floats of one kd plane of this block's weights
    constexpr int AITEMS = HV * 2, AITER = (AITEMS + NTHR - 1) / NTHR;
    constexpr int WITER = (WPL / 4 + NTHR - 1) / NTHR;
    static_assert(!PAIR || NT == 1, "pair layout has a single N tile");
    constexpr int VSK = PAIR ? 10 : 12, VS2 = VSK / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char k5_smem[];
    v2f* sA2 = reinterpret_cast<v2f*>(k5_smem);                           // [HV * VS2]
    v2f* sW2 = sA2 + HV * VS2;                                            // [2][WPL / 2]
    float* sRed = reinterpret_cast<float*>(sW2 + WPL);                    // [4 * NT * 16 * 2]
    typedef const volatile __attribute__((address_space(3))) v2f* lds_v2f_ptr;
    lds_v2f_ptr vA = (lds_v2f_ptr)sA2;
    lds_v2f_ptr vW = (lds_v2f_ptr)sW2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int by = blockIdx.y;
    const int n16 = p.n16;
    const int nchunk = p.rin_p >> 3;
    const int half = tid & 1;
    const bool has_xf = p.in_scale != nullptr;
    const int rowbase = ((wave * HH) * HW + (PAIR ? 2 * m : m)) * VS2 + kq;                 // float2 units
    const int bbase = kq * 16 + m;

    unsigned hoff[AITER];
    constexpr int FPW = 5, NFW = (AITER + FPW - 1) / FPW;
    unsigned fw[NFW];
#pragma unroll
    for (int q = 0; q < NFW; ++q) fw[q] = 0;
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
        const int i = tid + it * NTHR;
        const int v = (i < AITEMS) ? (i >> 1) : 0;
        const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
        hoff[it] = (unsigned)(((pd * p.H + ph) * p.W + pw) * p.in_cs + half * 4) * 4u;
        const unsigned face = (pd < PAD ? 1u : 0u) | (pd >= HD - PAD ? 2u : 0u) | (ph < PAD ? 4u : 0u) | (ph >= HH - PAD ? 8u : 0u) |
                              (pw < PAD ? 16u : 0u) | (pw >= HW - PAD ? 32u : 0u);
        fw[it / FPW] |= face << (6 * (it % FPW));
    }
    const unsigned safe_off = (unsigned)(((PAD * p.H + PAD) * p.W + PAD) * p.in_cs + half * 4) * 4u;

    f32x4 acc[MT][NT];
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    }

    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(ntiles, tile + tiles_per_block);
    int c = 0;

    float4 va[AITER], vw[WITER];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned vmask = 0;

    struct Box { int tx, ty, tz, n; };
    auto box_of = [&](int t) {
        Box b;
        b.tx = t % p.tiles_w; t /= p.tiles_w;
        b.ty = t % p.tiles_h; t /= p.tiles_h;
        b.tz = t % p.tiles_d; b.n = t / p.tiles_d;
        return b;
    };
    auto box_next = [&](Box b) {
        if (++b.tx == p.tiles_w) { b.tx = 0; if (++b.ty == p.tiles_h) { b.ty = 0; if (++b.tz == p.tiles_d) { b.tz = 0; ++b.n; } } }
        return b;
    };
    bool a_interior = false;
    auto load_a = [&](Box b, int cc) {
        const int d0 = b.tz * TD, h0 = b.ty * TH, w0 = b.tx * TW;
        if (has_xf) {
            sc = *reinterpret_cast<const float4*>(p.in_scale + cc * 8 + half * 4);
            sh = *reinterpret_cast<const float4*>(p.in_shift + cc * 8 + half * 4);
        }
        const long long org = (((long long)b.n * p.D + d0) * p.H + h0) * p.W + w0 - (long long)PAD * ((long long)p.H * p.W + p.W + 1);
        const char* base = reinterpret_cast<const char*>(p.in + org * p.in_cs + cc * 8);      // halo origin of the box
        a_interior = d0 >= PAD && h0 >= PAD && w0 >= PAD && d0 + TD + PAD <= p.D && h0 + TH + PAD <= p.H && w0 + TW + PAD <= p.W;
        if (a_interior) {
#pragma unroll
            for (int it = 0; it < AITER; ++it) {
                if ((it + 1) * NTHR <= AITEMS || tid + it * NTHR < AITEMS)
                    va[it] = *reinterpret_cast<const float4*>(base + hoff[it]);
                else
                    va[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
        vmask = 0;
        // full border box whose neighbours on the inner sides exist: the only out-of-volume items sit on halo faces that
        // coincide with volume faces (a box origin is a multiple of 4 >= PAD away from the low faces, or on them)
        if (d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W && (d0 + TD == p.D || d0 + TD + PAD <= p.D) &&
            (h0 + TH == p.H || h0 + TH + PAD <= p.H) && (w0 + TW == p.W || w0 + TW + PAD <= p.W)) {
            const unsigned bface = (d0 == 0 ? 1u : 0u) | (d0 + TD == p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) |
                                   (h0 + TH == p.H ? 8u : 0u) | (w0 == 0 ? 16u : 0u) | (w0 + TW == p.W ? 32u : 0u);
#pragma unroll
            for (int it = 0; it < AITER; ++it) {
                const bool ok = (fw[it / FPW] & (bface << (6 * (it % FPW)))) == 0u && ((it + 1) * NTHR <= AITEMS || tid + it * NTHR < AITEMS);
                va[it] = *reinterpret_cast<const float4*>(base + (ok ? hoff[it] : safe_off));
                vmask |= ok ? (1u << it) : 0u;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < AITER; ++it) {       // ragged box: per-item bounds
            const int i = tid + it * NTHR, v = i >> 1;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - PAD, gh = h0 + ph - PAD, gw = w0 + pw - PAD;
            const bool ok = i < AITEMS && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                val = *reinterpret_cast<const float4*>(base + hoff[it]);
                vmask |= 1u << it;
            }
            va[it] = val;
        }
    };
    const bool w_full = (by + 1) * NT <= n16;
    // kd plane `kd` of chunk cc: PTAPS taps x NT tiles x 128 floats of this block's N range
    auto load_w = [&](int cc, int kd) {
        const float* wsrc = p.wp + ((size_t)cc * NTAP + (size_t)kd * PTAPS) * n16 * 128 + (size_t)by * NT * 128;
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int i = (tid + it * NTHR) * 4;
            const int ts = i / (NT * 128), r = i % (NT * 128);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (((it + 1) * NTHR * 4 <= WPL || i < WPL) && (w_full || by * NT * 128 + r < n16 * 128))
                v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(wsrc) + (unsigned)(ts * n16 * 128 + r) * 4u);
            vw[it] = v;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int i = (tid + it * NTHR) * 4;
            if ((it + 1) * NTHR * 4 <= WPL || i < WPL) *reinterpret_cast<float4*>(&sW2[buf * (WPL / 2) + (i >> 1)]) = vw[it];
        }
    };

    Box box = box_of(tile);
    load_a(box, 0);
    load_w(0, 0);

    while (true) {
        __syncthreads();                       // the previous stage's readers are done with sA and both sW buffers
        {
            auto put = [&](int it, float4 val) {
                const int i = tid + it * NTHR;
                if ((it + 1) * NTHR <= AITEMS || i < AITEMS) {
                    if (PAIR) {
                        sA2[(i >> 1) * VS2 + half * 2] = v2f{val.x, val.y};
                        sA2[(i >> 1) * VS2 + half * 2 + 1] = v2f{val.z, val.w};
                    } else {
                        *reinterpret_cast<float4*>(&sA2[(i >> 1) * VS2 + half * 2]) = val;
                    }
                }
            };
            if (a_interior) {
#pragma unroll
                for (int it = 0; it < AITER; ++it) put(it, has_xf ? xform4(va[it], sc, sh, p.in_relu) : va[it]);
            } else if (has_xf) {
#pragma unroll
                for (int it = 0; it < AITER; ++it) {
                    const float4 t = xform4(va[it], sc, sh, p.in_relu);
                    const bool ok = (vmask >> it) & 1u;
                    put(it, make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f));
                }
            } else {
#pragma unroll
                for (int it = 0; it < AITER; ++it) {
                    const bool ok = (vmask >> it) & 1u;
                    put(it, make_float4(ok ? va[it].x : 0.f, ok ? va[it].y : 0.f, ok ? va[it].z : 0.f, ok ? va[it].w : 0.f));
                }
            }
        }
        store_w(0);
        __syncthreads();
        // ---- prefetch the next stage's input while this one computes
        int ntile = tile, nc = c + 1;
        Box nbox = box;
        if (nc == nchunk) { nc = 0; ntile = tile + 1; nbox = box_next(box); }
        const bool has_next = ntile < tile_end;
        if (has_next) load_a(nbox, nc);
#pragma unroll 1
        for (int kd = 0; kd < KS; ++kd) {
            // next plane of weights (or plane 0 of the next stage) into registers under this plane's MFMAs
            if (kd + 1 < KS) load_w(c, kd + 1);
            else if (has_next) load_w(nc, 0);
            const int wb = (kd & 1) * (WPL / 2);
            v2f ar[2][TH + KS - 1], br[2][KS][NT];
            auto load_group = [&](int kw, v2f (&aa)[TH + KS - 1], v2f (&bb)[KS][NT]) {
#pragma unroll
                for (int r = 0; r < TH + KS - 1; ++r) aa[r] = vA[rowbase + ((kd * HH + r) * HW + kw) * VS2];
#pragma unroll
                for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bb[kh][nt] = vW[wb + ((kh * KWN + kw) * NT + nt) * 64 + bbase];
            };
            load_group(0, ar[0], br[0]);
#if CTU_PIN_FWD
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int kw = 0; kw < KWN; ++kw) {
                if (kw + 1 < KWN) load_group(kw + 1, ar[(kw + 1) & 1], br[(kw + 1) & 1]);
                v2f (&aa)[TH + KS - 1] = ar[kw & 1];
                v2f (&bb)[KS][NT] = br[kw & 1];
#pragma unroll
                for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[kh][nt].x, aa[mt + kh].x, acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[kh][nt].y, aa[mt + kh].y, acc[mt][nt], 0, 0, 0);
                }
#if CTU_PIN_FWD
                // the next group's fragment reads one per MFMA under this group's MFMAs (see conv3d_fwd_k3_persist)
#pragma unroll
                for (int q = 0; q < 2 * KS * MT * NT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (kw + 1 < KWN && q < TH + KS - 1 + KS * NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#endif
            }
            if (kd + 1 < KS) {
                store_w((kd + 1) & 1);         // (its last readers finished before the barrier that ended plane kd - 1)
                __syncthreads();
            }
        }
        if (c == nchunk - 1) {
            const int n_img = box.n, d0 = box.tz * TD, h0 = box.ty * TH, w0 = box.tx * TW;
            const int gw = PAIR ? (w0 + 2 * m + (kq >> 1)) : (w0 + m);
            const bool box_full = d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W;
            float* obase = p.out + ((((size_t)n_img * p.D + d0 + wave) * p.H + h0) * p.W + gw) * p.out_cs;
            const int orow = p.W * p.out_cs;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = PAIR ? (kq & 1) * 4 : (by * NT + nt) * 16 + kq * 4;
                const bool cok = co < p.nout_p;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias) {
                    bv.x = co + 0 < p.nbias ? p.bias[co + 0] : 0.f; bv.y = co + 1 < p.nbias ? p.bias[co + 1] : 0.f;
                    bv.z = co + 2 < p.nbias ? p.bias[co + 2] : 0.f; bv.w = co + 3 < p.nbias ? p.bias[co + 3] : 0.f;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (cok && (box_full || (d0 + wave < p.D && h0 + mt < p.H && gw < p.W))) {
                        float4 o;
                        o.x = acc[mt][nt][0] + bv.x; o.y = acc[mt][nt][1] + bv.y;
                        o.z = acc[mt][nt][2] + bv.z; o.w = acc[mt][nt][3] + bv.w;
                        *reinterpret_cast<float4*>(obase + mt * orow + co) = o;
                        s1[nt][0] += o.x; s1[nt][1] += o.y; s1[nt][2] += o.z; s1[nt][3] += o.w;
                        s2[nt][0] += o.x * o.x; s2[nt][1] += o.y * o.y; s2[nt][2] += o.z * o.z; s2[nt][3] += o.w * o.w;
                    }
                    acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (!has_next) break;
        tile = ntile; c = nc; box = nbox;
    }
    if (p.stats) {
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a1 = s1[nt][r], a2 = s2[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
                if (PAIR) { a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32); }
                const int ch = PAIR ? (kq & 1) * 4 + r : nt * 16 + kq * 4 + r;
                if (m == 0 && (!PAIR || kq < 2)) {
                    sRed[(wave * NT * 16 + ch) * 2 + 0] = a1;
                    sRed[(wave * NT * 16 + ch) * 2 + 1] = a2;
                }
            }
        __syncthreads();
        const int nch = PAIR ? 8 : NT * 16;
        if (tid < nch) {
            const int co = (PAIR ? 0 : by * NT * 16) + tid;
            if (co < p.nout_p) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    a1 += sRed[(w * NT * 16 + tid) * 2 + 0];
                    a2 += sRed[(w * NT * 16 + tid) * 2 + 1];
                }
                float* row = p.stats + (size_t)blockIdx.x * 2 * p.nout_p;
                st_row(p.tail.counter != nullptr, row + co, a1);
                st_row(p.tail.counter != nullptr, row + p.nout_p + co, a2);
            }
        }
        if (p.tail.counter) bn_fwd_tail(p.tail, p.stats, gridDim.x, p.nout_p, gridDim.x * gridDim.y);
    }
}

// LDS bytes of conv3d_fwd_k5_persist<NT, PAIR>
template <int NT, bool PAIR>
constexpr size_t k5_persist_lds() {
    return (size_t)(8 * 8 * ((PAIR ? 32 : 16) + 4)) * (PAIR ? 10 : 12) * 4 + (size_t)2 * 5 * (PAIR ? 6 : 5) * NT * 128 * 4 +
           (size_t)4 * NT * 16 * 2 * 4;
}
// raise the kernel's dynamic-LDS limit once (above the 64 KB default)
template <int NT, bool PAIR>
static hipError_t k5_attr() {
    static hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_fwd_k5_persist<NT, PAIR>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)k5_persist_lds<NT, PAIR>());
    return rc;
}


template <int KS>
__global__ void pack_conv_w_pair8_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci,
                                         const int32_t* __restrict__ cinv, int nchunk, int mode) {
    pack_conv_w_pair8_elem<KS>(blockIdx.x * blockDim.x + threadIdx.x, w, wp, Co, Ci, cinv, nchunk, mode);
}

// ---- every weight tensor of the network in ONE launch: job table by value in the kernel arguments
constexpr int PACK_MAXJ = 48;
struct PackTable { ctu_pack_job j[PACK_MAXJ]; };

__global__ void pack_batch_kernel(PackTable tb) {
    const ctu_pack_job& q = tb.j[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (q.kind == 1) {
        const int n16 = (q.nout_p + 15) / 16;
        pack_convt_w_elem(idx, q.w, q.wp, q.Ci, q.Co, q.cinv, q.rin_p, n16 <= 1 ? 1 : (n16 <= 2 ? 2 : (n16 <= 4 ? 4 : 8)), q.mode);
    } else if (q.layout == 1) {
        if (q.k == 3) pack_conv_w_pair8_elem<3>(idx, q.w, q.wp, q.Co, q.Ci, q.cinv, q.rin_p / 8, q.mode);
        else pack_conv_w_pair8_elem<5>(idx, q.w, q.wp, q.Co, q.Ci, q.cinv, q.rin_p / 8, q.mode);
    } else if (q.k == 3) {
        pack_conv_w_elem<3>(idx, q.w, q.wp, q.Co, q.Ci, q.cinv, q.rin_p / 8, (q.nout_p + 15) / 16, q.mode);
    } else {
        pack_conv_w_elem<5>(idx, q.w, q.wp, q.Co, q.Ci, q.cinv, q.rin_p / 8, (q.nout_p + 15) / 16, q.mode);
    }
}

// ------------------------------------------------------------------ packing


template <int KS>
__global__ void pack_conv_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci,
                                   const int32_t* __restrict__ cinv, int nchunk, int n16, int mode) {
    pack_conv_w_elem<KS>(blockIdx.x * blockDim.x + threadIdx.x, w, wp, Co, Ci, cinv, nchunk, n16, mode);
}

// Launch shape: output-channel tiles per block (NT) and spatial tile.  Big volumes take the widest tile
// and up to 4 N-tiles per block (best operand reuse); small volumes (the 8^3/16^3 bottleneck levels)
// shrink both so that the grid still covers the 256 CUs.
inline void pick_launch(int N, int D, int H, int W, int k, int nout_p, int* nt, int* td, int* th, int* tw) {
    const int n16 = (nout_p + 15) / 16;
    *nt = n16 == 1 ? 1 : (n16 == 2 ? 2 : 4);
    pick_tile(W, td, th, tw);
    auto blocks = [&](int nt_, int td_, int th_, int tw_) {
        return (long)N * ceil_div(D, td_) * ceil_div(H, th_) * ceil_div(W, tw_) * ceil_div(n16, nt_);
    };
    if (*tw == 16 && *nt > 2) *nt = 2;              // this width runs the persistent kernels (at most 2 N-tiles)
    while (blocks(*nt, *td, *th, *tw) < 256 && *nt > 1) *nt >>= 1;
    if (blocks(*nt, *td, *th, *tw) < 256 && *tw > 4) { *td = 4; *th = 4; *tw = 4; }
}

// ------------------------------------------------------------------ weight gradient
struct WgP {
    const float* in;
    const float* in_scale;
    const float* in_shift;
    const float* g;
    float* ws;
    int in_cs, cin_p, in_relu, g_cs, cout_p;
    int N, D, H, W;
    int tiles_d, tiles_h, tiles_w, ntiles;
    int n_ci_t, n_co_t;
    // lazy BatchNorm+ReLU backward (conv3d_wgrad_k3s_kernel<.., LZ = true>): g is the gradient w.r.t. the ACTIVATED output,
    // lz_y the layer's raw output (same geometry and channel stride as g), lz_scale / lz_shift its BatchNorm vectors,
    // lz_coef = bn_bwd_finalize's [5][lz_cp] rows (k0, k1, k2, A, B); the raw-output gradient is written to lz_out
    const float* lz_y;
    float* lz_out;
    const float* lz_scale;
    const float* lz_shift;
    const float* lz_coef;
    int lz_cp;
};

// One block: one (ci-tile of 16, co-tile of 16) pair, KDS kd-planes of taps, a strided set of
// spatial tiles.  MFMA view: M = 16 input channels, N = 16 output channels, K = voxels.
template <int KS, int KDS, int TD, int TH, int TW>
__global__ __launch_bounds__(256, CTU_WG_OCC) void conv3d_wgrad_kernel(WgP p) {
    constexpr int PAD = (KS - 1) / 2;
    constexpr int HD = TD + KDS - 1, HH = TH + KS - 1, HW = TW + KS - 1;
    constexpr int HV = HD * HH * HW;
    constexpr int NVOX = TD * TH * TW;
    constexpr int BT = KDS * KS * KS;          // taps handled by this block
    constexpr int KSTEPS = NVOX / 4;           // MFMA K-steps per tile
    constexpr int KPW = KSTEPS / 4;            // per wave
    static_assert(TW % 4 == 0 && KSTEPS % 4 == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) float sA[HV * 16];
    __shared__ __attribute__((aligned(16))) float sG[NVOX * 16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int cit = blockIdx.y % p.n_ci_t, cot = blockIdx.y / p.n_ci_t;
    const int kd0 = blockIdx.z * KDS;
    const int ci0 = cit * 16, co0 = cot * 16;

    f32x4 acc[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging roles: 4 threads cover the 16 channels of one voxel
    const int quad = tid & 3;
    const bool a_ok = (ci0 + quad * 4) < p.cin_p, g_ok = (co0 + quad * 4) < p.cout_p;
    const bool has_xf = p.in_scale != nullptr;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_xf && a_ok) {
        sc = *reinterpret_cast<const float4*>(p.in_scale + ci0 + quad * 4);
        sh = *reinterpret_cast<const float4*>(p.in_shift + ci0 + quad * 4);
    }

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        int bx = tile;
        const int tx = bx % p.tiles_w; bx /= p.tiles_w;
        const int ty = bx % p.tiles_h; bx /= p.tiles_h;
        const int tz = bx % p.tiles_d; bx /= p.tiles_d;
        const int n_img = bx;
        const int d0 = tz * TD, h0 = ty * TH, w0 = tx * TW;
        __syncthreads();
        // staging loads go out in batches, branch-free (items outside the volume / channel range read element 0 of the
        // image and are zeroed at the LDS write): a load -> store pair per iteration would expose one round trip each
        constexpr int SB = 8;
        const size_t img = (size_t)n_img * p.D * p.H * p.W;
        for (int it0 = tid; it0 < HV * 4; it0 += SB * 256) {
            f32x4 vv[SB];
            unsigned okm = 0;
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const int it = it0 + q * 256;
                const int v = (it < HV * 4 ? it : it0) >> 2;
                const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
                const int gd = d0 + pd + kd0 - PAD, gh = h0 + ph - PAD, gw = w0 + pw - PAD;
                const bool ok = a_ok && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
                okm |= ok ? (1u << q) : 0u;
                const size_t off = ok ? ((size_t)(gd * p.H + gh) * p.W + gw) * p.in_cs + ci0 + quad * 4 : (size_t)0;
                vv[q] = *reinterpret_cast<const f32x4*>(p.in + img * p.in_cs + off);
            }
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const int it = it0 + q * 256;
                if (it < HV * 4) {
                    float4 val = make_float4(vv[q][0], vv[q][1], vv[q][2], vv[q][3]);
                    if (has_xf) val = xform4(val, sc, sh, p.in_relu);
                    if (!((okm >> q) & 1u)) val = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(&sA[(it >> 2) * 16 + quad * 4]) = val;
                }
            }
        }
        for (int it0 = tid; it0 < NVOX * 4; it0 += SB * 256) {
            f32x4 vv[SB];
            unsigned okm = 0;
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const int it = it0 + q * 256;
                const int v = (it < NVOX * 4 ? it : it0) >> 2;
                const int tw = v % TW, th = (v / TW) % TH, td = v / (TW * TH);
                const int gd = d0 + td, gh = h0 + th, gw = w0 + tw;
                const bool ok = g_ok && gd < p.D && gh < p.H && gw < p.W;
                okm |= ok ? (1u << q) : 0u;
                const size_t off = ok ? ((size_t)(gd * p.H + gh) * p.W + gw) * p.g_cs + co0 + quad * 4 : (size_t)0;
                vv[q] = *reinterpret_cast<const f32x4*>(p.g + img * p.g_cs + off);
            }
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const int it = it0 + q * 256;
                if (it < NVOX * 4)
                    *reinterpret_cast<f32x4*>(&sG[(it >> 2) * 16 + quad * 4]) = ((okm >> q) & 1u) ? vv[q] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < KPW; ++ks) {
            const int vt = (wave * KPW + ks) * 4 + kq;
            const int tw = vt % TW, th = (vt / TW) % TH, td = vt / (TW * TH);
            const float b = sG[vt * 16 + i];
            const int ab = ((td * HH + th) * HW + tw) * 16 + i;
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const int kd = t / (KS * KS), kh = (t / KS) % KS, kw = t % KS;
                const float a = sA[ab + ((kd * HH + kh) * HW + kw) * 16];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // sum the 4 waves through LDS (RT taps per round, reusing sA) -> ONE partial slab [BT][16 ci][16 co] per block
    constexpr int RT = (HV * 16 / 1024) < 8 ? (HV * 16 / 1024) : 8;
    static_assert(RT >= 1 && 4 * RT * 256 <= HV * 16, "reduction scratch must fit in sA");
    const size_t slab = (size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* dst = p.ws + slab * (BT * 256);
    for (int t0 = 0; t0 < BT; t0 += RT) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < BT; ++t)
            if (t >= t0 && t < t0 + RT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sA[(wave * RT + (t - t0)) * 256 + (kq * 4 + r) * 16 + i] = acc[t][r];
            }
        __syncthreads();
        const int nt = (BT - t0) < RT ? (BT - t0) : RT;
        for (int e = tid; e < nt * 256; e += 256)
            dst[t0 * 256 + e] = (sA[e] + sA[RT * 256 + e]) + (sA[2 * RT * 256 + e] + sA[3 * RT * 256 + e]);
    }
}

// dW[co][ci][t] = sum over the gx slabs of its (kd-group, ci-tile, co-tile).  A block owns 64 consecutive
// slab elements (coalesced 256-B reads); 4 thread groups stride over the slabs, combined in a fixed order.
template <int KS, int KDS>
__global__ __launch_bounds__(1024) void conv3d_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                  int Co, int Ci, const int32_t* __restrict__ pmap,
                                                                  int cin_p, int n_ci_t, int gx) {
    constexpr int TAPS = KS * KS * KS, BT = KDS * KS * KS;
    __shared__ float red[RPARTS][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int el = blockIdx.x * 64 + e;                 // element inside the [BT][16][16] slab
    const int pz = blockIdx.y;                          // (z * n_pairs + pair)
    float s = 0.f;
    if (el < BT * 256) {
        const float* src = ws + (size_t)pz * gx * (BT * 256) + el;
        for (int k = part; k < gx; k += RPARTS) s += src[(size_t)k * (BT * 256)];
    }
    red[part][e] = s;
    __syncthreads();
    if (part == 0 && el < BT * 256) {
        const float tot = red_total(red, e);
        const int n_pairs = gridDim.y / (TAPS / BT);
        const int z = pz / n_pairs, pair = pz % n_pairs;
        const int cit = pair % n_ci_t, cot = pair / n_ci_t;
        const int tl = el >> 8, i = (el >> 4) & 15, j = el & 15;
        const int cip = cit * 16 + i, co = cot * 16 + j;
        // padded position -> logical channel (or -1)
        const int ci = (cip < cin_p) ? (pmap ? pmap[cip] : (cip < Ci ? cip : -1)) : -1;
        if (ci >= 0 && co < Co) dw[((size_t)co * Ci + ci) * TAPS + z * BT + tl] = tot;
    }
}

// k = 3 layers at least 16 voxels wide take the persistent, software-pipelined weight-gradient kernel
inline bool use_k3s(int k, int W, int cin_p, int cout_p) { return k == 3 && W >= 16 && (CTU_K3S_ALL || cin_p == 8 || cout_p == 8); }

inline int wgrad_gx(int ntiles, int pairs_z) {
    int gx = CTU_WG_BLOCKS / pairs_z;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    return gx;
}


// ------------------------------------------------------------------ weight gradient, narrow layers (k = 3)
// When a side has only 8 (padded) channels, a plain 16x16 MFMA tile is half empty on that side.  Here a
// tile axis carries (w-shift, channel) pairs instead:
//   M index (s, ci):  A[(s,ci)][k = v] = a_in[v + (kd, kh, t0 + s) - pad][ci]
//   N index (s', co): B[k = v][(s',co)] = gout[v - s' e_w][co]
//   => entry [(s,ci),(s',co)] = dW[kd, kh, kw = t0 + s + s'][ci][co]
// (re-indexing u = v - s' e_w; the u_w = W-1 plane that a box decomposition of v never reaches for s' = 1
//  only multiplies a_in at w = W, i.e. zero padding, for the entries that are kept).
//   SM = SN = 2 (8 -> 8):   1 MFMA per (kd,kh) gives kw 0,1,2           ->  9 MFMAs per K-step (27 before)
//   SM = 1, SN = 2 (C -> 8): t0 = 0 keeps s'=0 (kw 0); t0 = 1 gives kw 1,2 -> 18
//   SM = 2, SN = 1 (8 -> C): t0 = 0 gives kw 0,1; t0 = 1 keeps s=1 (kw 2)  -> 18
// Persistent blocks walk contiguous voxel boxes with register prefetch of the next box.
// UP: the weight gradient of the FUSED decoder up-convolution (upconv_fused.hip): in = COARSE activations, g = the
// fine-grid gradient read at one output parity (blockIdx.y carries the parity), the 8 taps of that parity's 2x2x2
// sub-cube of the coarse halo -> dW_eff[parity][tap][ci][co] slabs.
// UP = 2 (8 padded output channels, gradient stride 8): the tile's 16 columns are (w-parity, c_out) -- the two fine voxels
// 2x, 2x+1 are 16 contiguous floats -- one (p_d, p_h) parity per block, 2 x 2 x 3 taps (12 MFMAs per K-step instead of 2 x 8).
// LZ: lazy BatchNorm+ReLU backward (models.py:27-32's BatchNorm3d + ReLU, backward).  p.g is the gradient w.r.t. the
// ACTIVATED output; the raw-output gradient  gy = [y sc + sh > 0] k0 g + (A y + B)  (bn_bwd_finalize_kernel's k0, A, B) is
// formed while the gradient box is written to LDS -- and stored to p.lz_out for the data-gradient kernel that follows --
// so the separate apply pass over the layer (read g, read y, write g) disappears.  Every staged item stores the value of the
// address it fetched (volume-face items fetch the box origin and store that voxel's value again; blocks that share a
// gradient tile store identical values), so the stores need no predicate.  Full boxes and channel tiles only (host check).
template <int SM, int SN, int UP = 0, bool LZ = false>
__global__ __launch_bounds__(256, CTU_K3S_OCC) void conv3d_wgrad_k3s_kernel(WgP p, int tiles_per_block) {
    static_assert(!UP || (SM == 1 && SN == 1), "the fused up-convolution uses the full 16 x 16 channel tile");
    // box: 4 x 4 x 16 voxels; 4 x 4 x 8 for the full 16 x 16 channel tile, whose 27 accumulators leave fewer staging registers
    constexpr int TD = 4, TH = 4, TW = (SM == 2 && SN == 2) ? 16 : 8, HD = 6, HH = 6, HW = TW + 2, HV = HD * HH * HW;
    constexpr int KPR = TW / 4, NKS = TH * KPR;       // K-steps (4 voxels) per row / per plane
    constexpr int CM = 16 / SM, CN = 16 / SN;        // channels per block on the input / output side
    constexpr int GW = TW + (SN - 1), GV = TD * TH * GW;
    constexpr int QN = (SM == 2 && SN == 2) ? 1 : ((SM == 1 && SN == 1) ? 3 : 2);     // w positions per (kd, kh) row
    constexpr int NMF = UP == 2 ? 12 : (UP ? 8 : 9 * QN);
    constexpr int AQ = CM / 4, GQ = CN / 4;
    constexpr int AITEMS = HV * AQ, AITER = (AITEMS + 255) / 256;
    constexpr int GITEMS = GV * GQ, GITER = (GITEMS + 255) / 256;

    // two LDS stages: while the MFMAs read stage cur, the prefetched next box is transformed and written into stage
    // cur ^ 1 between them (ONE barrier per box, no exposed write phase).  Every thread stores every item slot
    // (the slots beyond the box are padding), so the MFMA loop is a single basic block.
    constexpr int SAF = AITER * 256 * 4, SGF = GITER * 256 * 4;     // floats per stage
    static_assert(SAF >= HV * CM && SGF >= GV * CN, "stage size");
    __shared__ __attribute__((aligned(16))) float sA[2 * SAF];
    __shared__ __attribute__((aligned(16))) float sG[2 * SGF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int cig = blockIdx.y % p.n_ci_t, cog = UP ? (blockIdx.y / p.n_ci_t) % p.n_co_t : blockIdx.y / p.n_ci_t;
    const int par = UP ? blockIdx.y / (p.n_ci_t * p.n_co_t) : 0;          // output parity (pz, py, px) = bits 2, 1, 0
    const int pz = UP == 2 ? (par >> 1) & 1 : (par >> 2) & 1, py = UP == 2 ? par & 1 : (par >> 1) & 1, px = UP == 2 ? 0 : par & 1;
    const int gH = UP ? 2 * p.H : p.H, gW = UP ? 2 * p.W : p.W, gs = UP ? 2 : 1;      // gradient grid: row strides, voxel step
    const int ci0 = cig * CM, co0 = cog * CN;
    const bool has_xf = p.in_scale != nullptr;

    f32x4 acc[NMF];
#pragma unroll
    for (int t = 0; t < NMF; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int aq = tid % AQ, gq = tid % GQ;          // 256 % AQ == 0: a thread keeps its channel quad
    const bool a_ok = (ci0 + aq * 4) < p.cin_p, g_ok = (co0 + gq * 4) < p.cout_p;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);      // identity without transform
    if (has_xf && a_ok) {
        sc = *reinterpret_cast<const float4*>(p.in_scale + ci0 + aq * 4);
        sh = *reinterpret_cast<const float4*>(p.in_shift + ci0 + aq * 4);
    }
    const int xf_relu = has_xf ? p.in_relu : 0;
    const int boff = (SN == 2) ? ((i < 8) ? i : i - 16) : i;

    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(p.ntiles, tile + tiles_per_block);
    float4 va[AITER], vg[GITER];
    // LZ: this thread's channel quad of the BatchNorm-backward vectors (a quad beyond the tensor aliases quad 0, whose
    // address it fetches and stores), and the raw outputs fetched beside the gradient items
    float4 vy[LZ ? GITER : 1];
    float4 l_sc = make_float4(0.f, 0.f, 0.f, 0.f), l_sh = l_sc, l_k0 = l_sc, l_A = l_sc, l_B = l_sc;
    if constexpr (LZ) {
        const int lzc = (co0 + ((UP && !g_ok) ? 0 : gq * 4)) % p.lz_cp;
        l_sc = *reinterpret_cast<const float4*>(p.lz_scale + lzc);
        l_sh = *reinterpret_cast<const float4*>(p.lz_shift + lzc);
        l_k0 = *reinterpret_cast<const float4*>(p.lz_coef + lzc);
        l_A = *reinterpret_cast<const float4*>(p.lz_coef + 3 * p.lz_cp + lzc);
        l_B = *reinterpret_cast<const float4*>(p.lz_coef + 4 * p.lz_cp + lzc);
    }
    const long long lz_ydelta = LZ ? reinterpret_cast<const char*>(p.lz_y) - reinterpret_cast<const char*>(p.g) : 0;
    const long long lz_odelta = LZ ? reinterpret_cast<const char*>(p.lz_out) - reinterpret_cast<const char*>(p.g) : 0;

    // staging items: BYTE offsets relative to the box's halo origin (non-negative, so the loads take the
    // scalar-base + 32-bit-offset addressing form and cost no VALU address arithmetic)
    unsigned aoff[AITER], goff[GITER];
    constexpr int FPW = 5;                         // 6-bit halo-face codes per 32-bit word
    unsigned fw[(AITER + FPW - 1) / FPW], gw0 = 0; // gw0 bit it: gradient item it is the w-shift column (SN = 2)
#pragma unroll
    for (int q = 0; q < (AITER + FPW - 1) / FPW; ++q) fw[q] = 0;
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
        const int e = tid + it * 256, v = (e < AITEMS) ? e / AQ : 0;
        const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
        aoff[it] = (unsigned)(((pd * p.H + ph) * p.W + pw) * p.in_cs + ci0 + aq * 4) * 4u;
        const unsigned face = (pd == 0 ? 1u : 0u) | (pd == HD - 1 ? 2u : 0u) | (ph == 0 ? 4u : 0u) | (ph == HH - 1 ? 8u : 0u) |
                              (pw == 0 ? 16u : 0u) | (pw == HW - 1 ? 32u : 0u);
        fw[it / FPW] |= face << (6 * (it % FPW));
    }
#pragma unroll
    for (int it = 0; it < GITER; ++it) {
        const int e = tid + it * 256, v = (e < GITEMS) ? e / GQ : 0;
        const int tw = v % GW, th = (v / GW) % TH, td = v / (GW * TH);
        goff[it] = (unsigned)(((gs * td * gH + gs * th) * gW + gs * tw) * p.g_cs + co0 + gq * 4) * 4u;
        if (SN == 2 && tw == 0) gw0 |= 1u << it;
    }
    // a full box's only out-of-volume items sit on halo faces that coincide with volume faces: a load from the box
    // origin (always inside) stands in for them and the LDS write zeroes them
    // (UP: channel quads beyond the tensors -- 8 padded output channels in a 16-wide tile -- are masked per thread
    //  instead of sending the whole block down the ragged path)
    const unsigned safe_a = (unsigned)(((p.H + 1) * p.W + 1) * p.in_cs + ci0 + ((UP && !a_ok) ? 0 : aq * 4)) * 4u;
    const unsigned safe_g = (unsigned)((SN - 1) * p.g_cs + co0 + ((UP && !g_ok) ? 0 : gq * 4)) * 4u;
    const bool ch_full = UP || (ci0 + CM <= p.cin_p && co0 + CN <= p.cout_p);       // uniform

    // box coordinates advance incrementally (no div/mod per box)
    struct Box { int tx, ty, tz, n; };
    Box box;
    {
        int t = tile;
        box.tx = t % p.tiles_w; t /= p.tiles_w;
        box.ty = t % p.tiles_h; t /= p.tiles_h;
        box.tz = t % p.tiles_d; box.n = t / p.tiles_d;
    }
    auto box_next = [&](Box b) {
        if (++b.tx == p.tiles_w) { b.tx = 0; if (++b.ty == p.tiles_h) { b.ty = 0; if (++b.tz == p.tiles_d) { b.tz = 0; ++b.n; } } }
        return b;
    };

    // The loads only FETCH (raw values + a validity bit per item); the lazy-BatchNorm transform happens when the
    // values are written to LDS one stage later, so no wave ever waits for global memory inside load().
    unsigned amask = 0, gmask = 0;                 // bit it: item it of va[] / vg[] lies inside the volume
    const char* pf_abase = nullptr;                // uniform state of the box being fetched
    const char* pf_gbase = nullptr;
    unsigned pf_bface = 0;
    // prep(): scalar part of a fetch.  Returns whether the box is full (inside the volume up to its halo faces), the
    // precondition of the branch-free per-item forms below; ragged boxes go through load_ragged().
    auto prep = [&](Box b) -> bool {
        const int d0 = b.tz * TD, h0 = b.ty * TH, w0 = b.tx * TW;
        const long long org = (((long long)b.n * p.D + d0) * p.H + h0) * p.W + w0;
        pf_abase = reinterpret_cast<const char*>(p.in + (org - ((long long)p.H * p.W + p.W + 1)) * p.in_cs);
        if (UP) {
            const long long gorg = (((long long)b.n * 2 * p.D + 2 * d0 + pz) * gH + 2 * h0 + py) * gW + 2 * w0 + px;
            pf_gbase = reinterpret_cast<const char*>(p.g + gorg * p.g_cs);
        } else {
            pf_gbase = reinterpret_cast<const char*>(p.g + (org - (SN - 1)) * p.g_cs);
        }
        pf_bface = (d0 == 0 ? 1u : 0u) | (d0 + TD == p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) | (h0 + TH == p.H ? 8u : 0u) |
                   (w0 == 0 ? 16u : 0u) | (w0 + TW == p.W ? 32u : 0u);
        amask = 0; gmask = 0;
        return ch_full && d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W;
    };
    // a full box's only out-of-volume items sit on halo faces that coincide with volume faces: they fetch the box
    // origin instead (no branch) and are zeroed when written to LDS
    auto fetch_a = [&](int it) {
        const bool ok = (fw[it / FPW] & (pf_bface << (6 * (it % FPW)))) == 0u && (!UP || a_ok);
        va[it] = *reinterpret_cast<const float4*>(pf_abase + (ok ? aoff[it] : safe_a));
        amask |= ok ? (1u << it) : 0u;
    };
    auto fetch_g = [&](int it) {
        const bool ok = !((pf_bface & 16u) && ((gw0 >> it) & 1u)) && (!UP || g_ok);
        const unsigned off = ok ? goff[it] : safe_g;
        vg[it] = *reinterpret_cast<const float4*>(pf_gbase + off);
        if constexpr (LZ) vy[it] = *reinterpret_cast<const float4*>(pf_gbase + lz_ydelta + off);
        gmask |= ok ? (1u << it) : 0u;
    };
    auto load_ragged = [&](Box b) {                // per-item bounds checks (volume not a multiple of the box, channel tails)
        const int d0 = b.tz * TD, h0 = b.ty * TH, w0 = b.tx * TW;
#pragma unroll
        for (int it = 0; it < AITER; ++it) {
            const int e = tid + it * 256, v = e / AQ;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd - 1, gh = h0 + ph - 1, gw = w0 + pw - 1;
            const bool ok = e < AITEMS && a_ok && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H &&
                            (unsigned)gw < (unsigned)p.W;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                val = *reinterpret_cast<const float4*>(pf_abase + aoff[it]);
                amask |= 1u << it;
            }
            va[it] = val;
        }
#pragma unroll
        for (int it = 0; it < GITER; ++it) {
            const int e = tid + it * 256, v = e / GQ;
            const int tw = v % GW, th = (v / GW) % TH, td = v / (GW * TH);
            const int gd = d0 + td, gh = h0 + th, gw = w0 + tw - (SN - 1);
            const bool ok = e < GITEMS && g_ok && gd < p.D && gh < p.H && (unsigned)gw < (unsigned)p.W;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                val = *reinterpret_cast<const float4*>(pf_gbase + goff[it]);
                gmask |= 1u << it;
            }
            vg[it] = val;
        }
    };

    // fragment reads: one ds_read_b32 each at a compile-time offset from two per-lane bases, issued one K-step ahead
    typedef const volatile __attribute__((address_space(3))) float* lds_f_ptr;
    lds_f_ptr vA = (lds_f_ptr)sA + (((wave + pz) * HH + py) * HW + kq + px) * CM + i;      // (+ the parity's sub-cube origin)
    lds_f_ptr vG = (lds_f_ptr)sG + ((wave * TH * GW + kq + (SN - 1)) * CN + boff);

    // one staged item -> LDS stage st: transform (identity scale/shift without BatchNorm), zero what lies outside
    auto put_a = [&](int it, int st) {
        const float4 t = xform4(va[it], sc, sh, xf_relu);
        const bool ok = (amask >> it) & 1u;
        *reinterpret_cast<float4*>(&sA[st * SAF + (tid + it * 256) * 4]) =
            make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    };
    auto put_g = [&](int it, int st) {
        const bool ok = (gmask >> it) & 1u;
        float4 t = vg[it];
        if constexpr (LZ) {
            t = bn_bwd_lazy4(vg[it], vy[it], l_sc, l_sh, l_k0, l_A, l_B);
            // (pf_gbase still describes the box these registers were fetched from: prep() runs once per loop pass)
            *reinterpret_cast<float4*>(const_cast<char*>(pf_gbase) + lz_odelta + (ok ? goff[it] : safe_g)) = t;
        }
        *reinterpret_cast<float4*>(&sG[st * SGF + (tid + it * 256) * 4]) =
            make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    };

    // One box: NKS K-steps of 4 voxels per wave (plane td = wave, row th = ks / KPR, columns (ks % KPR) * 4 + kq).
    // The NKS * NMF (K-step, tap) MFMAs form one flat stream; their A fragments go through a ring of R registers
    // refilled R entries ahead (the slot an MFMA has just consumed), the B fragment one K-step ahead.  Between the
    // MFMAs of the first half of the K-steps the next box's global loads are issued (FETCH), between those of the
    // second half that box is transformed and written to the other LDS stage -- no phase of its own for either.
    constexpr bool INLOOP_FETCH = (SM == 2 && SN == 2);      // (the wider tiles run out of registers)
    int cur = 0;
    auto stage = [&](auto fetch_tag) {
        constexpr bool FETCH = decltype(fetch_tag)::value;
        constexpr int R = 12, NJ = NKS * NMF;
        constexpr int KW0 = NKS / 2, NIT = AITER + GITER;
        constexpr int IPF = (NIT + KW0 - 1) / KW0;
        lds_f_ptr cA = vA + cur * SAF;
        lds_f_ptr cG = vG + cur * SGF;
        float ar[R], br[2];
        auto a_read = [&](int j) -> float {
            const int ks = j / NMF, t = j % NMF;
            const int th = ks / KPR, tw4 = (ks % KPR) * 4;
            const int r = UP == 2 ? t / 3 : (UP ? (t >> 1) : t / QN), q = UP == 2 ? t % 3 : (UP ? (t & 1) : t % QN),
                      kd = UP ? (r >> 1) : r / 3, kh = UP ? (r & 1) : r % 3;
            return cA[((kd * HH + th + kh) * HW + tw4 + q) * CM];
        };
        auto b_read = [&](int ks) -> float { return cG[((ks / KPR) * GW + (ks % KPR) * 4) * CN]; };
        br[0] = b_read(0);
#pragma unroll
        for (int j = 0; j < R; ++j) ar[j] = a_read(j);
        // the whole ring is in flight before the first MFMA (one exposed LDS latency per box): a fake use of
        // every slot keeps the scheduler from trickling the fill reads in between the first MFMAs
        static_assert(R == 12, "fake-use list");
        asm volatile("" : "+v"(ar[0]), "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3]), "+v"(ar[4]), "+v"(ar[5]), "+v"(ar[6]),
                     "+v"(ar[7]), "+v"(ar[8]), "+v"(ar[9]), "+v"(ar[10]), "+v"(ar[11]), "+v"(br[0]));
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 1 < NKS) {
                br[(ks + 1) & 1] = b_read(ks + 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int t = 0; t < NMF; ++t) {
                const int j = ks * NMF + t;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[j % R], br[ks & 1], acc[t], 0, 0, 0);
                if (j + R < NJ) ar[j % R] = a_read(j + R);
                // pin the order: MFMA, then the refill of the slot it consumed (and, in the K-steps that also transform and
                // write staged items, a few of their VALU instructions in the MFMA's shadow instead of one block at the end)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (j + R < NJ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#if CTU_PIN_VALU
                if (ks >= KW0) __builtin_amdgcn_sched_group_barrier(0x002, CTU_PIN_VALU, 0);
#endif
            }
            if (FETCH && ks < KW0) {
#pragma unroll
                for (int q = 0; q < IPF; ++q) {
                    const int it = ks * IPF + q;
                    if (it < AITER) fetch_a(it);
                    else if (it < NIT) fetch_g(it - AITER);
                }
            }
            if (ks >= KW0) {
                // the NIT staged items spread evenly over the NKS - KW0 K-steps (gradient items, the costlier ones with LZ, first)
                constexpr int NK2 = NKS - KW0;
#pragma unroll
                for (int it = (ks - KW0) * NIT / NK2; it < (ks - KW0 + 1) * NIT / NK2; ++it) {
                    if (it < GITER) put_g(it, cur ^ 1);
                    else put_a(it - GITER, cur ^ 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (tile < tile_end) {
        if (prep(box)) {
#pragma unroll
            for (int it = 0; it < AITER; ++it) fetch_a(it);
#pragma unroll
            for (int it = 0; it < GITER; ++it) fetch_g(it);
        } else {
            load_ragged(box);
        }
#pragma unroll
        for (int it = 0; it < AITER; ++it) put_a(it, 0);
#pragma unroll
        for (int it = 0; it < GITER; ++it) put_g(it, 0);
    }
    __syncthreads();
    while (tile < tile_end) {
        box = box_next(box);
        bool full = false;
        if (tile + 1 < tile_end) {
            full = prep(box);
            if (!full) load_ragged(box);
        }
        // (after the last box the second half rewrites stale registers into the unread stage)
        if (INLOOP_FETCH && full) {
            stage(std::true_type{});
        } else {
            if (INLOOP_FETCH == false && full) {
#pragma unroll
                for (int it = 0; it < AITER; ++it) fetch_a(it);
#pragma unroll
                for (int it = 0; it < GITER; ++it) fetch_g(it);
            }
            stage(std::false_type{});
        }
        __syncthreads();                           // stage cur fully read, stage cur ^ 1 fully written
        cur ^= 1;
        ++tile;
    }
    // 4 waves -> one slab [NMF][16][16] per block (through sA, 4 * NMF * 256 floats <= HV * CM for CM = 8 needs rounds)
    constexpr int RT = (2 * SAF / 1024) < NMF ? (2 * SAF / 1024) : NMF;
    static_assert(RT >= 1, "reduction scratch");
    const size_t slab = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    float* dst = p.ws + slab * (NMF * 256);
    for (int t0 = 0; t0 < NMF; t0 += RT) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NMF; ++t)
            if (t >= t0 && t < t0 + RT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sA[(wave * RT + (t - t0)) * 256 + (kq * 4 + r) * 16 + i] = acc[t][r];
            }
        __syncthreads();
        const int nt = (NMF - t0) < RT ? (NMF - t0) : RT;
        for (int e = tid; e < nt * 256; e += 256)
            dst[t0 * 256 + e] = (sA[e] + sA[RT * 256 + e]) + (sA[2 * RT * 256 + e] + sA[3 * RT * 256 + e]);
    }
}

template <int SM, int SN>
__global__ __launch_bounds__(1024) void conv3d_wgrad_k3s_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                      int Co, int Ci, const int32_t* __restrict__ cinv,
                                                                      int cin_p, int n_ci_g, int gx) {
    constexpr int QN = (SM == 2 && SN == 2) ? 1 : ((SM == 1 && SN == 1) ? 3 : 2);
    constexpr int NMF = 9 * QN;
    constexpr int CM = 16 / SM, CN = 16 / SN;
    __shared__ float red[RPARTS][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int el = blockIdx.x * 64 + e;
    const int pair = blockIdx.y;
    float s = 0.f;
    if (el < NMF * 256) {
        const float* src = ws + (size_t)pair * gx * (NMF * 256) + el;
        for (int k = part; k < gx; k += RPARTS) s += src[(size_t)k * (NMF * 256)];
    }
    red[part][e] = s;
    __syncthreads();
    if (part != 0 || el >= NMF * 256) return;
    const float tot = red_total(red, e);
    const int t = el >> 8, i = (el >> 4) & 15, j = el & 15;
    const int r = t / QN, q = t % QN;
    const int sm = (SM == 2) ? i / 8 : 0, cil = (SM == 2) ? i % 8 : i;
    const int sn = (SN == 2) ? j / 8 : 0, col = (SN == 2) ? j % 8 : j;
    int kw;
    bool ok = true;
    if (SM == 1 && SN == 1) kw = q;
    else if (SM == 2 && SN == 2) { kw = sm + sn; ok = !(sm == 0 && sn == 1); }
    else if (SN == 2) { if (q == 0) { kw = 0; ok = sn == 0; } else kw = 1 + sn; }
    else { if (q == 0) kw = sm; else { kw = 2; ok = sm == 1; } }
    const int cig = pair % n_ci_g, cog = pair / n_ci_g;
    const int cip = cig * CM + cil, co = cog * CN + col;
    const int ci = (cip < cin_p) ? (cinv ? cinv[cip] : (cip < Ci ? cip : -1)) : -1;
    if (ok && ci >= 0 && co < Co) dw[((size_t)co * Ci + ci) * 27 + r * 3 + kw] = tot;
}

// geometry of the persistent k = 3 kernel for a layer: tile shape (SM, SN), box width, boxes, channel-tile pairs, grid
struct K3sGeom { int sm, sn, tw, ntiles, n_ci_g, pairs, gx, tpb, nmf; };
static K3sGeom k3s_geom(int N, int D, int H, int W, int cin_p, int cout_p) {
    K3sGeom g;
    g.sm = cin_p == 8 ? 2 : 1; g.sn = cout_p == 8 ? 2 : 1;
    g.tw = (g.sm == 2 && g.sn == 2) ? 16 : 8;
    g.nmf = (g.sm == 2 && g.sn == 2) ? 9 : ((g.sm == 1 && g.sn == 1) ? 27 : 18);
    g.ntiles = N * ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, g.tw);
    g.n_ci_g = ceil_div(cin_p, 16 / g.sm);
    g.pairs = g.n_ci_g * ceil_div(cout_p, 16 / g.sn);
    g.gx = wgrad_gx(g.ntiles, g.pairs);
    // small layers: with fewer than 4 boxes per block the 27-tap slab a block writes (and the reduce kernel re-reads)
    // outweighs its MFMA work -- one block per CU instead of two (64->64 @16^3: 28.4 -> 23.8 us, 32->32 @32^3: 34.4 -> 30.4)
    if (g.ntiles / g.gx < 4) {
        g.gx = (CTU_WG_BLOCKS / 2) / g.pairs;
        if (g.gx < 1) g.gx = 1;
        if (g.gx > g.ntiles) g.gx = g.ntiles;
    }
    g.tpb = ceil_div(g.ntiles, g.gx);
    g.gx = ceil_div(g.ntiles, g.tpb);
    return g;
}

// lazy BatchNorm backward inside the k = 3 weight-gradient kernel: full boxes and full channel tiles only
static bool k3s_lazy_ok(int D, int H, int W, int k, int cin_p, int cout_p) {
    if (!use_k3s(k, W, cin_p, cout_p)) return false;
    const int sm = cin_p == 8 ? 2 : 1, sn = cout_p == 8 ? 2 : 1, tw = (sm == 2 && sn == 2) ? 16 : 8;
    return D % 4 == 0 && H % 4 == 0 && W % tw == 0 && cin_p % (16 / sm) == 0 && cout_p % (16 / sn) == 0;
}

template <int SM, int SN>
static int launch_wgrad_k3s(WgP p, float* dw, int Co, int Ci, const int32_t* cinv, hipStream_t st) {
    const K3sGeom g = k3s_geom(p.N, p.D, p.H, p.W, p.cin_p, p.cout_p);
    p.tiles_d = ceil_div(p.D, 4); p.tiles_h = ceil_div(p.H, 4); p.tiles_w = ceil_div(p.W, g.tw);
    p.ntiles = g.ntiles;
    p.n_ci_t = g.n_ci_g;
    if (p.lz_y) conv3d_wgrad_k3s_kernel<SM, SN, 0, true><<<dim3(g.gx, g.pairs), 256, 0, st>>>(p, g.tpb);
    else conv3d_wgrad_k3s_kernel<SM, SN><<<dim3(g.gx, g.pairs), 256, 0, st>>>(p, g.tpb);
    CTU_CHECK_LAUNCH("conv3d_wgrad_k3s");
    conv3d_wgrad_k3s_reduce_kernel<SM, SN><<<dim3(ceil_div(g.nmf * 256, 64), g.pairs), 64 * RPARTS, 0, st>>>(
        p.ws, dw, Co, Ci, cinv, p.cin_p, p.n_ci_t, g.gx);
    CTU_CHECK_LAUNCH("conv3d_wgrad_k3s_reduce");
    return CTU_OK;
}

// ------------------------------------------------------------------ weight gradient, narrow layers (k = 5: the legacy nets)
// conv3d_wgrad_k3s_kernel's scheme -- (w-shift, channel) tile axes for 8-channel sides, persistent blocks, two LDS stages, the
// A-fragment register ring -- with what 125 taps change: ONE kd plane of taps per block (blockIdx.z; 5 x QN accumulators
// instead of 25 x QN: the staged input box is the 4 planes d0 + kd - 2 .. d0 + kd + 1 with an (h, w) halo of 2), and per kh
// row the w offsets t0 (kw = t0 + s + s'):
//   SM = SN = 2 (8 -> 8):    t0 = 0 -> kw 0,1 (s' = 0);  t0 = 2 -> kw 2,3,4                        10 MFMAs per K-step (25)
//   SM = 1, SN = 2 (C -> 8): t0 = 0,1,2,3 -> kw 0, 1, (2,3), 4                                     20
//   SM = 2, SN = 1 (8 -> C): t0 = 0,2,3 -> kw (0,1), (2,3), 4 (s = 1 of t0 = 3)                    15
//   SM = SN = 1:             t0 = kw                                                               25
// An entry that uses the gradient shift s' = 1 misses the term of the volume's last w column (the box decomposition of v
// never reaches u_w = W - 1), which multiplies a_in at w = W - 1 + kw - 2: zero padding only for kw >= 3 -- so kw <= 2 always
// come from s' = 0 entries (with pad 1 the k = 3 kernel needs that for kw <= 1 only).
template <int SM, int SN>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_k5s_kernel(WgP p, int tiles_per_block) {
    constexpr int UP = 0, PAD = 2;
    const int kd = blockIdx.z;                     // the kd plane of taps this block accumulates
    // box: 4 x 4 x 16 voxels; 4 x 4 x 8 for the full 16 x 16 channel tile, whose 27 accumulators leave fewer staging registers
    constexpr int TD = 4, TH = 4, TW = (SM == 2 && SN == 2) ? 16 : 8, HD = TD, HH = TH + 2 * PAD, HW = TW + 2 * PAD, HV = HD * HH * HW;
    constexpr int KPR = TW / 4, NKS = TH * KPR;       // K-steps (4 voxels) per row / per plane
    constexpr int CM = 16 / SM, CN = 16 / SN;        // channels per block on the input / output side
    constexpr int GW = TW + (SN - 1), GV = TD * TH * GW;
    constexpr int QN = (SM == 2 && SN == 2) ? 2 : ((SM == 1 && SN == 1) ? 5 : (SN == 2 ? 4 : 3));     // w positions t0 per kh row
    constexpr int NMF = 5 * QN;
    constexpr int AQ = CM / 4, GQ = CN / 4;
    constexpr int AITEMS = HV * AQ, AITER = (AITEMS + 255) / 256;
    constexpr int GITEMS = GV * GQ, GITER = (GITEMS + 255) / 256;

    // two LDS stages: while the MFMAs read stage cur, the prefetched next box is transformed and written into stage
    // cur ^ 1 between them (ONE barrier per box, no exposed write phase).  Every thread stores every item slot
    // (the slots beyond the box are padding), so the MFMA loop is a single basic block.
    constexpr int SAF = AITER * 256 * 4, SGF = GITER * 256 * 4;     // floats per stage
    static_assert(SAF >= HV * CM && SGF >= GV * CN, "stage size");
    __shared__ __attribute__((aligned(16))) float sA[2 * SAF];
    __shared__ __attribute__((aligned(16))) float sG[2 * SGF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int cig = blockIdx.y % p.n_ci_t, cog = UP ? (blockIdx.y / p.n_ci_t) % p.n_co_t : blockIdx.y / p.n_ci_t;
    const int par = UP ? blockIdx.y / (p.n_ci_t * p.n_co_t) : 0;          // output parity (pz, py, px) = bits 2, 1, 0
    const int pz = UP == 2 ? (par >> 1) & 1 : (par >> 2) & 1, py = UP == 2 ? par & 1 : (par >> 1) & 1, px = UP == 2 ? 0 : par & 1;
    const int gH = UP ? 2 * p.H : p.H, gW = UP ? 2 * p.W : p.W, gs = UP ? 2 : 1;      // gradient grid: row strides, voxel step
    const int ci0 = cig * CM, co0 = cog * CN;
    const bool has_xf = p.in_scale != nullptr;

    f32x4 acc[NMF];
#pragma unroll
    for (int t = 0; t < NMF; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int aq = tid % AQ, gq = tid % GQ;          // 256 % AQ == 0: a thread keeps its channel quad
    const bool a_ok = (ci0 + aq * 4) < p.cin_p, g_ok = (co0 + gq * 4) < p.cout_p;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);      // identity without transform
    if (has_xf && a_ok) {
        sc = *reinterpret_cast<const float4*>(p.in_scale + ci0 + aq * 4);
        sh = *reinterpret_cast<const float4*>(p.in_shift + ci0 + aq * 4);
    }
    const int xf_relu = has_xf ? p.in_relu : 0;
    const int boff = (SN == 2) ? ((i < 8) ? i : i - 16) : i;

    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(p.ntiles, tile + tiles_per_block);
    float4 va[AITER], vg[GITER];

    // staging items: BYTE offsets relative to the box's halo origin (non-negative, so the loads take the
    // scalar-base + 32-bit-offset addressing form and cost no VALU address arithmetic)
    unsigned aoff[AITER], goff[GITER];
    constexpr int FPW = 5;                         // 6-bit halo-face codes per 32-bit word
    unsigned fw[(AITER + FPW - 1) / FPW], gw0 = 0; // gw0 bit it: gradient item it is the w-shift column (SN = 2)
#pragma unroll
    for (int q = 0; q < (AITER + FPW - 1) / FPW; ++q) fw[q] = 0;
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
        const int e = tid + it * 256, v = (e < AITEMS) ? e / AQ : 0;
        const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
        aoff[it] = (unsigned)((((pd + kd) * p.H + ph) * p.W + pw) * p.in_cs + ci0 + aq * 4) * 4u;      // (halo origin of kd = 0)
        // plane pd of this block is volume plane d0 + pd + kd - PAD: outside below for a box on the low d face when
        // pd + kd < PAD, outside above for a box on the high face when pd + kd >= TD + PAD
        const unsigned face = (pd + kd < PAD ? 1u : 0u) | (pd + kd >= TD + PAD ? 2u : 0u) | (ph < PAD ? 4u : 0u) |
                              (ph >= HH - PAD ? 8u : 0u) | (pw < PAD ? 16u : 0u) | (pw >= HW - PAD ? 32u : 0u);
        fw[it / FPW] |= face << (6 * (it % FPW));
    }
#pragma unroll
    for (int it = 0; it < GITER; ++it) {
        const int e = tid + it * 256, v = (e < GITEMS) ? e / GQ : 0;
        const int tw = v % GW, th = (v / GW) % TH, td = v / (GW * TH);
        goff[it] = (unsigned)(((gs * td * gH + gs * th) * gW + gs * tw) * p.g_cs + co0 + gq * 4) * 4u;
        if (SN == 2 && tw == 0) gw0 |= 1u << it;
    }
    // a full box's only out-of-volume items sit on halo faces that coincide with volume faces: a load from the box
    // origin (always inside) stands in for them and the LDS write zeroes them
    // (UP: channel quads beyond the tensors -- 8 padded output channels in a 16-wide tile -- are masked per thread
    //  instead of sending the whole block down the ragged path)
    const unsigned safe_a = (unsigned)(((PAD * p.H + PAD) * p.W + PAD) * p.in_cs + ci0 + aq * 4) * 4u;
    const unsigned safe_g = (unsigned)((SN - 1) * p.g_cs + co0 + ((UP && !g_ok) ? 0 : gq * 4)) * 4u;
    const bool ch_full = UP || (ci0 + CM <= p.cin_p && co0 + CN <= p.cout_p);       // uniform

    // box coordinates advance incrementally (no div/mod per box)
    struct Box { int tx, ty, tz, n; };
    Box box;
    {
        int t = tile;
        box.tx = t % p.tiles_w; t /= p.tiles_w;
        box.ty = t % p.tiles_h; t /= p.tiles_h;
        box.tz = t % p.tiles_d; box.n = t / p.tiles_d;
    }
    auto box_next = [&](Box b) {
        if (++b.tx == p.tiles_w) { b.tx = 0; if (++b.ty == p.tiles_h) { b.ty = 0; if (++b.tz == p.tiles_d) { b.tz = 0; ++b.n; } } }
        return b;
    };

    // The loads only FETCH (raw values + a validity bit per item); the lazy-BatchNorm transform happens when the
    // values are written to LDS one stage later, so no wave ever waits for global memory inside load().
    unsigned amask = 0, gmask = 0;                 // bit it: item it of va[] / vg[] lies inside the volume
    const char* pf_abase = nullptr;                // uniform state of the box being fetched
    const char* pf_gbase = nullptr;
    unsigned pf_bface = 0;
    // prep(): scalar part of a fetch.  Returns whether the box is full (inside the volume up to its halo faces), the
    // precondition of the branch-free per-item forms below; ragged boxes go through load_ragged().
    auto prep = [&](Box b) -> bool {
        const int d0 = b.tz * TD, h0 = b.ty * TH, w0 = b.tx * TW;
        const long long org = (((long long)b.n * p.D + d0) * p.H + h0) * p.W + w0;
        pf_abase = reinterpret_cast<const char*>(p.in + (org - (long long)PAD * ((long long)p.H * p.W + p.W + 1)) * p.in_cs);
        if (UP) {
            const long long gorg = (((long long)b.n * 2 * p.D + 2 * d0 + pz) * gH + 2 * h0 + py) * gW + 2 * w0 + px;
            pf_gbase = reinterpret_cast<const char*>(p.g + gorg * p.g_cs);
        } else {
            pf_gbase = reinterpret_cast<const char*>(p.g + (org - (SN - 1)) * p.g_cs);
        }
        pf_bface = (d0 == 0 ? 1u : 0u) | (d0 + TD == p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) | (h0 + TH == p.H ? 8u : 0u) |
                   (w0 == 0 ? 16u : 0u) | (w0 + TW == p.W ? 32u : 0u);
        amask = 0; gmask = 0;
        // (the two halo layers beyond a full box's inner sides must exist as a whole: true for volumes that are multiples of the box)
        return ch_full && d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W && (d0 + TD == p.D || d0 + TD + PAD <= p.D) &&
               (h0 + TH == p.H || h0 + TH + PAD <= p.H) && (w0 + TW == p.W || w0 + TW + PAD <= p.W);
    };
    // a full box's only out-of-volume items sit on halo faces that coincide with volume faces: they fetch the box
    // origin instead (no branch) and are zeroed when written to LDS
    auto fetch_a = [&](int it) {
        const bool ok = (fw[it / FPW] & (pf_bface << (6 * (it % FPW)))) == 0u && (!UP || a_ok);
        va[it] = *reinterpret_cast<const float4*>(pf_abase + (ok ? aoff[it] : safe_a));
        amask |= ok ? (1u << it) : 0u;
    };
    auto fetch_g = [&](int it) {
        const bool ok = !((pf_bface & 16u) && ((gw0 >> it) & 1u)) && (!UP || g_ok);
        vg[it] = *reinterpret_cast<const float4*>(pf_gbase + (ok ? goff[it] : safe_g));
        gmask |= ok ? (1u << it) : 0u;
    };
    auto load_ragged = [&](Box b) {                // per-item bounds checks (volume not a multiple of the box, channel tails)
        const int d0 = b.tz * TD, h0 = b.ty * TH, w0 = b.tx * TW;
#pragma unroll
        for (int it = 0; it < AITER; ++it) {
            const int e = tid + it * 256, v = e / AQ;
            const int pw = v % HW, t2 = v / HW, ph = t2 % HH, pd = t2 / HH;
            const int gd = d0 + pd + kd - PAD, gh = h0 + ph - PAD, gw = w0 + pw - PAD;
            const bool ok = e < AITEMS && a_ok && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H &&
                            (unsigned)gw < (unsigned)p.W;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                val = *reinterpret_cast<const float4*>(pf_abase + aoff[it]);
                amask |= 1u << it;
            }
            va[it] = val;
        }
#pragma unroll
        for (int it = 0; it < GITER; ++it) {
            const int e = tid + it * 256, v = e / GQ;
            const int tw = v % GW, th = (v / GW) % TH, td = v / (GW * TH);
            const int gd = d0 + td, gh = h0 + th, gw = w0 + tw - (SN - 1);
            const bool ok = e < GITEMS && g_ok && gd < p.D && gh < p.H && (unsigned)gw < (unsigned)p.W;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                val = *reinterpret_cast<const float4*>(pf_gbase + goff[it]);
                gmask |= 1u << it;
            }
            vg[it] = val;
        }
    };

    // fragment reads: one ds_read_b32 each at a compile-time offset from two per-lane bases, issued one K-step ahead
    typedef const volatile __attribute__((address_space(3))) float* lds_f_ptr;
    lds_f_ptr vA = (lds_f_ptr)sA + (((wave + pz) * HH + py) * HW + kq + px) * CM + i;      // (+ the parity's sub-cube origin)
    lds_f_ptr vG = (lds_f_ptr)sG + ((wave * TH * GW + kq + (SN - 1)) * CN + boff);

    // one staged item -> LDS stage st: transform (identity scale/shift without BatchNorm), zero what lies outside
    auto put_a = [&](int it, int st) {
        const float4 t = xform4(va[it], sc, sh, xf_relu);
        const bool ok = (amask >> it) & 1u;
        *reinterpret_cast<float4*>(&sA[st * SAF + (tid + it * 256) * 4]) =
            make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    };
    auto put_g = [&](int it, int st) {
        const bool ok = (gmask >> it) & 1u;
        *reinterpret_cast<float4*>(&sG[st * SGF + (tid + it * 256) * 4]) =
            make_float4(ok ? vg[it].x : 0.f, ok ? vg[it].y : 0.f, ok ? vg[it].z : 0.f, ok ? vg[it].w : 0.f);
    };

    // One box: NKS K-steps of 4 voxels per wave (plane td = wave, row th = ks / KPR, columns (ks % KPR) * 4 + kq).
    // The NKS * NMF (K-step, tap) MFMAs form one flat stream; their A fragments go through a ring of R registers
    // refilled R entries ahead (the slot an MFMA has just consumed), the B fragment one K-step ahead.  Between the
    // MFMAs of the first half of the K-steps the next box's global loads are issued (FETCH), between those of the
    // second half that box is transformed and written to the other LDS stage -- no phase of its own for either.
    constexpr bool INLOOP_FETCH = (SM == 2 && SN == 2);      // (the wider tiles run out of registers)
    int cur = 0;
    auto stage = [&](auto fetch_tag) {
        constexpr bool FETCH = decltype(fetch_tag)::value;
        constexpr int R = 12, NJ = NKS * NMF;
        constexpr int KW0 = NKS / 2, NIT = AITER + GITER;
        constexpr int IPF = (NIT + KW0 - 1) / KW0;
        lds_f_ptr cA = vA + cur * SAF;
        lds_f_ptr cG = vG + cur * SGF;
        float ar[R], br[2];
        auto a_read = [&](int j) -> float {
            const int ks = j / NMF, t = j % NMF;
            const int th = ks / KPR, tw4 = (ks % KPR) * 4;
            const int kh = t / QN, q = t % QN;
            // w offsets t0 of the MFMAs of one kh row: kw = t0 + s + s' (see the reduce kernel for the entries kept)
            const int t0 = (SM == 1) ? q : ((SN == 1 && q == 2) ? 3 : 2 * q);
            return cA[((th + kh) * HW + tw4 + t0) * CM];
        };
        auto b_read = [&](int ks) -> float { return cG[((ks / KPR) * GW + (ks % KPR) * 4) * CN]; };
        br[0] = b_read(0);
#pragma unroll
        for (int j = 0; j < R; ++j) ar[j] = a_read(j);
        // the whole ring is in flight before the first MFMA (one exposed LDS latency per box): a fake use of
        // every slot keeps the scheduler from trickling the fill reads in between the first MFMAs
        static_assert(R == 12, "fake-use list");
        asm volatile("" : "+v"(ar[0]), "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3]), "+v"(ar[4]), "+v"(ar[5]), "+v"(ar[6]),
                     "+v"(ar[7]), "+v"(ar[8]), "+v"(ar[9]), "+v"(ar[10]), "+v"(ar[11]), "+v"(br[0]));
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 1 < NKS) {
                br[(ks + 1) & 1] = b_read(ks + 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int t = 0; t < NMF; ++t) {
                const int j = ks * NMF + t;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[j % R], br[ks & 1], acc[t], 0, 0, 0);
                if (j + R < NJ) ar[j % R] = a_read(j + R);
                // pin the order: MFMA, then the refill of the slot it consumed (and, in the K-steps that also transform and
                // write staged items, a few of their VALU instructions in the MFMA's shadow instead of one block at the end)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (j + R < NJ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#if CTU_PIN_VALU
                if (ks >= KW0) __builtin_amdgcn_sched_group_barrier(0x002, CTU_PIN_VALU, 0);
#endif
            }
            if (FETCH && ks < KW0) {
#pragma unroll
                for (int q = 0; q < IPF; ++q) {
                    const int it = ks * IPF + q;
                    if (it < AITER) fetch_a(it);
                    else if (it < NIT) fetch_g(it - AITER);
                }
            }
            if (ks >= KW0) {
                // the NIT staged items spread evenly over the NKS - KW0 K-steps (gradient items, the costlier ones with LZ, first)
                constexpr int NK2 = NKS - KW0;
#pragma unroll
                for (int it = (ks - KW0) * NIT / NK2; it < (ks - KW0 + 1) * NIT / NK2; ++it) {
                    if (it < GITER) put_g(it, cur ^ 1);
                    else put_a(it - GITER, cur ^ 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (tile < tile_end) {
        if (prep(box)) {
#pragma unroll
            for (int it = 0; it < AITER; ++it) fetch_a(it);
#pragma unroll
            for (int it = 0; it < GITER; ++it) fetch_g(it);
        } else {
            load_ragged(box);
        }
#pragma unroll
        for (int it = 0; it < AITER; ++it) put_a(it, 0);
#pragma unroll
        for (int it = 0; it < GITER; ++it) put_g(it, 0);
    }
    __syncthreads();
    while (tile < tile_end) {
        box = box_next(box);
        bool full = false;
        if (tile + 1 < tile_end) {
            full = prep(box);
            if (!full) load_ragged(box);
        }
        // (after the last box the second half rewrites stale registers into the unread stage)
        if (INLOOP_FETCH && full) {
            stage(std::true_type{});
        } else {
            if (INLOOP_FETCH == false && full) {
#pragma unroll
                for (int it = 0; it < AITER; ++it) fetch_a(it);
#pragma unroll
                for (int it = 0; it < GITER; ++it) fetch_g(it);
            }
            stage(std::false_type{});
        }
        __syncthreads();                           // stage cur fully read, stage cur ^ 1 fully written
        cur ^= 1;
        ++tile;
    }
    // 4 waves -> one slab [NMF][16][16] per block (through sA, 4 * NMF * 256 floats <= HV * CM for CM = 8 needs rounds)
    constexpr int RT = (2 * SAF / 1024) < NMF ? (2 * SAF / 1024) : NMF;
    static_assert(RT >= 1, "reduction scratch");
    const size_t slab = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* dst = p.ws + slab * (NMF * 256);
    for (int t0 = 0; t0 < NMF; t0 += RT) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NMF; ++t)
            if (t >= t0 && t < t0 + RT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sA[(wave * RT + (t - t0)) * 256 + (kq * 4 + r) * 16 + i] = acc[t][r];
            }
        __syncthreads();
        const int nt = (NMF - t0) < RT ? (NMF - t0) : RT;
        for (int e = tid; e < nt * 256; e += 256)
            dst[t0 * 256 + e] = (sA[e] + sA[RT * 256 + e]) + (sA[2 * RT * 256 + e] + sA[3 * RT * 256 + e]);
    }
}

template <int SM, int SN>
__global__ __launch_bounds__(1024) void conv3d_wgrad_k5s_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                      int Co, int Ci, const int32_t* __restrict__ cinv,
                                                                      int cin_p, int n_ci_g, int gx) {
    constexpr int QN = (SM == 2 && SN == 2) ? 2 : ((SM == 1 && SN == 1) ? 5 : (SN == 2 ? 4 : 3));
    constexpr int NMF = 5 * QN;
    constexpr int CM = 16 / SM, CN = 16 / SN;
    __shared__ float red[RPARTS][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int el = blockIdx.x * 64 + e;
    const int pair = blockIdx.y, kd = blockIdx.z;
    float s = 0.f;
    if (el < NMF * 256) {
        const float* src = ws + ((size_t)kd * gridDim.y + pair) * gx * (NMF * 256) + el;
        for (int k = part; k < gx; k += RPARTS) s += src[(size_t)k * (NMF * 256)];
    }
    red[part][e] = s;
    __syncthreads();
    if (part != 0 || el >= NMF * 256) return;
    const float tot = red_total(red, e);
    const int t = el >> 8, i = (el >> 4) & 15, j = el & 15;
    const int kh = t / QN, q = t % QN;
    const int sm = (SM == 2) ? i / 8 : 0, cil = (SM == 2) ? i % 8 : i;
    const int sn = (SN == 2) ? j / 8 : 0, col = (SN == 2) ? j % 8 : j;
    int kw;
    bool ok = true;
    if (SM == 1 && SN == 1) kw = q;
    else if (SM == 2 && SN == 2) {
        kw = 2 * q + sm + sn;
        ok = q == 0 ? sn == 0 : !(sm == 0 && sn == 1);        // t0 = 0: kw 0, 1 (s' = 0);  t0 = 2: kw 2, 3 (from s = 1), 4
    } else if (SN == 2) { kw = q + sn; ok = sn == 0 ? q <= 2 : q >= 2; }     // t0 = q: kw 0, 1, 2 from s' = 0; 3, 4 from s' = 1
    else { kw = (q == 2 ? 3 : 2 * q) + sm; ok = q < 2 || sm == 1; }      // t0 = 0, 2, 3
    const int cig = pair % n_ci_g, cog = pair / n_ci_g;
    const int cip = cig * CM + cil, co = cog * CN + col;
    const int ci = (cip < cin_p) ? (cinv ? cinv[cip] : (cip < Ci ? cip : -1)) : -1;
    if (ok && ci >= 0 && co < Co) dw[((size_t)co * Ci + ci) * 125 + (kd * 5 + kh) * 5 + kw] = tot;
}

// k = 5 layers with an 8-channel side (W >= 16) take the (w-shift, channel) tiles too
inline bool use_k5s(int k, int W, int cin_p, int cout_p) { return k == 5 && W >= 16 && (cin_p == 8 || cout_p == 8); }

static K3sGeom k5s_geom(int N, int D, int H, int W, int cin_p, int cout_p) {
    K3sGeom g;
    g.sm = cin_p == 8 ? 2 : 1; g.sn = cout_p == 8 ? 2 : 1;
    g.tw = (g.sm == 2 && g.sn == 2) ? 16 : 8;
    g.nmf = 5 * ((g.sm == 2 && g.sn == 2) ? 2 : ((g.sm == 1 && g.sn == 1) ? 5 : (g.sn == 2 ? 4 : 3)));
    g.ntiles = N * ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, g.tw);
    g.n_ci_g = ceil_div(cin_p, 16 / g.sm);
    g.pairs = g.n_ci_g * ceil_div(cout_p, 16 / g.sn);
    g.gx = wgrad_gx(g.ntiles, g.pairs * 5);
    g.tpb = ceil_div(g.ntiles, g.gx);
    g.gx = ceil_div(g.ntiles, g.tpb);
    return g;
}

template <int SM, int SN>
static int launch_wgrad_k5s(WgP p, float* dw, int Co, int Ci, const int32_t* cinv, hipStream_t st) {
    const K3sGeom g = k5s_geom(p.N, p.D, p.H, p.W, p.cin_p, p.cout_p);
    p.tiles_d = ceil_div(p.D, 4); p.tiles_h = ceil_div(p.H, 4); p.tiles_w = ceil_div(p.W, g.tw);
    p.ntiles = g.ntiles;
    p.n_ci_t = g.n_ci_g;
    conv3d_wgrad_k5s_kernel<SM, SN><<<dim3(g.gx, g.pairs, 5), 256, 0, st>>>(p, g.tpb);
    CTU_CHECK_LAUNCH("conv3d_wgrad_k5s");
    conv3d_wgrad_k5s_reduce_kernel<SM, SN><<<dim3(ceil_div(g.nmf * 256, 64), g.pairs, 5), 64 * RPARTS, 0, st>>>(
        p.ws, dw, Co, Ci, cinv, p.cin_p, p.n_ci_t, g.gx);
    CTU_CHECK_LAUNCH("conv3d_wgrad_k5s_reduce");
    return CTU_OK;
}

// dW_eff[parity 8][tap 8][cin_p][nout_p] from the slabs of the UP kernel (fixed-order parallel reduction)
__global__ __launch_bounds__(1024) void upconv_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dweff,
                                                                  int cin_p, int nout_p, int n_ci_g, int n_co_g, int gx) {
    __shared__ float red[RPARTS][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int el = blockIdx.x * 64 + e;                  // element of the [8][16][16] slab
    const int yb = blockIdx.y;                           // (parity * n_co_g + cog) * n_ci_g + cig
    float s = 0.f;
    const float* src = ws + (size_t)yb * gx * (8 * 256) + el;
    for (int k = part; k < gx; k += RPARTS) s += src[(size_t)k * (8 * 256)];
    red[part][e] = s;
    __syncthreads();
    if (part != 0) return;
    const float tot = red_total(red, e);
    const int t = el >> 8, i = (el >> 4) & 15, j = el & 15;
    const int cig = yb % n_ci_g, cog = (yb / n_ci_g) % n_co_g, par = yb / (n_ci_g * n_co_g);
    const int rp = cig * 16 + i, co = cog * 16 + j;
    if (rp < cin_p && co < nout_p) dweff[((size_t)(par * 8 + t) * cin_p + rp) * nout_p + co] = tot;
}

// UP = 2 slabs [12 taps = (dz*2+dy)*3+dxx][16 ci][16 = (px, co)] -> dW_eff[(pz,py,px)][(dz,dy,dxx-px)][cin_p][8]
__global__ __launch_bounds__(1024) void upconv_wgrad_reduce_pw_kernel(const float* __restrict__ ws, float* __restrict__ dweff,
                                                                     int cin_p, int n_ci_g, int gx) {
    __shared__ float red[RPARTS][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int el = blockIdx.x * 64 + e;                  // element of the [12][16][16] slab
    const int yb = blockIdx.y;                           // par4 * n_ci_g + cig
    float s = 0.f;
    const float* src = ws + (size_t)yb * gx * (12 * 256) + el;
    for (int k = part; k < gx; k += RPARTS) s += src[(size_t)k * (12 * 256)];
    red[part][e] = s;
    __syncthreads();
    if (part != 0) return;
    const float tot = red_total(red, e);
    const int t = el >> 8, i = (el >> 4) & 15, j = el & 15;
    const int cig = yb % n_ci_g, par4 = yb / n_ci_g;
    const int px = j >> 3, co = j & 7, jx = t % 3 - px, rp = cig * 16 + i;
    if ((jx == 0 || jx == 1) && rp < cin_p)
        dweff[((size_t)((par4 * 2 + px) * 8 + (t / 3) * 2 + jx) * cin_p + rp) * 8 + co] = tot;
}

struct UpWgGeom { int ntiles, n_ci_g, n_co_g, pairs, gx, tpb, nmf; };
static UpWgGeom upwg_geom(int N, int D, int H, int W, int cin_p, int nout_p, bool pw) {
    UpWgGeom g;
    if (pw) {
        g.ntiles = N * ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 8);
        g.n_ci_g = ceil_div(cin_p, 16); g.n_co_g = 1; g.pairs = 4 * g.n_ci_g; g.nmf = 12;
        g.gx = wgrad_gx(g.ntiles, g.pairs);
        g.tpb = ceil_div(g.ntiles, g.gx);
        g.gx = ceil_div(g.ntiles, g.tpb);
        return g;
    }
    g.nmf = 8;
    g.ntiles = N * ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 8);
    g.n_ci_g = ceil_div(cin_p, 16);
    g.n_co_g = ceil_div(nout_p, 16);
    g.pairs = 8 * g.n_ci_g * g.n_co_g;
    g.gx = wgrad_gx(g.ntiles, g.pairs);
    // small layers: with fewer than 4 boxes per block the 27-tap slab a block writes (and the reduce kernel re-reads)
    // outweighs its MFMA work -- one block per CU instead of two (64->64 @16^3: 28.4 -> 23.8 us, 32->32 @32^3: 34.4 -> 30.4)
    if (g.ntiles / g.gx < 4) {
        g.gx = (CTU_WG_BLOCKS / 2) / g.pairs;
        if (g.gx < 1) g.gx = 1;
        if (g.gx > g.ntiles) g.gx = g.ntiles;
    }
    g.tpb = ceil_div(g.ntiles, g.gx);
    g.gx = ceil_div(g.ntiles, g.tpb);
    return g;
}

}  // namespace

// k == 5: the same boxes, one block per CU -- when the boxes fill the chip (a block owns one whole box per stage)
static bool use_persist5(int k, int nt, int tw, int N, int D, int H, int W) {
    return k == 5 && tw == 16 && nt <= 2 && (long)N * ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 16) >= 256;
}

// =================================================================== C ABI
extern "C" int ctu_conv3d_layout(int k, int nout_p, int W) {
    return ((k == 3 || k == 5) && nout_p == 8 && W >= 32) ? 1 : 0;
}

extern "C" const char* ctu_conv3d_fwd_kernel_name(int N, int D, int H, int W, int k, int nout_p, int layout) {
    static thread_local char buf[64];
    if (layout == 1) return k == 3 ? "conv3d_fwd_k3_persist<1, true>" : "conv3d_fwd_k5_persist<1, true>";
    int nt, td, th, tw;
    pick_launch(N, D, H, W, k, nout_p, &nt, &td, &th, &tw);
    if (use_persist5(k, nt, tw, N, D, H, W)) snprintf(buf, sizeof(buf), "conv3d_fwd_k5_persist<%d, false>", nt);
    else if (k == 3 && tw == 16 && nt <= 2) snprintf(buf, sizeof(buf), "conv3d_fwd_k3_persist<%d, false>", nt);
    else snprintf(buf, sizeof(buf), "conv3d_fwd_kernel<%d, %d, %d, %d, %d>", k, nt, td, th, tw);
    return buf;
}

extern "C" const char* ctu_conv3d_wgrad_kernel_name(int W, int k, int cin_p, int cout_p) {
    static thread_local char buf[64];
    if (use_k3s(k, W, cin_p, cout_p))
        snprintf(buf, sizeof(buf), "conv3d_wgrad_k3s_kernel<%d, %d>", cin_p == 8 ? 2 : 1, cout_p == 8 ? 2 : 1);
    else if (use_k5s(k, W, cin_p, cout_p))
        snprintf(buf, sizeof(buf), "conv3d_wgrad_k5s_kernel<%d, %d>", cin_p == 8 ? 2 : 1, cout_p == 8 ? 2 : 1);
    else {
        int td, th, tw;
        pick_tile(W, &td, &th, &tw);
        snprintf(buf, sizeof(buf), "conv3d_wgrad_kernel<%d, %d, %d, %d, %d>", k, k == 3 ? 3 : 1, td, th, tw);
    }
    return buf;
}

extern "C" size_t ctu_conv3d_packed_floats(int k, int rin_p, int nout_p, int layout) {
    if ((k != 3 && k != 5) || rin_p <= 0 || nout_p <= 0) return 0;
    if (layout == 1) return nout_p == 8 ? (size_t)(rin_p / 8) * k * k * (k + 1) * 128 : 0;
    return (size_t)(rin_p / 8) * k * k * k * ceil_div(nout_p, 16) * 128;
}

// grid of the persistent kernels: boxes per block and blocks in x
static void persist_grid(int ntiles, int ny, int blocks_per_cu, int* gx, int* tpb) {
    int g = (256 * blocks_per_cu) / ny;
    if (g < 1) g = 1;
    if (g > ntiles) g = ntiles;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

// k == 3 volumes wide enough for the 4x4x16 box with at most 2 N-tiles per block use the persistent kernel
static bool use_persist(int k, int nt, int tw) { return k == 3 && tw == 16 && nt <= 2; }

extern "C" int ctu_conv3d_num_blocks(int N, int D, int H, int W, int k, int nout_p, int layout) {
    int gx, tpb;
    if (layout == 1) {
        persist_grid(N * ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 32), 1, k == 3 ? 2 : 1, &gx, &tpb);
        return gx;                                   // the persistent kernels write ONE stats row per block
    }
    int nt, td, th, tw;
    pick_launch(N, D, H, W, k, nout_p, &nt, &td, &th, &tw);
    const int ntiles = N * ceil_div(D, td) * ceil_div(H, th) * ceil_div(W, tw);
    if (use_persist5(k, nt, tw, N, D, H, W)) {
        persist_grid(ntiles, ceil_div(ceil_div(nout_p, 16), nt), 1, &gx, &tpb);
        return gx;
    }
    if (use_persist(k, nt, tw)) {
        persist_grid(ntiles, ceil_div(ceil_div(nout_p, 16), nt), nt == 1 ? CTU_FWD_OCC1 : 2, &gx, &tpb);
        return gx;
    }
    return ntiles;
}

extern "C" int ctu_pack_conv3d_weight(const float* w, float* wp, int Co, int Ci, int k, const int32_t* cinv,
                                      int rin_p, int nout_p, int mode, int layout, void* stream) {
    CTU_REQUIRE(k == 3 || k == 5, "pack_conv3d_weight: k=%d unsupported (3 or 5)", k);
    CTU_REQUIRE(rin_p % 8 == 0 && nout_p % 8 == 0, "pack_conv3d_weight: padded channels must be multiples of 8");
    CTU_REQUIRE(w && wp && Co > 0 && Ci > 0, "pack_conv3d_weight: null/empty argument");
    CTU_REQUIRE(layout == 0 || (layout == 1 && nout_p == 8), "pack_conv3d_weight: layout %d needs nout_p=8", layout);
    hipStream_t st = (hipStream_t)stream;
    if (layout == 1) {
        const int tot = (int)ctu_conv3d_packed_floats(k, rin_p, nout_p, 1);
        if (k == 3) pack_conv_w_pair8_kernel<3><<<ceil_div(tot, 256), 256, 0, st>>>(w, wp, Co, Ci, cinv, rin_p / 8, mode);
        else pack_conv_w_pair8_kernel<5><<<ceil_div(tot, 256), 256, 0, st>>>(w, wp, Co, Ci, cinv, rin_p / 8, mode);
        CTU_CHECK_LAUNCH("pack_conv3d_weight(pair8)");
        return CTU_OK;
    }
    const int n16 = ceil_div(nout_p, 16);
    const int total = (int)ctu_conv3d_packed_floats(k, rin_p, nout_p, 0);
    const int nb = ceil_div(total, 256);
    if (k == 3) pack_conv_w_kernel<3><<<nb, 256, 0, st>>>(w, wp, Co, Ci, cinv, rin_p / 8, n16, mode);
    else pack_conv_w_kernel<5><<<nb, 256, 0, st>>>(w, wp, Co, Ci, cinv, rin_p / 8, n16, mode);
    CTU_CHECK_LAUNCH("pack_conv3d_weight");
    return CTU_OK;
}

extern "C" size_t ctu_convt_packed_floats(int rin_p, int nout_p);

extern "C" int ctu_pack_batch(const ctu_pack_job* jobs, int n, void* stream) {
    CTU_REQUIRE(jobs && n > 0, "pack_batch: no jobs");
    hipStream_t st = (hipStream_t)stream;
    for (int j0 = 0; j0 < n; j0 += PACK_MAXJ) {
        const int nj = (n - j0) < PACK_MAXJ ? (n - j0) : PACK_MAXJ;
        PackTable tb;
        size_t mx = 1;
        for (int j = 0; j < nj; ++j) {
            const ctu_pack_job& q = jobs[j0 + j];
            CTU_REQUIRE(q.w && q.wp && q.Co > 0 && q.Ci > 0 && q.rin_p % 8 == 0 && q.nout_p % 8 == 0, "pack_batch: bad job %d", j0 + j);
            size_t tot;
            if (q.kind == 1) {
                CTU_REQUIRE(q.nout_p <= 128, "pack_batch: convT nout_p=%d", q.nout_p);
                tot = ctu_convt_packed_floats(q.rin_p, q.nout_p);
            } else {
                CTU_REQUIRE(q.kind == 0 && (q.k == 3 || q.k == 5) && (q.layout == 0 || (q.layout == 1 && q.nout_p == 8)),
                            "pack_batch: bad conv job %d", j0 + j);
                tot = ctu_conv3d_packed_floats(q.k, q.rin_p, q.nout_p, q.layout);
            }
            if (tot > mx) mx = tot;
            tb.j[j] = q;
        }
        pack_batch_kernel<<<dim3((unsigned)((mx + 255) / 256), nj), 256, 0, st>>>(tb);
        CTU_CHECK_LAUNCH("pack_batch");
    }
    return CTU_OK;
}

template <int KS, int NT>
static int launch_fwd(const ConvP& p0, int td, int th, int tw, hipStream_t st) {
    ConvP p = p0;
    p.n16 = ceil_div(p.nout_p, 16);
    p.tiles_d = ceil_div(p.D, td); p.tiles_h = ceil_div(p.H, th); p.tiles_w = ceil_div(p.W, tw);
    dim3 grid(p.N * p.tiles_d * p.tiles_h * p.tiles_w, ceil_div(p.nout_p, 16 * NT));
    if (tw == 16) conv3d_fwd_kernel<KS, NT, 4, 4, 16><<<grid, 256, 0, st>>>(p);
    else if (tw == 8) conv3d_fwd_kernel<KS, NT, 4, 8, 8><<<grid, 256, 0, st>>>(p);
    else if (KS == 3 && NT == 1 && p.rin_p % 32 == 0) {
        // small deep-level volumes: 32 channels per barrier pair (4x fewer exposed staging latencies)
        if constexpr (KS == 3 && NT == 1) conv3d_fwd_kernel<3, 1, 4, 4, 4, 4><<<grid, 256, 0, st>>>(p);
    } else conv3d_fwd_kernel<KS, NT, 4, 4, 4><<<grid, 256, 0, st>>>(p);
    CTU_CHECK_LAUNCH("conv3d_fwd");
    return CTU_OK;
}

extern "C" int ctu_conv3d_fwd(const float* in, int in_cs, int rin_p, const float* in_scale, const float* in_shift,
                              int in_relu, const float* wp, const float* bias, int nbias, float* out, int out_cs,
                              int nout_p, float* stats, int N, int D, int H, int W, int k, int layout, const ctu_bn_tail* tail, void* stream) {
    CTU_REQUIRE(k == 3 || k == 5, "conv3d_fwd: k=%d unsupported (3 or 5)", k);
    CTU_REQUIRE(in && wp && out, "conv3d_fwd: null pointer");
    CTU_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "conv3d_fwd: empty volume %dx%dx%dx%d", N, D, H, W);
    CTU_REQUIRE(rin_p > 0 && rin_p % 8 == 0 && nout_p > 0 && nout_p % 8 == 0,
                "conv3d_fwd: channel counts must be positive multiples of 8 (rin_p=%d nout_p=%d)", rin_p, nout_p);
    CTU_REQUIRE(in_cs >= rin_p && in_cs % 4 == 0 && out_cs >= nout_p && out_cs % 4 == 0, "conv3d_fwd: bad channel stride");
    CTU_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)wp & 15) == 0 && ((uintptr_t)out & 15) == 0,
                "conv3d_fwd: in/wp/out must be 16-byte aligned");
    CTU_REQUIRE((int64_t)D * H * W * in_cs < (1LL << 31), "conv3d_fwd: one batch item must stay below 2^31 elements");
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3d_fwd: scale/shift must come together");
    ConvP p;
    p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.wp = wp; p.bias = bias; p.out = out; p.stats = stats;
    CTU_REQUIRE(!tail || (stats && tail->counter && tail->gamma && tail->beta && tail->scale && tail->shift && tail->mean &&
                          tail->invstd && tail->C > 0 && tail->C <= nout_p && tail->count > 0),
                "conv3d_fwd: incomplete BatchNorm tail");
    p.tail = tail_or_off(tail);
    p.in_cs = in_cs; p.rin_p = rin_p; p.in_relu = in_relu; p.out_cs = out_cs; p.nout_p = nout_p;
    p.nbias = bias ? nbias : 0;
    p.N = N; p.D = D; p.H = H; p.W = W;
    hipStream_t st = (hipStream_t)stream;
    if (layout == 1) {
        CTU_REQUIRE(nout_p == 8, "conv3d_fwd: layout 1 needs nout_p=8");
        p.n16 = 1;
        p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, 4); p.tiles_w = ceil_div(W, 32);
        const int ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
        int gx, tpb;
        if (k == 5) {
            persist_grid(ntiles, 1, 1, &gx, &tpb);     // 1 resident block per CU (123 KB of LDS)
            { const hipError_t arc = k5_attr<1, true>(); CTU_REQUIRE(arc == hipSuccess, "conv3d_fwd: cannot raise the dynamic LDS limit of the k=5 kernel"); }
            conv3d_fwd_k5_persist<1, true><<<gx, 256, k5_persist_lds<1, true>(), st>>>(p, ntiles, tpb);
            CTU_CHECK_LAUNCH("conv3d_fwd_k5_persist<1, pair>");
            return CTU_OK;
        }
        persist_grid(ntiles, 1, 2, &gx, &tpb);         // 2 resident blocks per CU (LDS)
        conv3d_fwd_k3_persist<1, true><<<gx, 256, 0, st>>>(p, ntiles, tpb);
        CTU_CHECK_LAUNCH("conv3d_fwd_k3_persist<1, pair>");
        return CTU_OK;
    }
    CTU_REQUIRE(layout == 0, "conv3d_fwd: unknown layout %d", layout);
    int NT, td, th, tw;
    pick_launch(N, D, H, W, k, nout_p, &NT, &td, &th, &tw);
    if (use_persist5(k, NT, tw, N, D, H, W)) {
        p.n16 = ceil_div(p.nout_p, 16);
        p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, 4); p.tiles_w = ceil_div(W, 16);
        const int ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
        const int ny = ceil_div(p.n16, NT);
        int gx, tpb;
        persist_grid(ntiles, ny, 1, &gx, &tpb);
        if (NT == 1) {
            { const hipError_t arc = k5_attr<1, false>(); CTU_REQUIRE(arc == hipSuccess, "conv3d_fwd: cannot raise the dynamic LDS limit of the k=5 kernel"); }
            conv3d_fwd_k5_persist<1, false><<<dim3(gx, ny), 256, k5_persist_lds<1, false>(), st>>>(p, ntiles, tpb);
        } else {
            { const hipError_t arc = k5_attr<2, false>(); CTU_REQUIRE(arc == hipSuccess, "conv3d_fwd: cannot raise the dynamic LDS limit of the k=5 kernel"); }
            conv3d_fwd_k5_persist<2, false><<<dim3(gx, ny), 256, k5_persist_lds<2, false>(), st>>>(p, ntiles, tpb);
        }
        CTU_CHECK_LAUNCH("conv3d_fwd_k5_persist");
        return CTU_OK;
    }
    if (use_persist(k, NT, tw)) {
        // large layers: persistent, register-prefetching kernel
        p.n16 = ceil_div(p.nout_p, 16);
        p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, 4); p.tiles_w = ceil_div(W, 16);
        const int ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
        const int ny = ceil_div(p.n16, NT);
        int gx, tpb;
        persist_grid(ntiles, ny, NT == 1 ? CTU_FWD_OCC1 : 2, &gx, &tpb);
        if (NT == 1) conv3d_fwd_k3_persist<1, false><<<dim3(gx, ny), 256, 0, st>>>(p, ntiles, tpb);
        else conv3d_fwd_k3_persist<2, false><<<dim3(gx, ny), 256, 0, st>>>(p, ntiles, tpb);
        CTU_CHECK_LAUNCH("conv3d_fwd_k3_persist");
        return CTU_OK;
    }
    if (k == 3) {
        if (NT == 1) return launch_fwd<3, 1>(p, td, th, tw, st);
        if (NT == 2) return launch_fwd<3, 2>(p, td, th, tw, st);
        return launch_fwd<3, 4>(p, td, th, tw, st);
    }
    if (NT == 1) return launch_fwd<5, 1>(p, td, th, tw, st);
    if (NT == 2) return launch_fwd<5, 2>(p, td, th, tw, st);
    return launch_fwd<5, 4>(p, td, th, tw, st);
}

static void wgrad_geom(int N, int D, int H, int W, int k, int cin_p, int cout_p, int* ntiles, int* n_ci_t,
                       int* n_co_t, int* gz, int* gx, int* bt) {
    int td, th, tw;
    pick_tile(W, &td, &th, &tw);
    *ntiles = N * ceil_div(D, td) * ceil_div(H, th) * ceil_div(W, tw);
    *n_ci_t = ceil_div(cin_p, 16);
    *n_co_t = ceil_div(cout_p, 16);
    *gz = (k == 3) ? 1 : 5;
    *bt = (k == 3) ? 27 : 25;
    *gx = wgrad_gx(*ntiles, (*n_ci_t) * (*n_co_t) * (*gz));
}

extern "C" int ctu_channel_sum_num_blocks(int64_t nvox);

extern "C" size_t ctu_conv3d_wgrad_ws_floats(int N, int D, int H, int W, int k, int cin_p, int cout_p) {
    if (k != 3 && k != 5) return 0;
    int ntiles, nci, nco, gz, gx, bt;
    wgrad_geom(N, D, H, W, k, cin_p, cout_p, &ntiles, &nci, &nco, &gz, &gx, &bt);
    size_t slabs = (size_t)gz * nci * nco * gx * bt * 256;
    if (use_k3s(k, W, cin_p, cout_p)) {
        const K3sGeom g = k3s_geom(N, D, H, W, cin_p, cout_p);
        slabs = (size_t)g.pairs * g.gx * g.nmf * 256;
    }
    if (use_k5s(k, W, cin_p, cout_p)) {
        const K3sGeom g = k5s_geom(N, D, H, W, cin_p, cout_p);
        slabs = (size_t)5 * g.pairs * g.gx * g.nmf * 256;
    }
    const size_t bsum = (size_t)ctu_channel_sum_num_blocks((int64_t)N * D * H * W) * cout_p;
    return slabs > bsum ? slabs : bsum;
}

template <int KS, int KDS>
static int launch_wgrad(WgP p, float* dw, int Co, int Ci, const int32_t* cinv, int gz, int gx, hipStream_t st) {
    int td, th, tw;
    pick_tile(p.W, &td, &th, &tw);
    p.tiles_d = ceil_div(p.D, td); p.tiles_h = ceil_div(p.H, th); p.tiles_w = ceil_div(p.W, tw);
    dim3 grid(gx, p.n_ci_t * p.n_co_t, gz);
    if (tw == 16) conv3d_wgrad_kernel<KS, KDS, 4, 4, 16><<<grid, 256, 0, st>>>(p);
    else if (tw == 8) conv3d_wgrad_kernel<KS, KDS, 4, 8, 8><<<grid, 256, 0, st>>>(p);
    else conv3d_wgrad_kernel<KS, KDS, 4, 4, 4><<<grid, 256, 0, st>>>(p);
    CTU_CHECK_LAUNCH("conv3d_wgrad");
    constexpr int BT = KDS * KS * KS;
    const int n_pairs = p.n_ci_t * p.n_co_t;
    conv3d_wgrad_reduce_kernel<KS, KDS><<<dim3(ceil_div(BT * 256, 64), gz * n_pairs), 64 * RPARTS, 0, st>>>(
        p.ws, dw, Co, Ci, cinv, p.cin_p, p.n_ci_t, gx);
    CTU_CHECK_LAUNCH("conv3d_wgrad_reduce");
    return CTU_OK;
}

struct LazyBn { const float* y; const float* scale; const float* shift; const float* coef; float* out; int cp; };

static int conv3d_wgrad_impl(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                             int in_relu, const float* gout, int g_cs, int cout_p, float* dw, float* dbias, int Co,
                             int Ci, const int32_t* cinv, float* ws, int N, int D, int H, int W, int k, const LazyBn* lz,
                             void* stream);

extern "C" int ctu_conv3d_wgrad(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                int in_relu, const float* gout, int g_cs, int cout_p, float* dw, float* dbias, int Co,
                                int Ci, const int32_t* cinv, float* ws, int N, int D, int H, int W, int k,
                                void* stream) {
    return conv3d_wgrad_impl(in, in_cs, cin_p, in_scale, in_shift, in_relu, gout, g_cs, cout_p, dw, dbias, Co, Ci, cinv, ws,
                             N, D, H, W, k, nullptr, stream);
}

extern "C" int ctu_conv3d_wgrad_bn_supported(int N, int D, int H, int W, int k, int cin_p, int cout_p) {
    return (N > 0 && k3s_lazy_ok(D, H, W, k, cin_p, cout_p)) ? 1 : 0;
}

extern "C" int ctu_conv3d_wgrad_bn(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                   int in_relu, const float* ga, int g_cs, int cout_p, const float* y,
                                   const float* bn_scale, const float* bn_shift, const float* coef, float* gy_out,
                                   float* dw, int Co, int Ci, const int32_t* cinv, float* ws, int N, int D, int H, int W,
                                   int k, void* stream) {
    CTU_REQUIRE(y && bn_scale && bn_shift && coef && gy_out, "conv3d_wgrad_bn: null pointer");
    CTU_REQUIRE(k3s_lazy_ok(D, H, W, k, cin_p, cout_p), "conv3d_wgrad_bn: geometry not supported (ask ctu_conv3d_wgrad_bn_supported)");
    CTU_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)gy_out & 15) == 0 && gy_out != ga, "conv3d_wgrad_bn: y / gy_out alignment, gy_out must not alias ga");
    const LazyBn lz = {y, bn_scale, bn_shift, coef, gy_out, cout_p};
    return conv3d_wgrad_impl(in, in_cs, cin_p, in_scale, in_shift, in_relu, ga, g_cs, cout_p, dw, nullptr, Co, Ci, cinv, ws,
                             N, D, H, W, k, &lz, stream);
}

static int conv3d_wgrad_impl(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                             int in_relu, const float* gout, int g_cs, int cout_p, float* dw, float* dbias, int Co,
                             int Ci, const int32_t* cinv, float* ws, int N, int D, int H, int W, int k, const LazyBn* lz,
                             void* stream) {
    CTU_REQUIRE(k == 3 || k == 5, "conv3d_wgrad: k=%d unsupported (3 or 5)", k);
    CTU_REQUIRE(in && gout && dw && ws, "conv3d_wgrad: null pointer");
    CTU_REQUIRE(cin_p % 8 == 0 && cout_p % 8 == 0 && cin_p > 0 && cout_p > 0, "conv3d_wgrad: padded channels");
    CTU_REQUIRE(in_cs >= cin_p && in_cs % 4 == 0 && g_cs >= cout_p && g_cs % 4 == 0, "conv3d_wgrad: bad stride");
    CTU_REQUIRE(Co <= cout_p && Co > 0 && Ci > 0 && cin_p <= 256, "conv3d_wgrad: Co/Ci");
    CTU_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)gout & 15) == 0, "conv3d_wgrad: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    WgP p;
    p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.g = gout; p.ws = ws;
    p.in_cs = in_cs; p.cin_p = cin_p; p.in_relu = in_relu; p.g_cs = g_cs; p.cout_p = cout_p;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.lz_y = nullptr; p.lz_out = nullptr; p.lz_scale = nullptr; p.lz_shift = nullptr; p.lz_coef = nullptr; p.lz_cp = 0;
    if (lz) { p.lz_y = lz->y; p.lz_out = lz->out; p.lz_scale = lz->scale; p.lz_shift = lz->shift; p.lz_coef = lz->coef; p.lz_cp = lz->cp; }
    int gz, gx, bt;
    wgrad_geom(N, D, H, W, k, cin_p, cout_p, &p.ntiles, &p.n_ci_t, &p.n_co_t, &gz, &gx, &bt);
    if (use_k3s(k, W, cin_p, cout_p)) {
        // persistent, software-pipelined kernel; 8-channel sides use (shift, channel) MFMA tiles.
        // Workspace need is at most the generic bound.
        int rc2;
        if (cin_p == 8 && cout_p == 8) rc2 = launch_wgrad_k3s<2, 2>(p, dw, Co, Ci, cinv, st);
        else if (cout_p == 8) rc2 = launch_wgrad_k3s<1, 2>(p, dw, Co, Ci, cinv, st);
        else if (cin_p == 8) rc2 = launch_wgrad_k3s<2, 1>(p, dw, Co, Ci, cinv, st);
        else rc2 = launch_wgrad_k3s<1, 1>(p, dw, Co, Ci, cinv, st);
        if (rc2 != CTU_OK) return rc2;
        if (dbias) return ctu_channel_sum(gout, g_cs, cout_p, (int64_t)N * D * H * W, ws, dbias, Co, stream);
        return CTU_OK;
    }
    if (use_k5s(k, W, cin_p, cout_p)) {
        CTU_REQUIRE((int64_t)((4 + 2 * 2) * H + 8) * W * in_cs * 4 < (int64_t)1 << 31, "conv3d_wgrad: volume too large for 32-bit offsets");
        int rc2;
        if (cin_p == 8 && cout_p == 8) rc2 = launch_wgrad_k5s<2, 2>(p, dw, Co, Ci, cinv, st);
        else if (cout_p == 8) rc2 = launch_wgrad_k5s<1, 2>(p, dw, Co, Ci, cinv, st);
        else rc2 = launch_wgrad_k5s<2, 1>(p, dw, Co, Ci, cinv, st);
        if (rc2 != CTU_OK) return rc2;
        if (dbias) return ctu_channel_sum(gout, g_cs, cout_p, (int64_t)N * D * H * W, ws, dbias, Co, stream);
        return CTU_OK;
    }
    int rc = (k == 3) ? launch_wgrad<3, 3>(p, dw, Co, Ci, cinv, gz, gx, st)
                      : launch_wgrad<5, 1>(p, dw, Co, Ci, cinv, gz, gx, st);
    if (rc != CTU_OK) return rc;
    if (dbias) return ctu_channel_sum(gout, g_cs, cout_p, (int64_t)N * D * H * W, ws, dbias, Co, stream);
    return CTU_OK;
}

// ---- fused decoder up-convolution: weight gradient w.r.t. the composite weights (see upconv_fused.hip)
extern "C" size_t ctu_upconv_fused_wgrad_ws_floats(int N, int D, int H, int W, int cin_p, int nout_p) {
    const UpWgGeom a = upwg_geom(N, D, H, W, cin_p, nout_p, false), b = upwg_geom(N, D, H, W, cin_p, nout_p, true);
    const size_t na = (size_t)a.pairs * a.gx * a.nmf * 256, nb = (size_t)b.pairs * b.gx * b.nmf * 256;
    return na > nb ? na : nb;                            // either tiling (the w-parity tile needs nout_p = 8 = g_cs)
}

static int upconv_fused_wgrad_impl(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                   int in_relu, const float* gout, int g_cs, int nout_p, float* dweff, float* ws,
                                   int N, int D, int H, int W, const LazyBn* lz, void* stream);

extern "C" int ctu_upconv_fused_wgrad(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                      int in_relu, const float* gout, int g_cs, int nout_p, float* dweff, float* ws,
                                      int N, int D, int H, int W, void* stream) {
    return upconv_fused_wgrad_impl(in, in_cs, cin_p, in_scale, in_shift, in_relu, gout, g_cs, nout_p, dweff, ws, N, D, H, W,
                                   nullptr, stream);
}

// full coarse boxes (4 x 4 x 8) and full 16-channel input tiles
extern "C" int ctu_upconv_fused_wgrad_bn_supported(int N, int D, int H, int W, int cin_p, int nout_p) {
    return (N > 0 && D % 4 == 0 && H % 4 == 0 && W % 8 == 0 && cin_p % 16 == 0 && nout_p % 8 == 0) ? 1 : 0;
}

extern "C" int ctu_upconv_fused_wgrad_bn(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                         int in_relu, const float* ga, int g_cs, int nout_p, const float* y,
                                         const float* bn_scale, const float* bn_shift, const float* coef, float* gy_out,
                                         float* dweff, float* ws, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(y && bn_scale && bn_shift && coef && gy_out, "upconv_fused_wgrad_bn: null pointer");
    CTU_REQUIRE(ctu_upconv_fused_wgrad_bn_supported(N, D, H, W, cin_p, nout_p), "upconv_fused_wgrad_bn: geometry not supported");
    CTU_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)gy_out & 15) == 0 && gy_out != ga, "upconv_fused_wgrad_bn: y / gy_out alignment, gy_out must not alias ga");
    const LazyBn lz = {y, bn_scale, bn_shift, coef, gy_out, nout_p};
    return upconv_fused_wgrad_impl(in, in_cs, cin_p, in_scale, in_shift, in_relu, ga, g_cs, nout_p, dweff, ws, N, D, H, W, &lz, stream);
}

static int upconv_fused_wgrad_impl(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                   int in_relu, const float* gout, int g_cs, int nout_p, float* dweff, float* ws,
                                   int N, int D, int H, int W, const LazyBn* lz, void* stream) {
    CTU_REQUIRE(in && gout && dweff && ws, "upconv_fused_wgrad: null pointer");
    CTU_REQUIRE(cin_p % 8 == 0 && nout_p % 8 == 0 && cin_p > 0 && nout_p > 0 && nout_p <= 64, "upconv_fused_wgrad: padded channels");
    CTU_REQUIRE(in_cs >= cin_p && in_cs % 4 == 0 && g_cs >= nout_p && g_cs % 4 == 0, "upconv_fused_wgrad: bad stride");
    CTU_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)gout & 15) == 0, "upconv_fused_wgrad: 16-byte alignment");
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "upconv_fused_wgrad: scale/shift must come together");
    CTU_REQUIRE((int64_t)(2 * H * W + 2 * W + 2) * in_cs * 4 < (int64_t)1 << 31 && (int64_t)(8 * 2 * H + 8) * 2 * W * g_cs * 4 < (int64_t)1 << 31,
                "upconv_fused_wgrad: volume too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    const bool pw = nout_p == 8 && g_cs == 8;             // (w-parity, c_out) tile: the two fine voxels are 16 contiguous floats
    const UpWgGeom g = upwg_geom(N, D, H, W, cin_p, nout_p, pw);
    WgP p;
    p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.g = gout; p.ws = ws;
    p.in_cs = in_cs; p.cin_p = cin_p; p.in_relu = in_relu; p.g_cs = g_cs; p.cout_p = nout_p;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, 4); p.tiles_w = ceil_div(W, 8);
    p.ntiles = g.ntiles; p.n_ci_t = g.n_ci_g; p.n_co_t = g.n_co_g;
    p.lz_y = nullptr; p.lz_out = nullptr; p.lz_scale = nullptr; p.lz_shift = nullptr; p.lz_coef = nullptr; p.lz_cp = 0;
    if (lz) { p.lz_y = lz->y; p.lz_out = lz->out; p.lz_scale = lz->scale; p.lz_shift = lz->shift; p.lz_coef = lz->coef; p.lz_cp = lz->cp; }
    if (pw) {
        p.cout_p = 16;                                   // all 4 channel quads of the (w-parity, c_out) tile are real
        if (lz) conv3d_wgrad_k3s_kernel<1, 1, 2, true><<<dim3(g.gx, g.pairs), 256, 0, st>>>(p, g.tpb);
        else conv3d_wgrad_k3s_kernel<1, 1, 2><<<dim3(g.gx, g.pairs), 256, 0, st>>>(p, g.tpb);
        CTU_CHECK_LAUNCH("upconv_fused_wgrad(pw)");
        upconv_wgrad_reduce_pw_kernel<<<dim3(12 * 256 / 64, g.pairs), 64 * RPARTS, 0, st>>>(ws, dweff, cin_p, g.n_ci_g, g.gx);
        CTU_CHECK_LAUNCH("upconv_fused_wgrad_reduce(pw)");
        return CTU_OK;
    }
    if (lz) conv3d_wgrad_k3s_kernel<1, 1, 1, true><<<dim3(g.gx, g.pairs), 256, 0, st>>>(p, g.tpb);
    else conv3d_wgrad_k3s_kernel<1, 1, 1><<<dim3(g.gx, g.pairs), 256, 0, st>>>(p, g.tpb);
    CTU_CHECK_LAUNCH("upconv_fused_wgrad");
    upconv_wgrad_reduce_kernel<<<dim3(8 * 256 / 64, g.pairs), 64 * RPARTS, 0, st>>>(ws, dweff, cin_p, nout_p, g.n_ci_g, g.n_co_g, g.gx);
    CTU_CHECK_LAUNCH("upconv_fused_wgrad_reduce");
    return CTU_OK;
}
