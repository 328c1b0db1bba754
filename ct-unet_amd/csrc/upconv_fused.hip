// Fused ConvTranspose3d(C, C, k=2, s=2, bias) -> Conv3d(C, C_out, k=3, p=1, no bias) forward for gfx950: the first two
// layers of every decoder block (ctunet/pytorch/models.py:37-38) as ONE kernel on the coarse grid.
//
// Algebra (checked to 1e-13 in fp64 in scripts/experiments/fused_upblock_proto.py): for an output parity
// p in {0,1}^3 the pair is a 2x2x2 convolution of the COARSE input,
//     y[2i+p] = sum_{d in D(p)} x[i+d] . W_eff[p][d]  +  b_eff[border class of 2i+p],
//     W_eff[p][d][ci,co] = sum_{(t,a) in S(p,d)} sum_cm WT[ci,cm,a] W3[co,cm,t]
// (per axis: p=0: d=-1 <- (t=-1,a=1), d=0 <- (t=0,a=0),(t=1,a=1);  p=1: d=0 <- (t=-1,a=0),(t=0,a=1), d=+1 <- (t=1,a=0)),
// so a parity's taps are the 2x2x2 sub-cube of the 3x3x3 coarse halo that starts at p.  8 taps instead of 27 plus the
// transposed conv (4096 instead of 15872 FLOP per fine voxel for 32 -> 32 -> 8), and the fine-grid intermediate is never
// written.  The conv zero-pads the TRANSPOSED-CONV OUTPUT, so at the volume faces only the bias term changes:
// b_eff[class][co] = sum_{t inside} sum_cm bT[cm] W3[co,cm,t], 27 classes (per axis: interior / first / last fine voxel).
//
// Kernel: the persistent implicit GEMM of conv3d.hip (rows = output channels, columns = 16 coarse voxels along w, K =
// 8-channel chunks x taps; 4x4x16 coarse boxes, haloed box of one chunk in LDS at 12 floats per voxel, lazy-BatchNorm
// transform applied while staging, next (box, chunk) prefetched into registers under the MFMAs).  One staged box feeds
// all the block's parities: accumulators acc[parity][row][tile], PB * NTP = 8 tile-parities per block.  The epilogue
// scatters to the fine grid (float4 per lane), adds the border-class bias and carries the BatchNorm partial sums.
#include <type_traits>
#include "common.h"
#include "bn_tail.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

struct UpP {
    const float* in;
    const float* in_scale;
    const float* in_shift;
    const float* wp;          // [chunk][parity 8][tap 8][ntp][128]
    const float* beff;        // [27][nout_p]
    float* out;               // fine grid [N, 2D, 2H, 2W, out_cs]
    float* stats;             // [gridDim.y * gridDim.x][2][nout_p]
    int in_cs, rin_p, in_relu, out_cs, nout_p;
    int N, D, H, W;           // COARSE dims
    int tiles_d, tiles_h, tiles_w;
    ctu_bn_tail tail;         // counter != NULL: the last block of the launch finalizes the BatchNorm (bn_tail.h)
};

// PW (8 padded output channels): a 16-wide tile would be half empty, so the w-parity moves into the tile instead --
// columns = (p_w, c_out), 4 (p_d, p_h) parities per block, 2 x 2 x 3 taps (w offsets 0..2; offset 0 only feeds the
// p_w = 0 half, offset 2 only the p_w = 1 half, zeros elsewhere): 48 tile-taps instead of 64.
template <int PB, int NTP, bool PW = false>
__global__ __launch_bounds__(256, 2) void upconv_fused_fwd_kernel(UpP p, int ntiles, int tiles_per_block) {
    static_assert(PW ? (PB == 4 && NTP == 1) : ((PB * NTP == 8 || (PB == 4 && NTP == 1)) && (PB == 8 || PB == 4 || PB == 2)), "tile-parities per block");
    constexpr int NTW = PW ? 3 : 2, TAPS = 4 * NTW;      // taps along w / per parity
    constexpr int MT = 4, TD = 4, TH = 4, TW = 16;
    constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW;
    constexpr int NTHR = 256;
    constexpr int NTPT = NTP;                            // a block covers ALL 16-wide channel tiles of its PB parities
    constexpr int WFL = PB * TAPS * NTP * 128;           // weight floats of one (chunk, block)
    constexpr int AITEMS = HV * 2, AITER = (AITEMS + NTHR - 1) / NTHR;
    constexpr int WITER = (WFL / 4 + NTHR - 1) / NTHR;
    constexpr int VS2 = 6;                               // 12 floats per voxel: conflict-free ds_read_b64
    static_assert(WFL / 4 % NTHR == 0, "weight staging covers the stage exactly");

    __shared__ __attribute__((aligned(16))) v2f sA2[HV * VS2];
    __shared__ __attribute__((aligned(16))) v2f sW2[WFL / 2];
    typedef const volatile __attribute__((address_space(3))) v2f* lds_v2f_ptr;
    __shared__ float sRed[4 * NTP * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int by = blockIdx.y;                           // parity group: parities by*PB .. by*PB + PB-1
    const int nchunk = p.rin_p >> 3;
    const int half = tid & 1;
    const bool has_xf = p.in_scale != nullptr;
    const int pd0 = PW ? 0 : ((by * PB) >> 2) & 1, ph0 = PW ? 0 : ((by * PB) >> 1) & 1;      // run-time high parity bits

    // this wave owns coarse plane td = wave, its 4 M-tiles are the rows th = 0..3; base of row 0, tap (0,0,0) of the
    // block's first parity
    lds_v2f_ptr vA = (lds_v2f_ptr)sA2 + (((wave + pd0) * HH + ph0) * HW + m) * VS2 + kq;
    lds_v2f_ptr vW = (lds_v2f_ptr)sW2 + (kq * 16 + m);

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
    const unsigned safe_off = (unsigned)(((p.H + 1) * p.W + 1) * p.in_cs + half * 4) * 4u;

    f32x4 acc[PB][MT][NTP];
    float s1[NTP][4], s2[NTP][4];
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
#pragma unroll
        for (int pi = 0; pi < PB; ++pi)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[pi][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    }

    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(ntiles, tile + tiles_per_block);
    int c = 0;
    float4 va[AITER];
    // weight staging values: as float4 the array stays a stack object (scratch; the compiler copies the structs with
    // memcpy) -- harmless for the 255-VGPR variants, which have no registers to spare, but for PW it would be promoted
    // to 24 KB of LDS and halve the occupancy, so PW uses the plain vector type (registers)
    typedef typename std::conditional<PW, f32x4, float4>::type wv_t;
    wv_t vw[WITER];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned vmask = 0;

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
    bool a_interior = false;
    auto load_a = [&](Box b, int cc) {
        const int d0 = b.tz * TD, h0 = b.ty * TH, w0 = b.tx * TW;
        if (has_xf) {
            sc = *reinterpret_cast<const float4*>(p.in_scale + cc * 8 + half * 4);
            sh = *reinterpret_cast<const float4*>(p.in_shift + cc * 8 + half * 4);
        }
        const long long org = (((long long)b.n * p.D + d0) * p.H + h0) * p.W + w0 - ((long long)p.H * p.W + p.W + 1);
        const char* base = reinterpret_cast<const char*>(p.in + org * p.in_cs + cc * 8);
        a_interior = d0 >= 1 && h0 >= 1 && w0 >= 1 && d0 + TD < p.D && h0 + TH < p.H && w0 + TW < p.W;
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
        if (d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W) {          // full border box: branch-free
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
        for (int it = 0; it < AITER; ++it) {                               // ragged box: per-item bounds
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
    auto load_w = [&](int cc) {
        const char* wsrc = reinterpret_cast<const char*>(PW ? p.wp + (size_t)cc * 4 * TAPS * 128
                                                            : p.wp + ((size_t)cc * 8 + (size_t)by * PB) * 8 * NTPT * 128);
#pragma unroll
        for (int it = 0; it < WITER; ++it) vw[it] = *reinterpret_cast<const wv_t*>(wsrc + (unsigned)(tid + it * NTHR) * 16u);
    };
    if (tile >= tile_end) return;
    load_a(box, 0);
    load_w(0);

    while (true) {
        CTU_SETPRIO(CTU_PRIO_STAGE);           // staging phases outrank the other block's MFMA stream (common.h)
        __syncthreads();
        {
            auto put = [&](int it, float4 val) {
                const int i = tid + it * NTHR;
                if ((it + 1) * NTHR <= AITEMS || i < AITEMS) *reinterpret_cast<float4*>(&sA2[(i >> 1) * VS2 + half * 2]) = val;
            };
            if (a_interior) {
#pragma unroll
                for (int it = 0; it < AITER; ++it) put(it, has_xf ? xform4(va[it], sc, sh, p.in_relu) : va[it]);
            } else {
#pragma unroll
                for (int it = 0; it < AITER; ++it) {
                    const float4 t = has_xf ? xform4(va[it], sc, sh, p.in_relu) : va[it];
                    const bool ok = (vmask >> it) & 1u;
                    put(it, make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f));
                }
            }
#pragma unroll
            for (int it = 0; it < WITER; ++it) *reinterpret_cast<wv_t*>(&sW2[(tid + it * NTHR) * 2]) = vw[it];
        }
        __syncthreads();
        int ntile = tile, nc = c + 1;
        Box nbox = box;
        if (nc == nchunk) { nc = 0; ntile = tile + 1; nbox = box_next(box); }
        const bool has_next = ntile < tile_end;
        if (has_next) {
            load_a(nbox, nc);
            load_w(nc);
        }
        // ---- MFMAs: flat sequence of PB * 4 groups (parity pi, dz, dx); a group's 5 input rows feed the 2 dy taps of the
        // 4 M-tiles: 5 + 2*NTP fragment reads for 16*NTP MFMAs, read one group ahead
        CTU_SETPRIO(0);
        {
            constexpr int NG = PB * 2 * NTW;
            v2f ar[2][5], br[2][2][NTP];
            auto load_group = [&](int g, v2f (&aa)[5], v2f (&bb)[2][NTP]) {
                const int pi = g / (2 * NTW), dz = (g / NTW) & 1, dx = g % NTW;
                const int pid = PW ? (pi >> 1) : ((PB == 8) ? (pi >> 2) : 0), pih = PW ? (pi & 1) : ((PB >= 4) ? ((pi >> 1) & 1) : 0),
                          piw = PW ? 0 : (pi & 1);
#pragma unroll
                for (int r = 0; r < 5; ++r) aa[r] = vA[(((pid + dz) * HH + r + pih) * HW + piw + dx) * VS2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int nt = 0; nt < NTP; ++nt) bb[dy][nt] = vW[((pi * TAPS + (dz * 2 + dy) * NTW + dx) * NTP + nt) * 64];
            };
            load_group(0, ar[0], br[0]);
            // (the 128-accumulator variants have no registers for a second fragment set: they keep the compiler's order)
            constexpr bool PIN = CTU_PIN_FWD && PB * NTP <= 4;
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);      // the first group's reads stay out of the pinned sequence
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1, ar[(g + 1) & 1], br[(g + 1) & 1]);
                const int pi = g / (2 * NTW);
                v2f (&aa)[5] = ar[g & 1];
                v2f (&bb)[2][NTP] = br[g & 1];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
                    for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[pi][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[dy][nt].x, aa[mt + dy].x, acc[pi][mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[pi][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[dy][nt].y, aa[mt + dy].y, acc[pi][mt][nt], 0, 0, 0);
                }
                // pin the order: the next group's fragment reads go out one per MFMA under this group's MFMAs (left alone the
                // scheduler sinks each read to just above its first use and drains lgkmcnt(0) there)
                if constexpr (PIN) {
#pragma unroll
                    for (int q = 0; q < 4 * MT * NTP; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (g + 1 < NG && q < 5 + 2 * NTP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            }
        }
        CTU_SETPRIO(CTU_PRIO_STAGE);
        if (c == nchunk - 1) {
            // ---- epilogue of this coarse box: scatter its 8 * 256 fine voxels, border-class bias, BN partial sums
            const int d0 = box.tz * TD, h0 = box.ty * TH, w0 = box.tx * TW;
            const int Df = 2 * p.D, Hf = 2 * p.H, Wf = 2 * p.W;
            const int cd = d0 + wave, cw = w0 + m;
            const bool touches = d0 == 0 || h0 == 0 || w0 == 0 || d0 + TD >= p.D || h0 + TH >= p.H || w0 + TW >= p.W;   // uniform
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt) {
                const int co = PW ? (kq & 1) * 4 : nt * 16 + kq * 4;
                const bool cok = co < p.nout_p;
                float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cok) b0 = *reinterpret_cast<const float4*>(p.beff + co);             // class 0: interior
#pragma unroll
                for (int pi = 0; pi < PB; ++pi) {
                    const int pd = PW ? (pi >> 1) : pd0 + ((PB == 8) ? (pi >> 2) : 0), ph = PW ? (pi & 1) : ph0 + ((PB >= 4) ? ((pi >> 1) & 1) : 0),
                              pw = PW ? (kq >> 1) : (pi & 1);
                    const int fz = 2 * cd + pd, fx = 2 * cw + pw;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int ch = h0 + mt, fy = 2 * ch + ph;
                        if (cok && cd < p.D && ch < p.H && cw < p.W) {
                            float4 bv = b0;
                            if (touches) {
                                const int cz = fz == 0 ? 1 : (fz == Df - 1 ? 2 : 0), cy = fy == 0 ? 1 : (fy == Hf - 1 ? 2 : 0),
                                          cx = fx == 0 ? 1 : (fx == Wf - 1 ? 2 : 0);
                                bv = *reinterpret_cast<const float4*>(p.beff + (size_t)((cz * 3 + cy) * 3 + cx) * p.nout_p + co);
                            }
                            float4 o;
                            o.x = acc[pi][mt][nt][0] + bv.x; o.y = acc[pi][mt][nt][1] + bv.y;
                            o.z = acc[pi][mt][nt][2] + bv.z; o.w = acc[pi][mt][nt][3] + bv.w;
                            const size_t fv = (((size_t)box.n * Df + fz) * Hf + fy) * Wf + fx;
                            *reinterpret_cast<float4*>(p.out + fv * p.out_cs + co) = o;
                            s1[nt][0] += o.x; s1[nt][1] += o.y; s1[nt][2] += o.z; s1[nt][3] += o.w;
                            s2[nt][0] += o.x * o.x; s2[nt][1] += o.y * o.y; s2[nt][2] += o.z * o.z; s2[nt][3] += o.w * o.w;
                        }
                        acc[pi][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
        if (!has_next) break;
        tile = ntile; c = nc; box = nbox;
    }
    // ---- one BatchNorm partial row per block: reduce over the 16 voxel lanes and the 4 waves
    if (p.stats) {
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a1 = s1[nt][r], a2 = s2[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
                if (PW) { a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32); }      // the two w-parity halves hold the same channels
                const int ch = PW ? (kq & 1) * 4 + r : nt * 16 + kq * 4 + r;
                if (m == 0 && (!PW || kq < 2)) {
                    sRed[(wave * NTP * 16 + ch) * 2 + 0] = a1;
                    sRed[(wave * NTP * 16 + ch) * 2 + 1] = a2;
                }
            }
        __syncthreads();
        float* row = p.stats + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * p.nout_p;
        for (int ch = tid; ch < p.nout_p; ch += NTHR) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a1 += sRed[(w * NTP * 16 + ch) * 2 + 0];
                a2 += sRed[(w * NTP * 16 + ch) * 2 + 1];
            }
            st_row(p.tail.counter != nullptr, row + ch, a1);
            st_row(p.tail.counter != nullptr, row + p.nout_p + ch, a2);
        }
        if (p.tail.counter) bn_fwd_tail(p.tail, p.stats, gridDim.x * gridDim.y, p.nout_p, gridDim.x * gridDim.y);
    }
}

// ------------------------------------------------------------------ fused data gradient
// dX[i][ci] = sum_p sum_{d in D(p)} sum_co dy[2(i-d)+p][co] W_eff[p][d][ci][co]: the adjoint of the parity convolutions,
// again on the COARSE grid.  The fine-grid gradient is read as a strided space-to-depth view: a "chunk" is 8 channels
// of ONE output parity (fine voxel 2j + p of coarse voxel j), so the 6x6x18 coarse halo box of a chunk is staged exactly
// like an input chunk of the forward kernel, only with doubled voxel strides and a parity offset.  A chunk of parity p
// contributes through the 2x2x2 sub-cube of the halo that starts at 1 - p (coarse input j = i - d), with the taps of
// W_eff mirrored (tap bit = 1 - sub-cube bit).  rows = input channels ci (NT 16-wide tiles per block), K = chunks x 8 taps.
struct UpDP {
    const float* g;           // fine grid [N, 2D, 2H, 2W, g_cs], raw-output gradient of the fused op
    const float* wp;          // [chunk = parity*QC + q][tap 8][n16][128]
    float* out;               // coarse [N, D, H, W, out_cs]
    int g_cs, nout_p, out_cs, cin_p;
    int N, D, H, W;           // COARSE dims
    int tiles_d, tiles_h, tiles_w, n16;
};

template <int NT>
__global__ __launch_bounds__(256, 2) void upconv_fused_bwd_data_kernel(UpDP p, int ntiles, int tiles_per_block) {
    constexpr int MT = 4, TD = 4, TH = 4, TW = 16;
    constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW;
    constexpr int NTHR = 256;
    constexpr int WFL = 8 * NT * 128;
    constexpr int AITEMS = HV * 2, AITER = (AITEMS + NTHR - 1) / NTHR;
    constexpr int WITER = (WFL / 4 + NTHR - 1) / NTHR;
    constexpr int VS2 = 6;
    static_assert(WFL / 4 % NTHR == 0, "weight staging covers the stage exactly");

    __shared__ __attribute__((aligned(16))) v2f sA2[HV * VS2];
    __shared__ __attribute__((aligned(16))) v2f sW2[WFL / 2];
    typedef const volatile __attribute__((address_space(3))) v2f* lds_v2f_ptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int by = blockIdx.y;
    const int qc = p.nout_p >> 3;                        // 8-channel chunks per parity
    const int nchunk = 8 * qc;
    const int half = tid & 1;
    const int Hf = 2 * p.H, Wf = 2 * p.W;

    lds_v2f_ptr vA0 = (lds_v2f_ptr)sA2 + ((wave * HH) * HW + m) * VS2 + kq;
    lds_v2f_ptr vW = (lds_v2f_ptr)sW2 + (kq * 16 + m);

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
        hoff[it] = (unsigned)(((2 * pd * Hf + 2 * ph) * Wf + 2 * pw) * p.g_cs + half * 4) * 4u;      // fine strides
        const unsigned face = (pd == 0 ? 1u : 0u) | (pd == HD - 1 ? 2u : 0u) | (ph == 0 ? 4u : 0u) | (ph == HH - 1 ? 8u : 0u) |
                              (pw == 0 ? 16u : 0u) | (pw == HW - 1 ? 32u : 0u);
        fw[it / FPW] |= face << (6 * (it % FPW));
    }
    const unsigned safe_off = (unsigned)(((2 * Hf + 2) * Wf + 2) * p.g_cs + half * 4) * 4u;          // the box origin

    f32x4 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int tile = blockIdx.x * tiles_per_block;
    const int tile_end = min(ntiles, tile + tiles_per_block);
    if (tile >= tile_end) return;
    int c = 0;
    float4 va[AITER], vw[WITER];
    unsigned vmask = 0;

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
    bool a_interior = false;
    auto load_a = [&](Box b, int cc) {
        const int d0 = b.tz * TD, h0 = b.ty * TH, w0 = b.tx * TW;
        const int par = cc / qc, q = cc % qc;
        const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1;
        // fine voxel of the coarse halo origin (d0-1, h0-1, w0-1) at parity p (may lie outside: only valid items are read)
        const long long org = (((long long)b.n * 2 * p.D + 2 * (d0 - 1) + pz) * Hf + 2 * (h0 - 1) + py) * Wf + 2 * (w0 - 1) + px;
        const char* base = reinterpret_cast<const char*>(p.g + org * p.g_cs + q * 8);
        a_interior = d0 >= 1 && h0 >= 1 && w0 >= 1 && d0 + TD < p.D && h0 + TH < p.H && w0 + TW < p.W;
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
        if (d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W) {
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
        for (int it = 0; it < AITER; ++it) {
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
    const bool w_full = (by + 1) * NT <= p.n16;
    auto load_w = [&](int cc) {
        const float* wsrc = p.wp + ((size_t)cc * 8 * p.n16 + (size_t)by * NT) * 128;
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int i = (tid + it * NTHR) * 4;
            const int ts = i / (NT * 128), r = i % (NT * 128);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w_full || by * NT * 128 + r < p.n16 * 128)
                v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(wsrc) + (unsigned)(ts * p.n16 * 128 + r) * 4u);
            vw[it] = v;
        }
    };
    load_a(box, 0);
    load_w(0);

    while (true) {
        CTU_SETPRIO(CTU_PRIO_STAGE);           // staging phases outrank the other block's MFMA stream (common.h)
        __syncthreads();
        {
            auto put = [&](int it, float4 val) {
                const int i = tid + it * NTHR;
                if ((it + 1) * NTHR <= AITEMS || i < AITEMS) *reinterpret_cast<float4*>(&sA2[(i >> 1) * VS2 + half * 2]) = val;
            };
            if (a_interior) {
#pragma unroll
                for (int it = 0; it < AITER; ++it) put(it, va[it]);
            } else {
#pragma unroll
                for (int it = 0; it < AITER; ++it) {
                    const bool ok = (vmask >> it) & 1u;
                    put(it, make_float4(ok ? va[it].x : 0.f, ok ? va[it].y : 0.f, ok ? va[it].z : 0.f, ok ? va[it].w : 0.f));
                }
            }
#pragma unroll
            for (int it = 0; it < WITER; ++it) *reinterpret_cast<float4*>(&sW2[(tid + it * NTHR) * 2]) = vw[it];
        }
        __syncthreads();
        // this chunk's parity -> origin of its 2x2x2 sub-cube in the halo: 1 - p per axis
        const int par = c / qc;
        lds_v2f_ptr vA = vA0 + (((1 - ((par >> 2) & 1)) * HH + (1 - ((par >> 1) & 1))) * HW + (1 - (par & 1))) * VS2;
        int ntile = tile, nc = c + 1;
        Box nbox = box;
        if (nc == nchunk) { nc = 0; ntile = tile + 1; nbox = box_next(box); }
        const bool has_next = ntile < tile_end;
        if (has_next) {
            load_a(nbox, nc);
            load_w(nc);
        }
        CTU_SETPRIO(0);
        {
            v2f ar[2][5], br[2][2][NT];
            auto load_group = [&](int g, v2f (&aa)[5], v2f (&bb)[2][NT]) {
                const int ez = (g >> 1) & 1, ex = g & 1;
#pragma unroll
                for (int r = 0; r < 5; ++r) aa[r] = vA[((ez * HH + r) * HW + ex) * VS2];
#pragma unroll
                for (int ey = 0; ey < 2; ++ey)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bb[ey][nt] = vW[(((ez * 2 + ey) * 2 + ex) * NT + nt) * 64];
            };
            load_group(0, ar[0], br[0]);
#if CTU_PIN_FWD
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) load_group(g + 1, ar[(g + 1) & 1], br[(g + 1) & 1]);
                v2f (&aa)[5] = ar[g & 1];
                v2f (&bb)[2][NT] = br[g & 1];
#pragma unroll
                for (int ey = 0; ey < 2; ++ey) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[ey][nt].x, aa[mt + ey].x, acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[ey][nt].y, aa[mt + ey].y, acc[mt][nt], 0, 0, 0);
                }
#if CTU_PIN_FWD
#pragma unroll
                for (int q = 0; q < 4 * MT * NT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (g + 1 < 4 && q < 5 + 2 * NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#endif
            }
        }
        CTU_SETPRIO(CTU_PRIO_STAGE);
        if (c == nchunk - 1) {
            const int d0 = box.tz * TD, h0 = box.ty * TH, w0 = box.tx * TW;
            const int gw = w0 + m;
            float* obase = p.out + ((((size_t)box.n * p.D + d0 + wave) * p.H + h0) * p.W + gw) * p.out_cs;
            const int orow = p.W * p.out_cs;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = (by * NT + nt) * 16 + kq * 4;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (co < p.cin_p && d0 + wave < p.D && h0 + mt < p.H && gw < p.W) {
                        float4 o;
                        o.x = acc[mt][nt][0]; o.y = acc[mt][nt][1]; o.z = acc[mt][nt][2]; o.w = acc[mt][nt][3];
                        *reinterpret_cast<float4*>(obase + mt * orow + co) = o;
                    }
                    acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (!has_next) break;
        tile = ntile; c = nc; box = nbox;
    }
}

// data-gradient weights gathered from the forward packing: wpd[(p*qc + q)][e][n16][kq][n][j] = W_eff[p][tap = e ^ 7][pos][co],
// pos = n16*16 + n (padded input-channel position), co = q*8 + kq*2 + j
__global__ void upconv_pack_bwd_kernel(const float* __restrict__ wp, float* __restrict__ wpd, int cin_p, int nout_p) {
    const int n16 = (cin_p + 15) >> 4, qc = nout_p >> 3, ntpt = (nout_p + 15) >> 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 8 * qc * 8 * n16 * 128) return;
    int r = idx;
    const int j = r & 1; r >>= 1;
    const int n = r & 15; r >>= 4;
    const int kq = r & 3; r >>= 2;
    const int nt = r % n16; r /= n16;
    const int e = r & 7; r >>= 3;
    const int q = r % qc;
    const int par = r / qc;
    const int pos = nt * 16 + n, co = q * 8 + kq * 2 + j;
    float v = 0.f;
    if (pos < cin_p) {
        const int cf = pos >> 3, kqf = (pos & 7) >> 1, jf = pos & 1;
        v = wp[((((size_t)cf * 8 + par) * 8 + (e ^ 7)) * ntpt + (co >> 4)) * 128 + kqf * 32 + (co & 15) * 2 + jf];
    }
    wpd[idx] = v;
}

// ---- composite weights in fragment order: wp[chunk][parity][tap8 = (dz*2+dy)*2+dx][ntp][kq][n][j]
// per axis: the (t, a) pairs behind (parity bit, tap bit): t = conv tap index 0..2, a = transposed-conv tap 0..1
__device__ __forceinline__ int axis_pairs(int pbit, int jbit, int (&t)[2], int (&a)[2]) {
    if (pbit == 0) {
        if (jbit == 0) { t[0] = 0; a[0] = 1; return 1; }
        t[0] = 1; a[0] = 0; t[1] = 2; a[1] = 1; return 2;
    }
    if (jbit == 0) { t[0] = 0; a[0] = 0; t[1] = 1; a[1] = 1; return 2; }
    t[0] = 2; a[0] = 0; return 1;
}

// Packing runs once per optimizer step.  Step 1: transposes so that step 2 reads coalesced --
//   w3t[t 27][cm C][co nout_p] (zero beyond Co),  wtt[a 8][ci C][cm C].
__global__ void upconv_transpose_kernel(const float* __restrict__ wt, const float* __restrict__ w3, float* __restrict__ wtt,
                                        float* __restrict__ w3t, int C, int Co, int nout_p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n3 = 27 * C * nout_p, nt = 8 * C * C;
    if (idx < n3) {
        const int co = idx % nout_p, cm = (idx / nout_p) % C, t = idx / (nout_p * C);
        w3t[idx] = co < Co ? w3[((size_t)co * C + cm) * 27 + t] : 0.f;
    } else if (idx < n3 + nt) {
        const int k = idx - n3;
        const int cm = k % C, ci = (k / C) % C, a = k / (C * C);
        wtt[k] = wt[((size_t)ci * C + cm) * 8 + a];
    }
}

// Step 2: 8 lanes per (chunk, parity, tap, padded input channel, QUAD of output channels): a W_eff entry = sum over its
// <= 8 (conv tap, transposed-conv tap) pairs of a C-long dot product; each lane takes every 8th cm (one W3 float4 and one
// WT scalar per 4 multiply-adds: the kernel is bound by load instructions, not flops), a 3-step shuffle reduction combines
// the lanes and lane 0 writes the four values straight to their fragment positions
__global__ void upconv_pack_kernel(const float* __restrict__ wtt, const float* __restrict__ w3t, float* __restrict__ wp,
                                   int C, int nout_p, const int32_t* __restrict__ cinv, int nchunk, int ntpt) {
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = gidx & 7, idx = gidx >> 3;
    const int ncol4 = ntpt * 4;                          // output-channel quads per row of N-tiles
    if (idx >= nchunk * 8 * 8 * 8 * ncol4) return;      // (whole 8-lane groups leave together)
    int r = idx;
    const int co = (r % ncol4) * 4; r /= ncol4;
    const int r8 = r & 7; r >>= 3;                       // channel inside the chunk (= kq*2 + j)
    const int tap = r & 7; r >>= 3;
    const int par = r & 7; r >>= 3;
    const int c = r;
    const int rp = c * 8 + r8;
    const int ci = cinv ? cinv[rp] : (rp < C ? rp : -1);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ci >= 0 && co < nout_p) {                        // (nout_p is a multiple of 8: a quad is inside or outside as a whole)
        int tz[2], az[2], ty[2], ay[2], tx[2], ax[2];
        const int nz = axis_pairs((par >> 2) & 1, (tap >> 2) & 1, tz, az);
        const int ny = axis_pairs((par >> 1) & 1, (tap >> 1) & 1, ty, ay);
        const int nx = axis_pairs(par & 1, tap & 1, tx, ax);
        for (int iz = 0; iz < nz; ++iz)
            for (int iy = 0; iy < ny; ++iy)
                for (int ix = 0; ix < nx; ++ix) {
                    const float* a = wtt + ((size_t)((az[iz] * 2 + ay[iy]) * 2 + ax[ix]) * C + ci) * C;
                    const float* b = w3t + (size_t)((tz[iz] * 3 + ty[iy]) * 3 + tx[ix]) * C * nout_p + co;
                    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
                    for (int cm = sub; cm < C; cm += 8) {
                        const float av = a[cm];
                        const float4 bv = *reinterpret_cast<const float4*>(b + (size_t)cm * nout_p);
                        s.x = fmaf(av, bv.x, s.x); s.y = fmaf(av, bv.y, s.y); s.z = fmaf(av, bv.z, s.z); s.w = fmaf(av, bv.w, s.w);
                    }
                    v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
                }
    }
    float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { vv[q] += __shfl_xor(vv[q], 1); vv[q] += __shfl_xor(vv[q], 2); vv[q] += __shfl_xor(vv[q], 4); }
    if (sub == 0) {
        const int nt = co >> 4, n = co & 15, kq = r8 >> 1, j = r8 & 1;
        float* dst = wp + ((((size_t)c * 8 + par) * 8 + tap) * ntpt + nt) * 128 + kq * 32 + n * 2 + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * 2] = vv[q];
    }
}

// PW forward packing (8 padded output channels) gathered from the standard one:
// wpw[chunk][par4 = pd*2+ph][tap12 = (dz*2+dy)*3+dxx][kq][n = px*8+co][j] = W_eff[(pd,ph,px)][(dz,dy,dxx-px)] or 0
__global__ void upconv_pack_pw_kernel(const float* __restrict__ wp, float* __restrict__ wpw, int nchunk) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nchunk * 4 * 12 * 128) return;
    int r = idx;
    const int j = r & 1; r >>= 1;
    const int n = r & 15; r >>= 4;
    const int kq = r & 3; r >>= 2;
    const int tap = r % 12; r /= 12;
    const int par4 = r & 3;
    const int c = r >> 2;
    const int px = n >> 3, co = n & 7, dxx = tap % 3, dzy = tap / 3, jx = dxx - px;
    float v = 0.f;
    if (jx == 0 || jx == 1) v = wp[((size_t)(c * 8 + par4 * 2 + px) * 8 + dzy * 2 + jx) * 128 + kq * 32 + co * 2 + j];
    wpw[idx] = v;
}

// b_eff[class 27][nout_p]: class = (cz*3+cy)*3+cx, per axis 0 interior, 1 first fine voxel (t=0 outside), 2 last (t=2 outside).
// One block: S[t][co] = sum_cm bT[cm] w3t[t][cm][co] into LDS (coalesced over co), then the 27 class sums of valid taps.
__global__ __launch_bounds__(1024) void upconv_beff_kernel(const float* __restrict__ bt, const float* __restrict__ w3t,
                                                           float* __restrict__ beff, int C, int nout_p) {
    __shared__ float S[27 * 64];
    for (int i = threadIdx.x; i < 27 * nout_p; i += blockDim.x) {
        const int co = i % nout_p, t = i / nout_p;
        const float* b = w3t + (size_t)t * C * nout_p + co;
        float s = 0.f;
#pragma unroll 4
        for (int cm = 0; cm < C; ++cm) s = fmaf(bt[cm], b[(size_t)cm * nout_p], s);
        S[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * nout_p; i += blockDim.x) {
        const int co = i % nout_p, cls = i / nout_p;
        const int cz = cls / 9, cy = (cls / 3) % 3, cx = cls % 3;
        float v = 0.f;
        for (int t = 0; t < 27; ++t) {
            const int tz = t / 9, ty = (t / 3) % 3, tx = t % 3;
            if ((cz == 1 && tz == 0) || (cz == 2 && tz == 2) || (cy == 1 && ty == 0) || (cy == 2 && ty == 2) ||
                (cx == 1 && tx == 0) || (cx == 2 && tx == 2))
                continue;
            v += S[t * nout_p + co];
        }
        beff[i] = v;
    }
}

// ------------------------------------------------------------------ backward w.r.t. the three parameter tensors
// dW_eff[parity][tap][ci][co] (conv3d.hip, conv3d_wgrad_k3s_kernel<1,1,true>) and the fine-grid gradient dy are
// projected onto the reference's parameters (the composite is bilinear in WT and W3):
//   dWT[ci,cm,a] = sum_t sum_co dW_eff[pd(t,a)][ci][co] W3[co,cm,t]
//   dW3[co,cm,t] = sum_a sum_ci dW_eff[pd(t,a)][ci][co] WT[ci,cm,a]  +  V[t][co] bT[cm]
//   dbT[cm]      = sum_t sum_co V[t][co] W3[co,cm,t]
// V[t][co] = sum of dy over the fine voxels whose conv tap t lies inside the volume (per axis: t=0 excludes the first
// plane, t=2 the last) -- built from 27 box sums T[sel] (each axis: all / first plane / last plane).
constexpr int FS_SPLIT = 16;      // blocks per face / edge / corner selection (the full-volume sum goes through ctu_channel_sum)

template <class T>
__global__ __launch_bounds__(256) void upconv_face_sums_kernel(const T* __restrict__ dy, int cs, int cp, int N, int Df, int Hf,
                                                               int Wf, float* __restrict__ tpart) {
    const int sel = blockIdx.x + 1;                               // (sz*3 + sy)*3 + sx, 0 all / 1 first / 2 last; sel 0 is not computed here
    const int sz = sel / 9, sy = (sel / 3) % 3, sx = sel % 3;
    const int nz = sz ? 1 : Df, ny = sy ? 1 : Hf, nx = sx ? 1 : Wf;
    const int z0 = sz == 2 ? Df - 1 : 0, y0 = sy == 2 ? Hf - 1 : 0, x0 = sx == 2 ? Wf - 1 : 0;
    const int nq = cp >> 2;
    const int qd = threadIdx.x % nq;                              // 256 % nq == 0 for cp in {8, 16, 32, 64}
    const int64_t nvox = (int64_t)N * nz * ny * nx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t v = (int64_t)blockIdx.y * (256 / nq) + threadIdx.x / nq; v < nvox; v += (int64_t)FS_SPLIT * (256 / nq)) {
        int64_t r = v;
        const int x = x0 + (int)(r % nx); r /= nx;
        const int y = y0 + (int)(r % ny); r /= ny;
        const int z = z0 + (int)(r % nz);
        const int n = (int)(r / nz);
        const float4 g = ld4<T>(dy + ((((size_t)n * Df + z) * Hf + y) * Wf + x) * cs + qd * 4);
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
    }
    __shared__ float4 red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= nq; o >>= 1) {                         // threads with equal quad are nq apart
        if (threadIdx.x < o) {
            float4 b = red[threadIdx.x + o];
            float4& a = red[threadIdx.x];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        __syncthreads();
    }
    if (threadIdx.x < nq)
        *reinterpret_cast<float4*>(tpart + ((size_t)sel * FS_SPLIT + blockIdx.y) * cp + threadIdx.x * 4) = red[threadIdx.x];
}

__global__ void upconv_v_kernel(const float* __restrict__ tpart, const float* __restrict__ tall, int cp, float* __restrict__ V) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 27 * cp) return;
    const int co = idx % cp, t = idx / cp;
    const int tz = t / 9, ty = (t / 3) % 3, tx = t % 3;
    // per axis: tap 0 -> all - first, tap 1 -> all, tap 2 -> all - last
    float v = 0.f;
    for (int iz = 0; iz < (tz == 1 ? 1 : 2); ++iz)
        for (int iy = 0; iy < (ty == 1 ? 1 : 2); ++iy)
            for (int ix = 0; ix < (tx == 1 ? 1 : 2); ++ix) {
                const int sz = iz == 0 ? 0 : (tz == 0 ? 1 : 2), sy = iy == 0 ? 0 : (ty == 0 ? 1 : 2), sx = ix == 0 ? 0 : (tx == 0 ? 1 : 2);
                const float sign = ((iz + iy + ix) & 1) ? -1.f : 1.f;
                const int sel = (sz * 3 + sy) * 3 + sx;
                float s = 0.f;
                if (sel == 0) {
                    s = tall[co];
                } else {
                    const float* src = tpart + (size_t)sel * FS_SPLIT * cp + co;
                    for (int b = 0; b < FS_SPLIT; ++b) s += src[(size_t)b * cp];      // fixed order
                }
                v += sign * s;
            }
    V[idx] = v;
}

// (conv tap t in 0..2, transposed-conv tap a in 0..1) of one axis -> (parity bit, sub-cube tap bit) of its W_eff entry
__device__ __forceinline__ void axis_pd(int t, int a, int& pbit, int& jbit) {
    if (a == 1) { pbit = (t == 1) ? 1 : 0; jbit = (t == 2) ? 1 : 0; }        // (0,1)->(0,0) (1,1)->(1,0) (2,1)->(0,1)
    else { pbit = (t == 1) ? 0 : 1; jbit = (t == 0) ? 0 : 1; }               // (0,0)->(1,0) (1,0)->(0,1) (2,0)->(1,1)
}
__device__ __forceinline__ int weff_index(int t, int a) {                    // (parity*8 + tap) of the 3-D pair (t, a)
    int pz, jz, py, jy, px, jx;
    axis_pd(t / 9, (a >> 2) & 1, pz, jz);
    axis_pd((t / 3) % 3, (a >> 1) & 1, py, jy);
    axis_pd(t % 3, a & 1, px, jx);
    return ((pz * 4 + py * 2 + px) * 8) + (jz * 4 + jy * 2 + jx);
}

// 8 lanes per output element; grid covers dWT (C*C*8), then dW3 (Co*C*27), then dbT (C).  These sums are a few MFLOP:
// the time is the number of cache lines a wave touches per load, so (a) the dWT lanes read their c_out runs of dW_eff / W3
// as float4, (b) the dW3 outputs are dealt in c_out quads, c_out-fastest: a wave's (c_in lane, c_out quad) pairs read contiguous dW_eff
// rows as float4 (4 multiply-adds per load pair).
__global__ void upconv_project_kernel(const float* __restrict__ dweff, const float* __restrict__ V, const float* __restrict__ wtt,
                                      const float* __restrict__ w3t, const float* __restrict__ bt,
                                      const int32_t* __restrict__ imap, int C, int Co, int cin_p, int nout_p,
                                      float* __restrict__ dwt, float* __restrict__ dw3, float* __restrict__ dbt) {
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = gidx & 7;
    int idx = gidx >> 3;
    const int coq = (Co + 3) >> 2;                                // output-channel quads of the dW3 part
    const int n1 = C * C * 8, n2 = coq * C * 27, n3 = C;
    if (idx >= n1 + n2 + n3) return;
    float v = 0.f;
    float* dst;
    if (idx >= n1 && idx < n1 + n2) {                             // dW3[co .. co+3][cm][t], dealt co-quad-fastest
        const int k = idx - n1;
        const int co = (k % coq) * 4, cm = (k / coq) % C, t = k / (coq * C);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 8; ++a) {                              // 8 independent chains
            const float* g = dweff + (size_t)weff_index(t, a) * cin_p * nout_p + co;
            const float* w = wtt + (size_t)a * C * C + cm;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int ci = sub; ci < C; ci += 8) {
                const float4 gv = *reinterpret_cast<const float4*>(g + (size_t)(imap ? imap[ci] : ci) * nout_p);
                const float wv = w[(size_t)ci * C];
                s.x = fmaf(gv.x, wv, s.x); s.y = fmaf(gv.y, wv, s.y); s.z = fmaf(gv.z, wv, s.z); s.w = fmaf(gv.w, wv, s.w);
            }
            acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
        }
        float av[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (sub == 0 && co + q < Co) av[q] = fmaf(V[t * nout_p + co + q], bt[cm], av[q]);
            av[q] += __shfl_xor(av[q], 1); av[q] += __shfl_xor(av[q], 2); av[q] += __shfl_xor(av[q], 4);
            if (sub == 0 && co + q < Co) dw3[((size_t)(co + q) * C + cm) * 27 + t] = av[q];
        }
        return;
    }
    if (idx < n1) {                                               // dWT[ci][cm][a]
        const int a = idx & 7, cm = (idx >> 3) % C, ci = idx / (8 * C);
        const int pos = imap ? imap[ci] : ci;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                              // t = sub, sub + 8, sub + 16, sub + 24: independent chains
            const int t = sub + 8 * k;
            if (t < 27) {
                const float4* g = reinterpret_cast<const float4*>(dweff + ((size_t)weff_index(t, a) * cin_p + pos) * nout_p);
                const float4* w = reinterpret_cast<const float4*>(w3t + ((size_t)t * C + cm) * nout_p);
                float s = 0.f;
                for (int c4 = 0; c4 * 4 < Co; ++c4) {              // (rows are nout_p = 8k floats: 16-byte aligned)
                    const float4 gv = g[c4], wv = w[c4];
                    const int co = c4 * 4;
                    s = fmaf(gv.x, wv.x, s);
                    if (co + 1 < Co) s = fmaf(gv.y, wv.y, s);
                    if (co + 2 < Co) s = fmaf(gv.z, wv.z, s);
                    if (co + 3 < Co) s = fmaf(gv.w, wv.w, s);
                }
                v += s;
            }
        }
        dst = dwt + idx;
    } else {                                                      // dbT[cm]
        const int cm = idx - n1 - n2;
        for (int t = sub; t < 27; t += 8) {
            const float* w = w3t + ((size_t)t * C + cm) * nout_p;
            for (int co = 0; co < Co; ++co) v = fmaf(V[t * nout_p + co], w[co], v);
        }
        dst = dbt + cm;
    }
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    if (sub == 0) *dst = v;
}

static void up_grid(int ntiles, int ny, int* gx, int* tpb) {
    int g = 512 / ny;                                   // 2 resident blocks per CU
    if (g < 1) g = 1;
    if (g > ntiles) g = ntiles;
    *tpb = ceil_div(ntiles, g);
    *gx = ceil_div(ntiles, *tpb);
}

// parity groups (blockIdx.y) of the forward kernel: all 8 parities in one block for <= 16 output channels, unless the
// coarse volume has fewer than 256 boxes -- then 16-channel layers split the parities over two blocks so every CU gets one
static int up_ny(int nout_p, int ntiles) {
    if (nout_p == 16 && ntiles < 256) return 2;
    return nout_p <= 16 ? 1 : (nout_p <= 32 ? 2 : 4);
}

}  // namespace

// =================================================================== C ABI
extern "C" int ctu_upconv_fused_supported(int k, int D, int H, int W, int cin_p, int nout_p) {
    // coarse volume at least one 16-wide box, 8..64 padded output channels; the volume must be a multiple of the
    // 4x4x16 box only for speed, not for correctness
    return k == 3 && W >= 16 && D >= 1 && H >= 1 && cin_p % 8 == 0 && cin_p >= 8 && nout_p % 8 == 0 && nout_p >= 8 && nout_p <= 64;
}

static size_t up_std_packed(int cin_p, int nout_p) { return (size_t)(cin_p / 8) * 8 * 8 * ceil_div(nout_p, 16) * 128; }

// standard packing [chunk][parity 8][tap 8][n16][128]; for 8 padded output channels followed by the PW forward packing
extern "C" size_t ctu_upconv_fused_packed_floats(int cin_p, int nout_p) {
    return up_std_packed(cin_p, nout_p) + (nout_p == 8 ? (size_t)(cin_p / 8) * 4 * 12 * 128 : 0);
}

extern "C" int ctu_upconv_fused_num_blocks(int N, int D, int H, int W, int nout_p) {
    int gx, tpb;
    const int ntiles = N * ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 16);
    const int ny = up_ny(nout_p, ntiles);
    up_grid(ntiles, ny, &gx, &tpb);
    return gx * ny;
}

extern "C" size_t ctu_upconv_fused_pack_ws_floats(int C, int nout_p) { return (size_t)27 * C * nout_p + (size_t)8 * C * C; }

extern "C" int ctu_upconv_fused_pack(const float* wt, const float* bt, const float* w3, int C, int Co, const int32_t* cinv,
                                     int cin_p, int nout_p, float* wp, float* beff, float* ws, void* stream) {
    CTU_REQUIRE(wt && bt && w3 && wp && beff && ws, "upconv_fused_pack: null pointer");
    CTU_REQUIRE(C > 0 && Co > 0 && Co <= nout_p && cin_p % 8 == 0 && nout_p % 8 == 0 && (cinv || C <= cin_p),
                "upconv_fused_pack: C=%d Co=%d cin_p=%d nout_p=%d", C, Co, cin_p, nout_p);
    hipStream_t st = (hipStream_t)stream;
    const int ntpt = ceil_div(nout_p, 16);
    CTU_REQUIRE(((uintptr_t)ws & 15) == 0, "upconv_fused_pack: ws must be 16-byte aligned");
    float* w3t = ws;
    float* wtt = ws + (size_t)27 * C * nout_p;
    const int ntr = 27 * C * nout_p + 8 * C * C;
    upconv_transpose_kernel<<<ceil_div(ntr, 256), 256, 0, st>>>(wt, w3, wtt, w3t, C, Co, nout_p);
    CTU_CHECK_LAUNCH("upconv_fused_transpose");
    const int total = (cin_p / 8) * 8 * 8 * 8 * ntpt * 4 * 8;            // 8 lanes per quad of packed elements
    upconv_pack_kernel<<<ceil_div(total, 256), 256, 0, st>>>(wtt, w3t, wp, C, nout_p, cinv, cin_p / 8, ntpt);
    CTU_CHECK_LAUNCH("upconv_fused_pack");
    if (nout_p == 8) {
        const int tpw = (cin_p / 8) * 4 * 12 * 128;
        upconv_pack_pw_kernel<<<ceil_div(tpw, 256), 256, 0, st>>>(wp, wp + up_std_packed(cin_p, nout_p), cin_p / 8);
        CTU_CHECK_LAUNCH("upconv_fused_pack_pw");
    }
    upconv_beff_kernel<<<1, 1024, 0, st>>>(bt, w3t, beff, C, nout_p);
    CTU_CHECK_LAUNCH("upconv_fused_beff");
    return CTU_OK;
}

extern "C" int ctu_upconv_fused_fwd(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                    int in_relu, const float* wp, const float* beff, float* out, int out_cs, int nout_p,
                                    float* stats, int N, int D, int H, int W, const ctu_bn_tail* tail, void* stream) {
    CTU_REQUIRE(in && wp && beff && out, "upconv_fused_fwd: null pointer");
    CTU_REQUIRE(ctu_upconv_fused_supported(3, D, H, W, cin_p, nout_p), "upconv_fused_fwd: unsupported geometry (W=%d cin_p=%d nout_p=%d)",
                W, cin_p, nout_p);
    CTU_REQUIRE(in_cs >= cin_p && in_cs % 4 == 0 && out_cs >= nout_p && out_cs % 4 == 0, "upconv_fused_fwd: bad stride");
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "upconv_fused_fwd: scale/shift must come together");
    CTU_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)wp & 15) == 0 && ((uintptr_t)beff & 15) == 0,
                "upconv_fused_fwd: 16-byte alignment");
    CTU_REQUIRE((int64_t)N * D * H * W * 8 < (int64_t)1 << 31 && (int64_t)(2 * H * W + 2 * W + 2) * in_cs * 4 < (int64_t)1 << 31,
                "upconv_fused_fwd: volume too large for 32-bit offsets");
    UpP p;
    p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.wp = wp; p.beff = beff; p.out = out; p.stats = stats;
    CTU_REQUIRE(!tail || (stats && tail->counter && tail->gamma && tail->beta && tail->scale && tail->shift && tail->mean &&
                          tail->invstd && tail->C > 0 && tail->C <= nout_p && tail->count > 0),
                "upconv_fused_fwd: incomplete BatchNorm tail");
    p.tail = tail_or_off(tail);
    p.in_cs = in_cs; p.rin_p = cin_p; p.in_relu = in_relu; p.out_cs = out_cs; p.nout_p = nout_p;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, 4); p.tiles_w = ceil_div(W, 16);
    const int ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
    const int ny = up_ny(nout_p, ntiles);
    int gx, tpb;
    up_grid(ntiles, ny, &gx, &tpb);
    hipStream_t st = (hipStream_t)stream;
    if (nout_p == 16 && ny == 2) {
        upconv_fused_fwd_kernel<4, 1><<<dim3(gx, 2), 256, 0, st>>>(p, ntiles, tpb);
    } else if (nout_p == 8) {
        p.wp = wp + up_std_packed(cin_p, nout_p);                 // w-parity-in-tile packing
        upconv_fused_fwd_kernel<4, 1, true><<<dim3(gx, 1), 256, 0, st>>>(p, ntiles, tpb);
    } else if (ny == 1) upconv_fused_fwd_kernel<8, 1><<<dim3(gx, 1), 256, 0, st>>>(p, ntiles, tpb);
    else if (ny == 2) upconv_fused_fwd_kernel<4, 2><<<dim3(gx, 2), 256, 0, st>>>(p, ntiles, tpb);
    else upconv_fused_fwd_kernel<2, 4><<<dim3(gx, 4), 256, 0, st>>>(p, ntiles, tpb);
    CTU_CHECK_LAUNCH("upconv_fused_fwd");
    return CTU_OK;
}

extern "C" int ctu_channel_sum_num_blocks(int64_t nvox);
extern "C" int ctu_channel_sum(const float* x, int cs, int cp, int64_t nvox, float* partials, float* out, int C, void* stream);

extern "C" size_t ctu_upconv_fused_project_ws_floats(int nout_p, int64_t fine_nvox) {
    return (size_t)27 * FS_SPLIT * nout_p + (size_t)27 * nout_p + nout_p + (size_t)ctu_channel_sum_num_blocks(fine_nvox) * nout_p;
}

// dweff: [8][8][cin_p][nout_p] from ctu_upconv_fused_wgrad; gout: fine-grid gradient w.r.t. the fused op's RAW output
// [N,2D,2H,2W,g_cs]; pack_ws: the scratch ctu_upconv_fused_pack filled this step (transposed weights); imap: logical input
// channel -> padded position (NULL = identity).  Outputs in torch layouts: dwt [C][C][2][2][2], dbt [C], dw3 [Co][C][3][3][3].
extern "C" int ctu_lp_channel_sum(int dtype, const void* x, int cs, int cp, int64_t nvox, float* partials, float* out, int C, void* stream);

// dtype 0: gout is fp32; CTU_BF16 / CTU_F16: a 16-bit gradient tensor (the 16-bit path's ctu_lp_upconv_fused_project)
static int upconv_project_impl(int dtype, const float* dweff, const void* gout, int g_cs, int nout_p, int N, int D, int H, int W,
                               const float* bt, const float* pack_ws, const int32_t* imap, int C, int Co, int cin_p,
                               float* dwt, float* dbt, float* dw3, float* ws, void* stream) {
    CTU_REQUIRE(dweff && gout && bt && pack_ws && dwt && dbt && dw3 && ws, "upconv_fused_project: null pointer");
    CTU_REQUIRE(nout_p == 8 || nout_p == 16 || nout_p == 32 || nout_p == 64, "upconv_fused_project: nout_p=%d", nout_p);
    CTU_REQUIRE(C > 0 && Co > 0 && Co <= nout_p && g_cs >= nout_p && g_cs % 4 == 0, "upconv_fused_project: bad channels");
    CTU_REQUIRE(((uintptr_t)dweff & 15) == 0 && ((uintptr_t)pack_ws & 15) == 0, "upconv_fused_project: dweff / pack_ws must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float* tpart = ws;
    float* V = ws + (size_t)27 * FS_SPLIT * nout_p;
    float* tall = V + (size_t)27 * nout_p;
    float* csp = tall + nout_p;
    const int64_t fvox = (int64_t)N * 8 * D * H * W;
    int rc;
    if (dtype == 0) {
        rc = ctu_channel_sum((const float*)gout, g_cs, nout_p, fvox, csp, tall, nout_p, stream);          // full-volume sum per channel
        if (rc != CTU_OK) return rc;
        upconv_face_sums_kernel<float><<<dim3(26, FS_SPLIT), 256, 0, st>>>((const float*)gout, g_cs, nout_p, N, 2 * D, 2 * H, 2 * W, tpart);
    } else {
        rc = ctu_lp_channel_sum(dtype, gout, g_cs, nout_p, fvox, csp, tall, nout_p, stream);
        if (rc != CTU_OK) return rc;
        CTU_DISPATCH_LP(dtype, upconv_face_sums_kernel<T><<<dim3(26, FS_SPLIT), 256, 0, st>>>((const T*)gout, g_cs, nout_p, N, 2 * D, 2 * H, 2 * W, tpart));
    }
    CTU_CHECK_LAUNCH("upconv_face_sums");
    upconv_v_kernel<<<ceil_div(27 * nout_p, 64), 64, 0, st>>>(tpart, tall, nout_p, V);
    CTU_CHECK_LAUNCH("upconv_v");
    const float* w3t = pack_ws;
    const float* wtt = pack_ws + (size_t)27 * C * nout_p;
    const int total = (C * C * 8 + ((Co + 3) / 4) * C * 27 + C) * 8;
    upconv_project_kernel<<<ceil_div(total, 256), 256, 0, st>>>(dweff, V, wtt, w3t, bt, imap, C, Co, cin_p, nout_p, dwt, dw3, dbt);
    CTU_CHECK_LAUNCH("upconv_project");
    return CTU_OK;
}

extern "C" int ctu_upconv_fused_project(const float* dweff, const float* gout, int g_cs, int nout_p, int N, int D, int H, int W,
                                        const float* bt, const float* pack_ws, const int32_t* imap, int C, int Co, int cin_p,
                                        float* dwt, float* dbt, float* dw3, float* ws, void* stream) {
    return upconv_project_impl(0, dweff, gout, g_cs, nout_p, N, D, H, W, bt, pack_ws, imap, C, Co, cin_p, dwt, dbt, dw3, ws, stream);
}
extern "C" int ctu_lp_upconv_fused_project(int dtype, const float* dweff, const void* gout, int g_cs, int nout_p, int N, int D, int H,
                                           int W, const float* bt, const float* pack_ws, const int32_t* imap, int C, int Co, int cin_p,
                                           float* dwt, float* dbt, float* dw3, float* ws, void* stream) {
    CTU_REQUIRE(dtype == CTU_BF16 || dtype == CTU_F16, "lp_upconv_fused_project: dtype %d", dtype);
    return upconv_project_impl(dtype, dweff, gout, g_cs, nout_p, N, D, H, W, bt, pack_ws, imap, C, Co, cin_p, dwt, dbt, dw3, ws, stream);
}

// ---- fused data gradient: gin (coarse) from the fine-grid gradient of the fused op's raw output
extern "C" size_t ctu_upconv_fused_bwd_packed_floats(int cin_p, int nout_p) {
    return (size_t)8 * (nout_p / 8) * 8 * ceil_div(cin_p, 16) * 128;
}

extern "C" int ctu_upconv_fused_pack_bwd(const float* wp, int cin_p, int nout_p, float* wpd, void* stream) {
    CTU_REQUIRE(wp && wpd && cin_p % 8 == 0 && nout_p % 8 == 0, "upconv_fused_pack_bwd: bad argument");
    const int total = (int)ctu_upconv_fused_bwd_packed_floats(cin_p, nout_p);
    upconv_pack_bwd_kernel<<<ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(wp, wpd, cin_p, nout_p);
    CTU_CHECK_LAUNCH("upconv_fused_pack_bwd");
    return CTU_OK;
}

extern "C" int ctu_upconv_fused_bwd_data(const float* gout, int g_cs, int nout_p, const float* wpd, float* gin, int gin_cs,
                                         int cin_p, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(gout && wpd && gin, "upconv_fused_bwd_data: null pointer");
    CTU_REQUIRE(ctu_upconv_fused_supported(3, D, H, W, cin_p, nout_p), "upconv_fused_bwd_data: unsupported geometry");
    CTU_REQUIRE(g_cs >= nout_p && g_cs % 4 == 0 && gin_cs >= cin_p && gin_cs % 4 == 0, "upconv_fused_bwd_data: bad stride");
    CTU_REQUIRE(((uintptr_t)gout & 15) == 0 && ((uintptr_t)gin & 15) == 0 && ((uintptr_t)wpd & 15) == 0, "upconv_fused_bwd_data: alignment");
    CTU_REQUIRE((int64_t)(12 * 2 * H + 12) * 2 * W * g_cs * 4 + 64 < (int64_t)1 << 31, "upconv_fused_bwd_data: volume too large for 32-bit offsets");
    UpDP p;
    p.g = gout; p.wp = wpd; p.out = gin; p.g_cs = g_cs; p.nout_p = nout_p; p.out_cs = gin_cs; p.cin_p = cin_p;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.tiles_d = ceil_div(D, 4); p.tiles_h = ceil_div(H, 4); p.tiles_w = ceil_div(W, 16);
    p.n16 = ceil_div(cin_p, 16);
    const int ntiles = N * p.tiles_d * p.tiles_h * p.tiles_w;
    int NT = p.n16 >= 4 ? 4 : (p.n16 >= 2 ? 2 : 1);
    while (NT > 1 && (long)ntiles * ceil_div(p.n16, NT) < 256) NT >>= 1;      // few boxes: narrower blocks so every CU gets one
    const int ny = ceil_div(p.n16, NT);
    int gx, tpb;
    up_grid(ntiles, ny, &gx, &tpb);
    hipStream_t st = (hipStream_t)stream;
    if (NT == 4) upconv_fused_bwd_data_kernel<4><<<dim3(gx, ny), 256, 0, st>>>(p, ntiles, tpb);
    else if (NT == 2) upconv_fused_bwd_data_kernel<2><<<dim3(gx, ny), 256, 0, st>>>(p, ntiles, tpb);
    else upconv_fused_bwd_data_kernel<1><<<dim3(gx, ny), 256, 0, st>>>(p, ntiles, tpb);
    CTU_CHECK_LAUNCH("upconv_fused_bwd_data");
    return CTU_OK;
}
