// ConvTranspose3d(kernel 2, stride 2, bias) for gfx950 -- replaces nn.ConvTranspose3d at
// ctunet/pytorch/models.py:37 and :427-429 and its autograd backward.
//
// With kernel == stride every output voxel receives exactly one tap, so the op is a plain GEMM
//   [input voxels x Ci] . [Ci x (8 taps x Co)]  followed by a scatter to (2d+i, 2h+j, 2w+l).
// A block owns 64 consecutive input voxels (4 waves x one 16-voxel M-tile), keeps their
// activated channels in LDS (row stride Ci+4 floats -> conflict-free ds_read_b64) and walks the
// 8 taps, staging one tap's packed weights at a time.  v_mfma_f32_16x16x4_f32, exact fp32.
// The op is HBM-bound (the 8x larger output stream), see DESIGN.md.
#include "common.h"
#include "pack.h"

namespace {

struct CtP {
    const float* in;         // fwd: input [V][in_cs]; bwd-data: gout on the 2x grid
    const float* in_scale;
    const float* in_shift;
    const float* wp;
    const float* bias;
    float* out;              // fwd: output on the 2x grid; bwd-data: gin [V][out_cs]
    int in_cs, rin_p, in_relu, out_cs, nout_p, nbias;
    int ntt_total;           // 16-wide output tiles in the packed weights (power of two)
    int N, D, H, W;          // coarse (input-side) grid
    int64_t nvox;            // N*D*H*W
};

// Fine-grid voxel index of tap 0 for coarse voxel v (32-bit division: emulated 64-bit div/mod by runtime sizes is
// ~10x more expensive and used to dominate these HBM-bound kernels), and the per-tap offset on the fine grid.
__device__ __forceinline__ size_t fine_base(int64_t v, int D, int H, int W) {
    const unsigned u = (unsigned)v;
    const unsigned w = u % (unsigned)W, t1 = u / (unsigned)W;
    const unsigned h = t1 % (unsigned)H, t2 = t1 / (unsigned)H;
    const unsigned d = t2 % (unsigned)D, n = t2 / (unsigned)D;
    return (((size_t)n * (2 * D) + 2 * d) * (2 * H) + 2 * h) * (size_t)(2 * W) + 2 * w;
}
__device__ __forceinline__ int tap_off(int tap, int H, int W) {
    return ((tap >> 2) * (2 * H) + ((tap >> 1) & 1)) * (2 * W) + (tap & 1);
}

// MODE 0: forward (A staged once, one store per tap).  MODE 1: data gradient (A gathered per tap
// from the fine grid, accumulated over the 8 taps, one store).
// The MFMA operands are swapped (weights as the row operand, voxels as the column operand) so that a
// lane ends up with 4 CONSECUTIVE output channels of one voxel: the epilogue is one 16-byte store per
// lane instead of four scattered dword stores.  `tps` taps' weights are staged per barrier pair.
template <int NTT, int MODE>
__global__ __launch_bounds__(256) void convt2_kernel(CtP p, int tps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int AS = p.rin_p + 4;                       // LDS row stride (floats)
    float* sA = smem;                                 // [64][AS]
    float* sW = smem + 64 * AS;                       // [tps][rin_p/8][NTT][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    const int ng = p.rin_p >> 3;
    const int nq = p.rin_p >> 2;                      // float4 per voxel
    const bool has_xf = p.in_scale != nullptr;
    const int wfl = ng * NTT * 128;                   // this block's share of one tap: NTT of the ntt_total tiles
    const int by = blockIdx.y;
    const int64_t vme = v0 + wave * 16 + m;           // this lane's voxel (column of the MFMA tile)
    const size_t fme = vme < p.nvox ? fine_base(vme, p.D, p.H, p.W) : 0;
    // MODE 1 staging: this thread's gather items (voxel of item k = (tid + 256 k) / nq) and their fine-grid bases
    constexpr int GMAX = 8;                           // 64 * nq / 256 <= 8 for rin_p <= 128
    size_t gbase[GMAX];
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < GMAX; ++k) {
            const int it = tid + k * 256;
            const int64_t v = v0 + it / nq;
            gbase[k] = (it < 64 * nq && v < p.nvox) ? fine_base(v, p.D, p.H, p.W) : (size_t)-1;
        }
    }

    f32x4 acc[NTT];
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tap0 = 0; tap0 < 8; tap0 += tps) {
        __syncthreads();
        // Staging is all latency here (deep levels: a few hundred blocks, one per CU): every batch of loads is issued
        // together, branch-free (items past the end re-read a valid address and are not stored), then stored.
        if (MODE == 0 && tap0 == 0) {
            f32x4 va[GMAX];
#pragma unroll
            for (int k = 0; k < GMAX; ++k) {
                const int it = tid + k * 256;
                const int vl = it / nq, qd = it % nq;
                const bool ok = it < 64 * nq && v0 + vl < p.nvox;
                va[k] = *reinterpret_cast<const f32x4*>(p.in + (ok ? (size_t)(v0 + vl) * p.in_cs : (size_t)0) + qd * 4);
            }
#pragma unroll
            for (int k = 0; k < GMAX; ++k) {
                const int it = tid + k * 256;
                const int vl = it / nq, qd = it % nq;
                if (it < 64 * nq) {
                    float4 val = make_float4(va[k][0], va[k][1], va[k][2], va[k][3]);
                    if (has_xf)
                        val = xform4(val, *reinterpret_cast<const float4*>(p.in_scale + qd * 4),
                                     *reinterpret_cast<const float4*>(p.in_shift + qd * 4), p.in_relu);
                    if (v0 + vl >= p.nvox) val = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(&sA[vl * AS + qd * 4]) = val;
                }
            }
        }
        {
            // packed weights: [tap][g][ntt_total][128]; this block takes tiles by*NTT .. by*NTT+NTT-1
            const int tot = tps * wfl;
            for (int i0 = tid * 4; i0 < tot; i0 += 8 * 1024) {
                f32x4 wv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = (i0 + k * 1024 < tot) ? i0 + k * 1024 : i0;
                    const int tl = i / wfl, rem = i % wfl, g = rem / (NTT * 128), r = rem % (NTT * 128);
                    wv[k] = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)((tap0 + tl) * ng + g) * p.ntt_total + by * NTT) * 128 + r);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (i0 + k * 1024 < tot) *reinterpret_cast<f32x4*>(&sW[i0 + k * 1024]) = wv[k];
            }
        }
        if (MODE == 0) __syncthreads();
        for (int tl = 0; tl < tps; ++tl) {
            const int tap = tap0 + tl;
            if (MODE == 1) {
                // gather this tap's fine-grid voxels (the block's 64 coarse voxels) into sA
                __syncthreads();
                const int toff = tap_off(tap, p.H, p.W);
                f32x4 vg[GMAX];
#pragma unroll
                for (int k = 0; k < GMAX; ++k) {
                    const int qd = (tid + k * 256) % nq;
                    const bool ok = gbase[k] != (size_t)-1;
                    vg[k] = *reinterpret_cast<const f32x4*>(p.in + (ok ? (gbase[k] + toff) * p.in_cs : (size_t)0) + qd * 4);
                }
#pragma unroll
                for (int k = 0; k < GMAX; ++k) {
                    const int it = tid + k * 256;
                    if (it < 64 * nq) {
                        const int vl = it / nq, qd = it % nq;
                        const bool ok = gbase[k] != (size_t)-1;
                        *reinterpret_cast<f32x4*>(&sA[vl * AS + qd * 4]) = ok ? vg[k] : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int nt = 0; nt < NTT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const float* arow = &sA[(wave * 16 + m) * AS + kq * 2];
            const float* brow = &sW[tl * wfl + kq * 32 + m * 2];
            for (int g = 0; g < ng; ++g) {
                const float2 a = *reinterpret_cast<const float2*>(arow + g * 8);
#pragma unroll
                for (int nt = 0; nt < NTT; ++nt) {
                    const float2 b = *reinterpret_cast<const float2*>(brow + (g * NTT + nt) * 128);
                    // rows = output channels (weights), columns = voxels
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a.x, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a.y, acc[nt], 0, 0, 0);
                }
            }
            if (MODE == 0 && vme < p.nvox) {
                float* orow = p.out + (fme + tap_off(tap, p.H, p.W)) * p.out_cs;
#pragma unroll
                for (int nt = 0; nt < NTT; ++nt) {
                    const int co = (by * NTT + nt) * 16 + kq * 4;   // this lane: channels co .. co+3 of voxel vme
                    if (co < p.nout_p) {
                        float4 o;
                        o.x = acc[nt][0] + ((p.bias && co + 0 < p.nbias) ? p.bias[co + 0] : 0.f);
                        o.y = acc[nt][1] + ((p.bias && co + 1 < p.nbias) ? p.bias[co + 1] : 0.f);
                        o.z = acc[nt][2] + ((p.bias && co + 2 < p.nbias) ? p.bias[co + 2] : 0.f);
                        o.w = acc[nt][3] + ((p.bias && co + 3 < p.nbias) ? p.bias[co + 3] : 0.f);
                        *reinterpret_cast<float4*>(orow + co) = o;
                    }
                }
            }
        }
    }
    if (MODE == 1 && vme < p.nvox) {
        float* orow = p.out + (size_t)vme * p.out_cs;
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
            const int co = (by * NTT + nt) * 16 + kq * 4;
            if (co < p.nout_p)
                *reinterpret_cast<float4*>(orow + co) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
        }
    }
}

inline int pick_ntt(int nout_p) {
    const int n16 = (nout_p + 15) / 16;
    return n16 <= 1 ? 1 : (n16 <= 2 ? 2 : (n16 <= 4 ? 4 : 8));
}

// wp[tap][g][nt][kq][n][j]; one thread per packed element (gather, no memset needed).
// cinv: padded input-channel position -> logical channel (-1 = padding, NULL = identity).
__global__ void pack_convt_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Ci, int Co,
                                    const int32_t* __restrict__ cinv, int rin_p, int NTT, int mode) {
    pack_convt_w_elem(blockIdx.x * blockDim.x + threadIdx.x, w, wp, Ci, Co, cinv, rin_p, NTT, mode);
}

// ------------------------------------------------------------------ weight gradient
struct CtWgP {
    const float* in;
    const float* in_scale;
    const float* in_shift;
    const float* g;
    float* ws;
    int in_cs, cin_p, in_relu, g_cs, cout_p;
    int N, D, H, W;
    int64_t nvox;
    int ntiles, n_ci_t;
};

// M = MI x 16 input channels, N = NJ x 16 output channels, K = coarse voxels; one accumulator per (tap, tile pair).
// A block with 2 x 2 tiles reads the (8x larger) fine-grid gradient ONCE for 32 x 32 channels.  The bias gradient
// (sum of gout over voxels) is accumulated by the staging threads from the values they load anyway.
template <int MI, int NJ>
__global__ __launch_bounds__(256) void convt2_wgrad_kernel(CtWgP p, float* __restrict__ wsb) {
    constexpr int CA = 16 * MI, CG = 16 * NJ, QA = 4 * MI, QG = 4 * NJ;
    __shared__ __attribute__((aligned(16))) float sA[64 * CA];
    __shared__ __attribute__((aligned(16))) float sG[8 * 64 * CG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int cig = blockIdx.y % p.n_ci_t, cog = blockIdx.y / p.n_ci_t;
    const int ci0 = cig * CA, co0 = cog * CG;
    const int qa = tid % QA, qg = tid % QG;                    // 256 % QA == 0: fixed channel quad per thread
    const bool a_ok = (ci0 + qa * 4) < p.cin_p, g_ok = (co0 + qg * 4) < p.cout_p;
    const bool has_xf = p.in_scale != nullptr;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_xf && a_ok) {
        sc = *reinterpret_cast<const float4*>(p.in_scale + ci0 + qa * 4);
        sh = *reinterpret_cast<const float4*>(p.in_shift + ci0 + qa * 4);
    }
    f32x4 acc[8][MI][NJ];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NJ; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t v0 = (int64_t)tile * 64;
        __syncthreads();
        for (int it = tid; it < 64 * QA; it += 256) {
            const int vl = it / QA;
            const int64_t v = v0 + vl;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_ok && v < p.nvox) {
                val = *reinterpret_cast<const float4*>(p.in + (size_t)v * p.in_cs + ci0 + qa * 4);
                if (has_xf) val = xform4(val, sc, sh, p.in_relu);
            }
            *reinterpret_cast<float4*>(&sA[vl * CA + qa * 4]) = val;
        }
        // a thread owns channel quad qg of QG/4 voxels (vl = tid / QG + j * (256 / QG)); one 32-bit index decode per voxel
#pragma unroll
        for (int j = 0; j < QG / 4; ++j) {
            const int vl = tid / QG + j * (256 / QG);
            const int64_t v = v0 + vl;
            const bool ok = g_ok && v < p.nvox;
            // the 8 taps' loads are issued together, branch-free (items outside read voxel 0 / quad 0 and are zeroed)
            const float* gsrc = p.g + (ok ? fine_base(v, p.D, p.H, p.W) : 0) * p.g_cs + (g_ok ? co0 + qg * 4 : 0);
            f32x4 gl[8];
#pragma unroll
            for (int tap = 0; tap < 8; ++tap) gl[tap] = *reinterpret_cast<const f32x4*>(gsrc + (size_t)tap_off(tap, p.H, p.W) * p.g_cs);
#pragma unroll
            for (int tap = 0; tap < 8; ++tap) {
                const float4 gv = ok ? make_float4(gl[tap][0], gl[tap][1], gl[tap][2], gl[tap][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&sG[(tap * 64 + vl) * CG + qg * 4]) = gv;
                gsum.x += gv.x; gsum.y += gv.y; gsum.z += gv.z; gsum.w += gv.w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int vl = (wave * 4 + ks) * 4 + kq;
            float a[MI];
#pragma unroll
            for (int x = 0; x < MI; ++x) a[x] = sA[vl * CA + x * 16 + i];
#pragma unroll
            for (int tap = 0; tap < 8; ++tap) {
#pragma unroll
                for (int y = 0; y < NJ; ++y) {
                    const float b = sG[(tap * 64 + vl) * CG + y * 16 + i];
#pragma unroll
                    for (int x = 0; x < MI; ++x)
                        acc[tap][x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b, acc[tap][x][y], 0, 0, 0);
                }
            }
        }
    }
    // 4 waves -> one slab [8][MI][NJ][16 ci][16 co] per block, through sG (8 taps per round need 4*8*MI*NJ*256 floats)
    float* dst = p.ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (8 * MI * NJ * 256);
    constexpr int TPR = (8 * 64 * CG) / (4 * MI * NJ * 256);        // taps per round that fit in sG
    static_assert(TPR >= 1, "reduction scratch");
    for (int t0 = 0; t0 < 8; t0 += TPR) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t >= t0 && t < t0 + TPR)
#pragma unroll
                for (int x = 0; x < MI; ++x)
#pragma unroll
                    for (int y = 0; y < NJ; ++y)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            sG[((wave * TPR + (t - t0)) * MI * NJ + x * NJ + y) * 256 + (kq * 4 + r) * 16 + i] = acc[t][x][y][r];
        __syncthreads();
        const int nel = ((8 - t0) < TPR ? (8 - t0) : TPR) * MI * NJ * 256, stride = TPR * MI * NJ * 256;
        for (int e = tid; e < nel; e += 256)
            dst[t0 * MI * NJ * 256 + e] = (sG[e] + sG[stride + e]) + (sG[2 * stride + e] + sG[3 * stride + e]);
    }
    // bias-gradient partials: threads with the same channel quad, only the blocks of the first ci group
    if (cig == 0 && wsb) {
        __syncthreads();
        *reinterpret_cast<float4*>(&sA[tid * 4]) = gsum;
        __syncthreads();
        if (tid < CG) {
            const int q = tid >> 2, c = tid & 3;
            float t = 0.f;
            for (int k = q; k < 256; k += QG) t += sA[k * 4 + c];
            wsb[((size_t)cog * gridDim.x + blockIdx.x) * CG + tid] = t;
        }
    }
}

// dw[ci][co][tap]: a block owns 64 consecutive outputs (co fastest) and sums the gx slabs of their tile group.
template <int MI, int NJ>
__global__ __launch_bounds__(1024) void convt2_wgrad_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ wsb,
                                                                   float* __restrict__ dw, float* __restrict__ dbias, int Ci,
                                                                   int Co, const int32_t* __restrict__ imap, int n_ci_t,
                                                                   int gx) {
    constexpr int CA = 16 * MI, CG = 16 * NJ, SL = 8 * MI * NJ * 256;
    __shared__ float red[RPARTS][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;
    float s = 0.f;
    const int ndw = Ci * Co * 8;
    const bool ok = idx < ndw, okb = dbias && idx >= ndw && idx < ndw + Co;
    int co = 0, ci = 0, tap = 0;
    if (ok) {
        co = idx % Co; ci = (idx / Co) % Ci; tap = idx / (Co * Ci);
        const int cip = imap ? imap[ci] : ci;      // logical -> padded position
        const int cig = cip / CA, cog = co / CG, x = (cip % CA) >> 4, y = (co % CG) >> 4;
        const float* src = ws + ((size_t)(cog * n_ci_t + cig) * gx) * SL + ((tap * MI + x) * NJ + y) * 256 + (cip & 15) * 16 + (co & 15);
        for (int k = part; k < gx; k += RPARTS) s += src[(size_t)k * SL];
    } else if (okb) {
        co = idx - ndw;
        const float* src = wsb + ((size_t)(co / CG) * gx) * CG + co % CG;
        for (int k = part; k < gx; k += RPARTS) s += src[(size_t)k * CG];
    }
    red[part][e] = s;
    __syncthreads();
    if (part == 0 && ok) dw[((size_t)ci * Co + co) * 8 + tap] = red_total(red, e);
    if (part == 0 && okb) dbias[co] = red_total(red, e);
}

inline int ct_wgrad_gx(int ntiles, int pairs) {
    int gx = 512 / pairs;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    return gx;
}

template <int MODE>
int launch_convt(const CtP& p, hipStream_t st, const char* name) {
    const int ntt_total = pick_ntt(p.nout_p);
    const unsigned grid = (unsigned)ceil_div64(p.nvox, 64);
    int ntt = ntt_total;                                   // tiles per block: split across blockIdx.y until the grid fills the chip
    while (ntt > 1 && (long)grid * (ntt_total / ntt) < 256) ntt >>= 1;
    const size_t a_b = (size_t)64 * (p.rin_p + 4) * sizeof(float);
    const size_t w_b = (size_t)(p.rin_p / 8) * ntt * 128 * sizeof(float);      // one tap's share of the packed weights
    int tps = 8;                                                                // taps staged per barrier pair
    while (tps > 1 && a_b + tps * w_b > 72 * 1024) tps >>= 1;
    const size_t lds = a_b + tps * w_b;
    CTU_REQUIRE(lds <= 160 * 1024, "%s: rin_p=%d nout_p=%d needs %zu B of LDS", name, p.rin_p, p.nout_p, lds);
    CTU_REQUIRE(p.out_cs % 4 == 0 && ((uintptr_t)p.out & 15) == 0, "%s: output must be 16-byte aligned", name);
    CTU_REQUIRE(p.nvox < (1LL << 31), "%s: more than 2^31 coarse voxels", name);
    CtP q = p;
    q.ntt_total = ntt_total;
    const dim3 grid2(grid, ntt_total / ntt);
#define CT_LAUNCH(N_)                                                                                             \
    do {                                                                                                          \
        if (lds > 64 * 1024)                                                                                      \
            (void)hipFuncSetAttribute((const void*)convt2_kernel<N_, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                  \
        convt2_kernel<N_, MODE><<<grid2, 256, lds, st>>>(q, tps);                                                 \
    } while (0)
    switch (ntt) {
        case 1: CT_LAUNCH(1); break;
        case 2: CT_LAUNCH(2); break;
        case 4: CT_LAUNCH(4); break;
        default: CT_LAUNCH(8); break;
    }
#undef CT_LAUNCH
    CTU_CHECK_LAUNCH(name);
    return CTU_OK;
}

}  // namespace

// =================================================================== C ABI
extern "C" size_t ctu_convt_packed_floats(int rin_p, int nout_p) {
    if (rin_p <= 0 || nout_p <= 0) return 0;
    return (size_t)8 * (rin_p / 8) * pick_ntt(nout_p) * 128;
}

extern "C" int ctu_pack_convt_weight(const float* w, float* wp, int Ci, int Co, const int32_t* cinv, int rin_p,
                                     int nout_p, int mode, void* stream) {
    CTU_REQUIRE(w && wp && Ci > 0 && Co > 0, "pack_convt_weight: null/empty");
    CTU_REQUIRE(rin_p % 8 == 0 && nout_p % 8 == 0 && nout_p <= 128, "pack_convt_weight: rin_p=%d nout_p=%d", rin_p,
                nout_p);
    hipStream_t st = (hipStream_t)stream;
    const int total = (int)ctu_convt_packed_floats(rin_p, nout_p);
    pack_convt_w_kernel<<<ceil_div(total, 256), 256, 0, st>>>(w, wp, Ci, Co, cinv, rin_p, pick_ntt(nout_p), mode);
    CTU_CHECK_LAUNCH("pack_convt_weight");
    return CTU_OK;
}

extern "C" int ctu_convt2_fwd(const float* in, int in_cs, int rin_p, const float* in_scale, const float* in_shift,
                              int in_relu, const float* wp, const float* bias, int nbias, float* out, int out_cs,
                              int nout_p, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(in && wp && out, "convt2_fwd: null pointer");
    CTU_REQUIRE(rin_p > 0 && rin_p % 8 == 0 && nout_p > 0 && nout_p % 8 == 0 && nout_p <= 128,
                "convt2_fwd: rin_p=%d nout_p=%d", rin_p, nout_p);
    CTU_REQUIRE(in_cs >= rin_p && in_cs % 4 == 0 && out_cs >= nout_p, "convt2_fwd: bad stride");
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "convt2_fwd: scale/shift must come together");
    CtP p;
    p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.wp = wp; p.bias = bias; p.out = out;
    p.in_cs = in_cs; p.rin_p = rin_p; p.in_relu = in_relu; p.out_cs = out_cs; p.nout_p = nout_p;
    p.nbias = bias ? nbias : 0;
    p.N = N; p.D = D; p.H = H; p.W = W; p.nvox = (int64_t)N * D * H * W;
    return launch_convt<0>(p, (hipStream_t)stream, "convt2_fwd");
}

extern "C" int ctu_convt2_bwd_data(const float* gout, int g_cs, int rout_p, const float* wp, float* gin, int gin_cs,
                                   int nin_p, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(gout && wp && gin, "convt2_bwd_data: null pointer");
    CTU_REQUIRE(rout_p > 0 && rout_p % 8 == 0 && nin_p > 0 && nin_p % 8 == 0 && nin_p <= 128,
                "convt2_bwd_data: rout_p=%d nin_p=%d", rout_p, nin_p);
    CTU_REQUIRE(g_cs >= rout_p && g_cs % 4 == 0 && gin_cs >= nin_p, "convt2_bwd_data: bad stride");
    CtP p;
    p.in = gout; p.in_scale = nullptr; p.in_shift = nullptr; p.wp = wp; p.bias = nullptr; p.out = gin;
    p.in_cs = g_cs; p.rin_p = rout_p; p.in_relu = 0; p.out_cs = gin_cs; p.nout_p = nin_p; p.nbias = 0;
    p.N = N; p.D = D; p.H = H; p.W = W; p.nvox = (int64_t)N * D * H * W;
    return launch_convt<1>(p, (hipStream_t)stream, "convt2_bwd_data");
}

static void ct_wgrad_geom(int64_t nvox, int cin_p, int cout_p, int* mi, int* nj, int* n_ci_t, int* n_co_t, int* gx) {
    *mi = cin_p > 16 ? 2 : 1;
    *nj = cout_p > 16 ? 2 : 1;
    *n_ci_t = ceil_div(cin_p, 16 * *mi);
    *n_co_t = ceil_div(cout_p, 16 * *nj);
    *gx = ct_wgrad_gx((int)ceil_div64(nvox, 64), *n_ci_t * *n_co_t);
}

extern "C" size_t ctu_convt2_wgrad_ws_floats(int N, int D, int H, int W, int cin_p, int cout_p) {
    int mi, nj, nci, nco, gx;
    ct_wgrad_geom((int64_t)N * D * H * W, cin_p, cout_p, &mi, &nj, &nci, &nco, &gx);
    return (size_t)nci * nco * gx * 8 * mi * nj * 256 + (size_t)nco * gx * 16 * nj;
}

template <int MI, int NJ>
static int launch_ct_wgrad(CtWgP p, float* dw, float* dbias, int Ci, int Co, const int32_t* imap, int n_co_t, int gx,
                           hipStream_t st) {
    float* wsb = p.ws + (size_t)p.n_ci_t * n_co_t * gx * 8 * MI * NJ * 256;
    convt2_wgrad_kernel<MI, NJ><<<dim3(gx, p.n_ci_t * n_co_t), 256, 0, st>>>(p, dbias ? wsb : nullptr);
    CTU_CHECK_LAUNCH("convt2_wgrad");
    convt2_wgrad_reduce_kernel<MI, NJ><<<ceil_div(Ci * Co * 8 + Co, 64), 64 * RPARTS, 0, st>>>(p.ws, wsb, dw, dbias, Ci, Co, imap,
                                                                                           p.n_ci_t, gx);
    CTU_CHECK_LAUNCH("convt2_wgrad_reduce");
    return CTU_OK;
}

extern "C" int ctu_convt2_wgrad(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                int in_relu, const float* gout, int g_cs, int cout_p, float* dw, float* dbias, int Ci,
                                int Co, const int32_t* imap, float* ws, int N, int D, int H, int W, void* stream) {
    CTU_REQUIRE(in && gout && dw && ws, "convt2_wgrad: null pointer");
    CTU_REQUIRE(cin_p % 8 == 0 && cout_p % 8 == 0 && cin_p > 0 && cout_p > 0, "convt2_wgrad: padded channels");
    CTU_REQUIRE(in_cs >= cin_p && in_cs % 4 == 0 && g_cs >= cout_p && g_cs % 4 == 0, "convt2_wgrad: bad stride");
    hipStream_t st = (hipStream_t)stream;
    CtWgP p;
    p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.g = gout; p.ws = ws;
    p.in_cs = in_cs; p.cin_p = cin_p; p.in_relu = in_relu; p.g_cs = g_cs; p.cout_p = cout_p;
    p.N = N; p.D = D; p.H = H; p.W = W; p.nvox = (int64_t)N * D * H * W;
    p.ntiles = (int)ceil_div64(p.nvox, 64);
    CTU_REQUIRE(p.nvox < (1LL << 31), "convt2_wgrad: more than 2^31 coarse voxels");
    int mi, nj, n_co_t, gx;
    ct_wgrad_geom(p.nvox, cin_p, cout_p, &mi, &nj, &p.n_ci_t, &n_co_t, &gx);
    if (mi == 1 && nj == 1) return launch_ct_wgrad<1, 1>(p, dw, dbias, Ci, Co, imap, n_co_t, gx, st);
    if (mi == 1) return launch_ct_wgrad<1, 2>(p, dw, dbias, Ci, Co, imap, n_co_t, gx, st);
    if (nj == 1) return launch_ct_wgrad<2, 1>(p, dw, dbias, Ci, Co, imap, n_co_t, gx, st);
    return launch_ct_wgrad<2, 2>(p, dw, dbias, Ci, Co, imap, n_co_t, gx, st);
}
