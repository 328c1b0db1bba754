// 1x1x1 output head (+softmax/sigmoid/SP re-encoding, NCDHW out) and the fused Dice +
// cross-entropy loss for gfx950.  HBM-bound streaming kernels, one voxel per thread; reductions
// are two-stage and deterministic.
//
// Replaces: nn.Conv3d(lc_in, out, 1) + F.softmax / torch.sigmoid   ctunet/pytorch/models.py:223-224,255-259,507,538
//           UNetSP/UNetDO/UNetSPSmall re-encoding                   ctunet/pytorch/models.py:317-330,351-365,374-387
//           dice_loss                                               ctunet/utilities.py:35-50
//           softmax + nn.CrossEntropyLoss + argmax loss assembly    ctunet/pytorch/ProblemHandler.py:59-88,228-298
#include "common.h"
#include "bn_tail.h"

namespace {

constexpr int HB = 256;
constexpr int MAXCO = 4;

struct HeadP {
    const void* in;          // channels-last activations, element type T of the kernel template
    const float* in_scale;
    const float* in_shift;
    const float* w;
    const float* bias;
    const int32_t* imap;
    int in_cs, in_relu, Ci, Co, act, head_mode, N;
    int64_t V;
    float gscale;            // backward: the incoming output gradients are multiplied by this (float16 loss scale), 1 otherwise
};

// activated input channels of one voxel -> logits -> y (post softmax/sigmoid)
template <int CP, class T>
__device__ __forceinline__ void head_point(const HeadP& p, const float* sWp, const float* sB, int64_t gv,
                                           float (&a)[CP], float (&lg)[MAXCO], float (&u)[MAXCO],
                                           float (&y)[MAXCO]) {
    const T* src = reinterpret_cast<const T*>(p.in) + (size_t)gv * p.in_cs;
#pragma unroll
    for (int qd = 0; qd < CP / 4; ++qd) {
        float4 v = ld4<T>(src + qd * 4);
        if (p.in_scale) {
            const float4 sc = *reinterpret_cast<const float4*>(p.in_scale + qd * 4);
            const float4 sh = *reinterpret_cast<const float4*>(p.in_shift + qd * 4);
            v = xform4(v, sc, sh, p.in_relu);
        }
        a[qd * 4 + 0] = v.x; a[qd * 4 + 1] = v.y; a[qd * 4 + 2] = v.z; a[qd * 4 + 3] = v.w;
    }
#pragma unroll
    for (int co = 0; co < MAXCO; ++co) {
        float s = 0.f;
        if (co < p.Co) {
            // accumulate in logical channel order like a dot product over Ci
#pragma unroll
            for (int c = 0; c < CP; ++c) s = fmaf(a[c], sWp[co * CP + c], s);
            s += sB[co];
        }
        lg[co] = s;
    }
    if (p.act & 1) {
        float mx = -INFINITY;
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) if (co < p.Co) mx = fmaxf(mx, lg[co]);
        float den = 0.f;
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) { u[co] = (co < p.Co) ? expf(lg[co] - mx) : 0.f; den += u[co]; }
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) u[co] /= den;
    } else {
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) u[co] = lg[co];
    }
#pragma unroll
    for (int co = 0; co < MAXCO; ++co) y[co] = (p.act & 2) ? 1.f / (1.f + expf(-u[co])) : u[co];
}

template <int CP>
__device__ __forceinline__ void head_load_weights(const HeadP& p, float* sWp, float* sB) {
    for (int i = threadIdx.x; i < MAXCO * CP; i += blockDim.x) sWp[i] = 0.f;
    if (threadIdx.x < MAXCO) sB[threadIdx.x] = (threadIdx.x < p.Co && p.bias) ? p.bias[threadIdx.x] : 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < p.Co * p.Ci; i += blockDim.x) {
        const int co = i / p.Ci, ci = i % p.Ci;
        sWp[co * CP + (p.imap ? p.imap[ci] : ci)] = p.w[i];
    }
    __syncthreads();
}

__device__ __forceinline__ void softmax2(float a, float b, float& sa, float& sb) {
    const float mx = fmaxf(a, b);
    const float ea = expf(a - mx), eb = expf(b - mx);
    sa = ea / (ea + eb); sb = eb / (ea + eb);
}

template <int CP, class T>
__global__ __launch_bounds__(HB) void head_fwd_kernel(HeadP p, float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float sWp[MAXCO * CP];
    __shared__ float sB[MAXCO];
    head_load_weights<CP>(p, sWp, sB);
    const int64_t total = (int64_t)p.N * p.V;
    for (int64_t gv = (int64_t)blockIdx.x * HB + threadIdx.x; gv < total; gv += (int64_t)gridDim.x * HB) {
        float a[CP], lg[MAXCO], u[MAXCO], y[MAXCO];
        head_point<CP, T>(p, sWp, sB, gv, a, lg, u, y);
        const int64_t n = gv / p.V, v = gv % p.V;
        if (p.head_mode == 0) {
#pragma unroll
            for (int co = 0; co < MAXCO; ++co)
                if (co < p.Co) out0[(n * p.Co + co) * p.V + v] = y[co];
        } else {
            float s0 = y[0], s1 = y[1] + y[2], f0 = 1.f - y[1], f1 = y[1];
            if (p.head_mode == 2) {
                softmax2(s0, s1, s0, s1);
                softmax2(f0, f1, f0, f1);
            }
            out0[(n * 2 + 0) * p.V + v] = s0; out0[(n * 2 + 1) * p.V + v] = s1;
            out1[(n * 2 + 0) * p.V + v] = f0; out1[(n * 2 + 1) * p.V + v] = f1;
        }
    }
}

// partial layout per block: [MAXCO*CP dW][MAXCO db]
// Backward with CP/4 lanes per voxel: every lane owns one channel quad (16-byte coalesced loads/stores, ~50 VGPRs so
// the streaming loop is hidden by occupancy instead of being latency-bound at one voxel per 64-accumulator thread).
// Logits are quad partial sums reduced across the voxel's lanes; block partials keep the layout above.
template <int CP, class T>
// bn_partials != NULL: the first bn_cp input channels are relu(BN(raw conv output)) of ONE layer whose activated-output
// gradient is complete with this kernel's gin (the decoder's last conv feeding the head): the kernel also emits that
// BatchNorm's backward reduction rows {sum gz, sum gz * xhat} per block (layout of bn_relu_bwd_reduce).
__global__ __launch_bounds__(HB) void head_bwd_q_kernel(HeadP p, const float* __restrict__ g0,
                                                        const float* __restrict__ g1, T* __restrict__ gin,
                                                        int gin_cs, float* __restrict__ partials,
                                                        const float* __restrict__ bn_mean, const float* __restrict__ bn_invstd,
                                                        int bn_cp, float* __restrict__ bn_partials) {
    constexpr int Q = CP / 4;
    __shared__ float sWp[MAXCO * CP];
    __shared__ float sB[MAXCO];
    __shared__ float sRed[HB / 64][MAXCO * CP + MAXCO + 2 * CP];
    head_load_weights<CP>(p, sWp, sB);
    const int qd = threadIdx.x % Q;
    float wq[MAXCO][4], dw[MAXCO][4], db[MAXCO];
#pragma unroll
    for (int co = 0; co < MAXCO; ++co) {
        db[co] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { wq[co][j] = sWp[co * CP + qd * 4 + j]; dw[co][j] = 0.f; }
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.in_scale) {
        sc = *reinterpret_cast<const float4*>(p.in_scale + qd * 4);
        sh = *reinterpret_cast<const float4*>(p.in_shift + qd * 4);
    }
    const int xrelu = p.in_scale ? p.in_relu : 0;
    const bool bnq = bn_partials != nullptr && qd * 4 < bn_cp;
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, r1 = mu, r2 = mu;
    if (bnq) {
        mu = *reinterpret_cast<const float4*>(bn_mean + qd * 4);
        is = *reinterpret_cast<const float4*>(bn_invstd + qd * 4);
    }
    const int64_t total = (int64_t)p.N * p.V;
    constexpr int VPB = HB / Q;                       // voxels per block and iteration
    for (int64_t gv = (int64_t)blockIdx.x * VPB + threadIdx.x / Q; gv < total; gv += (int64_t)gridDim.x * VPB) {
        const float4 raw = ld4<T>(reinterpret_cast<const T*>(p.in) + (size_t)gv * p.in_cs + qd * 4);
        const float4 a = xform4(raw, sc, sh, xrelu);
        float lg[MAXCO], u[MAXCO], y[MAXCO], gy[MAXCO];
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) {
            float s = 0.f;
            if (co < p.Co) {
                s = fmaf(a.w, wq[co][3], fmaf(a.z, wq[co][2], fmaf(a.y, wq[co][1], a.x * wq[co][0])));
#pragma unroll
                for (int o = 1; o < Q; o <<= 1) s += __shfl_xor(s, o);
                s += sB[co];
            }
            lg[co] = s;
        }
        if (p.act & 1) {
            float mx = -INFINITY;
#pragma unroll
            for (int co = 0; co < MAXCO; ++co) if (co < p.Co) mx = fmaxf(mx, lg[co]);
            float den = 0.f;
#pragma unroll
            for (int co = 0; co < MAXCO; ++co) { u[co] = (co < p.Co) ? expf(lg[co] - mx) : 0.f; den += u[co]; }
#pragma unroll
            for (int co = 0; co < MAXCO; ++co) u[co] /= den;
        } else {
#pragma unroll
            for (int co = 0; co < MAXCO; ++co) u[co] = lg[co];
        }
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) y[co] = (p.act & 2) ? 1.f / (1.f + expf(-u[co])) : u[co];
        const int64_t n = gv / p.V, v = gv % p.V;
        if (p.head_mode == 0) {
#pragma unroll
            for (int co = 0; co < MAXCO; ++co) gy[co] = (co < p.Co) ? g0[(n * p.Co + co) * p.V + v] : 0.f;
        } else {
            float a0 = g0[(n * 2 + 0) * p.V + v], a1 = g0[(n * 2 + 1) * p.V + v];
            float b0 = g1[(n * 2 + 0) * p.V + v], b1 = g1[(n * 2 + 1) * p.V + v];
            if (p.head_mode == 2) {
                float s0, s1, f0, f1;
                softmax2(y[0], y[1] + y[2], s0, s1);
                softmax2(1.f - y[1], y[1], f0, f1);
                const float ds = a0 * s0 + a1 * s1, df = b0 * f0 + b1 * f1;
                a0 = s0 * (a0 - ds); a1 = s1 * (a1 - ds);
                b0 = f0 * (b0 - df); b1 = f1 * (b1 - df);
            }
            gy[0] = a0; gy[1] = a1 - b0 + b1; gy[2] = a1; gy[3] = 0.f;
        }
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) gy[co] *= p.gscale;      // (everything downstream is linear in the output gradients)
        float gu[MAXCO], gl[MAXCO];
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) gu[co] = (p.act & 2) ? gy[co] * y[co] * (1.f - y[co]) : gy[co];
        if (p.act & 1) {
            float dot = 0.f;
#pragma unroll
            for (int co = 0; co < MAXCO; ++co) if (co < p.Co) dot += gu[co] * u[co];
#pragma unroll
            for (int co = 0; co < MAXCO; ++co) gl[co] = (co < p.Co) ? u[co] * (gu[co] - dot) : 0.f;
        } else {
#pragma unroll
            for (int co = 0; co < MAXCO; ++co) gl[co] = (co < p.Co) ? gu[co] : 0.f;
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) {
            o.x = fmaf(gl[co], wq[co][0], o.x); o.y = fmaf(gl[co], wq[co][1], o.y);
            o.z = fmaf(gl[co], wq[co][2], o.z); o.w = fmaf(gl[co], wq[co][3], o.w);
            if (qd == 0) db[co] += gl[co];
#pragma unroll
            for (int j = 0; j < 4; ++j) dw[co][j] = fmaf(gl[co], av[j], dw[co][j]);
        }
        o = rnd4<T>(o);                      // the reduction sees what the BatchNorm-backward apply pass will read back
        st4<T>(gin + (size_t)gv * gin_cs + qd * 4, o);
        if (bnq) {
            float gz;
            gz = (fmaf(raw.x, sc.x, sh.x) > 0.f) ? o.x : 0.f; r1.x += gz; r2.x += gz * (raw.x - mu.x) * is.x;
            gz = (fmaf(raw.y, sc.y, sh.y) > 0.f) ? o.y : 0.f; r1.y += gz; r2.y += gz * (raw.y - mu.y) * is.y;
            gz = (fmaf(raw.z, sc.z, sh.z) > 0.f) ? o.z : 0.f; r1.z += gz; r2.z += gz * (raw.z - mu.z) * is.z;
            gz = (fmaf(raw.w, sc.w, sh.w) > 0.f) ? o.w : 0.f; r1.w += gz; r2.w += gz * (raw.w - mu.w) * is.w;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (bn_partials) {
        const float rv[8] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = rv[j];
#pragma unroll
            for (int o = 32; o >= Q; o >>= 1) s += __shfl_xor(s, o);        // over the lanes that own this quad
            if (lane < Q) sRed[wave][MAXCO * CP + MAXCO + (j >> 2) * CP + lane * 4 + (j & 3)] = s;
        }
    }
#pragma unroll
    for (int co = 0; co < MAXCO; ++co) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = dw[co][j];
#pragma unroll
            for (int o = 32; o >= Q; o >>= 1) s += __shfl_xor(s, o);        // over the lanes that own this quad
            if (lane < Q) sRed[wave][co * CP + lane * 4 + j] = s;
        }
        const float s = wave_sum(db[co]);
        if (lane == 0) sRed[wave][MAXCO * CP + co] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MAXCO * CP + MAXCO; i += HB) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < HB / 64; ++w) s += sRed[w][i];
        partials[(size_t)blockIdx.x * (MAXCO * CP + MAXCO) + i] = s;
    }
    if (bn_partials)
        for (int i = threadIdx.x; i < 2 * bn_cp; i += HB) {
            const int col = MAXCO * CP + MAXCO + (i / bn_cp) * CP + i % bn_cp;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < HB / 64; ++w) s += sRed[w][col];
            bn_partials[(size_t)blockIdx.x * 2 * bn_cp + i] = s;
        }
}

// one block per output (dW element or db element); fixed-order tree over the block partials
template <int CP>
__global__ __launch_bounds__(HB) void head_bwd_final_kernel(const float* __restrict__ partials, int nb, int Ci, int Co,
                                                            const int32_t* __restrict__ imap, float* __restrict__ dw,
                                                            float* __restrict__ db, const float* __restrict__ bn_partials,
                                                            int bn_cp, ctu_bn_bwd_tail tail) {
    const int i = blockIdx.x;
    if (i == Co * Ci + Co) {
        // one extra block: the BatchNorm-backward finalize of the layer feeding the head (rows written by the kernel before
        // this one, so plain visibility; same arithmetic as ctu_bn_bwd_finalize)
        bn_bwd_finalize_rows(tail, bn_partials, nb, bn_cp);
        return;
    }
    const int row = MAXCO * CP + MAXCO;
    int col;
    if (i < Co * Ci) {
        const int co = i / Ci, ci = i % Ci;
        col = co * CP + (imap ? imap[ci] : ci);
    } else {
        col = MAXCO * CP + (i - Co * Ci);
    }
    __shared__ double r[HB];
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += HB) s += (double)partials[(size_t)b * row + col];
    r[threadIdx.x] = s;
    __syncthreads();
    for (int o = HB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) r[threadIdx.x] += r[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (i < Co * Ci) dw[i] = (float)r[0];
        else db[i - Co * Ci] = (float)r[0];
    }
}

inline int head_blocks(int64_t total) {
    int64_t nb = ceil_div64(total, HB);
    if (nb > 2048) nb = 2048;
    return (int)(nb < 1 ? 1 : nb);
}

// ------------------------------------------------------------------ loss
constexpr int LOSS_BX = 1024;  // blocks per batch item (4 per CU: 16 B x 4 streams per lane in flight; 256 blocks of scalar loads
                               // ran at 1.6 TB/s -- 0.7 ms of the two-head 256^3 step)

__global__ __launch_bounds__(HB) void loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                      int64_t V, int dice_softmax, float* __restrict__ ws) {
    const int n = blockIdx.y;
    const float* p0 = pred + (size_t)n * 2 * V;
    const float* p1 = p0 + V;
    const float* t0 = tgt + (size_t)n * 2 * V;
    const float* t1 = t0 + V;
    float ce = 0.f, num = 0.f, d1 = 0.f, d2 = 0.f;
    auto point = [&](float a, float b, float ta, float tb) {
        const float mx = fmaxf(a, b);
        const float ea = expf(a - mx), eb = expf(b - mx);
        const float lse = mx + logf(ea + eb);
        ce += lse - ((tb > ta) ? b : a);          // argmax: first index wins ties
        float pa = a, pb = b;
        if (dice_softmax) { pa = ea / (ea + eb); pb = eb / (ea + eb); }
        num += pa * ta + pb * tb;
        d1 += pa * pa + pb * pb;
        d2 += ta * ta + tb * tb;
    };
    // four voxels per lane and step (16-byte loads of the four planes) where the planes are 16-byte aligned; scalar otherwise
    const bool vec = (V & 3) == 0 && ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(tgt)) & 15) == 0;
    if (vec) {
        const int64_t V4 = V >> 2;
        for (int64_t q = (int64_t)blockIdx.x * HB + threadIdx.x; q < V4; q += (int64_t)gridDim.x * HB) {
            const float4 a = reinterpret_cast<const float4*>(p0)[q], b = reinterpret_cast<const float4*>(p1)[q];
            const float4 ta = reinterpret_cast<const float4*>(t0)[q], tb = reinterpret_cast<const float4*>(t1)[q];
            point(a.x, b.x, ta.x, tb.x); point(a.y, b.y, ta.y, tb.y); point(a.z, b.z, ta.z, tb.z); point(a.w, b.w, ta.w, tb.w);
        }
    } else {
        for (int64_t v = (int64_t)blockIdx.x * HB + threadIdx.x; v < V; v += (int64_t)gridDim.x * HB) point(p0[v], p1[v], t0[v], t1[v]);
    }
    __shared__ float red[HB / 64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ce = wave_sum(ce); num = wave_sum(num); d1 = wave_sum(d1); d2 = wave_sum(d2);
    if (lane == 0) { red[wave][0] = ce; red[wave][1] = num; red[wave][2] = d1; red[wave][3] = d2; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < HB / 64; ++w) s += red[w][threadIdx.x];
        ws[((size_t)n * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = s;
    }
}

// ws tail: [N][2] = (num+eps, den+eps) per item, used by backward
// one wave: lane l sums the block partials l, l + 64, ... in double, then a fixed-order butterfly -> deterministic
__global__ void loss_final_kernel(float* __restrict__ ws, int N, int nbx, int64_t V, float ce_lambda,
                                  float dice_lambda, float* __restrict__ terms) {
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    const double eps = 0.0000001;
    double ce = 0.0, dsum = 0.0;
    float* tail = ws + (size_t)N * nbx * 4;
    for (int n = 0; n < N; ++n) {
        double c = 0.0, num = 0.0, d1 = 0.0, d2 = 0.0;
        for (int b = lane; b < nbx; b += 64) {
            const float4 r = *reinterpret_cast<const float4*>(ws + ((size_t)n * nbx + b) * 4);
            c += r.x; num += r.y; d1 += r.z; d2 += r.w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c += __shfl_xor(c, o); num += __shfl_xor(num, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o);
        }
        ce += c;
        dsum += (num + eps) / (d1 + d2 + eps);
        if (lane == 0) {
            tail[n * 2 + 0] = (float)(num + eps);
            tail[n * 2 + 1] = (float)(d1 + d2 + eps);
        }
    }
    if (lane == 0) {
        terms[0] = ce_lambda != 0.f ? (float)(ce_lambda * ce / ((double)N * (double)V)) : 0.f;
        terms[1] = dice_lambda != 0.f ? (float)(dice_lambda * (1.0 - 2.0 * dsum / N)) : 0.f;
    }
}

__global__ __launch_bounds__(HB) void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                      int N, int64_t V, float ce_lambda, float dice_lambda,
                                                      int dice_softmax, const float* __restrict__ tail,
                                                      const float* __restrict__ gscale_ce,
                                                      const float* __restrict__ gscale_dice, float* __restrict__ gpred,
                                                      int accumulate) {
    const int n = blockIdx.y;
    const float gs_ce = gscale_ce ? gscale_ce[0] : 1.f, gs_d = gscale_dice ? gscale_dice[0] : 1.f;
    const float kce = gs_ce * ce_lambda / ((float)N * (float)V);
    const float nume = tail[n * 2 + 0], dene = tail[n * 2 + 1];
    const float kd = gs_d * dice_lambda * (-2.f / (float)N);
    const float kd_t = kd / dene, kd_p = kd * nume * 2.f / (dene * dene);
    const size_t base = (size_t)n * 2 * V;
    auto point = [&](float a, float b, float ta, float tb, float& ga, float& gb) {
        const float mx = fmaxf(a, b);
        const float ea = expf(a - mx), eb = expf(b - mx);
        const float sa = ea / (ea + eb), sb = eb / (ea + eb);
        const bool cls1 = tb > ta;
        ga = kce * (sa - (cls1 ? 0.f : 1.f));
        gb = kce * (sb - (cls1 ? 1.f : 0.f));
        if (dice_lambda != 0.f) {
            const float pa = dice_softmax ? sa : a, pb = dice_softmax ? sb : b;
            float da = kd_t * ta - kd_p * pa, db = kd_t * tb - kd_p * pb;
            if (dice_softmax) {
                const float dot = da * sa + db * sb;
                da = sa * (da - dot); db = sb * (db - dot);
            }
            ga += da; gb += db;
        }
    };
    const bool vec = (V & 3) == 0 && ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(tgt) | reinterpret_cast<uintptr_t>(gpred)) & 15) == 0;
    if (vec) {
        const int64_t V4 = V >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(pred + base);
        const float4* b4 = reinterpret_cast<const float4*>(pred + base + V);
        const float4* ta4 = reinterpret_cast<const float4*>(tgt + base);
        const float4* tb4 = reinterpret_cast<const float4*>(tgt + base + V);
        float4* ga4 = reinterpret_cast<float4*>(gpred + base);
        float4* gb4 = reinterpret_cast<float4*>(gpred + base + V);
        for (int64_t q = (int64_t)blockIdx.x * HB + threadIdx.x; q < V4; q += (int64_t)gridDim.x * HB) {
            const float4 a = a4[q], b = b4[q], ta = ta4[q], tb = tb4[q];
            float4 ga, gb;
            point(a.x, b.x, ta.x, tb.x, ga.x, gb.x); point(a.y, b.y, ta.y, tb.y, ga.y, gb.y);
            point(a.z, b.z, ta.z, tb.z, ga.z, gb.z); point(a.w, b.w, ta.w, tb.w, ga.w, gb.w);
            if (accumulate) {
                const float4 oa = ga4[q], ob = gb4[q];
                ga.x += oa.x; ga.y += oa.y; ga.z += oa.z; ga.w += oa.w; gb.x += ob.x; gb.y += ob.y; gb.z += ob.z; gb.w += ob.w;
            }
            ga4[q] = ga; gb4[q] = gb;
        }
        return;
    }
    for (int64_t v = (int64_t)blockIdx.x * HB + threadIdx.x; v < V; v += (int64_t)gridDim.x * HB) {
        float ga, gb;
        point(pred[base + v], pred[base + V + v], tgt[base + v], tgt[base + V + v], ga, gb);
        if (accumulate) { ga += gpred[base + v]; gb += gpred[base + V + v]; }
        gpred[base + v] = ga;
        gpred[base + V + v] = gb;
    }
}

}  // namespace

// =================================================================== C ABI
static int fill_head(HeadP& p, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                     int in_relu, const float* w, const float* bias, const int32_t* imap, int Ci, int Co, int act,
                     int head_mode, int N, int64_t V, const char* name) {
    CTU_REQUIRE(in && w, "%s: null pointer", name);
    CTU_REQUIRE(cin_p == 8 || cin_p == 16 || cin_p == 32, "%s: cin_p=%d unsupported (8, 16 or 32)", name, cin_p);
    CTU_REQUIRE(Co >= 1 && Co <= MAXCO && Ci >= 1 && Ci <= cin_p, "%s: Co=%d Ci=%d", name, Co, Ci);
    CTU_REQUIRE(head_mode == 0 || Co == 3, "%s: SP re-encoding needs 3 output channels", name);
    CTU_REQUIRE(head_mode >= 0 && head_mode <= 2, "%s: head_mode=%d", name, head_mode);
    CTU_REQUIRE(in_cs >= cin_p && in_cs % 4 == 0, "%s: bad stride", name);
    CTU_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "%s: scale/shift must come together", name);
    p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.w = w; p.bias = bias; p.imap = imap;
    p.in_cs = in_cs; p.in_relu = in_relu; p.Ci = Ci; p.Co = Co; p.act = act; p.head_mode = head_mode; p.N = N; p.V = V; p.gscale = 1.f;
    return CTU_OK;
}

namespace {

template <class T>
int head_fwd_impl(const T* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift, int in_relu, const float* w,
                  const float* bias, const int32_t* imap, int Ci, int Co, int act, int head_mode, float* out0, float* out1, int N,
                  int64_t nvox_per_item, void* stream) {
    HeadP p;
    int rc = fill_head(p, in, in_cs, cin_p, in_scale, in_shift, in_relu, w, bias, imap, Ci, Co, act, head_mode, N,
                       nvox_per_item, "head_fwd");
    if (rc != CTU_OK) return rc;
    CTU_REQUIRE(out0 && (head_mode == 0 || out1), "head_fwd: null output");
    const int nb = head_blocks((int64_t)N * nvox_per_item);
    hipStream_t st = (hipStream_t)stream;
    if (cin_p == 8) head_fwd_kernel<8, T><<<nb, HB, 0, st>>>(p, out0, out1);
    else if (cin_p == 16) head_fwd_kernel<16, T><<<nb, HB, 0, st>>>(p, out0, out1);
    else head_fwd_kernel<32, T><<<nb, HB, 0, st>>>(p, out0, out1);
    CTU_CHECK_LAUNCH("head_fwd");
    return CTU_OK;
}

template <class T>
int head_bwd_impl(const T* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift, int in_relu, const float* w,
                  const float* bias, const int32_t* imap, int Ci, int Co, int act, int head_mode, const float* g0, const float* g1,
                  T* gin, int gin_cs, float* dw, float* db, float* ws, int N, int64_t nvox_per_item, const float* bn_mean,
                  const float* bn_invstd, int bn_cp, float* bn_partials, const ctu_bn_bwd_tail* tail, void* stream, float gscale = 1.f) {
    CTU_REQUIRE(!tail || (bn_partials && tail->gamma && tail->invstd && tail->dgamma && tail->dbeta && tail->coef &&
                          tail->C > 0 && tail->C <= bn_cp && tail->count > 0 && (!tail->running_mean || (tail->mean && tail->running_var))),
                "head_bwd: incomplete BatchNorm tail");
    HeadP p;
    int rc = fill_head(p, in, in_cs, cin_p, in_scale, in_shift, in_relu, w, bias, imap, Ci, Co, act, head_mode, N,
                       nvox_per_item, "head_bwd");
    if (rc != CTU_OK) return rc;
    p.gscale = gscale;
    CTU_REQUIRE(g0 && (head_mode == 0 || g1) && gin && dw && db && ws, "head_bwd: null pointer");
    CTU_REQUIRE(gin_cs >= cin_p && gin_cs % 4 == 0, "head_bwd: bad gin stride");
    CTU_REQUIRE(!bn_partials || (bn_mean && bn_invstd && in_scale && in_relu && bn_cp > 0 && bn_cp % 4 == 0 && bn_cp <= cin_p),
                "head_bwd: the BatchNorm reduction needs mean/invstd, a BN+ReLU input transform and bn_cp <= cin_p (bn_cp=%d)", bn_cp);
    const int nb = head_blocks((int64_t)N * nvox_per_item);
    hipStream_t st = (hipStream_t)stream;
    const int nfin = Co * Ci + Co + (tail ? 1 : 0);      // + the BatchNorm finalize block
    const ctu_bn_bwd_tail tl = bwd_tail_or_off(tail);
    if (cin_p == 8) {
        head_bwd_q_kernel<8, T><<<nb, HB, 0, st>>>(p, g0, g1, gin, gin_cs, ws, bn_mean, bn_invstd, bn_cp, bn_partials);
        head_bwd_final_kernel<8><<<nfin, HB, 0, st>>>(ws, nb, Ci, Co, imap, dw, db, bn_partials, bn_cp, tl);
    } else if (cin_p == 16) {
        head_bwd_q_kernel<16, T><<<nb, HB, 0, st>>>(p, g0, g1, gin, gin_cs, ws, bn_mean, bn_invstd, bn_cp, bn_partials);
        head_bwd_final_kernel<16><<<nfin, HB, 0, st>>>(ws, nb, Ci, Co, imap, dw, db, bn_partials, bn_cp, tl);
    } else {
        head_bwd_q_kernel<32, T><<<nb, HB, 0, st>>>(p, g0, g1, gin, gin_cs, ws, bn_mean, bn_invstd, bn_cp, bn_partials);
        head_bwd_final_kernel<32><<<nfin, HB, 0, st>>>(ws, nb, Ci, Co, imap, dw, db, bn_partials, bn_cp, tl);
    }
    CTU_CHECK_LAUNCH("head_bwd");
    return CTU_OK;
}

}  // namespace

extern "C" int ctu_head_fwd(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                            int in_relu, const float* w, const float* bias, const int32_t* imap, int Ci, int Co, int act,
                            int head_mode, float* out0, float* out1, int N, int64_t nvox_per_item, void* stream) {
    return head_fwd_impl<float>(in, in_cs, cin_p, in_scale, in_shift, in_relu, w, bias, imap, Ci, Co, act, head_mode, out0, out1, N,
                                nvox_per_item, stream);
}
extern "C" int ctu_lp_head_fwd(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                               int in_relu, const float* w, const float* bias, const int32_t* imap, int Ci, int Co, int act,
                               int head_mode, float* out0, float* out1, int N, int64_t nvox_per_item, void* stream) {
    CTU_DISPATCH_LP(dtype, return head_fwd_impl<T>((const T*)in, in_cs, cin_p, in_scale, in_shift, in_relu, w, bias, imap, Ci, Co, act,
                                                   head_mode, out0, out1, N, nvox_per_item, stream));
}

extern "C" size_t ctu_head_bwd_ws_floats(int N, int64_t nvox_per_item, int cin_p, int Co) {
    (void)Co;
    return (size_t)head_blocks((int64_t)N * nvox_per_item) * (MAXCO * cin_p + MAXCO);
}

extern "C" int ctu_head_bwd_num_blocks(int N, int64_t nvox_per_item) { return head_blocks((int64_t)N * nvox_per_item); }

extern "C" int ctu_head_bwd(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                            int in_relu, const float* w, const float* bias, const int32_t* imap, int Ci, int Co, int act,
                            int head_mode, const float* g0, const float* g1, float* gin, int gin_cs, float* dw,
                            float* db, float* ws, int N, int64_t nvox_per_item, void* stream) {
    return head_bwd_impl<float>(in, in_cs, cin_p, in_scale, in_shift, in_relu, w, bias, imap, Ci, Co, act, head_mode, g0, g1, gin,
                                gin_cs, dw, db, ws, N, nvox_per_item, nullptr, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int ctu_head_bwd_bn(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                               int in_relu, const float* w, const float* bias, const int32_t* imap, int Ci, int Co, int act,
                               int head_mode, const float* g0, const float* g1, float* gin, int gin_cs, float* dw,
                               float* db, float* ws, int N, int64_t nvox_per_item, const float* bn_mean,
                               const float* bn_invstd, int bn_cp, float* bn_partials, const ctu_bn_bwd_tail* tail, void* stream) {
    return head_bwd_impl<float>(in, in_cs, cin_p, in_scale, in_shift, in_relu, w, bias, imap, Ci, Co, act, head_mode, g0, g1, gin,
                                gin_cs, dw, db, ws, N, nvox_per_item, bn_mean, bn_invstd, bn_cp, bn_partials, tail, stream);
}
/* bn_partials NULL: plain head backward */
extern "C" int ctu_lp_head_bwd_bn(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                  int in_relu, const float* w, const float* bias, const int32_t* imap, int Ci, int Co, int act,
                                  int head_mode, const float* g0, const float* g1, void* gin, int gin_cs, float* dw,
                                  float* db, float* ws, int N, int64_t nvox_per_item, const float* bn_mean,
                                  const float* bn_invstd, int bn_cp, float* bn_partials, const ctu_bn_bwd_tail* tail, float gscale,
                                  void* stream) {
    CTU_DISPATCH_LP(dtype, return head_bwd_impl<T>((const T*)in, in_cs, cin_p, in_scale, in_shift, in_relu, w, bias, imap, Ci, Co, act,
                                                   head_mode, g0, g1, (T*)gin, gin_cs, dw, db, ws, N, nvox_per_item, bn_mean,
                                                   bn_invstd, bn_cp, bn_partials, tail, stream, gscale));
}

extern "C" size_t ctu_loss_ws_floats(int N, int64_t V) {
    (void)V;
    return (size_t)N * LOSS_BX * 4 + (size_t)N * 2;
}

extern "C" int ctu_loss_fwd(const float* pred, const float* target, int N, int64_t V, float ce_lambda,
                            float dice_lambda, int dice_softmax, float* terms, float* ws, void* stream) {
    CTU_REQUIRE(pred && target && terms && ws && N > 0 && V > 0, "loss_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    loss_fwd_kernel<<<dim3(LOSS_BX, N), HB, 0, st>>>(pred, target, V, dice_softmax, ws);
    CTU_CHECK_LAUNCH("loss_fwd");
    loss_final_kernel<<<1, 64, 0, st>>>(ws, N, LOSS_BX, V, ce_lambda, dice_lambda, terms);
    CTU_CHECK_LAUNCH("loss_final");
    return CTU_OK;
}

extern "C" int ctu_loss_bwd(const float* pred, const float* target, int N, int64_t V, float ce_lambda,
                            float dice_lambda, int dice_softmax, const float* ws, const float* gscale_ce,
                            const float* gscale_dice, float* gpred, int accumulate, void* stream) {
    CTU_REQUIRE(pred && target && ws && gpred && N > 0 && V > 0, "loss_bwd: bad argument");
    loss_bwd_kernel<<<dim3(LOSS_BX, N), HB, 0, (hipStream_t)stream>>>(pred, target, N, V, ce_lambda, dice_lambda,
                                                                      dice_softmax, ws + (size_t)N * LOSS_BX * 4,
                                                                      gscale_ce, gscale_dice, gpred, accumulate);
    CTU_CHECK_LAUNCH("loss_bwd");
    return CTU_OK;
}
