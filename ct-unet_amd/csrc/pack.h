// Weight re-layout ("packing") element functions shared by conv3d.hip and convt.hip: one thread per PACKED element
// (gather), so every element of the packed buffer is written and no memset is needed.
#pragma once
#include "common.h"

namespace {

template <int KS>
struct Taps {
    static constexpr int TAPS = KS * KS * KS;
    static constexpr int STAPS = (KS == 3) ? 27 : KS * KS;   // taps whose weights sit in LDS at once
    static constexpr int NSTAGE = TAPS / STAPS;
};

// wp index for (chunk c, stage s, tap-in-stage ts, 16-wide output tile, kq, n, j)
template <int KS>
__device__ __forceinline__ size_t wp_index(int rp, int np, int t, int n16) {
    constexpr int STAPS = Taps<KS>::STAPS, NSTAGE = Taps<KS>::NSTAGE;
    const int c = rp >> 3, kq = (rp & 7) >> 1, j = rp & 1;
    const int nt = np >> 4, n = np & 15;
    const int s = t / STAPS, ts = t % STAPS;
    return ((((size_t)c * NSTAGE + s) * STAPS + ts) * n16 + nt) * 128 + kq * 32 + n * 2 + j;
}

// One thread per PACKED element (gather): layout [chunk][stage][tap][n16 tile][kq][n][j].
// cinv maps a padded input-channel position to its logical channel (-1 = padding, NULL = identity).
template <int KS>
__device__ __forceinline__ void pack_conv_w_elem(int idx, const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci,
                                                 const int32_t* __restrict__ cinv, int nchunk, int n16, int mode) {
    constexpr int TAPS = Taps<KS>::TAPS, STAPS = Taps<KS>::STAPS, NSTAGE = Taps<KS>::NSTAGE;
    if (idx >= nchunk * TAPS * n16 * 128) return;
    int r = idx;
    const int j = r & 1; r >>= 1;
    const int n = r & 15; r >>= 4;
    const int kq = r & 3; r >>= 2;
    const int nt = r % n16; r /= n16;
    const int ts = r % STAPS; r /= STAPS;
    const int s = r % NSTAGE; r /= NSTAGE;
    const int c = r;
    const int rp = c * 8 + kq * 2 + j, np = nt * 16 + n, t = s * STAPS + ts;
    float v = 0.f;
    if (mode == 0) {
        const int ci = cinv ? cinv[rp] : (rp < Ci ? rp : -1);
        if (ci >= 0 && np < Co) v = w[((size_t)np * Ci + ci) * TAPS + t];
    } else {
        const int ci = cinv ? cinv[np] : (np < Ci ? np : -1);
        if (ci >= 0 && rp < Co) v = w[((size_t)rp * Ci + ci) * TAPS + (TAPS - 1 - t)];
    }
    wp[idx] = v;
}

// pair layout packing: [chunk][kd][kh][kw' (KS + 1)][kq][n = s*8+o][j]; gather, every element written
template <int KS>
__device__ __forceinline__ void pack_conv_w_pair8_elem(int idx, const float* __restrict__ w, float* __restrict__ wp, int Co,
                                                       int Ci, const int32_t* __restrict__ cinv, int nchunk, int mode) {
    constexpr int KWN = KS + 1, TAPS = KS * KS * KS;
    if (idx >= nchunk * KS * KS * KWN * 128) return;
    int r = idx;
    const int j = r & 1; r >>= 1;
    const int n = r & 15; r >>= 4;
    const int kq = r & 3; r >>= 2;
    const int kwp = r % KWN; r /= KWN;
    const int kh = r % KS; r /= KS;
    const int kd = r % KS; r /= KS;
    const int c = r;
    const int s = n >> 3, o = n & 7, kw = kwp - s;
    const int rp = c * 8 + kq * 2 + j;
    float v = 0.f;
    if (kw >= 0 && kw < KS) {
        const int t = (kd * KS + kh) * KS + kw;
        if (mode == 0) {
            const int ci = cinv ? cinv[rp] : (rp < Ci ? rp : -1);
            if (ci >= 0 && o < Co) v = w[((size_t)o * Ci + ci) * TAPS + t];
        } else {
            const int ci = cinv ? cinv[o] : (o < Ci ? o : -1);
            if (ci >= 0 && rp < Co) v = w[((size_t)rp * Ci + ci) * TAPS + (TAPS - 1 - t)];
        }
    }
    wp[idx] = v;
}

// ConvTranspose3d: wp[tap][g][nt][kq][n][j]
__device__ __forceinline__ void pack_convt_w_elem(int idx, const float* w, float* wp, int Ci, int Co, const int32_t* cinv, int rin_p, int NTT,
                                  int mode) {
    const int ng = rin_p >> 3;
    if (idx >= 8 * ng * NTT * 128) return;
    int r = idx;
    const int j = r & 1; r >>= 1;
    const int n = r & 15; r >>= 4;
    const int kq = r & 3; r >>= 2;
    const int nt = r % NTT; r /= NTT;
    const int g = r % ng; r /= ng;
    const int tap = r;
    const int rp = g * 8 + kq * 2 + j, np = nt * 16 + n;
    float v = 0.f;
    if (mode == 0) {
        const int ci = cinv ? cinv[rp] : (rp < Ci ? rp : -1);
        if (ci >= 0 && np < Co) v = w[((size_t)ci * Co + np) * 8 + tap];
    } else {
        const int ci = cinv ? cinv[np] : (np < Ci ? np : -1);
        if (ci >= 0 && rp < Co) v = w[((size_t)ci * Co + rp) * 8 + tap];
    }
    wp[idx] = v;
}


}  // namespace
