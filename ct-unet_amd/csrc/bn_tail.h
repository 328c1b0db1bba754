// In-launch BatchNorm finalize ("tail"): the block of a stats-producing launch that finishes LAST reduces the per-block
// partial rows and writes what the separate ctu_bn_finalize / ctu_bn_bwd_finalize launch would have written
// (reference: nn.BatchNorm3d's batch statistics, running-stat update and backward sums; models.py:27-32).
//
// Hand-off between blocks of one launch (the per-XCD L2s are not coherent): every partial row is stored write-through
// (sc1 = agent-scope atomic store), every storing wave drains its stores (s_waitcnt vmcnt(0)), the block's barrier, then ONE
// lane draws a ticket with an agent-scope ACQ_REL fetch_add -- the release orders this block's row stores before the ticket,
// the acquire in the block that draws the LAST ticket orders its row loads (sc1, past its L1) after every other block's
// release: the happens-before chain the memory model asks for (round 2 used a relaxed ticket and relied on the waitcnt
// alone).  The counter is one zero-initialised word per layer; the last block puts it back to zero, so a replayed
// graph needs no memset node.
#pragma once
#include <hip/hip_runtime.h>
#include "ctunet_hip.h"

namespace {

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a partial row element: write-through only when this launch finalizes itself (the tail's hand-off needs it); a plain store
// otherwise, which keeps the line in the XCD's L2 for the finalize launch that follows (an sc1 store drops it)
__device__ __forceinline__ void st_row(bool write_through, float* p, float v) {
#ifdef CTU_ROW_SC1                                   /* dev A/B: the round-2 behaviour (always write-through) */
    write_through = true;
#endif
    if (write_through) st_sc1(p, v);
    else *p = v;
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Called by EVERY thread of EVERY block of the launch, after the block's last row store.  True (in all threads) in the one
// block that arrived last.
__device__ __forceinline__ bool tail_last_block(unsigned* counter, unsigned nblocks_total) {
    __shared__ int s_tail_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev + 1u == nblocks_total;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tail_last = last;
    }
    __syncthreads();
    return s_tail_last != 0;
}

// Per-channel sums of the two halves of rows[nrows][2][cp] in double: a group of S lanes shares one channel (S a power of
// two, <= 64, so the butterfly stays inside a wave), lane stripe s sums rows s, s + S, ...  fin(c, sum1, sum2) runs in one
// lane per channel c < cp.  Fixed summation order -> deterministic.
template <class F>
__device__ __forceinline__ void tail_reduce_rows(const float* rows, int nrows, int cp, int C, F&& fin) {
    const int nthr = blockDim.x;
    int S = nthr / cp;
    S = S < 1 ? 1 : (S > 64 ? 64 : S);
    S = 1 << (31 - __clz(S));
    const int cpb = nthr / S, stripe = threadIdx.x % S;
    for (int c0 = 0; c0 < cp; c0 += cpb) {
        const int c = c0 + threadIdx.x / S;
        double s1 = 0.0, s2 = 0.0;
        if (c < C) {
            const float* col = rows + c;
            int r = stripe;
            for (; r + 7 * S < nrows; r += 8 * S) {          // 16 loads in flight per lane: the tail is pure latency
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a[u] = ld_sc1(col + (size_t)(r + u * S) * 2 * cp);
                    b[u] = ld_sc1(col + (size_t)(r + u * S) * 2 * cp + cp);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
            }
            for (; r < nrows; r += S) {
                s1 += (double)ld_sc1(col + (size_t)r * 2 * cp);
                s2 += (double)ld_sc1(col + (size_t)r * 2 * cp + cp);
            }
        }
        for (int o = 1; o < S; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        if (stripe == 0 && c < cp) fin(c, s1, s2);
    }
}

// forward: scale / shift / mean / invstd (+ running statistics), exactly bn_finalize_kernel's arithmetic
__device__ __forceinline__ void bn_fwd_finalize_rows(const ctu_bn_tail& t, const float* rows, int nrows, int cp) {
    if (t.num_batches_tracked && threadIdx.x == 0 && t.n_updates > 0) t.num_batches_tracked[0] += t.n_updates;
    tail_reduce_rows(rows, nrows, cp, t.C, [&](int c, double s1, double s2) {
        if (c >= t.C) { t.scale[c] = 0.f; t.shift[c] = 0.f; t.mean[c] = 0.f; t.invstd[c] = 0.f; return; }
        const double mean = s1 / t.count;
        double var = s2 / t.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)t.eps));
        const float sc = t.gamma[c] * invstd;
        t.scale[c] = sc;
        t.shift[c] = t.beta[c] - (float)mean * sc;
        t.mean[c] = (float)mean;
        t.invstd[c] = invstd;
        if (t.running_mean && t.n_updates > 0) {
            const float unb = (float)(var * (t.count / (t.count > 1.0 ? t.count - 1.0 : 1.0)));
            float rm = t.running_mean[c], rv = t.running_var[c];
            for (int u = 0; u < t.n_updates; ++u) {
                rm = (1.f - t.momentum) * rm + t.momentum * (float)mean;
                rv = (1.f - t.momentum) * rv + t.momentum * unb;
            }
            t.running_mean[c] = rm; t.running_var[c] = rv;
        }
    });
}

// backward: dgamma / dbeta / coef (+ the replayed running-stat update), exactly bn_bwd_finalize_kernel's arithmetic
__device__ __forceinline__ void bn_bwd_finalize_rows(const ctu_bn_bwd_tail& t, const float* rows, int nrows, int cp) {
    if (t.num_batches_tracked && t.running_mean && threadIdx.x == 0) t.num_batches_tracked[0] += 1;
    tail_reduce_rows(rows, nrows, cp, t.C, [&](int c, double s1, double s2) {
        if (c >= t.C) { t.coef[c] = 0.f; t.coef[cp + c] = 0.f; t.coef[2 * cp + c] = 0.f; t.coef[3 * cp + c] = 0.f; t.coef[4 * cp + c] = 0.f; return; }
        t.dbeta[c] = (float)s1;
        t.dgamma[c] = (float)s2;
        const float is_ = t.invstd[c], k0 = t.gamma[c] * is_, k1 = (float)(s1 / t.count), k2 = (float)(s2 / t.count);
        t.coef[c] = k0;
        t.coef[cp + c] = k1;
        t.coef[2 * cp + c] = k2;
        t.coef[3 * cp + c] = -k0 * k2 * is_;                      // (bn_bwd_finalize_kernel's A, B)
        t.coef[4 * cp + c] = -k0 * (k1 - k2 * t.mean[c] * is_);
        if (t.running_mean) {
            const double istd = (double)t.invstd[c];
            double var = 1.0 / (istd * istd) - (double)t.eps;
            if (var < 0.0) var = 0.0;
            const float unb = (float)(var * (t.count / (t.count > 1.0 ? t.count - 1.0 : 1.0)));
            t.running_mean[c] = (1.f - t.momentum) * t.running_mean[c] + t.momentum * t.mean[c];
            t.running_var[c] = (1.f - t.momentum) * t.running_var[c] + t.momentum * unb;
        }
    });
}

// ticket + finalize: called by every thread of every block of the launch after the block's last (sc1) row store
__device__ __forceinline__ void bn_fwd_tail(const ctu_bn_tail& t, const float* rows, int nrows, int cp, unsigned nblocks_total) {
    if (tail_last_block(t.counter, nblocks_total)) bn_fwd_finalize_rows(t, rows, nrows, cp);
}
__device__ __forceinline__ void bn_bwd_tail(const ctu_bn_bwd_tail& t, const float* rows, int nrows, int cp, unsigned nblocks_total) {
    if (tail_last_block(t.counter, nblocks_total)) bn_bwd_finalize_rows(t, rows, nrows, cp);
}

static inline ctu_bn_tail tail_or_off(const ctu_bn_tail* t) {
    return t ? *t : ctu_bn_tail{};
}
static inline ctu_bn_bwd_tail bwd_tail_or_off(const ctu_bn_bwd_tail* t) {
    return t ? *t : ctu_bn_bwd_tail{};
}

}  // namespace
